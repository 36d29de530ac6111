// Closest hit and any hit for scenes of a few dozen primitives, without a tree: every lane tests every
// primitive, the wave in lock-step, the records read once per wave (the loop counter is wave-uniform).
//
// The reference's triangle test (src/main.cpp:250-279) divides three times, and two of its three exits lie
// behind a division.  Its four dot products - det, and the numerators a, b, c of u, v, t - need none, and
// from them every exit but one can be decided exactly as the reference decides it:
//
//   D = |det|, A B C = a b c with det's sign taken out (the quotients x/det and X/D are the same number).
//   * |det| < 1e-4: the reference's own comparison.
//   * u = RN(a/det) < 0: a quotient rounds to a negative number iff it is negative and does not round to
//     -0.  D < 2^125 for tame rays and scenes, so A <= -2^-24 gives a quotient <= -2^-149: negative for
//     certain.  -2^-24 < A < 0 is left open ("vague").  A = -0 gives u = -0, which is not below 0.
//   * u > 1: floats A > D > 0 are at least one unit in the last place of D apart, more than 2^-24 D, so the
//     quotient lies beyond the midpoint of 1 and its successor and rounds above 1; A <= D rounds to at most 1.
//   * v < 0: as u < 0.
//   * u + v > 1: u and v are within 2^-24 (relative) of A/D and B/D, their rounded sum within 2^-24 of their
//     sum, the float A + B within 2^-24 of the sum of the numerators; so A + B > D (1 + 2^-20) means a sum
//     above 1 and A + B < D (1 - 2^-20) a sum of at most 1.  In between: vague.
//   * t = RN(c/det) > 1e-4: q = C * rcp(D), with the hardware's reciprocal (1 ulp), is within 2^-22 of t;
//     q > 1e-4 (1 + 2^-20) accepts, q < 1e-4 (1 - 2^-20) rejects, in between: vague.
//
// So a triangle is rejected for certain, accepted for certain with its distance known to 2^-22, or vague.
// The nearest of the accepted ones is found by comparing the q with a margin of 2^-19 (anything nearer than
// that to the best so far makes the ray vague), and ITS u, v, t are then computed with the reference's
// three divisions.  A vague ray is "in doubt": the caller has the exact traversal answer it, as for the
// certified walk (traverse_wide.hpp), whose argument carries over: every primitive is tested here, so the
// winner w with distance F is the only primitive the reference's test accepts that near (step 1; no tie:
// the others are at least a factor 1 + 2^-20 farther); the reference's running minimum never drops below F
// before it tests w (step 2); and it does test w if the box of w's leaf passes its test with tmin < F
// (step 3), which is evaluated here with the reference's arithmetic.
//
// ANY: the occlusion query (src/main.cpp:424-478).  A primitive accepted for certain with q < limit (1 -
// 2^-20) whose leaf's box passes the reference's test against `limit` is an occluder; if no primitive is
// accepted or vague before limit (1 + 2^-20), there is none.
#pragma once

#include "traverse.hpp"

namespace rta {

typedef float Quad __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) Quad *ConstQuads;

// Precondition: ray_is_tame(ray) on a tame tree for every active lane (`active` lanes only are answered; the
// others run along).  hit: the reference's answer unless doubt.
template <bool ANY, bool COUNT>
RTA_DEV void scan_flat(const DeviceTree &tree, const bool active, const Ray &ray, const float limit, HitRec &hit, bool &doubt,
                       WorkCount &work)
{
	constexpr float kTiny = 0x1p-24f;
	constexpr float kAcceptAbove = .0001f * (1.f + 0x1p-20f), kRejectBelow = .0001f * (1.f - 0x1p-20f);
	// the nearest accepted primitive so far: its distance to 2^-22, its numerators and determinant
	float best_q = ANY ? limit : FLT_MAX;
	float wa = 0.f, wb = 0.f, wc = 0.f, wd = 1.f;
	int32_t prim = -1;
	bool vague = false;
	const uint32_t n = tree.flat_prims;
	for (uint32_t p = 0; p < n; p++) {
		// (the records do not change while a kernel runs: read through the constant address space, a wave-uniform
		// address makes these scalar loads, and the record's words scalar operands)
		const ConstQuads rec = (ConstQuads)tree.prims + kPrimQuads * (size_t)p;
		const Quad p0 = rec[0], p1 = rec[1], p2 = rec[2];
		const uint32_t flags = __float_as_uint(p0.w);
		float q, na, nb, nc, nd;
		bool accepted, open;      // open: neither accepted nor rejected for certain
		if (flags & kPrimSphere) {
			if (COUNT) work.sph++;
			// the reference's own arithmetic: the distance is exact
			q = test_sphere(mk(p0.x, p0.y, p0.z), p1.x, ray);
			accepted = q < FLT_MAX;
			open = false;
			na = nb = 0.f;
			nc = q;
			nd = 1.f;
		} else {
			if (COUNT) work.tri++;
			const V3 v2 = mk(p0.x, p0.y, p0.z), e0 = mk(p1.x, p1.y, p1.z), e1 = mk(p2.x, p2.y, p2.z);
			const V3 pvec = cross(ray.d, e1);
			const float det = dot(e0, pvec);
			const V3 tvec = ray.o - v2;
			na = dot(tvec, pvec);
			const V3 qvec = cross(tvec, e0);
			nb = dot(ray.d, qvec);
			nc = dot(e1, qvec);
			nd = det;
			const uint32_t sign = __float_as_uint(det) & 0x80000000u;
			const float D = __builtin_fabsf(det);
			const float A = __uint_as_float(__float_as_uint(na) ^ sign), B = __uint_as_float(__float_as_uint(nb) ^ sign);
			const float C = __uint_as_float(__float_as_uint(nc) ^ sign);
			const float s = A + B;
			q = C * __builtin_amdgcn_rcpf(D);
			const bool rejected = !(D >= .0001f) || A <= -kTiny || A > D || B <= -kTiny || s > D * (1.f + 0x1p-20f) || !(q >= kRejectBelow);
			accepted = !rejected && A >= 0.f && B >= 0.f && s < D * (1.f - 0x1p-20f) && q > kAcceptAbove;
			open = !rejected && !accepted;
		}
		if (ANY) {
			// (best_q stays at the limit)
			const bool before = q < best_q * (1.f - 0x1p-20f), beyond = q > best_q * (1.f + 0x1p-20f);
			if (accepted && before && prim < 0)
				prim = (int32_t)p;
			vague = vague || ((accepted || open) && !beyond && !(accepted && before));
			if (!__ballot(active && prim < 0))
				break;
		} else {
			const bool nearer = q < best_q * (1.f - 0x1p-19f), farther = q > best_q * (1.f + 0x1p-19f);
			vague = vague || (open && !farther) || (accepted && !nearer && !farther);
			if (accepted && nearer) {
				best_q = q;
				prim = (int32_t)p;
				wa = na;
				wb = nb;
				wc = nc;
				wd = nd;
			}
		}
	}
	if (ANY && prim >= 0)
		vague = false;      // an occluder for certain, whatever else is unclear
	// the winner's u, v, t as the reference computes them (src/main.cpp:261-270)
	float best = ANY ? limit : FLT_MAX, u = 0.f, v = 0.f;
	if (!ANY && prim >= 0) {
		u = wa / wd;
		v = wb / wd;
		best = wc / wd;
	}
	doubt = active && vague;
	if (active && !vague && prim >= 0) {
		// step 3: the reference reaches the primitive's leaf
		const V3 rcp = mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
		const float4 b0 = tree.leaf_boxes[2 * (size_t)prim], b1 = tree.leaf_boxes[2 * (size_t)prim + 1];
		const float d = test_box_tame(mk(b0.x, b0.y, b0.z), mk(b1.x, b1.y, b1.z), ray, rcp, best);
		doubt = !(d < FLT_MAX);
	}
	hit = HitRec{prim, ANY ? 0.f : best, u, v};
}

}  // namespace rta
