// Closest hit and any hit for scenes of a few dozen primitives, without a tree: every lane tests every
// primitive, the wave in lock-step, the records read once per wave (the loop counter is wave-uniform).
//
// Triangles are tested without divisions (classify_triangle, traverse.hpp): rejected or accepted exactly as the
// reference's test (src/main.cpp:250-279) does, with an accepted one's distance q known to 2^-22, or left open.
// The nearest of the accepted ones is found by comparing the q with a margin of 2^-19 (anything nearer than
// that to the best so far, or open and not farther, makes the ray vague), and ITS u, v, t are then computed with
// the reference's three divisions.  A vague ray is "in doubt": the caller has the exact traversal answer it, as
// for the certified walk (traverse_wide.hpp), whose argument carries over: every primitive is tested here, so
// the winner w with distance F is the only primitive the reference's test accepts that near (step 1; no tie:
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

// Precondition: ray_is_tame(ray) on a tame tree for every active lane (`active` lanes only are answered; the
// others run along).  hit: the reference's answer unless doubt.
template <bool ANY, bool COUNT>
RTA_DEV void scan_flat(const DeviceTree &tree, const bool active, const Ray &ray, const float limit, HitRec &hit, bool &doubt,
                       WorkCount &work)
{
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
			TriNumerators num;
			const TriVerdict verdict = classify_triangle(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), ray, q, num);
			accepted = verdict == kTriAccepted;
			open = verdict == kTriOpen;
			na = num.a;
			nb = num.b;
			nc = num.c;
			nd = num.det;
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
			const bool nearer = q < best_q * kCloser, farther = q > best_q * kFarther;
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
	if (!ANY && prim >= 0)
		exact_hit(TriNumerators{wa, wb, wc, wd}, best, u, v);      // (a sphere: its exact distance over 1)
	doubt = active && vague;
	if (active && !vague && prim >= 0) {
		// step 3: the reference reaches the primitive's leaf
		const V3 rcp = mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
		const float4 b0 = tree.leaf_boxes[2 * (size_t)prim], b1 = tree.leaf_boxes[2 * (size_t)prim + 1];
		const float d = test_box_tame(mk(b0.x, b0.y, b0.z), mk(b1.x, b1.y, b1.z), ray, rcp, best);
		// (a primitive that no leaf holds - the reference never tests it - has no box: the exact traversal's)
		doubt = !(d < FLT_MAX) || b0.w != 1.f;
	}
	hit = HitRec{prim, ANY ? 0.f : best, u, v};
}

}  // namespace rta
