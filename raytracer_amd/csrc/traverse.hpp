// BVH traversal for gfx950: closest hit and any hit over the reference's own
// tree (src/main.cpp:250-478), one ray per lane, traversal stack in LDS.
//
// Device layouts
//   pair record, 64 B, one per inner node, holds BOTH children so that one
//   visit is four 16-byte loads from one aligned 64-byte line:
//     q0 = (L.min.x, L.min.y, L.min.z, L.max.x)
//     q1 = (L.max.y, L.max.z, R.min.x, R.min.y)
//     q2 = (R.min.z, R.max.x, R.max.y, R.max.z)
//     q3 = (L.ref, R.ref, -, -) as integers
//   a ref is either an inner node's pair index or kLeafBit | first primitive.
//   Boxes are only ever read from the parent, so the stack holds bare refs.
//
//   primitive record, 48 B (three 16-byte loads), in BVH order:
//     triangle  p0 = (v2.xyz, flags)  p1 = (v0-v2, 0)  p2 = (v1-v2, 0)
//     sphere    p0 = (center.xyz, flags)  p1 = (radius, 0, 0, 0)
//   flags bit 0: sphere; bit 1: last primitive of its leaf.  The edge vectors
//   are the reference's own first two subtractions (src/main.cpp:252-253),
//   done once at upload in fp32, so the test sees identical operands.
#pragma once

#include "device_math.hpp"
#include "kernels.hpp"

namespace rta {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kPrimSphere = 1u;
constexpr uint32_t kPrimLast = 2u;
constexpr int kBlock = 256;            // threads per workgroup of every traversal kernel

using DeviceTree = TreeRef;    // pairs, prims, ref of the root node

struct Ray {
	V3 o, d;
};

struct HitRec {
	int32_t prim;
	float t, u, v;
};

struct WorkCount {
	uint32_t inner, tri, sph;
};

// src/main.cpp:250-279 with e0/e1 precomputed
RTA_DEV float test_triangle(V3 v2, V3 e0, V3 e1, const Ray &r, float &u_out, float &v_out)
{
	V3 pvec = cross(r.d, e1);
	float det = dot(e0, pvec);
	if (det > -.0001f && det < .0001f)
		return FLT_MAX;
	V3 tvec = r.o - v2;
	float u = dot(tvec, pvec) / det;
	if (u < 0 || u > 1)
		return FLT_MAX;
	V3 qvec = cross(tvec, e0);
	float v = dot(r.d, qvec) / det;
	if (v < 0 || u + v > 1)
		return FLT_MAX;
	float t = dot(e1, qvec) / det;
	u_out = u;
	v_out = v;
	return t > .0001f ? t : FLT_MAX;
}

// The same test without its divisions, for a traversal that tests many triangles per ray in lock-step and
// proves its answer afterwards (traverse_flat.hpp).  (In the walk over the wide tree it saves a ninth of the
// vector instructions and no time: that kernel waits for its gathers.)
//
// The reference divides three times, and two of its three exits lie behind a division.  Its four dot
// products - det, and the numerators a, b, c of u, v, t - need none, and from them every exit but one can be
// decided exactly as the reference decides it:
//
//   D = |det|, A B C = a b c with det's sign taken out (the quotients x/det and X/D are the same number).
//   * |det| < 1e-4: the reference's own comparison.
//   * u = RN(a/det) < 0: a quotient rounds to a negative number iff it is negative and does not round to
//     -0.  A < -2^-100 D gives a quotient below -2^-100: negative for certain (D < 2^125 for tame rays and
//     scenes: the bound itself is a normal number).  -2^-100 D <= A < 0 is left open.  A = -0 gives u = -0,
//     which is not below 0.
//   * u > 1: floats A > D > 0 are at least one unit in the last place of D apart, more than 2^-24 D, so the
//     quotient lies beyond the midpoint of 1 and its successor and rounds above 1; A <= D rounds to at most 1.
//   * v < 0: as u < 0.
//   * u + v > 1: u and v are within 2^-24 (relative) of A/D and B/D, their rounded sum within 2^-24 of their
//     sum, the float A + B within 2^-24 of the sum of the numerators; so A + B > D (1 + 2^-20) means a sum
//     above 1 and A + B < D (1 - 2^-20) a sum of at most 1.  In between: open.
//   * t = RN(c/det) > 1e-4: q = C * rcp(D), with the hardware's reciprocal (1 ulp), is within 2^-22 of t;
//     q > 1e-4 (1 + 2^-20) accepts, q < 1e-4 (1 - 2^-20) rejects, in between: open.
//
// So a triangle is rejected as the reference rejects it, accepted as the reference accepts it with its
// distance q known to 2^-22 (relative), or left open (a ray in a million).  The numerators come back as the
// reference computes them: exact_hit() makes the reference's t, u, v of an accepted triangle from them.
// Precondition: a tame ray on a tame scene (nothing overflows; the bound on D).
enum TriVerdict : int { kTriRejected = 0, kTriAccepted = 1, kTriOpen = 2 };

struct TriNumerators {
	float a, b, c, det;
};

// det and the numerators of u, v, t as src/main.cpp:252-270 computes them
RTA_DEV TriNumerators triangle_numerators(V3 v2, V3 e0, V3 e1, const Ray &r)
{
	TriNumerators n;
	const V3 pvec = cross(r.d, e1);
	n.det = dot(e0, pvec);
	const V3 tvec = r.o - v2;
	n.a = dot(tvec, pvec);
	const V3 qvec = cross(tvec, e0);
	n.b = dot(r.d, qvec);
	n.c = dot(e1, qvec);
	return n;
}

RTA_DEV TriVerdict classify_triangle(V3 v2, V3 e0, V3 e1, const Ray &r, float &q, TriNumerators &n)
{
	constexpr float kAcceptAbove = .0001f * (1.f + 0x1p-20f), kRejectBelow = .0001f * (1.f - 0x1p-20f);
	n = triangle_numerators(v2, e0, e1, r);
	const uint32_t sign = __float_as_uint(n.det) & 0x80000000u;
	const float D = __builtin_fabsf(n.det);
	const float A = __uint_as_float(__float_as_uint(n.a) ^ sign), B = __uint_as_float(__float_as_uint(n.b) ^ sign);
	const float C = __uint_as_float(__float_as_uint(n.c) ^ sign);
	const float s = A + B;
	q = C * __builtin_amdgcn_rcpf(D);
	const float tiny = D * -0x1p-100f;      // a numerator below this gives a quotient below -2^-100
	const bool rejected = !(D >= .0001f) || A < tiny || A > D || B < tiny || s > D * (1.f + 0x1p-20f) || !(q >= kRejectBelow);
	const bool accepted = !rejected && A >= 0.f && B >= 0.f && s < D * (1.f - 0x1p-20f) && q > kAcceptAbove;
	return rejected ? kTriRejected : accepted ? kTriAccepted : kTriOpen;
}

// t, u, v of src/main.cpp:261-270 from the numerators of an accepted triangle
RTA_DEV void exact_hit(const TriNumerators &n, float &t, float &u, float &v)
{
	u = n.a / n.det;
	v = n.b / n.det;
	t = n.c / n.det;
}

// Two distances known to 2^-22 each (or exactly) that differ by more than this factor are in the order of
// their approximations; a bound times this factor is no nearer than the distance it bounds.
constexpr float kCloser = 1.f - 0x1p-19f, kFarther = 1.f + 0x1p-19f;

// Records that every lane of a wave reads at the same place (a loop over all primitives, a tree's first node): read
// through the constant address space - nothing writes them while a kernel runs - a wave-uniform address makes these
// scalar loads, one per wave instead of one per lane.
typedef float Quad __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) Quad *ConstQuads;
typedef const __attribute__((address_space(4))) uint32_t *ConstWords;

// src/main.cpp:281-301
RTA_DEV float test_sphere(V3 center, float radius, const Ray &r)
{
	V3 oc = center - r.o;
	float h = dot(r.d, oc);
	float c = norm2(oc) - radius * radius;
	float delta = h * h - c;
	if (delta < .001f)
		return FLT_MAX;
	float sq = __builtin_sqrtf(delta);
	float t = h - sq;
	if (t < .001f) {
		t = h + sq;
		if (t < .001f)
			return FLT_MAX;
	}
	return t;
}

// One axis of src/main.cpp:303-330: the direction's SIGN BIT picks the near
// plane; both quotients are true divisions.
RTA_DEV void slab(float lo, float hi, float o, float d, bool negative, float &tmin, float &tmax)
{
	float near_plane = negative ? hi : lo;
	float far_plane = negative ? lo : hi;
	float t1 = (near_plane - o) / d;
	float t2 = (far_plane - o) / d;
	tmin = greater(t1, tmin);
	tmax = lesser(t2, tmax);
}

// The same quotient from a precomputed reciprocal r = RN(1/d): multiply, then
// two residual corrections with fused multiply-adds.  The first correction
// leaves a faithful quotient, and a faithful quotient corrected once more with
// the correctly rounded reciprocal IS the correctly rounded quotient
// (Markstein's theorem), provided nothing under- or overflows on the way.
// rays_are_tame() establishes that proviso per ray; the quotient may differ
// from n/d in the sign of a zero only, which no comparison below can see.
RTA_DEV float quotient(float n, float d, float r)
{
	float q = n * r;
	float e = __builtin_fmaf(-d, q, n);
	q = __builtin_fmaf(e, r, q);
	e = __builtin_fmaf(-d, q, n);
	return __builtin_fmaf(e, r, q);
}

// Entry and exit distance of one axis for a tame ray.  With finite quotients
// the sign-bit selection of the reference equals taking the smaller quotient as
// entry and the larger as exit (division by a positive d keeps the order of the
// planes, by a negative d reverses it, and rounding is monotonic), so the planes
// need no per-lane select.
RTA_DEV void slab_tame(float lo, float hi, float o, float d, float r, float &t_in, float &t_out)
{
	const float a = quotient(lo - o, d, r);
	const float b = quotient(hi - o, d, r);
	t_in = __builtin_fminf(a, b);
	t_out = __builtin_fmaxf(a, b);
}

struct RaySigns {
	bool x, y, z;
};

RTA_DEV RaySigns signs_of(const Ray &r)
{
	return RaySigns{(__float_as_uint(r.d.x) >> 31) != 0, (__float_as_uint(r.d.y) >> 31) != 0,
	                (__float_as_uint(r.d.z) >> 31) != 0};
}

// True when every slab quotient of this ray can take the reciprocal route: each
// direction component has magnitude in [2^-40, 2^40] and each origin component
// is zero or has magnitude in [2^-50, 2^40].  Together with the scene-side
// condition checked at upload (every box coordinate is zero or has magnitude in
// [2^-78, 2^40]) a numerator plane - o is then zero or at least 2^-78 in
// magnitude, and quotient() stays clear of subnormals and overflow.
RTA_DEV bool component_tame(float d, float o)
{
	const float ad = __builtin_fabsf(d), ao = __builtin_fabsf(o);
	return ad >= 0x1p-40f && ad <= 0x1p40f && (ao == 0.f || (ao >= 0x1p-50f && ao <= 0x1p40f));
}

RTA_DEV bool ray_is_tame(const Ray &r)
{
	return component_tame(r.d.x, r.o.x) && component_tame(r.d.y, r.o.y) && component_tame(r.d.z, r.o.z);
}

RTA_DEV float test_box(V3 lo, V3 hi, const Ray &r, RaySigns s, float max_t)
{
	float tmin = 0.f, tmax = FLT_MAX;
	slab(lo.x, hi.x, r.o.x, r.d.x, s.x, tmin, tmax);
	slab(lo.y, hi.y, r.o.y, r.d.y, s.y, tmin, tmax);
	slab(lo.z, hi.z, r.o.z, r.d.z, s.z, tmin, tmax);
	return (tmax >= tmin && tmin < max_t && tmax > 0) ? tmin : FLT_MAX;
}

RTA_DEV float test_box_tame(V3 lo, V3 hi, const Ray &r, V3 rcp, float max_t)
{
	float ix, ox, iy, oy, iz, oz;
	slab_tame(lo.x, hi.x, r.o.x, r.d.x, rcp.x, ix, ox);
	slab_tame(lo.y, hi.y, r.o.y, r.d.y, rcp.y, iy, oy);
	slab_tame(lo.z, hi.z, r.o.z, r.d.z, rcp.z, iz, oz);
	// the reference folds max(t, tmin) from 0 and min(t, tmax) from FLT_MAX
	// (src/main.cpp:305-327); without NaNs the fold order does not matter
	const float tmin = __builtin_fmaxf(__builtin_fmaxf(ix, iy), __builtin_fmaxf(iz, 0.f));
	const float tmax = __builtin_fminf(__builtin_fminf(ox, oy), __builtin_fminf(oz, FLT_MAX));
	return (tmax >= tmin && tmin < max_t && tmax > 0) ? tmin : FLT_MAX;
}

// Entry and exit distance of a box from plain products with the reciprocal.
// Each product t' = RN(RN(plane - o) * RN(1/d)) is within a factor 1 +- 3*2^-24
// of the reference's quotient t = RN(RN(plane - o) / d) and has the same sign
// (tame rays and scenes: nothing under- or overflows), and min/max are
// monotonic, so the box's true entry distance lies within the same factor of
// t_in and, where t_out > 0, the true exit distance within it of t_out; t_in = 0
// means the true entry distance is 0, t_out <= 0 that the true one is <= 0.
// The traversal only ever COMPARES these distances (hit or miss, which child is
// nearer), so it decides from the approximations whenever they are further apart
// than the error and computes the quotients only for the rare close calls.
RTA_DEV void approx_box(V3 lo, V3 hi, const Ray &r, V3 rcp, float &t_in, float &t_out)
{
	const float ax = (lo.x - r.o.x) * rcp.x, bx = (hi.x - r.o.x) * rcp.x;
	const float ay = (lo.y - r.o.y) * rcp.y, by = (hi.y - r.o.y) * rcp.y;
	const float az = (lo.z - r.o.z) * rcp.z, bz = (hi.z - r.o.z) * rcp.z;
	const float ix = __builtin_fminf(ax, bx), iy = __builtin_fminf(ay, by), iz = __builtin_fminf(az, bz);
	const float ox = __builtin_fmaxf(ax, bx), oy = __builtin_fmaxf(ay, by), oz = __builtin_fmaxf(az, bz);
	t_in = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(ix, iy), iz), 0.f);
	t_out = __builtin_fminf(__builtin_fminf(ox, oy), oz);
}

// a > b * kApart (or a * kApart < b) proves a > b for the true distances:
// 1 + 2^-19 covers (1 + 3*2^-24) / (1 - 3*2^-24) and the rounding of the product
constexpr float kApart = 1.f + 0x1p-19f;

// Traversal of src/main.cpp:332-422 (closest, ANY = false) and :424-478 (any
// hit, ANY = true), kept as an explicit per-lane state machine: step() does one
// unit of work, either one inner-node visit (both child boxes) or one primitive
// test, and leaves in `node` what the lane does next.
//
// `stack` points at the lane's column of an LDS array laid out [depth][STRIDE],
// so consecutive lanes hit consecutive banks; upload guarantees the tree never
// needs more than `depth` entries.  For ANY, `limit` is the distance bound and
// the answer is prim >= 0.
//
// TAME_ONLY promises that every ray given to start() passes ray_is_tame() on a
// tame tree; the kernel then carries no code for the division-based box test.
template <bool ANY, bool COUNT, bool TAME_ONLY = false, int STRIDE = kBlock>
struct Traversal {
	Ray ray;
	V3 rcp;              // correctly rounded 1/direction (only used when tame)
	RaySigns sg;
	bool tame;
	bool more;           // false once the answer is final
	float best;
	int32_t prim;
	float u, v;
	uint32_t node;
	int sp;

	RTA_DEV void start(const DeviceTree &tree, const Ray &r, float limit)
	{
		ray = r;
		sg = signs_of(r);
		tame = TAME_ONLY || (tree.tame != 0 && ray_is_tame(r));
		rcp = mk(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
		best = ANY ? limit : FLT_MAX;
		prim = -1;
		u = v = 0.f;
		node = tree.root;
		sp = 0;
		more = true;
	}

	// A lane without a ray: more is false and result() says "no hit".
	RTA_DEV void rest()
	{
		ray = Ray{mk(0, 0, 0), mk(1, 1, 1)};
		rcp = mk(1, 1, 1);
		sg = RaySigns{false, false, false};
		tame = true;
		best = FLT_MAX;
		prim = -1;
		u = v = 0.f;
		node = 0;
		sp = 0;
		more = false;
	}

	RTA_DEV bool at_prim() const { return (node & kLeafBit) != 0; }

	// One unit of work: either one inner-node visit or one primitive test.
	// Precondition: more.  The record's first 48 bytes are loaded by the same three
	// instructions whichever kind the lane's node is: a load instruction occupies
	// the memory pipeline for the whole wave however few lanes take part, so a trip
	// in which some lanes visit nodes and others test primitives issues four loads
	// instead of seven.
	RTA_DEV void step(const DeviceTree &tree, uint32_t *stack, const int depth, WorkCount &work)
	{
		const bool leaf = at_prim();
		const float4 *rec = leaf ? tree.prims + kPrimQuads * (size_t)(node & ~kLeafBit) : tree.pairs + 4 * (size_t)node;
		const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
		// keep the three loads whole (a primitive does not use q1.w and q2.w; loading
		// them apart for nodes would cost a fifth instruction)
		asm volatile("" ::"v"(q1.w), "v"(q2.w));
		if (leaf)
			prim_unit(q0, q1, q2, stack, work);
		else
			inner_unit(q0, q1, q2, rec[3], stack, depth, work);
	}

	// Precondition: more && at_prim()
	RTA_DEV void test_prim(const DeviceTree &tree, uint32_t *stack, WorkCount &work)
	{
		const float4 *rec = tree.prims + kPrimQuads * (size_t)(node & ~kLeafBit);
		prim_unit(rec[0], rec[1], rec[2], stack, work);
	}

	// Precondition: more && !at_prim()
	RTA_DEV void visit_inner(const DeviceTree &tree, uint32_t *stack, const int depth, WorkCount &work)
	{
		const float4 *rec = tree.pairs + 4 * (size_t)node;
		inner_unit(rec[0], rec[1], rec[2], rec[3], stack, depth, work);
	}

	// p0..p2: the primitive's record
	RTA_DEV void prim_unit(const float4 p0, const float4 p1, const float4 p2, uint32_t *stack, WorkCount &work)
	{
		{
			const uint32_t i = node & ~kLeafBit;
			const uint32_t flags = __float_as_uint(p0.w);
			float t, tu = 0, tv = 0;
			if (flags & kPrimSphere) {
				if (COUNT) work.sph++;
				t = test_sphere(mk(p0.x, p0.y, p0.z), p1.x, ray);
			} else {
				if (COUNT) work.tri++;
				t = test_triangle(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), ray, tu, tv);
			}
			if (t < best) {           // strict: the first-visited primitive keeps a tie
				prim = (int32_t)i;
				if (ANY) {
					more = false;
					return;
				}
				best = t;
				u = tu;
				v = tv;
			}
			if (!(flags & kPrimLast))
				node = node + 1;      // next primitive of the same leaf
			else if (sp > 0)
				node = stack[(--sp) * STRIDE];
			else
				more = false;
		}
	}

	// q0..q3: the pair record of the node
	RTA_DEV void inner_unit(const float4 q0, const float4 q1, const float4 q2, const float4 q3, uint32_t *stack,
	                        const int depth, WorkCount &work)
	{
		{
			if (COUNT) work.inner++;
			const uint32_t ref1 = __float_as_uint(q3.x), ref2 = __float_as_uint(q3.y);
			const V3 lo1 = mk(q0.x, q0.y, q0.z), hi1 = mk(q0.w, q1.x, q1.y);
			const V3 lo2 = mk(q1.z, q1.w, q2.x), hi2 = mk(q2.y, q2.z, q2.w);
			bool hit1, hit2, one_near;      // one_near: the first child is visited first
			bool exact = !(TAME_ONLY || tame);
			if (!exact) {
				float i1, o1, i2, o2;
				approx_box(lo1, hi1, ray, rcp, i1, o1);
				approx_box(lo2, hi2, ray, rcp, i2, o2);
				const float far = best * kApart;      // may be +inf
				const float i1k = i1 * kApart, o1k = o1 * kApart, i2k = i2 * kApart, o2k = o2 * kApart;
				// the three conditions of the reference's box test (src/main.cpp:329),
				// each either proven false (miss) or all proven true (hit)
				const bool miss1 = o1 <= 0.f || o1k < i1 || i1 >= far;
				const bool miss2 = o2 <= 0.f || o2k < i2 || i2 >= far;
				hit1 = o1 > 0.f && o1 >= i1k && i1k < best;
				hit2 = o2 > 0.f && o2 >= i2k && i2k < best;
				exact = (!miss1 && !hit1) || (!miss2 && !hit2);
				if (ANY) {
					one_near = true;
				} else {
					// src/main.cpp:391-401 compares the boxes' distances, FLT_MAX for a miss
					const bool nearer = i1k < i2, not_nearer = i1 >= i2k;
					one_near = hit1 && (!hit2 || nearer);
					exact = exact || (hit1 && hit2 && !nearer && !not_nearer);
				}
			}
			if (exact) {
				float d1, d2;
				if (TAME_ONLY || tame) {
					d1 = test_box_tame(lo1, hi1, ray, rcp, best);
					d2 = test_box_tame(lo2, hi2, ray, rcp, best);
				} else {
					d1 = test_box(lo1, hi1, ray, sg, best);
					d2 = test_box(lo2, hi2, ray, sg, best);
				}
				hit1 = d1 < FLT_MAX;
				hit2 = d2 < FLT_MAX;
				one_near = ANY || d1 < d2;
			}
			// closest hit: nearer child first, and on equal distances the SECOND
			// child is "near" (src/main.cpp:391-401); any hit: first child first,
			// no ordering (src/main.cpp:460-473)
			const uint32_t first = one_near ? ref1 : ref2, second = one_near ? ref2 : ref1;
			const bool hit_first = one_near ? hit1 : hit2, hit_second = one_near ? hit2 : hit1;
			if (hit_first) {
				node = first;
				if (hit_second && sp < depth) {      // sp == depth cannot happen on a validated tree
					stack[sp * STRIDE] = second;
					sp++;
				}
			} else if (hit_second) {
				node = second;
			} else if (sp > 0) {
				node = stack[(--sp) * STRIDE];
			} else {
				more = false;
			}
		}
	}

	RTA_DEV HitRec result() const { return HitRec{prim, ANY ? 0.f : best, u, v}; }
};

// One ray per lane from start to finish.  Each trip of the loop gives every
// unfinished lane exactly one unit of work (an inner-node visit or a primitive
// test).  The loop condition is a wave-wide ballot: a convergent operation the
// compiler may not duplicate into the two branches, which keeps this ONE loop
// instead of a nest of "walk to a leaf" and "test the leaf" loops in which lanes
// wait for the slowest lane of each inner loop (measured: 23 % instead of 16 % of
// the lanes active per vector instruction on bounce rays).
template <bool ANY, bool COUNT, bool TAME_ONLY = false>
RTA_DEV void traverse_dyn(const DeviceTree &tree, const Ray &ray, float limit, uint32_t *stack, const int depth,
                          HitRec &hit, WorkCount &work)
{
	Traversal<ANY, COUNT, TAME_ONLY> tr;
	tr.start(tree, ray, limit);
	while (__ballot(tr.more)) {
		if (tr.more)
			tr.step(tree, stack, depth, work);
	}
	hit = tr.result();
}

// The same traversal as two nested loops: walk inner nodes down to a leaf, then
// test the leaf's primitives one after the other.  On trees with a handful of
// long leaves (cornell_box.txt: 23 primitives in 2 leaves) the lanes of a wave
// stay in step anyway, and the inner loops do without the per-trip choice
// between the two kinds of work (1.17-1.26 times the rays per second there; on
// dragon.txt, whose leaves hold 1.7 primitives, 0.7 times).
template <bool ANY, bool COUNT, bool TAME_ONLY = false>
RTA_DEV void traverse_leafwise(const DeviceTree &tree, const Ray &ray, float limit, uint32_t *stack, const int depth,
                               HitRec &hit, WorkCount &work)
{
	Traversal<ANY, COUNT, TAME_ONLY> tr;
	tr.start(tree, ray, limit);
	while (tr.more) {
		while (tr.more && !tr.at_prim())
			tr.visit_inner(tree, stack, depth, work);
		while (tr.more && tr.at_prim())
			tr.test_prim(tree, stack, work);
	}
	hit = tr.result();
}

// A wave works through more rays than it has lanes: rays differ widely in
// length (bounce rays on the dragon: 41 node visits and primitive tests on
// average, 95 at the 99th percentile), so a wave that took 64 of them and ran
// until the last was through would keep 43 % of its lanes busy.  Here, whenever
// `refill_at` lanes have run out of work, the wave stores their results and
// hands them new rays:
//   got = grab(want, first, dry)  reserves up to `want` rays, numbered first..first+got-1, and says
//                                 whether the source has run dry (wave-uniform; got < want alone
//                                 does not say so)
//   tag = load(i, ray, limit)   fetches ray number i; the tag comes back in
//   store(tag, hit)             once the ray is through
template <bool ANY, bool COUNT, bool TAME_ONLY, class Grab, class Load, class Store>
RTA_DEV void traverse_pulling(const DeviceTree &tree, const uint32_t refill_at, uint32_t *stack, const int depth,
                              WorkCount &work, Grab grab, Load load, Store store)
{
	Traversal<ANY, COUNT, TAME_ONLY> tr;
	tr.rest();
	uint32_t tag = 0;
	bool loaded = false, dry = false;
	for (;;) {
		const unsigned long long idle = __ballot(!tr.more);
		const uint32_t n_idle = (uint32_t)__popcll(idle);
		if (!dry && n_idle >= refill_at) {
			if (!tr.more && loaded) {
				store(tag, tr.result());
				loaded = false;
			}
			uint32_t first = 0;
			const uint32_t got = grab(n_idle, first, dry);
			const uint32_t rank =
			    __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
			if (!tr.more && rank < got) {
				Ray ray;
				float limit;
				tag = load(first + rank, ray, limit);
				tr.start(tree, ray, limit);
				loaded = true;
			}
		}
		if (__ballot(tr.more) == 0) {
			if (dry)
				break;
			continue;        // every lane idle and the source not known to be dry: the next trip refills
		}
		if (tr.more)
			tr.step(tree, stack, depth, work);
	}
	if (loaded)
		store(tag, tr.result());
}

// grab() for a wave that owns rays 0..count-1 outright
struct OwnChunk {
	uint32_t count, next;
	RTA_DEV uint32_t operator()(uint32_t want, uint32_t &first, bool &dry)
	{
		const uint32_t got = count - next < want ? count - next : want;
		first = next;
		next += got;
		dry = next == count;
		return got;
	}
};

// grab() for waves that share a list of `total` rays through a counter in memory.
// A wave reserves `chunk` rays at a time and hands them to its lanes as they fall
// idle: an atomic with a return value is a round trip to the memory side that the
// whole wave sits out, so one per refill (16-24 rays) costs a refilling wave more
// than its traversal (measured: 7.2 -> 3.8 Grays/s on dragon bounce rays).
struct SharedList {
	uint32_t total;
	uint32_t *head;
	uint32_t chunk;
	uint32_t next = 0, end = 0;      // what is left of the wave's reservation
	RTA_DEV uint32_t operator()(uint32_t want, uint32_t &first, bool &dry)
	{
		if (next == end) {
			uint32_t base = 0;
			if ((threadIdx.x & 63) == 0)
				base = atomicAdd(head, chunk);
			base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
			next = base < total ? base : total;
			end = total - next < chunk ? total : next + chunk;
		}
		const uint32_t got = end - next < want ? end - next : want;
		first = next;
		next += got;
		dry = got == 0;
		return got;
	}
};

}  // namespace rta
