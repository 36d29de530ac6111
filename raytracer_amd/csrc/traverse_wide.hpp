// Certified closest hit over the 4-wide tree (wide_tree.hpp).
//
// The reference's NearestHit (src/main.cpp:332-422) defines the result: the
// nearest primitive among those it tests, the first-tested one on equal
// distances.  What it tests depends on its visit order and on the rounding of
// its box tests, so a traversal in another order over another tree cannot in
// general be shown to return the same primitive - but for almost every ray it
// can be shown AFTER the fact.  The traversal below walks the wide tree in any
// order with approximate, conservative box tests, tests triangles with the
// reference's own arithmetic (test_triangle), and ends with a candidate (w, F).
// Let P be the set of triangles that HitTriangle accepts for this ray.
//
//   1. Conservative boxes (see wide_tree.cpp): every p in P with t_p <= F has
//      all its wide-tree boxes entered before best (1 + 2^-20) >= F (1 + 2^-20)
//      and left after 0, so the walk tests it.  Hence if the walk saw exactly
//      one triangle with t <= F (no tie), w is the only member of P that near.
//   2. The reference tests a subset of P's candidates, all others farther than
//      F, so its running minimum never drops below F before it tests w.
//   3. It does test w if the box of w's leaf passes ITS test with tmin < F
//      (evaluated here with the reference's arithmetic, test_box_tame): boxes of
//      the reference tree are nested and the slab quotients are monotonic in
//      the planes, so every ancestor of the leaf passes with tmin <= the
//      leaf's < F <= the running minimum (the builder checks the nesting).
//
// 1-3 give NearestHit = (w, F) with w's own t, u, v.  A miss of the walk means
// P is empty.  Rays for which a step fails (a tie, the leaf's box not passed
// with tmin < F, rays outside the range the padding was computed for, no room
// left in the lane's LDS column) are "in doubt" and are traced again by the
// exact traversal (traverse.hpp); nothing is ever answered from an unproven
// candidate.
//
// Because the order is free, a lane that reaches a triangle does not test it at
// once: it notes it in a list and goes on with the next node, and the wave runs
// the (long) triangle test only when enough lanes have one waiting.  The list
// shares the lane's LDS column with the node stack: stack entries grow from the
// bottom, list entries from the top.
#pragma once

#include "traverse.hpp"

namespace rta {

// REL: the tree's entries carry entry-relative distance bits (DeviceTree::wide_relative).
//
// ANY: the occlusion query of src/main.cpp:424-478 instead - is any primitive hit before `limit`?
// The reference answers yes as soon as a leaf it reaches holds a primitive with t < limit, and it
// reaches a leaf when the leaf's box and all boxes above pass its test against `limit`; the order
// plays no part.  The walk's "no" is therefore always the reference's (by step 1 it tested every
// triangle that is accepted before `limit`), and its "yes" is once the hit triangle's leaf box
// passes the reference's test (step 3).  `best` stays at `limit` throughout.
template <bool COUNT, bool REL, bool ANY = false, int STRIDE = kBlock>
struct WideTraversal {
	Ray ray;
	V3 rcp, off;         // correctly rounded 1/direction; -(origin - scene centre) * rcp
	uint32_t negative;   // bit k: direction component k is negative
	float best;
	int32_t prim;
	float u, v;
	uint32_t tie;        // another triangle at exactly `best`
	uint32_t doubt;      // the exact traversal must answer this ray
	uint32_t node;       // wide node to visit next, or kWideNone
	int sp, lp;          // stack entries [0, sp), list entries [lp, cap)
	float t_enter;       // where the ray enters the scene's box (DeviceTree::wide_relative), else 0

	RTA_DEV void rest(const int cap)
	{
		ray = Ray{mk(0, 0, 0), mk(1, 1, 1)};
		rcp = mk(1, 1, 1);
		off = mk(0, 0, 0);
		negative = 0;
		best = FLT_MAX;
		prim = -1;
		u = v = 0.f;
		tie = doubt = false;
		node = kWideNone;
		sp = 0;
		lp = cap;
		t_enter = 0.f;
	}

	RTA_DEV void start(const DeviceTree &tree, const Ray &r, const int cap, const float limit = FLT_MAX)
	{
		ray = r;
		rcp = mk(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
		const V3 from = r.o - mk(tree.wide_center[0], tree.wide_center[1], tree.wide_center[2]);
		off = mk(-(from.x * rcp.x), -(from.y * rcp.y), -(from.z * rcp.z));
		negative = (__float_as_uint(r.d.x) >> 31) | ((__float_as_uint(r.d.y) >> 31) << 1) | ((__float_as_uint(r.d.z) >> 31) << 2);
		best = ANY ? limit : FLT_MAX;
		prim = -1;
		u = v = 0.f;
		tie = false;
		sp = 0;
		lp = cap;
		// the rays the wide tree's padding covers; NaNs fail the comparisons
		const bool admitted =
		    ray_is_tame(r) && norm2(r.d) <= kWideMaxDir * kWideMaxDir * (1.f - 0x1p-20f) && norm2(from) <= tree.wide_reach2;
		doubt = !admitted;
		node = admitted ? 0u : kWideNone;
		// Spheres are outside the tree: each is tested here, with the reference's own arithmetic, before the
		// walk begins (step 1 of the proof holds for them trivially: every one is tested).
		if (admitted)
			for (uint32_t k = 0; k < tree.n_loose; k++) {
				// (the same sphere for every lane: scalar loads)
				const uint32_t i = ((ConstWords)tree.loose)[k];
				const ConstQuads rec = (ConstQuads)tree.prims + kPrimQuads * (size_t)i;
				const Quad p0 = rec[0], p1 = rec[1];
				const float t = test_sphere(mk(p0.x, p0.y, p0.z), p1.x, r);
				if (t < best) {
					prim = (int32_t)i;
					if (ANY) {
						node = kWideNone;      // an occluder: nothing left to walk
					} else {
						best = t;
						tie = false;
					}
				} else if (!ANY && t == best && t < FLT_MAX) {
					tie = true;
				}
			}
		t_enter = 0.f;
		if (REL) {
			// (an estimate is enough: distance_bits() is monotonic whatever t_enter is)
			const float ax = (tree.box_lo[0] - r.o.x) * rcp.x, bx = (tree.box_hi[0] - r.o.x) * rcp.x;
			const float ay = (tree.box_lo[1] - r.o.y) * rcp.y, by = (tree.box_hi[1] - r.o.y) * rcp.y;
			const float az = (tree.box_lo[2] - r.o.z) * rcp.z, bz = (tree.box_hi[2] - r.o.z) * rcp.z;
			t_enter = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)),
			                          __builtin_fmaxf(__builtin_fminf(az, bz), 0.f));
			t_enter = admitted && t_enter < FLT_MAX ? t_enter : 0.f;
		}
	}

	// The distance part of a key or entry: monotonic in t (t >= 0 or +inf), so that comparing two of them
	// never contradicts comparing the distances.  Scenes with few primitives leave room for the leading
	// bits of the float itself.  Scenes with millions leave 7 or 8 bits, less than a float's exponent: there
	// the distance counts from where the ray enters the scene's box (rays from outside have all their
	// boxes at much the same distance from the origin), as a short float of 4 exponent bits - 16 octaves
	// down from the scene's size - and what remains of mantissa.
	RTA_DEV uint32_t distance_bits(const DeviceTree &tree, float t, uint32_t dmask) const
	{
		if (REL) {
			const uint32_t x = __float_as_uint(__builtin_fmaxf(t - t_enter, 0.f));
			const uint32_t k = (x > tree.wide_key_base ? x : tree.wide_key_base) - tree.wide_key_base;
			return ((k < 0x07ffffffu ? k : 0x07ffffffu) << 4) & dmask;
		}
		return __float_as_uint(t) & dmask;
	}

	// A triangle the ray is likely to hit (what the last ray through the same pixel hit, say), listed
	// before the walk begins: tested first, it bounds the walk from its first node on.  Any triangle
	// may be tested at any time, so the proof does not care where the hint comes from.
	RTA_DEV void take_hint(uint32_t *stack, const int32_t hint)
	{
		if (hint >= 0 && !doubt) {
			lp--;
			stack[lp * STRIDE] = kWideLeaf | (uint32_t)hint;
		}
	}

	RTA_DEV bool busy(const int cap) const { return node != kWideNone || sp > 0 || lp < cap; }
	RTA_DEV bool has_node() const { return node != kWideNone || sp > 0; }
	RTA_DEV bool has_list(const int cap) const { return lp < cap; }
	// a visit files up to four references
	RTA_DEV bool has_room() const { return lp - sp >= 4; }

	RTA_DEV void give_up(const int cap)
	{
		doubt = true;
		node = kWideNone;
		sp = 0;
		lp = cap;
	}

	// Precondition: node != kWideNone && has_room()
	//
	// A child's key is its reference (bit 31: triangle; below, as many bits of node or primitive
	// number as the scene needs) with distance bits of the point where the ray enters its box in
	// between (distance_bits), or all ones for a miss.  Sorted as unsigned numbers the four keys
	// come out as nodes nearest first, then triangles nearest first, then misses, and the keys
	// are at once the stack and list entries.
	//
	// ROOT: the node is node 0, where every walk begins: its record is the same for all lanes and is read once per
	// wave through the scalar cache (constant address space: the tree does not change while a kernel runs)
	// instead of once per lane.
	template <bool ROOT = false>
	RTA_DEV void visit(const DeviceTree &tree, uint32_t *stack, WorkCount &work)
	{
		if (COUNT) work.inner++;
		const bool sx = (negative & 1u) != 0, sy = (negative & 2u) != 0, sz = (negative & 4u) != 0;
		// The record: 12 words of binary16 planes, two children per word (children 0,1 then 2,3),
		// in the order lo.x lo.y lo.z hi.x hi.y hi.z, relative to the scene's centre; then the four
		// references.  v_fma_mix_f32 reads a binary16 half as an operand, so nothing is converted.
		typedef _Float16 h2 __attribute__((ext_vector_type(2)));
		uint4 q0, q1, q2, refs;
		if (ROOT) {
			typedef uint32_t Words __attribute__((ext_vector_type(4)));
			const __attribute__((address_space(4))) Words *rec = (const __attribute__((address_space(4))) Words *)tree.wide;
			const Words w0 = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3];
			q0 = make_uint4(w0.x, w0.y, w0.z, w0.w);
			q1 = make_uint4(w1.x, w1.y, w1.z, w1.w);
			q2 = make_uint4(w2.x, w2.y, w2.z, w2.w);
			refs = make_uint4(w3.x, w3.y, w3.z, w3.w);
		} else {
			const uint4 *rec = reinterpret_cast<const uint4 *>(tree.wide) + 4 * (size_t)node;
			q0 = rec[0], q1 = rec[1], q2 = rec[2], refs = rec[3];
		}
		const uint32_t nx01 = sx ? q1.z : q0.x, nx23 = sx ? q1.w : q0.y, fx01 = sx ? q0.x : q1.z, fx23 = sx ? q0.y : q1.w;
		const uint32_t ny01 = sy ? q2.x : q0.z, ny23 = sy ? q2.y : q0.w, fy01 = sy ? q0.z : q2.x, fy23 = sy ? q0.w : q2.y;
		const uint32_t nz01 = sz ? q2.z : q1.x, nz23 = sz ? q2.w : q1.y, fz01 = sz ? q1.x : q2.z, fz23 = sz ? q1.y : q2.w;
		const float reach = best * (1.f + 0x1p-20f);      // +inf while nothing is hit
		const uint32_t dmask = 0x7fffffffu & ~tree.wide_index_mask;
		uint32_t key[4];
#define RTA_WIDE_HALF(word, part) ((float)__builtin_bit_cast(h2, word).part)
#define RTA_WIDE_CHILD(s, pair, part, ref)                                                        \
	{                                                                                             \
		const float ex = __builtin_fmaf(RTA_WIDE_HALF(nx##pair, part), rcp.x, off.x);             \
		const float ey = __builtin_fmaf(RTA_WIDE_HALF(ny##pair, part), rcp.y, off.y);             \
		const float ez = __builtin_fmaf(RTA_WIDE_HALF(nz##pair, part), rcp.z, off.z);             \
		const float xx = __builtin_fmaf(RTA_WIDE_HALF(fx##pair, part), rcp.x, off.x);             \
		const float xy = __builtin_fmaf(RTA_WIDE_HALF(fy##pair, part), rcp.y, off.y);             \
		const float xz = __builtin_fmaf(RTA_WIDE_HALF(fz##pair, part), rcp.z, off.z);             \
		const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(ex, ey), ez), 0.f);      \
		const float t1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(xx, xy), xz), reach);    \
		const uint32_t hit_key = ref | distance_bits(tree, t0, dmask);                             \
		key[s] = t0 <= t1 ? hit_key : 0xffffffffu;                                                \
	}
		RTA_WIDE_CHILD(0, 01, x, refs.x)
		RTA_WIDE_CHILD(1, 01, y, refs.y)
		RTA_WIDE_CHILD(2, 23, x, refs.z)
		RTA_WIDE_CHILD(3, 23, y, refs.w)
#undef RTA_WIDE_CHILD
#undef RTA_WIDE_HALF
#define RTA_WIDE_SORT(a, b)                                         \
	{                                                               \
		const uint32_t lo_ = key[a] < key[b] ? key[a] : key[b];     \
		const uint32_t hi_ = key[a] < key[b] ? key[b] : key[a];     \
		key[a] = lo_;                                               \
		key[b] = hi_;                                               \
	}
		RTA_WIDE_SORT(0, 1)
		RTA_WIDE_SORT(2, 3)
		RTA_WIDE_SORT(0, 2)
		RTA_WIDE_SORT(1, 3)
		RTA_WIDE_SORT(1, 2)
#undef RTA_WIDE_SORT
		// The nearest node is visited next; the other nodes go on the stack and the triangles
		// on the list, the farthest first.  Nodes come first in the order, so key[i] being a
		// node means that key[0] is one too.
		node = (int32_t)key[0] >= 0 ? (key[0] & tree.wide_index_mask) : kWideNone;
#pragma unroll
		for (int i = 3; i >= 1; i--)
			if ((int32_t)key[i] >= 0) {
				stack[sp * STRIDE] = key[i];
				sp++;
			}
#pragma unroll
		for (int i = 3; i >= 0; i--)
			if ((int32_t)key[i] < 0 && key[i] != 0xffffffffu) {
				lp--;
				stack[lp * STRIDE] = key[i];
			}
	}

	// Stack and list entries are the keys of visit(): below the triangle bit the distance at
	// which the ray enters the box as distance_bits() has it (rounded down), then as many bits
	// of node or primitive number as the scene needs (TreeRef::wide_index_mask).  An entry is
	// dropped when it comes up and its distance bits exceed those of the nearest hit so far:
	// distance_bits() is monotonic, so the box is then entered beyond that hit.

	// the next node from the stack whose box the ray may still have to enter (at most two tries per trip)
	RTA_DEV void take_node(const DeviceTree &tree, uint32_t *stack)
	{
		const uint32_t dmask = 0x7fffffffu & ~tree.wide_index_mask;
		const uint32_t reach = distance_bits(tree, best * (1.f + 0x1p-20f), dmask);
#pragma unroll
		for (int k = 0; k < 2; k++)
			if (node == kWideNone && sp > 0) {
				const uint32_t e = stack[(--sp) * STRIDE];
				if ((e & dmask) <= reach)
					node = e & tree.wide_index_mask;
			}
	}

	// Precondition: has_list(cap)
	RTA_DEV void test_listed(const DeviceTree &tree, uint32_t *stack, const int cap, WorkCount &work)
	{
		const uint32_t dmask = 0x7fffffffu & ~tree.wide_index_mask;
		const uint32_t reach = distance_bits(tree, best * (1.f + 0x1p-20f), dmask);
		// the nearer of the list's last two entries: a hit there spares the test of the other
		uint32_t e = stack[lp * STRIDE];
		if (lp + 1 < cap) {
			const uint32_t f = stack[(lp + 1) * STRIDE];
			if ((f & dmask) < (e & dmask)) {
				stack[(lp + 1) * STRIDE] = e;
				e = f;
			}
		}
		lp++;
		if ((e & dmask) > reach) {
			// (the other one is no nearer)
			if (lp < cap)
				lp++;
			return;
		}
		const uint32_t i = e & tree.wide_index_mask;
		if ((int32_t)i == prim)
			return;               // the hinted triangle, met again in the tree: no tie with itself
		if (COUNT) work.tri++;
		const float4 *rec = tree.prims + kPrimQuads * (size_t)i;
		const float4 p0 = rec[0], p1 = rec[1], p2 = rec[2];
		float tu = 0, tv = 0;
		const float t = test_triangle(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), ray, tu, tv);
		if (ANY) {
			if (t < best) {       // an occluder: the walk is over, certify() has the last word
				prim = (int32_t)i;
				node = kWideNone;
				sp = 0;
				lp = cap;
			}
		} else if (t < best) {
			best = t;
			prim = (int32_t)i;
			u = tu;
			v = tv;
			tie = false;
		} else if (t == best && t < FLT_MAX) {
			tie = true;
		}
	}

	// Steps 2 and 3 of the proof for a finished walk.
	RTA_DEV void certify(const DeviceTree &tree)
	{
		if (doubt || prim < 0)
			return;
		if (tie) {
			doubt = true;
			return;
		}
		const float4 b0 = tree.leaf_boxes[2 * (size_t)prim], b1 = tree.leaf_boxes[2 * (size_t)prim + 1];
		const float d = test_box_tame(mk(b0.x, b0.y, b0.z), mk(b1.x, b1.y, b1.z), ray, rcp, best);
		doubt = !(d < FLT_MAX);
	}

	RTA_DEV HitRec result() const { return HitRec{prim, best, u, v}; }
};

// A wave works through more rays than it has lanes (as traverse_pulling).
//   got = grab(want, first)                  reserves up to `want` rays
//   tag = load(i, ray, hint)                 fetches ray number i and a triangle to test first, or -1
//                                            (ANY: load(i, ray, limit))
//   store(finished, tag, hit, doubt)         called by ALL lanes of the wave together; `finished` lanes
//                                            hand in their ray's result, or `doubt`: trace it exactly
// `quorum`: lanes with a listed triangle that make the wave run the triangle test; `waiting_quorum`: lanes
// with nothing else to do that make it.
// per lane; lane 0's trips and runs are the wave's
struct WideCensus {
	uint32_t trips, visit_runs, visit_lanes, test_runs, test_lanes;
};

template <bool COUNT, bool REL, bool ANY, class Grab, class Load, class Store>
RTA_DEV void traverse_wide_pulling_as(const DeviceTree &tree, const uint32_t refill_at, const uint32_t quorum,
                                   const uint32_t waiting_quorum, uint32_t *stack, const int cap, WorkCount &work, WideCensus &census, Grab grab, Load load, Store store)
{
	WideTraversal<COUNT, REL, ANY> tr;
	tr.rest(cap);
	uint32_t tag = 0;
	bool loaded = false, dry = false;
	for (;;) {
		const unsigned long long idle = __ballot(!tr.busy(cap));
		const uint32_t n_idle = (uint32_t)__popcll(idle);
		if (!dry && n_idle >= refill_at) {
			const bool finished = loaded && !tr.busy(cap);
			if (finished)
				tr.certify(tree);
			store(finished, tag, tr.result(), tr.doubt != 0);
			if (finished)
				loaded = false;
			uint32_t first = 0;
			const uint32_t got = grab(n_idle, first, dry);
			const uint32_t rank =
			    __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
			if (!tr.busy(cap) && rank < got) {
				Ray ray;
				if constexpr (ANY) {
					float limit;
					tag = load(first + rank, ray, limit);
					tr.start(tree, ray, cap, limit);
				} else {
					int32_t hint;
					tag = load(first + rank, ray, hint);
					tr.start(tree, ray, cap);
					tr.take_hint(stack, hint);
				}
				loaded = true;
				// (every admitted ray's first node)
				if (tr.node == 0u)
					tr.template visit<true>(tree, stack, work);
			}
		}
		if (__ballot(tr.busy(cap)) == 0) {
			if (dry)
				break;
			// lanes whose rays were not admitted are through at once: the next trip hands them on
			continue;
		}
		// a lane whose column is full cannot visit; it waits for the triangle test to shorten
		// its list, and gives the ray up if there is no list to shorten
		if (tr.has_node() && !tr.has_room() && !tr.has_list(cap))
			tr.give_up(cap);
		const bool can_visit = tr.has_node() && tr.has_room();
		const bool listed = tr.has_list(cap);
		const uint32_t n_visit = (uint32_t)__popcll(__ballot(can_visit));
		const uint32_t n_listed = (uint32_t)__popcll(__ballot(listed));
		// lanes that can do nothing but test a triangle
		const uint32_t n_waiting = (uint32_t)__popcll(__ballot(listed && !can_visit));
		if (COUNT) census.trips++;
		if (n_listed >= quorum || n_waiting >= waiting_quorum || n_visit == 0 || __ballot(tr.has_node() && !tr.has_room())) {
			if (COUNT) census.test_runs++, census.test_lanes += listed;
			if (listed)
				tr.test_listed(tree, stack, cap, work);
		}
		if (n_visit) {
			if (COUNT) census.visit_runs++, census.visit_lanes += can_visit;
			if (can_visit) {
				tr.take_node(tree, stack);
				if (tr.node != kWideNone)
					tr.visit(tree, stack, work);
			}
		}
	}
	// the last rays
	if (loaded)
		tr.certify(tree);
	store(loaded, tag, tr.result(), tr.doubt != 0);
}

// The loop exists twice, once per kind of distance bits: the choice is the tree's, made once per wave.
template <bool COUNT, bool ANY = false, class Grab, class Load, class Store>
RTA_DEV void traverse_wide_pulling(const DeviceTree &tree, const uint32_t refill_at, const uint32_t quorum,
                                   const uint32_t waiting_quorum, uint32_t *stack, const int cap, WorkCount &work, WideCensus &census, Grab grab, Load load, Store store)
{
	if (tree.wide_relative)
		traverse_wide_pulling_as<COUNT, true, ANY>(tree, refill_at, quorum, waiting_quorum, stack, cap, work, census, grab, load, store);
	else
		traverse_wide_pulling_as<COUNT, false, ANY>(tree, refill_at, quorum, waiting_quorum, stack, cap, work, census, grab, load, store);
}

}  // namespace rta
