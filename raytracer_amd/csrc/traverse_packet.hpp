// The certified walk of traverse_wide.hpp for 64 rays that travel together (a
// camera's neighbouring pixels, a coherent batch): the WAVE walks the wide tree,
// one node at a time for all its rays, instead of every lane walking on its own.
//
// A visit reads the node's record once for the wave (scalar loads), every lane
// tests the four child boxes against its own ray, and a child is followed if ANY
// lane's ray enters it.  Triangle children are tested at once, for the lanes whose
// rays enter their boxes, with the reference's own arithmetic (test_triangle); node
// children go on ONE stack for the wave, which lives in two vector registers (lane
// j holds entry j), nearest first by the smallest entry distance among the lanes.
// There are no per-lane keys to sort, no per-lane stack and list in LDS, no lanes
// waiting for work of another kind; what it costs is that every lane sits through
// the union of the nodes that the wave's rays need, which is why it is only used
// where that union is about one ray's share (see the callers).
//
// The proof of traverse_wide.hpp carries over lane by lane.  A triangle p that
// HitTriangle accepts for lane L's ray at t_p <= F_L has all its padded boxes
// entered by that ray no later than t_p (1 + 2^-20) <= the lane's reach at any
// time of the walk (a lane's reach only shrinks towards F_L (1 + 2^-20)), so lane L
// votes for every ancestor of p, the wave visits them all - an entry is only
// dropped from the stack when its smallest entry distance among the voting lanes
// exceeds the largest reach among all lanes, which lane L's vote rules out - and
// at p's own box lane L passes the re-check against its current reach and tests p.
// Steps 2 and 3 (no tie; the leaf's box in the reference's tree passes the
// reference's own test) are checked per lane afterwards, as there.
#pragma once

#include "traverse_wide.hpp"

namespace rta {

// One step of a wave-wide reduction: the value of another lane, by a DPP control of gfx9 (quad
// permutes, row shifts, row broadcasts); a lane without a source keeps its own value.
template <int CTRL, int ROW_MASK>
RTA_DEV int dpp_from(int x)
{
	return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);
}

// Smallest / largest of a 32-bit integer over the 64 lanes of the wave; every lane must be active.
RTA_DEV int wave_min_int(int x)
{
	int a;
	a = dpp_from<0xb1, 0xf>(x), x = a < x ? a : x;       // quad_perm [1, 0, 3, 2]
	a = dpp_from<0x4e, 0xf>(x), x = a < x ? a : x;       // quad_perm [2, 3, 0, 1]
	a = dpp_from<0x114, 0xf>(x), x = a < x ? a : x;      // row_shr 4
	a = dpp_from<0x118, 0xf>(x), x = a < x ? a : x;      // row_shr 8
	a = dpp_from<0x142, 0xa>(x), x = a < x ? a : x;      // row_bcast 15 into rows 1 and 3
	a = dpp_from<0x143, 0xc>(x), x = a < x ? a : x;      // row_bcast 31 into rows 2 and 3
	return __builtin_amdgcn_readlane(x, 63);
}

RTA_DEV int wave_max_int(int x) { return -wave_min_int(-x); }      // (no value here is INT_MIN)

// Ascending order of two unsigned words, both wave-uniform.
#define RTA_PACKET_SORT(a, b)                          \
	{                                                  \
		const uint32_t lo_ = a < b ? a : b;            \
		const uint32_t hi_ = a < b ? b : a;            \
		a = lo_;                                       \
		b = hi_;                                       \
	}

//   mine            this lane holds a ray of the packet
//   store(finished, tag, hit, doubt)      as traverse_wide_pulling's: called by all lanes together
// Work counts: records read by the wave (lane 0 counts), not visits per ray.
template <bool COUNT, class Store>
RTA_DEV void walk_packet(const DeviceTree &tree, const bool mine, const Ray &given, const uint32_t tag, WorkCount &work, Store store)
{
	const uint32_t lane = threadIdx.x & 63;
	const Ray ray = mine ? given : Ray{mk(0, 0, 0), mk(1, 1, 1)};
	const V3 rcp = mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
	const V3 from = ray.o - mk(tree.wide_center[0], tree.wide_center[1], tree.wide_center[2]);
	const V3 off = mk(-(from.x * rcp.x), -(from.y * rcp.y), -(from.z * rcp.z));
	const bool sx = (__float_as_uint(ray.d.x) >> 31) != 0, sy = (__float_as_uint(ray.d.y) >> 31) != 0, sz = (__float_as_uint(ray.d.z) >> 31) != 0;
	// the rays the wide tree's padding covers; NaNs fail the comparisons
	const bool alive = mine && ray_is_tame(ray) && norm2(ray.d) <= kWideMaxDir * kWideMaxDir * (1.f - 0x1p-20f) && norm2(from) <= tree.wide_reach2;
	bool doubt = mine && !alive, tie = false;
	float best = FLT_MAX, u = 0.f, v = 0.f;
	int32_t prim = -1;
	// spheres are outside the tree: each is tested with the reference's own arithmetic before the walk begins
	for (uint32_t k = 0; k < tree.n_loose; k++) {
		const uint32_t i = ((ConstWords)tree.loose)[k];
		const ConstQuads rec = (ConstQuads)tree.prims + kPrimQuads * (size_t)i;
		const Quad p0 = rec[0], p1 = rec[1];
		const float t = alive ? test_sphere(mk(p0.x, p0.y, p0.z), p1.x, ray) : FLT_MAX;
		if (t < best) {
			best = t;
			prim = (int32_t)i;
			tie = false;
		} else if (t == best && t < FLT_MAX) {
			tie = true;
		}
	}

	// the wave's stack: entry j lives in lane j of these two registers
	uint32_t stack_ref = 0, stack_near = 0;
	int sp = 0;
	uint32_t node = 0;
	bool overflow = false;
	// largest reach among the wave's rays, as the bits of a float (they order like integers; -1 for no ray)
	int wave_far = __builtin_amdgcn_readfirstlane(__float_as_int(__builtin_inff()));
	typedef _Float16 h2 __attribute__((ext_vector_type(2)));
	bool walking = __ballot(alive) != 0;
	while (walking) {
		if (COUNT && lane == 0) work.inner++;
		const uint4 *rec = reinterpret_cast<const uint4 *>(tree.wide) + 4 * (size_t)node;
		const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], refs = rec[3];
		const uint32_t nx01 = sx ? q1.z : q0.x, nx23 = sx ? q1.w : q0.y, fx01 = sx ? q0.x : q1.z, fx23 = sx ? q0.y : q1.w;
		const uint32_t ny01 = sy ? q2.x : q0.z, ny23 = sy ? q2.y : q0.w, fy01 = sy ? q0.z : q2.x, fy23 = sy ? q0.w : q2.y;
		const uint32_t nz01 = sz ? q2.z : q1.x, nz23 = sz ? q2.w : q1.y, fz01 = sz ? q1.x : q2.z, fz23 = sz ? q1.y : q2.w;
		const float reach = alive ? best * (1.f + 0x1p-20f) : -1.f;      // +inf while nothing is hit
		float t0[4];
		bool hit[4];
#define RTA_PACKET_HALF(word, part) ((float)__builtin_bit_cast(h2, word).part)
#define RTA_PACKET_CHILD(s, pair, part)                                                           \
	{                                                                                             \
		const float ex = __builtin_fmaf(RTA_PACKET_HALF(nx##pair, part), rcp.x, off.x);           \
		const float ey = __builtin_fmaf(RTA_PACKET_HALF(ny##pair, part), rcp.y, off.y);           \
		const float ez = __builtin_fmaf(RTA_PACKET_HALF(nz##pair, part), rcp.z, off.z);           \
		const float xx = __builtin_fmaf(RTA_PACKET_HALF(fx##pair, part), rcp.x, off.x);           \
		const float xy = __builtin_fmaf(RTA_PACKET_HALF(fy##pair, part), rcp.y, off.y);           \
		const float xz = __builtin_fmaf(RTA_PACKET_HALF(fz##pair, part), rcp.z, off.z);           \
		t0[s] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(ex, ey), ez), 0.f);               \
		const float t1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(xx, xy), xz), reach);    \
		hit[s] = t0[s] <= t1;                                                                     \
	}
		RTA_PACKET_CHILD(0, 01, x)
		RTA_PACKET_CHILD(1, 01, y)
		RTA_PACKET_CHILD(2, 23, x)
		RTA_PACKET_CHILD(3, 23, y)
#undef RTA_PACKET_CHILD
#undef RTA_PACKET_HALF
		const uint32_t ref[4] = {refs.x, refs.y, refs.z, refs.w};
		// Triangle children first, in the record's order: tested for the lanes whose rays enter their boxes
		// before the nearest hit so far (the reach may have shrunk since the box test above).
		bool improved = false;
		uint32_t key[4];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			key[k] = 0xffffffffu;
			const unsigned long long votes = __ballot(hit[k]);
			if (votes == 0 || ref[k] == kWideNone)
				continue;
			if (ref[k] & kWideLeaf) {
				const uint32_t i = ref[k] & ~kWideLeaf;
				const bool want = hit[k] && t0[k] <= best * (1.f + 0x1p-20f) && (int32_t)i != prim;
				if (__ballot(want) == 0)
					continue;
				if (COUNT && lane == 0) work.tri++;
				const float4 *tri = tree.prims + kPrimQuads * (size_t)i;
				const float4 p0 = tri[0], p1 = tri[1], p2 = tri[2];
				if (want) {
					float tu = 0, tv = 0;
					const float t = test_triangle(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), ray, tu, tv);
					if (t < best) {
						best = t;
						prim = (int32_t)i;
						u = tu;
						v = tv;
						tie = false;
						improved = true;
					} else if (t == best && t < FLT_MAX) {
						tie = true;
					}
				}
			} else {
				// a node: the smallest entry distance among the voting lanes, rounded down to make room for
				// the child's place in the record (distances are >= 0: their bits order like integers)
				const int near = wave_min_int(hit[k] ? __float_as_int(t0[k]) : 0x7f800000);
				key[k] = ((uint32_t)near & ~3u) | (uint32_t)k;
			}
		}
		if (__ballot(improved))
			wave_far = wave_max_int(alive ? __float_as_int(best * (1.f + 0x1p-20f)) : __float_as_int(-1.f));
		// the node children, nearest first (wave-uniform words)
		RTA_PACKET_SORT(key[0], key[1])
		RTA_PACKET_SORT(key[2], key[3])
		RTA_PACKET_SORT(key[0], key[2])
		RTA_PACKET_SORT(key[1], key[3])
		RTA_PACKET_SORT(key[1], key[2])
#pragma unroll
		for (int k = 3; k >= 1; k--)
			if (key[k] != 0xffffffffu) {
				const uint32_t which = key[k] & 3u;
				const uint32_t r = which == 0 ? refs.x : (which == 1 ? refs.y : (which == 2 ? refs.z : refs.w));
				if (sp < 64) {
					if (lane == (uint32_t)sp) {
						stack_ref = r;
						stack_near = key[k] & ~3u;
					}
					sp++;
				} else {
					overflow = true;      // (a wide tree is some fifteen levels deep: three entries a level fit)
				}
			}
		node = kWideNone;
		if (key[0] != 0xffffffffu) {
			const uint32_t which = key[0] & 3u;
			node = which == 0 ? refs.x : (which == 1 ? refs.y : (which == 2 ? refs.z : refs.w));
		}
		// else the next entry of the stack that some ray may still have to enter
		while (node == kWideNone && sp > 0) {
			sp--;
			const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)stack_ref, sp);
			const int near = __builtin_amdgcn_readlane((int)stack_near, sp);
			if (near <= wave_far)
				node = r;
		}
		walking = node != kWideNone;
	}
	if (overflow && alive)
		doubt = true;
	// steps 2 and 3 of the proof, per lane (WideTraversal::certify)
	if (mine && !doubt && prim >= 0) {
		if (tie) {
			doubt = true;
		} else {
			const float4 b0 = tree.leaf_boxes[2 * (size_t)prim], b1 = tree.leaf_boxes[2 * (size_t)prim + 1];
			const float d = test_box_tame(mk(b0.x, b0.y, b0.z), mk(b1.x, b1.y, b1.z), ray, rcp, best);
			doubt = !(d < FLT_MAX);
		}
	}
	store(mine, tag, HitRec{prim, best, u, v}, doubt);
}

#undef RTA_PACKET_SORT

}  // namespace rta
