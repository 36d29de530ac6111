// rta_upload_scene on the GPU (see upload_device.hpp).  What context.cpp's flatten()
// and wide_tree.cpp's build_wide_tree() do on one host thread - 3 s for a scene of
// 10 M triangles - is done here by kernels over the caller's arrays:
//
//   1. the description's checks (index ranges, every node reached once, every
//      primitive in at most one leaf, depth within the traversal stack) by a
//      breadth-first walk of the reference's tree (src/main.cpp:20-23), one launch
//      per level;
//   2. pair records in depth-first order (first child first, the numbering of the
//      host path), primitive and shading records, the reference's node records;
//   3. the wide tree of the certified walk (traverse_wide.hpp) on the reference's
//      topology: leaves cut down to single triangles, boxes and longest edges from
//      the bottom up, collapse to four children per node by opening the child with
//      the largest surface, depth-first numbers from subtree sizes, padded
//      binary16 planes.
//
// The arrays come out as the host path's do, byte for byte (tests compare them).
// Anything this path does not handle - an invalid description, whose error the
// host path words; a leaf of more than kMaxLeaf primitives, which one thread
// would have to cut down - is left to the host path.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <string>
#include <vector>

#include <rocprim/device/device_scan.hpp>

#include "upload_device.hpp"
#include "wide_tree.hpp"

namespace rta {

namespace {

constexpr int kT = 256;
constexpr int kMaxLeaf = 16;           // primitives of a reference leaf that one thread cuts down to single triangles
constexpr int kMaxDepth = 64;          // the reference's traversal stack (src/main.cpp:339)
constexpr uint32_t kLeaf = 0x80000000u;
constexpr uint32_t kSphereFlag = 1u, kLastFlag = 2u;

struct Flags {
	uint32_t bad;            // the description is invalid
	uint32_t big_leaf;       // a leaf of more than kMaxLeaf primitives
	uint32_t not_nested;     // a child's box reaches outside its parent's
	uint32_t not_tame;       // a box coordinate out of the ordinary range, or an inverted box
	uint32_t nonfinite;      // a triangle with a non-finite vertex
	uint32_t spheres;        // number of spheres
	uint32_t next_count;     // nodes appended to the next level
	uint32_t overflow;       // a plane beyond binary16's range
	unsigned long long widened_sum;      // sum over children of widening / extent, in units of 2^-32
	unsigned long long widened_n;
};

struct Ref {
	const float *bounds;
	const int32_t *index, *count;
	int32_t n_nodes, n_prims;
};

inline unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + kT - 1) / kT); }

__device__ inline float lesser(float a, float b) { return b < a ? b : a; }      // std::min
__device__ inline float greater(float a, float b) { return a < b ? b : a; }     // std::max

__global__ void k_check_prims(int32_t n, const int32_t *type, const int32_t *material, const int32_t *light, int32_t n_materials,
                              int32_t n_lights, const float *geom, Flags *flags)
{
	const int32_t k = blockIdx.x * kT + threadIdx.x;
	if (k >= n)
		return;
	const int32_t t = type[k];
	if ((t != RTA_PRIM_TRIANGLE && t != RTA_PRIM_SPHERE) || material[k] < 0 || material[k] >= n_materials || light[k] < -1 ||
	    light[k] >= n_lights)
		flags->bad = 1;
	if (t == RTA_PRIM_SPHERE)
		atomicAdd(&flags->spheres, 1u);
	else {
		bool finite = true;
		for (int a = 0; a < 9; a++)
			finite = finite && __builtin_isfinite(geom[(size_t)k * 9 + a]);
		if (!finite)
			flags->nonfinite = 1;
	}
}

// One level of the breadth-first walk: the nodes order[begin, end) check themselves and append their
// children, side by side, behind `end`.  covered[k]: 1 once a leaf holds primitive k, 3 for the last of its leaf.
__global__ void k_expand(const Ref ref, int32_t begin, int32_t end, int32_t *order, uint32_t *visited, uint32_t *covered,
                         float4 *leaf_boxes, int32_t *extra, Flags *flags)
{
	const int32_t i = begin + blockIdx.x * kT + threadIdx.x;
	if (i >= end)
		return;
	const int32_t node = order[i];
	const int32_t idx = ref.index[node], cnt = ref.count[node];
	const float *nb = ref.bounds + (size_t)node * 6;
	if (cnt > 0) {
		if (idx < 0 || (int64_t)idx + cnt > ref.n_prims) {
			flags->bad = 1;
			return;
		}
		if (cnt > kMaxLeaf) {
			flags->big_leaf = 1;
			return;
		}
		for (int32_t k = idx; k < idx + cnt; k++) {
			if (atomicExch(&covered[k], k == idx + cnt - 1 ? 3u : 1u))
				flags->bad = 1;      // the primitive belongs to two leaves
			leaf_boxes[(size_t)k * 2] = make_float4(nb[0], nb[1], nb[2], 1.f);      // (1: the primitive is in a leaf)
			leaf_boxes[(size_t)k * 2 + 1] = make_float4(nb[3], nb[4], nb[5], 0.f);
		}
		extra[node] = 2 * cnt - 2;
		return;
	}
	if (cnt < 0 || idx <= 0 || (int64_t)idx + 1 >= ref.n_nodes) {
		flags->bad = 1;
		return;
	}
	const bool again = atomicExch(&visited[idx], 1u) != 0;
	const bool again2 = atomicExch(&visited[idx + 1], 1u) != 0;
	if (again || again2) {
		flags->bad = 1;          // a node is reachable twice
		return;
	}
	if (node != 0)
		for (int c = 0; c < 2; c++) {
			const float *cb = ref.bounds + (size_t)(idx + c) * 6;
			for (int a = 0; a < 3; a++)
				if (!(cb[a] >= nb[a] && cb[3 + a] <= nb[3 + a]))
					flags->not_nested = 1;
		}
	const uint32_t at = atomicAdd(&flags->next_count, 2u);
	order[end + at] = idx;
	order[end + at + 1] = idx + 1;
}

// inner nodes below and including each node, one level at a time from the bottom
__global__ void k_inner_counts(const Ref ref, int32_t begin, int32_t end, const int32_t *order, int32_t *inner)
{
	const int32_t i = begin + blockIdx.x * kT + threadIdx.x;
	if (i >= end)
		return;
	const int32_t node = order[i];
	const int32_t idx = ref.index[node];
	inner[node] = ref.count[node] > 0 ? 0 : 1 + inner[idx] + inner[idx + 1];
}

// depth-first numbers of the inner nodes (their pair records), first child first, one level at a time from the top
__global__ void k_pair_numbers(const Ref ref, int32_t begin, int32_t end, const int32_t *order, const int32_t *inner, int32_t *pair_of)
{
	const int32_t i = begin + blockIdx.x * kT + threadIdx.x;
	if (i >= end)
		return;
	const int32_t node = order[i];
	if (ref.count[node] > 0)
		return;
	const int32_t idx = ref.index[node], p = pair_of[node];
	pair_of[idx] = ref.count[idx] > 0 ? -1 : p + 1;
	pair_of[idx + 1] = ref.count[idx + 1] > 0 ? -1 : p + 1 + inner[idx];
}

__device__ inline uint32_t ref_of(const Ref &ref, const int32_t *pair_of, int32_t node)
{
	return ref.count[node] > 0 ? (kLeaf | (uint32_t)ref.index[node]) : (uint32_t)pair_of[node];
}

// pair records of the reached inner nodes
__global__ void k_pairs(const Ref ref, int32_t reached, const int32_t *order, const int32_t *pair_of, float4 *pairs)
{
	const int32_t i = blockIdx.x * kT + threadIdx.x;
	if (i >= reached)
		return;
	const int32_t node = order[i];
	if (ref.count[node] > 0)
		return;
	const int32_t l = ref.index[node], r = l + 1;
	const float *a = ref.bounds + (size_t)l * 6, *b = ref.bounds + (size_t)r * 6;
	float4 *rec = pairs + (size_t)pair_of[node] * 4;
	rec[0] = make_float4(a[0], a[1], a[2], a[3]);
	rec[1] = make_float4(a[4], a[5], b[0], b[1]);
	rec[2] = make_float4(b[2], b[3], b[4], b[5]);
	rec[3] = make_float4(__uint_as_float(ref_of(ref, pair_of, l)), __uint_as_float(ref_of(ref, pair_of, r)), 0.f, 0.f);
}

// the reference's own node records (all of them, reached or not) and the checks on the boxes' coordinates
__global__ void k_ref_nodes(const Ref ref, float4 *out, Flags *flags)
{
	const int32_t k = blockIdx.x * kT + threadIdx.x;
	if (k >= ref.n_nodes)
		return;
	const float *b = ref.bounds + (size_t)k * 6;
	out[(size_t)k * 2] = make_float4(b[0], b[1], b[2], b[3]);
	out[(size_t)k * 2 + 1] = make_float4(b[4], b[5], __uint_as_float((uint32_t)ref.index[k]), __uint_as_float((uint32_t)ref.count[k]));
	bool tame = b[0] <= b[3] && b[1] <= b[4] && b[2] <= b[5];
	for (int a = 0; a < 6; a++) {
		const float m = __builtin_fabsf(b[a]);
		tame = tame && (m == 0.f || (m >= 0x1p-78f && m <= 0x1p40f));
	}
	if (!tame)
		flags->not_tame = 1;
}

__global__ void k_prims(int32_t n, const int32_t *type, const int32_t *material, const int32_t *light, const float *geom,
                        const float *normals, const uint32_t *covered, float4 *prims, float4 *shade)
{
	const int32_t k = blockIdx.x * kT + threadIdx.x;
	if (k >= n)
		return;
	const float *g = geom + (size_t)k * 9, *nn = normals + (size_t)k * 9;
	// an uncovered primitive terminates its own (never visited) leaf
	uint32_t flags = ((covered[k] & 2u) || !covered[k]) ? kLastFlag : 0u;
	float4 *rec = prims + (size_t)k * kPrimQuads;
	if (type[k] == RTA_PRIM_SPHERE) {
		flags |= kSphereFlag;
		rec[0] = make_float4(g[0], g[1], g[2], __uint_as_float(flags));
		rec[1] = make_float4(g[3], 0.f, 0.f, 0.f);
		rec[2] = make_float4(0.f, 0.f, 0.f, 0.f);
	} else {
		// e0 = v0 - v2, e1 = v1 - v2 (src/main.cpp:252-253), rounded once in fp32
		rec[0] = make_float4(g[6], g[7], g[8], __uint_as_float(flags));
		rec[1] = make_float4(g[0] - g[6], g[1] - g[7], g[2] - g[8], 0.f);
		rec[2] = make_float4(g[3] - g[6], g[4] - g[7], g[5] - g[8], 0.f);
	}
	rec[3] = make_float4(0.f, 0.f, 0.f, 0.f);
	shade[(size_t)k * 3 + 0] = make_float4(nn[0], nn[1], nn[2], __uint_as_float((uint32_t)material[k]));
	shade[(size_t)k * 3 + 1] = make_float4(nn[3], nn[4], nn[5], __uint_as_float((uint32_t)light[k]));
	shade[(size_t)k * 3 + 2] = make_float4(nn[6], nn[7], nn[8], __uint_as_float((uint32_t)type[k]));
}

// ------------------------------------------------------------------ the wide tree

// The binary tree that is collapsed: node k < n_nodes is the reference's node k; a leaf of several
// triangles is the root of a little tree of its own whose other nodes follow at n_nodes + extra_at[k].
struct Binary {
	float *lo, *hi;          // 3 each
	int32_t *left, *right;   // children, or -1
	int32_t *prim;           // >= 0: a triangle
	float *edge;             // longest edge vector of a triangle below
};

__device__ inline void binary_leaf(const Binary &b, int32_t at, int32_t prim, const float *geom)
{
	const float *g = geom + (size_t)prim * 9;
	double e0 = 0, e1 = 0;
	for (int k = 0; k < 3; k++) {
		b.lo[(size_t)at * 3 + k] = lesser(g[k], lesser(g[3 + k], g[6 + k]));
		b.hi[(size_t)at * 3 + k] = greater(g[k], greater(g[3 + k], g[6 + k]));
		// the edge vectors as the kernels hold them: v0 - v2 and v1 - v2 rounded to binary32
		const float a = g[k] - g[6 + k], c = g[3 + k] - g[6 + k];
		e0 += (double)a * a;
		e1 += (double)c * c;
	}
	b.edge[at] = (float)(__builtin_sqrt(e0 > e1 ? e0 : e1) * (1 + 1e-6));
	b.left[at] = b.right[at] = -1;
	b.prim[at] = prim;
}

__device__ inline void binary_join(const Binary &b, int32_t at, int32_t l, int32_t r)
{
	for (int k = 0; k < 3; k++) {
		b.lo[(size_t)at * 3 + k] = lesser(b.lo[(size_t)l * 3 + k], b.lo[(size_t)r * 3 + k]);
		b.hi[(size_t)at * 3 + k] = greater(b.hi[(size_t)l * 3 + k], b.hi[(size_t)r * 3 + k]);
	}
	b.edge[at] = greater(b.edge[l], b.edge[r]);
	b.left[at] = l;
	b.right[at] = r;
	b.prim[at] = -1;
}

__device__ inline float centroid_of(const float *geom, int32_t prim, int axis)
{
	const float *g = geom + (size_t)prim * 9;
	return (g[axis] + g[3 + axis] + g[6 + axis]) * (1.f / 3.f);
}

// The reference's leaves, cut down to single triangles: the median of the centroids on their longest
// axis, ties by primitive number (wide_tree.cpp: Builder::cut).  One thread per reached leaf.
__global__ void k_leaf_trees(const Ref ref, int32_t reached, const int32_t *order, const float *geom, const int32_t *extra_at, Binary b)
{
	const int32_t i = blockIdx.x * kT + threadIdx.x;
	if (i >= reached)
		return;
	const int32_t node = order[i];
	const int32_t idx = ref.index[node], cnt = ref.count[node];
	if (cnt <= 0)
		return;
	if (cnt == 1) {
		binary_leaf(b, node, idx, geom);
		return;
	}
	int32_t ids[kMaxLeaf], made[kMaxLeaf];
	int32_t range_begin[kMaxLeaf], range_end[kMaxLeaf], range_at[kMaxLeaf];
	for (int k = 0; k < cnt; k++)
		ids[k] = idx + k;
	int sp = 0, n_made = 0;
	int32_t next = ref.n_nodes + extra_at[node];
	range_begin[0] = 0, range_end[0] = cnt, range_at[0] = node;
	sp = 1;
	while (sp > 0) {
		sp--;
		const int32_t begin = range_begin[sp], end = range_end[sp], at = range_at[sp];
		if (end - begin == 1) {
			binary_leaf(b, at, ids[begin], geom);
			continue;
		}
		float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
		for (int32_t k = begin; k < end; k++)
			for (int a = 0; a < 3; a++) {
				const float c = centroid_of(geom, ids[k], a);
				lo[a] = lesser(lo[a], c);
				hi[a] = greater(hi[a], c);
			}
		int axis = 0;
		for (int a = 1; a < 3; a++)
			if (hi[a] - lo[a] > hi[axis] - lo[axis])
				axis = a;
		// in full order by (centroid, number): the lower half is what a selection of the median leaves there
		for (int32_t k = begin + 1; k < end; k++) {
			const int32_t x = ids[k];
			const float cx = centroid_of(geom, x, axis);
			int32_t j = k;
			while (j > begin) {
				const int32_t y = ids[j - 1];
				const float cy = centroid_of(geom, y, axis);
				if (!(cx < cy || (cx == cy && x < y)))
					break;
				ids[j] = y;
				j--;
			}
			ids[j] = x;
		}
		const int32_t mid = begin + (end - begin) / 2;
		const int32_t l = next++, r = next++;
		b.left[at] = l;
		b.right[at] = r;
		made[n_made++] = at;
		range_begin[sp] = begin, range_end[sp] = mid, range_at[sp] = l;
		sp++;
		range_begin[sp] = mid, range_end[sp] = end, range_at[sp] = r;
		sp++;
	}
	// parents were made before their children: joined in reverse, every node finds its children done
	for (int k = n_made - 1; k >= 0; k--)
		binary_join(b, made[k], b.left[made[k]], b.right[made[k]]);
}

// the reference's inner nodes, one level at a time from the bottom
__global__ void k_join_level(const Ref ref, int32_t begin, int32_t end, const int32_t *order, Binary b)
{
	const int32_t i = begin + blockIdx.x * kT + threadIdx.x;
	if (i >= end)
		return;
	const int32_t node = order[i];
	if (ref.count[node] > 0)
		return;
	const int32_t idx = ref.index[node];
	binary_join(b, node, idx, idx + 1);
}

__device__ inline double surface_of(const Binary &b, int32_t at)
{
	const double x = (double)b.hi[(size_t)at * 3] - b.lo[(size_t)at * 3], y = (double)b.hi[(size_t)at * 3 + 1] - b.lo[(size_t)at * 3 + 1],
	             z = (double)b.hi[(size_t)at * 3 + 2] - b.lo[(size_t)at * 3 + 2];
	return x * y + y * z + z * x;
}

// One level of wide nodes: node w stands on binary node wide_at[w]; it starts with that node's two
// children and replaces the inner child with the largest surface by its two children until it has
// four or none is left to open.  Its inner children are the wide nodes of the next level.
__global__ void k_wide_expand(int32_t begin, int32_t end, int32_t *wide_at, int32_t *kids_out, int32_t *child_out, const Binary b,
                              Flags *flags)
{
	const int32_t w = begin + blockIdx.x * kT + threadIdx.x;
	if (w >= end)
		return;
	const int32_t at = wide_at[w];
	int32_t kids[4] = {b.left[at], b.right[at], -1, -1};
	int n = 2;
	while (n < 4) {
		int best = -1;
		double area = -1;
		for (int k = 0; k < n; k++)
			if (b.prim[kids[k]] < 0) {
				const double s = surface_of(b, kids[k]);
				if (s > area) {
					area = s;
					best = k;
				}
			}
		if (best < 0)
			break;
		const int32_t open = kids[best];
		for (int k = n; k > best + 1; k--)
			kids[k] = kids[k - 1];
		kids[best] = b.left[open];
		kids[best + 1] = b.right[open];
		n++;
	}
	for (int k = 0; k < 4; k++) {
		int32_t child = -1;
		if (k < n && b.prim[kids[k]] < 0) {
			child = end + (int32_t)atomicAdd(&flags->next_count, 1u);
			wide_at[child] = kids[k];
		}
		kids_out[(size_t)w * 4 + k] = k < n ? kids[k] : -1;
		child_out[(size_t)w * 4 + k] = child;
	}
}

// wide nodes below and including each wide node, one level at a time from the bottom
__global__ void k_wide_counts(int32_t begin, int32_t end, const int32_t *child, int32_t *below)
{
	const int32_t w = begin + blockIdx.x * kT + threadIdx.x;
	if (w >= end)
		return;
	int32_t n = 1;
	for (int k = 0; k < 4; k++)
		if (child[(size_t)w * 4 + k] >= 0)
			n += below[child[(size_t)w * 4 + k]];
	below[w] = n;
}

// depth-first numbers, children in their order, one level at a time from the top
__global__ void k_wide_numbers(int32_t begin, int32_t end, const int32_t *child, const int32_t *below, int32_t *number)
{
	const int32_t w = begin + blockIdx.x * kT + threadIdx.x;
	if (w >= end)
		return;
	int32_t next = number[w] + 1;
	for (int k = 0; k < 4; k++) {
		const int32_t c = child[(size_t)w * 4 + k];
		if (c >= 0) {
			number[c] = next;
			next += below[c];
		}
	}
}

// binary16 neighbours of x, as bit patterns: the largest one <= x, the smallest one >= x.  (A
// conversion lands on one of the two neighbours however it rounds; the comparison settles which.)
__device__ inline uint16_t half_bits(_Float16 h) { return __builtin_bit_cast(uint16_t, h); }
__device__ inline _Float16 half_of(uint16_t b) { return __builtin_bit_cast(_Float16, b); }

__device__ inline uint16_t half_below(double x)
{
	const _Float16 h = (_Float16)(float)x;
	if ((double)h <= x)
		return half_bits(h);
	const uint16_t b = half_bits(h);
	if ((b & 0x7fffu) == 0)
		return 0x8001u;                   // below zero: the smallest negative subnormal
	return (b & 0x8000u) ? (uint16_t)(b + 1) : (uint16_t)(b - 1);
}

__device__ inline uint16_t half_above(double x)
{
	const _Float16 h = (_Float16)(float)x;
	if ((double)h >= x)
		return half_bits(h);
	const uint16_t b = half_bits(h);
	if ((b & 0x7fffu) == 0)
		return 0x0001u;
	return (b & 0x8000u) ? (uint16_t)(b - 1) : (uint16_t)(b + 1);
}

struct Padding {
	double s_max, cmax, radius;      // farthest admitted origin, largest centre coordinate, the scene's radius
	double center[3];                // as binary32 values
};

// the padding of wide_tree.cpp for a box over triangles whose longest edge is e
__device__ inline double pad_of(const Padding &p, double e)
{
	const double u = 0x1p-24;
	const double rho = 16.02e4 * u * (double)kWideMaxDir * e * e * (p.s_max + e) + 2 * u * e;
	const double dt = 16.02e4 * u * e * e * (p.s_max + e);
	const double slab = 8 * u * (p.s_max + p.cmax + p.radius);
	return 2 * (rho + (double)kWideMaxDir * dt + slab);
}

__device__ inline unsigned long long wave_sum(unsigned long long v)
{
	for (int m = 32; m >= 1; m >>= 1) {
		const uint32_t lo = __shfl_xor((uint32_t)v, m), hi = __shfl_xor((uint32_t)(v >> 32), m);
		v += ((unsigned long long)hi << 32) | lo;
	}
	return v;
}

// The records: planes relative to the centre, as binary16 rounded outwards, and the references.
__global__ void k_wide_emit(int32_t n_wide, const int32_t *kids, const int32_t *child, const int32_t *number, const Binary b,
                            const Padding pad, float4 *nodes, Flags *flags)
{
	const int32_t w = blockIdx.x * kT + threadIdx.x;
	unsigned long long widened_sum = 0, widened_n = 0;
	if (w < n_wide) {
		uint16_t plane[6][4];
		uint32_t ref[4];
		bool overflow = false;
		for (int k = 0; k < 4; k++) {
			const int32_t kid = kids[(size_t)w * 4 + k];
			if (kid < 0) {
				for (int a = 0; a < 3; a++) {
					plane[a][k] = 0x7c00u;          // +inf
					plane[3 + a][k] = 0xfc00u;      // -inf
				}
				ref[k] = kWideNone;
				continue;
			}
			const double p = pad_of(pad, (double)b.edge[kid]);
			double extent = 0, widened = 0;
			for (int a = 0; a < 3; a++) {
				const float lo = b.lo[(size_t)kid * 3 + a], hi = b.hi[(size_t)kid * 3 + a];
				plane[a][k] = half_below((double)lo - p - pad.center[a]);
				plane[3 + a][k] = half_above((double)hi + p - pad.center[a]);
				const double lo_h = (double)half_of(plane[a][k]), hi_h = (double)half_of(plane[3 + a][k]);
				if (!__builtin_isfinite(lo_h) || !__builtin_isfinite(hi_h))
					overflow = true;
				extent += (hi_h - lo_h) / 3;
				widened += ((hi_h - lo_h) - ((double)hi - lo)) / 6;
			}
			if (extent > 0) {
				const double share = widened / extent;
				widened_sum += (unsigned long long)((share < 0 ? 0 : (share > 1024 ? 1024 : share)) * 0x1p32);
				widened_n++;
			}
			ref[k] = b.prim[kid] >= 0 ? (kWideLeaf | (uint32_t)b.prim[kid]) : (uint32_t)number[child[(size_t)w * 4 + k]];
		}
		if (overflow)
			flags->overflow = 1;
		float4 *rec = nodes + (size_t)number[w] * 4;
		uint32_t words[12];
		for (int a = 0; a < 6; a++) {
			words[a * 2] = (uint32_t)plane[a][0] | ((uint32_t)plane[a][1] << 16);
			words[a * 2 + 1] = (uint32_t)plane[a][2] | ((uint32_t)plane[a][3] << 16);
		}
		for (int q = 0; q < 3; q++)
			rec[q] = make_float4(__uint_as_float(words[q * 4]), __uint_as_float(words[q * 4 + 1]), __uint_as_float(words[q * 4 + 2]),
			                     __uint_as_float(words[q * 4 + 3]));
		rec[3] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), __uint_as_float(ref[2]), __uint_as_float(ref[3]));
	}
	widened_sum = wave_sum(widened_sum);
	widened_n = wave_sum(widened_n);
	if ((threadIdx.x & 63) == 0 && widened_n) {
		atomicAdd(&flags->widened_sum, widened_sum);
		atomicAdd(&flags->widened_n, widened_n);
	}
}

// temporaries of one upload
struct Arena {
	std::vector<void *> ptrs;
	hipError_t err = hipSuccess;
	template <typename T>
	T *get(size_t count)
	{
		void *p = nullptr;
		if (err == hipSuccess)
			err = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
		if (err == hipSuccess)
			ptrs.push_back(p);
		return (T *)p;
	}
	~Arena()
	{
		for (void *p : ptrs)
			(void)hipFree(p);
	}
};

inline float below(double x)
{
	float f = (float)x;
	return (double)f <= x ? f : std::nextafterf(f, -INFINITY);
}

double seconds_since(std::chrono::steady_clock::time_point t0)
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

bool upload_goes_to_device(const rta_scene_desc &d)
{
	if (const char *v = getenv("RTA_UPLOAD_DEVICE"))
		return atoi(v) != 0;
	return (size_t)d.n_prims > kWideSahLimit;
}

#define UP_TRY(call)                                 \
	do {                                             \
		hipError_t e__ = (call);                     \
		if (e__ != hipSuccess)                       \
			return ctx->hip_fail(e__, #call);        \
	} while (0)

int upload_scene_on_device(rta_context *ctx, const rta_scene_desc &d)
{
	const auto t_start = std::chrono::steady_clock::now();
	const bool verbose = getenv("RTA_WIDE_VERBOSE") != nullptr;
	// what the kernels cannot look at: counts, null arrays, the lights (a handful)
	if (d.n_prims <= 0 || d.n_nodes <= 0 || d.n_materials <= 0 || d.n_lights < 0)
		return kUploadNotHandled;
	if (!d.prim_type || !d.prim_material || !d.prim_light || !d.prim_geom || !d.prim_normals || !d.node_bounds || !d.node_index ||
	    !d.node_count || !d.materials || (d.n_lights && (!d.light_color || !d.light_prim)))
		return kUploadNotHandled;
	if ((uint32_t)d.n_prims >= kLeaf || (uint32_t)d.n_nodes >= kLeaf)
		return kUploadNotHandled;
	for (int32_t l = 0; l < d.n_lights; l++) {
		const int32_t p = d.light_prim[l];
		if (p < 0 || p >= d.n_prims || d.prim_type[p] != RTA_PRIM_TRIANGLE)
			return kUploadNotHandled;
	}
	const int32_t P = d.n_prims, N = d.n_nodes;
	hipStream_t s = ctx->stream;
	Arena A;
	int32_t *d_type = A.get<int32_t>(P), *d_material = A.get<int32_t>(P), *d_light = A.get<int32_t>(P);
	float *d_geom = A.get<float>((size_t)P * 9), *d_normals = A.get<float>((size_t)P * 9), *d_bounds = A.get<float>((size_t)N * 6);
	int32_t *d_index = A.get<int32_t>(N), *d_count = A.get<int32_t>(N);
	int32_t *order = A.get<int32_t>(N), *extra = A.get<int32_t>(N), *inner = A.get<int32_t>(N), *pair_of = A.get<int32_t>(N);
	uint32_t *visited = A.get<uint32_t>(N), *covered = A.get<uint32_t>(P);
	Flags *flags = A.get<Flags>(1);
	if (A.err != hipSuccess)
		return ctx->hip_fail(A.err, "hipMalloc (scene upload)");
	UP_TRY(ctx->leaf_boxes.reserve((size_t)P * 32));
	UP_TRY(hipMemcpyAsync(d_type, d.prim_type, (size_t)P * 4, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemcpyAsync(d_material, d.prim_material, (size_t)P * 4, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemcpyAsync(d_light, d.prim_light, (size_t)P * 4, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemcpyAsync(d_geom, d.prim_geom, (size_t)P * 36, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemcpyAsync(d_normals, d.prim_normals, (size_t)P * 36, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemcpyAsync(d_bounds, d.node_bounds, (size_t)N * 24, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemcpyAsync(d_index, d.node_index, (size_t)N * 4, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemcpyAsync(d_count, d.node_count, (size_t)N * 4, hipMemcpyHostToDevice, s));
	UP_TRY(hipMemsetAsync(flags, 0, sizeof(Flags), s));
	UP_TRY(hipMemsetAsync(visited, 0, (size_t)N * 4, s));
	UP_TRY(hipMemsetAsync(covered, 0, (size_t)P * 4, s));
	UP_TRY(hipMemsetAsync(extra, 0, (size_t)N * 4, s));
	UP_TRY(hipMemsetAsync(ctx->leaf_boxes.ptr, 0, (size_t)P * 32, s));
	UP_TRY(hipMemsetAsync(order, 0, 4, s));           // the root
	UP_TRY(hipMemsetAsync(pair_of, 0, 4, s));         // its pair record, if it is an inner node
	UP_TRY(hipStreamSynchronize(s));
	const double t_copied = seconds_since(t_start);
	const Ref ref = {d_bounds, d_index, d_count, N, P};
	Flags host;

	// 1. the checks, level by level
	hipLaunchKernelGGL(k_check_prims, dim3(grid_for(P)), dim3(kT), 0, s, P, d_type, d_material, d_light, d.n_materials, d.n_lights,
	                   d_geom, flags);
	std::vector<int32_t> level_begin = {0};
	int32_t begin = 0, end = 1;
	while (begin < end) {
		level_begin.push_back(end);
		if ((int)level_begin.size() - 2 > kMaxDepth)
			return kUploadNotHandled;         // deeper than the traversal stack: the host path reports it
		hipLaunchKernelGGL(k_expand, dim3(grid_for(end - begin)), dim3(kT), 0, s, ref, begin, end, order, visited, covered,
		                   ctx->leaf_boxes.as<float4>(), extra, flags);
		UP_TRY(hipMemcpyAsync(&host, flags, sizeof host, hipMemcpyDeviceToHost, s));
		UP_TRY(hipStreamSynchronize(s));
		if (host.bad || host.big_leaf)
			return kUploadNotHandled;
		begin = end;
		end += (int32_t)host.next_count;
		UP_TRY(hipMemsetAsync(&flags->next_count, 0, 4, s));
	}
	const int32_t reached = end;
	const int levels = (int)level_begin.size() - 1;      // level_begin[l] .. level_begin[l + 1]: the nodes at depth l
	const int max_depth = levels - 1;

	// 2. the records
	for (int l = levels - 1; l >= 0; l--)
		hipLaunchKernelGGL(k_inner_counts, dim3(grid_for(level_begin[l + 1] - level_begin[l])), dim3(kT), 0, s, ref, level_begin[l],
		                   level_begin[l + 1], order, inner);
	for (int l = 0; l < levels; l++)
		hipLaunchKernelGGL(k_pair_numbers, dim3(grid_for(level_begin[l + 1] - level_begin[l])), dim3(kT), 0, s, ref, level_begin[l],
		                   level_begin[l + 1], order, inner, pair_of);
	int32_t n_inner = 0;
	UP_TRY(hipMemcpyAsync(&n_inner, inner, 4, hipMemcpyDeviceToHost, s));      // inner[0]: the root's
	UP_TRY(hipStreamSynchronize(s));
	const size_t n_pairs = std::max<size_t>((size_t)n_inner, 1);
	UP_TRY(ctx->pairs.reserve(n_pairs * 64));
	UP_TRY(ctx->prims.reserve((size_t)P * 16 * kPrimQuads));
	UP_TRY(ctx->shade_prims.reserve((size_t)P * 48));
	UP_TRY(ctx->ref_nodes.reserve((size_t)N * 32));
	if (n_inner == 0)
		UP_TRY(hipMemsetAsync(ctx->pairs.ptr, 0, 64, s));
	hipLaunchKernelGGL(k_pairs, dim3(grid_for(reached)), dim3(kT), 0, s, ref, reached, order, pair_of, ctx->pairs.as<float4>());
	hipLaunchKernelGGL(k_ref_nodes, dim3(grid_for(N)), dim3(kT), 0, s, ref, ctx->ref_nodes.as<float4>(), flags);
	hipLaunchKernelGGL(k_prims, dim3(grid_for(P)), dim3(kT), 0, s, P, d_type, d_material, d_light, d_geom, d_normals, covered,
	                   ctx->prims.as<float4>(), ctx->shade_prims.as<float4>());
	UP_TRY(hipMemcpyAsync(&host, flags, sizeof host, hipMemcpyDeviceToHost, s));
	UP_TRY(hipStreamSynchronize(s));
	UP_TRY(hipGetLastError());
	if (host.bad)
		return kUploadNotHandled;
	const bool tame = !host.not_tame;
	const double t_flat = seconds_since(t_start);

	// 3. the wide tree (the refusals of wide_tree.cpp, in its order)
	WideTree wide;      // (the scalars only; the arrays stay on the device)
	auto refuse = [&](const char *why) {
		wide.enabled = false;
		wide.why_not = why;
	};
	wide.enabled = true;
	std::vector<int32_t> wide_level = {0};
	int32_t n_wide = 0;
	int64_t n_covered = 0;
	if (!tame)
		refuse("box coordinates out of the ordinary range");
	else if (d.node_count[0] > 0)
		refuse("the root is a leaf");
	else if (d.n_prims >= (1 << 26))
		refuse("more primitives than a stack entry can number");
	else if (host.not_nested)
		refuse("a child's box reaches outside its parent's");
	else if (host.spheres)
		refuse("spheres in a scene beyond the size of the heuristic build");
	else if (host.nonfinite)
		refuse("non-finite vertex");
	if (wide.enabled) {
		// where the little trees of the cut leaves go: behind the reference's nodes, in node order
		size_t temp_bytes = 0;
		int32_t *extra_at = A.get<int32_t>(N);
		(void)rocprim::exclusive_scan(nullptr, temp_bytes, extra, extra_at, 0, (size_t)N, rocprim::plus<int32_t>(), s);
		void *temp = A.get<char>(temp_bytes);
		if (A.err != hipSuccess)
			return ctx->hip_fail(A.err, "hipMalloc (wide tree)");
		UP_TRY(rocprim::exclusive_scan(temp, temp_bytes, extra, extra_at, 0, (size_t)N, rocprim::plus<int32_t>(), s));
		int32_t last_at = 0, last_extra = 0;
		UP_TRY(hipMemcpyAsync(&last_at, extra_at + (N - 1), 4, hipMemcpyDeviceToHost, s));
		UP_TRY(hipMemcpyAsync(&last_extra, extra + (N - 1), 4, hipMemcpyDeviceToHost, s));
		UP_TRY(hipStreamSynchronize(s));
		const size_t NB = (size_t)N + (size_t)last_at + (size_t)last_extra;
		// triangles in leaves = leaves of the binary tree = (its nodes + 1) / 2 over the reached part
		n_covered = ((int64_t)reached + last_at + last_extra + 1) / 2;
		Binary b;
		b.lo = A.get<float>(NB * 3);
		b.hi = A.get<float>(NB * 3);
		b.left = A.get<int32_t>(NB);
		b.right = A.get<int32_t>(NB);
		b.prim = A.get<int32_t>(NB);
		b.edge = A.get<float>(NB);
		// at most one wide node per inner binary node
		const size_t max_wide = std::max<size_t>((size_t)n_covered, 1);
		int32_t *wide_at = A.get<int32_t>(max_wide), *kids = A.get<int32_t>(max_wide * 4), *child = A.get<int32_t>(max_wide * 4);
		int32_t *below_w = A.get<int32_t>(max_wide), *number = A.get<int32_t>(max_wide);
		if (A.err != hipSuccess)
			return ctx->hip_fail(A.err, "hipMalloc (wide tree)");
		if (n_covered < 2)
			refuse("fewer than two triangles");
		else {
			hipLaunchKernelGGL(k_leaf_trees, dim3(grid_for(reached)), dim3(kT), 0, s, ref, reached, order, d_geom, extra_at, b);
			for (int l = levels - 1; l >= 0; l--)
				hipLaunchKernelGGL(k_join_level, dim3(grid_for(level_begin[l + 1] - level_begin[l])), dim3(kT), 0, s, ref, level_begin[l],
				                   level_begin[l + 1], order, b);
			// the collapse, level by level
			UP_TRY(hipMemsetAsync(wide_at, 0, 4, s));          // the root stands on binary node 0
			UP_TRY(hipMemsetAsync(number, 0, 4, s));
			int32_t wb = 0, we = 1;
			while (wb < we) {
				wide_level.push_back(we);
				hipLaunchKernelGGL(k_wide_expand, dim3(grid_for(we - wb)), dim3(kT), 0, s, wb, we, wide_at, kids, child, b, flags);
				UP_TRY(hipMemcpyAsync(&host, flags, sizeof host, hipMemcpyDeviceToHost, s));
				UP_TRY(hipStreamSynchronize(s));
				wb = we;
				we += (int32_t)host.next_count;
				UP_TRY(hipMemsetAsync(&flags->next_count, 0, 4, s));
			}
			n_wide = we;
			const int wl = (int)wide_level.size() - 1;
			for (int l = wl - 1; l >= 0; l--)
				hipLaunchKernelGGL(k_wide_counts, dim3(grid_for(wide_level[l + 1] - wide_level[l])), dim3(kT), 0, s, wide_level[l],
				                   wide_level[l + 1], child, below_w);
			for (int l = 0; l < wl; l++)
				hipLaunchKernelGGL(k_wide_numbers, dim3(grid_for(wide_level[l + 1] - wide_level[l])), dim3(kT), 0, s, wide_level[l],
				                   wide_level[l + 1], child, below_w, number);
			// the scene's sphere, from the box of binary node 0
			float root_lo[3], root_hi[3];
			UP_TRY(hipMemcpyAsync(root_lo, b.lo, 12, hipMemcpyDeviceToHost, s));
			UP_TRY(hipMemcpyAsync(root_hi, b.hi, 12, hipMemcpyDeviceToHost, s));
			UP_TRY(hipStreamSynchronize(s));
			double c[3], r2 = 0, cmax = 0;
			for (int k = 0; k < 3; k++) {
				c[k] = .5 * ((double)root_lo[k] + root_hi[k]);
				const double h = .5 * ((double)root_hi[k] - root_lo[k]);
				r2 += h * h;
				cmax = std::max(cmax, std::fabs(c[k]));
			}
			const double radius = std::sqrt(r2);
			if (!(radius > 0) || !std::isfinite(radius) || !std::isfinite(cmax))
				refuse("degenerate scene box");
			else {
				for (int k = 0; k < 3; k++)
					wide.center[k] = (float)c[k];
				UP_TRY(ctx->wide_nodes.reserve((size_t)n_wide * 64));
				std::vector<double> candidates(std::begin(kWideReachRadii), std::end(kWideReachRadii));
				if (getenv("RTA_WIDE_REACH"))
					candidates = {atof(getenv("RTA_WIDE_REACH"))};
				for (size_t trial = 0; trial < candidates.size() && wide.enabled; trial++) {
					const double reach_radii = candidates[trial];
					Padding pad;
					pad.s_max = reach_radii * radius;
					pad.cmax = cmax;
					pad.radius = radius;
					for (int k = 0; k < 3; k++)
						pad.center[k] = (double)wide.center[k];
					wide.reach = below((reach_radii - 1.001) * radius - 1e-6 * cmax);
					wide.span = (float)(pad.s_max * 1.001);
					if (!(wide.reach > 0)) {
						refuse("degenerate scene box");
						break;
					}
					UP_TRY(hipMemsetAsync(&flags->overflow, 0, 4 + 2 * sizeof(unsigned long long), s));      // overflow and the two sums behind it
					hipLaunchKernelGGL(k_wide_emit, dim3(grid_for(n_wide)), dim3(kT), 0, s, n_wide, kids, child, number, b, pad,
					                   ctx->wide_nodes.as<float4>(), flags);
					UP_TRY(hipMemcpyAsync(&host, flags, sizeof host, hipMemcpyDeviceToHost, s));
					UP_TRY(hipStreamSynchronize(s));
					if (host.overflow) {
						refuse("coordinates beyond the range of the wide tree's binary16 planes");
						break;
					}
					wide.mean_pad_over_extent = host.widened_n ? (double)host.widened_sum * 0x1p-32 / (double)host.widened_n : 0;
					if (wide.mean_pad_over_extent <= kWideMaxPadding)
						break;
				}
				if (wide.enabled && wide.mean_pad_over_extent > kWideHopelessPadding)
					refuse("triangles too large for the padding the walk needs");
			}
		}
		UP_TRY(hipGetLastError());
		if (wide.enabled) {
			wide.depth = (int)wide_level.size() - 1;
			wide.n_nodes = (uint32_t)n_wide;
			wide.index_mask = 1;
			while (wide.index_mask < (uint32_t)d.n_prims || wide.index_mask < wide.n_nodes)
				wide.index_mask = wide.index_mask * 2 + 1;
		}
	}
	UP_TRY(hipStreamSynchronize(s));

	// the context's view of the scene, as rta_upload_scene's host path sets it
	ctx->have_wide = wide.enabled;
	ctx->have_leaf_boxes = wide.enabled;      // (this path certifies by the wide tree's conditions only)
	ctx->have_flat_boxes = false;            // (and serves scenes far beyond the size that is scanned)
	if (wide.enabled) {
		memcpy(ctx->wide_center, wide.center, 12);
		ctx->wide_reach2 = wide.reach * wide.reach * (1.f - 0x1p-20f);
		ctx->wide_reach = wide.reach;
		ctx->wide_depth = wide.depth;
		ctx->wide_index_mask = wide.index_mask;
		ctx->wide_n_nodes = wide.n_nodes;
		ctx->wide_span = wide.span;
		ctx->n_loose = 0;
	}
	ctx->root_ref = d.node_count[0] > 0 ? (kLeaf | (uint32_t)d.node_index[0]) : 0u;
	ctx->n_pairs = (uint32_t)n_pairs;
	ctx->tame = tame;
	// the box around the root's children (the root's own box is never tested, src/main.cpp:332-350)
	ctx->box_known = false;
	memset(ctx->box, 0, sizeof ctx->box);
	if (tame && d.node_count[0] == 0) {
		const float *a = d.node_bounds + (size_t)d.node_index[0] * 6, *bb = a + 6;
		for (int k = 0; k < 3; k++) {
			ctx->box[k] = std::min(a[k], bb[k]);
			ctx->box[3 + k] = std::max(a[3 + k], bb[3 + k]);
		}
		ctx->box_known = true;
	}
	ctx->stack_depth = std::max(max_depth, 1);
	if (verbose)
		fprintf(stderr, "scene upload on the device: %.3f s (arrays copied after %.3f s, records made after %.3f s); wide tree: %s%s, %u nodes, "
		                "depth %d, padding/extent %.4f, reach %g\n",
		        seconds_since(t_start), t_copied, t_flat, wide.enabled ? "built" : "none: ", wide.why_not.c_str(), wide.n_nodes, wide.depth,
		        wide.mean_pad_over_extent, wide.reach);
	return kUploadDone;
}

}  // namespace rta
