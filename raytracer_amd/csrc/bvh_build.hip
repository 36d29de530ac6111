// The reference's BVH builder (src/main.cpp:133-248) on the GPU, producing the
// SAME tree: node numbering, boxes (signed zeros included) and primitive order
// are bit-identical to the host builder in host_scene.cpp.
//
// The reference builds depth-first with an in-place two-pointer partition.  Both
// are reproduced level by level:
//
// * The partition `while (i <= j) if (c[i] < split) i++; else swap(a[i], a[j--])`
//   has a closed form.  With nL elements below the split ("L", the others "R"),
//   holes p_1 < ... < p_K = the R elements at positions < nL, and m_1 > ... > m_K
//   = the L elements at positions >= nL (m_0 = one past the end):
//       an L below nL stays; the L at m_k goes to p_k;
//       the R at p_k goes to m_(k-1) - 1; the R at position nL (if it is one)
//       goes to m_K - 1; every other R moves down by one.
//   Ranks come from one exclusive scan of the R flags over the whole primitive
//   array per level, so all nodes of a level are partitioned by the same launches.
// * A node's box folds its primitives' boxes in order with `a < b ? a : b`, so
//   among equal candidates (+0 and -0 are the only ones that differ in bits) the
//   LAST one wins.  Boxes are reduced with 64-bit atomics on (value, position)
//   keys that encode exactly that rule.
// * Depth-first node numbers follow from subtree sizes once the topology is
//   known: the children of a node get the two numbers that are free when the
//   node is split, and the left subtree is numbered before the right one.
//
// Inputs with non-finite coordinates are refused (the fold's NaN behaviour is
// not reproduced); the caller then uses the host builder.

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <string>
#include <vector>

#include "bvh_build.hpp"
#include "raytracer_amd.h"

namespace rta {

namespace {

constexpr int kThreads = 256;
constexpr int kScanItems = 16;                       // per thread
constexpr int kScanTile = kThreads * kScanItems;     // per workgroup
constexpr unsigned long long kLoInit = ~0ull;
constexpr unsigned long long kHiInit = 0ull;

struct Nodes {
	int32_t *first, *count;       // primitive range
	int32_t *left;                // first child (the second follows it) or -1
	int32_t *lturns;              // ancestors entered as a first child: the reference's pending-stack depth
	int32_t *axis;                // split axis, or -1 when the node is not split
	float *split;
	int32_t *n_left_prims;
	unsigned long long *lo_key, *hi_key;     // 3 each
	float *bounds;                // 6 each
	int32_t *splits;              // split nodes in the subtree
	int32_t *id, *next_free;      // depth-first number; first free number when the node is reached
};

__device__ inline float lesser(float a, float b) { return a < b ? a : b; }
__device__ inline float greater(float a, float b) { return a > b ? a : b; }

// Per primitive: box and centroid as src/main.cpp:133-168,212.
__global__ void k_prepare(int32_t n, const int32_t *type, const float *geom, float *plo, float *phi, float *cen,
                          int32_t *order, int32_t *seg, int32_t *bad)
{
	const int32_t k = blockIdx.x * kThreads + threadIdx.x;
	if (k >= n)
		return;
	const float *g = geom + (size_t)k * 9;
	float lo[3], hi[3], c[3];
	if (type[k] == RTA_PRIM_SPHERE) {
		const float r = g[3];
		for (int a = 0; a < 3; a++) {
			lo[a] = g[a] - r;
			hi[a] = g[a] + r;
			c[a] = g[a];
		}
	} else {
		for (int a = 0; a < 3; a++) {
			const float v0 = g[a], v1 = g[3 + a], v2 = g[6 + a];
			lo[a] = lesser(lesser(v0, v1), v2);
			hi[a] = greater(greater(v0, v1), v2);
			c[a] = ((v0 + v1) + v2) * (1.f / 3.f);
		}
	}
	bool finite = true;
	for (int a = 0; a < 3; a++) {
		plo[(size_t)k * 3 + a] = lo[a];
		phi[(size_t)k * 3 + a] = hi[a];
		cen[(size_t)k * 3 + a] = c[a];
		finite = finite && __builtin_isfinite(lo[a]) && __builtin_isfinite(hi[a]) && __builtin_isfinite(c[a]);
	}
	if (!finite)
		*bad = 1;
	order[k] = k;
	seg[k] = 0;
}

// Unsigned key with the order of the floats, +0 and -0 alike.
__device__ inline uint32_t order_key(float v)
{
	if (v == 0.f)
		v = 0.f;
	const uint32_t u = __float_as_uint(v);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ inline unsigned long long shfl_xor64(unsigned long long v, int m)
{
	const uint32_t lo = __shfl_xor((uint32_t)v, m), hi = __shfl_xor((uint32_t)(v >> 32), m);
	return ((unsigned long long)hi << 32) | lo;
}

// Box of every node created by the previous level: smallest lo / largest hi, and
// among equal values the one latest in primitive order.
__global__ void k_bounds(int32_t n, const int32_t *order, const int32_t *seg, const float *plo, const float *phi, Nodes nd)
{
	const int32_t pos = blockIdx.x * kThreads + threadIdx.x;
	const int32_t node = pos < n ? seg[pos] : -1;
	unsigned long long lo[3], hi[3];
	if (node >= 0) {
		const int32_t k = order[pos];
		for (int a = 0; a < 3; a++) {
			lo[a] = ((unsigned long long)order_key(plo[(size_t)k * 3 + a]) << 32) | (0xffffffffu - (uint32_t)pos);
			hi[a] = ((unsigned long long)order_key(phi[(size_t)k * 3 + a]) << 32) | (uint32_t)pos;
		}
	} else {
		for (int a = 0; a < 3; a++) {
			lo[a] = kLoInit;
			hi[a] = kHiInit;
		}
	}
	// a wave whose lanes all belong to one node sends one set of atomics
	const int32_t node0 = __shfl(node, 0);
	if (__ballot(node != node0) == 0) {
		if (node0 < 0)
			return;
		for (int m = 32; m >= 1; m >>= 1)
			for (int a = 0; a < 3; a++) {
				const unsigned long long l = shfl_xor64(lo[a], m), h = shfl_xor64(hi[a], m);
				lo[a] = l < lo[a] ? l : lo[a];
				hi[a] = h > hi[a] ? h : hi[a];
			}
		if ((threadIdx.x & 63) == 0)
			for (int a = 0; a < 3; a++) {
				atomicMin(&nd.lo_key[(size_t)node0 * 3 + a], lo[a]);
				atomicMax(&nd.hi_key[(size_t)node0 * 3 + a], hi[a]);
			}
	} else if (node >= 0) {
		for (int a = 0; a < 3; a++) {
			atomicMin(&nd.lo_key[(size_t)node * 3 + a], lo[a]);
			atomicMax(&nd.hi_key[(size_t)node * 3 + a], hi[a]);
		}
	}
}

// Per new node: the box from the winning positions, then the reference's split
// decision (src/main.cpp:187-200): longest axis, first of equals, at its midpoint.
__global__ void k_decide(int32_t begin, int32_t end, const int32_t *order, const float *plo, const float *phi, Nodes nd)
{
	const int32_t node = begin + blockIdx.x * kThreads + threadIdx.x;
	if (node >= end)
		return;
	float b[6];
	for (int a = 0; a < 3; a++) {
		const uint32_t pl = 0xffffffffu - (uint32_t)nd.lo_key[(size_t)node * 3 + a];
		const uint32_t ph = (uint32_t)nd.hi_key[(size_t)node * 3 + a];
		b[a] = plo[(size_t)order[pl] * 3 + a];
		b[3 + a] = phi[(size_t)order[ph] * 3 + a];
	}
	for (int a = 0; a < 6; a++)
		nd.bounds[(size_t)node * 6 + a] = b[a];
	int axis = -1;
	float split = 0.f;
	// a node reached with 64 second children pending is given up as a leaf (src/main.cpp:182-185)
	if (nd.count[node] > 2 && nd.lturns[node] < 64) {
		const float ext[3] = {b[3] - b[0], b[4] - b[1], b[5] - b[2]};
		axis = 0;
		if (ext[1] > ext[0])
			axis = 1;
		if (ext[2] > ext[axis])
			axis = 2;
		split = b[axis] + ext[axis] * .5f;
	}
	nd.axis[node] = axis;
	nd.split[node] = split;
	nd.left[node] = -1;
}

// 1 where the primitive goes to the far side of its node's split plane.
__global__ void k_flags(int32_t n, const int32_t *order, const int32_t *seg, const float *cen, Nodes nd, uint32_t *flags)
{
	const int32_t pos = blockIdx.x * kThreads + threadIdx.x;
	if (pos >= n)
		return;
	const int32_t node = seg[pos];
	uint32_t f = 0;
	if (node >= 0) {
		const int axis = nd.axis[node];
		if (axis >= 0)
			f = cen[(size_t)order[pos] * 3 + axis] < nd.split[node] ? 0u : 1u;
	}
	flags[pos] = f;
}

// Exclusive prefix sum of `flags` into sums[0..n] in three launches.
__global__ void k_scan_tiles(int32_t n, const uint32_t *flags, uint32_t *tile_sums)
{
	__shared__ uint32_t part[kThreads];
	const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
	uint32_t s = 0;
	for (int k = 0; k < kScanItems; k++)
		if (base + k < (size_t)n)
			s += flags[base + k];
	part[threadIdx.x] = s;
	__syncthreads();
	for (int w = kThreads / 2; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w)
			part[threadIdx.x] += part[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0)
		tile_sums[blockIdx.x] = part[0];
}

__global__ void k_scan_tile_sums(int32_t n_tiles, uint32_t *tile_sums)
{
	// one workgroup: a running total over chunks of kThreads tiles
	__shared__ uint32_t part[kThreads];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0)
		carry = 0;
	__syncthreads();
	for (int32_t base = 0; base < n_tiles; base += kThreads) {
		const int32_t i = base + threadIdx.x;
		const uint32_t v = i < n_tiles ? tile_sums[i] : 0u;
		part[threadIdx.x] = v;
		__syncthreads();
		for (int d = 1; d < kThreads; d <<= 1) {
			const uint32_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
			__syncthreads();
			part[threadIdx.x] += add;
			__syncthreads();
		}
		if (i < n_tiles)
			tile_sums[i] = carry + part[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == kThreads - 1)
			carry += part[threadIdx.x];
		__syncthreads();
	}
}

__global__ void k_scan_finish(int32_t n, const uint32_t *flags, const uint32_t *tile_sums, uint32_t *sums)
{
	__shared__ uint32_t part[kThreads];
	const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
	uint32_t local[kScanItems];
	uint32_t s = 0;
	for (int k = 0; k < kScanItems; k++) {
		local[k] = s;
		if (base + k < (size_t)n)
			s += flags[base + k];
	}
	part[threadIdx.x] = s;
	__syncthreads();
	for (int d = 1; d < kThreads; d <<= 1) {
		const uint32_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
		__syncthreads();
		part[threadIdx.x] += add;
		__syncthreads();
	}
	const uint32_t before = tile_sums[blockIdx.x] + part[threadIdx.x] - s;
	for (int k = 0; k < kScanItems; k++)
		if (base + k <= (size_t)n)          // sums has n + 1 entries
			sums[base + k] = before + local[k];
}

// Per new node: primitives on the near side; a proper split gets two new nodes.
__global__ void k_children(int32_t begin, int32_t end, const uint32_t *sums, Nodes nd, int32_t *node_total)
{
	const int32_t node = begin + blockIdx.x * kThreads + threadIdx.x;
	if (node >= end || nd.axis[node] < 0)
		return;
	const int32_t first = nd.first[node], count = nd.count[node];
	const int32_t n_right = (int32_t)(sums[first + count] - sums[first]);
	const int32_t n_left = count - n_right;
	nd.n_left_prims[node] = n_left;
	if (n_left == 0 || n_left == count)
		return;                         // stays a leaf; its primitives are permuted all the same
	const int32_t c = atomicAdd(node_total, 2);
	nd.left[node] = c;
	for (int side = 0; side < 2; side++) {
		const int32_t ch = c + side;
		nd.first[ch] = side ? first + n_left : first;
		nd.count[ch] = side ? n_right : n_left;
		nd.lturns[ch] = nd.lturns[node] + (side ? 0 : 1);
		nd.left[ch] = -1;
		nd.axis[ch] = -1;
		for (int a = 0; a < 3; a++) {
			nd.lo_key[(size_t)ch * 3 + a] = kLoInit;
			nd.hi_key[(size_t)ch * 3 + a] = kHiInit;
		}
	}
}

struct Place {
	int32_t node, first, count, n_left;
	int32_t r_before;       // far-side primitives of the node before this position
	bool far;
};

__device__ inline bool place_of(int32_t pos, const int32_t *seg, const uint32_t *flags, const uint32_t *sums,
                                const Nodes &nd, Place &p)
{
	p.node = seg[pos];
	if (p.node < 0 || nd.axis[p.node] < 0)
		return false;
	p.first = nd.first[p.node];
	p.count = nd.count[p.node];
	p.n_left = nd.n_left_prims[p.node];
	p.r_before = (int32_t)(sums[pos] - sums[p.first]);
	p.far = flags[pos] != 0;
	return true;
}

// Positions of the holes (far-side primitives below the boundary, in ascending
// order) and of the misplaced near-side primitives (from the top down), each
// list stored at the start of the node's own range.
__global__ void k_tables(int32_t n, const int32_t *seg, const uint32_t *flags, const uint32_t *sums, Nodes nd,
                         int32_t *holes, int32_t *misplaced)
{
	const int32_t pos = blockIdx.x * kThreads + threadIdx.x;
	if (pos >= n)
		return;
	Place p;
	if (!place_of(pos, seg, flags, sums, nd, p))
		return;
	const int32_t boundary = p.first + p.n_left;
	if (pos < boundary) {
		if (p.far)
			holes[p.first + p.r_before] = pos;
	} else if (!p.far) {
		const int32_t l_before = (pos - p.first) - p.r_before;
		misplaced[p.first + (p.n_left - l_before - 1)] = pos;
	}
}

__global__ void k_move(int32_t n, const int32_t *order, const int32_t *seg, const uint32_t *flags, const uint32_t *sums,
                       Nodes nd, const int32_t *holes, const int32_t *misplaced, int32_t *order_out, int32_t *seg_out)
{
	const int32_t pos = blockIdx.x * kThreads + threadIdx.x;
	if (pos >= n)
		return;
	Place p;
	if (!place_of(pos, seg, flags, sums, nd, p)) {
		order_out[pos] = order[pos];
		seg_out[pos] = -1;
		return;
	}
	const int32_t boundary = p.first + p.n_left, end = p.first + p.count;
	const int32_t n_holes = (int32_t)(sums[boundary] - sums[p.first]);
	int32_t dest;
	if (pos < boundary) {
		if (!p.far) {
			dest = pos;
		} else {
			const int32_t k = p.r_before;          // hole number, from 0
			dest = (k == 0 ? end : misplaced[p.first + k - 1]) - 1;
		}
	} else if (!p.far) {
		const int32_t l_before = (pos - p.first) - p.r_before;
		dest = holes[p.first + (p.n_left - l_before - 1)];
	} else if (pos == boundary) {
		dest = (n_holes == 0 ? end : misplaced[p.first + n_holes - 1]) - 1;
	} else {
		dest = pos - 1;
	}
	order_out[dest] = order[pos];
	const int32_t left = nd.left[p.node];
	seg_out[dest] = left < 0 ? -1 : (dest < boundary ? left : left + 1);
}

// Split nodes per subtree, one level at a time from the bottom.
__global__ void k_count_splits(int32_t begin, int32_t end, Nodes nd)
{
	const int32_t node = begin + blockIdx.x * kThreads + threadIdx.x;
	if (node >= end)
		return;
	const int32_t l = nd.left[node];
	nd.splits[node] = l < 0 ? 0 : 1 + nd.splits[l] + nd.splits[l + 1];
}

// Depth-first numbers, one level at a time from the top (the root has number 0
// and finds number 1 free).
__global__ void k_number(int32_t begin, int32_t end, Nodes nd)
{
	const int32_t node = begin + blockIdx.x * kThreads + threadIdx.x;
	if (node >= end)
		return;
	const int32_t l = nd.left[node];
	if (l < 0)
		return;
	const int32_t free0 = nd.next_free[node];
	nd.id[l] = free0;
	nd.id[l + 1] = free0 + 1;
	nd.next_free[l] = free0 + 2;
	nd.next_free[l + 1] = free0 + 2 + 2 * nd.splits[l];
}

__global__ void k_emit(int32_t total, Nodes nd, float *out_bounds, int32_t *out_index, int32_t *out_count)
{
	const int32_t node = blockIdx.x * kThreads + threadIdx.x;
	if (node >= total)
		return;
	const int32_t id = nd.id[node], l = nd.left[node];
	for (int a = 0; a < 6; a++)
		out_bounds[(size_t)id * 6 + a] = nd.bounds[(size_t)node * 6 + a];
	out_index[id] = l < 0 ? nd.first[node] : nd.id[l];
	out_count[id] = l < 0 ? nd.count[node] : 0;
}

__global__ void k_init_root(int32_t n, Nodes nd, int32_t *node_total, int32_t *bad)
{
	nd.first[0] = 0;
	nd.count[0] = n;
	nd.lturns[0] = 0;
	nd.left[0] = -1;
	nd.axis[0] = -1;
	nd.id[0] = 0;
	nd.next_free[0] = 1;
	for (int a = 0; a < 3; a++) {
		nd.lo_key[a] = kLoInit;
		nd.hi_key[a] = kHiInit;
	}
	*node_total = 1;
	*bad = 0;
}

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

inline unsigned grid_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }

}  // namespace

std::string device_build_bvh(hipStream_t s, int32_t n, const int32_t *prim_type, const float *geom, BuiltTree &out,
                             bool &unsupported)
{
	unsupported = false;
	if (n <= 0)
		return "no primitives";
	const size_t cap = (size_t)n * 2;       // a binary tree over n primitives has fewer than 2n nodes
	Arena A;
	int32_t *d_type = A.get<int32_t>(n);
	float *d_geom = A.get<float>((size_t)n * 9);
	float *plo = A.get<float>((size_t)n * 3), *phi = A.get<float>((size_t)n * 3), *cen = A.get<float>((size_t)n * 3);
	int32_t *order[2] = {A.get<int32_t>(n), A.get<int32_t>(n)};
	int32_t *seg[2] = {A.get<int32_t>(n), A.get<int32_t>(n)};
	uint32_t *flags = A.get<uint32_t>(n), *sums = A.get<uint32_t>((size_t)n + 1);
	const int32_t n_tiles = (int32_t)(((size_t)n + 1 + kScanTile - 1) / kScanTile);
	uint32_t *tile_sums = A.get<uint32_t>(n_tiles);
	int32_t *holes = A.get<int32_t>(n), *misplaced = A.get<int32_t>(n);
	int32_t *counters = A.get<int32_t>(2);          // node total, "non-finite input"
	Nodes nd;
	nd.first = A.get<int32_t>(cap);
	nd.count = A.get<int32_t>(cap);
	nd.left = A.get<int32_t>(cap);
	nd.lturns = A.get<int32_t>(cap);
	nd.axis = A.get<int32_t>(cap);
	nd.split = A.get<float>(cap);
	nd.n_left_prims = A.get<int32_t>(cap);
	nd.lo_key = A.get<unsigned long long>(cap * 3);
	nd.hi_key = A.get<unsigned long long>(cap * 3);
	nd.bounds = A.get<float>(cap * 6);
	nd.splits = A.get<int32_t>(cap);
	nd.id = A.get<int32_t>(cap);
	nd.next_free = A.get<int32_t>(cap);
	float *out_bounds = A.get<float>(cap * 6);
	int32_t *out_index = A.get<int32_t>(cap), *out_count = A.get<int32_t>(cap);
	if (A.err != hipSuccess)
		return std::string("hipMalloc: ") + hipGetErrorString(A.err);

#define BUILD_TRY(call)                                                          \
	do {                                                                         \
		hipError_t e__ = (call);                                                 \
		if (e__ != hipSuccess)                                                   \
			return std::string(#call) + ": " + hipGetErrorString(e__);           \
	} while (0)

	BUILD_TRY(hipMemcpyAsync(d_type, prim_type, (size_t)n * 4, hipMemcpyHostToDevice, s));
	BUILD_TRY(hipMemcpyAsync(d_geom, geom, (size_t)n * 36, hipMemcpyHostToDevice, s));
	// destroyed on every way out of this function
	struct Events {
		hipEvent_t a = nullptr, b = nullptr;
		~Events()
		{
			if (a) (void)hipEventDestroy(a);
			if (b) (void)hipEventDestroy(b);
		}
	} ev;
	BUILD_TRY(hipEventCreate(&ev.a));
	BUILD_TRY(hipEventCreate(&ev.b));
	(void)hipEventRecord(ev.a, s);

	hipLaunchKernelGGL(k_init_root, dim3(1), dim3(1), 0, s, n, nd, counters, counters + 1);
	hipLaunchKernelGGL(k_prepare, dim3(grid_for(n)), dim3(kThreads), 0, s, n, d_type, d_geom, plo, phi, cen, order[0],
	                   seg[0], counters + 1);

	std::vector<int32_t> level_begin = {0};      // first node of every level, then the total
	int32_t begin = 0, end = 1;
	int cur = 0;
	int32_t host_counters[2] = {1, 0};
	while (begin < end) {
		level_begin.push_back(end);
		const unsigned gn = grid_for(n), gl = grid_for(end - begin);
		hipLaunchKernelGGL(k_bounds, dim3(gn), dim3(kThreads), 0, s, n, order[cur], seg[cur], plo, phi, nd);
		hipLaunchKernelGGL(k_decide, dim3(gl), dim3(kThreads), 0, s, begin, end, order[cur], plo, phi, nd);
		hipLaunchKernelGGL(k_flags, dim3(gn), dim3(kThreads), 0, s, n, order[cur], seg[cur], cen, nd, flags);
		hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(kThreads), 0, s, n, flags, tile_sums);
		hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kThreads), 0, s, n_tiles, tile_sums);
		hipLaunchKernelGGL(k_scan_finish, dim3(n_tiles), dim3(kThreads), 0, s, n, flags, tile_sums, sums);
		hipLaunchKernelGGL(k_children, dim3(gl), dim3(kThreads), 0, s, begin, end, sums, nd, counters);
		hipLaunchKernelGGL(k_tables, dim3(gn), dim3(kThreads), 0, s, n, seg[cur], flags, sums, nd, holes, misplaced);
		hipLaunchKernelGGL(k_move, dim3(gn), dim3(kThreads), 0, s, n, order[cur], seg[cur], flags, sums, nd, holes,
		                   misplaced, order[cur ^ 1], seg[cur ^ 1]);
		cur ^= 1;
		BUILD_TRY(hipMemcpyAsync(host_counters, counters, 8, hipMemcpyDeviceToHost, s));
		BUILD_TRY(hipStreamSynchronize(s));
		if (host_counters[1]) {
			unsupported = true;
			return "non-finite primitive coordinates";
		}
		begin = end;
		end = host_counters[0];
		if ((size_t)end > cap)
			return "node count exceeds 2n";
	}
	const int32_t total = end;
	const int levels = (int)level_begin.size() - 1;
	for (int l = levels - 1; l >= 0; l--)
		hipLaunchKernelGGL(k_count_splits, dim3(grid_for(level_begin[l + 1] - level_begin[l])), dim3(kThreads), 0, s,
		                   level_begin[l], level_begin[l + 1], nd);
	for (int l = 0; l < levels; l++)
		hipLaunchKernelGGL(k_number, dim3(grid_for(level_begin[l + 1] - level_begin[l])), dim3(kThreads), 0, s,
		                   level_begin[l], level_begin[l + 1], nd);
	hipLaunchKernelGGL(k_emit, dim3(grid_for(total)), dim3(kThreads), 0, s, total, nd, out_bounds, out_index, out_count);
	(void)hipEventRecord(ev.b, s);
	BUILD_TRY(hipGetLastError());

	out.perm.resize(n);
	out.node_bounds.resize((size_t)total * 6);
	out.node_index.resize(total);
	out.node_count.resize(total);
	BUILD_TRY(hipMemcpyAsync(out.perm.data(), order[cur], (size_t)n * 4, hipMemcpyDeviceToHost, s));
	BUILD_TRY(hipMemcpyAsync(out.node_bounds.data(), out_bounds, (size_t)total * 24, hipMemcpyDeviceToHost, s));
	BUILD_TRY(hipMemcpyAsync(out.node_index.data(), out_index, (size_t)total * 4, hipMemcpyDeviceToHost, s));
	BUILD_TRY(hipMemcpyAsync(out.node_count.data(), out_count, (size_t)total * 4, hipMemcpyDeviceToHost, s));
	BUILD_TRY(hipStreamSynchronize(s));
	float ms = 0;
	(void)hipEventElapsedTime(&ms, ev.a, ev.b);
	out.device_ms = ms;
	out.levels = levels;
#undef BUILD_TRY
	return "";
}

}  // namespace rta
