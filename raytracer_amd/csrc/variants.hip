// Design-space probes: the closest-hit traversal with ONE design decision
// changed at a time, so that the decision can be priced on the hardware
// (tools/probe_layouts.py runs them under rocprofv3).  Every variant returns
// the same bits as the production kernel; none is used by the renderer.
//
//   1  traversal stack in private (scratch) memory instead of LDS
//   2  division-based box test for every ray (six IEEE divisions per box)
//      instead of the reciprocal + two fused corrections
//   3  the reference's node array as it is (32-byte nodes, src/main.cpp:20-23):
//      a visit loads the two children's records from index and index+1, the
//      stack holds node numbers, and a popped node's (index, count) is re-read
//   4  two nested loops per ray (walk inner nodes down to a leaf, then test all
//      of the leaf's primitives) instead of one loop whose every trip is one
//      node visit or one primitive test
//   5  struct-of-arrays nodes: the 14 words of a pair record in 14 arrays, so
//      that a visit issues 14 four-byte gathers instead of four 16-byte loads
//      from one 64-byte line (one ray per lane: neighbouring lanes are at
//      unrelated nodes, so the arrays buy no coalescing)
//   100        not a variant but a census: per ray the trips of its wave, its own
//              units of work, and the wave's trips that ran both branches
//   101..164   the wave runs a branch only when at least (variant - 100) lanes
//              want it or it is the majority's (1 = always: the production loop
//              with each branch loading its own record)
//   200 + d    as 101 with the LDS stack sized for d entries: fewer workgroups
//              per CU, i.e. what occupancy is worth (d >= the tree's depth)

#include <cstdlib>

#include "probe.hpp"
#include "shading.hpp"
#include "traverse.hpp"
#include "traverse_wide.hpp"

namespace rta {

namespace {

__global__ __launch_bounds__(kBlock) void k_variant_private_stack(const DeviceTree tree, const float *rays, uint64_t n,
                                                                  rta_hit *hits)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	uint32_t stack[64];
	const float *r = rays + i * 6;
	Traversal<false, false, false, 1> tr;
	WorkCount wc = {0, 0, 0};
	tr.start(tree, Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])}, FLT_MAX);
	while (__ballot(tr.more)) {
		if (tr.more)
			tr.step(tree, stack, 64, wc);
	}
	const HitRec h = tr.result();
	hits[i] = rta_hit{h.prim, h.t, h.u, h.v};
}

// LDS stack, but every box test by division: the production code path for
// rays that are not tame, applied to all rays.
__global__ __launch_bounds__(kBlock) void k_variant_division(DeviceTree tree, const float *rays, uint64_t n,
                                                             rta_hit *hits, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	tree.tame = 0;
	const float *r = rays + i * 6;
	HitRec h;
	WorkCount wc = {0, 0, 0};
	traverse_dyn<false, false>(tree, Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])}, FLT_MAX,
	                           lds_stack + threadIdx.x, depth, h, wc);
	hits[i] = rta_hit{h.prim, h.t, h.u, h.v};
}

// Reference node layout: nodes[k] = (min.xyz, max.x) (max.yz, index, count).
__global__ __launch_bounds__(kBlock) void k_variant_reference_nodes(const DeviceTree tree, const float4 *nodes,
                                                                    const float *rays, uint64_t n, rta_hit *hits,
                                                                    const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	uint32_t *stack = lds_stack + threadIdx.x;
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float *r = rays + i * 6;
	const Ray ray = {mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
	const bool tame = tree.tame != 0 && ray_is_tame(ray);
	const RaySigns sg = signs_of(ray);
	const V3 rcp = mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
	float best = FLT_MAX, bu = 0, bv = 0;
	int32_t prim = -1;
	uint32_t node = 0;
	int sp = 0;
	for (;;) {
		const float4 me = nodes[2 * (size_t)node + 1];          // (.., .., index, count): the re-read
		const int32_t index = __float_as_int(me.z), count = __float_as_int(me.w);
		if (count > 0) {
			for (int32_t k = index; k < index + count; k++) {
				const float4 p0 = tree.prims[kPrimQuads * (size_t)k], p1 = tree.prims[kPrimQuads * (size_t)k + 1];
				float t, tu = 0, tv = 0;
				if (__float_as_uint(p0.w) & kPrimSphere) {
					t = test_sphere(mk(p0.x, p0.y, p0.z), p1.x, ray);
				} else {
					const float4 p2 = tree.prims[kPrimQuads * (size_t)k + 2];
					t = test_triangle(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), ray, tu, tv);
				}
				if (t < best) {
					best = t;
					prim = k;
					bu = tu;
					bv = tv;
				}
			}
			if (sp == 0)
				break;
			node = stack[(--sp) * kBlock];
		} else {
			const float4 a0 = nodes[2 * (size_t)index], a1 = nodes[2 * (size_t)index + 1];
			const float4 b0 = nodes[2 * (size_t)index + 2], b1 = nodes[2 * (size_t)index + 3];
			float d1, d2;
			if (tame) {
				d1 = test_box_tame(mk(a0.x, a0.y, a0.z), mk(a0.w, a1.x, a1.y), ray, rcp, best);
				d2 = test_box_tame(mk(b0.x, b0.y, b0.z), mk(b0.w, b1.x, b1.y), ray, rcp, best);
			} else {
				d1 = test_box(mk(a0.x, a0.y, a0.z), mk(a0.w, a1.x, a1.y), ray, sg, best);
				d2 = test_box(mk(b0.x, b0.y, b0.z), mk(b0.w, b1.x, b1.y), ray, sg, best);
			}
			const bool one_near = d1 < d2;
			const uint32_t first = one_near ? (uint32_t)index : (uint32_t)index + 1;
			const uint32_t second = one_near ? (uint32_t)index + 1 : (uint32_t)index;
			const bool hit_first = (one_near ? d1 : d2) < FLT_MAX, hit_second = (one_near ? d2 : d1) < FLT_MAX;
			if (hit_first) {
				node = first;
				if (hit_second && sp < depth)
					stack[(sp++) * kBlock] = second;
			} else if (hit_second) {
				node = second;
			} else {
				if (sp == 0)
					break;
				node = stack[(--sp) * kBlock];
			}
		}
	}
	hits[i] = rta_hit{prim, best, bu, bv};
}

// Production records and stack, but the ray walks inner nodes until it holds a
// leaf and then tests the whole leaf: in a wave, every lane waits for the
// slowest lane of each of the two inner loops.
__global__ __launch_bounds__(kBlock) void k_variant_nested_loops(const DeviceTree tree, const float *rays, uint64_t n,
                                                                 rta_hit *hits, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	uint32_t *stack = lds_stack + threadIdx.x;
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float *r = rays + i * 6;
	const Ray ray = {mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
	const bool tame = tree.tame != 0 && ray_is_tame(ray);
	const RaySigns sg = signs_of(ray);
	const V3 rcp = mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
	float best = FLT_MAX, bu = 0, bv = 0;
	int32_t prim = -1;
	uint32_t node = tree.root;
	int sp = 0;
	bool more = true;
	while (more) {
		while (more && !(node & kLeafBit)) {
			const float4 *rec = tree.pairs + 4 * (size_t)node;
			const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
			const uint32_t ref1 = __float_as_uint(q3.x), ref2 = __float_as_uint(q3.y);
			float d1, d2;
			if (tame) {
				d1 = test_box_tame(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), ray, rcp, best);
				d2 = test_box_tame(mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w), ray, rcp, best);
			} else {
				d1 = test_box(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), ray, sg, best);
				d2 = test_box(mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w), ray, sg, best);
			}
			const bool one_near = d1 < d2;
			const uint32_t first = one_near ? ref1 : ref2, second = one_near ? ref2 : ref1;
			const bool hit_first = (one_near ? d1 : d2) < FLT_MAX, hit_second = (one_near ? d2 : d1) < FLT_MAX;
			if (hit_first) {
				node = first;
				if (hit_second && sp < depth)
					stack[(sp++) * kBlock] = second;
			} else if (hit_second) {
				node = second;
			} else if (sp > 0) {
				node = stack[(--sp) * kBlock];
			} else {
				more = false;
			}
		}
		if (!more)
			break;
		uint32_t k = node & ~kLeafBit;
		for (;;) {
			const float4 p0 = tree.prims[kPrimQuads * (size_t)k], p1 = tree.prims[kPrimQuads * (size_t)k + 1];
			const uint32_t flags = __float_as_uint(p0.w);
			float t, tu = 0, tv = 0;
			if (flags & kPrimSphere) {
				t = test_sphere(mk(p0.x, p0.y, p0.z), p1.x, ray);
			} else {
				const float4 p2 = tree.prims[kPrimQuads * (size_t)k + 2];
				t = test_triangle(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), ray, tu, tv);
			}
			if (t < best) {
				best = t;
				prim = (int32_t)k;
				bu = tu;
				bv = tv;
			}
			if (flags & kPrimLast)
				break;
			k++;
		}
		if (sp > 0)
			node = stack[(--sp) * kBlock];
		else
			more = false;
	}
	hits[i] = rta_hit{prim, best, bu, bv};
}

// Experiment: a trip runs a branch only if at least `threshold` lanes want it
// or it is the majority's; the minority's lanes wait for company.
// pairs (64-byte records) -> 14 arrays of n_pairs words each
__global__ __launch_bounds__(kBlock) void k_pairs_to_soa(const float4 *pairs, uint32_t n_pairs, float *soa)
{
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= n_pairs)
		return;
	const float *rec = (const float *)(pairs + 4 * (size_t)p);
	for (int f = 0; f < 14; f++)
		soa[(size_t)f * n_pairs + p] = rec[f];
}

__global__ __launch_bounds__(kBlock) void k_variant_soa_nodes(const DeviceTree tree, const float *soa, uint32_t n_pairs,
                                                              const float *rays, uint64_t n, rta_hit *hits,
                                                              const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	uint32_t *stack = lds_stack + threadIdx.x;
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float *r = rays + i * 6;
	Traversal<false, false> tr;
	WorkCount wc = {0, 0, 0};
	tr.start(tree, Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])}, FLT_MAX);
	while (__ballot(tr.more)) {
		if (!tr.more)
			continue;
		if (tr.at_prim()) {
			tr.test_prim(tree, stack, wc);
		} else {
			float w[14];
			for (int f = 0; f < 14; f++)
				w[f] = soa[(size_t)f * n_pairs + tr.node];
			tr.inner_unit(make_float4(w[0], w[1], w[2], w[3]), make_float4(w[4], w[5], w[6], w[7]),
			              make_float4(w[8], w[9], w[10], w[11]), make_float4(w[12], w[13], 0.f, 0.f), stack, depth, wc);
		}
	}
	const HitRec h = tr.result();
	hits[i] = rta_hit{h.prim, h.t, h.u, h.v};
}

// Variants 100 and up (see the list at the top).
__global__ __launch_bounds__(kBlock) void k_variant_threshold(const DeviceTree tree, const float *rays, uint64_t n,
                                                              rta_hit *hits, const int depth, const int threshold)
{
	extern __shared__ uint32_t lds_stack[];
	uint32_t *stack = lds_stack + threadIdx.x;
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float *r = rays + i * 6;
	Traversal<false, false> tr;
	WorkCount wc = {0, 0, 0};
	tr.start(tree, Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])}, FLT_MAX);
	uint32_t trips = 0, units = 0, both = 0;
	for (;;) {
		const bool want_prim = tr.more && tr.at_prim(), want_inner = tr.more && !tr.at_prim();
		const int n_prim = __popcll(__ballot(want_prim)), n_inner = __popcll(__ballot(want_inner));
		if (n_prim + n_inner == 0)
			break;
		trips++;
		units += tr.more;
		both += n_prim > 0 && n_inner > 0;
		if (n_inner >= threshold || n_inner >= n_prim) {
			if (want_inner)
				tr.visit_inner(tree, stack, depth, wc);
		}
		if (n_prim >= threshold || n_prim > n_inner) {
			if (want_prim)
				tr.test_prim(tree, stack, wc);
		}
	}
	const HitRec h = tr.result();
	if (threshold == 0)      // statistics instead of the hit: trips of the wave, units of the lane, trips with both branches
		hits[i] = rta_hit{(int32_t)trips, (float)units, (float)both, 0.f};
	else
		hits[i] = rta_hit{h.prim, h.t, h.u, h.v};
}

// Certified traversal of the wide tree; a ray left in doubt gets prim = -2 and is
// answered by k_wide_redo.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_wide_batch(
    const DeviceTree tree, const float *rays, uint64_t n, rta_hit *hits, const int cap, const uint32_t rays_per_lane,
    const uint32_t refill_at, const uint32_t quorum, const uint32_t waiting_quorum, unsigned long long *census,
    uint32_t *heads, const uint32_t chunk)
{
	extern __shared__ uint32_t lds_stack[];
	const uint64_t per_wave = 64 * (uint64_t)rays_per_lane;
	const uint64_t begin = ((uint64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64) * per_wave;
	WorkCount wc = {0, 0, 0};
	WideCensus cs = {0, 0, 0, 0, 0};
	if (heads) {
		// measurement: as the renderer hands out rays, 64 lists with a shared counter each
		const uint32_t shard = blockIdx.x % 64;
		const uint64_t lo = n * shard / 64, hi = n * (shard + 1) / 64;
		const float *mine = rays + lo * 6;
		rta_hit *out = hits + lo;
		traverse_wide_pulling<COUNT>(
		    tree, refill_at, quorum, waiting_quorum, lds_stack + threadIdx.x, cap, wc, cs,
		    SharedList{(uint32_t)(hi - lo), heads + shard, chunk},
		    [&](uint32_t i, Ray &ray, int32_t &hint) {
			    const float *r = mine + (size_t)i * 6;
			    ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
			    hint = -1;
			    return i;
		    },
		    [&](bool finished, uint32_t i, const HitRec &h, bool doubt) {
			    if (finished)
				    out[i] = doubt ? rta_hit{-2, 0.f, 0.f, 0.f} : rta_hit{h.prim, h.t, h.u, h.v};
		    });
	} else
	if (begin < n) {
		const float *mine = rays + begin * 6;
		rta_hit *out = hits + begin;
		traverse_wide_pulling<COUNT>(
		    tree, refill_at, quorum, waiting_quorum, lds_stack + threadIdx.x, cap, wc, cs,
		    OwnChunk{(uint32_t)(n - begin < per_wave ? n - begin : per_wave), 0u},
		    [&](uint32_t i, Ray &ray, int32_t &hint) {
			    const float *r = mine + (size_t)i * 6;
			    ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
			    hint = -1;
			    return i;
		    },
		    [&](bool finished, uint32_t i, const HitRec &h, bool doubt) {
			    if (finished)
				    out[i] = doubt ? rta_hit{-2, 0.f, 0.f, 0.f} : rta_hit{h.prim, h.t, h.u, h.v};
			    if (COUNT && finished && doubt)
				    wc.sph++;
		    });
	}
	if (COUNT) {
		if (wc.inner) atomicAdd(&census[0], (unsigned long long)wc.inner);
		if (wc.tri) atomicAdd(&census[1], (unsigned long long)wc.tri);
		if (wc.sph) atomicAdd(&census[2], (unsigned long long)wc.sph);
		if ((threadIdx.x & 63) == 0) {
			atomicAdd(&census[3], (unsigned long long)cs.trips);
			atomicAdd(&census[4], (unsigned long long)cs.visit_runs);
			atomicAdd(&census[6], (unsigned long long)cs.test_runs);
		}
		if (cs.visit_lanes) atomicAdd(&census[5], (unsigned long long)cs.visit_lanes);
		if (cs.test_lanes) atomicAdd(&census[7], (unsigned long long)cs.test_lanes);
	}
}

__global__ __launch_bounds__(kBlock) void k_wide_redo(const DeviceTree tree, const float *rays, uint64_t n, rta_hit *hits,
                                                      const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	const bool mine = i < n && hits[i].prim == -2;
	if (__ballot(mine) == 0)
		return;
	Traversal<false, false> tr;
	WorkCount wc = {0, 0, 0};
	tr.rest();
	if (mine) {
		const float *r = rays + i * 6;
		tr.start(tree, Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])}, FLT_MAX);
	}
	while (__ballot(tr.more)) {
		if (tr.more)
			tr.step(tree, lds_stack + threadIdx.x, depth, wc);
	}
	if (mine) {
		const HitRec h = tr.result();
		hits[i] = rta_hit{h.prim, h.t, h.u, h.v};
	}
}

// out[i] = op(x[i], y[i]) with the kernels' own primitives (rta_debug_arith)
__global__ __launch_bounds__(kBlock) void k_arith(int op, const float *x, const float *y, float *out, uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const float a = x[i], b = y[i];
	float r;
	switch (op) {
	case 0: r = a / b; break;
	case 1: r = __builtin_sqrtf(a); break;
	case 2: r = pm_sinf(a); break;
	case 3: r = pm_cosf(a); break;
	case 4: r = pm_logf(a); break;
	case 5: r = pm_powf(a, b); break;
	case 6: r = __builtin_fmaf(a, b, a); break;
	case 8: r = quotient(a, b, 1.f / b); break;
	default: r = a * b + a; break;
	}
	out[i] = r;
}

}  // namespace

static int env_int(const char *name, int fallback)
{
	const char *v = getenv(name);
	return v && *v ? atoi(v) : fallback;
}

void launch_wide_batch(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, int depth, bool redo)
{
	const int cap = env_int("RTA_WIDE_CAP", 20), rpl = env_int("RTA_WIDE_RPL", 8), refill = env_int("RTA_WIDE_REFILL", 24),
	          quorum = env_int("RTA_WIDE_QUORUM", 24), wq = env_int("RTA_WIDE_WAITING", 64), count = env_int("RTA_WIDE_COUNT", 0);
	const uint64_t per_block = (uint64_t)kBlock * (uint64_t)rpl;
	const unsigned blocks = (unsigned)((n + per_block - 1) / per_block);
	const size_t lds = (size_t)cap * kBlock * sizeof(uint32_t);
	static uint32_t *heads = nullptr;
	if (env_int("RTA_WIDE_SHARED", 0)) {
		if (!heads)
			(void)hipMalloc((void **)&heads, 64 * sizeof(uint32_t));
		(void)hipMemsetAsync(heads, 0, 64 * sizeof(uint32_t), s);
	}
	uint32_t *use_heads = env_int("RTA_WIDE_SHARED", 0) ? heads : nullptr;
	if (count) {
		unsigned long long *census = nullptr, host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		(void)hipMalloc((void **)&census, sizeof host);
		(void)hipMemsetAsync(census, 0, sizeof host, s);
		hipLaunchKernelGGL(k_wide_batch<true>, dim3(blocks), dim3(kBlock), lds, s, tree, rays, n, hits, cap, (uint32_t)rpl,
		                   (uint32_t)refill, (uint32_t)quorum, (uint32_t)wq, census, use_heads, (uint32_t)env_int("RTA_WIDE_CHUNK", 256));
		(void)hipMemcpyAsync(host, census, sizeof host, hipMemcpyDeviceToHost, s);
		(void)hipStreamSynchronize(s);
		(void)hipFree(census);
		fprintf(stderr, "wide: %.2f node visits, %.2f triangle tests per ray, %llu of %llu rays in doubt; per 64 rays %.1f trips, "
		        "%.1f visit runs (%.1f lanes), %.1f test runs (%.1f lanes)\n",
		        (double)host[0] / (double)n, (double)host[1] / (double)n, host[2], (unsigned long long)n,
		        (double)host[3] * 64 / (double)n, (double)host[4] * 64 / (double)n, (double)host[5] / (double)(host[4] ? host[4] : 1),
		        (double)host[6] * 64 / (double)n, (double)host[7] / (double)(host[6] ? host[6] : 1));
	} else {
		hipLaunchKernelGGL(k_wide_batch<false>, dim3(blocks), dim3(kBlock), lds, s, tree, rays, n, hits, cap, (uint32_t)rpl,
		                   (uint32_t)refill, (uint32_t)quorum, (uint32_t)wq, (unsigned long long *)nullptr, use_heads, (uint32_t)env_int("RTA_WIDE_CHUNK", 256));
	}
	if (redo)
		hipLaunchKernelGGL(k_wide_redo, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock),
		                   (size_t)depth * kBlock * sizeof(uint32_t), s, tree, rays, n, hits, depth);
}

void launch_pairs_to_soa(hipStream_t s, const float4 *pairs, uint32_t n_pairs, float *soa)
{
	hipLaunchKernelGGL(k_pairs_to_soa, dim3((n_pairs + kBlock - 1) / kBlock), dim3(kBlock), 0, s, pairs, n_pairs, soa);
}

void launch_variant(hipStream_t s, int variant, const TreeRef &tree, const float4 *reference_nodes, const float *soa_nodes,
                    uint32_t n_pairs, const float *rays, uint64_t n, rta_hit *hits, int depth)
{
	const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
	const size_t lds = (size_t)depth * kBlock * sizeof(uint32_t);
	if (variant >= 200) {
		// occupancy experiment: LDS sized for (variant - 200) stack entries
		hipLaunchKernelGGL(k_variant_threshold, dim3(blocks), dim3(kBlock), (size_t)(variant - 200) * kBlock * 4, s, tree,
		                   rays, n, hits, depth, 1);
		return;
	}
	if (variant >= 100) {
		hipLaunchKernelGGL(k_variant_threshold, dim3(blocks), dim3(kBlock), lds, s, tree, rays, n, hits, depth,
		                   variant - 100);
		return;
	}
	switch (variant) {
	case 1:
		hipLaunchKernelGGL(k_variant_private_stack, dim3(blocks), dim3(kBlock), 0, s, tree, rays, n, hits);
		break;
	case 2:
		hipLaunchKernelGGL(k_variant_division, dim3(blocks), dim3(kBlock), lds, s, tree, rays, n, hits, depth);
		break;
	case 4:
		hipLaunchKernelGGL(k_variant_nested_loops, dim3(blocks), dim3(kBlock), lds, s, tree, rays, n, hits, depth);
		break;
	case 5:
		hipLaunchKernelGGL(k_variant_soa_nodes, dim3(blocks), dim3(kBlock), lds, s, tree, soa_nodes, n_pairs, rays, n, hits,
		                   depth);
		break;
	default:
		hipLaunchKernelGGL(k_variant_reference_nodes, dim3(blocks), dim3(kBlock), lds, s, tree, reference_nodes, rays, n,
		                   hits, depth);
		break;
	}
}

void launch_arith(hipStream_t s, int op, const float *x, const float *y, float *out, uint64_t n)
{
	hipLaunchKernelGGL(k_arith, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, op, x, y, out, n);
}

}  // namespace rta
