// HIP kernels of the wavefront path tracer (gfx950, wave64) and their
// launchers.  The reference's per-pixel loop (src/main.cpp:657-714) becomes
// three kernels per wavefront step over a fixed pool of path slots:
//
//   advance    everything between two traversals of a path: shade the hit (one
//              bounce of RayTrace, src/main.cpp:507-589: emission, light
//              sampling, BSDF sampling, Russian roulette), retire a finished
//              sample into its block sum, hand out the next sample block, draw
//              the camera ray (src/main.cpp:663-687)
//   shadow     occlusion query for the sampled light (src/main.cpp:424-478) and
//              the deferred radiance add
//   intersect  closest hit (src/main.cpp:332-422)
//
// followed once by block-sum accumulation and the tone-mapping resolve
// (src/main.cpp:692-706).
//
// A path slot works through one (pixel, sample block) stream at a time and
// carries its xorshift32 state from sample to sample, so a block's samples are
// drawn and summed strictly in order.  Work is handed out per WAVE: the 64
// slots of a wave take one 8x8 pixel tile and a run of consecutive sample
// blocks of it, lane l rendering the tile's pixel l; the wave draws its next
// (tile, block run) from one atomic counter once all its lanes are through.
// Camera rays of a wave therefore always go through the same tile, which keeps
// the lanes of the intersect kernel on the same side of the tree.  Results
// depend only on (seed, pixel, block), never on this schedule.
//
// Lane i of every advance launch works on slot i, so the slot state (16-byte
// arrays, read and written only where a path's stage needs them) moves fully
// coalesced.  Rays reach the traversal kernels through queues compacted with a
// wave64 ballot + prefix count and one atomic per wave: the slots of fresh
// camera rays (coherent: a wave's rays cross one tile) and of continuing paths
// (incoherent bounce rays) go to different queues, occlusion queries to a third
// with their payload.  Keeping the two kinds of closest-hit rays apart is the
// divergence control: a traversal wave then holds rays of one kind, whose
// node-visit counts differ far less than across kinds.  Each queue is cut into
// kShards sub-queues (picked by workgroup) so that the appending atomics spread
// over many addresses.

#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>

#include "shading.hpp"
#include "traverse.hpp"
#include "traverse_wide.hpp"
#include "traverse_cells.hpp"
#include "traverse_flat.hpp"
#include "traverse_packet.hpp"

namespace rta {

namespace {

// slot states kept in the bounce word (ray1.w): a bounce number (with kHasColour: the path has gathered
// radiance, or an occlusion query may add some, and col[slot] holds it; without, its radiance is 0 and
// col[slot] is neither written nor read), or
constexpr uint32_t kHasColour = 0x01000000u;
constexpr uint32_t kWaiting = 0xfffffffdu;     // through with its blocks; waits for the wave's next tile
constexpr uint32_t kRetired = 0xfffffffeu;     // no work left for this wave
constexpr uint32_t kEnded = 0xffffffffu;       // path finished, its occlusion query pending; retired by the next pass
// k_sky_map values up to this say that no sample of the pixel can reach a leaf's box (u1 >= 2^-24 > this)
constexpr float kNoSampleReaches = 0x1p-25f;
// kinds of closest-hit rays, queued apart; kOddRays are the (rare) rays whose
// box tests need the division-based form (see ray_is_tame), of either origin;
// kDoubtRays are bounce rays whose certified walk over the wide tree could not
// prove its answer (traverse_wide.hpp), queued by the traversal kernel itself for
// the exact traversal that follows it
constexpr int kCameraRays = 0, kBounceRays = 1, kOddRays = 2, kDoubtRays = 3;

RTA_DEV float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
RTA_DEV float4 f4u(V3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
RTA_DEV V3 xyz(float4 q) { return mk(q.x, q.y, q.z); }

// Wave-level queue append: one atomic per wave64.  Must be reached by every
// lane of the wave (pass pred = false for lanes with nothing to add).
RTA_DEV uint32_t queue_push(bool pred, uint32_t *counter)
{
	const unsigned long long mask = __ballot(pred);
	if (mask == 0)
		return 0;
	const uint32_t before =
	    __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
	const int leader = __ffsll((long long)mask) - 1;
	uint32_t base = 0;
	if ((int)(threadIdx.x & 63) == leader)
		base = atomicAdd(counter, (uint32_t)__popcll(mask));
	base = __shfl(base, leader);
	return base + before;
}

RTA_DEV void count_flush(Counters *cnt, const WorkCount &wc)
{
	if (wc.inner) atomicAdd(&cnt->inner, (unsigned long long)wc.inner);
	if (wc.tri) atomicAdd(&cnt->tri, (unsigned long long)wc.tri);
	if (wc.sph) atomicAdd(&cnt->sph, (unsigned long long)wc.sph);
}

// The closest-hit ray queues: [step parity][kind][shard][capacity] slot numbers,
// lengths in Counters::n_ray in the same order.
RTA_DEV uint32_t ray_queue_index(int par, int kind, uint32_t shard) { return ((uint32_t)par * kShards + shard) * kRayKinds + (uint32_t)kind; }

// The ray a slot holds: ray1 = (direction, bounce word), ray0 = (origin, pdf of the direction).  A pinhole
// camera's rays (bounce word 0) all start at the pinhole with pdf 1: their ray0 is neither written nor read.
RTA_DEV bool pinhole(const RenderArgs &a) { return !(a.cam.defocus_angle > 0); }

RTA_DEV Ray slot_ray(const RenderArgs &a, uint32_t slot)
{
	const float4 r1 = a.ray1[slot];
	V3 o = mk(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2]);
	if (!pinhole(a) || __float_as_uint(r1.w) != 0u)
		o = xyz(a.ray0[slot]);
	return Ray{o, xyz(r1)};
}

// A wave's camera rays (lanes with `camera`) and bounce rays (lanes with `bounce`; no lane has
// both) join their queues with ONE atomic on the shard's 64-bit pair of counters: an atomic that
// returns a value is a round trip to the memory side that the wave sits out.  Must be reached by
// every lane of the wave.
// With cell lists (RenderArgs::camera_cells) a camera sample is queued before its ray exists: as its two random
// numbers and its pixel (`sample`), from which k_camera_cells makes the ray - with all its lanes, where the
// few lanes of a k_advance wave that have a new sample would run the ray's logarithm, sine and cosine alone.
RTA_DEV void ray_queues_push(const RenderArgs &a, int par, bool camera, bool bounce, uint32_t slot, const Ray &ray, const float4 sample)
{
	// (a camera sample's ray is looked at where it is made)
	const bool tame = (camera && a.camera_cells) || (a.tree.tame != 0 && ray_is_tame(ray));
	const uint32_t shard = blockIdx.x % kShards;
	const unsigned long long cam = __ballot(camera && tame), bnc = __ballot(bounce && tame);
	if (cam | bnc) {
		const uint32_t qc = ray_queue_index(par, kCameraRays, shard), qb = ray_queue_index(par, kBounceRays, shard);
		unsigned long long base = 0;
		const int leader = __ffsll((long long)(cam | bnc)) - 1;
		if ((int)(threadIdx.x & 63) == leader)
			base = atomicAdd(reinterpret_cast<unsigned long long *>(&a.cnt->n_ray[qc]),
			                 (unsigned long long)__popcll(cam) | ((unsigned long long)__popcll(bnc) << 32));
		const uint32_t base_c = __shfl((uint32_t)base, leader), base_b = __shfl((uint32_t)(base >> 32), leader);
		if (camera && tame) {
			const uint32_t at = base_c + __builtin_amdgcn_mbcnt_hi((uint32_t)(cam >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cam, 0u));
			a.q_ray[(size_t)qc * a.shard_cap + at] = slot;
			if (a.camera_cells)
				a.q_camera[((size_t)par * kShards + shard) * a.shard_cap + at] = sample;
		}
		if (bounce && tame) {
			const uint32_t at = base_b + __builtin_amdgcn_mbcnt_hi((uint32_t)(bnc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bnc, 0u));
			if (a.leafwise)      // (the other traversal kernels find the slot number in the entry below)
				a.q_ray[(size_t)qb * a.shard_cap + at] = slot;
			float4 *e = a.q_bounce + 2 * (((size_t)par * kShards + shard) * a.shard_cap + at);
			e[0] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
			e[1] = make_float4(ray.d.y, ray.d.z, __uint_as_float(slot), 0.f);
		}
	}
	const bool odd = (camera || bounce) && !tame;
	if (__ballot(odd)) {
		const uint32_t qo = ray_queue_index(par, kOddRays, shard);
		const uint32_t ato = queue_push(odd, &a.cnt->n_ray[qo]);
		if (odd)
			a.q_ray[(size_t)qo * a.shard_cap + ato] = slot;
	}
}

// ------------------------------------------------------------------ advance

// The random numbers of one camera sample (src/main.cpp:665-683), in the
// reference's order: the lens point, then the two of the pixel filter.
struct CameraDraws {
	float lens_x, lens_y, u1, u2;
};

RTA_DEV CameraDraws camera_draws(const RenderArgs &a, uint32_t &rng)
{
	CameraDraws d = {0.f, 0.f, 0.f, 0.f};
	if (a.cam.defocus_angle > 0) {
		// src/main.cpp:105-113,665-669; the accept test is x*y < 1 as written there
		for (;;) {
			d.lens_x = rng_float(rng) * 2 - 1;
			d.lens_y = rng_float(rng) * 2 - 1;
			if (d.lens_x * d.lens_y < 1)
				break;
		}
	}
	do {
		d.u1 = rng_float(rng);
	} while (d.u1 == 0);
	d.u2 = rng_float(rng);
	return d;
}

// One camera sample of pixel (pu, pv) from its random numbers (src/main.cpp:663-687).  `cell`: the cell of
// the image-plane map (RenderArgs::cell_w x cell_h) that the sample is aimed at, -1 beyond the map.
RTA_DEV Ray camera_ray(const RenderArgs &a, uint32_t pu, uint32_t pv, const CameraDraws &d, int32_t &cell)
{
	const V3 cam_o = mk(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2]);
	V3 origin = cam_o;
	if (a.cam.defocus_angle > 0) {
		const V3 lu = mk(a.cam.lens_u[0], a.cam.lens_u[1], a.cam.lens_u[2]);
		const V3 lv = mk(a.cam.lens_v[0], a.cam.lens_v[1], a.cam.lens_v[2]);
		origin = (cam_o + lu * d.lens_x) + lv * d.lens_y;
	}
	// Gaussian pixel filter by Box-Muller, sigma = 0.5 px (src/main.cpp:674-683)
	const float radius = .5f * __builtin_sqrtf(-2 * pm_logf(d.u1));
	float sj, cj;
	pm_sincosf(2 * kPiF * d.u2, &sj, &cj);
	const float ju = radius * cj;
	const float jv = radius * sj;
	const V3 ul = mk(a.cam.upper_left[0], a.cam.upper_left[1], a.cam.upper_left[2]);
	const V3 du = mk(a.cam.du[0], a.cam.du[1], a.cam.du[2]);
	const V3 dv = mk(a.cam.dv[0], a.cam.dv[1], a.cam.dv[2]);
	const float fx = (float)pu + .5f + ju, fy = (float)pv + .5f + jv;
	const V3 target = (ul + du * fx) + dv * fy;
	const V3 dir = unit(target - origin);
	const int ix = (int)__builtin_floorf(fx) + kSkyBorder, iy = (int)__builtin_floorf(fy) + kSkyBorder;
	cell = ix >= 0 && ix < a.cell_w && iy >= 0 && iy < a.cell_h ? iy * a.cell_w + ix : -1;
	return Ray{origin, dir};
}

// True only if the ray provably misses the box around the whole scene, and with
// it every node: a child's box lies inside it and the slab quotients are
// monotonic in the planes, so in the reference's own arithmetic both children of
// the root then fail their box tests (src/main.cpp:381-401) and NearestHit
// returns "no hit" without testing a primitive.  Decided from products with an
// approximate reciprocal (1 ulp), which are within a factor 1 +- 2^-21 of the
// quotients and have their sign (tame rays and scenes), with a margin of 2^-18.
RTA_DEV bool misses_the_scene(const TreeRef &tree, const Ray &r)
{
	if (!tree.box_known || !ray_is_tame(r))
		return false;
	const V3 rcp = mk(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
	float t_in, t_out;
	approx_box(mk(tree.box_lo[0], tree.box_lo[1], tree.box_lo[2]), mk(tree.box_hi[0], tree.box_hi[1], tree.box_hi[2]), r,
	           rcp, t_in, t_out);
	return t_out < 0.f || t_out * (1.f + 0x1p-18f) < t_in;
}

// One bounce of RayTrace on a hit (src/main.cpp:516-588): emission, the light sample whose visibility an
// occlusion query decides later, the BSDF sample, Russian roulette.  `bounce`: segments of the path before
// this one; prev_pdf: pdf of the direction that led here; thr, color, rng: the path's, updated.
// Returns whether the path goes on: along next_ray, whose direction was sampled with pdf next_pdf.
struct ShadowQuery {
	bool wanted;         // an occlusion query (o, d, limit dist) decides whether c joins the path's radiance
	V3 o, d, c;
	float dist;
};

// LIGHTS = false: for scenes known to have no light (no emission, no light sampling: less code and fewer registers)
template <bool LIGHTS = true>
RTA_DEV bool shade_hit(const RenderArgs &a, const uint32_t slot, const uint32_t bounce, const Ray &ray, const float prev_pdf,
                       const float4 H, uint32_t &rng, V3 &thr, V3 &color, Ray &next_ray, float &next_pdf, ShadowQuery &shadow)
{
	const int32_t prim = __float_as_int(H.x);
	const float t = H.y, bu = H.z, bv = H.w;
	const float4 s0 = a.shade_prims[3 * (size_t)prim];
	const float4 s1 = a.shade_prims[3 * (size_t)prim + 1];
	const float4 s2 = a.shade_prims[3 * (size_t)prim + 2];
	const int32_t mat_id = __float_as_int(s0.w), light_id = __float_as_int(s1.w);
	const bool is_sphere = __float_as_int(s2.w) == RTA_PRIM_SPHERE;
	// the slot's next camera ray goes through the same pixel (hints are triangles: the walk
	// tests every sphere anyway)
	if (a.hint && bounce == 0 && !a.camera_cells)
		a.hint[slot] = is_sphere ? -1 : prim;

	// src/main.cpp:353-354,365-368
	const V3 point = ray.o + ray.d * t;
	V3 normal;
	if (is_sphere) {
		const float4 g0 = a.tree.prims[kPrimQuads * (size_t)prim], g1 = a.tree.prims[kPrimQuads * (size_t)prim + 1];
		normal = (point - xyz(g0)) / g1.x;
	} else {
		const float bw = 1 - bu - bv;
		normal = unit((xyz(s0) * bu + xyz(s1) * bv) + xyz(s2) * bw);
	}
	bool front = true;
	if (dot(ray.d, normal) > 0) {
		normal = -normal;
		front = false;
	}
	const float4 m0 = a.materials[2 * (size_t)mat_id], m1 = a.materials[2 * (size_t)mat_id + 1];
	const Material mat = {xyz(m0), m0.w, m1.x, m1.y};

	const Frame fr = frame_from_normal(normal);
	const V3 v = to_local(fr, -ray.d);

	// emitted light, MIS-weighted after the first bounce (src/main.cpp:531-543)
	if (LIGHTS && front && light_id >= 0) {
		const float4 *L = a.lights + 7 * (size_t)light_id;
		const V3 le = xyz(L[0]);
		if (bounce == 0 || a.n_lights == 0) {
			color = color + thr * le;
		} else {
			const float pmf = 1.f / (float)a.n_lights;
			const float lpdf = pmf * triangle_pdf(xyz(L[1]), xyz(L[2]), xyz(L[3]), ray.o, point, normal);
			const float w = power_heuristic(prev_pdf, lpdf);
			color = color + (thr * le) * w;
		}
	}

	// next-event estimation on one uniformly chosen triangle light
	// (src/main.cpp:545-571); the visibility test is deferred to k_shadow
	if (LIGHTS && a.n_lights > 0) {
		const uint32_t li = rng_next(rng) % (uint32_t)a.n_lights;
		const float4 *L = a.lights + 7 * (size_t)li;
		// src/main.cpp:115-131
		const float r1u = rng_float(rng);
		const float r2u = rng_float(rng);
		float b0, b1;
		if (r1u < r2u) {
			b0 = r1u / 2;
			b1 = r2u - b0;
		} else {
			b1 = r2u / 2;
			b0 = r1u - b1;
		}
		const float b2 = 1 - b0 - b1;
		const V3 lp = (xyz(L[1]) * b0 + xyz(L[2]) * b1) + xyz(L[3]) * b2;
		const V3 ln = unit((xyz(L[4]) * b0 + xyz(L[5]) * b1) + xyz(L[6]) * b2);
		V3 ld = lp - point;
		const float dist = norm(ld);
		ld = ld / dist;
		const V3 l = to_local(fr, ld);
		if (dot(ld, ln) < 0) {
			const float pmf = 1.f / (float)a.n_lights;
			const float lpdf = pmf * triangle_pdf(xyz(L[1]), xyz(L[2]), xyz(L[3]), point, lp, ln);
			const float bpdf = bsdf_pdf(v, l, mat);
			const float w = power_heuristic(lpdf, bpdf);
			const V3 f = bsdf_eval(v, l, mat);
			shadow.c = (((thr * xyz(L[0])) * f) * w) / lpdf;
			shadow.o = point;
			shadow.d = ld;
			shadow.dist = dist - .0001f;
			shadow.wanted = true;
		}
	}

	// src/main.cpp:573-588
	const BsdfSample bs = sample_bsdf(v, mat, rng);
	thr = thr * (bs.f / bs.pdf);
	bool stop = false;
	if (bounce > 3) {
		const float p = greater(thr.x, greater(thr.y, thr.z));
		if (rng_float(rng) < (1 - p))
			stop = true;
		else
			thr = thr / p;
	}
	if (!stop && bounce + 1 < (uint32_t)a.max_bounces) {
		next_ray = Ray{point, to_world(fr, bs.l)};
		next_pdf = bs.pdf;
		return true;
	}
	return false;
}

// Everything between two traversals of a slot, in one pass over the slot state:
//
//   shade     one bounce of RayTrace on the hit the traversal left
//             (src/main.cpp:507-589);
//   retire    a finished sample joins its block sum (src/main.cpp:689), a
//             finished block is written out, a wave whose lanes are all through
//             draws its next (tile, block run);
//   generate  the next sample's camera ray (src/main.cpp:663-687).
//
// A path in its first segment has throughput 1 and radiance 0 by definition, so
// neither array is read for it, and a sample that ends in the pass it was shaded
// in never writes them.  `par`: parity of the step whose intersect launch traces
// the rays queued here.
// LIGHTS = false: for scenes known to have no light (less code in the shading, fewer registers, more waves per SIMD)
template <bool LIGHTS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LIGHTS ? 6 : 8, LIGHTS ? 6 : 8))) void k_advance(const RenderArgs a, const int par)
{
	uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63;
	// How far the group's list of streams had been handed out when this launch began or since (it only ever
	// grows): a plain load of a wave-uniform address, which goes through the scalar cache - a quarter of a
	// million waves asking the memory side for ONE word, even just to read it, queue up there.
	const uint32_t handed_out = a.cnt->next_item;
	float4 r1 = make_float4(0.f, 0.f, 0.f, __uint_as_float(kRetired)), H = make_float4(0.f, 0.f, 0.f, 0.f);
	if (a.sort_lanes) {
		// Which slot a lane works on is free (all state lives in memory, indexed by slot), and what a
		// slot needs differs widely: a bounce to shade, a finished sample to retire and a camera ray to
		// make, or nothing.  Left where they are, the kinds mix in every wave and each wave runs every
		// branch with half its lanes off.  So the workgroup's lanes trade slots first: slots to shade
		// go to the first lanes, slots with nothing to do to the middle, slots to retire and regenerate
		// to the last - a wave then holds one kind of work, two where the kinds meet.  What the trade
		// costs is coalescing: a wave's loads and stores then touch four times the lines, which pays
		// only where shading is heavy (scenes with lights).
		__shared__ uint32_t traded[kBlock];
		__shared__ float4 traded_r1[kBlock], traded_hit[kBlock];
		__shared__ uint32_t counts[kBlock / 64][2];
		float4 own_r1 = make_float4(0.f, 0.f, 0.f, __uint_as_float(kRetired)), own_hit = make_float4(0.f, 0.f, 0.f, 0.f);
		if (slot < a.n_slots)
			own_r1 = a.ray1[slot];
		const uint32_t st = __float_as_uint(own_r1.w);
		// 0: shade, 1: nothing, 2: retire / draw a stream / generate
		int kind = 1;
		if (st < kWaiting) {
			own_hit = a.hit[slot];
			kind = __float_as_int(own_hit.x) < 0 ? 2 : 0;
		} else if (st != kRetired)
			kind = 2;
		const unsigned long long shade = __ballot(kind == 0), idle = __ballot(kind == 1);
		if (lane == 0) {
			counts[threadIdx.x / 64][0] = (uint32_t)__popcll(shade);
			counts[threadIdx.x / 64][1] = (uint32_t)__popcll(idle);
		}
		__syncthreads();
		uint32_t before0 = 0, before1 = 0, before2 = 0, total0 = 0, total1 = 0;
		for (uint32_t w = 0; w < kBlock / 64; w++) {
			const uint32_t n0 = counts[w][0], n1 = counts[w][1];
			if (w < threadIdx.x / 64) {
				before0 += n0;
				before1 += n1;
				before2 += 64 - n0 - n1;
			}
			total0 += n0;
			total1 += n1;
		}
		if (total1 == kBlock)
			return;               // nothing left in any of the workgroup's slots
		const unsigned long long mine = kind == 0 ? shade : (kind == 1 ? idle : ~(shade | idle));
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
		const uint32_t to = rank + (kind == 0 ? before0 : (kind == 1 ? total0 + before1 : total0 + total1 + before2));
		traded[to] = slot;
		traded_r1[to] = own_r1;
		traded_hit[to] = own_hit;
		__syncthreads();
		slot = traded[threadIdx.x];
		r1 = traded_r1[threadIdx.x];
		H = traded_hit[threadIdx.x];
	} else if (slot < a.n_slots)
		r1 = a.ray1[slot];
	uint32_t state = __float_as_uint(r1.w);
	if (__ballot(state != kRetired) == 0)
		return;                   // whole waves leave together

	bool alive = false, ended = false;
	Ray next_ray = {mk(0, 0, 0), mk(1, 1, 1)};
	ShadowQuery shadow = {false, mk(0, 0, 0), mk(0, 0, 0), mk(0, 0, 0), 0.f};
	V3 color = mk(0, 0, 0);
	uint32_t rng = 0;

	if (state < kWaiting) {
		const uint32_t bounce = state & (kHasColour - 1);
		if (!a.sort_lanes)
			H = a.hit[slot];
		rng = a.rng[slot];
		V3 thr = mk(1, 1, 1);
		if (bounce > 0) {
			thr = xyz(a.thr[slot]);
			if (state & kHasColour)
				color = xyz(a.col[slot]);
		}
		const int32_t prim = __float_as_int(H.x);

		if (prim < 0) {
			// src/main.cpp:511-514
			color = color + thr * mk(a.sky[0], a.sky[1], a.sky[2]);
		} else {
			float4 r0 = make_float4(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2], 1.f);
			if (bounce > 0 || !pinhole(a))
				r0 = a.ray0[slot];
			float next_pdf = 0.f;
			alive = shade_hit<LIGHTS>(a, slot, bounce, Ray{xyz(r0), xyz(r1)}, r0.w, H, rng, thr, color, next_ray, next_pdf, shadow);
			if (alive) {
				// (with lights nearly every path gathers at once: not worth the test)
				const bool gathered = LIGHTS || ((__float_as_uint(color.x) | __float_as_uint(color.y) | __float_as_uint(color.z)) != 0);
				a.ray0[slot] = make_float4(next_ray.o.x, next_ray.o.y, next_ray.o.z, next_pdf);
				a.ray1[slot] = make_float4(next_ray.d.x, next_ray.d.y, next_ray.d.z, __uint_as_float((bounce + 1) | (gathered ? kHasColour : 0u)));
				a.thr[slot] = f4(thr, 0.f);
			}
		}
		if (alive || shadow.wanted) {
			// the radiance so far is needed again: by the next bounce, or by the
			// occlusion query that may add the light's share to it (nothing to keep while it is 0)
			if (LIGHTS || ((__float_as_uint(color.x) | __float_as_uint(color.y) | __float_as_uint(color.z)) != 0))
				a.col[slot] = f4(color, 0.f);
			if (!alive)
				state = kEnded;         // retired by the next pass, after k_shadow has run
			a.rng[slot] = rng;
		} else {
			ended = true;
		}
	} else if (state == kEnded) {
		color = xyz(a.col[slot]);
		rng = a.rng[slot];
		ended = true;
	}
	// the occlusion query of the light sampled at the hit joins its queue here, not at the end of the kernel: ten
	// registers less to keep through the loop below (every lane of the wave is here)
	if (LIGHTS) {
		const uint32_t shard = blockIdx.x % kShards;
		const uint32_t at_sh = queue_push(shadow.wanted, &a.cnt->n_shadow[shard]);
		if (shadow.wanted) {
			float4 *e = a.shadow + 3 * ((size_t)shard * a.shard_cap + at_sh);
			e[0] = f4(shadow.o, shadow.dist);
			e[1] = f4u(shadow.d, slot);      // slot < n_slots here: a wanted query implies a live lane
			e[2] = f4(shadow.c, 0.f);
		}
	}

	// Retire, draw new streams, generate - and again, for as long as some lane's
	// new camera sample provably ends in the sky (it cannot reach a leaf's box by
	// sky_map, or its ray misses the box around the whole scene): such a sample is
	// over at once (src/main.cpp:511-514 with throughput 1 and radiance 0), without
	// a queue entry, a traversal or a word of slot state moved.  Lanes whose ray must
	// be traced keep it and wait for the loop to end, hence the small fast_trips.
	uint4 M = make_uint4(0u, 0u, 0u, 0u);     // pixel, current block, end of the block run, index in the padded pixel list
	float4 A = make_float4(0.f, 0.f, 0.f, 0.f);     // block sum, samples finished in the block
	bool need_ray = false, have_ray = false, in_registers = false, meta_changed = false;
	uint32_t answered_here = 0;
	uint32_t sky_pixel = 0xffffffffu, sky_pu = 0, sky_pv = 0;     // the pixel sky_u1 belongs to
	uint32_t repeat = 1;          // samples that the finished sample stands for
	float sky_u1 = 2.f;
	// With a sky map, a sample that does need a ray keeps its random numbers and the ray is
	// made after the loop: per trip only a few lanes get that far, and the ray's logarithm,
	// sine and cosine (binary64 polynomials) would run for them alone trip after trip.
	CameraDraws held = {0.f, 0.f, 0.f, 0.f};
	const bool ray_later = a.sky_map != nullptr;
	int32_t cell = -1;
	for (uint32_t trip = 0;; trip++) {
		if (ended) {
			// the finished sample joins its block sum (src/main.cpp:689)
			if (!in_registers) {
				M = a.meta[slot];
				A = a.acc[slot];
				in_registers = true;
			}
			uint32_t done = __float_as_uint(A.w) + repeat;
			for (uint32_t r = 0; r < repeat; r++) {      // (1, or the rest of a block of sky samples)
				A.x += color.x;
				A.y += color.y;
				A.z += color.z;
			}
			repeat = 1;
			const uint32_t first = M.y * a.sps;
			const uint32_t len = a.spp - first < a.sps ? a.spp - first : a.sps;
			need_ray = true;
			if (done >= len) {
				a.partial[(size_t)(M.y - a.first_block) * a.n_pad + M.w] = make_float4(A.x, A.y, A.z, 0.f);
				A = make_float4(0.f, 0.f, 0.f, 0.f);
				done = 0;
				M.y++;
				meta_changed = true;
				if (M.y >= M.z) {
					state = kWaiting;
					need_ray = false;
				} else {
					rng = stream_seed(a.seed, M.x, M.y);
				}
			}
			A.w = __uint_as_float(done);
			ended = false;
		}

		// Lanes without a stream draw the next ones of the group's list as soon as
		// grab_at of them are waiting, or at once when nothing else is left to do in
		// the wave.  The list is ordered: parts of the image heaviest first
		// (part_order, dealt round-robin to the slot groups), within a part its
		// 64 / lane_blocks pixels for the first block run, then for the second, ...
		// so that the streams a wave draws together are neighbours in the image.
		const unsigned long long waiting = __ballot(state == kWaiting);
		const uint32_t n_wait = (uint32_t)__popcll(waiting);
		if (n_wait && (n_wait >= a.grab_at || __ballot(state != kWaiting && state != kRetired) == 0)) {
			// (The first pass after k_init_slots finds every lane of every wave waiting.  A quarter of a million
			// waves taking their places in line at ONE counter cost that pass 2.4 ms, so there the places are
			// the slots' own numbers and the host starts the counter behind them.)
			uint32_t base = blockIdx.x * kBlock + (threadIdx.x & ~63u);
			if (!(a.first_pass && trip == 0)) {      // (lanes that are through with a stream within the pass ask like everyone)
				// (once the list was seen dry no wave needs to ask: the waves that retire together at the end of a
				// render would otherwise queue at the counter again)
				base = handed_out;
				if (handed_out < a.n_streams) {
					if (lane == 0)
						base = atomicAdd(&a.cnt->next_item, n_wait);
					base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
				}
			}
			const uint32_t got = base < a.n_streams ? (a.n_streams - base < n_wait ? a.n_streams - base : n_wait) : 0u;
			const uint32_t rank =
			    __builtin_amdgcn_mbcnt_hi((uint32_t)(waiting >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)waiting, 0u));
			if (state == kWaiting) {
				if (rank < got) {
					const uint32_t stream = base + rank;
					const uint32_t k = stream / a.streams_per_part, within = stream - k * a.streams_per_part;
					const uint32_t which = a.part_order[k * a.n_groups + a.group];
					const uint32_t tile = which / a.lane_blocks, part = which - tile * a.lane_blocks;
					const uint32_t part_pixels = 64 / a.lane_blocks;
					const uint32_t run = within / part_pixels, px = within - run * part_pixels;
					const uint32_t padded = tile * 64 + part * part_pixels + px;
					const uint32_t first = a.first_block + run * a.run_blocks;
					const int32_t pixel = a.pixels[padded];
					if (pixel >= 0) {          // (a tile that leaves the frame has streams without a pixel)
						M.x = (uint32_t)pixel;
						M.y = first;
						M.z = first + a.run_blocks < a.end_block ? first + a.run_blocks : a.end_block;
						M.w = padded;
						in_registers = true;
						meta_changed = true;
						rng = stream_seed(a.seed, M.x, M.y);
						A = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
						need_ray = true;
						state = 0;             // (a camera ray carries bounce 0)
					}
				} else {
					state = kRetired;          // the list is dry
				}
			}
			// slots that retire are through for the rest of the render
			if (got < n_wait && lane == 0)
				atomicAdd(&a.cnt->retired[(blockIdx.x * (kBlock / 64) + threadIdx.x / 64) % kShards], n_wait - got);
		}
		// every lane retired (not so on entry): the wave is through
		if (__ballot(state != kRetired) == 0)
			break;

		bool answered = false;
		if (need_ray && !have_ray) {
			if (sky_pixel != M.x) {
				sky_pixel = M.x;
				sky_pu = M.x % a.width;
				sky_pv = M.x / a.width;
				sky_u1 = a.sky_map ? a.sky_map[M.x] : 2.f;
			}
			if (trip < a.bulk_trips && sky_u1 <= kNoSampleReaches) {
				// Every sample of this pixel ends in the sky (u1 is a positive multiple of
				// 2^-24): the rest of the block is the sky's colour added once per sample,
				// and its random numbers are never looked at (the next block has its own seed).
				const uint32_t first = M.y * a.sps;
				const uint32_t len = a.spp - first < a.sps ? a.spp - first : a.sps;
				repeat = len - __float_as_uint(A.w);
				answered = true;
			} else if (const CameraDraws draws = camera_draws(a, rng); trip < a.fast_trips && draws.u1 >= sky_u1) {
				answered = true;      // too close to the pixel's centre to reach any leaf's box
			} else if (ray_later) {
				held = draws;
				have_ray = true;
			} else {
				next_ray = camera_ray(a, sky_pu, sky_pv, draws, cell);
				have_ray = true;
				answered = trip < a.fast_trips && misses_the_scene(a.tree, next_ray);
			}
		}
		if (__ballot(answered) == 0)
			break;
		if (answered) {
			color = mk(0, 0, 0) + mk(1, 1, 1) * mk(a.sky[0], a.sky[1], a.sky[2]);
			ended = true;
			need_ray = false;
			have_ray = false;
			answered_here += repeat;
		}
	}

	if (ray_later && need_ray && !a.camera_cells)
		next_ray = camera_ray(a, sky_pu, sky_pv, held, cell);
	if (need_ray) {
		a.acc[slot] = A;
		a.rng[slot] = rng;
		if (meta_changed)
			a.meta[slot] = M;
		if (!pinhole(a))
			a.ray0[slot] = make_float4(next_ray.o.x, next_ray.o.y, next_ray.o.z, 1.f);     // bsdf_pdf = 1
		a.ray1[slot] = make_float4(next_ray.d.x, next_ray.d.y, next_ray.d.z, __uint_as_float(0u));     // bounce 0
	} else if (state >= kWaiting && state != __float_as_uint(r1.w)) {
		// (a slot without a ray in flight has no use for a direction: three registers less to keep through the kernel)
		a.ray1[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(state));
	}
	// rays answered above count like traced ones (each stands for one NearestHit call)
	for (int m = 32; m >= 1; m >>= 1)
		answered_here += __shfl_xor(answered_here, m);
	if (answered_here && lane == 0)
		atomicAdd(&a.cnt->early[(blockIdx.x * (kBlock / 64) + threadIdx.x / 64) % kShards], (unsigned long long)answered_here);

	// camera rays are coherent (a wave's rays cross one tile), bounce rays are not:
	// they are queued apart so that a traversal wave holds rays of one kind
	ray_queues_push(a, par, need_ray, alive, slot, next_ray, make_float4(held.u1, held.u2, __uint_as_float(sky_pixel), 0.f));
}

// ------------------------------------------------------------------ intersect

// Housekeeping done by the first workgroup of the first traversal launch of a
// step: count the step's rays, empty the queues the previous step consumed,
// empty the occlusion queue.
RTA_DEV void step_housekeeping(const RenderArgs &a, int par)
{
	if (threadIdx.x < kRayKinds * kShards) {
		// (the queue of rays to trace again fills while this launch runs; its rays are counted already)
		const uint32_t n = threadIdx.x % kRayKinds != (uint32_t)kDoubtRays ? a.cnt->n_ray[ray_queue_index(par, 0, 0) + threadIdx.x] : 0;
		if (n)
			atomicAdd(&a.cnt->rays, (unsigned long long)n);
		if (n && a.wide_cap && (threadIdx.x % kRayKinds == (uint32_t)kBounceRays || (a.camera_walk && !a.camera_cells && threadIdx.x % kRayKinds == (uint32_t)kCameraRays)))
			atomicAdd(&a.cnt->walked, (unsigned long long)n);
		if (n && a.camera_cells && threadIdx.x % kRayKinds == (uint32_t)kCameraRays)
			atomicAdd(&a.cnt->listed, (unsigned long long)n);
		if (n && a.tree.flat_prims && (threadIdx.x % kRayKinds == (uint32_t)kBounceRays || (!a.camera_cells && threadIdx.x % kRayKinds == (uint32_t)kCameraRays)))
			atomicAdd(&a.cnt->scanned, (unsigned long long)n);
		a.cnt->n_ray[ray_queue_index(par ^ 1, 0, 0) + threadIdx.x] = 0;
	}
	if (threadIdx.x < 2 * kShards)
		a.cnt->head[(uint32_t)(par ^ 1) * 2 * kShards + threadIdx.x] = 0;

	if (threadIdx.x < kShards)
		a.cnt->n_shadow[threadIdx.x] = 0;
}

// Bounce rays are traced by few waves per sub-queue (one per rays_per_lane x 64
// entries) that draw their rays from it as their lanes run out of work
// (traverse_pulling); the waves of a sub-queue share its `head` counter, so a wave
// that drew long rays simply draws fewer.
RTA_DEV uint32_t pulling_blocks(uint32_t n_rays, uint32_t rpl) { return (n_rays + kBlock * rpl - 1) / (kBlock * rpl); }

// Camera and bounce rays.  Grid: first the workgroups that pull bounce rays
// (kShards sub-queues x the most a sub-queue can need), then one workgroup per
// kBlock camera-ray entries (kShards x shard_cap / kBlock); shards innermost so
// that the first workgroups of all shards start together.  A workgroup with
// nothing to do leaves at once.  The grid is sized by RenderArgs::grid_cap, the
// host's bound on what a sub-queue can hold this step.
RTA_DEV uint32_t pulling_grid(uint32_t shard_cap, uint32_t rpl) { return kShards * pulling_blocks(shard_cap, rpl); }

template <bool COUNT, bool WIDE, int PART = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_intersect(const RenderArgs a, const int par, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	if (blockIdx.x == 0 && PART != 2)
		step_housekeeping(a, par);
	WorkCount wc = {0, 0, 0};
	const uint32_t n_pulling = pulling_grid(a.grid_cap, a.rays_per_lane);
	if (PART == 1 && blockIdx.x >= n_pulling)
		return;
	if (PART == 2 && blockIdx.x < n_pulling)
		return;
	// with camera_walk the camera rays' workgroups pull them through the walk as the bounce rays' do
	const bool camera_pulls = WIDE && a.camera_walk && blockIdx.x >= n_pulling;
	if (blockIdx.x < n_pulling || camera_pulls) {
		const uint32_t block = camera_pulls ? blockIdx.x - n_pulling : blockIdx.x;
		const int kind = camera_pulls ? kCameraRays : kBounceRays;
		const uint32_t shard = block % kShards, part = block / kShards;
		const uint32_t q = ray_queue_index(par, kind, shard);
		const uint32_t total = a.cnt->n_ray[q];
		// a short list is spread over up to 8 workgroups rather than worked through by
		// one: towards the end of a render the steps are as long as their longest wave
		const uint32_t spread = (total + a.spread_blocks * kBlock - 1) / (a.spread_blocks * kBlock);
		const uint32_t rays_per_lane = spread < a.rays_per_lane ? (spread ? spread : 1u) : a.rays_per_lane;
		if (part < pulling_blocks(total, rays_per_lane)) {
			const float4 *queue = a.q_bounce + 2 * ((size_t)par * kShards + shard) * a.shard_cap;
			uint32_t *head = &a.cnt->head[((uint32_t)par * kShards + shard) * 2 + (camera_pulls ? 1u : 0u)];
			if (WIDE) {
				WideCensus census = {0, 0, 0, 0, 0};
				const uint32_t qd = ray_queue_index(par, kDoubtRays, shard);
				traverse_wide_pulling<COUNT>(
				    a.tree, a.walk_refill_at, a.walk_quorum, a.walk_waiting, lds_stack + threadIdx.x, (int)a.wide_cap, wc, census,
				    SharedList{total, head, 32u * rays_per_lane},
				    [&](uint32_t i, Ray &ray, int32_t &hint) {
					    if (camera_pulls) {
						    // camera rays are queued by slot number only
						    const uint32_t slot = a.q_ray[(size_t)q * a.shard_cap + i];
						    ray = slot_ray(a, slot);
						    hint = a.hint ? a.hint[slot] : -1;
						    return slot;
					    }
					    const float4 r0 = queue[2 * (size_t)i], r1 = queue[2 * (size_t)i + 1];
					    ray = Ray{mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y)};
					    hint = -1;
					    return __float_as_uint(r1.z);
				    },
				    [&](bool finished, uint32_t slot, const HitRec &h, bool doubt) {
					    if (finished && !doubt)
						    a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
					    // a ray whose answer is not proven goes to the exact traversal (k_intersect_odd)
					    const bool again = finished && doubt;
					    if (__ballot(again)) {
						    const uint32_t at = queue_push(again, &a.cnt->n_ray[qd]);
						    if (again)
							    a.q_ray[(size_t)qd * a.shard_cap + at] = slot;
					    }
				    });
			} else
			traverse_pulling<false, COUNT, true>(
			    a.tree, a.refill_at, lds_stack + threadIdx.x, depth, wc,
			    SharedList{total, head, 32u * rays_per_lane},
			    [&](uint32_t i, Ray &ray, float &limit) {
				    const float4 r0 = queue[2 * (size_t)i], r1 = queue[2 * (size_t)i + 1];
				    ray = Ray{mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y)};
				    limit = FLT_MAX;
				    return __float_as_uint(r1.z);
			    },
			    [&](uint32_t slot, const HitRec &h) {
				    a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
			    });
		}
	} else {
		const uint32_t rem = blockIdx.x - n_pulling;
		const uint32_t shard = rem % kShards, part = rem / kShards;
		const uint32_t q = ray_queue_index(par, kCameraRays, shard);
		const uint32_t total = a.cnt->n_ray[q];
		const uint32_t i = part * kBlock + threadIdx.x;
		if (i < total) {
			const uint32_t slot = a.q_ray[(size_t)q * a.shard_cap + i];
			Ray ray = slot_ray(a, slot);
			HitRec h;
			traverse_dyn<false, COUNT, true>(a.tree, ray, FLT_MAX, lds_stack + threadIdx.x, depth, h, wc);
			a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
		}
	}
	if (COUNT)
		count_flush(a.cnt, wc);
}

// A pinhole camera's rays, answered from the triangle lists of their cells on the image plane
// (traverse_cells.hpp, cells.hip), one ray per lane; runs before the step's k_intersect.  Grid: one workgroup
// per kBlock entries of the camera sub-queues, shards innermost.
//
// With RenderArgs::shade_first_hit a ray that hits is shaded here as well - bounce 0 of RayTrace
// (src/main.cpp:507-589) with throughput 1 and radiance 0, from the slot's random stream where k_advance left
// it - and its bounce ray joins the step's bounce queue, which the traversal kernel traces next, its occlusion
// query the step's occlusion queue, which k_shadow answers next: a camera sample that hits takes one pass of
// k_advance less, and the lanes that shade are compact here.  A slot's bounce ray and occlusion query go to the
// sub-queues its camera ray came from (the slot's own, see ray_queues_push), so no sub-queue receives more than
// its slots.
template <bool COUNT, bool LIGHTS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LIGHTS ? 7 : 8, LIGHTS ? 7 : 8))) void k_camera_cells(const RenderArgs a, const int par)
{
	const uint32_t shard = blockIdx.x % kShards, part = blockIdx.x / kShards;
	const uint32_t q = ray_queue_index(par, kCameraRays, shard);
	const uint32_t total = a.cnt->n_ray[q];
	if (part * kBlock >= total)
		return;
	const uint32_t i = part * kBlock + threadIdx.x;
	const bool active = i < total;
	uint32_t slot = 0;
	int32_t cell = -1;
	Ray ray = {mk(0, 0, 0), mk(1, 1, 1)};
	if (active) {
		slot = a.q_ray[(size_t)q * a.shard_cap + i];
		// the sample as k_advance drew it: the pixel filter's two random numbers and the pixel (src/main.cpp:674-687)
		const float4 e = a.q_camera[((size_t)par * kShards + shard) * a.shard_cap + i];
		const uint32_t pixel = __float_as_uint(e.z);
		ray = camera_ray(a, pixel % a.width, pixel / a.width, CameraDraws{0.f, 0.f, e.x, e.y}, cell);
	}
	// the slot does not know its ray yet: whoever looks at the ray after this kernel finds it there
	bool ray_wanted = false;
	// (a ray that needs the division-based box test goes the way of those the lists leave in doubt)
	const bool odd = active && !(a.tree.tame != 0 && ray_is_tame(ray));
	if (odd)
		cell = -1;
	WorkCount wc = {0, 0, 0};
	HitRec h;
	bool doubt;
	closest_from_cell<COUNT>(a.tree, active, ray, cell, a.cell_start, a.cell_list, a.cell_keys, h, doubt, wc);
	bool onward = false;
	Ray next_ray = {mk(0, 0, 0), mk(1, 1, 1)};
	ShadowQuery shadow = {false, mk(0, 0, 0), mk(0, 0, 0), mk(0, 0, 0), 0.f};
	if (active && !doubt) {
		const float4 H = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
		if (a.shade_first_hit && h.prim >= 0) {
			uint32_t rng = a.rng[slot];
			V3 thr = mk(1, 1, 1), color = mk(0, 0, 0);
			float next_pdf = 0.f;
			onward = shade_hit<LIGHTS>(a, slot, 0u, ray, 1.f, H, rng, thr, color, next_ray, next_pdf, shadow);      // (a pinhole's rays: pdf 1)
			a.rng[slot] = rng;
			// (as in k_advance: with lights nearly every path gathers at once)
			const bool gathered = LIGHTS || (__float_as_uint(color.x) | __float_as_uint(color.y) | __float_as_uint(color.z)) != 0;
			if (onward) {
				a.ray0[slot] = make_float4(next_ray.o.x, next_ray.o.y, next_ray.o.z, next_pdf);
				a.ray1[slot] = make_float4(next_ray.d.x, next_ray.d.y, next_ray.d.z, __uint_as_float(1u | (gathered ? kHasColour : 0u)));
				a.thr[slot] = f4(thr, 0.f);
				if (gathered)
					a.col[slot] = f4(color, 0.f);
			} else {
				// the sample is over: the next pass of k_advance retires it
				a.col[slot] = f4(color, 0.f);
				a.ray1[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(kEnded));
			}
		} else {
			a.hit[slot] = H;
			ray_wanted = h.prim >= 0;      // k_advance shades it
		}
	}
	// a ray whose answer is not proven goes to the exact traversal (k_intersect_odd)
	const bool again = active && doubt;
	if (again || ray_wanted)
		a.ray1[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(0u));
	if (__ballot(again)) {
		const uint32_t qd = ray_queue_index(par, kDoubtRays, shard);
		const uint32_t at = queue_push(again, &a.cnt->n_ray[qd]);
		if (again)
			a.q_ray[(size_t)qd * a.shard_cap + at] = slot;
	}
	// the occlusion query of the light sampled at the hit joins the step's queue: k_shadow runs after this kernel
	if (LIGHTS && __ballot(shadow.wanted)) {
		const uint32_t at_sh = queue_push(shadow.wanted, &a.cnt->n_shadow[shard]);
		if (shadow.wanted) {
			float4 *e = a.shadow + 3 * ((size_t)shard * a.shard_cap + at_sh);
			e[0] = f4(shadow.o, shadow.dist);
			e[1] = f4u(shadow.d, slot);
			e[2] = f4(shadow.c, 0.f);
		}
	}
	if (COUNT && wc.tri)
		atomicAdd(&a.cnt->cell_tests, (unsigned long long)wc.tri);
	if (__ballot(onward)) {
		const bool tame = a.tree.tame != 0 && ray_is_tame(next_ray);
		const uint32_t qb = ray_queue_index(par, kBounceRays, shard), qo = ray_queue_index(par, kOddRays, shard);
		const uint32_t at = queue_push(onward && tame, &a.cnt->n_ray[qb]);
		if (onward && tame) {
			if (a.leafwise)      // (the other traversal kernels find the slot number in the entry below)
				a.q_ray[(size_t)qb * a.shard_cap + at] = slot;
			float4 *e = a.q_bounce + 2 * (((size_t)par * kShards + shard) * a.shard_cap + at);
			e[0] = make_float4(next_ray.o.x, next_ray.o.y, next_ray.o.z, next_ray.d.x);
			e[1] = make_float4(next_ray.d.y, next_ray.d.z, __uint_as_float(slot), 0.f);
		}
		const uint32_t ato = queue_push(onward && !tame, &a.cnt->n_ray[qo]);
		if (onward && !tame)
			a.q_ray[(size_t)qo * a.shard_cap + ato] = slot;
	}
	if (COUNT)
		count_flush(a.cnt, wc);
}

// k_intersect for trees with long leaves (RenderArgs::leafwise): bounce and
// camera rays alike take one lane each through traverse_leafwise.  Grid: one
// workgroup per kBlock entries of the bounce sub-queues, then of the camera
// sub-queues, shards innermost.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_intersect_leafwise(const RenderArgs a, const int par, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	if (blockIdx.x == 0)
		step_housekeeping(a, par);
	const uint32_t per_kind = kShards * (a.grid_cap / kBlock);
	const int kind = blockIdx.x < per_kind ? kBounceRays : kCameraRays;
	const uint32_t rem = blockIdx.x < per_kind ? blockIdx.x : blockIdx.x - per_kind;
	const uint32_t shard = rem % kShards, part = rem / kShards;
	const uint32_t q = ray_queue_index(par, kind, shard);
	const uint32_t total = a.cnt->n_ray[q];
	const uint32_t i = part * kBlock + threadIdx.x;
	WorkCount wc = {0, 0, 0};
	if (i < total) {
		const uint32_t slot = a.q_ray[(size_t)q * a.shard_cap + i];
		Ray ray = slot_ray(a, slot);
		HitRec h;
		traverse_leafwise<false, COUNT, true>(a.tree, ray, FLT_MAX, lds_stack + threadIdx.x, depth, h, wc);
		a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
	}
	if (COUNT)
		count_flush(a.cnt, wc);
}

// k_intersect for scenes of a few dozen primitives (TreeRef::flat_prims): every ray tests them all
// (traverse_flat.hpp), one ray per lane; what that leaves in doubt goes to k_intersect_odd.  The grid of
// k_intersect_leafwise; bounce rays come with their queue entries, camera rays (where no cell lists serve
// them) from their slots.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_intersect_flat(const RenderArgs a, const int par)
{
	if (blockIdx.x == 0)
		step_housekeeping(a, par);
	const uint32_t per_kind = kShards * (a.grid_cap / kBlock);
	const int kind = blockIdx.x < per_kind ? kBounceRays : kCameraRays;
	const uint32_t rem = blockIdx.x < per_kind ? blockIdx.x : blockIdx.x - per_kind;
	const uint32_t shard = rem % kShards, part = rem / kShards;
	const uint32_t q = ray_queue_index(par, kind, shard);
	const uint32_t total = a.cnt->n_ray[q];
	if (part * kBlock >= total)
		return;
	const uint32_t i = part * kBlock + threadIdx.x;
	const bool active = i < total;
	uint32_t slot = 0;
	Ray ray = {mk(0, 0, 0), mk(1, 1, 1)};
	if (active && kind == kBounceRays) {
		const float4 *e = a.q_bounce + 2 * (((size_t)par * kShards + shard) * a.shard_cap + i);
		const float4 r0 = e[0], r1 = e[1];
		ray = Ray{mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y)};
		slot = __float_as_uint(r1.z);
	} else if (active) {
		slot = a.q_ray[(size_t)q * a.shard_cap + i];
		ray = slot_ray(a, slot);
	}
	WorkCount wc = {0, 0, 0};
	HitRec h;
	bool doubt;
	scan_flat<false, COUNT>(a.tree, active, ray, FLT_MAX, h, doubt, wc);
	if (active && !doubt)
		a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
	if (__ballot(doubt)) {
		const uint32_t qd = ray_queue_index(par, kDoubtRays, shard);
		const uint32_t at = queue_push(doubt, &a.cnt->n_ray[qd]);
		if (doubt)
			a.q_ray[(size_t)qd * a.shard_cap + at] = slot;
	}
	if (COUNT)
		count_flush(a.cnt, wc);
}

// The rays that need the division-based box test, and the rays that the
// certified walk of the launch before left to the exact traversal.  Any grid that
// is a multiple of kShards works: the workgroups of a shard stride over its
// sub-queues.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_intersect_odd(const RenderArgs a, const int par, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	const uint32_t shard = blockIdx.x % kShards, part = blockIdx.x / kShards, parts = gridDim.x / kShards;
	WorkCount wc = {0, 0, 0};
	for (int kind = kOddRays; kind <= kDoubtRays; kind++) {
		const uint32_t q = ray_queue_index(par, kind, shard);
		const uint32_t total = a.cnt->n_ray[q];
		if (kind == kDoubtRays && part == 0 && threadIdx.x == 0 && total)
			atomicAdd(&a.cnt->retraced, (unsigned long long)total);
		for (uint32_t i = part * kBlock + threadIdx.x; i < total; i += parts * kBlock) {
			const uint32_t slot = a.q_ray[(size_t)q * a.shard_cap + i];
			Ray ray = slot_ray(a, slot);
			HitRec h;
			traverse_dyn<false, COUNT, false>(a.tree, ray, FLT_MAX, lds_stack + threadIdx.x, depth, h, wc);
			a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
		}
	}
	if (COUNT)
		count_flush(a.cnt, wc);
}

// ------------------------------------------------------------------ shadow

template <bool COUNT, bool LEAFWISE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shadow(const RenderArgs a, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	// grid: shards innermost, as the ray queues' (kShards x grid_cap / kBlock workgroups)
	const uint32_t shard = blockIdx.x % kShards, part = blockIdx.x / kShards;
	const uint32_t i = part * kBlock + threadIdx.x;
	const uint32_t n = a.cnt->n_shadow[shard];
	if (i == 0 && n)
		atomicAdd(&a.cnt->shadow_rays, (unsigned long long)n);
	WorkCount wc = {0, 0, 0};
	if (i < n) {
		const float4 *e = a.shadow + 3 * ((size_t)shard * a.shard_cap + i);
		const float4 e0 = e[0], e1 = e[1], e2 = e[2];
		Ray ray = {xyz(e0), xyz(e1)};
		HitRec h;
		if (LEAFWISE)
			traverse_leafwise<true, COUNT>(a.tree, ray, e0.w, lds_stack + threadIdx.x, depth, h, wc);
		else
			traverse_dyn<true, COUNT>(a.tree, ray, e0.w, lds_stack + threadIdx.x, depth, h, wc);
		if (h.prim < 0) {
			// src/main.cpp:569: the light's contribution joins the path radiance
			const uint32_t slot = __float_as_uint(e1.w);
			float4 C = a.col[slot];
			C.x += e2.x;
			C.y += e2.y;
			C.z += e2.z;
			a.col[slot] = C;
		}
	}
	if (COUNT)
		count_flush(a.cnt, wc);
}

// k_shadow for scenes of a few dozen primitives (TreeRef::flat_prims): every query tests them all
// (traverse_flat.hpp).  A query that this cannot answer for certain, or whose ray needs the division-based box
// test, is traced by the replaying traversal on the spot: such queries are too rare for a queue of their own.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_shadow_flat(const RenderArgs a, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	const uint32_t shard = blockIdx.x % kShards, part = blockIdx.x / kShards;
	const uint32_t i = part * kBlock + threadIdx.x;
	const uint32_t n = a.cnt->n_shadow[shard];
	if (i == 0 && n)
		atomicAdd(&a.cnt->shadow_rays, (unsigned long long)n);
	if (part * kBlock >= n)
		return;
	WorkCount wc = {0, 0, 0};
	const float4 *e = a.shadow + 3 * ((size_t)shard * a.shard_cap + (i < n ? i : 0u));
	const float4 e0 = e[0], e1 = e[1];
	const Ray ray = {xyz(e0), xyz(e1)};
	const bool tame = ray_is_tame(ray);
	HitRec h;
	bool doubt;
	scan_flat<true, COUNT>(a.tree, i < n && tame, ray, e0.w, h, doubt, wc);
	if (i < n && (doubt || !tame))
		traverse_leafwise<true, COUNT>(a.tree, ray, e0.w, lds_stack + threadIdx.x, depth, h, wc);
	if (i < n && h.prim < 0) {
		// src/main.cpp:569: the light's contribution joins the path radiance
		const float4 e2 = e[2];
		const uint32_t slot = __float_as_uint(e1.w);
		float4 C = a.col[slot];
		C.x += e2.x;
		C.y += e2.y;
		C.z += e2.z;
		a.col[slot] = C;
	}
	if (COUNT)
		count_flush(a.cnt, wc);
}

// Occlusion queries of one wave by the any-hit walk over the wide tree (traverse_wide.hpp).
//   tag = load(i, ray, limit)    the wave's i-th query
//   fetch(tag, ray, limit)       the same query again, by its tag
//   done(tag, occluded)          its answer
// A query that the walk cannot answer for certain is traced by the replaying traversal on the
// spot, by the lanes that have just finished (their LDS columns are empty; `stack` must hold
// max(cap, depth) entries per lane): such queries are too rare for a queue of their own.
constexpr uint32_t kShadowWalkRefillAt = 24, kShadowWalkQuorum = 24, kShadowWalkWaiting = 64;

template <class Grab, class Load, class Fetch, class Done>
RTA_DEV void occluded_by_walk(const DeviceTree &tree, uint32_t *stack, const int cap, const int depth, Grab grab, Load load,
                              Fetch fetch, Done done)
{
	WorkCount wc = {0, 0, 0};
	WideCensus cs = {0, 0, 0, 0, 0};
	traverse_wide_pulling<false, true>(
	    tree, kShadowWalkRefillAt, kShadowWalkQuorum, kShadowWalkWaiting, stack, cap, wc, cs, grab, load,
	    [&](bool finished, uint32_t tag, const HitRec &h, bool doubt) {
		    bool occluded = h.prim >= 0;
		    const bool again = finished && doubt;
		    if (__ballot(again)) {
			    Traversal<true, false> tr;
			    tr.rest();
			    if (again) {
				    Ray ray;
				    float limit;
				    fetch(tag, ray, limit);
				    tr.start(tree, ray, limit);
			    }
			    while (__ballot(tr.more)) {
				    if (tr.more)
					    tr.step(tree, stack, depth, wc);
			    }
			    if (again)
				    occluded = tr.result().prim >= 0;
		    }
		    if (finished)
			    done(tag, occluded);
	    });
}

// k_shadow for scenes with a wide tree: a wave takes 64 consecutive queue entries through the any-hit walk.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_shadow_walk(const RenderArgs a, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	const uint32_t shard = blockIdx.x % kShards, part = blockIdx.x / kShards;
	const uint32_t n = a.cnt->n_shadow[shard];
	if (part == 0 && threadIdx.x == 0 && n)
		atomicAdd(&a.cnt->shadow_rays, (unsigned long long)n);
	// a wave works through rays_per_lane x 64 consecutive entries, handing its lanes new queries as they finish (a short
	// queue is spread over more waves, as the bounce rays' is)
	const uint32_t spread = (n + a.spread_blocks * kBlock - 1) / (a.spread_blocks * kBlock);
	const uint32_t per_wave = 64u * (spread < a.rays_per_lane ? (spread ? spread : 1u) : a.rays_per_lane);
	const uint32_t begin = (part * (kBlock / 64) + threadIdx.x / 64) * per_wave;
	if (begin >= n)
		return;
	const float4 *queue = a.shadow + 3 * (size_t)shard * a.shard_cap;
	const auto fetch = [&](uint32_t i, Ray &ray, float &limit) {
		const float4 *e = queue + 3 * (size_t)i;
		const float4 e0 = e[0], e1 = e[1];
		ray = Ray{xyz(e0), xyz(e1)};
		limit = e0.w;
	};
	occluded_by_walk(
	    a.tree, lds_stack + threadIdx.x, (int)a.wide_cap, depth, OwnChunk{n - begin < per_wave ? n - begin : per_wave, 0u},
	    [&](uint32_t i, Ray &ray, float &limit) {
		    fetch(begin + i, ray, limit);
		    return begin + i;
	    },
	    fetch,
	    [&](uint32_t i, bool occluded) {
		    if (!occluded) {
			    // src/main.cpp:569: the light's contribution joins the path radiance
			    const float4 *e = queue + 3 * (size_t)i;
			    const float4 e1 = e[1], e2 = e[2];
			    const uint32_t slot = __float_as_uint(e1.w);
			    float4 C = a.col[slot];
			    C.x += e2.x;
			    C.y += e2.y;
			    C.z += e2.z;
			    a.col[slot] = C;
		    }
	    });
}

// ------------------------------------------------------------------ item weights

// How expensive will the parts of the tiles be?  One ray through the centre of
// every pixel; a part's weight is the number of its pixels that see geometry.
// Heavier parts are handed out first, so that the render does not end on a few
// waves that drew a long item late.  Only the schedule depends on this, never a
// result, so the ray needs none of the reference's arithmetic.
__global__ __launch_bounds__(kBlock) void k_part_weights(const RenderArgs a, uint32_t *weights, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;      // index in the padded pixel list
	const int32_t pixel = i < a.n_pad ? a.pixels[i] : -1;
	bool seen = false;
	// pixels none of whose samples can reach a leaf's box need no slot at all (k_sum_blocks knows their sums)
	const bool near = pixel >= 0 && !(a.sky_map && a.sky_map[pixel] <= kNoSampleReaches);
	if (near) {
		const uint32_t pu = (uint32_t)pixel % a.width, pv = (uint32_t)pixel / a.width;
		const V3 o = mk(a.cam.origin[0], a.cam.origin[1], a.cam.origin[2]);
		const V3 ul = mk(a.cam.upper_left[0], a.cam.upper_left[1], a.cam.upper_left[2]);
		const V3 du = mk(a.cam.du[0], a.cam.du[1], a.cam.du[2]);
		const V3 dv = mk(a.cam.dv[0], a.cam.dv[1], a.cam.dv[2]);
		const V3 target = (ul + du * ((float)pu + .5f)) + dv * ((float)pv + .5f);
		HitRec h;
		WorkCount wc = {0, 0, 0};
		traverse_dyn<true, false>(a.tree, Ray{o, unit(target - o)}, FLT_MAX, lds_stack + threadIdx.x, depth, h, wc);
		seen = h.prim >= 0;
	}
	// a wave covers one tile; its lanes split into lane_blocks parts of 64 / lane_blocks pixels
	const uint32_t part_pixels = 64 / a.lane_blocks, lane = threadIdx.x & 63;
	const unsigned long long votes = __ballot(seen), any_near = __ballot(near);
	if (i < a.n_pad && lane % part_pixels == 0) {
		const unsigned long long mask = part_pixels == 64 ? ~0ull : ((1ull << part_pixels) - 1) << lane;
		weights[(i / 64) * a.lane_blocks + lane / part_pixels] =
		    (any_near & mask) ? (uint32_t)__popcll(votes & mask) : kNothingToTrace;
	}
}

// ------------------------------------------------------------------ sky map

// Which samples need no ray at all?  A primitive is only tested once the ray has
// passed the box test of its leaf (src/main.cpp:381-401), so a ray that misses
// every leaf's box - by far more than binary32 rounding can move the slab
// quotients - ends in the sky.  For a pinhole camera all rays of the image start
// at one point, and a box is then missed by every ray whose point on the image
// plane lies outside the rectangle around the box's eight projected corners.
// A lens camera's rays start anywhere on the lens; seen from all of its points a
// corner covers a small parallelogram on the image plane (Projection::lens_x),
// and the rectangle is taken around those.
//
// k_cover_leaves marks the pixel-sized cells of the image plane (the frame and
// kSkyBorder cells around it) that some leaf's rectangle, widened by the margin,
// touches; gridDim.y waves per (pair record, child).
__global__ __launch_bounds__(kBlock) void k_cover_leaves(const TreeRef tree, const uint32_t n_pairs, const Projection pr,
                                                        const int map_w, const int map_h, uint8_t *cover)
{
	const uint32_t item = blockIdx.x * (kBlock / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
	if (item >= 2 * n_pairs)
		return;
	const float4 *rec = tree.pairs + 4 * (size_t)(item >> 1);
	const float4 q3 = rec[3];
	const uint32_t ref = (item & 1) ? __float_as_uint(q3.y) : __float_as_uint(q3.x);
	if (!(ref & kLeafBit))
		return;
	const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
	const float lo[3] = {(item & 1) ? q1.z : q0.x, (item & 1) ? q1.w : q0.y, (item & 1) ? q2.x : q0.z};
	const float hi[3] = {(item & 1) ? q2.y : q0.w, (item & 1) ? q2.z : q1.x, (item & 1) ? q2.w : q1.y};
	double x0 = 1e9, x1 = -1e9, y0 = 1e9, y1 = -1e9;
	bool everywhere = false;      // a corner that is not safely in front of the camera: no statement about this box
	for (int c = 0; c < 8; c++) {
		double w[3], depth = 0;
		for (int k = 0; k < 3; k++) {
			w[k] = (double)(((c >> k) & 1) ? hi[k] : lo[k]) - pr.o[k];
			depth += w[k] * pr.n[k];
		}
		const double scale = pr.h / depth;
		if (!(scale > 0) || !(scale < 1e6))
			everywhere = true;
		double x = 0, y = 0;
		for (int k = 0; k < 3; k++) {
			const double p = pr.o[k] + w[k] * scale - pr.ul[k];
			x += p * pr.du[k];
			y += p * pr.dv[k];
		}
		x /= pr.du2;
		y /= pr.dv2;
		if (!(x > -1e9 && x < 1e9 && y > -1e9 && y < 1e9))
			everywhere = true;
		// where the lens's other points see this corner (nothing for a pinhole)
		const double off = scale > 1 ? scale - 1 : 1 - scale;
		const double sx = off * pr.lens_x * (1 + 1e-9), sy = off * pr.lens_y * (1 + 1e-9);
		x0 = x - sx < x0 ? x - sx : x0, x1 = x + sx > x1 ? x + sx : x1;
		y0 = y - sy < y0 ? y - sy : y0, y1 = y + sy > y1 ? y + sy : y1;
	}
	if (everywhere) {
		x0 = y0 = -1e9;
		x1 = y1 = 1e9;
	}
	// widened by the margin (and by more than the rounding of these few binary64 operations)
	x0 -= pr.margin + 1e-9 * (x0 < 0 ? -x0 : x0), x1 += pr.margin + 1e-9 * (x1 < 0 ? -x1 : x1);
	y0 -= pr.margin + 1e-9 * (y0 < 0 ? -y0 : y0), y1 += pr.margin + 1e-9 * (y1 < 0 ? -y1 : y1);
	// cells [i, i + 1] x [j, j + 1] of the map, whose cell (kSkyBorder, kSkyBorder) is pixel (0, 0)
	const double lim = 1e8;
	const long long i0 = (long long)floor(x0 < -lim ? -lim : x0) + kSkyBorder, i1 = (long long)floor(x1 > lim ? lim : x1) + kSkyBorder;
	const long long j0 = (long long)floor(y0 < -lim ? -lim : y0) + kSkyBorder, j1 = (long long)floor(y1 > lim ? lim : y1) + kSkyBorder;
	const int ci0 = (int)(i0 < 0 ? 0 : i0), ci1 = (int)(i1 > map_w - 1 ? map_w - 1 : i1);
	const int cj0 = (int)(j0 < 0 ? 0 : j0), cj1 = (int)(j1 > map_h - 1 ? map_h - 1 : j1);
	if (ci1 < ci0 || cj1 < cj0)
		return;
	// the rectangle's rows are dealt to the gridDim.y waves that look at this leaf (most leaves cover a
	// few cells; a leaf that fills the frame would otherwise keep one wave busy for milliseconds)
	for (int row = cj0 + (int)blockIdx.y; row <= cj1; row += (int)gridDim.y)
		for (int col = ci0 + (int)lane; col <= ci1; col += 64)
			cover[(size_t)row * map_w + col] = 1;
}

// The pixel filter moves a sample by .5 sqrt(-2 ln u1) pixels from the pixel's
// centre (src/main.cpp:674-683).  If every covered cell is further than that
// from the centre, the sample's ray ends in the sky whatever its angle u2: that
// is u1 >= exp(-2 distance^2), stored here per pixel, rounded up (rounding only
// ever sends a sample the long way).  2 = never.  Cells beyond kSkyBorder are
// not looked at and count as covered.
__global__ __launch_bounds__(kBlock) void k_sky_map(const uint8_t *cover, const int width, const int height, float *sky_map)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= (uint32_t)width * (uint32_t)height)
		return;
	const int px = (int)(i % (uint32_t)width), py = (int)(i / (uint32_t)width), map_w = width + 2 * kSkyBorder;
	float d2 = (kSkyBorder + .5f) * (kSkyBorder + .5f);
	for (int dj = -kSkyBorder; dj <= kSkyBorder; dj++)
		for (int di = -kSkyBorder; di <= kSkyBorder; di++)
			if (cover[(size_t)(py + kSkyBorder + dj) * map_w + px + kSkyBorder + di]) {
				const float dx = di ? (float)(di < 0 ? -di : di) - .5f : 0.f, dy = dj ? (float)(dj < 0 ? -dj : dj) - .5f : 0.f;
				d2 = lesser(d2, dx * dx + dy * dy);
			}
	// __expf is good to a few ulp; the factor covers that
	sky_map[i] = d2 > 0.f ? __expf(-2.f * d2) * (1.f + 0x1p-10f) + 0x1p-120f : 2.f;
}

// ------------------------------------------------------------------ finish

// Pixel sum = block sums added in block order from zero (`resume`: from the sum
// of the blocks before these).  `pixels` is the padded per-tile list (64 entries
// per tile, -1 where the tile leaves the frame).
//
// A pixel none of whose samples can reach a leaf's box (sky_map, see k_sky_map) has block sums that are known
// without looking: every sample of it is the sky's colour (src/main.cpp:511-514 with throughput 1 and
// radiance 0) added to the block's sum in turn, whoever added it - k_advance sample by sample or a block at a
// time, or nobody: the parts of the image that only see the sky never enter the work lists and their block
// sums are never written.  The same additions in the same order are made here instead of reading them back
// (four fifths of dragon.txt's pixels: 3.4 GB of block sums per 1080p x 1024 spp frame neither written nor read).
__global__ __launch_bounds__(kBlock) void k_sum_blocks(const float4 *partial, const int32_t *pixels, uint32_t n_pad,
                                                       uint32_t n_blocks, float *frame, bool resume, const float *sky_map,
                                                       const V3 sky, uint32_t first_block, uint32_t sps, uint32_t spp)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_pad)
		return;
	const int32_t pixel = pixels[i];
	if (pixel < 0)
		return;
	float *dst = frame + (size_t)pixel * 3;
	float x = 0, y = 0, z = 0;
	if (resume) {
		x = dst[0];
		y = dst[1];
		z = dst[2];
	}
	if (sky_map && sky_map[pixel] <= kNoSampleReaches) {
		const V3 color = mk(0, 0, 0) + mk(1, 1, 1) * sky;
		V3 A = mk(0, 0, 0);
		uint32_t summed = 0;
		for (uint32_t b = 0; b < n_blocks; b++) {
			const uint32_t first = (first_block + b) * sps;
			const uint32_t len = spp - first < sps ? spp - first : sps;
			if (len != summed) {      // (all blocks but the image's last have sps samples)
				A = mk(0, 0, 0);
				for (uint32_t k = 0; k < len; k++) {
					A.x += color.x;
					A.y += color.y;
					A.z += color.z;
				}
				summed = len;
			}
			x += A.x;
			y += A.y;
			z += A.z;
		}
	} else
		for (uint32_t b = 0; b < n_blocks; b++) {
			const float4 p = partial[(size_t)b * n_pad + i];
			x += p.x;
			y += p.y;
			z += p.z;
		}
	dst[0] = x;
	dst[1] = y;
	dst[2] = z;
}

// src/main.cpp:692-706: mean, blue/green error markers, tone map, truncation,
// B,G,R byte order.
RTA_DEV uint8_t to_byte(float x)
{
	// float -> int as the x86 cvttss2si the reference compiles to: NaN and
	// out-of-range become INT_MIN, whose low byte is 0
	int v;
	if (!(x > -2147483648.f && x < 2147483648.f))
		v = (int)0x80000000;
	else
		v = (int)x;
	return (uint8_t)v;
}

__global__ __launch_bounds__(kBlock) void k_resolve(const float *frame, const int32_t *pixels, uint32_t n, float samples,
                                                    float exposure, uint8_t *bgr)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	if (pixels && pixels[i] < 0)
		return;
	const size_t p = pixels ? (size_t)pixels[i] : (size_t)i;
	V3 c = mk(frame[p * 3], frame[p * 3 + 1], frame[p * 3 + 2]) / samples;
	if (c.x < 0 || c.y < 0 || c.z < 0)
		c = mk(0, 0, 1);
	if (c.x != c.x || c.y != c.y || c.z != c.z)
		c = mk(0, 1, 0);
	bgr[p * 3 + 2] = to_byte(255 * tonemap(c.x, exposure));
	bgr[p * 3 + 1] = to_byte(255 * tonemap(c.y, exposure));
	bgr[p * 3 + 0] = to_byte(255 * tonemap(c.z, exposure));
}

__global__ __launch_bounds__(kBlock) void k_init_slots(const RenderArgs a, uint32_t n_slots)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_slots)
		return;
	a.ray1[i] = make_float4(0.f, 0.f, 1.f, __uint_as_float(kWaiting));
	if (a.hint)
		a.hint[i] = -1;
}

// ------------------------------------------------------------------ ray batches

// Ray batches: a wave owns a run of rays_per_lane x 64 consecutive rays (see
// traverse_pulling); small batches get one ray per lane so that they still
// spread over the whole GPU.
constexpr uint32_t kBatchRefillAt = 16;

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_batch_closest(const DeviceTree tree, const float *rays, uint64_t n,
                                                          rta_hit *hits, const int depth, Counters *cnt,
                                                          const uint32_t rays_per_lane, const uint32_t *order,
                                                          const uint32_t *count_from)
{
	extern __shared__ uint32_t lds_stack[];
	// (a list whose length only the device knows: the rays a walk handed on; n is then its capacity)
	if (count_from)
		n = *count_from;
	// wave w of the grid owns rays [w 64 kBatchRaysPerLane, (w + 1) 64 kBatchRaysPerLane)
	const uint64_t per_wave = 64 * (uint64_t)rays_per_lane;
	const uint64_t first = ((uint64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64) * per_wave;
	// (with count_from the grid may be smaller than the list: its waves stride over it)
	const uint64_t stride = (uint64_t)gridDim.x * (kBlock / 64) * per_wave;
	WorkCount wc = {0, 0, 0};
	for (uint64_t begin = first; begin < n; begin += stride) {
		// the wave's rays: numbers begin.. of the batch, or of the tracing order
		const float *mine = order ? rays : rays + begin * 6;
		rta_hit *out = order ? hits : hits + begin;
		const uint32_t *which = order ? order + begin : nullptr;
		traverse_pulling<false, COUNT, false>(
		    tree, kBatchRefillAt, lds_stack + threadIdx.x, depth, wc,
		    OwnChunk{(uint32_t)(n - begin < per_wave ? n - begin : per_wave), 0u},
		    [&](uint32_t i, Ray &ray, float &limit) {
			    if (which)
				    i = which[i];
			    const float *r = mine + (size_t)i * 6;
			    ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
			    limit = FLT_MAX;
			    return i;
		    },
		    [&](uint32_t i, const HitRec &h) { out[i] = rta_hit{h.prim, h.t, h.u, h.v}; });
	}
	if (COUNT)
		count_flush(cnt, wc);
}

// The same batch by the certified walk over the wide tree (traverse_wide.hpp).  The rays it cannot
// answer are listed in `doubts` (ray numbers, *n_doubts of them) for k_batch_closest to trace.
constexpr uint32_t kBatchWalkRefillAt = 24, kBatchWalkQuorum = 24, kBatchWalkWaiting = 64;

// (seven waves per SIMD: at eight, 64 registers, the kernel spills ten of them in its loop)
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_batch_walk(
    const DeviceTree tree, const float *rays, uint64_t n, rta_hit *hits, const int cap, const uint32_t rays_per_lane,
    const uint32_t *order, uint32_t *doubts, uint32_t *n_doubts, Counters *cnt)
{
	extern __shared__ uint32_t lds_stack[];
	const uint64_t per_wave = 64 * (uint64_t)rays_per_lane;
	const uint64_t begin = ((uint64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64) * per_wave;
	if (begin >= n)
		return;
	const uint32_t *which = order ? order + begin : nullptr;
	WorkCount wc = {0, 0, 0};
	WideCensus cs = {0, 0, 0, 0, 0};
	traverse_wide_pulling<COUNT>(
	    tree, kBatchWalkRefillAt, kBatchWalkQuorum, kBatchWalkWaiting, lds_stack + threadIdx.x, cap, wc, cs,
	    OwnChunk{(uint32_t)(n - begin < per_wave ? n - begin : per_wave), 0u},
	    [&](uint32_t i, Ray &ray, int32_t &hint) {
		    const uint32_t number = which ? which[i] : (uint32_t)begin + i;
		    const float *r = rays + (size_t)number * 6;
		    ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
		    hint = -1;
		    return number;
	    },
	    [&](bool finished, uint32_t number, const HitRec &h, bool doubt) {
		    if (finished && !doubt)
			    hits[number] = rta_hit{h.prim, h.t, h.u, h.v};
		    // the wave's rays in doubt join the list with one atomic (rare: most waves have none)
		    const uint32_t at = queue_push(finished && doubt, n_doubts);
		    if (finished && doubt)
			    doubts[at] = number;
	    });
	if (COUNT)
		count_flush(cnt, wc);
}

// The same batch, 64 consecutive rays at a time through the tree TOGETHER (traverse_packet.hpp): for batches
// whose neighbouring rays travel together (a camera's).  One packet per wave; no LDS.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_batch_packet(const DeviceTree tree, const float *rays, uint64_t n, rta_hit *hits,
                                                         const uint32_t *order, uint32_t *doubts, uint32_t *n_doubts, Counters *cnt)
{
	// a wave's 64 lanes: 64 consecutive rays of the batch, or of the tracing order
	const uint64_t place = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	const bool mine = place < n;
	const uint64_t number = mine && order ? order[place] : place;
	Ray ray = {mk(0, 0, 0), mk(1, 1, 1)};
	if (mine) {
		const float *r = rays + (size_t)number * 6;
		ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
	}
	WorkCount wc = {0, 0, 0};
	walk_packet<COUNT>(tree, mine, ray, (uint32_t)number, wc, [&](bool finished, uint32_t which, const HitRec &h, bool doubt) {
		if (finished && !doubt)
			hits[which] = rta_hit{h.prim, h.t, h.u, h.v};
		const uint32_t at = queue_push(finished && doubt, n_doubts);
		if (finished && doubt)
			doubts[at] = which;
	});
	if (COUNT)
		count_flush(cnt, wc);
}

// The same batch for a scene of a few dozen primitives (TreeRef::flat_prims): one ray per lane, every primitive
// tested (traverse_flat.hpp); rays that need the division-based box test are in doubt from the start.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_batch_flat(const DeviceTree tree, const float *rays, uint64_t n, rta_hit *hits,
                                                       uint32_t *doubts, uint32_t *n_doubts, Counters *cnt)
{
	const uint64_t number = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	const bool mine = number < n;
	Ray ray = {mk(0, 0, 0), mk(1, 1, 1)};
	if (mine) {
		const float *r = rays + (size_t)number * 6;
		ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
	}
	const bool tame = ray_is_tame(ray);
	WorkCount wc = {0, 0, 0};
	HitRec h;
	bool doubt;
	scan_flat<false, COUNT>(tree, mine && tame, ray, FLT_MAX, h, doubt, wc);
	doubt = mine && (doubt || !tame);
	if (mine && !doubt)
		hits[number] = rta_hit{h.prim, h.t, h.u, h.v};
	const uint32_t at = queue_push(doubt, n_doubts);
	if (doubt)
		doubts[at] = (uint32_t)number;
	if (COUNT)
		count_flush(cnt, wc);
}

// Occlusion queries likewise; what the scan leaves in doubt is traced by the replaying traversal on the spot.
__global__ __launch_bounds__(kBlock) void k_batch_occluded_flat(const DeviceTree tree, const float *rays, const float *tmax,
                                                                uint64_t n, uint8_t *out, const int depth)
{
	extern __shared__ uint32_t lds_stack[];
	const uint64_t number = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	const bool mine = number < n;
	Ray ray = {mk(0, 0, 0), mk(1, 1, 1)};
	float limit = 0.f;
	if (mine) {
		const float *r = rays + (size_t)number * 6;
		ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
		limit = tmax[number];
	}
	const bool tame = ray_is_tame(ray);
	WorkCount wc = {0, 0, 0};
	HitRec h;
	bool doubt;
	scan_flat<true, false>(tree, mine && tame, ray, limit, h, doubt, wc);
	if (mine && (doubt || !tame))
		traverse_dyn<true, false>(tree, ray, limit, lds_stack + threadIdx.x, depth, h, wc);
	if (mine)
		out[number] = h.prim >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_batch_occluded(const DeviceTree tree, const float *rays, const float *tmax,
                                                           uint64_t n, uint8_t *out, const int depth,
                                                           const uint32_t rays_per_lane, const uint32_t *order)
{
	extern __shared__ uint32_t lds_stack[];
	const uint64_t per_wave = 64 * (uint64_t)rays_per_lane;
	const uint64_t begin = ((uint64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64) * per_wave;
	if (begin >= n)
		return;
	const float *mine = order ? rays : rays + begin * 6, *limits = order ? tmax : tmax + begin;
	uint8_t *flags = order ? out : out + begin;
	const uint32_t *which = order ? order + begin : nullptr;
	WorkCount wc = {0, 0, 0};
	traverse_pulling<true, false, false>(
	    tree, kBatchRefillAt, lds_stack + threadIdx.x, depth, wc,
	    OwnChunk{(uint32_t)(n - begin < per_wave ? n - begin : per_wave), 0u},
	    [&](uint32_t i, Ray &ray, float &limit) {
		    if (which)
			    i = which[i];
		    const float *r = mine + (size_t)i * 6;
		    ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
		    limit = limits[i];
		    return i;
	    },
	    [&](uint32_t i, const HitRec &h) { flags[i] = h.prim >= 0 ? 1 : 0; });
}

// k_batch_occluded by the any-hit walk over the wide tree.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_batch_occluded_walk(
    const DeviceTree tree, const float *rays, const float *tmax, uint64_t n, uint8_t *out, const int cap, const int depth,
    const uint32_t rays_per_lane, const uint32_t *order)
{
	extern __shared__ uint32_t lds_stack[];
	const uint64_t per_wave = 64 * (uint64_t)rays_per_lane;
	const uint64_t begin = ((uint64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64) * per_wave;
	if (begin >= n)
		return;
	const uint32_t *which = order ? order + begin : nullptr;
	const auto fetch = [&](uint32_t number, Ray &ray, float &limit) {
		const float *r = rays + (size_t)number * 6;
		ray = Ray{mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5])};
		limit = tmax[number];
	};
	occluded_by_walk(
	    tree, lds_stack + threadIdx.x, cap, depth, OwnChunk{(uint32_t)(n - begin < per_wave ? n - begin : per_wave), 0u},
	    [&](uint32_t i, Ray &ray, float &limit) {
		    const uint32_t number = which ? which[i] : (uint32_t)begin + i;
		    fetch(number, ray, limit);
		    return number;
	    },
	    fetch, [&](uint32_t number, bool occluded) { out[number] = occluded ? 1 : 0; });
}

}  // namespace

// ------------------------------------------------------------------ launchers

static inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static inline uint32_t batch_rays_per_lane(uint64_t n) { return n >= (8u << 20) ? 4u : 1u; }
static inline unsigned batch_blocks(uint64_t n) { return blocks_for((n + batch_rays_per_lane(n) - 1) / batch_rays_per_lane(n)); }
static inline size_t stack_bytes(int depth) { return (size_t)depth * kBlock * sizeof(uint32_t); }

void launch_init_slots(hipStream_t s, const RenderArgs &a, uint32_t n_slots)
{
	hipLaunchKernelGGL(k_init_slots, dim3(blocks_for(n_slots)), dim3(kBlock), 0, s, a, n_slots);
}

void launch_part_weights(hipStream_t s, const RenderArgs &a, uint32_t *weights, int depth)
{
	hipLaunchKernelGGL(k_part_weights, dim3(blocks_for(a.n_pad)), dim3(kBlock), stack_bytes(depth), s, a, weights, depth);
}

void launch_advance(hipStream_t s, const RenderArgs &a, int par)
{
	if (a.n_lights > 0)
		hipLaunchKernelGGL(k_advance<true>, dim3(blocks_for(a.n_slots)), dim3(kBlock), 0, s, a, par);
	else
		hipLaunchKernelGGL(k_advance<false>, dim3(blocks_for(a.n_slots)), dim3(kBlock), 0, s, a, par);
}

void launch_sky_map(hipStream_t s, const TreeRef &tree, uint32_t n_pairs, const Projection &pr, uint32_t width,
                    uint32_t height, uint8_t *cover, float *sky_map)
{
	const int map_w = (int)width + 2 * kSkyBorder, map_h = (int)height + 2 * kSkyBorder;
	(void)hipMemsetAsync(cover, 0, (size_t)map_w * map_h, s);
	// (a tree of a few leaves can have leaves that fill the frame; a tree of millions has waves enough)
	const unsigned waves_per_leaf = n_pairs < (1u << 16) ? 32 : 1;
	hipLaunchKernelGGL(k_cover_leaves, dim3((2 * n_pairs + kBlock / 64 - 1) / (kBlock / 64), waves_per_leaf), dim3(kBlock), 0, s,
	                   tree, n_pairs, pr, map_w, map_h, cover);
	hipLaunchKernelGGL(k_sky_map, dim3(blocks_for((uint64_t)width * height)), dim3(kBlock), 0, s, cover, (int)width,
	                   (int)height, sky_map);
}

void launch_intersect(hipStream_t s, const RenderArgs &a, int par, int depth, bool count, int stages)
{
	// (the same arithmetic as pulling_grid / per_kind in the kernels)
	const unsigned blocks = kShards * ((a.grid_cap + kBlock * a.rays_per_lane - 1) / (kBlock * a.rays_per_lane)) + kShards * (a.grid_cap / kBlock);
	const unsigned leafwise_blocks = 2 * kShards * (a.grid_cap / kBlock);
	const unsigned camera_blocks = kShards * (a.grid_cap / kBlock);
	if (a.camera_cells && (stages & kStageCamera)) {
		// Camera rays from their cells' lists first (their bounce rays may join this step's); the traversal kernel
		// that follows is launched without its workgroups for camera rays (they come last in every grid).
		if (count && a.n_lights > 0)
			hipLaunchKernelGGL((k_camera_cells<true, true>), dim3(camera_blocks), dim3(kBlock), 0, s, a, par);
		else if (count)
			hipLaunchKernelGGL((k_camera_cells<true, false>), dim3(camera_blocks), dim3(kBlock), 0, s, a, par);
		else if (a.n_lights > 0)
			hipLaunchKernelGGL((k_camera_cells<false, true>), dim3(camera_blocks), dim3(kBlock), 0, s, a, par);
		else
			hipLaunchKernelGGL((k_camera_cells<false, false>), dim3(camera_blocks), dim3(kBlock), 0, s, a, par);
	}
	if (!(stages & kStageWalk))
		;
	else if (a.tree.flat_prims && count)
		hipLaunchKernelGGL(k_intersect_flat<true>, dim3(a.camera_cells ? camera_blocks : leafwise_blocks), dim3(kBlock), 0, s, a, par);
	else if (a.tree.flat_prims)
		hipLaunchKernelGGL(k_intersect_flat<false>, dim3(a.camera_cells ? camera_blocks : leafwise_blocks), dim3(kBlock), 0, s, a, par);
	else if (a.leafwise && count)
		hipLaunchKernelGGL(k_intersect_leafwise<true>, dim3(a.camera_cells ? camera_blocks : leafwise_blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
	else if (a.leafwise)
		hipLaunchKernelGGL(k_intersect_leafwise<false>, dim3(a.camera_cells ? camera_blocks : leafwise_blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
	else if (a.wide_cap && a.camera_cells) {
		const size_t lds = stack_bytes(std::max(depth, (int)a.wide_cap));
		if (count)
			hipLaunchKernelGGL((k_intersect<true, true>), dim3(blocks - camera_blocks), dim3(kBlock), lds, s, a, par, depth);
		else
			hipLaunchKernelGGL((k_intersect<false, true>), dim3(blocks - camera_blocks), dim3(kBlock), lds, s, a, par, depth);
	} else if (a.wide_cap) {
		// one LDS column per lane serves the exact traversal's stack (camera rays) and the certified walk
		const size_t lds = stack_bytes(std::max(depth, (int)a.wide_cap));
		static const bool split = getenv("RTA_SPLIT_INTERSECT") != nullptr;      // measurement: the two parts as two launches
		if (count)
			hipLaunchKernelGGL((k_intersect<true, true>), dim3(blocks), dim3(kBlock), lds, s, a, par, depth);
		else if (split) {
			hipLaunchKernelGGL((k_intersect<false, true, 1>), dim3(blocks), dim3(kBlock), lds, s, a, par, depth);
			hipLaunchKernelGGL((k_intersect<false, true, 2>), dim3(blocks), dim3(kBlock), lds, s, a, par, depth);
		} else
			hipLaunchKernelGGL((k_intersect<false, true>), dim3(blocks), dim3(kBlock), lds, s, a, par, depth);
	} else if (count)
		hipLaunchKernelGGL((k_intersect<true, false>), dim3(a.camera_cells ? blocks - camera_blocks : blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
	else if (a.camera_cells)
		hipLaunchKernelGGL((k_intersect<false, false>), dim3(blocks - camera_blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
	else if (getenv("RTA_SPLIT_INTERSECT")) {
		hipLaunchKernelGGL((k_intersect<false, false, 1>), dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
		hipLaunchKernelGGL((k_intersect<false, false, 2>), dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
	} else
		hipLaunchKernelGGL((k_intersect<false, false>), dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
	// on a tame tree odd rays are rare: one workgroup per sub-queue is plenty
	const unsigned odd_blocks = a.tree.tame ? kShards : kShards * (a.grid_cap / kBlock);
	if (!(stages & kStageOdd))
		;
	else if (count)
		hipLaunchKernelGGL(k_intersect_odd<true>, dim3(odd_blocks), dim3(kBlock), stack_bytes(depth), s, a, par, depth);
	else
		hipLaunchKernelGGL(k_intersect_odd<false>, dim3(odd_blocks), dim3(kBlock), stack_bytes(depth), s, a, par,
		                   depth);
}

void launch_shadow(hipStream_t s, const RenderArgs &a, int depth, bool count)
{
	// (the same bound per sub-queue as the ray queues')
	const unsigned blocks = kShards * (a.grid_cap / kBlock);
	if (a.wide_cap && !count)
		hipLaunchKernelGGL(k_shadow_walk, dim3(blocks), dim3(kBlock), stack_bytes(std::max(depth, (int)a.wide_cap)), s, a, depth);
	else if (a.tree.flat_prims && count)
		hipLaunchKernelGGL(k_shadow_flat<true>, dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, depth);
	else if (a.tree.flat_prims)
		hipLaunchKernelGGL(k_shadow_flat<false>, dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, depth);
	else if (count && a.leafwise)
		hipLaunchKernelGGL((k_shadow<true, true>), dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, depth);
	else if (count)
		hipLaunchKernelGGL((k_shadow<true, false>), dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, depth);
	else if (a.leafwise)
		hipLaunchKernelGGL((k_shadow<false, true>), dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, depth);
	else
		hipLaunchKernelGGL((k_shadow<false, false>), dim3(blocks), dim3(kBlock), stack_bytes(depth), s, a, depth);
}

void launch_sum_blocks(hipStream_t s, const RenderArgs &a, uint32_t first_block, uint32_t n_blocks, float *frame, bool resume)
{
	hipLaunchKernelGGL(k_sum_blocks, dim3(blocks_for(a.n_pad)), dim3(kBlock), 0, s, a.partial, a.pixels, a.n_pad, n_blocks,
	                   frame, resume, a.sky_map, V3{a.sky[0], a.sky[1], a.sky[2]}, first_block, a.sps, a.spp);
}

void launch_resolve(hipStream_t s, const float *frame, const int32_t *pixels, uint32_t n, float samples, float exposure,
                    uint8_t *bgr)
{
	hipLaunchKernelGGL(k_resolve, dim3(blocks_for(n)), dim3(kBlock), 0, s, frame, pixels, n, samples, exposure, bgr);
}

void launch_batch_closest(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, int depth,
                          Counters *cnt, const uint32_t *order, const uint32_t *count_from)
{
	if (cnt)
		hipLaunchKernelGGL(k_batch_closest<true>, dim3(batch_blocks(n)), dim3(kBlock), stack_bytes(depth), s, tree, rays,
		                   n, hits, depth, cnt, batch_rays_per_lane(n), order, count_from);
	else if (count_from)
		// one ray per lane and a grid for up to a 64th of the capacity (the list is short); workgroups beyond the
		// list's end leave at once, and a longer list is worked through in strides
		hipLaunchKernelGGL(k_batch_closest<false>, dim3(std::max(1u, blocks_for(n / 64))), dim3(kBlock), stack_bytes(depth), s,
		                   tree, rays, n, hits, depth, cnt, 1u, order, count_from);
	else
		hipLaunchKernelGGL(k_batch_closest<false>, dim3(batch_blocks(n)), dim3(kBlock), stack_bytes(depth), s, tree,
		                   rays, n, hits, depth, cnt, batch_rays_per_lane(n), order, count_from);
}

void launch_batch_walk(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, int cap,
                       const uint32_t *order, uint32_t *doubts, uint32_t *n_doubts, Counters *cnt)
{
	// long runs per wave once the batch fills the GPU several times over
	const uint32_t rays_per_lane = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, n >> 20));
	const dim3 grid(blocks_for((n + rays_per_lane - 1) / rays_per_lane));
	(void)hipMemsetAsync(n_doubts, 0, sizeof(uint32_t), s);
	if (cnt)
		hipLaunchKernelGGL(k_batch_walk<true>, grid, dim3(kBlock), stack_bytes(cap), s, tree, rays, n, hits, cap, rays_per_lane,
		                   order, doubts, n_doubts, cnt);
	else
		hipLaunchKernelGGL(k_batch_walk<false>, grid, dim3(kBlock), stack_bytes(cap), s, tree, rays, n, hits, cap, rays_per_lane,
		                   order, doubts, n_doubts, cnt);
}

void launch_batch_flat(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, uint32_t *doubts,
                       uint32_t *n_doubts, Counters *cnt)
{
	(void)hipMemsetAsync(n_doubts, 0, sizeof(uint32_t), s);
	if (cnt)
		hipLaunchKernelGGL(k_batch_flat<true>, dim3(blocks_for(n)), dim3(kBlock), 0, s, tree, rays, n, hits, doubts, n_doubts, cnt);
	else
		hipLaunchKernelGGL(k_batch_flat<false>, dim3(blocks_for(n)), dim3(kBlock), 0, s, tree, rays, n, hits, doubts, n_doubts, cnt);
}

void launch_batch_occluded_flat(hipStream_t s, const TreeRef &tree, const float *rays, const float *tmax, uint64_t n, uint8_t *out,
                                int depth)
{
	hipLaunchKernelGGL(k_batch_occluded_flat, dim3(blocks_for(n)), dim3(kBlock), stack_bytes(depth), s, tree, rays, tmax, n, out, depth);
}

void launch_batch_packet(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, const uint32_t *order,
                         uint32_t *doubts, uint32_t *n_doubts, Counters *cnt)
{
	(void)hipMemsetAsync(n_doubts, 0, sizeof(uint32_t), s);
	if (cnt)
		hipLaunchKernelGGL(k_batch_packet<true>, dim3(blocks_for(n)), dim3(kBlock), 0, s, tree, rays, n, hits, order, doubts, n_doubts, cnt);
	else
		hipLaunchKernelGGL(k_batch_packet<false>, dim3(blocks_for(n)), dim3(kBlock), 0, s, tree, rays, n, hits, order, doubts, n_doubts, cnt);
}

void launch_batch_occluded_walk(hipStream_t s, const TreeRef &tree, const float *rays, const float *tmax, uint64_t n,
                                uint8_t *out, int cap, int depth, const uint32_t *order)
{
	const uint32_t rays_per_lane = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, n >> 20));
	hipLaunchKernelGGL(k_batch_occluded_walk, dim3(blocks_for((n + rays_per_lane - 1) / rays_per_lane)), dim3(kBlock),
	                   stack_bytes(std::max(cap, depth)), s, tree, rays, tmax, n, out, cap, depth, rays_per_lane, order);
}

void launch_batch_occluded(hipStream_t s, const TreeRef &tree, const float *rays, const float *tmax, uint64_t n,
                           uint8_t *out, int depth, const uint32_t *order)
{
	hipLaunchKernelGGL(k_batch_occluded, dim3(batch_blocks(n)), dim3(kBlock), stack_bytes(depth), s, tree, rays, tmax, n,
	                   out, depth, batch_rays_per_lane(n), order);
}

}  // namespace rta
