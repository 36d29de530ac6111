// rta_trace_closest_static: closest hit of a device-resident ray batch with no
// allocation of any kind.  The batch is traced in the order of its rays' origins
// (sort_rays.hip), which needs per-ray workspace - sort keys, ray numbers, the
// sorted order and the radix sort's own temporary storage.  rta_trace_closest
// takes that workspace from a grow-only hipMalloc'ed buffer of the context;
// here it is a file-scope array in device memory, sized for kStaticMaxRays rays,
// that every GPU the library runs on holds from the moment the code object is
// loaded until the process ends.
//
// In host code the array's NAME is only a handle for hipGetSymbolAddress: each
// GPU has its own copy at its own address, looked up once per device
// (static_scratch_init, called by rta_create with that device current, and
// again by the entry point if that look-up failed) and kept in
// g_scratch[device] until the process ends (a hipDeviceReset by the caller
// invalidates it along with every context).  Everything below passes that pointer.
//
// All calls on one GPU share that GPU's array, so two such calls that overlap
// in time on one GPU race, also from different contexts.  The array keeps the
// last call's contents; the only word that must start at zero (the count of
// out-of-order keys that k_ray_keys adds to, later the count of rays that the
// certified walk hands on) is cleared by launch_ray_keys and by launch_batch_walk.

#include <hip/hip_runtime.h>

#include <mutex>

#include "context.hpp"

namespace rta {

namespace {

constexpr uint32_t kStaticMaxRays = RTA_STATIC_MAX_RAYS;
constexpr size_t kEach = (size_t)kStaticMaxRays * sizeof(uint32_t);      // one array of 4-byte entries (a multiple of 256)
constexpr size_t kTempBytes = (size_t)8 << 20;                           // for the radix sort; checked per call
constexpr size_t kScratchBytes = 4 * kEach + kTempBytes + 256;

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_lock;
void *g_scratch[kMaxDevices] = {};      // device -> address of that device's `scratch`

}  // namespace

hipError_t static_scratch_init(int device)
{
	if (device < 0 || device >= kMaxDevices)
		return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> hold(g_lock);
	if (g_scratch[device])
		return hipSuccess;
	int current = -1;
	hipError_t e = hipGetDevice(&current);
	if (e != hipSuccess)
		return e;
	if (current != device)
		return hipErrorInvalidDevice;      // the address is that of the CURRENT device's copy
	void *ptr = nullptr;
	e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
	if (e != hipSuccess)
		return e;
	if (!ptr)
		return hipErrorInvalidSymbol;
	g_scratch[device] = ptr;
	return hipSuccess;
}

}  // namespace rta

using namespace rta;

extern "C" int rta_trace_closest_static(rta_context *ctx, const float *rays, int64_t n, rta_hit *hits)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (!ctx->have_scene)
		return ctx->fail(RTA_E_NO_SCENE, "no scene uploaded");
	if (n < 0 || (n > 0 && (!rays || !hits)))
		return ctx->fail(RTA_E_INVALID, "bad ray batch arguments");
	if (n > (int64_t)kStaticMaxRays)
		return ctx->fail(RTA_E_LIMIT, "rta_trace_closest_static takes at most " + std::to_string(kStaticMaxRays) +
		                                  " rays per call (its workspace is a fixed array); split the batch or use rta_trace_closest");
	if (n == 0)
		return RTA_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	// (looked up when the context was created; if that failed, once more here: a look-up allocates nothing)
	if (hipError_t e = static_scratch_init(ctx->device); e != hipSuccess)
		return ctx->fail(RTA_E_DEVICE, std::string("the static workspace of this GPU was not found: hipGetSymbolAddress: ") + hipGetErrorString(e));
	char *base = nullptr;
	{
		std::lock_guard<std::mutex> hold(g_lock);
		base = (char *)g_scratch[ctx->device];
	}
	const uint32_t count = (uint32_t)n;
	const uint32_t *order = nullptr;
	if (ctx->box_known) {
		const size_t temp = ray_order_temp_bytes(count);
		if (temp > kTempBytes)
			return ctx->fail(RTA_E_LIMIT, "the radix sort asks for more temporary storage than the static workspace holds");
		uint32_t *keys = (uint32_t *)base, *keys_sorted = (uint32_t *)(base + kEach), *index = (uint32_t *)(base + 2 * kEach),
		         *sorted = (uint32_t *)(base + 3 * kEach), *descents = (uint32_t *)(base + 4 * kEach + kTempBytes);
		launch_ray_keys(ctx->stream, rays, count, ctx->box, ctx->box + 3, keys, index, descents);
		HIP_TRY(ctx, launch_ray_sort(ctx->stream, count, keys, keys_sorted, index, sorted, base + 4 * kEach, temp));
		order = sorted;
	}
	if (ctx->have_wide && order) {
		// The certified walk first (as rta_trace_closest): once the order is sorted, the keys' array is free to
		// take the list of the rays that the walk hands on, and the word that counted the out-of-order keys
		// their number (cleared by launch_batch_walk); the replaying traversal then runs over that list.
		uint32_t *doubts = (uint32_t *)base, *n_doubts = (uint32_t *)(base + 4 * kEach + kTempBytes);
		launch_batch_walk(ctx->stream, tree_of(ctx), rays, (uint64_t)n, hits, kBatchWalkCap, order, doubts, n_doubts);
		launch_batch_closest(ctx->stream, tree_of(ctx), rays, (uint64_t)n, hits, ctx->stack_depth, nullptr, doubts, n_doubts);
	} else
		launch_batch_closest(ctx->stream, tree_of(ctx), rays, (uint64_t)n, hits, ctx->stack_depth, nullptr, order);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return RTA_OK;
}
