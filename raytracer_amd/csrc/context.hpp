// The device context behind rta_context: the GPU copies of a scene, the
// wavefront workspace and the streams.  Shared by the translation units that
// implement the device half of include/raytracer_amd.h.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "kernels.hpp"

namespace rta {

void set_global_error(const std::string &text);      // api.cpp
const char *get_global_error();

constexpr int kMaxGroups = 8;             // slot groups = concurrent HIP streams per render

struct DeviceBuffer {
	void *ptr = nullptr;
	size_t bytes = 0;

	~DeviceBuffer() { release(); }
	void release()
	{
		if (ptr)
			(void)hipFree(ptr);
		ptr = nullptr;
		bytes = 0;
	}
	// grow-only
	hipError_t reserve(size_t want)
	{
		if (want <= bytes)
			return hipSuccess;
		release();
		hipError_t e = hipMalloc(&ptr, want);
		if (e == hipSuccess)
			bytes = want;
		return e;
	}
	template <typename T>
	T *as() const { return static_cast<T *>(ptr); }
};

}  // namespace rta

struct rta_context {
	int device = -1;
	hipStream_t stream = nullptr;         // group 0 and everything that is not a wavefront step
	hipStream_t extra[rta::kMaxGroups - 1] = {};  // the other slot groups
	std::string error;

	// scene
	bool have_scene = false;
	rta::DeviceBuffer pairs, prims, shade_prims, materials, lights, ref_nodes;
	rta::DeviceBuffer wide_nodes, leaf_boxes;     // the 4-wide tree of the certified traversal (wide_tree.hpp)
	bool have_wide = false;
	bool have_flat_boxes = false;                 // leaf_boxes alone is there (a scene with more spheres than `loose` lists)
	bool have_leaf_boxes = false;                 // leaf_boxes and loose are there (with a wide tree, and where only the tree was refused)
	float wide_center[3] = {0, 0, 0}, wide_reach2 = 0, wide_reach = 0;
	int wide_depth = 0;
	uint32_t wide_index_mask = 0;
	float wide_span = 1;
	uint32_t n_loose = 0;                     // the wide tree's spheres
	rta::DeviceBuffer loose;
	rta::DeviceBuffer soa_nodes;      // variant 5 only, made from `pairs` on first use
	uint32_t n_pairs = 0;
	bool have_soa = false;
	uint32_t root_ref = 0;
	bool tame = false;           // box coordinates allow the reciprocal slab test (traverse.hpp)
	bool box_known = false;      // `box` is the box around the root's two children
	float box[6] = {0, 0, 0, 0, 0, 0};
	int stack_depth = 1;
	int32_t n_prims = 0, n_lights = 0, n_nodes = 0;
	uint32_t wide_n_nodes = 0;

	// wavefront workspace
	rta::DeviceBuffer ray0, ray1, hit, thr, col, rng, acc, meta, hint, q_ray, q_bounce, q_camera, shadow, partial, pixels, frame, bgr, counters;
	rta::DeviceBuffer part_weights, part_order;
	rta::DeviceBuffer cover, sky_map;         // see k_sky_map
	rta::DeviceBuffer cell_counts, cell_starts, cell_list, cell_keys, cell_unsorted, cell_temp;     // per-cell triangle lists of the prepared view (cells.hip)
	rta::DeviceBuffer ray_order;              // see tracing_order
	rta::DeviceBuffer doubts;                 // rays a batch's certified walk hands on to the exact traversal
	rta::DeviceBuffer io_a, io_b, io_c;       // staging for host-pointer calls
	rta::Counters *host_counters = nullptr;   // pinned, kMaxGroups entries
	hipEvent_t batch_ev[4] = {};              // around the stages of a ray batch: order, walk, replay (rta_last_batch_stats)
	rta_batch_stats batch = {};               // what the last ray batch did

	// What a render prepares before its first wavefront step depends only on the
	// scene, the camera, the frame size and the shard: the owned pixel list, the
	// map of samples that need no ray, the work list heaviest first and the list of
	// sky-only parts.  A render with the same key as the one before (the next frame
	// of a progressive render, the next step of a benchmark) finds them in the
	// buffers where that render left them.
	struct Prepared {
		bool valid = false;
		rta_camera cam;
		int32_t width = 0, height = 0, shard_index = 0, shard_count = 0;
		uint32_t lane_blocks = 0;
		bool with_sky_map = false;
		bool want_cells = false, with_cells = false;      // camera rays from per-cell triangle lists: asked for, built
		uint32_t cell_entries = 0;
		uint32_t n_listed = 0, n_far = 0;
		uint64_t far_pixels = 0;
	} prepared;

	int fail(int code, const std::string &text)
	{
		error = text;
		return code;
	}
	int hip_fail(hipError_t e, const char *what)
	{
		error = std::string(what) + ": " + hipGetErrorString(e);
		return RTA_E_DEVICE;
	}
};

#define HIP_TRY(ctx, call)                                                  \
	do {                                                                    \
		hipError_t e__ = (call);                                            \
		if (e__ != hipSuccess)                                              \
			return (ctx)->hip_fail(e__, #call);                             \
	} while (0)

// context.cpp: rta_trace_closest with the certified walk switched on or off
extern "C" int rta_internal_trace_closest(rta_context *ctx, const float *rays, int64_t n, rta_hit *hits, int mem, int walk);

namespace rta {

// the uploaded tree as the kernels take it
TreeRef tree_of(const rta_context *ctx);

// static_trace.hip: looks up the CURRENT device's copy of the library's static
// workspace array (once per device); the device must be `device`
hipError_t static_scratch_init(int device);

}  // namespace rta
