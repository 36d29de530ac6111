// Device context: owns the GPU copies of a scene, the wavefront workspace and
// the stream, validates what is uploaded, and drives the kernels of kernels.hip.
// Implements the device half of include/raytracer_amd.h.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_build.hpp"
#include "context.hpp"
#include "host_scene.hpp"
#include "kernels.hpp"
#include "upload_device.hpp"
#include "wide_tree.hpp"

namespace rta {

namespace {

constexpr uint32_t kLeaf = 0x80000000u;
constexpr uint32_t kSphereFlag = 1u, kLastFlag = 2u;
constexpr int kMaxDepth = 64;             // the reference's traversal stack (src/main.cpp:339)
constexpr int kTile = 8;                  // pixels per tile side; 64 pixels = one wave64

}  // namespace

}  // namespace rta

using namespace rta;

extern "C" {

int rta_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int rta_create(int device, rta_context **out)
{
	if (!out)
		return RTA_E_INVALID;
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
		set_global_error(e != hipSuccess
		                     ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e)
		                     : "no such GPU device " + std::to_string(device) + " (" + std::to_string(n) + " present)");
		return RTA_E_NO_DEVICE;
	}
	hipDeviceProp_t prop;
	if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
		set_global_error(std::string("hipSetDevice: ") + hipGetErrorString(e));
		return RTA_E_NO_DEVICE;
	}
	if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
		set_global_error(std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
		return RTA_E_NO_DEVICE;
	}
	// this GPU's copy of the static workspace of rta_trace_closest_static (the device is current); a
	// failed look-up only concerns that entry point, which tries again and reports it
	(void)static_scratch_init(device);
	rta_context *ctx = new rta_context();
	ctx->device = device;
	e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	for (int g = 0; g < kMaxGroups - 1 && e == hipSuccess; g++)
		e = hipStreamCreateWithFlags(&ctx->extra[g], hipStreamNonBlocking);
	for (hipEvent_t &ev : ctx->batch_ev)
		if (e == hipSuccess)
			e = hipEventCreate(&ev);
	if (e != hipSuccess ||
	    (e = hipHostMalloc((void **)&ctx->host_counters, sizeof(Counters) * kMaxGroups, hipHostMallocDefault)) != hipSuccess ||
	    (e = ctx->counters.reserve(sizeof(Counters) * kMaxGroups)) != hipSuccess) {
		set_global_error(std::string("context set-up: ") + hipGetErrorString(e));
		rta_destroy(ctx);
		return RTA_E_DEVICE;
	}
	*out = ctx;
	return RTA_OK;
}

void rta_destroy(rta_context *ctx)
{
	if (!ctx)
		return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) {
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipStreamDestroy(ctx->stream);
	}
	for (hipStream_t x : ctx->extra)
		if (x) {
			(void)hipStreamSynchronize(x);
			(void)hipStreamDestroy(x);
		}
	for (hipEvent_t ev : ctx->batch_ev)
		if (ev)
			(void)hipEventDestroy(ev);
	if (ctx->host_counters)
		(void)hipHostFree(ctx->host_counters);
	delete ctx;
}

const char *rta_last_error(const rta_context *ctx) { return ctx ? ctx->error.c_str() : get_global_error(); }

void *rta_stream(rta_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int rta_last_batch_stats(const rta_context *ctx, rta_batch_stats *out)
{
	if (!ctx || !out)
		return RTA_E_INVALID;
	*out = ctx->batch;
	return RTA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ upload

namespace rta {

TreeRef tree_of(const rta_context *ctx)
{
	TreeRef t;
	t.pairs = ctx->pairs.as<const float4>();
	t.prims = ctx->prims.as<const float4>();
	t.root = ctx->root_ref;
	t.tame = ctx->tame ? 1u : 0u;
	t.box_known = ctx->box_known ? 1u : 0u;
	memcpy(t.box_lo, ctx->box, 12);
	memcpy(t.box_hi, ctx->box + 3, 12);
	t.wide = ctx->have_wide ? ctx->wide_nodes.as<const float4>() : nullptr;
	t.leaf_boxes = ctx->leaf_boxes.as<const float4>();
	memcpy(t.wide_center, ctx->wide_center, 12);
	t.wide_reach2 = ctx->wide_reach2;
	t.wide_index_mask = ctx->wide_index_mask;
	t.n_loose = ctx->have_leaf_boxes ? ctx->n_loose : 0u;
	t.loose = ctx->loose.as<const uint32_t>();
	t.flat_prims = 0;
	t.wide_index_bits = (uint32_t)__builtin_popcount(ctx->wide_index_mask);
	// with fewer than 11 bits left for the distance, a float's leading bits say too little
	static const char *relative = getenv("RTA_WIDE_RELATIVE");      // measurement
	t.wide_relative = relative ? (uint32_t)atoi(relative) : (t.wide_index_bits > 20 && ctx->box_known ? 1u : 0u);
	// 16 octaves down from the first power of two beyond the longest way through the scene
	uint32_t top;
	memcpy(&top, &ctx->wide_span, 4);
	top = (top & 0x7f800000u) + (1u << 23);
	t.wide_key_base = top > (16u << 23) ? top - (16u << 23) : 0u;
	return t;
}

}  // namespace rta

namespace {

struct Flattened {
	std::vector<float4> pairs, prims, shade, mats, lights;
	std::vector<float4> ref_nodes;     // the reference's 32-byte node records, for the layout probe only
	uint32_t root = 0;
	int depth = 0;
	bool tame = true;
	bool box_known = false;            // `box`: lo xyz, hi xyz around the root's two children
	float box[6] = {0, 0, 0, 0, 0, 0};
	bool over_limit = false;           // the problem reported is a size limit, not a malformed description
};

inline float as_float(uint32_t u)
{
	float f;
	memcpy(&f, &u, 4);
	return f;
}

// The material and light records (a handful; shared by both upload paths).  The description's
// index ranges have been checked.
void small_records(const rta_scene_desc &d, std::vector<float4> &mats, std::vector<float4> &lights)
{
	mats.resize((size_t)d.n_materials * 2);
	for (int32_t m = 0; m < d.n_materials; m++) {
		const float *s = d.materials + (size_t)m * 6;
		mats[(size_t)m * 2 + 0] = make_float4(s[0], s[1], s[2], s[3]);
		mats[(size_t)m * 2 + 1] = make_float4(s[4], s[5], 0, 0);
	}
	lights.assign(std::max<size_t>((size_t)d.n_lights, 1) * 7, make_float4(0, 0, 0, 0));
	for (int32_t l = 0; l < d.n_lights; l++) {
		int32_t p = d.light_prim[l];
		const float *c = d.light_color + (size_t)l * 3;
		const float *g = d.prim_geom + (size_t)p * 9, *nn = d.prim_normals + (size_t)p * 9;
		float4 *L = &lights[(size_t)l * 7];
		L[0] = make_float4(c[0], c[1], c[2], as_float((uint32_t)p));
		for (int q = 0; q < 3; q++) {
			L[1 + q] = make_float4(g[q * 3], g[q * 3 + 1], g[q * 3 + 2], 0);
			L[4 + q] = make_float4(nn[q * 3], nn[q * 3 + 1], nn[q * 3 + 2], 0);
		}
	}
}

// Checks everything the kernels take for granted and builds their layouts.
// Returns an empty string on success, else what is wrong.
std::string flatten(const rta_scene_desc &d, Flattened &out)
{
	if (d.n_prims <= 0 || d.n_nodes <= 0 || d.n_materials <= 0 || d.n_lights < 0)
		return "scene needs at least one primitive, one node and one material";
	if (!d.prim_type || !d.prim_material || !d.prim_light || !d.prim_geom || !d.prim_normals || !d.node_bounds ||
	    !d.node_index || !d.node_count || !d.materials || (d.n_lights && (!d.light_color || !d.light_prim)))
		return "scene description has null arrays";
	if ((uint32_t)d.n_prims >= kLeaf || (uint32_t)d.n_nodes >= kLeaf)
		return "too many primitives or nodes";

	for (int32_t k = 0; k < d.n_prims; k++) {
		if (d.prim_type[k] != RTA_PRIM_TRIANGLE && d.prim_type[k] != RTA_PRIM_SPHERE)
			return "primitive " + std::to_string(k) + ": unknown type";
		if (d.prim_material[k] < 0 || d.prim_material[k] >= d.n_materials)
			return "primitive " + std::to_string(k) + ": material index out of range";
		if (d.prim_light[k] < -1 || d.prim_light[k] >= d.n_lights)
			return "primitive " + std::to_string(k) + ": light index out of range";
	}
	for (int32_t l = 0; l < d.n_lights; l++) {
		int32_t p = d.light_prim[l];
		if (p < 0 || p >= d.n_prims)
			return "light " + std::to_string(l) + ": primitive index out of range";
		if (d.prim_type[p] != RTA_PRIM_TRIANGLE)
			return "light " + std::to_string(l) + ": only triangles emit (src/parser.cpp:297-300)";
	}

	// Walk the tree from the root: every node reached once, children adjacent,
	// leaves partition [0, n_prims), depth within the traversal stack.
	std::vector<int32_t> pair_of(d.n_nodes, -1);      // inner node -> pair record
	std::vector<uint8_t> covered(d.n_prims, 0);
	std::vector<uint8_t> last_flag(d.n_prims, 0);
	struct Item {
		int32_t node, depth;
	};
	std::vector<Item> todo;
	std::vector<int32_t> inner_order;
	todo.push_back({0, 0});
	int64_t visited = 0;
	int max_depth = 0;
	while (!todo.empty()) {
		Item it = todo.back();
		todo.pop_back();
		if (++visited > d.n_nodes)
			return "node graph is not a tree (a node is reachable twice)";
		max_depth = std::max(max_depth, it.depth);
		int32_t idx = d.node_index[it.node], cnt = d.node_count[it.node];
		if (cnt > 0) {
			if (idx < 0 || (int64_t)idx + cnt > d.n_prims)
				return "leaf " + std::to_string(it.node) + ": primitive range out of bounds";
			for (int32_t k = idx; k < idx + cnt; k++) {
				if (covered[k])
					return "primitive " + std::to_string(k) + " belongs to two leaves";
				covered[k] = 1;
			}
			last_flag[idx + cnt - 1] = 1;
		} else {
			if (cnt < 0)
				return "node " + std::to_string(it.node) + ": negative count";
			if (idx <= 0 || (int64_t)idx + 1 >= d.n_nodes)
				return "node " + std::to_string(it.node) + ": child index out of range";
			if (pair_of[it.node] >= 0)
				return "node graph is not a tree";
			pair_of[it.node] = (int32_t)inner_order.size();
			inner_order.push_back(it.node);
			// second child first so that the first child is numbered next (depth-first, first child first)
			todo.push_back({idx + 1, it.depth + 1});
			todo.push_back({idx, it.depth + 1});
		}
	}
	if (max_depth > kMaxDepth) {
		out.over_limit = true;
		return "tree depth " + std::to_string(max_depth) + " exceeds the traversal stack of " + std::to_string(kMaxDepth) +
		       " entries (the reference's own, src/main.cpp:339)";
	}
	// primitives outside every leaf are unreachable, which is legal for a
	// caller-made tree but would leave their records without a leaf terminator
	out.depth = std::max(max_depth, 1);

	auto ref_of = [&](int32_t node) -> uint32_t {
		return d.node_count[node] > 0 ? (kLeaf | (uint32_t)d.node_index[node]) : (uint32_t)pair_of[node];
	};
	out.root = ref_of(0);
	// the reciprocal slab test needs every box coordinate to be zero or of
	// ordinary magnitude (see ray_is_tame in traverse.hpp); NaN fails both tests
	out.tame = true;
	for (size_t k = 0; k < (size_t)d.n_nodes * 6; k++) {
		const float m = std::fabs(d.node_bounds[k]);
		if (!(m == 0.f || (m >= 0x1p-78f && m <= 0x1p40f)))
			out.tame = false;
	}
	// The reciprocal test takes the smaller quotient of an axis as the entry and
	// the larger as the exit, which equals the reference's selection by the
	// direction's sign bit only while min <= max.  A caller-made tree with an
	// inverted box is traversed with the division-based test throughout.
	for (int32_t k = 0; k < d.n_nodes && out.tame; k++) {
		const float *b = d.node_bounds + (size_t)k * 6;
		if (!(b[0] <= b[3] && b[1] <= b[4] && b[2] <= b[5]))
			out.tame = false;
	}
	// The box around the root's children (the root's own box is never tested,
	// src/main.cpp:332-350, and a caller-made tree need not have a tight one).
	if (out.tame && d.node_count[0] == 0) {
		const float *a = d.node_bounds + (size_t)d.node_index[0] * 6, *b = a + 6;
		for (int k = 0; k < 3; k++) {
			out.box[k] = std::min(a[k], b[k]);
			out.box[3 + k] = std::max(a[3 + k], b[3 + k]);
		}
		out.box_known = true;
	}
	out.pairs.resize(std::max<size_t>(inner_order.size(), 1) * 4, make_float4(0, 0, 0, 0));
	for (size_t p = 0; p < inner_order.size(); p++) {
		int32_t l = d.node_index[inner_order[p]], r = l + 1;
		const float *a = d.node_bounds + (size_t)l * 6, *b = d.node_bounds + (size_t)r * 6;
		out.pairs[p * 4 + 0] = make_float4(a[0], a[1], a[2], a[3]);
		out.pairs[p * 4 + 1] = make_float4(a[4], a[5], b[0], b[1]);
		out.pairs[p * 4 + 2] = make_float4(b[2], b[3], b[4], b[5]);
		out.pairs[p * 4 + 3] = make_float4(as_float(ref_of(l)), as_float(ref_of(r)), 0.f, 0.f);
	}

	out.ref_nodes.resize((size_t)d.n_nodes * 2);
	for (int32_t k = 0; k < d.n_nodes; k++) {
		const float *b = d.node_bounds + (size_t)k * 6;
		out.ref_nodes[(size_t)k * 2] = make_float4(b[0], b[1], b[2], b[3]);
		out.ref_nodes[(size_t)k * 2 + 1] =
		    make_float4(b[4], b[5], as_float((uint32_t)d.node_index[k]), as_float((uint32_t)d.node_count[k]));
	}
	out.prims.assign((size_t)d.n_prims * kPrimQuads, make_float4(0, 0, 0, 0));
	out.shade.resize((size_t)d.n_prims * 3);
	for (int32_t k = 0; k < d.n_prims; k++) {
		const float *g = d.prim_geom + (size_t)k * 9, *nn = d.prim_normals + (size_t)k * 9;
		// an uncovered primitive terminates its own (never visited) leaf
		uint32_t flags = (last_flag[k] || !covered[k]) ? kLastFlag : 0;
		if (d.prim_type[k] == RTA_PRIM_SPHERE) {
			flags |= kSphereFlag;
			out.prims[(size_t)k * kPrimQuads + 0] = make_float4(g[0], g[1], g[2], as_float(flags));
			out.prims[(size_t)k * kPrimQuads + 1] = make_float4(g[3], 0, 0, 0);
			out.prims[(size_t)k * kPrimQuads + 2] = make_float4(0, 0, 0, 0);
		} else {
			// e0 = v0 - v2, e1 = v1 - v2 (src/main.cpp:252-253), rounded once in fp32
			out.prims[(size_t)k * kPrimQuads + 0] = make_float4(g[6], g[7], g[8], as_float(flags));
			out.prims[(size_t)k * kPrimQuads + 1] = make_float4(g[0] - g[6], g[1] - g[7], g[2] - g[8], 0);
			out.prims[(size_t)k * kPrimQuads + 2] = make_float4(g[3] - g[6], g[4] - g[7], g[5] - g[8], 0);
		}
		out.shade[(size_t)k * 3 + 0] = make_float4(nn[0], nn[1], nn[2], as_float((uint32_t)d.prim_material[k]));
		out.shade[(size_t)k * 3 + 1] = make_float4(nn[3], nn[4], nn[5], as_float((uint32_t)d.prim_light[k]));
		out.shade[(size_t)k * 3 + 2] = make_float4(nn[6], nn[7], nn[8], as_float((uint32_t)d.prim_type[k]));
	}
	small_records(d, out.mats, out.lights);
	return "";
}

template <typename T>
int upload_vec(rta_context *ctx, DeviceBuffer &buf, const std::vector<T> &v)
{
	HIP_TRY(ctx, buf.reserve(v.size() * sizeof(T)));
	HIP_TRY(ctx, hipMemcpyAsync(buf.ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
	return RTA_OK;
}


// Owned pixels in walk order: tiles row-major, pixels row-major inside a tile.
// With `padded`, every tile contributes exactly 64 entries and the positions
// that fall outside the frame hold -1; otherwise those are dropped.
void shard_pixels(int32_t w, int32_t h, int32_t shard, int32_t n_shards, bool padded, std::vector<int32_t> &out)
{
	out.clear();
	const int32_t tx = (w + kTile - 1) / kTile, ty = (h + kTile - 1) / kTile;
	for (int32_t t = 0; t < tx * ty; t++) {
		if (t % n_shards != shard)
			continue;
		const int32_t x0 = (t % tx) * kTile, y0 = (t / tx) * kTile;
		for (int32_t y = y0; y < y0 + kTile; y++)
			for (int32_t x = x0; x < x0 + kTile; x++) {
				if (x < w && y < h)
					out.push_back(y * w + x);
				else if (padded)
					out.push_back(-1);
			}
	}
}

}  // namespace

extern "C" {

int rta_validate_scene(const rta_scene_desc *desc, char *err, size_t errlen)
{
	if (!desc)
		return RTA_E_INVALID;
	Flattened f;
	std::string problem = flatten(*desc, f);
	if (problem.empty())
		return RTA_OK;
	set_global_error("invalid scene: " + problem);
	if (err && errlen)
		snprintf(err, errlen, "%s", problem.c_str());
	return f.over_limit ? RTA_E_LIMIT : RTA_E_INVALID;
}

int rta_scene_build_bvh_device(rta_context *ctx, rta_scene *scene, float *device_ms)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (!scene)
		return ctx->fail(RTA_E_INVALID, "null scene");
	if (device_ms)
		*device_ms = 0;
	HostScene &sc = *scene->impl;
	if (sc.prim_count() == 0)
		return ctx->fail(RTA_E_INVALID, "scene has no primitives");
	if (!sc.perm.empty())
		return ctx->fail(RTA_E_INVALID, "BVH already built for this scene");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	try {
		BuiltTree tree;
		bool unsupported = false;
		const std::string err =
		    device_build_bvh(ctx->stream, sc.prim_count(), sc.prim_type.data(), sc.geom.data(), tree, unsupported);
		if (unsupported) {
			sc.build_bvh();
			return RTA_OK;
		}
		if (!err.empty())
			return ctx->fail(RTA_E_DEVICE, "BVH build: " + err);
		sc.adopt_tree(std::move(tree.perm), std::move(tree.node_bounds), std::move(tree.node_index),
		              std::move(tree.node_count));
		if (device_ms)
			*device_ms = tree.device_ms;
		return RTA_OK;
	} catch (const SceneError &e) {
		return ctx->fail(e.code, e.text);
	} catch (const std::exception &e) {
		return ctx->fail(RTA_E_INVALID, e.what());
	}
}

int rta_upload_scene(rta_context *ctx, const rta_scene_desc *desc)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (!desc)
		return ctx->fail(RTA_E_INVALID, "null scene description");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc;
	if (upload_goes_to_device(*desc)) {
		// large scenes: checks, layouts and the wide tree by kernels (upload_device.hip); what that path
		// does not handle, an invalid description included, comes back here
		ctx->have_scene = false;
		ctx->prepared.valid = false;
		rc = upload_scene_on_device(ctx, *desc);
		if (rc < 0)
			return rc;
		if (rc == kUploadDone) {
			std::vector<float4> mats, lights;
			small_records(*desc, mats, lights);
			if ((rc = upload_vec(ctx, ctx->materials, mats)) || (rc = upload_vec(ctx, ctx->lights, lights)))
				return rc;
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the staging vectors die with this call
			ctx->have_soa = false;
			ctx->n_prims = desc->n_prims;
			ctx->n_nodes = desc->n_nodes;
			ctx->n_lights = desc->n_lights;
			ctx->have_scene = true;
			return RTA_OK;
		}
	}
	Flattened f;
	std::string problem = flatten(*desc, f);
	if (!problem.empty())
		return ctx->fail(f.over_limit ? RTA_E_LIMIT : RTA_E_INVALID, "invalid scene: " + problem);
	ctx->have_scene = false;
	ctx->prepared.valid = false;
	if ((rc = upload_vec(ctx, ctx->pairs, f.pairs)) || (rc = upload_vec(ctx, ctx->prims, f.prims)) ||
	    (rc = upload_vec(ctx, ctx->shade_prims, f.shade)) || (rc = upload_vec(ctx, ctx->materials, f.mats)) ||
	    (rc = upload_vec(ctx, ctx->lights, f.lights)) || (rc = upload_vec(ctx, ctx->ref_nodes, f.ref_nodes)))
		return rc;
	// the second tree, for the certified traversal
	ctx->have_wide = false;
	ctx->have_leaf_boxes = false;
	WideTree wide = build_wide_tree(*desc, f.tame);
	ctx->have_flat_boxes = false;
	if (wide.boxes_complete) {
		// what certifies an answer found without the reference's tree: by the walk, from a camera's cell lists, by
		// the scan of a small scene
		if ((rc = upload_vec(ctx, ctx->leaf_boxes, wide.leaf_boxes)))
			return rc;
		ctx->n_loose = wide.certifiable ? (uint32_t)wide.spheres.size() : 0u;
		if (wide.certifiable && !wide.spheres.empty() && (rc = upload_vec(ctx, ctx->loose, wide.spheres)))
			return rc;
	}
	if (wide.enabled) {
		if ((rc = upload_vec(ctx, ctx->wide_nodes, wide.nodes)))
			return rc;
		memcpy(ctx->wide_center, wide.center, 12);
		ctx->wide_reach2 = wide.reach * wide.reach * (1.f - 0x1p-20f);
		ctx->wide_reach = wide.reach;
		ctx->wide_depth = wide.depth;
		ctx->wide_index_mask = wide.index_mask;
		ctx->wide_n_nodes = wide.n_nodes;
		ctx->wide_span = wide.span;
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the staging vectors die with this call
	ctx->have_wide = wide.enabled;
	ctx->have_leaf_boxes = wide.certifiable;
	ctx->have_flat_boxes = wide.boxes_complete;
	if (getenv("RTA_WIDE_VERBOSE"))
		fprintf(stderr, "wide tree: %s%s, %u nodes, depth %d, padding/extent %.4f, reach %g\n", wide.enabled ? "built" : "none: ",
		        wide.why_not.c_str(), wide.n_nodes, wide.depth, wide.mean_pad_over_extent, wide.reach);
	ctx->root_ref = f.root;
	ctx->n_pairs = (uint32_t)(f.pairs.size() / 4);
	ctx->box_known = f.box_known;
	memcpy(ctx->box, f.box, sizeof ctx->box);
	ctx->have_soa = false;
	ctx->tame = f.tame;
	ctx->stack_depth = f.depth;
	ctx->n_prims = desc->n_prims;
	ctx->n_nodes = desc->n_nodes;
	ctx->n_lights = desc->n_lights;
	ctx->have_scene = true;
	return RTA_OK;
}

int64_t rta_shard_pixels(int32_t width, int32_t height, int32_t shard_index, int32_t shard_count, int32_t *out)
{
	if (width <= 0 || height <= 0 || shard_count <= 0 || shard_index < 0 || shard_index >= shard_count)
		return RTA_E_INVALID;
	std::vector<int32_t> px;
	shard_pixels(width, height, shard_index, shard_count, false, px);
	if (out)
		memcpy(out, px.data(), px.size() * sizeof(int32_t));
	return (int64_t)px.size();
}

// ------------------------------------------------------------------ ray batches

// LDS entries per lane of the batch walks.  RTA_BATCH_CAP (measurement) may move it within what the walk
// can work with: below 8 a visit finds no room for its four references, beyond 64 no workgroup fits a CU.
static int batch_walk_cap()
{
	const char *v = getenv("RTA_BATCH_CAP");
	return v ? std::min(std::max(atoi(v), 8), 64) : kBatchWalkCap;
}

// Big batches over a tree that does not fit the caches are traced in the order
// of sort_rays.hip (same results, each ray is traced on its own).  Sets `order`
// to the device array of ray numbers, or to null when the batch is traced as it
// comes: small batches, small trees (the order costs more than it saves) and
// trees without a known box.
// `packets`: the batch is a pinhole camera's - in order, 64 consecutive rays travelling together, all from one
// point - and `order` lists its rays in compact patches of neighbouring directions, for the walk by packets.
static int packets_mode(const rta_context *ctx)
{
	const char *v = getenv("RTA_BATCH_PACKETS");      // measurement and tests: 0 never, 2 every closest-hit batch
	return ctx->have_wide ? (v ? atoi(v) : 1) : 0;
}

static int tracing_order(rta_context *ctx, const float *d_rays, int64_t n, const uint32_t *&order, bool &packets, bool closest)
{
	order = nullptr;
	packets = false;
	const size_t tree_bytes = (size_t)ctx->n_pairs * 64 + (size_t)ctx->n_prims * 16 * kPrimQuads;
	if (!ctx->box_known || n < (1 << 16) || n >= ((int64_t)1 << 32) || tree_bytes < ((size_t)8 << 20))
		return RTA_OK;
	const uint32_t count = (uint32_t)n;
	const size_t temp = (ray_order_temp_bytes(count) + 255) & ~(size_t)255, each = ((size_t)count * 4 + 255) & ~(size_t)255;
	constexpr uint32_t kStride = 64;      // words between two counters: each in a 128-byte line (and L2 channel) of its own
	HIP_TRY(ctx, ctx->ray_order.reserve(4 * each + temp + (size_t)kDescentCounters * kStride * 4));
	char *base = ctx->ray_order.as<char>();
	uint32_t *keys = (uint32_t *)base, *keys_sorted = (uint32_t *)(base + each), *index = (uint32_t *)(base + 2 * each),
	         *sorted = (uint32_t *)(base + 3 * each), *descents = (uint32_t *)(base + 4 * each + temp);
	// (the direction keys go where the origin sort would put its sorted keys: only one of the two sorts runs)
	const int mode = closest ? packets_mode(ctx) : 0;
	launch_ray_keys(ctx->stream, d_rays, count, ctx->box, ctx->box + 3, keys, index, descents, kStride, mode ? keys_sorted : nullptr);
	// a batch that comes in order (a camera's rays, say) is not sorted by its origins
	uint32_t counters[kDescentCounters * kStride];
	HIP_TRY(ctx, hipMemcpyAsync(counters, descents, sizeof counters, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	uint64_t out_of_order = 0, spread_waves = 0, elsewhere_waves = 0;
	for (uint32_t k = 0; k < kDescentCounters; k++) {
		out_of_order += counters[k * kStride];
		spread_waves += counters[k * kStride + 1];
		elsewhere_waves += counters[k * kStride + 2];
	}
	if (out_of_order * 16 < count) {
		// At most one wave in 64 whose rays go apart, and one origin: the rays walk the tree 64 at a time, in
		// the order of their directions (a scan line's 64 pixels are a strip; 64 neighbours in that order a patch:
		// 59 instead of 95 node records per packet on the 10 M-triangle soup).
		if (mode && spread_waves * 64 <= (count + 63) / 64 && elsewhere_waves == 0) {
			HIP_TRY(ctx, launch_ray_sort(ctx->stream, count, keys_sorted, keys, index, sorted, base + 4 * each, temp, true));
			order = sorted;
			packets = true;
		}
		return RTA_OK;
	}
	HIP_TRY(ctx, launch_ray_sort(ctx->stream, count, keys, keys_sorted, index, sorted, base + 4 * each, temp));
	order = sorted;
	return RTA_OK;
}

// > 0: the scene is one that rays scan instead of walking a tree (traverse_flat.hpp) - a few dozen primitives, no wide
// tree, and what certifying an answer needs (a tame tree, its leaves' boxes) - and has this many primitives
static uint32_t flat_prims_of(const rta_context *ctx)
{
	static const char *flat = getenv("RTA_FLAT");      // measurement: 0 never, 1 only scenes without a wide tree, 2 whenever possible
	// (960 x 540 x 64 spp frames of scenes of 12 / 25 / 34 / 50 / 63 primitives that do have a wide tree: 3.33 / 4.82 / 5.38 /
	// 6.28 / 6.02 ms scanned, 3.76 / 5.21 / 5.89 / 6.98 / 6.73 ms walked)
	const int mode = flat ? atoi(flat) : 2;
	const bool scan = mode != 0 && (mode == 2 || !ctx->have_wide) && ctx->n_prims > 0 && ctx->n_prims <= (int32_t)kFlatMaxPrims &&
	                  ctx->have_flat_boxes && ctx->tame;
	return scan ? (uint32_t)ctx->n_prims : 0u;
}

// rta_trace_closest; `walk`: the batch goes through the certified walk first where the scene has a wide tree
// (not in the public header: the probe library asks for the batch without the walk)
int rta_internal_trace_closest(rta_context *ctx, const float *rays, int64_t n, rta_hit *hits, int mem, int walk)
{
	if (!ctx)
		return RTA_E_INVALID;
	const bool count_work = (mem & RTA_TRACE_COUNT_WORK) != 0;
	if (mem & RTA_TRACE_EXACT_ONLY)
		walk = 0;
	mem &= ~(RTA_TRACE_COUNT_WORK | RTA_TRACE_EXACT_ONLY);
	if (!ctx->have_scene)
		return ctx->fail(RTA_E_NO_SCENE, "no scene uploaded");
	if (n < 0 || (n > 0 && (!rays || !hits)) || (mem != RTA_MEM_HOST && mem != RTA_MEM_DEVICE))
		return ctx->fail(RTA_E_INVALID, "bad ray batch arguments");
	if (n == 0)
		return RTA_OK;
	if ((uint64_t)n > (uint64_t)0x7fffffff * 256)
		return ctx->fail(RTA_E_LIMIT, "ray batch too large for one launch");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const float *d_rays = rays;
	rta_hit *d_hits = hits;
	if (mem == RTA_MEM_HOST) {
		HIP_TRY(ctx, ctx->io_a.reserve((size_t)n * 24));
		HIP_TRY(ctx, ctx->io_b.reserve((size_t)n * sizeof(rta_hit)));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.ptr, rays, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
		d_rays = ctx->io_a.as<const float>();
		d_hits = ctx->io_b.as<rta_hit>();
	}
	rta_batch_stats bs;
	memset(&bs, 0, sizeof bs);
	bs.rays = n;
	hipStream_t s = ctx->stream;
	HIP_TRY(ctx, hipEventRecord(ctx->batch_ev[0], s));
	const uint32_t *order = nullptr;
	bool packets = false;
	static const bool exact_only = getenv("RTA_BATCH_EXACT") != nullptr;      // measurement: price the walk
	// (a scene that is scanned: every ray meets every primitive, in whatever order the rays come)
	TreeRef tree = tree_of(ctx);
	tree.flat_prims = walk && n < ((int64_t)1 << 32) && !exact_only ? flat_prims_of(ctx) : 0u;
	if (!tree.flat_prims)
		if (int rc = tracing_order(ctx, d_rays, n, order, packets, true))
			return rc;
	if (packets_mode(ctx) == 2)
		packets = true;           // (forced: whatever order the batch is traced in, 64 of its rays at a time)
	bs.sorted = order != nullptr;
	static const int cap = batch_walk_cap();
	// the work counts of a counting call: group 0's counters, which no render is using meanwhile
	Counters *cnt = count_work ? ctx->counters.as<Counters>() : nullptr;
	auto counted = [&](uint64_t &nodes, uint64_t &prims) -> hipError_t {
		if (!cnt)
			return hipSuccess;
		hipError_t e = hipMemcpyAsync(ctx->host_counters, cnt, sizeof(Counters), hipMemcpyDeviceToHost, s);
		if (e == hipSuccess)
			e = hipStreamSynchronize(s);
		nodes = ctx->host_counters->inner;
		prims = ctx->host_counters->tri + ctx->host_counters->sph;
		return e;
	};
	if (cnt)
		HIP_TRY(ctx, hipMemsetAsync(cnt, 0, sizeof(Counters), s));
	// the list of the rays the walk hands on: 4 bytes per ray; without room for it the batch is traced as before
	const size_t list_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
	if (walk && (ctx->have_wide || tree.flat_prims) && n < ((int64_t)1 << 32) && !exact_only && ctx->doubts.reserve(list_bytes + 256) != hipSuccess) {
		(void)hipGetLastError();
		walk = false;
		tree.flat_prims = 0;
		if (int rc = tracing_order(ctx, d_rays, n, order, packets, true))
			return rc;
	}
	HIP_TRY(ctx, hipEventRecord(ctx->batch_ev[1], s));
	if (walk && (ctx->have_wide || tree.flat_prims) && n < ((int64_t)1 << 32) && !exact_only) {
		// the certified walk (or the scan of a small scene) answers what it can prove; the rest goes through the
		// replaying traversal
		uint32_t *list = ctx->doubts.as<uint32_t>(), *count = (uint32_t *)(ctx->doubts.as<char>() + list_bytes);
		if (tree.flat_prims)
			launch_batch_flat(s, tree, d_rays, (uint64_t)n, d_hits, list, count, cnt);
		else if (packets) {
			launch_batch_packet(s, tree_of(ctx), d_rays, (uint64_t)n, d_hits, order, list, count, cnt);
			bs.packets = 1;
		} else
			launch_batch_walk(s, tree_of(ctx), d_rays, (uint64_t)n, d_hits, cap, order, list, count, cnt);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipEventRecord(ctx->batch_ev[2], s));
		uint32_t in_doubt = 0;
		HIP_TRY(ctx, hipMemcpyAsync(&in_doubt, count, sizeof in_doubt, hipMemcpyDeviceToHost, s));
		HIP_TRY(ctx, hipStreamSynchronize(s));
		HIP_TRY(ctx, counted(bs.walk_node_visits, bs.walk_tri_tests));
		if (cnt)
			HIP_TRY(ctx, hipMemsetAsync(cnt, 0, sizeof(Counters), s));
		bs.walked = tree.flat_prims ? 0 : 1;
		bs.scanned = tree.flat_prims ? 1 : 0;
		bs.retraced_rays = in_doubt;
		if (in_doubt)
			launch_batch_closest(s, tree_of(ctx), d_rays, (uint64_t)in_doubt, d_hits, ctx->stack_depth, cnt, list);
	} else {
		HIP_TRY(ctx, hipEventRecord(ctx->batch_ev[2], s));
		launch_batch_closest(s, tree_of(ctx), d_rays, (uint64_t)n, d_hits, ctx->stack_depth, cnt, order);
	}
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipEventRecord(ctx->batch_ev[3], s));
	if (mem == RTA_MEM_HOST)
		HIP_TRY(ctx, hipMemcpyAsync(hits, d_hits, (size_t)n * sizeof(rta_hit), hipMemcpyDeviceToHost, s));
	HIP_TRY(ctx, hipStreamSynchronize(s));
	HIP_TRY(ctx, counted(bs.replay_node_visits, bs.replay_prim_tests));
	float ms = 0;
	HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->batch_ev[0], ctx->batch_ev[1]));
	bs.order_ms = ms;
	HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->batch_ev[1], ctx->batch_ev[2]));
	bs.walk_ms = ms;
	HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->batch_ev[2], ctx->batch_ev[3]));
	bs.replay_ms = ms;
	ctx->batch = bs;
	return RTA_OK;
}

int rta_trace_closest(rta_context *ctx, const float *rays, int64_t n, rta_hit *hits, int mem)
{
	return rta_internal_trace_closest(ctx, rays, n, hits, mem, 1);
}

int rta_trace_occluded(rta_context *ctx, const float *rays, const float *tmax, int64_t n, uint8_t *out, int mem)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (!ctx->have_scene)
		return ctx->fail(RTA_E_NO_SCENE, "no scene uploaded");
	if (n < 0 || (n > 0 && (!rays || !tmax || !out)))
		return ctx->fail(RTA_E_INVALID, "bad ray batch arguments");
	if (n == 0)
		return RTA_OK;
	if ((uint64_t)n > (uint64_t)0x7fffffff * 256)
		return ctx->fail(RTA_E_LIMIT, "ray batch too large for one launch");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const float *d_rays = rays, *d_tmax = tmax;
	uint8_t *d_out = out;
	if (mem == RTA_MEM_HOST) {
		HIP_TRY(ctx, ctx->io_a.reserve((size_t)n * 24));
		HIP_TRY(ctx, ctx->io_b.reserve((size_t)n * 4));
		HIP_TRY(ctx, ctx->io_c.reserve((size_t)n));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.ptr, rays, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->io_b.ptr, tmax, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
		d_rays = ctx->io_a.as<const float>();
		d_tmax = ctx->io_b.as<const float>();
		d_out = ctx->io_c.as<uint8_t>();
	}
	const uint32_t *order = nullptr;
	bool packets = false;
	static const bool exact_only = getenv("RTA_BATCH_EXACT") != nullptr;      // measurement: price the walk
	TreeRef tree = tree_of(ctx);
	tree.flat_prims = exact_only ? 0u : flat_prims_of(ctx);
	if (!tree.flat_prims)
		if (int rc = tracing_order(ctx, d_rays, n, order, packets, false))
			return rc;
	if (tree.flat_prims)
		launch_batch_occluded_flat(ctx->stream, tree, d_rays, d_tmax, (uint64_t)n, d_out, ctx->stack_depth);
	else if (ctx->have_wide && n < ((int64_t)1 << 32) && !exact_only)
		launch_batch_occluded_walk(ctx->stream, tree_of(ctx), d_rays, d_tmax, (uint64_t)n, d_out, kBatchWalkCap, ctx->stack_depth, order);
	else
		launch_batch_occluded(ctx->stream, tree_of(ctx), d_rays, d_tmax, (uint64_t)n, d_out, ctx->stack_depth, order);
	HIP_TRY(ctx, hipGetLastError());
	if (mem == RTA_MEM_HOST)
		HIP_TRY(ctx, hipMemcpyAsync(out, d_out, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return RTA_OK;
}

// ------------------------------------------------------------------ render

int rta_resolve(rta_context *ctx, const float *accum, int32_t width, int32_t height, int32_t samples, float exposure,
                uint8_t *bgr, int mem)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (!accum || !bgr || width <= 0 || height <= 0)
		return ctx->fail(RTA_E_INVALID, "bad resolve arguments");
	const size_t n = (size_t)width * height;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const float *d_acc = accum;
	uint8_t *d_bgr = bgr;
	if (mem == RTA_MEM_HOST) {
		HIP_TRY(ctx, ctx->io_a.reserve(n * 12));
		HIP_TRY(ctx, ctx->io_c.reserve(n * 3));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.ptr, accum, n * 12, hipMemcpyHostToDevice, ctx->stream));
		d_acc = ctx->io_a.as<const float>();
		d_bgr = ctx->io_c.as<uint8_t>();
	}
	launch_resolve(ctx->stream, d_acc, nullptr, (uint32_t)n, (float)samples, exposure, d_bgr);
	HIP_TRY(ctx, hipGetLastError());
	if (mem == RTA_MEM_HOST)
		HIP_TRY(ctx, hipMemcpyAsync(bgr, d_bgr, n * 3, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return RTA_OK;
}

// Can samples be decided on the image plane (see k_sky_map)?  Only for a camera
// (pinhole, or a lens parallel to the image plane) that sees all of the scene's
// box well in front of it, with perpendicular image axes.  The margin by which
// the projected boxes are widened is 0.05 pixels + 1e-5 radians: orders of
// magnitude more than the binary32 roundings of a sample's point on the image
// plane, of its ray and of the reference's slab test can move either.
static bool projection_of(const rta_camera &cam, const TreeRef &tree, Projection &pr)
{
	if (!tree.box_known)
		return false;
	for (int k = 0; k < 3; k++) {
		pr.o[k] = cam.origin[k];
		pr.ul[k] = cam.upper_left[k];
		pr.du[k] = cam.du[k];
		pr.dv[k] = cam.dv[k];
	}
	pr.n[0] = pr.du[1] * pr.dv[2] - pr.du[2] * pr.dv[1];
	pr.n[1] = pr.du[2] * pr.dv[0] - pr.du[0] * pr.dv[2];
	pr.n[2] = pr.du[0] * pr.dv[1] - pr.du[1] * pr.dv[0];
	double dudv = 0, n2 = 0;
	pr.h = pr.du2 = pr.dv2 = 0;
	for (int k = 0; k < 3; k++) {
		pr.h += (pr.ul[k] - pr.o[k]) * pr.n[k];
		pr.du2 += pr.du[k] * pr.du[k];
		pr.dv2 += pr.dv[k] * pr.dv[k];
		dudv += pr.du[k] * pr.dv[k];
		n2 += pr.n[k] * pr.n[k];
	}
	if (!(pr.du2 > 0) || !(pr.dv2 > 0) || !(fabs(dudv) <= 1e-6 * sqrt(pr.du2 * pr.dv2)) || !(fabs(pr.h) > 0) || !(n2 > 0))
		return false;
	for (int c = 0; c < 8; c++) {
		double depth = 0;
		for (int k = 0; k < 3; k++)
			depth += ((double)((c >> k) & 1 ? tree.box_hi[k] : tree.box_lo[k]) - pr.o[k]) * pr.n[k];
		const double scale = pr.h / depth;
		if (!(scale > 0) || !(scale < 1e6))
			return false;
	}
	// Lens cameras (src/main.cpp:665-669): origins o + a lens_u + b lens_v, a and b in [-1, 1).  The
	// lens must lie in a plane parallel to the image plane (it does for rta_camera_from_settings).
	pr.lens_x = pr.lens_y = 0;
	if (cam.defocus_angle > 0) {
		double lu_n = 0, lv_n = 0, lu_du = 0, lv_du = 0, lu_dv = 0, lv_dv = 0, lu2 = 0, lv2 = 0;
		for (int k = 0; k < 3; k++) {
			lu_n += cam.lens_u[k] * pr.n[k], lv_n += cam.lens_v[k] * pr.n[k];
			lu_du += cam.lens_u[k] * pr.du[k], lv_du += cam.lens_v[k] * pr.du[k];
			lu_dv += cam.lens_u[k] * pr.dv[k], lv_dv += cam.lens_v[k] * pr.dv[k];
			lu2 += (double)cam.lens_u[k] * cam.lens_u[k], lv2 += (double)cam.lens_v[k] * cam.lens_v[k];
		}
		if (!(fabs(lu_n) <= 1e-6 * sqrt(lu2 * n2)) || !(fabs(lv_n) <= 1e-6 * sqrt(lv2 * n2)))
			return false;
		pr.lens_x = (fabs(lu_du) + fabs(lv_du)) / pr.du2;
		pr.lens_y = (fabs(lu_dv) + fabs(lv_dv)) / pr.dv2;
		if (!std::isfinite(pr.lens_x) || !std::isfinite(pr.lens_y))
			return false;
	}
	// one pixel is sqrt(min(du2, dv2)) wide on a plane |h| / |n| away
	const double pixel_angle = sqrt(std::min(pr.du2, pr.dv2)) / (fabs(pr.h) / sqrt(n2));
	pr.margin = 0.05 + 1e-5 / pixel_angle;
	return std::isfinite(pr.margin) && pr.margin < 1.0;
}


int rta_render(rta_context *ctx, const rta_camera *camera, const rta_render_params *p, float *accum, uint8_t *bgr,
               rta_stats *stats)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (!ctx->have_scene)
		return ctx->fail(RTA_E_NO_SCENE, "no scene uploaded");
	if (!camera || !p)
		return ctx->fail(RTA_E_INVALID, "null camera or parameters");
	if (p->width <= 0 || p->height <= 0 || p->samples <= 0)
		return ctx->fail(RTA_E_INVALID, "width, height and samples must be positive");
	const int32_t n_shards = p->shard_count <= 0 ? 1 : p->shard_count;
	if (p->shard_index < 0 || p->shard_index >= n_shards)
		return ctx->fail(RTA_E_INVALID, "shard_index out of range");
	if ((int64_t)p->width * p->height > 0x7fffffff / 3)
		return ctx->fail(RTA_E_LIMIT, "frame too large");
	HIP_TRY(ctx, hipSetDevice(ctx->device));

	std::vector<int32_t> px;          // 64 entries per owned tile, -1 outside the frame
	shard_pixels(p->width, p->height, p->shard_index, n_shards, true, px);
	const uint32_t n_pad = (uint32_t)px.size();
	const uint32_t n_tiles = n_pad / 64;
	uint32_t n_owned = 0;
	for (int32_t q : px)
		n_owned += q >= 0;
	const uint32_t spp = (uint32_t)p->samples;
	const uint32_t sps = p->samples_per_stream <= 0 ? spp : std::min((uint32_t)p->samples_per_stream, spp);
	// the sample blocks this call renders: all of them, or the caller's range of them (sample-range sharding)
	const uint32_t all_blocks = (spp + sps - 1) / sps;
	const bool whole = p->block_begin == 0 && p->block_end == 0;
	if (!whole && (p->block_begin < 0 || p->block_end <= p->block_begin || (uint32_t)p->block_end > all_blocks))
		return ctx->fail(RTA_E_INVALID, "block range out of order or beyond the " + std::to_string(all_blocks) + " sample blocks");
	const uint32_t range_begin = whole ? 0u : (uint32_t)p->block_begin, range_end = whole ? all_blocks : (uint32_t)p->block_end;
	const bool partial_sum = range_begin != 0 || range_end != all_blocks;
	if (partial_sum && bgr)
		return ctx->fail(RTA_E_INVALID, "a render of part of the sample blocks yields partial sums: pass bgr = NULL and resolve the summed frame");
	const uint32_t n_blocks = range_end - range_begin;
	const uint32_t run_blocks = p->run_blocks > 0 ? std::min((uint32_t)p->run_blocks, n_blocks) : 1u;
	const uint32_t n_runs = (n_blocks + run_blocks - 1) / run_blocks;
	// A wave's 64 lanes work on 64 / lane_blocks pixels of a tile (whole rows of it)
	// x lane_blocks block runs: lanes on the same pixel cost the same, so the
	// fewer pixels a wave spans the less its lanes wait for each other where the
	// image changes (dragon's silhouette).  Needs enough runs to go round.
	uint32_t lane_blocks = 1;
	while (lane_blocks < 8 && lane_blocks * 2 <= n_runs)
		lane_blocks *= 2;
	const uint32_t n_groups = (n_runs + lane_blocks - 1) / lane_blocks;
	const uint64_t items64 = (uint64_t)n_tiles * lane_blocks * n_groups;      // one per wave at a time
	const size_t frame_px = (size_t)p->width * p->height;
	rta_stats st;
	memset(&st, 0, sizeof st);

	// The slot pool: `groups` equal parts of whole workgroups, never more waves
	// than there are (tile, block run) pairs to hand out.
	uint64_t want_waves = ((p->slots > 0 ? (uint64_t)p->slots : (1u << 24)) + 63) / 64;
	want_waves = std::min<uint64_t>(want_waves, std::max<uint64_t>(items64, 1));
	int groups = p->streams > 0 ? std::min(p->streams, kMaxGroups) : 4;
	const uint64_t waves_per_block = 256 / 64;
	// (the workspace is sized for this many; how many of them the render uses is settled once the
	// image's work is known, see below)
	const uint64_t max_group_blocks = (want_waves + groups * waves_per_block - 1) / (groups * waves_per_block);
	if (items64 + max_group_blocks * 256 * groups >= 0xffffffffull || (uint64_t)n_blocks * n_pad >= 0xffffffffull)
		return ctx->fail(RTA_E_LIMIT, "tiles x sample blocks exceeds 2^32; raise samples_per_stream");
	// A wave draws new streams when all its lanes are waiting.  Drawing earlier
	// (measured at 4..32 waiting lanes) fills idle lanes but loses more than it
	// gains: the wave's camera rays stop coming from one part of the image at one
	// sample index.
	static const char *grab_env = getenv("RTA_GRAB_AT");      // measurement
	const uint32_t grab_at = grab_env ? (uint32_t)std::min(std::max(atoi(grab_env), 1), 64) : 64u;

	// workspace (grow-only, reused across calls)
	const size_t S = (size_t)max_group_blocks * 256 * groups;
	HIP_TRY(ctx, ctx->ray0.reserve(S * 16));
	HIP_TRY(ctx, ctx->ray1.reserve(S * 16));
	HIP_TRY(ctx, ctx->hit.reserve(S * 16));
	HIP_TRY(ctx, ctx->thr.reserve(S * 16));
	HIP_TRY(ctx, ctx->col.reserve(S * 16));
	HIP_TRY(ctx, ctx->rng.reserve(S * 4));
	HIP_TRY(ctx, ctx->acc.reserve(S * 16));
	HIP_TRY(ctx, ctx->meta.reserve(S * 16));
	HIP_TRY(ctx, ctx->hint.reserve(S * 4));
	// the ray queues, for the largest pool this call may use (their layout follows the pool, below)
	{
		const size_t max_cap = (size_t)((max_group_blocks + kShards - 1) / kShards) * 256;
		HIP_TRY(ctx, ctx->q_ray.reserve((size_t)2 * kRayKinds * kShards * max_cap * groups * 4));
		HIP_TRY(ctx, ctx->q_bounce.reserve((size_t)2 * kShards * max_cap * 2 * groups * 16));
		HIP_TRY(ctx, ctx->q_camera.reserve((size_t)2 * kShards * max_cap * groups * 16));
		if (ctx->n_lights > 0)
			HIP_TRY(ctx, ctx->shadow.reserve((size_t)groups * kShards * max_cap * 48));      // kShards sub-queues per group
	}
	// Block sums are kept until their pixel's earlier blocks are in, 16 bytes per
	// (pixel, block).  Beyond the budget the samples are rendered in passes over
	// consecutive groups of blocks, each pass continuing the pixels' sums.
	const uint64_t budget = (p->flags & RTA_RENDER_SMALL_PASSES) ? 0 : (uint64_t)32 << 30;
	const uint32_t blocks_per_group = lane_blocks * run_blocks;
	const uint64_t bytes_per_group = (uint64_t)blocks_per_group * std::max<uint32_t>(n_pad, 1) * 16;
	const uint32_t groups_per_pass = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_groups, budget / bytes_per_group));
	const uint32_t pass_blocks = std::min<uint64_t>(n_blocks, (uint64_t)groups_per_pass * blocks_per_group);
	HIP_TRY(ctx, ctx->partial.reserve(std::max<size_t>((size_t)pass_blocks * n_pad, 1) * 16));
	HIP_TRY(ctx, ctx->pixels.reserve(std::max<size_t>(n_pad, 1) * 4));
	const uint32_t n_parts = n_tiles * lane_blocks;
	HIP_TRY(ctx, ctx->part_weights.reserve(std::max<size_t>(n_parts, 1) * 4));
	HIP_TRY(ctx, ctx->part_order.reserve(std::max<size_t>(n_parts, 1) * 4));

	const bool to_device = p->out_mem == RTA_MEM_DEVICE;
	float *d_frame = nullptr;
	uint8_t *d_bgr = nullptr;
	if (to_device) {
		d_frame = accum;
		d_bgr = bgr;
		if (!d_frame && bgr) {
			HIP_TRY(ctx, ctx->frame.reserve(frame_px * 12));
			d_frame = ctx->frame.as<float>();
		}
	} else {
		HIP_TRY(ctx, ctx->frame.reserve(frame_px * 12));
		d_frame = ctx->frame.as<float>();
		if (bgr) {
			HIP_TRY(ctx, ctx->bgr.reserve(frame_px * 3));
			d_bgr = ctx->bgr.as<uint8_t>();
		}
	}

	RenderArgs base;
	memset(&base, 0, sizeof base);
	base.tree = tree_of(ctx);
	base.shade_prims = ctx->shade_prims.as<const float4>();
	base.materials = ctx->materials.as<const float4>();
	base.lights = ctx->lights.as<const float4>();
	base.n_lights = ctx->n_lights;
	base.partial = ctx->partial.as<float4>();
	base.pixels = ctx->pixels.as<const int32_t>();
	base.n_tiles = n_tiles;
	base.n_pad = n_pad;
	base.n_blocks = n_blocks;
	base.run_blocks = run_blocks;
	base.lane_blocks = lane_blocks;
	base.part_order = ctx->part_order.as<const uint32_t>();
	// per pass of k_advance a lane may finish so many sample blocks of pixels that only see the sky,
	// and so many single samples that need no ray (measured on dragon.txt: 2-4 best, 64 loses 2 %:
	// the wave's other lanes wait with their rays meanwhile)
	base.bulk_trips = 64;
	base.fast_trips = 4;
	base.spp = spp;
	base.sps = sps;
	base.width = (uint32_t)p->width;
	base.seed = p->seed;
	base.max_bounces = p->max_bounces > 0 ? std::min(p->max_bounces, 1 << 20) : 10;      // (the slot state numbers bounces in 24 bits)
	memcpy(base.sky, p->sky, sizeof base.sky);
	base.cam = *camera;
	Projection projection;
	const bool with_sky_map = n_owned > 0 && projection_of(*camera, base.tree, projection);
	if (with_sky_map) {
		HIP_TRY(ctx, ctx->cover.reserve(((size_t)p->width + 2 * kSkyBorder) * ((size_t)p->height + 2 * kSkyBorder)));
		HIP_TRY(ctx, ctx->sky_map.reserve(frame_px * 4));
		base.sky_map = ctx->sky_map.as<const float>();
	}
	// Bounce rays take the certified walk over the wide tree where the scene has one (triangles
	// only, short leaves); 20 LDS entries per lane keep 8 workgroups on a CU
	{
		static const char *sort_lanes = getenv("RTA_SORT_LANES");      // measurement
		base.sort_lanes = sort_lanes ? (uint32_t)atoi(sort_lanes) : (ctx->n_lights > 0 ? 1u : 0u);
	}
	base.wide_cap = (ctx->have_wide && !(p->flags & RTA_RENDER_EXACT_ONLY)) ? 20u : 0u;
	// Without the walk, long leaves (4 primitives and more on average) are traversed as nested loops: see
	// traverse_leafwise.  With it they do not matter: the wide tree has a topology of its own, and only the few
	// rays the walk hands on meet the reference's leaves - which is what makes a mesh standing on a floor of
	// two large triangles renderable at all (the reference's midpoint split leaves half the mesh in one leaf
	// with a floor triangle whose box spans the scene: 13 s per 1080p x 256 spp frame replayed, 0.2 s walked).
	base.leafwise = !base.wide_cap && (uint64_t)ctx->n_prims >= 4 * ((uint64_t)ctx->n_pairs + 1) ? 1u : 0u;
	// A scene of a few dozen primitives needs no tree: every ray tests them all, in lock-step and mostly without
	// divisions (traverse_flat.hpp); answers are certified against the leaves' boxes as the walk's are.  (The queues
	// are those of the leafwise kernels.)
	base.tree.flat_prims = (p->flags & RTA_RENDER_EXACT_ONLY) ? 0u : flat_prims_of(ctx);
	if (base.tree.flat_prims) {
		base.leafwise = 1u;
		base.wide_cap = 0u;
	}
	// Camera rays join them when the camera - all of its lens (src/main.cpp:665-669), if it has one - lies within the
	// wide tree's reach (the walk looks at each ray's origin again: this only keeps rays that it would all hand on
	// out of it); each starts with the triangle that the slot's last camera ray hit: the same pixel, most often the
	// same triangle
	{
		double d2 = 0, lu2 = 0, lv2 = 0;
		for (int k = 0; k < 3; k++) {
			d2 += ((double)camera->origin[k] - ctx->wide_center[k]) * ((double)camera->origin[k] - ctx->wide_center[k]);
			lu2 += (double)camera->lens_u[k] * camera->lens_u[k], lv2 += (double)camera->lens_v[k] * camera->lens_v[k];
		}
		const double farthest = sqrt(d2) + (camera->defocus_angle > 0.f ? sqrt(lu2) + sqrt(lv2) : 0.0);
		static const char *camera_walk = getenv("RTA_CAMERA_WALK"), *hints = getenv("RTA_HINTS");      // measurement
		base.camera_walk = base.wide_cap && farthest * farthest < 0.99 * (double)ctx->wide_reach * ctx->wide_reach &&
		                           (camera_walk ? atoi(camera_walk) != 0 : true)
		                       ? 1u
		                       : 0u;
		base.hint = base.camera_walk && (hints ? atoi(hints) != 0 : true) ? ctx->hint.as<int32_t>() : nullptr;
	}
	// ... or are answered without any tree, from the triangle lists of their cells on the image plane (cells.hip),
	// where samples can be placed on the image plane at all (the conditions of the sky map) and an answer can be
	// certified (the leaf boxes of the reference's tree are nested); with or without a wide tree
	const int map_w = p->width + 2 * kSkyBorder, map_h = p->height + 2 * kSkyBorder;
	const size_t n_cells = (size_t)map_w * (size_t)map_h;
	bool want_cells = false;
	{
		static const char *cells = getenv("RTA_CAMERA_CELLS");      // measurement
		want_cells = !(camera->defocus_angle > 0.f) && with_sky_map && ctx->have_leaf_boxes && ctx->tame && !(p->flags & RTA_RENDER_EXACT_ONLY) &&
		             n_cells < ((size_t)1 << 30) && (cells ? atoi(cells) != 0 : true);
	}
	// the walk's thresholds: idle lanes that make a wave draw new rays, lanes with a listed triangle (lanes with
	// nothing else to do) that make it run the triangle test.  Measured on dragon.txt, camera and bounce rays:
	// refill at 8 / 16 / 32 loses 4 / 1 / 0 %, quorum 16 or 32 loses 1 %, waiting 8 or 32 within 0.5 %.
	{
		static const char *refill = getenv("RTA_WALK_REFILL"), *quorum = getenv("RTA_WALK_QUORUM"), *waiting = getenv("RTA_WALK_WAITING");
		// (a threshold beyond the 64 lanes of a wave would never be met: the walk's loop would spin)
		auto lanes = [](const char *v, uint32_t fallback) { return v ? (uint32_t)std::min(std::max(atoi(v), 1), 64) : fallback; };
		base.walk_refill_at = lanes(refill, 24u);
		base.walk_quorum = lanes(quorum, 24u);
		base.walk_waiting = lanes(waiting, 16u);
	}
	base.rays_per_lane = 8;       // measured on dragon.txt: 4 and 12 lose 1-2 %
	{
		static const char *spread = getenv("RTA_SPREAD");      // measurement
		base.spread_blocks = spread ? (uint32_t)std::min(std::max(atoi(spread), 1), 1024) : 8u;
	}
	base.refill_at = 24;          // likewise 16 (8 loses 3 %)

	const bool time_kernels = (p->flags & RTA_RENDER_TIME_KERNELS) != 0;
	const bool count_work = (p->flags & RTA_RENDER_COUNT_WORK) != 0;
	const bool lights = ctx->n_lights > 0;
	hipStream_t s = ctx->stream;

	hipEvent_t ev_begin = nullptr, ev_mid = nullptr, ev_end = nullptr;
	std::vector<hipEvent_t> ev_pool;        // begin/end pairs around timed launches
	std::vector<int> ev_kind;               // 0 intersect, 1 shadow, 2 advance, 3 camera cells, 4 the rays left over
	auto cleanup = [&]() {
		for (hipEvent_t e : {ev_begin, ev_mid, ev_end})
			if (e)
				(void)hipEventDestroy(e);
		for (hipEvent_t e : ev_pool)
			(void)hipEventDestroy(e);
	};
#define HIP_TRY_R(call)                                      \
	do {                                                     \
		hipError_t e__ = (call);                             \
		if (e__ != hipSuccess) {                             \
			cleanup();                                       \
			return ctx->hip_fail(e__, #call);                \
		}                                                    \
	} while (0)

	HIP_TRY_R(hipEventCreate(&ev_begin));
	HIP_TRY_R(hipEventCreate(&ev_mid));
	HIP_TRY_R(hipEventCreate(&ev_end));
	// What is prepared here depends only on scene, camera, frame and shard; a render with the
	// same key as the last one finds it all where that render left it (rta_context::Prepared).
	rta_context::Prepared &pre = ctx->prepared;
	const bool reuse = pre.valid && memcmp(&pre.cam, camera, sizeof(rta_camera)) == 0 && pre.width == p->width &&
	                   pre.height == p->height && pre.shard_index == p->shard_index && pre.shard_count == n_shards &&
	                   pre.lane_blocks == lane_blocks && pre.with_sky_map == with_sky_map && pre.want_cells == want_cells;
	pre.valid = false;
	if (n_pad && !reuse)
		HIP_TRY_R(hipMemcpyAsync(ctx->pixels.ptr, px.data(), (size_t)n_pad * 4, hipMemcpyHostToDevice, s));
	HIP_TRY_R(hipEventRecord(ev_begin, s));
	if (with_sky_map && !reuse)
		launch_sky_map(s, base.tree, ctx->n_pairs, projection, (uint32_t)p->width, (uint32_t)p->height,
		               ctx->cover.as<uint8_t>(), ctx->sky_map.as<float>());
	bool with_cells = reuse ? pre.with_cells : false;
	if (want_cells && !reuse) {
		const size_t temp_bytes = cell_scan_temp_bytes(n_cells + 1);
		HIP_TRY_R(ctx->cell_counts.reserve((n_cells + 1) * 4));
		HIP_TRY_R(ctx->cell_starts.reserve((n_cells + 1) * 4));
		HIP_TRY_R(ctx->cell_temp.reserve(std::max<size_t>(temp_bytes, 16)));
		HIP_TRY_R(launch_cell_count(s, base.tree, ctx->wide_n_nodes, (uint32_t)ctx->n_prims, projection, map_w, map_h,
		                            ctx->cell_counts.as<uint32_t>(), ctx->cell_starts.as<uint32_t>(), ctx->cell_temp.ptr, temp_bytes));
		uint32_t entries = 0;
		HIP_TRY_R(hipMemcpyAsync(&entries, ctx->cell_starts.as<uint32_t>() + n_cells, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY_R(hipStreamSynchronize(s));
		// (lists of a scene whose triangles all cover the frame would be of no use)
		if (entries <= (1u << 26)) {
			const size_t room = std::max<size_t>(entries, 1), sort_bytes = cell_sort_temp_bytes(room);
			HIP_TRY_R(ctx->cell_list.reserve(room * 4));
			HIP_TRY_R(ctx->cell_keys.reserve(room * 8));
			HIP_TRY_R(ctx->cell_unsorted.reserve(room * 12));      // keys, then triangle numbers
			HIP_TRY_R(ctx->cell_temp.reserve(std::max<size_t>(sort_bytes, 16)));
			HIP_TRY_R(launch_cell_fill(s, base.tree, ctx->wide_n_nodes, (uint32_t)ctx->n_prims, projection, map_w, map_h,
			                           ctx->cell_counts.as<uint32_t>(), ctx->cell_starts.as<uint32_t>(), entries,
			                           reinterpret_cast<uint32_t *>(ctx->cell_unsorted.as<unsigned long long>() + room),
			                           ctx->cell_unsorted.as<unsigned long long>(), ctx->cell_list.as<uint32_t>(),
			                           ctx->cell_keys.as<unsigned long long>(), ctx->cell_temp.ptr, sort_bytes));
			with_cells = true;
		}
		pre.cell_entries = entries;
	}
	if (with_cells) {
		base.camera_cells = 1;
		base.hint = nullptr;          // (hints are the walk's)
		static const char *first_hit = getenv("RTA_SHADE_FIRST_HIT");      // measurement
		base.shade_first_hit = (first_hit ? atoi(first_hit) != 0 : true) ? 1u : 0u;
		base.cell_start = ctx->cell_starts.as<const uint32_t>();
		base.cell_list = ctx->cell_list.as<const uint32_t>();
		base.cell_keys = ctx->cell_keys.as<const unsigned long long>();
		base.cell_w = map_w;
		base.cell_h = map_h;
	}
	// Heaviest parts of the image first: one ray per pixel tells which parts see
	// geometry, a counting sort (stable, so equal parts keep their order) makes the
	// list the waves draw their items from.
	uint32_t n_listed = 0, n_far = 0;
	uint64_t far_pixels = 0;
	if (reuse) {
		n_listed = pre.n_listed;
		n_far = pre.n_far;
		far_pixels = pre.far_pixels;
	} else if (n_parts) {
		std::vector<uint32_t> weight(n_parts), order(n_parts);
		launch_part_weights(s, base, ctx->part_weights.as<uint32_t>(), ctx->stack_depth);
		HIP_TRY_R(hipMemcpyAsync(weight.data(), ctx->part_weights.ptr, (size_t)n_parts * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY_R(hipStreamSynchronize(s));
		// parts that only see the sky come off the list: k_sum_blocks knows their block sums
		// without anybody writing them
		std::vector<uint32_t> far;
		uint32_t start[66] = {0};
		for (uint32_t w : weight)
			if (w != kNothingToTrace)
				start[64 - std::min(w, 64u) + 1]++;
		for (int k = 1; k < 66; k++)
			start[k] += start[k - 1];
		for (uint32_t k = 0; k < n_parts; k++)
			if (weight[k] != kNothingToTrace)
				order[start[64 - std::min(weight[k], 64u)]++] = k;
			else
				far.push_back(k);
		n_far = (uint32_t)far.size();
		n_listed = n_parts - n_far;
		const uint32_t part_pixels = 64 / lane_blocks;
		for (uint32_t which : far)
			for (uint32_t k = 0; k < part_pixels; k++)
				far_pixels += px[(size_t)(which / lane_blocks) * 64 + (which % lane_blocks) * part_pixels + k] >= 0;
		HIP_TRY_R(hipMemcpyAsync(ctx->part_order.ptr, order.data(), (size_t)n_listed * 4, hipMemcpyHostToDevice, s));
		HIP_TRY_R(hipStreamSynchronize(s));       // `order` and `far` die here
	}
	pre.cam = *camera;
	pre.width = p->width;
	pre.height = p->height;
	pre.shard_index = p->shard_index;
	pre.shard_count = n_shards;
	pre.lane_blocks = lane_blocks;
	pre.with_sky_map = with_sky_map;
	pre.want_cells = want_cells;
	pre.with_cells = with_cells;
	pre.n_listed = n_listed;
	pre.n_far = n_far;
	pre.far_pixels = far_pixels;
	pre.valid = true;
	// The slot pool.  Every pass of k_advance reads a word of every slot, whatever is left to do in
	// it, and a render lasts at least as many wavefront steps as its longest chain of path segments,
	// so a small job (a rank's share of a frame split over 8 GPUs) is fastest with a pool of about one
	// slot per stream that needs rays (dragon 1080p x 1024 spp in streams of 8 samples: 32 M such
	// streams, a ninth of the frame's), up to the 16 Mi that serve the whole frame best.  Measured, one
	// eighth of that frame: 27.5 / 24.7 / 23.5 / 24.9 / 37 ms with 16 / 8 / 4 / 2 / 1 Mi slots; a quarter:
	// 43.3 / 42.4 / 42.6 / 47.7 ms with 16 / 8 / 4 / 2 Mi.
	uint64_t group_blocks = max_group_blocks;
	if (p->slots <= 0) {
		const uint64_t pass_runs = (std::min<uint64_t>(n_blocks, pass_blocks) + run_blocks - 1) / run_blocks;
		const uint64_t listed_streams = (uint64_t)n_listed * (64 / lane_blocks) * pass_runs;
		const uint64_t want = std::max<uint64_t>(listed_streams, (uint64_t)1 << 20);
		group_blocks = std::max<uint64_t>(1, std::min<uint64_t>(max_group_blocks, (want / 64 + groups * waves_per_block - 1) / (groups * waves_per_block)));
	}
	const uint32_t group_slots = (uint32_t)(group_blocks * 256);
	const uint32_t n_slots = group_slots * (uint32_t)groups;
	base.n_slots = group_slots;
	// per group: 2 parities x kRayKinds kinds x kShards sub-queues, each able to take
	// everything its share of the group's workgroups can append in one launch
	const uint32_t shard_cap = (uint32_t)((group_blocks + kShards - 1) / kShards) * 256;
	const size_t group_queue = (size_t)2 * kRayKinds * kShards * shard_cap;
	const size_t group_bounce = (size_t)2 * kShards * shard_cap * 2;      // quads
	// one RenderArgs and one stream per slot group
	std::vector<RenderArgs> ga((size_t)groups, base);
	std::vector<hipStream_t> gs((size_t)groups);
	for (int g = 0; g < groups; g++) {
		const size_t off = (size_t)g * group_slots;
		RenderArgs &a = ga[g];
		a.ray0 = ctx->ray0.as<float4>() + off;
		a.ray1 = ctx->ray1.as<float4>() + off;
		a.hit = ctx->hit.as<float4>() + off;
		a.thr = ctx->thr.as<float4>() + off;
		a.col = ctx->col.as<float4>() + off;
		a.rng = ctx->rng.as<uint32_t>() + off;
		a.acc = ctx->acc.as<float4>() + off;
		a.meta = ctx->meta.as<uint4>() + off;
		if (base.hint)
			a.hint = base.hint + off;
		a.shadow = ctx->shadow.as<float4>() ? ctx->shadow.as<float4>() + 3 * (size_t)g * kShards * shard_cap : nullptr;
		a.q_ray = ctx->q_ray.as<uint32_t>() + (size_t)g * group_queue;
		a.q_bounce = ctx->q_bounce.as<float4>() + (size_t)g * group_bounce;
		a.q_camera = ctx->q_camera.as<float4>() + (size_t)g * (group_bounce / 2);
		a.shard_cap = shard_cap;
		a.grid_cap = shard_cap;
		a.cnt = ctx->counters.as<Counters>() + g;
		gs[g] = g == 0 ? ctx->stream : ctx->extra[g - 1];
	}

	HIP_TRY_R(hipEventRecord(ev_mid, s));
	for (int g = 1; g < groups; g++)
		HIP_TRY_R(hipStreamWaitEvent(gs[g], ev_mid, 0));
	// The wavefront loop.  Progress lives on the device; the host reads it back
	// every `poll` steps only to notice the end (steps past the end find every
	// slot retired and do nothing).  The groups' steps are issued round-robin on
	// their own streams and run concurrently.
	uint64_t steps = 0;
	auto timed = [&](hipStream_t st_, int kind, auto &&launch) -> hipError_t {
		if (!time_kernels) {
			launch();
			return hipSuccess;
		}
		hipEvent_t e0 = nullptr, e1 = nullptr;
		hipError_t e;
		if ((e = hipEventCreate(&e0)) != hipSuccess)
			return e;
		ev_pool.push_back(e0);
		if ((e = hipEventCreate(&e1)) != hipSuccess)
			return e;
		ev_pool.push_back(e1);
		ev_kind.push_back(kind);
		if ((e = hipEventRecord(e0, st_)) != hipSuccess)
			return e;
		launch();
		return hipEventRecord(e1, st_);
	};
	for (uint32_t g0 = 0; g0 < n_groups && n_parts; g0 += groups_per_pass) {
		const uint32_t g1 = std::min(n_groups, g0 + groups_per_pass);
		const uint32_t first_block = range_begin + g0 * blocks_per_group;
		const uint32_t end_block = (uint32_t)std::min<uint64_t>(range_end, (uint64_t)range_begin + (uint64_t)g1 * blocks_per_group);
		for (int g = 0; g < kMaxGroups; g++) {
			memset(&ctx->host_counters[g], 0, sizeof(Counters));
			// (the first pass of k_advance hands stream i to slot i without asking)
			ctx->host_counters[g].next_item = g < groups ? group_slots : 0;
		}
		HIP_TRY_R(hipMemcpyAsync(ctx->counters.ptr, ctx->host_counters, sizeof(Counters) * kMaxGroups, hipMemcpyHostToDevice, s));
		HIP_TRY_R(hipStreamSynchronize(s));       // host_counters is reused for read-back below; the groups are idle
		if (n_far) {
			// (k_sum_blocks knows these parts' block sums without anybody writing them)
			// each of these samples stands for one NearestHit call of the reference, like
			// the ones k_advance answers
			const uint64_t pass_samples = std::min<uint64_t>(spp, (uint64_t)end_block * sps) - (uint64_t)first_block * sps;
			st.rays += far_pixels * pass_samples;
			st.early_rays += far_pixels * pass_samples;
		}
		for (int g = 0; g < groups; g++) {
			// the sorted parts are dealt round-robin to the slot groups; a part's
			// streams are its pixels x the block runs of the pass
			const uint32_t parts_here = (n_listed + (uint32_t)groups - 1 - (uint32_t)g) / (uint32_t)groups;
			ga[g].first_block = first_block;
			ga[g].end_block = end_block;
			ga[g].group = (uint32_t)g;
			ga[g].n_groups = (uint32_t)groups;
			ga[g].streams_per_part = (64 / lane_blocks) * ((end_block - first_block + run_blocks - 1) / run_blocks);
			// Counters::next_item is bumped by every waiting wave even once the list is dry
			// (at most once per lane and pass), so the list's length plus the pool must
			// stay clear of 2^32
			if ((uint64_t)parts_here * ga[g].streams_per_part + n_slots >= 0xffffffffull) {
				cleanup();
				return ctx->fail(RTA_E_LIMIT, "pixels x sample blocks exceeds 2^32; raise samples_per_stream");
			}
			ga[g].n_streams = parts_here * ga[g].streams_per_part;
			ga[g].grab_at = grab_at;
			ga[g].grid_cap = shard_cap;
			launch_init_slots(gs[g], ga[g], group_slots);
		}
		bool done = false, fresh = true;      // fresh: the slots come straight from k_init_slots
		int poll = 16;
		while (!done) {
			for (int k = 0; k < poll; k++) {
				for (int g = 0; g < groups; g++) {
					RenderArgs &a = ga[g];
					a.first_pass = fresh ? 1u : 0u;
					const int par = (int)(steps & 1);
					// advance queues the step's rays; shadow comes before intersect, whose
					// first workgroup empties the occlusion queue for the next step
					HIP_TRY_R(timed(gs[g], 2, [&] { launch_advance(gs[g], a, par); }));
					// camera rays from their cells' lists: what their hits queue (a bounce ray, an occlusion query)
					// joins this step's queues
					if (a.camera_cells)
						HIP_TRY_R(timed(gs[g], 3, [&] { launch_intersect(gs[g], a, par, ctx->stack_depth, count_work, kStageCamera); }));
					if (lights)
						HIP_TRY_R(timed(gs[g], 1, [&] {
							launch_shadow(gs[g], a, ctx->stack_depth, count_work);
						}));
					if (time_kernels) {
						// (the three launches of the closest-hit part, each with its own pair of events)
						HIP_TRY_R(timed(gs[g], 0, [&] { launch_intersect(gs[g], a, par, ctx->stack_depth, count_work, kStageWalk); }));
						HIP_TRY_R(timed(gs[g], 4, [&] { launch_intersect(gs[g], a, par, ctx->stack_depth, count_work, kStageOdd); }));
					} else
						launch_intersect(gs[g], a, par, ctx->stack_depth, count_work, kStageWalk | kStageOdd);
				}
				steps++;
				fresh = false;
			}
			HIP_TRY_R(hipGetLastError());
			for (int g = 0; g < groups; g++)
				HIP_TRY_R(hipMemcpyAsync(&ctx->host_counters[g], ctx->counters.as<Counters>() + g, sizeof(Counters),
				                         hipMemcpyDeviceToHost, gs[g]));
			done = true;
			uint64_t live = 0;
			for (int g = 0; g < groups; g++) {
				HIP_TRY_R(hipStreamSynchronize(gs[g]));
				uint32_t slots = g < groups ? group_slots : 0;      // those that still hold or may get a path
				for (uint32_t r : ctx->host_counters[g].retired)
					slots -= r;
				done = done && slots == 0;
				live += slots;
				// A slot that has retired stays retired for the rest of the pass, so the slots still
				// live bound what any sub-queue can receive from now on: the traversal grids shrink
				// with the work instead of covering the queues' capacity.
				const uint64_t bound = ((uint64_t)slots + 255) / 256 * 256;
				ga[g].grid_cap = (uint32_t)std::min<uint64_t>(shard_cap, std::max<uint64_t>(bound, 256));
			}
			// near the end the host looks more often: steps past the end are empty launches
			if (live * 8 < n_slots)
				poll = 4;
		}
		for (int g = 0; g < groups; g++) {
			const Counters &c = ctx->host_counters[g];
			uint64_t early = 0;
			for (unsigned long long e : c.early)
				early += e;
			st.rays += c.rays + early;
			st.early_rays += early;
			st.shadow_rays += c.shadow_rays;
			st.inner_visits += c.inner;
			st.tri_tests += c.tri;
			st.sph_tests += c.sph;
			st.retraced_rays += c.retraced;
			st.walked_rays += c.walked;
			st.listed_rays += c.listed;
			st.cell_tests += c.cell_tests;
			st.scanned_rays += c.scanned;
		}
		// all groups are idle here (synchronised above)
		if (d_frame && n_owned)
			launch_sum_blocks(s, base, first_block, end_block - first_block, d_frame, g0 > 0);
	}
	if (d_frame && d_bgr && n_owned)
		launch_resolve(s, d_frame, base.pixels, n_pad, (float)spp, p->exposure, d_bgr);
	HIP_TRY_R(hipEventRecord(ev_end, s));
	HIP_TRY_R(hipGetLastError());
	HIP_TRY_R(hipStreamSynchronize(s));

	if (!to_device && n_owned) {
		// copy out only what this shard owns, row by row of its tiles
		std::vector<float> h_frame;
		std::vector<uint8_t> h_bgr;
		if (accum) {
			h_frame.resize(frame_px * 3);
			HIP_TRY_R(hipMemcpy(h_frame.data(), d_frame, frame_px * 12, hipMemcpyDeviceToHost));
			for (int32_t q : px)
				if (q >= 0)
					memcpy(accum + (size_t)q * 3, h_frame.data() + (size_t)q * 3, 12);
		}
		if (bgr) {
			h_bgr.resize(frame_px * 3);
			HIP_TRY_R(hipMemcpy(h_bgr.data(), d_bgr, frame_px * 3, hipMemcpyDeviceToHost));
			for (int32_t q : px)
				if (q >= 0)
					memcpy(bgr + (size_t)q * 3, h_bgr.data() + (size_t)q * 3, 3);
		}
	}

	float ms = 0;
	HIP_TRY_R(hipEventElapsedTime(&ms, ev_begin, ev_end));
	st.gpu_ms = ms;
	for (size_t k = 0; k < ev_kind.size(); k++) {
		float t = 0;
		HIP_TRY_R(hipEventElapsedTime(&t, ev_pool[2 * k], ev_pool[2 * k + 1]));
		if (ev_kind[k] == 0) {
			st.intersect_ms += t;
			st.intersect_launches++;
		} else if (ev_kind[k] == 2) {
			st.advance_ms += t;
			st.advance_launches++;
		} else if (ev_kind[k] == 3) {
			st.camera_ms += t;
			st.camera_launches++;
		} else if (ev_kind[k] == 4) {
			st.leftover_ms += t;
		} else {
			st.shadow_ms += t;
			st.shadow_launches++;
		}
	}
	st.samples = (uint64_t)n_owned * (std::min<uint64_t>(spp, (uint64_t)range_end * sps) - (uint64_t)range_begin * sps);
	st.steps = steps;
	if (stats)
		*stats = st;
	cleanup();
#undef HIP_TRY_R
	return RTA_OK;
}

}  // extern "C"
