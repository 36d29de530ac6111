// The probe library (libraytracer_amd_probe.so): measurement and test hooks that
// work on a context of the product library but are no part of it - traversal
// variants (variants.hip) and a device arithmetic probe.  Declared in
// include/raytracer_amd_probe.h.
#include <cstdint>
#include <cstring>

#include "context.hpp"
#include "probe.hpp"
#include "raytracer_amd_probe.h"

using namespace rta;

extern "C" {

int rta_debug_trace_variant(rta_context *ctx, int variant, const float *rays, int64_t n, rta_hit *hits, int mem)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (variant == 0)
		return rta_trace_closest(ctx, rays, n, hits, mem);
	if (variant == 10)      // every ray through the replaying traversal, as scenes without a wide tree are traced
		return rta_internal_trace_closest(ctx, rays, n, hits, mem, 0);
	if (!ctx->have_scene)
		return ctx->fail(RTA_E_NO_SCENE, "no scene uploaded");
	if (variant == 20 || variant == 21) {
		if (!ctx->have_wide)
			return ctx->fail(RTA_E_INVALID, "the scene has no wide tree");
		if (n <= 0 || !rays || !hits || mem != RTA_MEM_DEVICE)
			return ctx->fail(RTA_E_INVALID, "variants 20 and 21 take device-resident ray batches");
		HIP_TRY(ctx, hipSetDevice(ctx->device));
		launch_wide_batch(ctx->stream, tree_of(ctx), rays, (uint64_t)n, hits, ctx->stack_depth, variant == 20);
		HIP_TRY(ctx, hipGetLastError());
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return RTA_OK;
	}
	if (variant < 1 || (variant > 5 && variant < 100) || variant > 360 || (variant >= 200 && variant - 200 < ctx->stack_depth) || n <= 0 || !rays || !hits || mem != RTA_MEM_DEVICE)
		return ctx->fail(RTA_E_INVALID, "variants 1..4 take device-resident ray batches");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (variant == 5 && !ctx->have_soa) {
		HIP_TRY(ctx, ctx->soa_nodes.reserve((size_t)ctx->n_pairs * 14 * 4));
		launch_pairs_to_soa(ctx->stream, ctx->pairs.as<const float4>(), ctx->n_pairs, ctx->soa_nodes.as<float>());
		ctx->have_soa = true;
	}
	launch_variant(ctx->stream, variant, tree_of(ctx), ctx->ref_nodes.as<const float4>(), ctx->soa_nodes.as<const float>(),
	               ctx->n_pairs, rays, (uint64_t)n, hits, ctx->stack_depth);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return RTA_OK;
}

int rta_debug_arith(rta_context *ctx, int op, const float *x, const float *y, float *out, int64_t n)
{
	if (!ctx)
		return RTA_E_INVALID;
	if (n <= 0 || !x || !y || !out)
		return ctx->fail(RTA_E_INVALID, "bad arguments");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, ctx->io_a.reserve((size_t)n * 4));
	HIP_TRY(ctx, ctx->io_b.reserve((size_t)n * 4));
	HIP_TRY(ctx, ctx->io_c.reserve((size_t)n * 4));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->io_a.ptr, x, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->io_b.ptr, y, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
	launch_arith(ctx->stream, op, ctx->io_a.as<const float>(), ctx->io_b.as<const float>(), ctx->io_c.as<float>(),
	             (uint64_t)n);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipMemcpyAsync(out, ctx->io_c.ptr, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return RTA_OK;
}

int rta_debug_scene_buffer(rta_context *ctx, int which, void *out, size_t cap, size_t *bytes)
{
	if (!ctx || !bytes)
		return RTA_E_INVALID;
	if (!ctx->have_scene)
		return ctx->fail(RTA_E_NO_SCENE, "no scene uploaded");
	uint32_t scalars[16] = {0};
	const void *src = nullptr;
	size_t n = 0;
	switch (which) {
	case 0: src = ctx->pairs.ptr, n = (size_t)ctx->n_pairs * 64; break;
	case 1: src = ctx->prims.ptr, n = (size_t)ctx->n_prims * 16 * kPrimQuads; break;
	case 2: src = ctx->shade_prims.ptr, n = (size_t)ctx->n_prims * 48; break;
	case 3: src = ctx->ref_nodes.ptr, n = (size_t)ctx->n_nodes * 32; break;
	case 4: src = ctx->wide_nodes.ptr, n = ctx->have_wide ? (size_t)ctx->wide_n_nodes * 64 : 0; break;
	case 5: src = ctx->leaf_boxes.ptr, n = ctx->have_leaf_boxes ? (size_t)ctx->n_prims * 32 : 0; break;
	// the cell lists of the view that the last render prepared (cells.hip)
	case 7:
	case 8:
	case 9: {
		const rta_context::Prepared &pre = ctx->prepared;
		if (!pre.valid || !pre.with_cells) {
			n = 0;
			break;
		}
		const size_t cells = ((size_t)pre.width + 2 * kSkyBorder) * ((size_t)pre.height + 2 * kSkyBorder);
		if (which == 7)
			src = ctx->cell_starts.ptr, n = (cells + 1) * 4;
		else if (which == 8)
			src = ctx->cell_list.ptr, n = (size_t)pre.cell_entries * 4;
		else
			src = ctx->cell_keys.ptr, n = (size_t)pre.cell_entries * 8;
		break;
	}
	case 6: {
		scalars[0] = ctx->n_pairs, scalars[1] = ctx->root_ref, scalars[2] = ctx->tame, scalars[3] = ctx->box_known;
		memcpy(scalars + 4, ctx->box, 24);
		scalars[10] = (uint32_t)ctx->stack_depth, scalars[11] = ctx->have_wide;
		if (ctx->have_wide) {
			scalars[12] = (uint32_t)ctx->wide_depth, scalars[13] = ctx->wide_index_mask;
			memcpy(scalars + 14, &ctx->wide_reach, 4);
			memcpy(scalars + 15, &ctx->wide_span, 4);
		}
		*bytes = sizeof scalars;
		if (out) {
			if (cap < sizeof scalars)
				return ctx->fail(RTA_E_INVALID, "buffer too small");
			memcpy(out, scalars, sizeof scalars);
		}
		return RTA_OK;
	}
	default: return ctx->fail(RTA_E_INVALID, "no such buffer");
	}
	*bytes = n;
	if (!out || n == 0)
		return RTA_OK;
	if (cap < n)
		return ctx->fail(RTA_E_INVALID, "buffer too small");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpy(out, src, n, hipMemcpyDeviceToHost));
	return RTA_OK;
}

}  // extern "C"
