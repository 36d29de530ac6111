/* Probe entry points: measurement and test hooks exported by
 * libraytracer_amd_probe.so, which works on contexts of libraytracer_amd.so and is
 * not needed to render.  They have no counterpart in the reference. */
#ifndef RAYTRACER_AMD_PROBE_H
#define RAYTRACER_AMD_PROBE_H

#include "raytracer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Device arithmetic probe for tests: out[i] = op(x[i], y[i]) evaluated on the
 * GPU with the kernels' own primitives.  op: 0 x/y, 1 sqrtf(x), 2 pm_sinf,
 * 3 pm_cosf, 4 pm_logf, 5 pm_powf(x,y), 6 fmaf(x,y,x), 7 x*y+x (unfused),
 * 8 the traversal's reciprocal-based quotient of x by y. */
int rta_debug_arith(rta_context *ctx, int op, const float *x, const float *y, float *out, int64_t n);

/* Measurement hook: closest hit as rta_trace_closest (variant 0) or with one
 * design decision changed - 1 traversal stack in private memory, 2 division-based
 * box test for every ray, 3 the reference's 32-byte node array with a node-number
 * stack, 4 two nested loops per ray instead of one, 5 struct-of-arrays nodes;
 * 20 the certified walk over the wide tree followed by the exact traversal of the
 * rays it leaves in doubt, 21 the walk alone: a ray in doubt comes back with
 * prim = -2, every other answer is one the walk has proven; 100 and up: scheduling and
 * occupancy experiments listed in csrc/variants.hip.  Same results (100 returns
 * a census instead of hits); all but 0 take device pointers only. */
int rta_debug_trace_variant(rta_context *ctx, int variant, const float *rays, int64_t n, rta_hit *hits, int mem);

/* Test hook: the device layouts of the uploaded scene, copied to the host.  which: 0 pair records, 1 primitive
 * records, 2 shading records, 3 the reference's node records, 4 the wide tree's nodes, 5 the leaf boxes; 7, 8, 9: the
 * camera's cell lists of the view that the last render prepared (per cell of the image plane - the frame and three
 * cells around it, row by row - where its entries start, one more word for their total; the entries' triangle
 * numbers; the entries' keys: cell << 32 | bits of the least t at which the triangle can be hit through the cell),
 * empty where that render made none;
 * 6: sixteen 32-bit words of scalars (pair count, root reference, tame, box known, the box's six floats, stack
 * depth, wide tree present, its depth, index mask, reach, span).  *bytes receives the size; out may be NULL to
 * ask for it, else cap must be at least that.  Used to compare the two upload paths (host-built and
 * device-built layouts). */
int rta_debug_scene_buffer(rta_context *ctx, int which, void *out, size_t cap, size_t *bytes);

#ifdef __cplusplus
}
#endif

#endif
