/*
 * raytracer_amd.h -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The reference renderer (melby-md/raytracer) is one translation unit with no
 * plugin or FFI surface: its only contract is `raytracer FILE -> image.bmp`
 * (src/main.cpp:594-605,717).  The seam this library sits behind is the state
 * of the reference's `Scene` after parsing and BVH construction
 * (src/main.cpp:605-622) on the input side and the BGR byte image handed to
 * the BMP writer (src/main.cpp:650,698-706,717) on the output side.  Each entry
 * point below names the reference code it replaces.
 *
 * Conventions
 *  - every function returns RTA_OK (0) or a negative RTA_E* code and never
 *    exits or throws; rta_last_error() gives the text of the last failure on
 *    a context, rta_global_error() of the last failure without a context;
 *  - all arrays are plain, densely packed, caller-owned; the library copies
 *    what it needs during the call;
 *  - pointers are host pointers unless a `mem` argument says RTA_MEM_DEVICE,
 *    in which case they are device pointers on the context's GPU;
 *  - a context is bound to ONE GPU and is driven by one host thread at a time;
 *    multi-GPU jobs use one process (and one context) per GPU;
 *  - the compute entry points fail with RTA_E_NO_DEVICE when no GPU is
 *    present; there is no CPU fallback;
 *  - all device work runs on the context's own non-blocking HIP streams and
 *    every compute call returns only after that work has finished.  RTA_MEM_DEVICE
 *    buffers are NOT ordered against the caller's streams: whatever produces an
 *    input (rays, tmax, accum) must have completed before the call - synchronise
 *    the producing stream, or make rta_stream() wait on an event of it - and
 *    outputs may be read from any stream once the call has returned.
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	RTA_OK = 0,
	RTA_E_INVALID = -1,     /* bad argument or malformed scene description */
	RTA_E_PARSE = -2,       /* scene file syntax error */
	RTA_E_IO = -3,          /* file could not be read or written */
	RTA_E_NO_DEVICE = -4,   /* no usable GPU / HIP runtime failure at start-up */
	RTA_E_DEVICE = -5,      /* HIP call failed */
	RTA_E_NO_SCENE = -6,    /* compute call before rta_upload_scene */
	RTA_E_LIMIT = -7        /* size beyond what the kernels support */
};

enum { RTA_MEM_HOST = 0, RTA_MEM_DEVICE = 1 };
enum { RTA_PRIM_TRIANGLE = 0, RTA_PRIM_SPHERE = 1 };     /* src/main.h:1-4 */

/* ------------------------------------------------------------------ */
/* host side: scene file, BVH, camera, BMP                             */
/* ------------------------------------------------------------------ */

typedef struct rta_scene rta_scene;

/* Parse a scene file.  Replaces LoadScene (src/parser.cpp:238-417): same
 * grammar, defaults, sticky material / area-light state and squared-roughness
 * storage.  On a syntax error returns RTA_E_PARSE and writes the reference's
 * message "ERROR:<file>[<byte offset + 1>]: <msg>" (src/parser.cpp:20-30) to
 * err; an unreadable file gives RTA_E_IO and "Couldn't open '<file>'"
 * (src/parser.cpp:261-262).  Unlike the reference there is no 16384-object /
 * 64-material / 64-light cap (src/main.h:46-48). */
int rta_scene_load(const char *path, rta_scene **out, char *err, size_t errlen);
int rta_scene_parse(const char *name, const char *text, rta_scene **out, char *err, size_t errlen);

/* A scene of n_tri bare triangles (9 floats each: v0 v1 v2) with the parser's
 * flat normals (src/parser.cpp:337-343), default material, no lights. */
int rta_scene_from_triangles(const float *verts, int64_t n_tri, rta_scene **out);
void rta_scene_free(rta_scene *scene);

/* Replaces BuildBVH + UpdateBounds + the light re-link (src/main.cpp:133-248,
 * 617-622): midpoint split on the longest axis, swap partition, children at
 * consecutive indices; permutes the scene's primitives exactly as the reference
 * permutes `objects`.  Like the reference's, the builder bounds only the number
 * of pending second children (64, src/main.cpp:182-185), not the depth of the
 * tree; a tree deeper than 64 levels (degenerate, right-leaning input) can be
 * built and described but not uploaded: rta_upload_scene answers RTA_E_LIMIT. */
int rta_scene_build_bvh(rta_scene *scene);

typedef struct {
	float camera[3], look_at[3], up[3];
	float fov, defocus_angle, exposure;
	float sky[3];
	int32_t width, height, samples;
} rta_scene_settings;      /* the `render { }` block, src/parser.cpp:387-412 */

int rta_scene_get_settings(const rta_scene *scene, rta_scene_settings *out);
int rta_scene_set_settings(rta_scene *scene, const rta_scene_settings *in);

/* Flat views of a scene (valid until the scene is changed or freed).  This is
 * also the upload format: a caller with its own loader fills one by hand. */
typedef struct {
	int32_t n_prims, n_nodes, n_materials, n_lights;
	/* primitives in BVH order (src/main.h:28-36) */
	const int32_t *prim_type;      /* RTA_PRIM_* */
	const int32_t *prim_material;  /* index into materials */
	const int32_t *prim_light;     /* index into lights or -1 */
	const float *prim_geom;        /* 9 floats: triangle v0 v1 v2 | sphere center, radius, 5 unused */
	const float *prim_normals;     /* 9 floats: n0 n1 n2 (unused for spheres) */
	/* BVH nodes (src/main.cpp:20-23): node 0 is the root; count > 0 marks a
	 * leaf whose primitives are [index, index+count); otherwise `index` is the
	 * first child and index+1 the second */
	const float *node_bounds;      /* 6 floats: min xyz, max xyz */
	const int32_t *node_index;
	const int32_t *node_count;
	const float *materials;        /* 6 floats: color rgb, roughness (alpha = r^2 as stored by the parser), ior, metallic */
	const float *light_color;      /* 3 floats */
	const int32_t *light_prim;     /* primitive (BVH order) that emits */
} rta_scene_desc;

int rta_scene_describe(const rta_scene *scene, rta_scene_desc *out);
/* The checks rta_upload_scene applies (index ranges, tree shape, no primitive
 * in two leaves: RTA_E_INVALID; depth <= 64, the reference's traversal stack,
 * src/main.cpp:339: RTA_E_LIMIT), without a GPU.  err receives what is wrong. */
int rta_validate_scene(const rta_scene_desc *desc, char *err, size_t errlen);
/* BVH-order position -> position in the scene file (the build's permutation). */
int rta_scene_permutation(const rta_scene *scene, const int32_t **perm, int32_t *n);

typedef struct {
	float origin[3], upper_left[3], du[3], dv[3], lens_u[3], lens_v[3];
	float defocus_angle;
} rta_camera;

/* Replaces the camera set-up of src/main.cpp:624-648. */
int rta_camera_from_settings(const rta_scene_settings *s, rta_camera *out);

/* Replaces WriteBMP (src/main.cpp:48-81): 24-bit, bottom-up, rows padded to 4
 * bytes; bgr is height*width*3 bytes, top row first. */
int rta_write_bmp(const char *path, int32_t width, int32_t height, const uint8_t *bgr);

const char *rta_global_error(void);

/* ------------------------------------------------------------------ */
/* device side                                                         */
/* ------------------------------------------------------------------ */

typedef struct rta_context rta_context;

int rta_device_count(void);
int rta_create(int device, rta_context **out);
void rta_destroy(rta_context *ctx);
const char *rta_last_error(const rta_context *ctx);

/* The same tree as rta_scene_build_bvh - node numbering, boxes and primitive
 * order bit for bit - built on the context's GPU (BuildBVH, src/main.cpp:133-248:
 * level by level, the two-pointer partition of src/main.cpp:201-210 in closed
 * form).  Meant for large inputs: 10 M triangles take seconds on the host.  The
 * scene ends up exactly as after rta_scene_build_bvh; scenes with non-finite
 * coordinates are handed to the host builder.  *device_ms (optional) receives
 * the time of the build kernels, 0 when the host builder ran. */
int rta_scene_build_bvh_device(rta_context *ctx, rta_scene *scene, float *device_ms);

/* Validates the description (index ranges, tree shape, every primitive in at
 * most one leaf; RTA_E_LIMIT for a tree deeper than 64 levels) and builds the
 * device layouts.  Replaces the hand-over of `Scene`/`BVHTree` to the render
 * loop (src/main.cpp:607-615).  A node whose box has min > max on some axis is
 * legal (the reference would take it as it is); such a tree is traversed with
 * the division-based box test only, which picks the planes by the direction's
 * sign bit exactly as src/main.cpp:303-330 does. */
int rta_upload_scene(rta_context *ctx, const rta_scene_desc *desc);

typedef struct {
	int32_t prim;       /* BVH-order primitive index, -1 on a miss */
	float t;            /* FLT_MAX on a miss */
	float u, v;         /* barycentric weights of v0 and v1 (triangles) */
} rta_hit;

/* Closest hit of n rays (6 floats each: origin xyz, direction xyz).  Replaces
 * NearestHit (src/main.cpp:332-422) with the same tree, visit order, tie rule
 * and arithmetic, so prim/t/u/v equal the reference's bit for bit.  Batches of
 * 65536 rays and more over a tree of more than 8 MiB are traced in the order
 * of their origins (16 bytes of workspace per ray) unless they come in that
 * order; hits[i] always belongs to ray i. */
int rta_trace_closest(rta_context *ctx, const float *rays, int64_t n, rta_hit *hits, int mem);

/* rta_trace_closest for callers that cannot allocate device memory (tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in
 * the reference.  Like every compute call it runs on the context's own stream
 * and returns when the work is done, so it is not itself capturable into a
 * caller's HIP graph.  rays and hits are DEVICE pointers on
 * the context's GPU; results equal rta_trace_closest's bit for bit.  The batch
 * is traced in the order of its rays' origins, and the workspace for that order
 * (16 bytes per ray + the sort's temporary storage) is a fixed array in device
 * memory that the library holds on every GPU it runs on (24 MiB + 256 bytes)
 * instead of a buffer that grows with n: the call makes no hipMalloc, hipFree
 * or other allocating call, and rejects n > RTA_STATIC_MAX_RAYS with
 * RTA_E_LIMIT.  All such calls on one GPU share that array: calls that overlap
 * in time on one GPU (from several contexts or threads) race. */
#define RTA_STATIC_MAX_RAYS (1 << 20)
int rta_trace_closest_static(rta_context *ctx, const float *rays, int64_t n, rta_hit *hits);

/* What the last rta_trace_closest call on this context did, kernel by
 * kernel: HIP events recorded on the context's stream
 * around the launches.  The reference has no counterpart (it carries neither
 * timing code nor ray counters, src/main.cpp:594-720); benchmarks price the
 * traversal kernels with it.  The work counts are filled only by a call whose
 * `mem` argument carried RTA_TRACE_COUNT_WORK (slower kernels). */
enum {                                /* OR-ed into the `mem` argument of rta_trace_closest */
	RTA_TRACE_COUNT_WORK = 0x100,     /* count node visits and primitive tests (rta_batch_stats) */
	RTA_TRACE_EXACT_ONLY = 0x200      /* every ray through the traversal that replays the reference's visit order, as
	                                   * RTA_RENDER_EXACT_ONLY does for a render (same hits either way) */
};

typedef struct {
	int64_t rays;
	int32_t sorted;               /* the batch was traced in an order of its own making: of its rays' origins, or (packets) directions */
	int32_t walked;               /* the batch went through the walk over the wide tree first */
	int32_t packets;              /* ... 64 rays at a time together (a pinhole camera's batch: one origin, neighbours travel together) */
	int32_t scanned;              /* 1: the scene has so few primitives that every ray tested them all, without a tree */
	uint64_t retraced_rays;       /* of a walked batch: rays left to the order-replaying traversal */
	double order_ms;              /* sort keys and radix sort */
	double walk_ms;               /* the walk's kernel */
	double replay_ms;             /* the order-replaying kernel (the whole batch when !walked) */
	uint64_t walk_node_visits;    /* 64-byte nodes of the wide tree read by the walk */
	uint64_t walk_tri_tests;      /* triangle records (48 bytes) tested by the walk */
	uint64_t replay_node_visits;  /* 64-byte pair records read by the order-replaying traversal */
	uint64_t replay_prim_tests;   /* primitive records (48 bytes) it tested */
} rta_batch_stats;

int rta_last_batch_stats(const rta_context *ctx, rta_batch_stats *out);

/* Any hit closer than tmax[i].  Replaces Occluded (src/main.cpp:424-478). */
int rta_trace_occluded(rta_context *ctx, const float *rays, const float *tmax, int64_t n, uint8_t *out, int mem);

typedef struct {
	int32_t width, height, samples;
	float exposure;
	float sky[3];
	uint32_t seed;                /* base of the per-(pixel, block) stream seeds */
	int32_t samples_per_stream;   /* samples served by one stream; <= 0: all of a pixel's samples */
	int32_t max_bounces;          /* <= 0: 10, the reference's value (src/main.cpp:501) */
	/* pixel ownership: 8x8 tiles numbered row-major; this call renders the
	 * tiles with tile_id % shard_count == shard_index (1 / 0: whole frame) */
	int32_t shard_index, shard_count;
	int32_t slots;                /* paths in flight; <= 0: library default */
	int32_t out_mem;              /* RTA_MEM_* of accum and bgr */
	int32_t flags;                /* RTA_RENDER_* */
	int32_t run_blocks;           /* consecutive sample blocks of a tile a wave takes at once (scheduling
	                               * only, results do not depend on it); <= 0: library default */
	int32_t streams;              /* parts the slot pool is cut into, each stepping on its own HIP stream
	                               * (scheduling only); <= 0: library default, at most 8 */
	/* sample-range sharding: render only the sample blocks [block_begin, block_end) of every owned
	 * pixel (a block = samples_per_stream samples; 0, 0 = all blocks).  accum then receives the sum
	 * over those blocks only and bgr must be NULL: the caller adds the partial frames of all ranges
	 * (a sum-reduce over the GPUs) and calls rta_resolve.  The union of the ranges draws exactly the
	 * samples of an unsplit render; the frame differs from it only by the order of the last fp32
	 * additions, which is why tile sharding (shard_index / shard_count), exact, is the default. */
	int32_t block_begin, block_end;
} rta_render_params;

enum {
	RTA_RENDER_TIME_KERNELS = 1,   /* bracket every advance, shadow and intersect launch with HIP events (stats.*_ms) */
	RTA_RENDER_COUNT_WORK = 2,     /* count node visits / primitive tests (slower kernels) */
	RTA_RENDER_SMALL_PASSES = 4,   /* hold as few sample blocks in memory at a time as possible (see rta_render) */
	RTA_RENDER_EXACT_ONLY = 8      /* every ray through the traversal that replays the reference's visit order; without it
	                                * bounce rays and occlusion queries of triangle scenes walk a second, 4-wide tree in any
	                                * order and are answered from that walk only where it proves NearestHit's / Occluded's
	                                * result (same bits either way) */
};

typedef struct {
	uint64_t samples, rays, shadow_rays;    /* rays = closest-hit queries; shadow_rays = occlusion queries */
	uint64_t steps;                         /* wavefront iterations */
	uint64_t inner_visits, tri_tests, sph_tests;   /* RTA_RENDER_COUNT_WORK only */
	double gpu_ms;                          /* whole render on the context's stream */
	double intersect_ms, shadow_ms;         /* RTA_RENDER_TIME_KERNELS only: summed launch durations */
	uint64_t intersect_launches, shadow_launches;
	uint64_t early_rays;                    /* of `rays`: camera rays that provably miss every leaf's box (the
	                                         * reference's NearestHit would test no primitive), answered where
	                                         * they are generated and never queued for traversal; rays - early_rays
	                                         * is the number of closest-hit queries that walked the tree */
	double advance_ms;                      /* RTA_RENDER_TIME_KERNELS only: summed durations of the advance launches */
	uint64_t advance_launches;
	uint64_t retraced_rays;                 /* of `rays`: bounce rays whose walk over the wide tree could not prove its
	                                         * answer (an exact tie, a hit on the rim of its leaf's box, ...) and that the
	                                         * order-replaying traversal answered instead */
	uint64_t walked_rays;                   /* of `rays`: handed to the walk over the wide tree (0 for scenes without one) */
	uint64_t listed_rays;                   /* of `rays`: a pinhole camera's rays answered from the triangle list of their cell
	                                         * on the image plane, without a tree (retraced ones included) */
	uint64_t cell_tests;                    /* RTA_RENDER_COUNT_WORK only: of tri_tests, those made for listed_rays */
	double camera_ms, leftover_ms;          /* RTA_RENDER_TIME_KERNELS only: summed durations of the launches that answer
	                                         * listed_rays, and of those that trace what the certified answers left over;
	                                         * intersect_ms is the traversal kernel's alone */
	uint64_t camera_launches;
	uint64_t scanned_rays;                  /* of `rays`: answered by testing every primitive of a scene of a few dozen, without
	                                         * a tree (retraced ones included; 0 for larger scenes) */
} rta_stats;

/* The render loop.  Replaces src/main.cpp:657-714 (row loop, per-sample ray
 * generation, RayTrace, accumulation, tone map, B,G,R store) on the GPU.
 * Random streams: the reference's single stream per image row
 * (src/main.cpp:659) serialises a row, so here every (pixel, block) pair owns a
 * xorshift32 stream: block b of pixel p = row*width + col serves samples
 * [b*samples_per_stream, (b+1)*samples_per_stream) and is seeded
 *     h = mix(p + seed); h = mix(h ^ (b*0x9e3779b9 + 0x7f4a7c15)); h ? h : 0x9e3779b9
 * with mix(h) = { h ^= h>>16; h *= 0x7feb352d; h ^= h>>15; h *= 0x846ca68b;
 * h ^= h>>16 }.  A block's samples are summed in order starting from 0, and a
 * pixel's block sums are summed in block order starting from 0, all in fp32,
 * so the result does not depend on scheduling, slot count or sharding.
 * Memory: a block's sum waits (16 bytes per pixel and block) until the pixel's
 * earlier blocks are in; when all blocks together would need more than 32 GiB
 * they are rendered in passes over consecutive blocks (RTA_RENDER_SMALL_PASSES:
 * always, in the smallest passes), with the same sums.  A pinhole camera
 * outside the scene's box adds 5 bytes per pixel (which samples need no ray).
 * accum: height*width*3 floats, per-pixel radiance SUM over samples (may be
 * NULL); bgr: height*width*3 bytes B,G,R top row first (may be NULL).  Only the
 * pixels owned by this shard are written. */
int rta_render(rta_context *ctx, const rta_camera *camera, const rta_render_params *params, float *accum,
               uint8_t *bgr, rta_stats *stats);

/* Tone-map a full frame of radiance sums (e.g. one gathered from several GPUs)
 * to B,G,R bytes: src/main.cpp:692-706.  Both pointers live in `mem`. */
int rta_resolve(rta_context *ctx, const float *accum, int32_t width, int32_t height, int32_t samples, float exposure,
                uint8_t *bgr, int mem);

/* Pixels (row*width + col) owned by a shard, in the order the renderer walks
 * them.  Call with out == NULL to get the count. */
int64_t rta_shard_pixels(int32_t width, int32_t height, int32_t shard_index, int32_t shard_count, int32_t *out);

void *rta_stream(rta_context *ctx);     /* the hipStream_t all work of the context is enqueued on */

#ifdef __cplusplus
}
#endif
#endif
