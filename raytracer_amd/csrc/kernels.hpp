// Shared declarations between the HIP kernels (kernels.hip) and the host-side
// context that drives them (context.cpp).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "raytracer_amd.h"

namespace rta {

// 16-byte quads per primitive record: three hold it, the fourth keeps a record inside one 64-byte line
constexpr int kPrimQuads = 4;
constexpr uint32_t kShards = 64;      // sub-queues per ray queue
constexpr uint32_t kRayKinds = 4;     // camera rays, bounce rays, rays needing the division-based box test, rays to trace again

// Device-resident progress and statistics of one slot group (see RenderArgs).
struct Counters {
	uint32_t unused0;
	uint32_t next_item;      // streams of the group's list already handed out
	uint32_t unused1;
	uint32_t pad;
	unsigned long long rays, shadow_rays;
	// closest-hit queue lengths, [step parity][shard][kind]: the camera and the bounce counter of a
	// shard are the halves of one aligned 64-bit word, which k_advance bumps with a single atomic
	alignas(8) uint32_t n_ray[2 * kRayKinds * kShards];
	uint32_t n_shadow[kShards];                  // pending occlusion queries per sub-queue (one counter would have every wave of an advance launch wait in line for it)
	uint32_t head[2 * kShards * 2];              // queue entries already drawn by traversal waves, [step parity][shard][bounce, camera]
	unsigned long long inner, tri, sph;     // RTA_RENDER_COUNT_WORK only
	// closest-hit queries answered where the ray was generated; the sum of the words counts (every wave of an
	// advance launch may add to it: on ONE word they would queue up behind one another)
	unsigned long long early[kShards];
	unsigned long long retraced;            // closest-hit queries that the certified walk left to the exact traversal
	unsigned long long walked;              // closest-hit queries handed to the certified walk (retraced ones included)
	// slots that are through for the rest of the pass; the sum of the words counts (the waves of a render's end retire
	// together: on ONE word they would queue up behind one another)
	uint32_t retired[kShards];
	unsigned long long listed;              // camera rays answered from their cell's triangle list (retraced ones included)
	unsigned long long cell_tests;          // RTA_RENDER_COUNT_WORK only: triangle tests of those (also counted in `tri`)
	unsigned long long scanned;             // closest-hit queries that tested every primitive of a small scene (retraced ones included)
};

struct TreeRef {
	const float4 *pairs;
	const float4 *prims;
	uint32_t root;
	uint32_t tame;           // every box coordinate is 0 or within [2^-78, 2^40] in magnitude
	uint32_t box_known;      // box_lo/box_hi hold the box around the root's two children (tame trees with an inner root)
	float box_lo[3], box_hi[3];
	// The 4-wide tree of the certified traversal (traverse_wide.hpp, wide_tree.hpp); null when the scene has none.
	const float4 *wide;          // 4 quads per node
	const float4 *leaf_boxes;    // 2 quads per primitive: the box of its leaf in the reference's tree (lo, 1) (hi, 0); all 0: in no leaf
	float wide_center[3];        // a ray may take the wide tree when |origin - wide_center|^2 <= wide_reach2
	float wide_reach2;           //   and |direction|^2 <= kWideMaxDir^2
	uint32_t wide_index_mask;    // 2^bits - 1: the bits of a child reference, stack or list entry that number the node or primitive
	uint32_t wide_index_bits;    //   and that number of bits
	uint32_t wide_relative;      // distance bits of the entries count from the ray's entry into the scene's box (big scenes)
	uint32_t wide_key_base;      //   and are the bits of the float distance above this pattern (WideTraversal::distance_bits)
	uint32_t n_loose;            // spheres, which the walk tests one by one for every ray:
	const uint32_t *loose;       //   their primitive numbers
	uint32_t flat_prims;         // > 0: the scene has this few primitives, and rays may test them all instead of walking a tree (traverse_flat.hpp)
};

constexpr uint32_t kFlatMaxPrims = 64;          // scenes of up to this many primitives may be scanned (TreeRef::flat_prims)
constexpr uint32_t kWideLeaf = 0x80000000u;     // child reference of a wide node: kWideLeaf | primitive,
constexpr uint32_t kWideNone = 0xffffffffu;     //   a node number, or kWideNone for an empty child
constexpr float kWideMaxDir = 2.f;              // longest ray direction the wide tree's padding is computed for
constexpr int kBatchWalkCap = 22;               // entries of a lane's LDS column (node stack + triangle list) in the batch walks:
                                                //   seven workgroups (their registers' limit) fill a CU's 160 KB of LDS

// Everything a wavefront kernel needs, passed by value.  The slot pool is cut
// into groups that run their wavefront steps on different HIP streams, so that
// one group's kernels fill the GPU while another group's kernel drains its last
// long rays; a RenderArgs describes one group (pointers already offset to it).
struct RenderArgs {
	TreeRef tree;
	const float4 *shade_prims;     // 3 per primitive: (n0, material) (n1, light) (n2, type)
	const float4 *materials;       // 2 per material: (color, alpha) (ior, metallic, -, -)
	const float4 *lights;          // 7 per light: (color, prim) v0 v1 v2 n0 n1 n2
	int32_t n_lights;

	// per path slot
	float4 *ray0;                  // origin xyz, pdf of the sampled direction (not kept for a pinhole camera's rays)
	float4 *ray1;                  // direction xyz, bounce word or slot state
	float4 *hit;                   // primitive, t, u, v
	float4 *thr;                   // throughput rgb (kept from the second segment of a path on)
	float4 *col;                   // path radiance rgb (likewise)
	uint32_t *rng;                 // xorshift32 state of the slot's stream
	float4 *acc;                   // block sum rgb, samples finished in the current block
	uint4 *meta;                   // pixel, current block, end of the block run, index in the padded pixel list
	uint32_t n_slots;
	uint32_t *q_ray;               // closest-hit queues, [step parity][kind][shard][shard_cap] slot numbers
	// The bounce-ray queues also carry the rays, [step parity][shard][shard_cap] x 2 quads (origin, direction.x)
	// (direction.yz, slot, -): traversal waves take their rays in queue order from consecutive
	// lines instead of gathering them from the slot arrays, whose latency a refilling wave would sit out
	float4 *q_bounce;
	uint32_t shard_cap;            // entries per sub-queue, a multiple of the workgroup size
	uint32_t grid_cap;             // entries per sub-queue that a traversal launch's grid covers: a multiple of the workgroup
	                               // size, <= shard_cap, and at least the number of slots that can still queue a ray
	uint32_t rays_per_lane, refill_at;     // bounce-ray traversal: see traverse_pulling
	uint32_t spread_blocks;        //   a sub-queue's rays are spread over at least so many workgroups (fewer rays per lane)
	uint32_t leafwise;             // the tree has long leaves: traverse_leafwise, one ray per lane (rays_per_lane = 1)
	uint32_t walk_refill_at, walk_quorum, walk_waiting;      // traverse_wide_pulling's thresholds for the render's rays
	uint32_t sort_lanes;           // k_advance: the lanes of a workgroup trade slots so that its waves hold one kind of work each
	uint32_t camera_walk;          // camera rays take the certified walk too (the camera lies within the wide tree's reach)
	int32_t *hint;                 //   and start with the triangle the slot's last camera ray hit, [slot]; -1: none
	uint32_t wide_cap;             // > 0: bounce rays take the certified walk over the wide tree, with this many LDS entries per lane
	// Camera rays answered from per-cell triangle lists (cells.hip) instead: the camera queues' entries then
	// come with (direction, cell) in q_camera, [step parity][shard][shard_cap] - the cell is the ray's place on
	// the image plane, -1: none -, and cell_start[cell] .. cell_start[cell + 1] are the cell's entries of
	// cell_list (triangle numbers)
	uint32_t camera_cells;
	float4 *q_camera;
	uint32_t shade_first_hit;      //   and those that hit are shaded where they are answered (k_camera_cells; scenes without lights)
	const uint32_t *cell_start, *cell_list;
	const unsigned long long *cell_keys;      // per entry: cell << 32 | bits of the nearest t the entry's triangle can be hit at
	int32_t cell_w, cell_h;        // the map: the frame and kSkyBorder cells around it
	// occlusion queue, 3 per entry: (origin, limit) (direction, slot) (radiance, -)
	float4 *shadow;
	Counters *cnt;                 // this group's

	float4 *partial;               // block sums of the pass, [block - first_block][padded pixel]
	const int32_t *pixels;         // this shard's pixels, 64 per tile, -1 where a tile leaves the frame
	uint32_t n_tiles, n_pad;       // n_pad = 64 * n_tiles
	uint32_t n_blocks, run_blocks; // sample blocks per pixel; consecutive blocks a lane works through as one stream
	uint32_t lane_blocks;          // a tile is cut into lane_blocks parts of 64 / lane_blocks pixels (1, 2, 4 or 8)
	uint32_t first_block, end_block;       // the pass renders blocks [first_block, end_block)
	uint32_t group, n_groups;      // this slot group and how many there are
	uint32_t streams_per_part;     // pixels of a part x block runs of the pass
	uint32_t n_streams;            // in this group's list (Counters::next_item counts them out)
	uint32_t grab_at;              // waiting lanes that make a wave draw new streams
	uint32_t first_pass;           // k_advance: the pass right after k_init_slots (slot i draws stream i, Counters::next_item starts at n_slots)
	// Pinhole cameras looking at a tame tree: per pixel, the value u1 of the pixel
	// filter's first random number from which on a sample stays clear of every
	// leaf's box on the image plane (see k_sky_map); null when there is no such map.
	const float *sky_map;
	uint32_t bulk_trips;           // whole sample blocks a lane may finish per pass (pixels that only see the sky)
	uint32_t fast_trips;           // single samples a lane may finish per pass without tracing
	const uint32_t *part_order;    // tile * lane_blocks + part, heaviest first (see k_part_weights)
	uint32_t spp, sps;
	uint32_t width;
	uint32_t seed;
	int32_t max_bounces;
	float sky[3];
	rta_camera cam;
};

void launch_init_slots(hipStream_t s, const RenderArgs &a, uint32_t n_slots);
void launch_part_weights(hipStream_t s, const RenderArgs &a, uint32_t *weights, int depth);
// weight of a part none of whose pixels can send a sample to a leaf's box (see k_sky_map)
constexpr uint32_t kNothingToTrace = 0xffffffffu;
void launch_advance(hipStream_t s, const RenderArgs &a, int par);
// Where a pinhole camera sees a point: on the image plane at o + (point - o) * h / ((point - o) . n),
// pixel coordinates x = (that - ul) . du / du2, y likewise with dv.
struct Projection {
	double o[3], ul[3], du[3], dv[3], n[3];
	double h, du2, dv2;
	double margin;       // pixels by which every projected box is widened
	// Lens cameras: a ray starts at o + a lens_u + b lens_v with a, b in [-1, 1).  The lens lies in a
	// plane parallel to the image plane, so a point at depth h / s is seen at its pinhole position
	// shifted by (1 - s)(a lens_u + b lens_v): at most |1 - s| lens_x pixels across and |1 - s| lens_y
	// pixels down.  Both are 0 for a pinhole.
	double lens_x, lens_y;
};
constexpr int kSkyBorder = 3;     // cells around the frame that the cover map includes, and the search radius

// cover: (width + 2 kSkyBorder) x (height + 2 kSkyBorder) bytes; sky_map: width x height floats
void launch_sky_map(hipStream_t s, const TreeRef &tree, uint32_t n_pairs, const Projection &pr, uint32_t width,
                    uint32_t height, uint8_t *cover, float *sky_map);
// cells.hip: per-cell triangle lists of a pinhole camera.  counts, starts: map_w x map_h + 1 words each.  The triangles are
// those of the wide tree's n_wide nodes, or, for a scene without one (tree.wide == null), the n_prims primitive records
// that are triangles in some leaf.  The count pass leaves the number of entries in starts[map_w x map_h]; the fill pass
// writes them into `list`, nearest first.
size_t cell_scan_temp_bytes(size_t n);
hipError_t launch_cell_count(hipStream_t s, const TreeRef &tree, uint32_t n_wide, uint32_t n_prims, const Projection &pr, int map_w,
                             int map_h, uint32_t *counts, uint32_t *starts, void *temp, size_t temp_bytes);
size_t cell_sort_temp_bytes(size_t entries);
// list, keys: the cells' entries nearest first; keys[k] = cell number << 32 | bits of the least t at which the test can accept
// list[k] for a ray through the cell
hipError_t launch_cell_fill(hipStream_t s, const TreeRef &tree, uint32_t n_wide, uint32_t n_prims, const Projection &pr, int map_w,
                            int map_h, uint32_t *counts, const uint32_t *starts, size_t entries, uint32_t *list_unsorted,
                            unsigned long long *keys_unsorted, uint32_t *list, unsigned long long *keys, void *temp, size_t temp_bytes);
// The closest-hit part of a wavefront step, three launches in this order: camera rays from their cells' lists
// (k_camera_cells, where the render has such lists), the traversal proper (k_intersect / k_intersect_leafwise), the rays
// left over for the exact traversal (k_intersect_odd).  `stages`: which of them to issue.
constexpr int kStageCamera = 1, kStageWalk = 2, kStageOdd = 4, kStageAll = 7;
void launch_intersect(hipStream_t s, const RenderArgs &a, int par, int depth, bool count, int stages = kStageAll);
void launch_shadow(hipStream_t s, const RenderArgs &a, int depth, bool count);
// the pass's blocks (first_block .. first_block + n_blocks - 1, in a.partial) joined to the pixels' sums in `frame`
void launch_sum_blocks(hipStream_t s, const RenderArgs &a, uint32_t first_block, uint32_t n_blocks, float *frame, bool resume);
void launch_resolve(hipStream_t s, const float *frame, const int32_t *pixels, uint32_t n, float samples, float exposure,
                    uint8_t *bgr);
// `order`: null, or the n ray numbers in the order in which they are to be traced (n < 2^32);
// `count_from`: null, or where the device finds how many of the n entries of `order` are in use
void launch_batch_closest(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, int depth,
                          Counters *cnt, const uint32_t *order = nullptr, const uint32_t *count_from = nullptr);
// the certified walk over the wide tree; rays left in doubt are listed for launch_batch_closest (order = doubts)
// (`cnt`: null, or where the walk adds its node visits and triangle tests)
void launch_batch_walk(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, int cap,
                       const uint32_t *order, uint32_t *doubts, uint32_t *n_doubts, Counters *cnt = nullptr);
// the same for a scene that rays scan (tree.flat_prims > 0; traverse_flat.hpp); n < 2^32
void launch_batch_flat(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, uint32_t *doubts,
                       uint32_t *n_doubts, Counters *cnt = nullptr);
void launch_batch_occluded_flat(hipStream_t s, const TreeRef &tree, const float *rays, const float *tmax, uint64_t n, uint8_t *out,
                                int depth);
// the same by packets of 64 rays that walk the tree together (traverse_packet.hpp): consecutive rays of the batch, or
// of `order`; n < 2^32
void launch_batch_packet(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, const uint32_t *order,
                         uint32_t *doubts, uint32_t *n_doubts, Counters *cnt = nullptr);
// the any-hit walk over the wide tree (n < 2^32); queries it cannot answer for certain are traced exactly in the same launch
void launch_batch_occluded_walk(hipStream_t s, const TreeRef &tree, const float *rays, const float *tmax, uint64_t n,
                                uint8_t *out, int cap, int depth, const uint32_t *order);
void launch_batch_occluded(hipStream_t s, const TreeRef &tree, const float *rays, const float *tmax, uint64_t n,
                           uint8_t *out, int depth, const uint32_t *order = nullptr);
// sort_rays.hip: the tracing order of a big batch (by origin cell and direction octant)
size_t ray_order_temp_bytes(uint32_t n);
constexpr uint32_t kDescentCounters = 64;      // counters at descents[k * stride], whose sum is the number of out-of-order rays
// With stride >= 3, descents[k * stride + 1] count the waves whose 64 consecutive rays do NOT travel together
// (directions more than 2 degrees apart, or origins more than 1/64 of the scene's box) and descents[k * stride + 2]
// the waves with a ray that does not start where the batch's first ray starts.  direction_keys: null, or where the
// rays' direction keys go (the order that turns a pinhole camera's scan lines into compact patches).
void launch_ray_keys(hipStream_t s, const float *rays, uint32_t n, const float box_lo[3], const float box_hi[3], uint32_t *keys,
                     uint32_t *index, uint32_t *descents, uint32_t stride = 1, uint32_t *direction_keys = nullptr);
// all_bits: the keys are direction keys (32 bits), not origin keys (24)
hipError_t launch_ray_sort(hipStream_t s, uint32_t n, uint32_t *keys, uint32_t *keys_sorted, uint32_t *index, uint32_t *order,
                           void *temp, size_t temp_bytes, bool all_bits = false);
}  // namespace rta
