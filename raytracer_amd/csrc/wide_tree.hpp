// A second, 4-wide tree over the same primitives for the certified traversal
// (traverse_wide.hpp).  Built on the host at upload from the scene description;
// the reference's own tree (src/main.cpp:170-248) stays the one that defines
// every result.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "raytracer_amd.h"

namespace rta {

struct WideTree {
	bool enabled = false;
	bool boxes_complete = false;      // leaf_boxes holds every primitive's leaf, and the reference's boxes are nested
	bool certifiable = false;         // leaf_boxes and spheres are complete (also where the tree itself was refused)
	std::string why_not;              // when !enabled
	// 4 x 16 bytes per node (one 64-byte line): twelve words of binary16 planes of the four
	// children, relative to `center` and rounded outwards (lo.x lo.y lo.z hi.x hi.y hi.z, each as
	// two words: children 0,1 and children 2,3), then the four child references.  A reference is
	// a node number, kWideLeaf | primitive, or kWideNone; an empty child's box is inverted
	// (lo = +inf, hi = -inf).
	std::vector<float4> nodes;
	// 2 x 16 bytes per primitive: the box of its leaf in the reference's tree (lo, hi)
	std::vector<float4> leaf_boxes;
	// spheres are not in the tree: the walk tests each of them for every ray (at most kWideMaxSpheres)
	std::vector<int32_t> spheres;
	float center[3] = {0, 0, 0};      // rays may take the wide tree when |origin - center| <= reach
	float reach = 0;                  //   and |direction| <= kWideMaxDir
	float span = 0;                   // >= reach + the radius of the scene's sphere: no admitted ray of unit direction is in the scene beyond
	int depth = 0;                    // levels of wide nodes
	uint32_t n_nodes = 0;
	uint32_t index_mask = 0;          // 2^bits - 1 >= every node and primitive number
	double mean_pad_over_extent = 0;  // diagnostic: padding relative to the padded child boxes' mean extent
};

// Reach + scene radius, in scene radii.  Rays that start within the reach are admitted; the padding grows
// with it.  The smallest covers the sphere around the scene's box (bounce rays start there); the builder
// takes the widest of these (rays from outside: a camera's, a batch's) whose padding stays below
// kWideMaxPadding of the boxes' extents, binary16 rounding of the planes included.
constexpr double kWideReachRadii[] = {4.0, 3.0, 2.5, 2.25, 2.05};
constexpr double kWideMaxPadding = 0.03;
// The padding grows with the square of a triangle's longest edge (the reference's determinant threshold is
// absolute): a scene of large triangles would get boxes that all overlap.  Beyond this mean padding at the
// smallest reach the scene gets no wide tree.
constexpr double kWideHopelessPadding = 0.25;

// spheres a scene may have and still get a wide tree over its triangles (every ray tests them all, one after the
// other, from scalar operands: at this many they cost a ray about as much as its walk)
constexpr size_t kWideMaxSpheres = 48;

// primitives up to which the wide tree gets a surface-area-heuristic topology of its own (a host-side build,
// O(n log n)); beyond, it keeps the reference tree's
constexpr size_t kWideSahLimit = 2u << 20;

// `tame`: every box coordinate of the reference's tree is ordinary (see ray_is_tame)
WideTree build_wide_tree(const rta_scene_desc &d, bool tame);

}  // namespace rta
