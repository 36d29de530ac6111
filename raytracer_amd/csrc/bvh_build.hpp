// The reference's BVH builder on the GPU (bvh_build.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace rta {

struct BuiltTree {
	std::vector<int32_t> perm;          // BVH-order position -> input position
	std::vector<float> node_bounds;     // 6 per node
	std::vector<int32_t> node_index, node_count;
	float device_ms = 0;                // kernels only, between the upload of the primitives and the read-back
	int levels = 0;
};

// Builds the tree of src/main.cpp:133-248 for n primitives given as on the host
// (type, 9 floats each: v0 v1 v2, or centre and radius).  Returns "" or an error
// text; `unsupported` is set when the input is one the device path declines
// (non-finite coordinates) and the host builder should be used.
std::string device_build_bvh(hipStream_t s, int32_t n, const int32_t *prim_type, const float *geom, BuiltTree &out,
                             bool &unsupported);

}  // namespace rta
