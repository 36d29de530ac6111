// Launchers of the probe library's kernels (variants.hip).
#pragma once

#include "kernels.hpp"

namespace rta {

void launch_arith(hipStream_t s, int op, const float *x, const float *y, float *out, uint64_t n);
// variants.hip: closest hit with one design decision changed (1 private stack,
// 2 division-based box test, 3 reference node layout); measurement only
void launch_variant(hipStream_t s, int variant, const TreeRef &tree, const float4 *reference_nodes, const float *soa_nodes,
                    uint32_t n_pairs, const float *rays, uint64_t n, rta_hit *hits, int depth);
// certified traversal of the wide tree over a device-resident batch, then (`redo`) the exact traversal for the
// rays left in doubt; without it those rays keep prim = -2
void launch_wide_batch(hipStream_t s, const TreeRef &tree, const float *rays, uint64_t n, rta_hit *hits, int depth, bool redo = true);
void launch_pairs_to_soa(hipStream_t s, const float4 *pairs, uint32_t n_pairs, float *soa);

}  // namespace rta
