// Order in which a large ray batch is traced over a tree that does not fit the
// caches: by where the rays start (Morton code of the origin in the scene's box,
// 7 bits per axis) and which way they point (direction octant).  Rays that a wave
// traces together then walk the same upper levels of the tree and start in the
// same leaves, so most of their node records come from the cache of the wave
// before: incoherent rays over the 10 M-triangle soup go from 0.51 to 0.99
// Grays/s, the same hits in the same places of the output (only the order of
// tracing changes, every ray's traversal is its own).
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.hpp"

namespace rta {

namespace {

constexpr int kBlock = 256;
constexpr int kBitsPerAxis = 7;
constexpr unsigned kKeyBits = 3 * kBitsPerAxis + 3;

__device__ inline uint32_t spread3(uint32_t v)      // 7 bits -> every third bit
{
	uint32_t out = 0;
	for (int b = 0; b < kBitsPerAxis; b++)
		out |= ((v >> b) & 1u) << (3 * b);
	return out;
}

__device__ inline uint32_t ray_key(const float *r, const float lo_x, const float lo_y, const float lo_z, const float sx,
                                   const float sy, const float sz)
{
	// position in the box scaled to [0, 128), rays that start outside (or nowhere: NaN) at the nearest face
	const float fx = __builtin_fminf(__builtin_fmaxf((r[0] - lo_x) * sx, 0.f), 127.f);
	const float fy = __builtin_fminf(__builtin_fmaxf((r[1] - lo_y) * sy, 0.f), 127.f);
	const float fz = __builtin_fminf(__builtin_fmaxf((r[2] - lo_z) * sz, 0.f), 127.f);
	const uint32_t cell = spread3((uint32_t)fx) | (spread3((uint32_t)fy) << 1) | (spread3((uint32_t)fz) << 2);
	const uint32_t octant = (__float_as_uint(r[3]) >> 31) | ((__float_as_uint(r[4]) >> 31) << 1) | ((__float_as_uint(r[5]) >> 31) << 2);
	return (cell << 3) | octant;
}

// Where a direction points, as a 32-bit Morton code of its octahedral map (16 bits per axis: 3e-5 of the sphere's
// quarter arc per step): rays from one point sorted by it come out in compact patches of neighbouring directions.
__device__ inline uint32_t spread2(uint32_t v)      // 16 bits -> every second bit
{
	v = (v | (v << 8)) & 0x00ff00ffu;
	v = (v | (v << 4)) & 0x0f0f0f0fu;
	v = (v | (v << 2)) & 0x33333333u;
	v = (v | (v << 1)) & 0x55555555u;
	return v;
}

__device__ inline uint32_t direction_key(const float *r)
{
	const float x = r[3], y = r[4], z = r[5];
	const float s = __builtin_fabsf(x) + __builtin_fabsf(y) + __builtin_fabsf(z);
	float px = x / s, py = y / s;
	if (z < 0.f) {
		const float qx = (1.f - __builtin_fabsf(py)) * (px < 0.f ? -1.f : 1.f), qy = (1.f - __builtin_fabsf(px)) * (py < 0.f ? -1.f : 1.f);
		px = qx;
		py = qy;
	}
	// [-1, 1] -> [0, 65535]; NaN and the like land at 0
	const float fx = __builtin_fminf(__builtin_fmaxf((px * .5f + .5f) * 65535.f, 0.f), 65535.f);
	const float fy = __builtin_fminf(__builtin_fmaxf((py * .5f + .5f) * 65535.f, 0.f), 65535.f);
	return spread2((uint32_t)fx) | (spread2((uint32_t)fy) << 1);
}

// keys[i], index[i] = i, and in descents[k * stride], k < kDescentCounters, together the number of rays whose
// key is below the key of the ray before them (0 for a batch that is in order already).  One atomic per
// workgroup, spread over counters that a caller with room for them keeps in different 128-byte lines: atomics on
// one line are served one after the other by its L2 channel (a million of them, one per wave of 64 Mi
// incoherent rays on 64 neighbouring words, took 5 ms).
__global__ __launch_bounds__(kBlock) void k_ray_keys(const float *rays, const uint32_t n, const float lo_x, const float lo_y,
                                                    const float lo_z, const float sx, const float sy, const float sz,
                                                    uint32_t *keys, uint32_t *index, uint32_t *descents, const uint32_t stride,
                                                    uint32_t *direction_keys)
{
	__shared__ uint32_t block_descents, block_spread, block_elsewhere;
	if (threadIdx.x == 0)
		block_descents = block_spread = block_elsewhere = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	bool descent = false, apart = false, elsewhere = false;
	if (i < n) {
		const float *r = rays + (size_t)i * 6;
		const uint32_t key = ray_key(r, lo_x, lo_y, lo_z, sx, sy, sz);
		keys[i] = key;
		index[i] = i;
		descent = i > 0 && key < ray_key(rays + (size_t)(i - 1) * 6, lo_x, lo_y, lo_z, sx, sy, sz);
		if (stride >= 3) {
			// does this ray travel with the first ray of its wave?  Directions within 2 degrees (cos^2 = 0.99878), origins
			// within 2 of the 128 cells per axis that the keys are made of; NaNs count as apart
			const float *f = rays + (size_t)(i & ~63u) * 6;
			const float dd = r[3] * f[3] + r[4] * f[4] + r[5] * f[5];
			const float d2 = r[3] * r[3] + r[4] * r[4] + r[5] * r[5], f2 = f[3] * f[3] + f[4] * f[4] + f[5] * f[5];
			const float ox = (r[0] - f[0]) * sx, oy = (r[1] - f[1]) * sy, oz = (r[2] - f[2]) * sz;
			apart = !(dd > 0.f && dd * dd >= 0.99878f * d2 * f2 && ox * ox + oy * oy + oz * oz <= 4.f);
			// does it start where the batch's first ray starts (a pinhole camera's rays do)?
			const float px = (r[0] - rays[0]) * sx, py = (r[1] - rays[1]) * sy, pz = (r[2] - rays[2]) * sz;
			elsewhere = !(px * px + py * py + pz * pz <= 1e-8f);
		}
		if (direction_keys)
			direction_keys[i] = direction_key(r);
	}
	const unsigned long long votes = __ballot(descent), spread = __ballot(apart), others = __ballot(elsewhere);
	if ((threadIdx.x & 63) == 0 && others)
		atomicAdd(&block_elsewhere, 1u);
	if ((threadIdx.x & 63) == 0 && votes)
		atomicAdd(&block_descents, (uint32_t)__popcll(votes));
	if ((threadIdx.x & 63) == 0 && spread)
		atomicAdd(&block_spread, 1u);
	__syncthreads();
	if (threadIdx.x == 0 && block_descents)
		atomicAdd(descents + (size_t)(blockIdx.x % kDescentCounters) * stride, block_descents);
	if (threadIdx.x == 0 && block_spread)
		atomicAdd(descents + (size_t)(blockIdx.x % kDescentCounters) * stride + 1, block_spread);
	if (threadIdx.x == 0 && block_elsewhere)
		atomicAdd(descents + (size_t)(blockIdx.x % kDescentCounters) * stride + 2, block_elsewhere);
}

}  // namespace

size_t ray_order_temp_bytes(uint32_t n)
{
	size_t bytes = 0;
	uint32_t *none = nullptr;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, none, none, none, none, n, 0, 32u, hipStream_t(nullptr));
	return bytes;
}

void launch_ray_keys(hipStream_t s, const float *rays, uint32_t n, const float box_lo[3], const float box_hi[3], uint32_t *keys,
                     uint32_t *index, uint32_t *descents, uint32_t stride, uint32_t *direction_keys)
{
	float scale[3];
	for (int k = 0; k < 3; k++) {
		const float extent = box_hi[k] - box_lo[k];
		scale[k] = extent > 0.f ? 128.f / extent : 0.f;
	}
	(void)hipMemsetAsync(descents, 0, ((size_t)(kDescentCounters - 1) * stride + (stride >= 3 ? 3 : 1)) * sizeof(uint32_t), s);
	hipLaunchKernelGGL(k_ray_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, rays, n, box_lo[0], box_lo[1], box_lo[2],
	                   scale[0], scale[1], scale[2], keys, index, descents, stride, direction_keys);
}

hipError_t launch_ray_sort(hipStream_t s, uint32_t n, uint32_t *keys, uint32_t *keys_sorted, uint32_t *index, uint32_t *order,
                           void *temp, size_t temp_bytes, bool all_bits)
{
	return rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_sorted, index, order, n, 0, all_bits ? 32u : kKeyBits, s);
}

}  // namespace rta
