// What does the memory system deliver for scattered record reads out of a table far beyond the
// caches (1 GiB against 4 MiB of L2 per XCD and 256 MiB of Infinity Cache), by record size?
//
//   hipcc --offload-arch=gfx950 -O3 -o build/hbm_gather_probe tools/hbm_gather_probe.hip
//   build/hbm_gather_probe                                           (on the GPU box)
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- build/hbm_gather_probe   (calibrates the counter:
//                                                          the bytes each kernel asks for are known)
//
// Every lane reads `trips` pseudo-random records, 16 bytes per load instruction, the record aligned to
// its size: 32 B (two loads), 64 B (four: a node of the 4-wide tree), 128 B (eight).  The indices do not
// depend on what is read, so the loads of a lane overlap freely: this is the ceiling for a traversal's
// node reads, not a model of its dependent chain.  `stream` reads the same table front to back.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                  \
	do {                                                                                          \
		hipError_t e_ = (x);                                                                      \
		if (e_ != hipSuccess) {                                                                   \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                               \
			exit(1);                                                                              \
		}                                                                                         \
	} while (0)

__device__ inline uint32_t next_index(uint32_t &state, uint32_t mask)
{
	state = state * 1664525u + 1013904223u;
	return (state >> 4) & mask;
}

// QUADS: 16-byte loads per record
template <int QUADS>
__global__ __launch_bounds__(256) void k_gather(const float4 *table, uint32_t mask, int trips, float *out)
{
	uint32_t state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
	float acc = 0.f;
	for (int t = 0; t < trips; t++) {
		const float4 *r = table + (size_t)QUADS * next_index(state, mask);
#pragma unroll
		for (int q = 0; q < QUADS; q++) {
			const float4 a = r[q];
			acc += a.x + a.w;
		}
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_stream(const float4 *table, size_t quads, float *out)
{
	float acc = 0.f;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) {
		const float4 a = table[i];
		acc += a.x + a.w;
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <class Launch>
float timed(Launch launch)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	launch();
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, a, b));
	CHECK(hipEventDestroy(a));
	CHECK(hipEventDestroy(b));
	return ms;
}

int main()
{
	const size_t bytes = (size_t)1 << 30;
	const int trips = 64, blocks = 256 * 8 * 8;
	float4 *table = nullptr;
	float *out = nullptr;
	CHECK(hipMalloc((void **)&table, bytes));
	CHECK(hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(float)));
	CHECK(hipMemset(table, 0, bytes));
	const double records = (double)blocks * 256 * trips;
	float ms;
	ms = timed([&] { hipLaunchKernelGGL(k_stream, dim3(256 * 32), dim3(256), 0, 0, table, bytes / 16, out); });
	printf("stream 1 GiB          %8.3f ms  %7.1f GB/s\n", ms, bytes / (ms * 1e-3) / 1e9);
	ms = timed([&] { hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(bytes / 32 - 1), trips, out); });
	printf("gather  32-B records  %8.3f ms  %7.1f GB/s asked for  %6.2f G records/s\n", ms, records * 32 / (ms * 1e-3) / 1e9, records / (ms * 1e-3) / 1e9);
	ms = timed([&] { hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(bytes / 64 - 1), trips, out); });
	printf("gather  64-B records  %8.3f ms  %7.1f GB/s asked for  %6.2f G records/s\n", ms, records * 64 / (ms * 1e-3) / 1e9, records / (ms * 1e-3) / 1e9);
	ms = timed([&] { hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(bytes / 128 - 1), trips, out); });
	printf("gather 128-B records  %8.3f ms  %7.1f GB/s asked for  %6.2f G records/s\n", ms, records * 128 / (ms * 1e-3) / 1e9, records / (ms * 1e-3) / 1e9);
	printf("(each kernel ran twice; per launch the gathers ask for %.0f records)\n", records);
	CHECK(hipFree(table));
	CHECK(hipFree(out));
	return 0;
}
