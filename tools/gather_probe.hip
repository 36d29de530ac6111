// What does a gather of 64-byte records cost on gfx950, per pattern?
//
//   hipcc --offload-arch=gfx950 -O3 -o build/gather_probe tools/gather_probe.hip
//   build/gather_probe            (on the GPU box)
//
// Every lane of every wave reads `trips` pseudo-random records of a table that
// stays in L2 (1 MiB, the size of a small scene's tree).  Patterns:
//   own4x16    four 16-byte loads of the lane's own record: 4 instructions, each
//              touching 64 different cache lines (what a traversal trip does)
//   own1x16    one 16-byte load of the lane's own record
//   own4x4     four 4-byte loads of the lane's own record
//   quad4x16   four 16-byte loads, the four lanes of a quad reading the four
//              quarters of ONE record (of quad lane k in load k): the same
//              bytes per lane and per wave as own4x16, 16 lines per instruction
//   none       the index arithmetic alone
// and the first two again with loads marked non-temporal and with smaller
// tables (16 KiB fits the CU's vector cache).
// Prints nanoseconds per wave and trip with the GPU full (8 workgroups of 256
// per CU), i.e. the time a CU's memory pipeline needs per trip / 32 waves.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
	do {                                                                                          \
		hipError_t e_ = (x);                                                                      \
		if (e_ != hipSuccess) {                                                                   \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                               \
			exit(1);                                                                              \
		}                                                                                         \
	} while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline uint32_t next_index(uint32_t &state, uint32_t mask)
{
	state = state * 1664525u + 1013904223u;
	return (state >> 8) & mask;
}

template <int PATTERN>
__global__ __launch_bounds__(256) void k_gather(const float4 *table, uint32_t mask, int trips, float *out)
{
	uint32_t state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
	float acc = 0.f;
	const uint32_t quarter = threadIdx.x & 3;
	for (int t = 0; t < trips; t++) {
		const uint32_t mine = next_index(state, mask);      // < number of records by construction
		if (PATTERN == 0) {
			const float4 *r = table + 4 * (size_t)mine;
			const float4 a = r[0], b = r[1], c = r[2], d = r[3];
			acc += a.x + b.y + c.z + d.w;
		} else if (PATTERN == 1) {
			const float4 a = table[4 * (size_t)mine];
			acc += a.x + a.w;
		} else if (PATTERN == 2) {
			const float *r = reinterpret_cast<const float *>(table + 4 * (size_t)mine);
			acc += r[0] + r[5] + r[10] + r[15];
		} else if (PATTERN == 3) {
			// all 64 lanes are active here, so every quad lane's index is defined
			float4 got[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t theirs = (uint32_t)__shfl((int)mine, (int)((threadIdx.x & 60) | k), 64);
				got[k] = table[4 * (size_t)theirs + quarter];
			}
			acc += got[0].x + got[1].y + got[2].z + got[3].w;
		} else if (PATTERN == 5) {
			const v4f *r = reinterpret_cast<const v4f *>(table + 4 * (size_t)mine);
			const v4f a = __builtin_nontemporal_load(r), b = __builtin_nontemporal_load(r + 1),
			          c = __builtin_nontemporal_load(r + 2), d = __builtin_nontemporal_load(r + 3);
			acc += a.x + b.y + c.z + d.w;
		} else if (PATTERN == 6) {
			const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(table + 4 * (size_t)mine));
			acc += a.x + a.w;
		} else {
			acc += (float)mine;
		}
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int PATTERN>
float run(const float4 *table, uint32_t mask, int trips, float *out, int blocks)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	hipLaunchKernelGGL(k_gather<PATTERN>, dim3(blocks), dim3(256), 0, 0, table, mask, trips, out);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	hipLaunchKernelGGL(k_gather<PATTERN>, dim3(blocks), dim3(256), 0, 0, table, mask, trips, out);
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
	const uint32_t records = 1u << 14;      // x 64 B = 1 MiB
	const int trips = 2000, blocks = 256 * 8;
	std::vector<float> host((size_t)records * 16);
	for (size_t i = 0; i < host.size(); i++)
		host[i] = (float)(i % 97) * 0.25f;
	float4 *table = nullptr;
	float *out = nullptr;
	CHECK(hipMalloc((void **)&table, host.size() * sizeof(float)));
	CHECK(hipMalloc((void **)&out, (size_t)blocks * 256 * sizeof(float)));
	CHECK(hipMemcpy(table, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
	const char *names[] = {"own4x16", "own1x16", "own4x4", "quad4x16", "none", "own4x16 nontemporal", "own1x16 nontemporal",
	                       "own4x16, 16 KiB table", "own1x16, 16 KiB table", "own4x16, 256 KiB table"};
	float ms[10];
	ms[0] = run<0>(table, records - 1, trips, out, blocks);
	ms[1] = run<1>(table, records - 1, trips, out, blocks);
	ms[2] = run<2>(table, records - 1, trips, out, blocks);
	ms[3] = run<3>(table, records - 1, trips, out, blocks);
	ms[4] = run<4>(table, records - 1, trips, out, blocks);
	ms[5] = run<5>(table, records - 1, trips, out, blocks);
	ms[6] = run<6>(table, records - 1, trips, out, blocks);
	ms[7] = run<0>(table, 255, trips, out, blocks);
	ms[8] = run<1>(table, 255, trips, out, blocks);
	ms[9] = run<0>(table, 4095, trips, out, blocks);
	// a CU holds 32 of the grid's waves at a time; per CU one trip of one wave takes
	const double wave_trips_per_cu = (double)blocks * 4 * trips / 256.0;
	for (int p = 0; p < 10; p++)
		printf("%-24s %8.3f ms   %7.1f ns per wave-trip per CU  (%5.1f cycles at 2.4 GHz)\n", names[p], ms[p],
		       ms[p] * 1e6 / wave_trips_per_cu, ms[p] * 1e6 / wave_trips_per_cu * 2.4);
	CHECK(hipFree(table));
	CHECK(hipFree(out));
	return 0;
}
