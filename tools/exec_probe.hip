// Does a vector instruction cost less when few of its 64 lanes are active?
// Each wave runs a dependent chain of v_fma_f32 under different EXEC masks:
// all lanes, the first 32, the first 16, every fourth lane (16 lanes, one in
// every group of four), one lane.  Build: hipcc --offload-arch=gfx950 -O3 -o exec_probe tools/exec_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void k_chain(float *out, int iters, int pattern)
{
	const int lane = threadIdx.x & 63;
	bool on;
	switch (pattern) {
	case 0: on = true; break;
	case 1: on = lane < 32; break;
	case 2: on = lane < 16; break;
	case 3: on = (lane & 3) == 0; break;
	case 4: on = lane == 0; break;
	default: on = lane >= 16 && lane < 32; break;
	}
	float a = (float)threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f;
	if (on) {
		for (int i = 0; i < iters; i++) {
			// four independent chains, so that the ALU pipeline rather than its latency sets the pace
			a = __builtin_fmaf(a, b, c);
			d = __builtin_fmaf(d, b, c);
			e = __builtin_fmaf(e, b, c);
			f = __builtin_fmaf(f, b, c);
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = a + d + e + f;
}

int main()
{
	const int blocks = 256 * 8 * 4, iters = 20000;      // 32 waves per CU
	float *out;
	hipMalloc(&out, (size_t)blocks * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const char *names[] = {"64 lanes", "lanes 0-31", "lanes 0-15", "every 4th lane (16)", "1 lane", "lanes 16-31"};
	for (int p = 0; p < 6; p++) {
		hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, out, iters, p);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, out, iters, p);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		const double insts = (double)blocks * 4 * iters * 4;      // wave instructions
		printf("%-22s %8.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz, 1024 SIMDs)\n", names[p], ms,
		       ms * 1e-3 * 2.4e9 * 1024 / insts);
	}
	return 0;
}
