// valu_rate.hip — how many cycles does a SIMD need per wave64 VALU instruction on this GPU?
// Each wave runs N dependent-free v_fma_f32 / v_cndmask / v_max3 (8 independent accumulators) in a loop;
// grid = 256 CUs x 4 SIMDs x W waves.  Prints cycles per instruction per SIMD for W = 1..8.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters) {
	float a[8];
	for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
	const float b = 1.0001f, c = 0.5f;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 16; r++)
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (KIND == 0) a[i] = __builtin_fmaf(a[i], b, c);
				else if (KIND == 1) a[i] = a[i] > c ? a[i] * b : a[i];           // cmp + cndmask + mul
				else a[i] = fmaxf(fmaxf(a[i], b), c) + 1.0f;                     // max3 + add
			}
	}
	float s = 0;
	for (int i = 0; i < 8; i++) s += a[i];
	out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
	float* d;
	hipMalloc(&d, 256 * 4 * 16 * 64 * sizeof(float));
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
	const int iters = 4000;
	for (int kind = 0; kind < 3; kind++)
		for (int w = 1; w <= 8; w *= 2) {
			hipEvent_t e0, e1;
			hipEventCreate(&e0); hipEventCreate(&e1);
			const int blocks = p.multiProcessorCount * 4 * w;
			for (int rep = 0; rep < 2; rep++) {
				hipEventRecord(e0);
				if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters);
				else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters);
				else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, iters);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
			}
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			const double insts_per_wave = (double)iters * 16 * 8 * (kind == 0 ? 1 : (kind == 1 ? 3 : 2));
			const double cycles = ms * 1e-3 * p.clockRate * 1e3;
			printf("kind %d (%s) waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD\n", kind,
			       kind == 0 ? "v_fma_f32" : kind == 1 ? "cmp+cndmask+mul" : "max3+add", w, ms, cycles / (insts_per_wave * w));
		}
	return 0;
}
