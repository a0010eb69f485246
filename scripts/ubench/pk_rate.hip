// pk_rate.hip — does a packed binary32 instruction (v_pk_mul_f32 / v_pk_add_f32: two IEEE binary32 results per
// lane) issue at the rate of a plain v_mul_f32 on this GPU, also with an SGPR-pair operand and with op_sel
// broadcasting one half?  Prints cycles per wave-instruction per SIMD for 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, const float* in, int iters) {
	f2 a[8];
	for (int i = 0; i < 8; i++) a[i] = f2{ threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i };
	const f2 b = f2{ 1.0001f, 0.9999f };
	const float s0 = in[0], s1 = in[1];          // uniform: live in SGPRs
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 16; r++)
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (KIND == 0) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y)); }
				else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				else if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(f2{ s0, s1 }));
				else if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(b));
				else if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
				else if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
				else if (KIND == 6) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].x) : "s"(s0)); asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i].y) : "s"(s1)); }
				else if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "+v"(a[i]) : "s"(f2{ s0, s1 }), "v"(a[i]));
			}
	}
	f2 s = f2{ 0, 0 };
	for (int i = 0; i < 8; i++) s += a[i];
	out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}
template <int KIND> void run(const char* name, int per, float* d, float* in, const hipDeviceProp_t& p) {
	const int iters = 2000;
	for (int w = 1; w <= 8; w *= 2) {
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		const int blocks = p.multiProcessorCount * 4 * w;
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, in, iters);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
		}
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		const double insts = (double)iters * 16 * 8 * per, cycles = ms * 1e-3 * p.clockRate * 1e3;
		printf("%-44s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD, %.2f per binary32 result pair\n", name, w, ms,
		       cycles / (insts * w), cycles / ((double)iters * 16 * 8 * w));
	}
}
int main() {
	float *d, *in, h[2] = { 1.0001f, 0.9999f };
	hipMalloc(&d, 256 * 4 * 16 * 64 * sizeof(float)); hipMalloc(&in, 8);
	hipMemcpy(in, h, 8, hipMemcpyHostToDevice);
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
	run<0>("2 x v_mul_f32 (vgpr)", 2, d, in, p);
	run<6>("2 x v_mul_f32 (sgpr operand)", 2, d, in, p);
	run<1>("v_pk_mul_f32 (vgpr pairs)", 1, d, in, p);
	run<2>("v_pk_mul_f32 (sgpr pair operand)", 1, d, in, p);
	run<3>("v_pk_mul_f32 op_sel_hi:[1,0] (broadcast)", 1, d, in, p);
	run<7>("v_pk_mul_f32 sgpr pair x broadcast vgpr", 1, d, in, p);
	run<4>("v_pk_add_f32", 1, d, in, p);
	run<5>("v_pk_fma_f32", 1, d, in, p);
	return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
