// scripts/ubench/recip_exact.hip — exhaustive check of a cheaper correctly rounded reciprocal for rt_kernels.hip (the light's
// 1/s and 1/(s*s), Source.cpp:488 and glm::normalize): r0 = v_rcp_f32(x) (1 ulp), e = fma(-x, r0, 1), r = fma(e, r0, r0)
// (Markstein) against the IEEE division 1.0f / x, over all 2^32 binary32 bit patterns (NaN excluded).  Prints the number of
// mismatches and the smallest and largest |x| that mismatches, overall and inside [2^-126, 2^126].
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off recip_exact.hip -o recip_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void check(unsigned long long* n_bad, unsigned int* lo, unsigned int* hi, unsigned long long* n_mid, unsigned int* lo_mid, unsigned int* hi_mid,
                      unsigned long long* n_bad2) {
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	unsigned long long nb = 0, nm = 0, nb2 = 0;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
		const unsigned int bits = (unsigned int)i;
		const float x = __uint_as_float(bits);
		if (x != x) continue;
		const float ref = 1.0f / x;
		const float r0 = __builtin_amdgcn_rcpf(x);
		const float e = __builtin_fmaf(-x, r0, 1.0f);
		const float r = __builtin_fmaf(e, r0, r0);
		const float e2 = __builtin_fmaf(-x, r, 1.0f);
		const float r2 = __builtin_fmaf(e2, r, r);
		const unsigned int mag = bits & 0x7fffffffu;
		const float ax = fabsf(x);
		const bool mid = ax >= 0x1p-126f && ax <= 0x1p126f;
		if (__float_as_uint(r) != __float_as_uint(ref)) {
			nb++;
			atomicMin(lo, mag); atomicMax(hi, mag);
			if (mid) { nm++; atomicMin(lo_mid, mag); atomicMax(hi_mid, mag); }
		}
		if (mid && __float_as_uint(r2) != __float_as_uint(ref)) nb2++;
	}
	if (nb) atomicAdd(n_bad, nb);
	if (nm) atomicAdd(n_mid, nm);
	if (nb2) atomicAdd(n_bad2, nb2);
}

int main() {
	unsigned long long *d, h[3] = { 0, 0, 0 };
	unsigned int *dl, hl[4] = { 0xffffffffu, 0u, 0xffffffffu, 0u };
	hipMalloc(&d, sizeof h); hipMalloc(&dl, sizeof hl);
	hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(dl, hl, sizeof hl, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, dl, dl + 1, d + 1, dl + 2, dl + 3, d + 2);
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hl, dl, sizeof hl, hipMemcpyDeviceToHost);
	float f[4]; memcpy(f, hl, sizeof f);
	printf("one step vs 1.0f / x: %llu mismatches, |x| from %g (0x%08x) to %g (0x%08x)\n", h[0], f[0], hl[0], f[1], hl[1]);
	printf("  with |x| in [2^-126, 2^126]: %llu, |x| from %g (0x%08x) to %g (0x%08x)\n", h[1], f[2], hl[2], f[3], hl[3]);
	printf("two steps, |x| in [2^-126, 2^126]: %llu mismatches\n", h[2]);
	return 0;
}
