// scripts/ubench/div65025.hip — exhaustive check of the two cheaper forms of Source.cpp:488's
// `x / 255.0 / 255.0f` (binary64 divisions, result stored in a binary32) used by rt_kernels.hip div65025:
//   (a) one binary32 IEEE division x / 65025.0f            -> must equal the binary64 form for EVERY x
//   (b) the division-free form q' = fma(fma(-q, 65025, x), c, q), q = x * c, c = RN(1/65025)
// over all 2^32 binary32 bit patterns (NaN inputs excluded).  Prints the mismatch counts and, for (b),
// the smallest and largest |x| that mismatches.   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off div65025.hip -o div65025
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void check(unsigned long long* bad_a, unsigned long long* bad_b, unsigned int* lo_b, unsigned int* hi_b, unsigned long long* bad_b_mid) {
	const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
	unsigned long long na = 0, nb = 0, nmid = 0;
	for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
		const unsigned int bits = (unsigned int)i;
		const float x = __uint_as_float(bits);
		if (x != x) continue;
		const float ref = (float)(((double)x / 255.0) / (double)255.0f);
		const float a = x / 65025.0f;
		const float c = 1.0f / 65025.0f;
		const float q = x * c;
		const float b = __builtin_fmaf(__builtin_fmaf(-q, 65025.0f, x), c, q);
		if (__float_as_uint(a) != __float_as_uint(ref)) na++;
		if (__float_as_uint(b) != __float_as_uint(ref)) {
			nb++;
			const unsigned int mag = bits & 0x7fffffffu;
			atomicMin(lo_b, mag);
			atomicMax(hi_b, mag);
			const float ax = fabsf(x);
			if (ax >= 0x1p-100f && ax <= 0x1p100f) nmid++;
		}
	}
	if (na) atomicAdd(bad_a, na);
	if (nb) atomicAdd(bad_b, nb);
	if (nmid) atomicAdd(bad_b_mid, nmid);
}

int main() {
	unsigned long long *d, h[3] = { 0, 0, 0 };
	unsigned int *dl, hl[2] = { 0xffffffffu, 0u };
	hipMalloc(&d, sizeof h); hipMalloc(&dl, sizeof hl);
	hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(dl, hl, sizeof hl, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, d + 1, dl, dl + 1, d + 2);
	if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hl, dl, sizeof hl, hipMemcpyDeviceToHost);
	float flo, fhi; memcpy(&flo, &hl[0], 4); memcpy(&fhi, &hl[1], 4);
	printf("x/65025.0f vs binary64 form: %llu mismatches\n", h[0]);
	printf("division-free form vs binary64 form: %llu mismatches, |x| from %g to %g; %llu of them with |x| in [2^-100, 2^100]\n", h[1], flo, fhi, h[2]);
	return (h[0] == 0) ? 0 : 2;
}
