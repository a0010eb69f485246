// atomic_min64.hip — how fast are 64-bit atomicMin updates of a per-pixel visibility buffer on this GPU?  (What a
// rasterising primary-visibility pass would do: one atomicMin of (depth bits : triangle index) per accepted
// triangle/pixel pair.)  Threads in a wave hit neighbouring pixels, as the pixels of one triangle would.
// hipcc --offload-arch=gfx950 -O2 atomic_min64.hip -o atomic_min64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int MODE>
__global__ void __launch_bounds__(256) k(u64* buf, unsigned int n_pix, unsigned int width, int per_thread, unsigned int seed) {
	const unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned int h = (tid >> 3) * 2654435761u + seed;          // 8 lanes share a "triangle": a 4x2 pixel patch somewhere
	for (int k = 0; k < per_thread; k++) {
		h = h * 1664525u + 1013904223u;
		const unsigned int base = (h >> 4) % n_pix;
		const unsigned int pix = (base + (tid & 3) + ((tid >> 2) & 1) * width) % n_pix;
		const u64 key = ((u64)(0x3f800000u + (h & 0xfffff)) << 32) | tid;
		if (MODE == 0) atomicMin(&buf[pix], key);
		else if (MODE == 1) { if (key < __hip_atomic_load(&buf[pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&buf[pix], key); }
		else buf[pix] = key;                                      // plain stores, for scale
	}
}
template <int MODE> void run(const char* name, u64* d, unsigned int n_pix, unsigned int width, long long total) {
	const int per = 16;
	const unsigned int threads = (unsigned int)(total / per);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		hipMemset(d, 0xff, (size_t)n_pix * 8);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<MODE>, dim3(threads / 256), dim3(256), 0, 0, d, n_pix, width, per, 12345u + rep);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
	}
	printf("%-28s %9u pixels, %lld updates: %.3f ms = %.1f G updates/s\n", name, n_pix, total, best, total / best / 1e6);
}
int main() {
	for (unsigned int side : { 800u, 2880u }) {
		const unsigned int n_pix = side * side;
		u64* d; hipMalloc(&d, (size_t)n_pix * 8);
		const long long total = 1ll << 24;
		run<0>("atomicMin u64", d, n_pix, side, total);
		run<1>("load + atomicMin if smaller", d, n_pix, side, total);
		run<2>("plain 8-byte store", d, n_pix, side, total);
		hipFree(d);
	}
	return 0;
}
