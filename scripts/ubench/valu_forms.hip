// valu_forms.hip — cycles per wave64 instruction per SIMD for individual instruction FORMS (inline asm, so the
// compiler cannot repack or vectorise them): is the 4-cycle cost of v_fma_f32 a property of FMA, of the VOP3
// encoding, or of three register sources?  8 waves per SIMD, 8 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM)                                                                         \
	__global__ void __launch_bounds__(64) NAME(float* out, int iters, float sb, float sc) {      \
		float a0 = threadIdx.x + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
		float vb = threadIdx.x * 0.5f + 1.0f, vc = threadIdx.x * 0.25f + 2.0f;                    \
		unsigned long long m = 0x5555555555555555ull;                                             \
		for (int it = 0; it < iters; it++) {                                                      \
			_Pragma("unroll") for (int r = 0; r < 16; r++) {                                      \
				asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
				             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
				             : "v"(vb), "v"(vc), "s"(sb), "s"(sc), "s"(m) : "vcc");                \
			}                                                                                     \
		}                                                                                         \
		out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
	}
// operands: %8 = vb, %9 = vc (VGPR), %10 = sb, %11 = sc (SGPR), %12 = mask (SGPR pair)
#define A_FMA_VVV(d) "v_fma_f32 " d ", " d ", %8, %9\n\t"
#define A_FMA_VSV(d) "v_fma_f32 " d ", " d ", %10, %9\n\t"
#define A_FMAC(d) "v_fmac_f32 " d ", %8, %9\n\t"
#define A_FMAC_S(d) "v_fmac_f32 " d ", %10, %9\n\t"
#define A_MUL_E32(d) "v_mul_f32_e32 " d ", %8, " d "\n\t"
#define A_MUL_S(d) "v_mul_f32_e32 " d ", %10, " d "\n\t"
#define A_ADD_E64(d) "v_add_f32_e64 " d ", " d ", %10\n\t"
#define A_MAX_E32(d) "v_max_f32_e32 " d ", %8, " d "\n\t"
#define A_MAX3(d) "v_max3_f32 " d ", " d ", %8, %9\n\t"
#define A_MED3(d) "v_med3_f32 " d ", " d ", %8, %9\n\t"
#define A_CNDMASK_VCC(d) "v_cndmask_b32_e32 " d ", " d ", %8, vcc\n\t"
#define A_CNDMASK_S(d) "v_cndmask_b32_e64 " d ", " d ", %8, %12\n\t"
#define A_CMP_VCC(d) "v_cmp_lt_f32_e32 vcc, %8, " d "\n\t"
#define A_CMP_S(d) "v_cmp_lt_f32_e64 vcc, " d ", %10\n\t"
#define A_SUB_E32(d) "v_sub_f32_e32 " d ", " d ", %8\n\t"
#define A_MOV(d) "v_mov_b32_e32 " d ", %8\n\t"
#define A_RCP(d) "v_rcp_f32_e32 " d ", " d "\n\t"
#define A_CVT_UB(d) "v_cvt_f32_ubyte1_e32 " d ", " d "\n\t"
KERNEL(k_fma_vvv, A_FMA_VVV) KERNEL(k_fma_vsv, A_FMA_VSV) KERNEL(k_fmac, A_FMAC) KERNEL(k_fmac_s, A_FMAC_S)
KERNEL(k_mul_e32, A_MUL_E32) KERNEL(k_mul_s, A_MUL_S) KERNEL(k_add_e64, A_ADD_E64) KERNEL(k_max_e32, A_MAX_E32)
KERNEL(k_max3, A_MAX3) KERNEL(k_med3, A_MED3) KERNEL(k_cnd_vcc, A_CNDMASK_VCC) KERNEL(k_cnd_s, A_CNDMASK_S)
KERNEL(k_cmp_vcc, A_CMP_VCC) KERNEL(k_cmp_s, A_CMP_S) KERNEL(k_sub, A_SUB_E32) KERNEL(k_mov, A_MOV) KERNEL(k_rcp, A_RCP) KERNEL(k_cvt, A_CVT_UB)
typedef float v2f __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, ASM)                                                                        \
	__global__ void __launch_bounds__(64) NAME(float* out, int iters, float sb, float sc) {      \
		v2f a0 = {threadIdx.x + 1.f, 2.f}, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
		v2f vb = {threadIdx.x * 0.5f + 1.0f, 3.f}, vc = {threadIdx.x * 0.25f + 2.0f, 5.f};        \
		v2f s2 = {sb, sc};                                                                        \
		for (int it = 0; it < iters; it++) {                                                      \
			_Pragma("unroll") for (int r = 0; r < 16; r++) {                                      \
				asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7") \
				             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
				             : "v"(vb), "v"(vc), "s"(s2));                                         \
			}                                                                                     \
		}                                                                                         \
		const v2f t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                      \
		out[blockIdx.x * 64 + threadIdx.x] = t.x + t.y;                                           \
	}
#define P_FMA_VVV(d) "v_pk_fma_f32 " d ", " d ", %8, %9\n\t"
#define P_FMA_SVV(d) "v_pk_fma_f32 " d ", %10, " d ", %9\n\t"
#define P_FMA_SVV_SEL(d) "v_pk_fma_f32 " d ", %10, " d ", %9 op_sel:[0,1,1] op_sel_hi:[1,1,1]\n\t"
#define P_FMA_SVV_SEL0(d) "v_pk_fma_f32 " d ", %10, " d ", %9 op_sel_hi:[1,0,0]\n\t"
#define P_MUL_VV(d) "v_pk_mul_f32 " d ", " d ", %8\n\t"
#define P_MUL_SV(d) "v_pk_mul_f32 " d ", %10, " d "\n\t"
#define P_ADD_SV(d) "v_pk_add_f32 " d ", %10, " d "\n\t"
KERNEL2(kp_fma_vvv, P_FMA_VVV) KERNEL2(kp_fma_svv, P_FMA_SVV) KERNEL2(kp_fma_sel, P_FMA_SVV_SEL) KERNEL2(kp_fma_sel0, P_FMA_SVV_SEL0)
KERNEL2(kp_mul_vv, P_MUL_VV) KERNEL2(kp_mul_sv, P_MUL_SV) KERNEL2(kp_add_sv, P_ADD_SV)
#define A_MOV_S(d) "v_mov_b32_e32 " d ", %10\n\t"
#define A_SUB_S(d) "v_sub_f32_e32 " d ", %10, " d "\n\t"
#define A_MIN_E32(d) "v_min_f32_e32 " d ", %8, " d "\n\t"
#define A_ADD_VV(d) "v_add_f32_e32 " d ", %8, " d "\n\t"
#define A_MUL_LIT(d) "v_mul_f32_e32 " d ", 0x3f800347, " d "\n\t"
#define A_MUL_INL(d) "v_mul_f32_e32 " d ", 0.5, " d "\n\t"
#define A_AND(d) "v_and_b32_e32 " d ", %8, " d "\n\t"
#define A_LSHL(d) "v_lshlrev_b32_e32 " d ", 1, " d "\n\t"
KERNEL(k_mov_s, A_MOV_S) KERNEL(k_sub_s, A_SUB_S) KERNEL(k_min, A_MIN_E32) KERNEL(k_add_vv, A_ADD_VV) KERNEL(k_mul_lit, A_MUL_LIT) KERNEL(k_mul_inl, A_MUL_INL)
KERNEL(k_and, A_AND) KERNEL(k_lshl, A_LSHL)
// op_sel semantics of v_pk_fma_f32 as rt_kernels.hip slab_hit_oct relies on them
__global__ void k_opsel_check(float* out, float s0, float s1) {
	const v2f sp = {s0, s1};
	const v2f a = {2.0f, 3.0f}, b = {10.0f, 20.0f};
	v2f lo, hi, mix;
	asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "s"(sp), "v"(a), "v"(b));                  // (s0*a.x+b.x, s1*a.x+b.x)
	asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(hi) : "s"(sp), "v"(a), "v"(b));   // (s0*a.y+b.y, s1*a.y+b.y)
	asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(mix) : "s"(sp), "v"(a), "v"(a));  // (s0*a.x+a.y, s1*a.x+a.y)
	if (threadIdx.x == 0) { out[0] = lo.x; out[1] = lo.y; out[2] = hi.x; out[3] = hi.y; out[4] = mix.x; out[5] = mix.y; }
}
typedef void (*kern)(float*, int, float, float);
int main() {
	float* d;
	if (hipMalloc(&d, 256 * 4 * 16 * 64 * sizeof(float)) != hipSuccess) return 1;
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
	printf("%d CUs, clock %d MHz\n", p.multiProcessorCount, p.clockRate / 1000);
	{
		hipLaunchKernelGGL(k_opsel_check, dim3(1), dim3(64), 0, 0, d, 5.0f, 7.0f);
		float h[6];
		if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
		const float want[6] = { 5 * 2 + 10.f, 7 * 2 + 10.f, 5 * 3 + 20.f, 7 * 3 + 20.f, 5 * 2 + 3.f, 7 * 2 + 3.f };
		bool ok = true;
		for (int i = 0; i < 6; i++) ok = ok && h[i] == want[i];
		printf("v_pk_fma_f32 op_sel check: %s (%g %g | %g %g | %g %g)\n", ok ? "PASS" : "FAIL", h[0], h[1], h[2], h[3], h[4], h[5]);
		if (!ok) return 2;
	}
	const int iters = 200, w = 8;
	struct { const char* name; kern k; } ks[] = {
		{ "v_fma_f32 v,v,v,v", k_fma_vvv }, { "v_fma_f32 v,v,s,v", k_fma_vsv }, { "v_fmac_f32 v,v,v", k_fmac }, { "v_fmac_f32 v,s,v", k_fmac_s },
		{ "v_mul_f32_e32 v,v,v", k_mul_e32 }, { "v_mul_f32_e32 v,s,v", k_mul_s }, { "v_add_f32_e64 v,v,s", k_add_e64 }, { "v_max_f32_e32", k_max_e32 },
		{ "v_max3_f32", k_max3 }, { "v_med3_f32", k_med3 }, { "v_cndmask_b32_e32 vcc", k_cnd_vcc }, { "v_cndmask_b32_e64 s[n:n+1]", k_cnd_s },
		{ "v_cmp_lt_f32_e32 vcc", k_cmp_vcc }, { "v_cmp_lt_f32_e64 vcc,v,s", k_cmp_s }, { "v_sub_f32_e32", k_sub }, { "v_mov_b32", k_mov },
		{ "v_rcp_f32", k_rcp }, { "v_cvt_f32_ubyte1", k_cvt },
		{ "v_mov_b32 v,s", k_mov_s }, { "v_sub_f32_e32 v,s,v", k_sub_s }, { "v_min_f32_e32 v,v,v", k_min }, { "v_add_f32_e32 v,v,v", k_add_vv },
		{ "v_mul_f32_e32 v,literal,v", k_mul_lit }, { "v_mul_f32_e32 v,0.5,v", k_mul_inl }, { "v_and_b32 v,v,v", k_and }, { "v_lshlrev_b32 v,1,v", k_lshl },
		{ "v_pk_fma_f32 v,v,v,v", kp_fma_vvv }, { "v_pk_fma_f32 v,s,v,v", kp_fma_svv }, { "v_pk_fma_f32 v,s,v,v sel hi", kp_fma_sel }, { "v_pk_fma_f32 v,s,v,v sel lo", kp_fma_sel0 },
		{ "v_pk_mul_f32 v,v,v", kp_mul_vv }, { "v_pk_mul_f32 v,s,v", kp_mul_sv }, { "v_pk_add_f32 v,s,v", kp_add_sv },
	};
	for (auto& e : ks) {
		hipEvent_t e0, e1;
		(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		const int blocks = p.multiProcessorCount * 4 * w;
		for (int rep = 0; rep < 2; rep++) {
			(void)hipEventRecord(e0);
			hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
			(void)hipEventRecord(e1);
			(void)hipEventSynchronize(e1);
		}
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		const double insts = (double)iters * 16 * 8 * w;
		printf("%-30s %.3f ms  %.2f cycles per wave64 instruction per SIMD (at the nominal clock)\n", e.name, ms, ms * 1e-3 * p.clockRate * 1e3 / insts);
	}
	return 0;
}
