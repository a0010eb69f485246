// rt_kernels.hip — the per-pixel render loop of the reference (getPixelData Source.cpp:530-628,
// castRay Source.cpp:277-405, processLighting Source.cpp:460-494) as ONE fused gfx950 kernel.
//
// Two forms of the kernel, same arithmetic, same bits (MI355X-first, not the reference's call tree):
//   * render_lockstep_kernel — what every launch uses unless rt_scene_set_kernel asks otherwise.  One lane == one pixel; a
//     wavefront (64 lanes) takes one 8x8 pixel tile after the other and all its lanes stay in step: primary castRay -> (only if
//     a lane hit a mirror) the bounce loop of Source.cpp:572-580 on those lanes -> per light one shadow castRay and the sum of
//     Source.cpp:476-489 -> store.  Measured faster than, or equal to, the state machine on every scene tried, bounce depths
//     that diverge inside a tile included (profiles/r03_config3_study.txt: 61 of 64 lanes active on the depth-16 mirror scene).
//     Brute force (ACCEL 0/1): every lane visits the SAME box or triangle per loop trip, so the scene's float4 planes are read
//     through the scalar cache (s_load, operands in SGPRs).  Tree scenes: ACCEL 3 walks the binary tree once per wavefront for a
//     coherent tile of rays ("packets", cast_ray_packet: node records by scalar loads, the stack in one VGPR), ACCEL 2 per lane
//     through the 4-wide quantised tree (rt_lbvh_trav.inl); shadow rays of both go through the light's direction grid
//     (rt_lgrid_trav.inl) and the scene-spanning triangles kept out of the tree by direction masks (hoisted_by_mask);
//   * render_kernel — the state machine (RT_KERNEL_STATE_MACHINE, and the RT_SHADE_GLSL checking mode): the reference's
//     recursion-by-call flattened into per-lane stages with ONE closest-hit loop in the instruction stream; a lane holds "the ray
//     it is waiting on" (primary, bounce or shadow), and lanes whose pixel is finished are refilled with fresh pixels from the
//     wave's tile run (ballot + mbcnt prefix ranks).  It pays for that generality on every pass of its loop;
//   * one launch renders up to RT_MAX_FRAMES frames (camera poses) of the same scene.
//
// Numerics: this TU is compiled with -ffp-contract=off and uses only unfused binary32 + - * / sqrt
// in the reference's order (and binary64 at the two places the reference does): frames reproduce
// the reference's CPU frames bit for bit.  Everything that skips work (boxes, limits, the
// division-free prefilter) only ever skips; no filter decides a hit.
#include "rt_internal.h"
#include <type_traits>

#include <cstdlib>
#include <cmath>

namespace rt {

// Source.cpp:391 compares `t > 0.00001` with t promoted to double.  For a binary32 t this is the
// same predicate as t > EPS_BELOW where EPS_BELOW is the largest binary32 <= 0.00001 (double):
// (float)0.00001 = 0x3727C5AC = 9.99999974737875e-06 < 1e-5, so that float itself.
#define RT_EPS_BELOW 9.99999974737875163555145263671875e-06f
#define RT_PREFILTER_REL 9.5367431640625e-07f      /* 2^-20 */
static_assert((double)RT_EPS_BELOW <= 0.00001, "EPS_BELOW must not exceed the double literal");

// pixel tile of one wavefront pass: RT_TILE_W x RT_TILE_H = 64 lanes
#ifndef RT_TILE_W
#define RT_TILE_W 8
#endif
#define RT_TILE_H (64 / RT_TILE_W)

enum { ST_IDLE = 0, ST_PRIMARY = 1, ST_BOUNCE = 2, ST_SHADOW = 3 };
enum { CTR_PRIMARY = 0, CTR_BOUNCE, CTR_SHADOW, CTR_PLANE, CTR_POINT, CTR_EDGE, CTR_BOX, CTR_PXHIT, CTR_PXMIRROR, CTR_PXLIT, CTR_N };

__host__ __device__ __forceinline__ int lds_stride_of(int n_tris) { return (n_tris + RT_GROUP - 1) / RT_GROUP * RT_GROUP; }

struct Hit {
	float t;      // closest.distance, 1e9f when nothing was hit (Source.cpp:278)
	int idx;      // closest.cur as an index, -1 == NULL
	vec3 p;       // closest.point
};

// Wave-level event counts for tuning (RT_PROFILE builds only; scripts/prof_build.sh).
#ifdef RT_PROFILE
struct WaveProf { unsigned int v[10]; unsigned long long t[10]; unsigned long long last; unsigned long long* g; };
#define RT_PROF(p, k) ((p).v[k]++)
// segment stamp: adds the cycles since the previous stamp to segment k (diagnostic builds only)
#define RT_STAMP(p, k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
	__builtin_amdgcn_s_waitcnt(0xC07F); (p).t[k] += now_ - (p).last; (p).last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
struct WaveProf { };
#define RT_PROF(p, k) ((void)0)
#define RT_STAMP(p, k) ((void)0)
#endif
enum { WP_TRACE = 0, WP_GROUP, WP_INBOX, WP_TRI, WP_MAYBE, WP_CAND, WP_C0, WP_C1, WP_ACCEPT, WP_ITER };

struct LaneCounters {
	unsigned int plane, point, edge, box;
};

__device__ __forceinline__ vec3 xyz(float4 v) { return make3(v.x, v.y, v.z); }

// ---- castRay, brute force (Source.cpp:301-402) ---------------------------------------------
// `tri` is the global planes; the loop index is wave-uniform, so every lane addresses the same
// float4.  `live` masks lanes that hold no
// ray.  Wave-level early outs replace the reference's per-ray `continue`s: the edge terms are
// only evaluated when some lane of the wave still needs them, lane results are unaffected.
template <bool COUNT, typename TriPtr>
__device__ __forceinline__ Hit cast_ray_brute(TriPtr tri, int stride, int n, vec3 o, vec3 d, int skip,
                                              bool live, LaneCounters& lc) {
	Hit h;
	h.t = 1e9f;
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	for (int i = 0; i < n; i++) {
		const float4 N4 = tri[PL_N * stride + i];
		const float4 P14 = tri[PL_P1 * stride + i];
		const vec3 nn = xyz(N4), p1 = xyz(P14);
		const float t = dot(p1 - o, nn) / dot(d, nn);                       // Source.cpp:383
		const bool visit = live && (i != skip);                             // Source.cpp:302
		const bool fwd = visit && !(t < 0.0f);                              // Source.cpp:385
		const bool cand = fwd && (t < h.t) && (t > RT_EPS_BELOW);           // Source.cpp:391
		if (COUNT) { lc.plane += visit; lc.point += fwd; lc.edge += cand; }
		if (__any(cand)) {
			const vec3 I = make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);   // Source.cpp:389
			const float4 E0 = tri[PL_E0 * stride + i];
			const bool c0 = cand && (dot(nn, cross(xyz(E0), I - p1)) > 0.0f);    // Source.cpp:392
			if (COUNT) lc.edge += c0;
			if (__any(c0)) {
				const float4 P24 = tri[PL_P2 * stride + i];
				const float4 E1 = tri[PL_E1 * stride + i];
				const bool c1 = c0 && (dot(nn, cross(xyz(E1), I - xyz(P24))) > 0.0f);   // Source.cpp:393
				if (COUNT) lc.edge += c1;
				if (__any(c1)) {
					const float4 P34 = tri[PL_P3 * stride + i];
					const float4 E2 = tri[PL_E2 * stride + i];
					const bool c2 = c1 && (dot(nn, cross(xyz(E2), I - xyz(P34))) > 0.0f);   // Source.cpp:394
					if (c2) { h.t = t; h.idx = i; h.p = I; }                    // Source.cpp:396
				}
			}
		}
	}
	return h;
}

// ---- castRay, brute force, pruned but exact -----------------------------------------------------
// Same closest hit, bit for bit, with less work per triangle than the loop above (which stays as
// the instrumented reference-order form used for the work counters).  Three observations:
//  (1) A triangle can only be accepted if eps < t < closest.distance (Source.cpp:391).  Any
//      upper bound `lim` on the distance of a possible hit may stand in for the initial 1e9:
//      * no hit point lies outside the scene's bounding box, so t <= t_exit(box) (the box is
//        padded on the host by 1e-3 of the scene extent and the exit distance by 1e-4, orders of
//        magnitude above the rounding of I = o + t*d) — in a closed room this prunes every wall
//        plane behind the wall that is actually hit;
//      * a shadow ray only decides "is there a hit with t <= dist" (Source.cpp:476), so its
//        limit is the float after dist and it may stop at its first accepted hit.
//  (2) t > eps already implies !(t < 0) (Source.cpp:385), and the self-skip (Source.cpp:302) only
//      matters for a triangle that would otherwise be accepted, so both leave the per-triangle path.
//  (3) For primary rays the numerator dot(p1 - camera, n) of Source.cpp:383 is the same for every
//      pixel: it is computed once per workgroup (same operations, same bits) and read from LDS.
// Lane predicates as wave masks.  The conditions of Source.cpp:385-396 are kept as 64-bit masks in SGPRs — one v_cmp
// per comparison writing the mask, s_and/s_or to combine, s_cmp + branch when no lane is left — instead of per-lane
// bools: a compound bool handed to ballot() is materialised through a VGPR (v_cndmask + v_cmp_ne per ballot) and
// `a && b` guards b's arithmetic with an exec save/restore and a branch; with masks the arithmetic runs on every
// lane (it costs the same issue slot) and only the masks decide.  lane_in() turns a mask back into a lane condition
// for free (it becomes the select's own condition operand).
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask lanes(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool lane_in(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// The signs of the three edge terms of Source.cpp:392-394 without their 51 operations, where that is safe: with
// M_k = n x edge_k made on the host (rt_scene.cpp edge_filter_planes, which holds the error analysis) and w = I - p1,
// s_k = M_k . w - c_k has the sign of the reference's dot(n, cross(edge_k, I - p_k)) whenever |s_k| exceeds
// B = G |w|_1 + H, a bound on the rounding of both.  Returns the candidate lanes that pass all three tests for sure;
// `undecided` gets the candidates for which some term is inside its bound and no other term surely fails — those
// lanes (a hit point within ~2^-20 of the triangle's size of an edge line) take the reference's own arithmetic.
// M0.w = G, M1.w = c_1, M2.w = H.
__device__ __forceinline__ lanemask edge_filter(float4 M0, float4 M1, float4 M2, vec3 w, lanemask cand, lanemask& undecided) {
	const float w1 = (fabsf(w.x) + fabsf(w.y)) + fabsf(w.z);
	// below 1e-20 the bound is raised to it (underflow cannot add that much); above 1e20, or NaN, nothing is decided
	// (a bound that is NaN or infinite — a ray or a triangle that is not finite — fails the comparison with 1e20 before fmaxf drops it;
	// within the guard every partial product is below 1e27: no term can be NaN)
	const float B0 = __builtin_fmaf(M0.w, w1, M2.w);
	const float B = fmaxf(B0, 1e-20f);
	const float s0 = __builtin_fmaf(M0.x, w.x, __builtin_fmaf(M0.y, w.y, M0.z * w.z));
	const float s1 = __builtin_fmaf(M1.x, w.x, __builtin_fmaf(M1.y, w.y, __builtin_fmaf(M1.z, w.z, -M1.w)));
	const float s2 = __builtin_fmaf(M2.x, w.x, __builtin_fmaf(M2.y, w.y, M2.z * w.z));
	const float least = fminf(fminf(s0, s1), s2);
	const lanemask passes = lanes(least > B);                  // every term surely positive
	const lanemask fails = lanes(-least > B);                  // the smallest term surely negative: some test of Source.cpp:392-394 fails
	undecided = cand & ~((fails | passes) & lanes(B0 < 1e20f));
	return cand & passes & ~undecided;
}

// Edge tests and acceptance for ONE triangle whose exact t is known (Source.cpp:389-396).
template <typename TriPtr>
__device__ __forceinline__ void accept_triangle(TriPtr tri, int stride, int i, vec3 nn, vec3 p1, vec3 o, vec3 d, float t,
                                                lanemask cand, int skip, bool first_only, Hit& h, float& lim, float& lim_hi, WaveProf& wp) {
	const vec3 I = make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);   // Source.cpp:389
	const vec3 w = I - p1;
	lanemask open;
	lanemask c2 = edge_filter(tri[PL_M0 * stride + i], tri[PL_M1 * stride + i], tri[PL_M2 * stride + i], w, cand, open);
	if (open != 0) {                                                        // some lane too close to an edge line to call: the reference's arithmetic
		const float4 E0 = tri[PL_E0 * stride + i];
		const lanemask c0 = open & lanes(dot(nn, cross(xyz(E0), w)) > 0.0f);            // Source.cpp:392
		RT_PROF(wp, WP_C0);
		if (c0 != 0) {
			const float4 P24 = tri[PL_P2 * stride + i];
			const float4 E1 = tri[PL_E1 * stride + i];
			const lanemask c1 = c0 & lanes(dot(nn, cross(xyz(E1), I - xyz(P24))) > 0.0f);   // Source.cpp:393
			RT_PROF(wp, WP_C1);
			if (c1 != 0) {
				const float4 P34 = tri[PL_P3 * stride + i];
				const float4 E2 = tri[PL_E2 * stride + i];
				c2 |= c1 & lanes(dot(nn, cross(xyz(E2), I - xyz(P34))) > 0.0f);         // Source.cpp:394
			}
		}
	}
	c2 &= lanes(i != skip);                                                 // Source.cpp:302
	if (c2 == 0) return;
	RT_PROF(wp, WP_ACCEPT);
	if (lane_in(c2)) {                                                      // Source.cpp:396
		h.t = t; h.idx = i; h.p = I;
		lim = first_only ? -1.0f : t;
		lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	}
}

// Ray / padded-box slab test with precomputed inv = 1/d and oi = -o/d: does the ray meet the box
// within [0, lim_hi]?  v_min/v_max drop NaN operands (0 * inf on a slab the ray lies in): that axis
// then does not constrain, which errs on the side of "hit".  Rays whose boxes are not to be
// trusted (far_guard) come in with inv = oi = 0: every box then tests as hit.
__device__ __forceinline__ bool slab_hit(float4 lo, float4 hi, vec3 inv, vec3 oi, float lim_hi) {
	const float x0 = __builtin_fmaf(lo.x, inv.x, oi.x), x1 = __builtin_fmaf(hi.x, inv.x, oi.x);
	const float y0 = __builtin_fmaf(lo.y, inv.y, oi.y), y1 = __builtin_fmaf(hi.y, inv.y, oi.y);
	const float z0 = __builtin_fmaf(lo.z, inv.z, oi.z), z1 = __builtin_fmaf(hi.z, inv.z, oi.z);
	const float tmin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
	const float tmax = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), lim_hi));
	return !(tmin > tmax);
}

// The loop itself, over clusters of groups of RT_GROUP consecutive triangles.
//  (4) Each group has a padded bounding box, and runs of groups whose common box is small against
//      the scene form a cluster with a box of its own (host: flatten_planes).  A hit point lies in
//      its triangle's box up to rounding that the pad exceeds by orders of magnitude, so a group
//      (cluster) whose box no ray of the wave meets within [0, lim] is skipped as a whole: one slab
//      test instead of RT_GROUP plane tests plus the edge terms of every plane the rays happen to
//      cross (the infinite planes of a small object are crossed by most rays of a frame, its box
//      by few).
//  (5) Per triangle the wave first forms ta = num * rcp(den): v_rcp_f32 is within 1 ulp, so ta is
//      within 3 * 2^-24 of the correctly rounded quotient t whenever rcp(den) is a normal number.
//      A triangle with ta outside [eps*(1 - 2^-20), lim*(1 + 2^-20)] cannot satisfy eps < t < lim
//      and is dropped without the IEEE division; everything else (including NaN and any den whose
//      reciprocal is not a normal float) goes on to the exact t = num / den and the reference's
//      own comparisons.
// Neither filter ever decides a hit; both only skip work, and NaNs always fall through to the
// exact path.  (Measured alternatives, kept out: testing all boxes branch-free into a wave mask
// first, and a straight-line exact path per group, shorten a lone wave's latency but cost
// instructions and registers; at steady state they are 10-20 % slower than this form.)
// Wave-uniform scene data is read through the scalar cache (constant address space: the loads
// become s_load_*, their results live in SGPRs and feed the VALU as scalar operands); the scene is
// read-only for the lifetime of a launch.
typedef float v4f __attribute__((ext_vector_type(4)));
// A wave-uniform 64-bit base plus a 32-bit byte offset: the form `s_load_dwordx4 sdst, sbase, soffset` takes, so
// indexing costs one scalar shift/add per load instead of the four of a 64-bit address (sign extension, 64-bit
// shift, add, add-with-carry).  Every buffer addressed this way is below 4 GiB (checked at commit).
struct c4ptr {
	const __attribute__((address_space(4))) char* p;
	unsigned int off;
	__device__ __forceinline__ float4 operator[](unsigned int i) const {
		const v4f v = *(const __attribute__((address_space(4))) v4f*)(p + (off + i * 16u));
		return make_float4(v.x, v.y, v.z, v.w);
	}
	__device__ __forceinline__ c4ptr offset_floats(unsigned int f) const { c4ptr r; r.p = p; r.off = off + f * 4u; return r; }
	// entries i .. i+3 (a 64-byte record) in one s_load_dwordx16
	__device__ __forceinline__ void quad(unsigned int i, float4& a, float4& b, float4& c, float4& d) const {
		typedef float v16f_ __attribute__((ext_vector_type(16)));
		const v16f_ v = *(const __attribute__((address_space(4))) v16f_*)(p + (off + i * 16u));
		a = make_float4(v[0], v[1], v[2], v[3]);
		b = make_float4(v[4], v[5], v[6], v[7]);
		c = make_float4(v[8], v[9], v[10], v[11]);
		d = make_float4(v[12], v[13], v[14], v[15]);
	}
	// entries i and i+1 (i even: a 32-byte record) in one s_load_dwordx8
	__device__ __forceinline__ void pair(unsigned int i, float4& a, float4& b) const {
		typedef float v8f_ __attribute__((ext_vector_type(8)));
		const v8f_ v = *(const __attribute__((address_space(4))) v8f_*)(p + (off + i * 16u));
		a = make_float4(v[0], v[1], v[2], v[3]);
		b = make_float4(v[4], v[5], v[6], v[7]);
	}
};
__device__ __forceinline__ c4ptr c4ptr_of(const void* base) {
	c4ptr r;
	r.p = (const __attribute__((address_space(4))) char*)(unsigned long long)base;
	r.off = 0u;
	return r;
}

// max(x0, y0, z0, 0) and min(x1, y1, z1, lim) of a slab test as four instructions in ONE asm block: written with
// fmaxf/fminf the compiler puts a canonicalising v_max_f32 x, x, x in front of every use of an operand it cannot prove
// quiet (the loop-carried limit) — one instruction of twelve in every box test — and around separate asm statements it
// pads with s_nops.  NaN operands are dropped exactly as by fmaxf/fminf (minNum/maxNum, the instructions' own behaviour).
__device__ __forceinline__ bool slab_interval_hit(float x0, float y0, float z0, float x1, float y1, float z1, float lim) {
	float tmin, tmax;
	asm("v_max_f32 %0, 0, %4\n\tv_max3_f32 %0, %2, %3, %0\n\tv_min_f32 %1, %7, %8\n\tv_min3_f32 %1, %5, %6, %1"
	    : "=&v"(tmin), "=&v"(tmax) : "v"(x0), "v"(y0), "v"(z0), "v"(x1), "v"(y1), "v"(z1), "v"(lim));
	return !(tmin > tmax);
}

// Slab test against an octant-ordered box record (near.xyz, far.xyz): for a wave whose rays all
// point into the same octant the near/far faces are known per axis, so the per-axis min/max of
// slab_hit falls away: 6 fma + max3/max + min3/min + compare.
__device__ __forceinline__ bool slab_hit_oct(float4 a, float4 b, vec3 inv, vec3 oi, float lim_hi) {
	return slab_interval_hit(__builtin_fmaf(a.x, inv.x, oi.x), __builtin_fmaf(a.y, inv.y, oi.y), __builtin_fmaf(a.z, inv.z, oi.z),
	                         __builtin_fmaf(a.w, inv.x, oi.x), __builtin_fmaf(b.x, inv.y, oi.y), __builtin_fmaf(b.y, inv.z, oi.z), lim_hi);
}

// The same test on a packet record (rt_lbvh.hip k_packet: near.x near.y far.x far.y | near.z far.z ref ref), the six
// multiply-adds as three packed ones (v_pk_fma_f32: two lanes' worth of fma per issue slot, same results): ixy = (inv.x, inv.y),
// izz = (inv.z, inv.z), likewise oxy, ozz.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool slab_hit_pk(float4 a, float4 b, f2v ixy, f2v izz, f2v oxy, f2v ozz, float lim_hi) {
	const f2v nxy = { a.x, a.y }, fxy = { a.z, a.w }, zz = { b.x, b.y };
	const f2v n = __builtin_elementwise_fma(nxy, ixy, oxy), f = __builtin_elementwise_fma(fxy, ixy, oxy), z = __builtin_elementwise_fma(zz, izz, ozz);
	return slab_interval_hit(n.x, n.y, z.x, f.x, f.y, z.y, lim_hi);
}

// OCT: every live ray of the wave points into octant `oct` (bit k = component k negative).
template <bool PRIMARY, bool OCT>
__device__ __forceinline__ Hit cast_ray_pruned(const DevScene& sc, c4ptr tri, int oct, const float* num0, vec3 o, vec3 d,
                                               vec3 inv, vec3 oi, int skip, float lim, bool first_only, WaveProf& wp) {
	RT_PROF(wp, WP_TRACE);
	RT_STAMP(wp, 7);        // bounds + slab set-up (since the last outer stamp)
	const int stride = sc.stride, n_clusters = sc.n_clusters;
	Hit h;
	h.t = 1e9f;
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	float lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	const float eps_lo = RT_EPS_BELOW * (1.0f - RT_PREFILTER_REL);
	const c4ptr cbox = tri.offset_floats((unsigned int)sc.oct_clusters + (unsigned int)oct * (unsigned int)sc.oct_cluster_pitch * 8u);
	const c4ptr gbox = tri.offset_floats((unsigned int)sc.oct_groups + (unsigned int)oct * (unsigned int)sc.oct_group_pitch * 8u);
	c4ptr crec = cbox;                                   // (running byte offsets: one scalar add per record)
	for (int c = 0; c < n_clusters; c++, crec.off += 32u) {
		// one record per cluster: its box and its run of groups.  A cluster of one group has that
		// group's box, so its test stands for the group's and the group record is never fetched.
		int g0, g1;
		RT_PROF(wp, WP_GROUP);
		if constexpr (OCT) {
			float4 ca, cb;
			crec.pair(0, ca, cb);
			g0 = __float_as_int(cb.z);
			g1 = g0 + __float_as_int(cb.w);
			if (__builtin_amdgcn_ballot_w64(slab_hit_oct(ca, cb, inv, oi, lim_hi)) == 0) continue;
		} else {
			const float4 clo = tri[PL_CLO * stride + c];
			const float4 chi = tri[PL_CHI * stride + c];
			g0 = __float_as_int(clo.w);
			g1 = g0 + __float_as_int(chi.w);
			if (__builtin_amdgcn_ballot_w64(slab_hit(clo, chi, inv, oi, lim_hi)) == 0) continue;
		}
		const bool single = g1 - g0 == 1;
		c4ptr grec = gbox.offset_floats((unsigned int)g0 * 8u);
		for (int g = g0; g < g1; g++, grec.off += 32u) {
			if (!single) {
				RT_PROF(wp, WP_GROUP);
				if constexpr (OCT) {
					float4 ga, gb;
					grec.pair(0, ga, gb);
					if (__builtin_amdgcn_ballot_w64(slab_hit_oct(ga, gb, inv, oi, lim_hi)) == 0) continue;
				} else {
					if (__builtin_amdgcn_ballot_w64(slab_hit(tri[PL_GLO * stride + g], tri[PL_GHI * stride + g], inv, oi, lim_hi)) == 0) continue;
				}
			}
			RT_PROF(wp, WP_INBOX);
#pragma unroll
			for (int k = 0; k < RT_GROUP; k++) {
				const int i = g * RT_GROUP + k;
				// (no `i >= n` test: the slots of a last, incomplete group are zero planes — den = 0, t = NaN, never a candidate)
				RT_PROF(wp, WP_TRI);
				const vec3 nn = xyz(tri[PL_N * stride + i]);
				vec3 p1 = make3(0.f, 0.f, 0.f);
				float num;
				if constexpr (PRIMARY) {
					num = num0[i];
				} else {
					p1 = xyz(tri[PL_P1 * stride + i]);
					num = dot(p1 - o, nn);
				}
				const float den = dot(d, nn);
				// eps_lo <= num/den <= lim_hi  <=>  num lies between eps_lo*den and lim_hi*den (either order)
				const float qa = lim_hi * den, qb = eps_lo * den;
				// +-normal qb: both products are accurate to 2^-24; anything else goes on to the exact test
				const lanemask maybe = (lanes(true) & ~lanes(__builtin_amdgcn_classf(qb, 0x108))) | lanes(__builtin_amdgcn_fmed3f(num, qa, qb) == num);
				if (maybe == 0) continue;
				RT_PROF(wp, WP_MAYBE);
				const float t = num / den;                                      // Source.cpp:383
				const lanemask cand = maybe & lanes(t > RT_EPS_BELOW) & lanes(t < lim);     // Source.cpp:385, 391
				if (cand == 0) continue;
				RT_PROF(wp, WP_CAND);
				if constexpr (PRIMARY) p1 = xyz(tri[PL_P1 * stride + i]);
				accept_triangle(tri, stride, i, nn, p1, o, d, t, cand, skip, first_only, h, lim, lim_hi, wp);
			}
		}
	}
	return h;
}

// ---- castRay over the tree, wave-uniform ("packet") ----------------------------------------------
// For a wave whose live rays share an octant the binary tree is walked ONCE for the whole wave:
// one 64-byte scalar load brings both children of a node, already ordered near-first for the
// octant with near/far faces per axis (rt_lbvh.hip k_packet); a child no ray of the wave meets
// within its limit is dropped, a leaf's triangle is tested with the reference's arithmetic, the
// farther inner child waits on a wave-uniform stack kept in the lanes of one VGPR.  No per-lane
// stack, no sorting, no divergence, every load scalar.  The order of visits is not the index order,
// so ties go to the lower index explicitly (as in rt_lbvh_trav.inl).
// COUNT: the work counters of SURVEY 8d for the lanes whose ray is still searching (lim >= 0) — what
// the walk costs each ray it carries, in the units the per-lane traversal counts for itself.
// How a trace handles the triangles kept out of the tree: behind their boxes, or — by_mask — by the lanes' direction masks hm
// (hoisted_by_mask).
struct HoistCtx {
	unsigned int hm = 0xffffffffu;
	bool by_mask = false;
};
template <bool COUNT, typename Rec>
__device__ __forceinline__ void packet_leaf(Rec T, int i, vec3 o, vec3 d, int skip, bool first_only,
                                            float eps_lo, Hit& h, float& lim, float& lim_hi, LaneCounters& lc, WaveProf& wp) {
	RT_PROF(wp, WP_TRI);
	const float4 N4 = T[0], P14 = T[1];
	const vec3 nn = xyz(N4), p1 = xyz(P14);
	const float num = dot(p1 - o, nn);
	const float den = dot(d, nn);
	// (no division-free prefilter here: nine of ten triangles whose box a packet reaches are candidates)
	const float t = num / den;                                              // Source.cpp:383
	const lanemask better = lanes(t < lim) | (lanes(t == h.t) & lanes(i < h.idx));   // Source.cpp:391 + index order
	const lanemask cand = lanes(t > RT_EPS_BELOW) & better;
	if constexpr (COUNT) { const bool act = lim >= 0.0f && i != skip; lc.plane += act; lc.point += act && !(t < 0.0f); lc.edge += act && lane_in(cand); }
	if (cand == 0) return;
	RT_PROF(wp, WP_CAND);
	const vec3 I = make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);      // Source.cpp:389
	const vec3 w = I - p1;
	lanemask open = cand, c2 = 0;
	if constexpr (!COUNT) c2 = edge_filter(T[2], T[3], T[4], w, cand, open);       // (the counting kernel takes the reference's steps one by one)
	if (open != 0) {
		const float4 P24 = T[5], P34 = T[6], E0 = T[7], E1 = T[8], E2 = T[9];
		const lanemask c0 = open & lanes(dot(nn, cross(xyz(E0), w)) > 0.0f);            // Source.cpp:392
		if constexpr (COUNT) lc.edge += lane_in(c0) && i != skip;
		if (c0 != 0) {
			const lanemask c1 = c0 & lanes(dot(nn, cross(xyz(E1), I - xyz(P24))) > 0.0f);   // Source.cpp:393
			if constexpr (COUNT) lc.edge += lane_in(c1) && i != skip;
			if (c1 != 0) c2 |= c1 & lanes(dot(nn, cross(xyz(E2), I - xyz(P34))) > 0.0f);   // Source.cpp:394
		}
	}
	c2 &= lanes(i != skip);                                                 // Source.cpp:302
	if (c2 == 0) return;
	if (lane_in(c2)) {                                                      // Source.cpp:396
		h.t = t; h.idx = i; h.p = I;
		lim = first_only ? -1.0f : t;
		lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	}
}

// v_writelane with a wave-uniform value and lane index (both SGPRs: the lane index goes through m0)
__device__ __forceinline__ void wave_set_lane(int& v, int value, int lane) {
	int m0_saved;
	asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
	             : "+v"(v), "=&s"(m0_saved) : "s"(value), "s"(lane));
}

// The triangles kept out of the tree (DevScene::hoist) for rays whose direction cell says which of them they can meet: `hm` is the
// lane's bit mask over the hoisted list (primary rays: FrameParams::hoist_mask, by the pixel's coarse cell of the view; rt_vis.inl
// hoist_mask_kernel).  No box tests and no box records: the lanes of a tile nearly always share one mask — take one lane's mask as
// a scalar, test the triangles of its set bits on the whole wave (the reference's arithmetic, ties to the lower index), repeat
// with what other lanes still hold.  Lanes with lim < 0 are done.
template <bool COUNT>
__device__ __forceinline__ void hoisted_by_mask(const DevScene& sc, unsigned int hm, vec3 o, vec3 d, int skip, bool first_only,
                                                Hit& h, float& lim, float& lim_hi, LaneCounters& lc, WaveProf& wp) {
	const c4ptr hrec = c4ptr_of(sc.hoist);
	const float eps_lo = RT_EPS_BELOW * (1.0f - RT_PREFILTER_REL);
	unsigned int left = lim >= 0.0f ? hm : 0u;
	for (;;) {
		const unsigned long long any = __builtin_amdgcn_ballot_w64(left != 0u);
		if (any == 0ull) break;
		unsigned int m = __builtin_amdgcn_readlane(left, __builtin_ctzll(any));
		left &= ~m;
		while (m != 0u) {
			const int j = __builtin_ctz(m);
			m &= m - 1u;
			if (j >= sc.hoist_count) break;
			// (the triangle's index comes with the record's first 32 bytes — rt_lbvh.hip k_hoist — which the test needs anyway: a separate
			// load of it was one more round trip through the scalar cache in front of every test)
			const c4ptr R = hrec.offset_floats((unsigned int)j * (RT_AOS_F4 * 4u));
			float4 N4, P14;
			R.pair(0, N4, P14);
			const int i = __float_as_int(P14.w);
			const bool want = lim >= 0.0f && ((hm >> j) & 1u) != 0u && i != skip;
			if (__builtin_amdgcn_ballot_w64(want) == 0ull) continue;
			RT_PROF(wp, WP_INBOX);
			packet_leaf<COUNT>(R, i, o, d, skip, first_only, eps_lo, h, lim, lim_hi, lc, wp);
		}
	}
}

// `by_mask` (wave-uniform): the hoisted triangles by the lanes' direction masks `hm` instead of behind their boxes
template <bool COUNT>
__device__ __forceinline__ Hit cast_ray_packet(const DevScene& sc, int oct, vec3 o, vec3 d, vec3 inv, vec3 oi,
                                               int skip, float lim, bool first_only, LaneCounters& lc, WaveProf& wp, HoistCtx hc = HoistCtx()) {
	Hit h;
	h.t = 1e9f;
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	float lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	const float eps_lo = RT_EPS_BELOW * (1.0f - RT_PREFILTER_REL);
	const c4ptr rec = c4ptr_of(sc.skiplist).offset_floats((unsigned int)oct * (unsigned int)sc.skip_count * 16u);
	const c4ptr aos = c4ptr_of(sc.tri_aos);
#ifdef RT_PROFILE
	int steps_ = 0;
#endif
	const f2v ixy = { inv.x, inv.y }, izz = { inv.z, inv.z }, oxy = { oi.x, oi.y }, ozz = { oi.z, oi.z };
	int stackv = 0;          // wave-uniform stack: entry k lives in lane k (a Karras tree over 63 key bits is at most 63 deep)
	int sp = 0;
	int node = 0;
#ifdef RT_EXPERIMENT
	if (!(sc.debug & 32))                                                     // (attribution: no tree walk, hoisted triangles only)
#endif
	for (;;) {
		float4 q0, q1, q2, q3;
		rec.quad((unsigned int)node * 4u, q0, q1, q2, q3);
		const int refA = __float_as_int(q1.z), refB = __float_as_int(q1.w);
		RT_PROF(wp, WP_GROUP);
		if constexpr (COUNT) lc.box += lim >= 0.0f;
		const bool hitA = __builtin_amdgcn_ballot_w64(slab_hit_pk(q0, q1, ixy, izz, oxy, ozz, lim_hi)) != 0;
		int next = -1;
		if (hitA) {
			if (refA < 0) packet_leaf<COUNT>(aos.offset_floats((unsigned int)(~refA) * (RT_AOS_F4 * 4u)), ~refA, o, d, skip, first_only, eps_lo, h, lim, lim_hi, lc, wp);
			else next = refA;
		}
		// the farther child is tested after a near LEAF has had its chance to shorten the limits
		if constexpr (COUNT) lc.box += lim >= 0.0f;
		const bool hitB = __builtin_amdgcn_ballot_w64(slab_hit_pk(q2, q3, ixy, izz, oxy, ozz, lim_hi)) != 0;
		if (hitB) {
			if (refB < 0) packet_leaf<COUNT>(aos.offset_floats((unsigned int)(~refB) * (RT_AOS_F4 * 4u)), ~refB, o, d, skip, first_only, eps_lo, h, lim, lim_hi, lc, wp);
			else if (next >= 0) wave_set_lane(stackv, refB, sp++);
			else next = refB;
		}
		if (next < 0) {
			if (sp == 0) break;
			sp--;
			next = __builtin_amdgcn_readlane(stackv, sp);
		}
		node = next;
#ifdef RT_PROFILE
		steps_++;
#endif
	}
	// The triangles kept out of the tree (DevScene::hoist), each behind its octant-ordered box — AFTER the walk: they are the
	// scene-spanning ones (the walls of the reference's room), mostly behind what the tree holds, and with the limits the walk
	// leaves their boxes are only met where a ray still has to reach them.
#ifdef RT_EXPERIMENT
	if (sc.debug & 16) return h;
#endif
	if (hc.by_mask) {
		if (sc.hoist_count > 0) hoisted_by_mask<COUNT>(sc, hc.hm, o, d, skip, first_only, h, lim, lim_hi, lc, wp);
	} else if (sc.hoist_count > 0 && __builtin_amdgcn_ballot_w64(lim >= 0.0f) != 0) {
		const c4ptr hrec = c4ptr_of(sc.hoist);
		const c4ptr hbox = hrec.offset_floats((unsigned int)sc.hoist_count * (RT_AOS_F4 * 4u) + (unsigned int)oct * (unsigned int)sc.hoist_count * 8u);
		for (int j = 0; j < sc.hoist_count;) {
			const c4ptr R = hbox.offset_floats((unsigned int)j * 8u);
			const float4 a0 = R[0], a1 = R[1];
			const int second = __float_as_int(a1.z);         // a partner with the same box follows (rt_lbvh.hip k_hoist), or -1
			RT_PROF(wp, WP_INBOX);
			if constexpr (COUNT) lc.box += lim >= 0.0f;
			if (__builtin_amdgcn_ballot_w64(slab_hit_oct(a0, a1, inv, oi, lim_hi)) != 0) {
				packet_leaf<COUNT>(hrec.offset_floats((unsigned int)j * (RT_AOS_F4 * 4u)), __float_as_int(a1.w), o, d, skip, first_only, eps_lo, h, lim, lim_hi, lc, wp);
				if (second >= 0) packet_leaf<COUNT>(hrec.offset_floats((unsigned int)(j + 1) * (RT_AOS_F4 * 4u)), second, o, d, skip, first_only, eps_lo, h, lim, lim_hi, lc, wp);
			}
			j += second >= 0 ? 2 : 1;
		}
	}
#ifdef RT_PROFILE
	if (wp.g && __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)) == (int)__lane_id()) {   // histogram of node steps per walk: counters 20..27
		const int b = steps_ < 16 ? 0 : steps_ < 32 ? 1 : steps_ < 48 ? 2 : steps_ < 64 ? 3 : steps_ < 96 ? 4 : steps_ < 128 ? 5 : steps_ < 192 ? 6 : 7;
		atomicAdd(&wp.g[20 + b], 1ull);
	}
#endif
	return h;
}

// ---- the triangles kept out of the tree (DevScene::hoist), for the per-lane traversal ------------
// Tested for every ray before the walk, from scalar loads, in index order with the reference's own
// strict comparison (Source.cpp:391), each behind a test of its own padded box (skipped when no ray
// of the wave meets it); the walk then starts from the hit and limit they leave.
template <bool COUNT>
__device__ __forceinline__ void hoisted_tests(const DevScene& sc, vec3 o, vec3 d, vec3 inv, vec3 oi, int skip, bool live, bool first_only,
                                              Hit& h, float& lim, float& lim_hi, LaneCounters& lc, WaveProf& wp) {
	c4ptr T = c4ptr_of(sc.hoist);
	// The lanes of a tile nearly always point into one octant: then that octant's copy of the boxes (near and far faces
	// known per axis, as the packet walk reads them) saves the per-axis min/max of the general slab test; a wave of mixed
	// octants reads the copy of octant 0 (near = lo, far = hi) with the general test.
	const unsigned int lane_oct = (__float_as_uint(inv.x) >> 31) | ((__float_as_uint(inv.y) >> 31) << 1) | ((__float_as_uint(inv.z) >> 31) << 2);
	const unsigned long long lv = __builtin_amdgcn_ballot_w64(live);
	const unsigned int oct = __builtin_amdgcn_readlane(lane_oct, lv ? __builtin_ctzll(lv) : 0);
	const bool one_octant = __builtin_amdgcn_ballot_w64(live && lane_oct != oct) == 0ull;
	c4ptr B = T.offset_floats((unsigned int)sc.hoist_count * (RT_AOS_F4 * 4u) + (one_octant ? oct : 0u) * (unsigned int)sc.hoist_count * 8u);
	for (int j = 0; j < sc.hoist_count;) {
		const float4 b0 = B[0], b1 = B[1];
		const int second = __float_as_int(b1.z);             // a partner with the same box follows (rt_lbvh.hip k_hoist), or -1
		const int members = second >= 0 ? 2 : 1;
		if constexpr (COUNT) lc.box += live;
		const bool meets = one_octant ? __builtin_amdgcn_ballot_w64(live && slab_hit_oct(b0, b1, inv, oi, lim_hi)) != 0
		                              : __builtin_amdgcn_ballot_w64(live && slab_hit(b0, make_float4(b0.w, b1.x, b1.y, 0.f), inv, oi, lim_hi)) != 0;
		for (int m = 0; m < members && meets; m++) {
			RT_PROF(wp, WP_TRI);
			const c4ptr Tm = T.offset_floats((RT_AOS_F4 * 4) * m);
			const int i = m == 0 ? __float_as_int(b1.w) : second;
			const float4 N4 = Tm[0], P14 = Tm[1];
			const vec3 nn = xyz(N4), p1 = xyz(P14);
			const float t = dot(p1 - o, nn) / dot(d, nn);                           // Source.cpp:383
			const lanemask visit = lanes(live) & lanes(i != skip);                  // Source.cpp:302
			const lanemask cand = visit & lanes(t > RT_EPS_BELOW) & (lanes(t < lim) | (lanes(t == h.t) & lanes(i < h.idx))); // Source.cpp:385, 391; index order when a tree hit came first (rt_vis.inl)
			if constexpr (COUNT) { lc.plane += lane_in(visit); lc.point += lane_in(visit) && !(t < 0.0f); lc.edge += lane_in(cand); }
			if (cand == 0) continue;
			const vec3 I = make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);      // Source.cpp:389
			const vec3 w = I - p1;
			lanemask open = cand, c2 = 0;
			if constexpr (!COUNT) c2 = edge_filter(Tm[2], Tm[3], Tm[4], w, cand, open);
			if (open != 0) {
				const float4 P24 = Tm[5], P34 = Tm[6], E0 = Tm[7], E1 = Tm[8], E2 = Tm[9];
				const lanemask c0 = open & lanes(dot(nn, cross(xyz(E0), w)) > 0.0f);            // Source.cpp:392
				if constexpr (COUNT) lc.edge += lane_in(c0);
				if (c0 != 0) {
					const lanemask c1 = c0 & lanes(dot(nn, cross(xyz(E1), I - xyz(P24))) > 0.0f);   // Source.cpp:393
					if constexpr (COUNT) lc.edge += lane_in(c1);
					if (c1 != 0) c2 |= c1 & lanes(dot(nn, cross(xyz(E2), I - xyz(P34))) > 0.0f);   // Source.cpp:394
				}
			}
			if (c2 == 0) continue;
			if (lane_in(c2)) {                                                      // Source.cpp:396
				h.t = t; h.idx = i; h.p = I;
				lim = first_only ? -1.0f : t;
				lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
			}
		}
		j += members;
		T = T.offset_floats((RT_AOS_F4 * 4) * members);
		B = B.offset_floats(8 * members);
	}
}

// Upper bound on the distance of any hit: where the ray leaves the (padded) scene box.
__device__ __forceinline__ float exit_bound(const DevScene& sc, vec3 o, vec3 d) {
	const float inf = __builtin_huge_valf();
	float tx = ((d.x > 0.f ? sc.bb_hi[0] : sc.bb_lo[0]) - o.x) * __builtin_amdgcn_rcpf(d.x);
	float ty = ((d.y > 0.f ? sc.bb_hi[1] : sc.bb_lo[1]) - o.y) * __builtin_amdgcn_rcpf(d.y);
	float tz = ((d.z > 0.f ? sc.bb_hi[2] : sc.bb_lo[2]) - o.z) * __builtin_amdgcn_rcpf(d.z);
	tx = (d.x != 0.f) ? tx : inf;
	ty = (d.y != 0.f) ? ty : inf;
	tz = (d.z != 0.f) ? tz : inf;
	const float te = fminf(fminf(tx, ty), tz);           // fminf drops NaN operands
	return te + fabsf(te) * 1e-4f;                       // NaN (all-NaN ray) is dropped by the caller's fminf
}

// Source.cpp:488: `1.0f / (dist * dist) * l.intensity / 255.0 / 255.0f` — the two divisions are binary64
// (255.0 is a double literal) and the result is rounded to binary32 when it is stored in `modify`.
// For every binary32 x that is the same value as ONE binary32 division x / 65025.0f: the exact
// quotient x / 65025 is never closer to a binary32 rounding boundary than 2^-42 relative (65025 * odd
// 25-bit midpoint needs 39+ significant bits, x has 24), while the two binary64 roundings move it by
// 2^-52 at most, so both round to the same binary32.  Checked exhaustively, bit for bit, over all 2^31
// non-negative binary32 values and +inf against the binary64 form.
#ifndef RT_NT_STORE
#define RT_NT_STORE 0
#endif
#ifndef RT_DIV65025_FAST
#define RT_DIV65025_FAST 0
#endif
__device__ __forceinline__ float div65025(float x) {
#if RT_DIV65025_FAST
	// division by the constant without the division sequence (Markstein): exhaustively equal to x / 65025.0f
	// for every finite binary32 x with |x| in [2^-100, 2^100] (scripts/ubench/div65025.hip), the IEEE sequence outside
	const float c = 1.0f / 65025.0f;
	const float q = x * c;
	const float q2 = __builtin_fmaf(__builtin_fmaf(-q, 65025.0f, x), c, q);
	const float ax = fabsf(x);
	return (ax >= 0x1p-100f && ax <= 0x1p100f) ? q2 : x / 65025.0f;
#else
	return x / 65025.0f;
#endif
}

// One pixel into the frame: three binary32 channels (Source.cpp:612-614), or — FrameParams::rgba8 — the 8-bit pixel
// the reference's viewer ends up displaying: the GL unorm conversion of glTexImage2D(GL_RGB, GL_FLOAT) (Source.cpp:1270)
// as SURVEY 8a fixes it, q = (int)(c * 255.0f + 0.5f), NaN and negative -> 0, packed R, G, B, 255 in memory order.
__device__ __forceinline__ unsigned int unorm8(float c) {
	c = c > 0.0f ? c : 0.0f;                                  // (NaN fails the comparison)
	c = c > 1.0f ? 1.0f : c;
	return (unsigned int)(c * 255.0f + 0.5f);
}
__device__ __forceinline__ void store_pixel(const FrameParams& fp, int frame, long long pixel, vec3 c) {
	if (fp.rgba8) {
		unsigned int* out = (unsigned int*)fp.out + (long long)frame * fp.frame_stride + pixel;
		const unsigned int px = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | 0xff000000u;
#if RT_NT_STORE
		__builtin_nontemporal_store(px, out);
#else
		*out = px;
#endif
	} else {
		float* out = fp.out + (long long)frame * fp.frame_stride + pixel * 3;   // Source.cpp:550
#if RT_NT_STORE
		// Source.cpp:612-614 as streaming stores (off; -DRT_NT_STORE=1 to try): a frame is written once and never read by the kernel,
		// and at 100 MB (3840x2160) it pushes the scene out of the L2s it passes through.  Measured plain | streaming, us/frame: 102k
		// triangles 800x800 18.85 | 18.35, 3840x2160 139.6 | 138.8, 7680x4320 512.8 | 498.7; room 800x800 10.52 | 10.46, 7680x4320 520.0 |
		// 520.1 — but the 128-byte lines that two tiles of a tree kernel share then go to memory twice: WRITE_SIZE / frame bytes 1.00 ->
		// 1.33 at 800x800, 1.43 at 3840x2160, 1.04 -> 1.11 at 7680x4320 (brute force, whose runs go down a column, stays at 1.000).
		// Up to 2.7 % of time against 11-43 % more bytes written: left off.
		__builtin_nontemporal_store(c.x, out); __builtin_nontemporal_store(c.y, out + 1); __builtin_nontemporal_store(c.z, out + 2);
#else
		out[0] = c.x; out[1] = c.y; out[2] = c.z;                               // Source.cpp:612-614
#endif
	}
}

// ---- pixel <-> work index ------------------------------------------------------------------
// Work items are numbered tile-major: item g -> tile g/64 (tiles run left to right, then up),
// pixel g%64 inside the 8x8 tile.  A wave owns `tiles_per_wave` consecutive tiles.
struct PixelId {
	int x, ly;      // column, LOCAL row of this shard
	bool inside;
};

// (tx0, ty0) is the wave-uniform position of tile g0 >> 6; a lane's item g is in that tile or in
// one of the next few, so no per-lane division is needed.
__device__ __forceinline__ PixelId pixel_of(unsigned int g, unsigned int base_tile, int tx0, int ty0, int tiles_x,
                                            int width, int local_rows) {
	PixelId p;
	const int in = (int)(g & 63u);
	int tx = tx0 + (int)((g >> 6) - base_tile), ty = ty0;
	while (tx >= tiles_x) { tx -= tiles_x; ty++; }
	p.x = tx * RT_TILE_W + (in & (RT_TILE_W - 1));
	p.ly = ty * RT_TILE_H + (in / RT_TILE_W);
	p.inside = (p.x < width) && (p.ly < local_rows);
	return p;
}

__device__ __forceinline__ int global_row(int ly, const FrameParams& fp) {
	// rt_internal.h FrameParams: local row -> tile of this shard -> global tile -> global row
	if (fp.tile_stride == 1 && fp.tile_first == 0) return ly;          // whole frame (wave-uniform branch)
	const int lt = ly / fp.tile_rows;
	return (fp.tile_first + lt * fp.tile_stride) * fp.tile_rows + (ly - lt * fp.tile_rows);
}

// Source.cpp:534-545: u, v in double then rounded; dir = (trans * vec4(u, v, 0.5, 1)).xyz
__device__ __forceinline__ vec3 primary_dir(int x, int y, const FrameParams& fp, int frame, const float* pose = nullptr) {
	// (x - w/2.0) / (float)w in double, rounded to float.  For w, h < 2^22 the numerator is a
	// half-integer below 2^22 and the denominator an integer: both exact in binary32, and rounding a
	// binary64 quotient of binary32 operands to binary32 equals the correctly rounded binary32
	// quotient (53 >= 2*24 + 2), so one IEEE binary32 division gives the same bits.
	float u, v;
	if (fp.uv_fast) {
		// the same quotients without the division sequence (FrameParams::rcp_w; the launcher checked every
		// column and row of this frame size against the IEEE division, bit for bit)
		const float hf = (float)fp.height, wf = (float)fp.width;
		const float nv = (float)y - hf * 0.5f, nu = (float)x - wf * 0.5f;
		const float qv = nv * fp.rcp_h, qu = nu * fp.rcp_w;
		v = __builtin_fmaf(__builtin_fmaf(-qv, hf, nv), fp.rcp_h, qv);
		u = __builtin_fmaf(__builtin_fmaf(-qu, wf, nu), fp.rcp_w, qu);
	} else if (fp.width < (1 << 22) && fp.height < (1 << 22)) {
		v = ((float)y - (float)fp.height * 0.5f) / (float)fp.height;
		u = ((float)x - (float)fp.width * 0.5f) / (float)fp.width;
	} else {
		// (cold: frames of 4M+ pixels a side.  The empty asm keeps the compiler from hoisting the
		// binary64 halves out of the pixel loop, where they would cost two VGPRs or a spill.)
		int hh = fp.height, ww = fp.width;
		asm volatile("" : "+s"(hh), "+s"(ww));
		v = (float)(((double)y - (double)hh / 2.0) / (double)(float)hh);
		u = (float)(((double)x - (double)ww / 2.0) / (double)(float)ww);
	}
	const float* m = pose ? pose : fp.pose[frame];       // 3..5: column 0, 6..8: column 1, 9..11: m2 * 0.5f + m3 * 1.0f (set_pose)
	vec3 r;
	r.x = (m[3] * u + m[6] * v) + m[9];
	r.y = (m[4] * u + m[7] * v) + m[10];
	r.z = (m[5] * u + m[8] * v) + m[11];
	return r;
}

// primary_dir for a launch whose u, v are division-free (FrameParams::uv_fast) — the branch-free body of the above
__device__ __forceinline__ vec3 primary_dir_plain(int x, int y, const FrameParams& fp, int frame, const float* pose = nullptr) {
	const float hf = (float)fp.height, wf = (float)fp.width;
	const float nv = (float)y - hf * 0.5f, nu = (float)x - wf * 0.5f;
	const float qv = nv * fp.rcp_h, qu = nu * fp.rcp_w;
	const float v = __builtin_fmaf(__builtin_fmaf(-qv, hf, nv), fp.rcp_h, qv);
	const float u = __builtin_fmaf(__builtin_fmaf(-qu, wf, nu), fp.rcp_w, qu);
	const float* m = pose ? pose : fp.pose[frame];
	vec3 r;
	r.x = (m[3] * u + m[6] * v) + m[9];
	r.y = (m[4] * u + m[7] * v) + m[10];
	r.z = (m[5] * u + m[8] * v) + m[11];
	return r;
}

template <bool COUNT>
__device__ __forceinline__ void flush_counters(unsigned long long* g, const unsigned int (&v)[CTR_N]) {
	if (!COUNT || g == nullptr) return;
#pragma unroll
	for (int k = 0; k < CTR_N; k++) {
		unsigned int s = v[k];
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
		if ((threadIdx.x & 63) == 0 && s) atomicAdd(&g[k], (unsigned long long)s);
	}
}

// ---- the render kernel -----------------------------------------------------------------------
// ACCEL 0: brute force, the scene read through the scalar cache, with a per-workgroup LDS table of the primary
// rays' numerators (scenes up to RT_LDS_MAX_TRIS triangles); 1: the same without that table (forced brute force on
// larger scenes); 2: LBVH, per-lane walk of the 4-wide tree (rt_lbvh_trav.inl); 3: LBVH, wave-uniform packet walk.
// `shadow`: the lane's ray is a shadow ray towards a light at distance `dist`.
// `all_primary` (wave-uniform): every live lane holds a primary ray from the camera.
// `li`: for a shadow ray, the index of the light it is aimed at (per lane; the state machine's lanes may be at different lights).
template <int ACCEL, bool COUNT>
__device__ __forceinline__ Hit trace(const DevScene& sc, const float4* lds_tri, int lds_stride, const float* lds_num0,
                                     vec3 o, vec3 d, int skip, bool live, bool shadow, float dist, int li, bool all_primary,
                                     LaneCounters& lc, WaveProf& wp);

} // namespace rt

#include "rt_lbvh_trav.inl"
#include "rt_lgrid_trav.inl"

namespace rt {

// Shadow rays of lights that have a direction grid (rt_lgrid.hip), for a wave whose lanes may be aimed at different lights:
// one pass per light present.  Returns the lanes that were served; their result is in `hg`.
template <bool COUNT>
__device__ __forceinline__ bool shadow_by_grids(const DevScene& sc, vec3 o, vec3 d, int skip, bool live, bool shadow, float dist, int li,
                                                Hit& hg, LaneCounters& lc, WaveProf& wp) {
	bool served = false;
	const bool wants = live && shadow && li >= 0 && li < sc.n_lgrid;
	unsigned long long todo = __builtin_amdgcn_ballot_w64(wants);
	while (todo) {
		const int l = __builtin_amdgcn_readlane(li, __builtin_ctzll(todo));
		const bool mine = wants && li == l;
		todo &= ~__builtin_amdgcn_ballot_w64(mine);
		const __attribute__((address_space(4))) LightGrid* G = (const __attribute__((address_space(4))) LightGrid*)(unsigned long long)(sc.lgrid + l);
		if (G->R <= 0) continue;                                   // this light has no grid: the tree, with everybody else
		const float4 L4 = c4ptr_of(sc.lights)[(unsigned int)l];
		const Hit x = grid_shadow<COUNT>(sc, l, make3(L4.x, L4.y, L4.z), o, d, skip, mine, dist, lc, wp);
		if (mine) { hg = x; served = true; }
	}
	return served;
}

// castRay over the tree for lanes of any kind (state-machine kernel, ray batches): per lane (ACCEL 2) or by packets (3)
template <int ACCEL, bool COUNT>
__device__ __forceinline__ Hit trace_tree(const DevScene& sc, const float4* lds_tri, vec3 o, vec3 d, int skip, bool live, bool shadow, float dist,
                                          LaneCounters& lc, WaveProf& wp) {
	if constexpr (ACCEL == 2) {
		return cast_ray_lbvh<COUNT>(sc, (int*)lds_tri, o, d, skip, live, shadow, dist, lc, wp);
	} else {
		// packets: one walk of the tree per octant present in the wave (one pass for almost every wave)
		const vec3 ds = make3(fabsf(d.x) < 1e-30f ? 1e-30f : d.x, fabsf(d.y) < 1e-30f ? 1e-30f : d.y, fabsf(d.z) < 1e-30f ? 1e-30f : d.z);
		const bool near = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) < sc.far_guard;   // false for NaN too
		vec3 inv = make3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
		vec3 oi = make3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
		const float ex = fminf(fminf(fmaxf(__builtin_fmaf(sc.bb_lo[0], inv.x, oi.x), __builtin_fmaf(sc.bb_hi[0], inv.x, oi.x)),
		                             fmaxf(__builtin_fmaf(sc.bb_lo[1], inv.y, oi.y), __builtin_fmaf(sc.bb_hi[1], inv.y, oi.y))),
		                       fmaxf(__builtin_fmaf(sc.bb_lo[2], inv.z, oi.z), __builtin_fmaf(sc.bb_hi[2], inv.z, oi.z)));
		float lim = near ? fminf(ex + fabsf(ex) * 1e-4f, 1e9f) : 1e9f;
		if (shadow) lim = fminf(lim, __uint_as_float(__float_as_uint(dist) + 1u));
		lim = fminf(lim, 1e9f);
		if (!near) { inv = make3(0.f, 0.f, 0.f); oi = make3(0.f, 0.f, 0.f); }   // boxes not trusted: every box tests as hit for this lane
		const unsigned int lane_oct = (__float_as_uint(ds.x) >> 31) | ((__float_as_uint(ds.y) >> 31) << 1) | ((__float_as_uint(ds.z) >> 31) << 2);
		unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
		Hit h;
		h.t = 1e9f;
		h.idx = -1;
		h.p = make3(0.f, 0.f, 0.f);
		while (todo) {
			const unsigned int oct = __builtin_amdgcn_readlane(lane_oct, __builtin_ctzll(todo));
			const bool mine = live && lane_oct == oct;
			todo &= ~__builtin_amdgcn_ballot_w64(mine);
			RT_PROF(wp, WP_TRACE);
			const Hit hp = cast_ray_packet<COUNT>(sc, (int)oct, o, d, inv, oi, skip, mine ? lim : -1.0f, shadow, lc, wp);
			if (mine) h = hp;
		}
		return h;
	}
}

template <int ACCEL, bool COUNT>
__device__ __forceinline__ Hit trace(const DevScene& sc, const float4* lds_tri, int lds_stride, const float* lds_num0,
                                     vec3 o, vec3 d, int skip, bool live, bool shadow, float dist, int li, bool all_primary,
                                     LaneCounters& lc, WaveProf& wp) {
	if constexpr (ACCEL == 2 || ACCEL == 3) {
		// shadow rays of lights with a direction grid are answered from it; everything else walks the tree
		if (sc.n_lgrid > 0 && __builtin_amdgcn_ballot_w64(live && shadow) != 0ull) {
			Hit hg;
			hg.t = 1e9f; hg.idx = -1; hg.p = make3(0.f, 0.f, 0.f);
			const bool served = shadow_by_grids<COUNT>(sc, o, d, skip, live, shadow, dist, li, hg, lc, wp);
			const bool rest = live && !served;
			if (__builtin_amdgcn_ballot_w64(rest) == 0ull) return hg;
			const Hit hw = trace_tree<ACCEL, COUNT>(sc, lds_tri, o, d, skip, rest, shadow, dist, lc, wp);
			return served ? hg : hw;
		}
		return trace_tree<ACCEL, COUNT>(sc, lds_tri, o, d, skip, live, shadow, dist, lc, wp);
	} else if constexpr (COUNT) {
		// instrumented, reference-order loop: the counters follow Source.cpp's own short-circuits
		return cast_ray_brute<true>(sc.planes, sc.stride, sc.n_tris, o, d, skip, live, lc);
	} else {
		// limit of the search (see cast_ray_pruned); idle lanes get a limit nothing passes
		// Slab arithmetic t = b * inv - o * inv must stay finite: a direction component below 1e-30
		// is widened to 1e-30 (its slab interval is then ~(-1e28, 1e28) around any origin inside the
		// slab, empty for origins outside), and the host switches culling off (far_guard) for origins
		// or scenes large enough for the products to overflow or to lose the pad to rounding.
		const vec3 ds = make3(fabsf(d.x) < 1e-30f ? 1e-30f : d.x, fabsf(d.y) < 1e-30f ? 1e-30f : d.y, fabsf(d.z) < 1e-30f ? 1e-30f : d.z);
		const bool near = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) < sc.far_guard;   // false for NaN too
		vec3 inv = make3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
		vec3 oi = make3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
		// where the ray leaves the padded scene box: no hit lies beyond (same arithmetic as the slabs)
		const float ex = fminf(fminf(fmaxf(__builtin_fmaf(sc.bb_lo[0], inv.x, oi.x), __builtin_fmaf(sc.bb_hi[0], inv.x, oi.x)),
		                             fmaxf(__builtin_fmaf(sc.bb_lo[1], inv.y, oi.y), __builtin_fmaf(sc.bb_hi[1], inv.y, oi.y))),
		                       fmaxf(__builtin_fmaf(sc.bb_lo[2], inv.z, oi.z), __builtin_fmaf(sc.bb_hi[2], inv.z, oi.z)));
		float lim = near ? fminf(ex + fabsf(ex) * 1e-4f, 1e9f) : 1e9f;       // 1e9: Source.cpp:278
		if (shadow) lim = fminf(lim, __uint_as_float(__float_as_uint(dist) + 1u));   // first float > dist (dist >= +0)
		lim = fminf(lim, 1e9f);                                              // (dist == inf -> NaN -> dropped)
		if (!live) lim = -1.0f;
		if (!near) { inv = make3(0.f, 0.f, 0.f); oi = make3(0.f, 0.f, 0.f); }   // every box tests as hit
		// octant of the wave: sign bits of the (clamped) direction; uniform over the live lanes?
		const unsigned int lane_oct = (__float_as_uint(ds.x) >> 31) | ((__float_as_uint(ds.y) >> 31) << 1) | ((__float_as_uint(ds.z) >> 31) << 2);
		const unsigned long long lv = __builtin_amdgcn_ballot_w64(live);
		const int oct = (int)__builtin_amdgcn_readlane(lane_oct, lv ? __builtin_ctzll(lv) : 0);
		const bool oct_uniform = __builtin_amdgcn_ballot_w64(live && (lane_oct != (unsigned int)oct || !near)) == 0ull;
		const c4ptr tri = c4ptr_of(sc.planes);
		if constexpr (ACCEL == 0) {
			if (oct_uniform) {
				if (all_primary) return cast_ray_pruned<true, true>(sc, tri, oct, lds_num0, o, d, inv, oi, skip, lim, false, wp);
				return cast_ray_pruned<false, true>(sc, tri, oct, lds_num0, o, d, inv, oi, skip, lim, shadow, wp);
			}
			return cast_ray_pruned<false, false>(sc, tri, 0, lds_num0, o, d, inv, oi, skip, lim, shadow, wp);
		} else {
			if (oct_uniform) return cast_ray_pruned<false, true>(sc, tri, oct, nullptr, o, d, inv, oi, skip, lim, shadow, wp);
			return cast_ray_pruned<false, false>(sc, tri, 0, nullptr, o, d, inv, oi, skip, lim, shadow, wp);
		}
	}
}

// castRay by brute force for ONE kind of ray (the lockstep kernel below knows at every call site whether
// its rays are primary, bounce or shadow rays, so each site carries only the variants it can need).
enum { RAY_PRIMARY = 0, RAY_BOUNCE = 1, RAY_SHADOW = 2 };

// The scene constants a trace needs (box, guard, offsets, counts: ~17 SGPRs) RE-READ from the kernel arguments —
// a few scalar loads per trace — instead of living in SGPRs for the whole kernel, where the register allocator
// spills them to VGPR lanes and every use costs a v_readlane (a vector-ALU slot).  DevScene is the first argument
// of every kernel that gets here (so it sits at offset 0 of the kernel-argument segment); the empty asm keeps the
// loads from being hoisted back out of the tile loop; only the words a trace uses survive as loads.
__device__ __forceinline__ DevScene scene_from_kernel_args(const DevScene& sc_arg) {
	DevScene sc = sc_arg;
#if defined(__HIP_DEVICE_COMPILE__)
	static_assert(sizeof(DevScene) % 4 == 0, "copied word by word");
	const __attribute__((address_space(4))) unsigned int* w = (const __attribute__((address_space(4))) unsigned int*)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(w));
	unsigned int* dst = (unsigned int*)&sc;
#pragma unroll
	for (unsigned int k = 0; k < sizeof(DevScene) / 4; k++) dst[k] = w[k];
#endif
	return sc;
}
// The same for the scalar fields of FrameParams (everything behind the poses), which follows DevScene in the argument segment of
// the render kernels (DevScene, FrameParams, int): what a tile needs of the frame at its END — the output pointer, strides, the
// frame size — is read there instead of living in spilled SGPRs through the walks (each use of a spilled SGPR is a v_readlane).
struct RenderKernelArgs { DevScene sc; FrameParams fp; int tiles_per_wave; };
__device__ __forceinline__ FrameParams frame_from_kernel_args(const FrameParams& fp_arg) {
	FrameParams fp = fp_arg;
#if defined(__HIP_DEVICE_COMPILE__)
	static_assert(sizeof(FrameParams) % 4 == 0 && offsetof(FrameParams, n_frames) % 4 == 0 && offsetof(RenderKernelArgs, fp) % 4 == 0, "copied word by word");
	const __attribute__((address_space(4))) unsigned int* w = (const __attribute__((address_space(4))) unsigned int*)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(w));
	w += offsetof(RenderKernelArgs, fp) / 4;
	unsigned int* dst = (unsigned int*)&fp;
#pragma unroll
	for (unsigned int k = offsetof(FrameParams, n_frames) / 4; k < sizeof(FrameParams) / 4; k++) dst[k] = w[k];
#endif
	return fp;
}
// ... and one frame's pose (12 floats at a wave-uniform offset of the same segment)
__device__ __forceinline__ void pose_from_kernel_args(const FrameParams& fp_arg, int frame, float m[12]) {
#if defined(__HIP_DEVICE_COMPILE__)
	const __attribute__((address_space(4))) float* w = (const __attribute__((address_space(4))) float*)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(w));
	w += offsetof(RenderKernelArgs, fp) / 4 + (unsigned int)frame * 12u;
#pragma unroll
	for (int k = 0; k < 12; k++) m[k] = w[k];
#else
	for (int k = 0; k < 12; k++) m[k] = fp_arg.pose[frame][k];
#endif
}
// The exit bound of a trace (where the ray leaves the padded scene box: no hit lies beyond) only ever skips work — and it has
// none to skip for a ray that STARTS inside the box: everything the scene holds lies before the exit.  A bounce ray starts at a hit
// point, i.e. on a triangle (the box is padded by 1e-3 of the scene, orders of magnitude above the rounding of o + t*d); a primary
// ray starts at the camera, and the launcher says per frame whether that is inside (FrameParams::origin_inside, wave-uniform).
// Those traces search up to the reference's own 1e9 (Source.cpp:278) and save the slab arithmetic: 13 vector instructions.
#ifndef RT_SKIP_IDLE_EXIT
#define RT_SKIP_IDLE_EXIT 1
#endif

template <int KIND>
__device__ __forceinline__ bool exit_bound_is_idle(bool from_inside) {
#if RT_SKIP_IDLE_EXIT
	if constexpr (KIND == RAY_BOUNCE) return true;
	else if constexpr (KIND == RAY_PRIMARY) return from_inside;
#endif
	return false;
}
template <int ACCEL, int KIND>
__device__ __forceinline__ Hit trace_brute(const DevScene& sc_arg, const float* lds_num0, vec3 o, vec3 d, int skip, bool live, float dist, WaveProf& wp, bool from_inside = false) {
	static_assert(ACCEL == 0 || ACCEL == 1, "brute force only");
	const DevScene sc = scene_from_kernel_args(sc_arg);
	// (the comments of trace<> above apply word for word: clamped direction, exit bound, far_guard, octant)
	const vec3 ds = make3(fabsf(d.x) < 1e-30f ? 1e-30f : d.x, fabsf(d.y) < 1e-30f ? 1e-30f : d.y, fabsf(d.z) < 1e-30f ? 1e-30f : d.z);
	const bool near = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) < sc.far_guard;
	vec3 inv = make3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
	vec3 oi = make3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
	float lim = 1e9f;                                                    // 1e9: Source.cpp:278 (the box's diagonal as a free limit for bounce rays: no gain, measured)
	if (!exit_bound_is_idle<KIND>(from_inside)) {
		const float ex = fminf(fminf(fmaxf(__builtin_fmaf(sc.bb_lo[0], inv.x, oi.x), __builtin_fmaf(sc.bb_hi[0], inv.x, oi.x)),
		                             fmaxf(__builtin_fmaf(sc.bb_lo[1], inv.y, oi.y), __builtin_fmaf(sc.bb_hi[1], inv.y, oi.y))),
		                       fmaxf(__builtin_fmaf(sc.bb_lo[2], inv.z, oi.z), __builtin_fmaf(sc.bb_hi[2], inv.z, oi.z)));
		lim = fminf(ex + fabsf(ex) * 1e-4f, 1e9f);
	}
	const bool some_far = __builtin_amdgcn_ballot_w64(live && !near) != 0ull;      // (rays from beyond far_guard: rare, wave-uniform branch)
	if (some_far && !near) { lim = 1e9f; inv = make3(0.f, 0.f, 0.f); oi = make3(0.f, 0.f, 0.f); }
	// a shadow ray's limit is the first float > dist (Source.cpp:476); the exit bound could only repeat what that already says
	if constexpr (KIND == RAY_SHADOW) lim = __uint_as_float(__float_as_uint(dist) + 1u);
	lim = fminf(lim, 1e9f);
	if (!live) lim = -1.0f;
	const unsigned int lane_oct = (__float_as_uint(ds.x) >> 31) | ((__float_as_uint(ds.y) >> 31) << 1) | ((__float_as_uint(ds.z) >> 31) << 2);
	const unsigned long long lv = __builtin_amdgcn_ballot_w64(live);
	const int oct = (int)__builtin_amdgcn_readlane(lane_oct, lv ? __builtin_ctzll(lv) : 0);
	const bool oct_uniform = !some_far && __builtin_amdgcn_ballot_w64(live && lane_oct != (unsigned int)oct) == 0ull;
	const c4ptr tri = c4ptr_of(sc.planes);
	constexpr bool first_only = KIND == RAY_SHADOW;
	if (oct_uniform) {
		if constexpr (KIND == RAY_PRIMARY && ACCEL == 0) return cast_ray_pruned<true, true>(sc, tri, oct, lds_num0, o, d, inv, oi, skip, lim, false, wp);
		else return cast_ray_pruned<false, true>(sc, tri, oct, lds_num0, o, d, inv, oi, skip, lim, first_only, wp);
	}
	return cast_ray_pruned<false, false>(sc, tri, 0, lds_num0, o, d, inv, oi, skip, lim, first_only, wp);
}

// castRay over the LBVH by packets for ONE kind of ray (lockstep kernel): one walk per direction octant
// present among the live lanes (one pass for almost every wave).  Same set-up as trace<3>.
template <int KIND, bool COUNT>
__device__ __forceinline__ Hit trace_packet(const DevScene& sc_arg, vec3 o, vec3 d, int skip, bool live, float dist, LaneCounters& lc, WaveProf& wp, HoistCtx hc = HoistCtx(),
                                             bool from_inside = false) {
	const DevScene sc = scene_from_kernel_args(sc_arg);
	const vec3 ds = make3(fabsf(d.x) < 1e-30f ? 1e-30f : d.x, fabsf(d.y) < 1e-30f ? 1e-30f : d.y, fabsf(d.z) < 1e-30f ? 1e-30f : d.z);
	// An origin inside a scene box that lies within far_guard is within far_guard — hit points always, the camera when the launcher
	// says so: no check then.  (Packets only: 3840x2160 128.05 -> 127.15 us/frame, 800x800 18.95 -> 18.90; the brute-force traces
	// came out 0.3 % slower with it and keep the check.)
	const bool near_known = RT_SKIP_IDLE_EXIT && sc.box_within_guard != 0 && (KIND != RAY_PRIMARY || from_inside);
	const bool near = near_known || fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) < sc.far_guard;   // false for NaN too
	vec3 inv = make3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
	vec3 oi = make3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
	float lim = 1e9f;
	if (!exit_bound_is_idle<KIND>(from_inside)) {
		const float ex = fminf(fminf(fmaxf(__builtin_fmaf(sc.bb_lo[0], inv.x, oi.x), __builtin_fmaf(sc.bb_hi[0], inv.x, oi.x)),
		                             fmaxf(__builtin_fmaf(sc.bb_lo[1], inv.y, oi.y), __builtin_fmaf(sc.bb_hi[1], inv.y, oi.y))),
		                       fmaxf(__builtin_fmaf(sc.bb_lo[2], inv.z, oi.z), __builtin_fmaf(sc.bb_hi[2], inv.z, oi.z)));
		lim = fminf(ex + fabsf(ex) * 1e-4f, 1e9f);
	}
	if (!near_known && __builtin_amdgcn_ballot_w64(live && !near) != 0ull) {          // rays from beyond far_guard: every box tests as hit for them
		if (!near) { lim = 1e9f; inv = make3(0.f, 0.f, 0.f); oi = make3(0.f, 0.f, 0.f); }
	}
	if constexpr (KIND == RAY_SHADOW) lim = __uint_as_float(__float_as_uint(dist) + 1u);   // (no exit bound: see trace_brute)
	lim = fminf(lim, 1e9f);
	const unsigned int lane_oct = (__float_as_uint(ds.x) >> 31) | ((__float_as_uint(ds.y) >> 31) << 1) | ((__float_as_uint(ds.z) >> 31) << 2);
	unsigned long long todo = __builtin_amdgcn_ballot_w64(live);
	Hit h;
	h.t = 1e9f;
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	while (todo) {
		const unsigned int oct = __builtin_amdgcn_readlane(lane_oct, __builtin_ctzll(todo));
		const bool mine = live && lane_oct == oct;
		todo &= ~__builtin_amdgcn_ballot_w64(mine);
		RT_PROF(wp, WP_TRACE);
		const Hit hp = cast_ray_packet<COUNT>(sc, (int)oct, o, d, inv, oi, skip, mine ? lim : -1.0f, KIND == RAY_SHADOW, lc, wp, hc);
		if (mine) h = hp;
	}
	return h;
}

// COUNT (tree kernels only): the work counters of SURVEY 8d, as the state-machine kernel keeps them
template <int ACCEL, int KIND, bool COUNT>
__device__ __forceinline__ Hit trace_kind(const DevScene& sc, const float* lds_num0, vec3 o, vec3 d, int skip, bool live, float dist, LaneCounters& lc, WaveProf& wp,
                                          HoistCtx hc = HoistCtx(), bool from_inside = false) {
	if constexpr (ACCEL == 3) return trace_packet<KIND, COUNT>(sc, o, d, skip, live, dist, lc, wp, hc, from_inside);
	else if constexpr (ACCEL == 2) return cast_ray_lbvh<COUNT>(sc, (int*)lds_num0, o, d, skip, live, KIND == RAY_SHADOW, dist, lc, wp, hc, exit_bound_is_idle<KIND>(from_inside));   // (lds_num0: the workgroup's LDS, here the stacks)
	else {
		static_assert(!COUNT, "the brute-force counters follow the reference-order loop of the state-machine kernel");
		return trace_brute<ACCEL, KIND>(sc, lds_num0, o, d, skip, live, dist, wp, from_inside);
	}
}

// The shadow ray of light `li` (wave-uniform) for the lockstep kernel: through the light's direction grid when the commit
// made one (LBVH scenes, rt_lgrid.hip), through the tree or the brute-force loop otherwise.
template <int ACCEL, bool COUNT>
__device__ __forceinline__ Hit trace_shadow(const DevScene& sc_arg, const float* lds_num0, int li, vec3 L, vec3 o, vec3 d, int skip, bool live, float dist,
                                            LaneCounters& lc, WaveProf& wp) {
#ifdef RT_EXPERIMENT
	if (sc_arg.debug & 64) { Hit h; h.t = 1e9f; h.idx = -1; h.p = make3(0.f, 0.f, 0.f); return h; }     // (attribution: no shadow work at all)
#endif
	if constexpr (ACCEL == 2 || ACCEL == 3) {
		if (li < sc_arg.n_lgrid) {
			const __attribute__((address_space(4))) LightGrid* G = (const __attribute__((address_space(4))) LightGrid*)(unsigned long long)(sc_arg.lgrid + li);
			if (G->R > 0) {
				if constexpr (ACCEL == 3) return grid_shadow<COUNT>(scene_from_kernel_args(sc_arg), li, L, o, d, skip, live, dist, lc, wp);
				else return grid_shadow<COUNT>(sc_arg, li, L, o, d, skip, live, dist, lc, wp);
			}
		}
	}
	return trace_kind<ACCEL, RAY_SHADOW, COUNT>(sc_arg, lds_num0, o, d, skip, live, dist, lc, wp);
}

} // namespace rt

#include "rt_vis.inl"

namespace rt {

// Per-lane attribute fetch by hit index (shininess Source.cpp:570/575, normal :561, colour :489).
template <int ACCEL>
__device__ __forceinline__ float4 tri_plane(const DevScene& sc, const float4* lds_tri, int lds_stride, int plane, int idx) {
	return sc.planes[plane * sc.stride + idx];
}

template <int ACCEL, bool COUNT>
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 8      /* 64 VGPRs: 8 waves per SIMD; +6 % at steady state over 7 (measured) */
#endif
#ifndef RT_MIN_WAVES_PACKET
#define RT_MIN_WAVES_PACKET 8
#endif
#ifndef RT_MIN_WAVES_LBVH
#define RT_MIN_WAVES_LBVH 7    /* measured: 7 beats 6 by 3 %, 8 and 5 lose */
#endif
__global__ void __launch_bounds__(256, ACCEL == 2 ? RT_MIN_WAVES_LBVH : (ACCEL == 3 ? RT_MIN_WAVES_PACKET : RT_MIN_WAVES)) render_kernel(DevScene sc, FrameParams fp, int tiles_per_wave) {
#ifdef RT_PROFILE
	const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
	extern __shared__ float4 s_tri[];
	const int lds_stride = lds_stride_of(sc.n_tris);
	float* s_num0 = (float*)s_tri;
	int frame = fp.n_frames > 1 ? (int)(blockIdx.x / fp.blocks_per_frame) : 0;      // workgroup-uniform
	unsigned int block_in_frame = blockIdx.x - (unsigned int)frame * fp.blocks_per_frame;
	if constexpr (ACCEL == 2 || ACCEL == 3) {
		if (fp.band_blocks > 0) {          // XCD-aware order (rt_internal.h FrameParams::band_blocks)
			const unsigned int xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
			frame = (int)(j / fp.band_blocks);
			const unsigned int band = (xcd + (unsigned int)frame) & 7u;
			block_in_frame = band * fp.band_blocks + (j - (unsigned int)frame * fp.band_blocks);
			if (block_in_frame >= fp.blocks_per_frame) return;       // (the last band is padded; no barrier in these kernels)
		}
	}
	const vec3 cam = make3(fp.pose[frame][0], fp.pose[frame][1], fp.pose[frame][2]);
	if constexpr (ACCEL == 0) {
		// numerator of Source.cpp:383 for rays that start at the camera: the one per-workgroup table
		// (the scene itself is read through the scalar cache, see cast_ray_pruned)
		for (int i = threadIdx.x; i < lds_stride; i += blockDim.x)
			s_num0[i] = dot(xyz(sc.planes[PL_P1 * sc.stride + i]) - cam, xyz(sc.planes[PL_N * sc.stride + i]));
		__syncthreads();
	}

	const int tiles_x = (fp.width + RT_TILE_W - 1) / RT_TILE_W;
	const int tiles_y = (fp.local_rows + RT_TILE_H - 1) / RT_TILE_H;
	// work items fit 32 bits: width * height * 3 < 2^31 is checked on the host (Source.cpp:550 uses int)
	const unsigned int n_items = (unsigned int)tiles_x * (unsigned int)tiles_y * 64u;
	const unsigned int wave = __builtin_amdgcn_readfirstlane(block_in_frame * (blockDim.x >> 6) + (threadIdx.x >> 6));
	unsigned int next = wave * (unsigned int)tiles_per_wave * 64u;  // wave-uniform cursor
	unsigned int end = next + (unsigned int)tiles_per_wave * 64u;
	if (end > n_items || end < next) end = n_items;
	if (next > n_items) next = n_items;
	// position of the cursor's tile, kept in step with `next` (uniform, one division per wave)
	unsigned int base_tile = next >> 6;
	int ty0 = (int)(base_tile / (unsigned int)tiles_x), tx0 = (int)(base_tile - (unsigned int)ty0 * (unsigned int)tiles_x);

	// per-lane ray state
	int stage = ST_IDLE;
	vec3 o = cam, d = make3(0.f, 0.f, 0.f);
	int skip = -1;
	int out_index = 0;
	int bounces = 0;
	float dim = 1.0f;
	// shading record (what processLighting gets: hit.point, hit.cur) + light cursor + accumulator
	vec3 hp = make3(0.f, 0.f, 0.f);
	int hidx = -1;
	int li = 0;
	float dist = 0.f, intensity = 0.f;
	vec3 color = make3(0.f, 0.f, 0.f);
	bool lit_any = false;

	unsigned int ctr[CTR_N];
#pragma unroll
	for (int k = 0; k < CTR_N; k++) ctr[k] = 0;
	LaneCounters lc = { 0, 0, 0, 0 };
	WaveProf wp = {};
#ifdef RT_PROFILE
	wp.last = __builtin_amdgcn_s_memtime();
	wp.g = fp.counters;
	__builtin_amdgcn_s_waitcnt(0xC07F);
	wp.t[0] = wp.last - t_start;       // argument loads + LDS staging
#endif

	for (;;) {
		// ---- regeneration: idle lanes take the next pixels of the wave's tile run ---------------
		if (next < end) {
			const bool want = (stage == ST_IDLE);
			const unsigned long long m = __ballot(want);
			if (m) {
				const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0));
				const unsigned int g = next + (unsigned int)rank;
				next += (unsigned int)__popcll(m);
				if (want && g < end) {
					const PixelId px = pixel_of(g, base_tile, tx0, ty0, tiles_x, fp.width, fp.local_rows);
					if (px.inside) {
						const int y = global_row(px.ly, fp);
						out_index = px.ly * fp.width + px.x;                  // pixel of the shard's local rows (Source.cpp:550)
						d = primary_dir(px.x, y, fp, frame);
						o = cam;                                             // Source.cpp:545
						skip = -1;
						stage = ST_PRIMARY;
						if (COUNT) ctr[CTR_PRIMARY]++;
					}
				}
				// advance the uniform tile position to the new cursor
				const unsigned int nt = next >> 6;
				tx0 += (int)(nt - base_tile);
				while (tx0 >= tiles_x) { tx0 -= tiles_x; ty0++; }
				base_tile = nt;
			}
		}
		RT_STAMP(wp, 1);      // regeneration + primary ray set-up
		const bool live = (stage != ST_IDLE);
		if (!__any(live)) {
			if (next >= end) break;
			continue;
		}

		// ---- the one closest-hit loop ---------------------------------------------------------
		const bool all_primary = __builtin_amdgcn_ballot_w64(live && stage != ST_PRIMARY) == 0;
		const Hit h = trace<ACCEL, COUNT>(sc, s_tri, lds_stride, s_num0, o, d, skip, live, stage == ST_SHADOW, dist, li,
		                                  all_primary, lc, wp);
		RT_PROF(wp, WP_ITER);
		RT_STAMP(wp, 2);      // closest-hit search
		const bool hit = h.idx >= 0;

		// ---- stage transitions ----------------------------------------------------------------
		bool shade = false;       // start processLighting on (hp, hidx)
		bool finish = false;      // write the pixel
		bool mirror = false;      // reflect off the triangle just hit and trace on
		float4 N4 = make_float4(0.f, 0.f, 0.f, 0.f);
		if (hit && stage != ST_SHADOW) N4 = tri_plane<ACCEL>(sc, s_tri, lds_stride, PL_N, h.idx);
		if (stage == ST_PRIMARY) {
			if (!hit) {
				color = make3(0.f, 0.f, 0.f);                                // Source.cpp:623-625
				dim = 1.0f;
				finish = true;
			} else {
				if (COUNT) ctr[CTR_PXHIT]++;
				if (N4.w > 0.5f && !fp.shade_glsl) {                         // Source.cpp:570 (Compute.shader:257-330 has no mirrors)
					bounces = 0;
					stage = ST_BOUNCE;
					mirror = true;                                           // Source.cpp:561, 572-573
					if (COUNT) ctr[CTR_PXMIRROR]++;
				} else {
					hp = h.p; hidx = h.idx; dim = 1.0f;                      // Source.cpp:597
					shade = true;
				}
			}
		} else if (stage == ST_BOUNCE) {
			if (hit && bounces < fp.max_bounces && N4.w > 0.5f) {            // Source.cpp:575
				bounces++;
				mirror = true;                                               // Source.cpp:576-578
			} else if (hit) {                                                // Source.cpp:582
				hp = h.p; hidx = h.idx;
				dim = (float)(bounces * 1);                                  // Source.cpp:585
				shade = true;
			} else {
				color = make3(0.f, 0.f, 0.f);                                // Source.cpp:590
				dim = 1.0f;
				finish = true;
			}
		} else if (stage == ST_SHADOW) {
			if (!hit || h.t > dist) {                                        // Source.cpp:476, Compute.shader:203
				const float f = 1.0f / (dist * dist) * intensity;            // Source.cpp:488, Compute.shader:214
				if (fp.shade_glsl) {
					color.x = color.x + f / 255.0f;                          // Compute.shader:214: the scalar `result`, all binary32
				} else {
					const float modify = div65025(f);                        // ... / 255.0 / 255.0f: binary64 divisions, rounded to binary32
					const float cr = tri_plane<ACCEL>(sc, s_tri, lds_stride, PL_P1, hidx).w;
					const float cg = tri_plane<ACCEL>(sc, s_tri, lds_stride, PL_P2, hidx).w;
					const float cb = tri_plane<ACCEL>(sc, s_tri, lds_stride, PL_P3, hidx).w;
					color.x = color.x + modify * cr;                         // Source.cpp:489
					color.y = color.y + modify * cg;
					color.z = color.z + modify * cb;
				}
				lit_any = true;
			}
			li++;
			if (li >= sc.n_lights) {
				finish = true;
				if (fp.shade_glsl) {                                         // Compute.shader:330: (indirect + direct) * hit.color / 255.0, indirect == 0
					const float direct = 0.0f + color.x;
					color.x = direct * tri_plane<ACCEL>(sc, s_tri, lds_stride, PL_P1, hidx).w / 255.0f;
					color.y = direct * tri_plane<ACCEL>(sc, s_tri, lds_stride, PL_P2, hidx).w / 255.0f;
					color.z = direct * tri_plane<ACCEL>(sc, s_tri, lds_stride, PL_P3, hidx).w / 255.0f;
				}
			}
		}

		if (mirror) {
			// newNormal + reflect + normalize, then the next castRay starts on the mirror (one copy of
			// this code serves Source.cpp:561/572-573 and Source.cpp:576-578)
			const vec3 normal = face_normal(d, xyz(N4));
			d = normalize(reflect(d, normal));
			o = h.p;
			skip = h.idx;
			if (COUNT) ctr[CTR_BOUNCE]++;
		}
		if (shade) {
			color = make3(0.f, 0.f, 0.f);                                    // Source.cpp:462
			li = 0;
			lit_any = false;
			stage = ST_SHADOW;
			if (sc.n_lights <= 0) finish = true;
		}
		if (stage == ST_SHADOW && !finish) {
			// next shadow ray (Source.cpp:470-474)
			const float4 L = sc.lights[li];
			const vec3 ray = make3(L.x - hp.x, L.y - hp.y, L.z - hp.z);
			const float dd = dot(ray, ray);
			const float s = sqrtf(dd);
			const float inv = 1.0f / s;                                      // glm::normalize: v * inversesqrt(dot(v,v))
			d = ray * inv;
			dist = s;                                                        // magnitude(ray), Source.cpp:26-28
			intensity = L.w;
			o = hp;
			skip = hidx;
			if (COUNT) ctr[CTR_SHADOW]++;
		}
		if (finish) {
			if (COUNT && lit_any) ctr[CTR_PXLIT]++;
			vec3 c = color;
			c.x = c.x * dim; c.y = c.y * dim; c.z = c.z * dim;               // Source.cpp:586; dim == 1 (exact) on the matte and miss paths
			if (!fp.shade_glsl) {                                            // (Compute.shader:342-345 stores unclamped)
				if (c.x > 1.0f) c.x = 1.0f;                                  // Source.cpp:602-610
				if (c.y > 1.0f) c.y = 1.0f;
				if (c.z > 1.0f) c.z = 1.0f;
			}
			store_pixel(fp, frame, out_index, c);
			stage = ST_IDLE;
			lit_any = false;
		}
		RT_STAMP(wp, 3);      // stage transitions, shading, store
	}

#ifdef RT_PROFILE
	if (!COUNT && fp.counters != nullptr && (threadIdx.x & 63) == 0) {
		for (int k = 0; k < 10; k++) atomicAdd(&fp.counters[k], (unsigned long long)wp.v[k]);
		for (int k = 0; k < 10; k++) atomicAdd(&fp.counters[10 + k], wp.t[k]);
	}
#endif
	if (COUNT) {
		ctr[CTR_PLANE] = lc.plane; ctr[CTR_POINT] = lc.point; ctr[CTR_EDGE] = lc.edge; ctr[CTR_BOX] = lc.box;
		flush_counters<COUNT>(fp.counters, ctr);
	}
}

// reflect(I, face_normal(I, n)) (Source.cpp:443-453, glm reflect: I - (N * dot(N, I)) * 2) with ONE dot product: N is n or -n, and
// dot(-n, I) is exactly -dot(n, I) — every product and both sums change sign and nothing else (round-to-nearest is symmetric), and
// dot(n, I) is dot(I, n) product by product — so the second dot product of the pair is the first one with newNormal's sign.
__device__ __forceinline__ vec3 reflect_off_face(vec3 I, vec3 n) {
	const float dn = dot(I, n);
	const bool keep = dn < 0.0f;                                             // newNormal: dot == 0 flips
	const vec3 N = keep ? n : n * -1.0f;
	const float dNI = keep ? dn : -dn;
	return I - (N * dNI) * 2.0f;
}

// ---- the render kernel in lockstep form (what launches use unless the state machine is asked for) ---
// The state machine above exists for divergence in bounce depth: any lane may be at any stage, so every
// pass of its loop runs the whole stage-transition and shading code.  But the pixels of an 8x8 tile
// nearly always take the SAME path (a wall is a mirror or it is not; neighbouring pixels between two
// facing mirrors bounce the same number of times), and then that generality is pure overhead: measured
// on the room scene, the 3840x2160 depth-16 mirror scene and random soups with a quarter of the
// triangles mirrors at depth 8 and 16, this form is 0-19 % faster and never slower.
// Here a wavefront walks its tiles one after the other and all 64 lanes stay in step:
// primary castRay -> (only if some lane hit a mirror) the bounce loop of Source.cpp:572-580 on those
// lanes -> per light one shadow castRay and the accumulation of Source.cpp:476-489 -> store.  Every call
// site knows its kind of ray (trace_brute<KIND>), the lights are wave-uniform (scalar loads), colour and
// shininess come in one load (plane COL), and nothing is re-decided per pass: about 40 % fewer vector
// instructions per pixel than the state machine on the room scene.  Same arithmetic, same bits.
#ifndef RT_LOCKSTEP_TILES
#define RT_LOCKSTEP_TILES 4
#endif
// PLAIN: a whole frame (no row tiles) whose u, v are division-free — the launcher's common case, compiled without the
// row mapping and the slow forms of Source.cpp:534-535, whose wave-uniform conditions otherwise sit in spilled SGPRs
// and are re-read for every tile.  Instantiated for the tree kernels and for brute force with the numerator table (launch_render).
// COUNT (tree kernels): the work counters of rt_render_stats, counted in the very traversal a frame gets.
template <int ACCEL, bool PLAIN, bool COUNT = false>
__global__ void __launch_bounds__(256, ACCEL == 2 ? RT_MIN_WAVES_LBVH : (ACCEL == 3 ? RT_MIN_WAVES_PACKET : RT_MIN_WAVES)) render_lockstep_kernel(DevScene sc, FrameParams fp, int tiles_per_wave) {
	extern __shared__ float4 s_tri[];
	float* s_num0 = (float*)s_tri;
	// (A copy of the hoisted triangles' records in LDS, read by broadcast instead of through the scalar cache, was measured SLOWER:
	// 102k triangles 3840x2160 213.6 against 199.8 us/frame — the records then sit in vector registers the kernel does not have.)
	int frame = fp.n_frames > 1 ? (int)(blockIdx.x / fp.blocks_per_frame) : 0;      // workgroup-uniform
	unsigned int block_in_frame = blockIdx.x - (unsigned int)frame * fp.blocks_per_frame;
	if constexpr (ACCEL == 2 || ACCEL == 3) {
		if (fp.band_blocks > 0) {          // XCD-aware order (rt_internal.h FrameParams::band_blocks), as in render_kernel
			const unsigned int xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
			frame = (int)(j / fp.band_blocks);
			const unsigned int band = (xcd + (unsigned int)frame) & 7u;
			block_in_frame = band * fp.band_blocks + (j - (unsigned int)frame * fp.band_blocks);
			if (block_in_frame >= fp.blocks_per_frame) return;       // (the last band is padded; no barrier in this instance)
		}
	}
	const vec3 cam = make3(fp.pose[frame][0], fp.pose[frame][1], fp.pose[frame][2]);
	if constexpr (ACCEL == 0) {
		// numerator of Source.cpp:383 for rays that start at the camera (see render_kernel)
		const int lds_stride = lds_stride_of(sc.n_tris);
		for (int i = threadIdx.x; i < lds_stride; i += blockDim.x)
			s_num0[i] = dot(xyz(sc.planes[PL_P1 * sc.stride + i]) - cam, xyz(sc.planes[PL_N * sc.stride + i]));
		__syncthreads();
	}
	const int tiles_x = (fp.width + RT_TILE_W - 1) / RT_TILE_W;
	const int tiles_y = (fp.local_rows + RT_TILE_H - 1) / RT_TILE_H;
	// A wave's run of tiles goes DOWN a column of tiles — wave w has column w % tiles_x of the band of tiles_per_wave tile rows
	// w / tiles_x.  The tiles left and right of it are then rendered by the neighbouring waves at about the same time, and the
	// 128-byte lines two of them share (a tile row is 96 bytes of RGB) are completed in the L2 before it writes them back; with
	// runs ALONG a row the halves of such a line were stored a tile's time apart and went to memory twice (measured, room 7680x4320,
	// WRITE_SIZE / frame bytes at 1 | 2 | 4 | 8 tiles per wave: 1.000 | 1.008 | 1.035 | 1.079, now 1.000 throughout;
	// scripts/write_ratio.sh).  The tree kernels too: 102k triangles, two tiles per wave, along the row | down the column 3840x2160
	// 134.0 | 132.3 us/frame with 1.048 | 1.01 of the frame's bytes written, 7680x4320 489 | 483 and 1.035 | 1.00.  (When the tile loop
	// re-derived its position from the wave's number the column form cost the packet kernel seven spilled vector registers; with
	// tx, ty and `left` carried along it has none and four spilled SGPRs fewer than the row form.)
	const unsigned int wave = __builtin_amdgcn_readfirstlane(block_in_frame * (blockDim.x >> 6) + (threadIdx.x >> 6));
	int tx, ty, left;                                                 // this wave's next tile, and how many it still has
	if constexpr (ACCEL != 2) {
		const unsigned int band_of_wave = wave / (unsigned int)tiles_x;
		tx = (int)(wave - band_of_wave * (unsigned int)tiles_x);
		ty = (int)(band_of_wave * (unsigned int)tiles_per_wave);      // (tiles_per_wave <= 1024 and wave < 2^21: no wrap)
		left = min(tiles_per_wave, tiles_y - ty);
	} else {
		// (the per-lane walk keeps runs along the row: down the column costs it four more spilled vector registers, 800x800 21.4 -> 21.8
		// us/frame; it runs one tile per wave up to 8K anyway)
		const unsigned int n_tiles = (unsigned int)tiles_x * (unsigned int)tiles_y;
		const unsigned int tile = wave * (unsigned int)tiles_per_wave;
		ty = (int)(tile / (unsigned int)tiles_x);
		tx = (int)(tile - (unsigned int)ty * (unsigned int)tiles_x);
		left = tile < n_tiles ? (int)min((unsigned int)tiles_per_wave, n_tiles - tile) : 0;
	}
	if (left <= 0) return;
	const int lx = (int)(threadIdx.x & (RT_TILE_W - 1)), lyi = (int)((threadIdx.x & 63u) / RT_TILE_W);
	WaveProf wp = {};
#ifdef RT_PROFILE
	wp.g = fp.counters;
#endif
	unsigned int ctr[CTR_N];
#pragma unroll
	for (int k = 0; k < CTR_N; k++) ctr[k] = 0;
	LaneCounters lc = { 0, 0, 0, 0 };
	// primary hits of the tree's triangles from the visibility pass (rt_vis.inl)?  >= 0: yes, with that many listed triangles to test per pixel
	int vis_list_n = -1;
	if constexpr (ACCEL == 2 || ACCEL == 3) {
		if (fp.vis != nullptr) {
			vis_list_n = __builtin_amdgcn_readfirstlane(fp.vis_count[frame]);
			if (vis_list_n > RT_VIS_LIST_MAX) vis_list_n = -1;                  // too many triangles the pass could not project: this frame walks the tree
		}
	}
	for (; left > 0; left--) {
		const int x = tx * RT_TILE_W + lx, ly = ty * RT_TILE_H + lyi;
		// (what a tile needs of the frame, the pose and the scene is read from the kernel arguments tile by tile where it is used, and
		// again at the tile's end: held through the walks it sat in spilled SGPRs — packet kernel 39 -> 18 of them, 102k triangles
		// 3840x2160 155 -> 147 us/frame; brute force +-0 on the room at 800x800, -1 % on the 4K mirrors and the 8K room)
		const FrameParams fp_top = frame_from_kernel_args(fp);
		const bool inside = (x < fp_top.width) && (ly < fp_top.local_rows);
		if constexpr (ACCEL != 2) ty++;
		else if (++tx == tiles_x) { tx = 0; ty++; }
		// ---- primary ray (Source.cpp:534-548)
		vec3 d, cam_t = cam;
		Hit h;
		HoistCtx hc;
		{
			float m[12];
			pose_from_kernel_args(fp, frame, m);
			cam_t = make3(m[0], m[1], m[2]);
			if constexpr (PLAIN) d = primary_dir_plain(x, ly, fp_top, frame, m);
			else d = primary_dir(x, global_row(ly, fp_top), fp_top, frame, m);
			if ((ACCEL == 2 || ACCEL == 3) && fp_top.hoist_mask != nullptr) {   // which hoisted triangles this pixel's direction can meet (rt_vis.inl)
				hc.by_mask = true;
				const int gy = PLAIN ? ly : global_row(ly, fp_top);
				const int cu = min((int)((float)x * fp_top.hoist_cell_x), RT_HOIST_CELLS - 1), cv = min((int)((float)gy * fp_top.hoist_cell_y), RT_HOIST_CELLS - 1);
				hc.hm = inside ? fp_top.hoist_mask[(frame * RT_HOIST_CELLS + cv) * RT_HOIST_CELLS + cu] : 0u;
			}
		}
		if (vis_list_n >= 0) h = trace_primary_vis<ACCEL == 3 ? 3 : 2, COUNT>(sc, fp, frame, vis_list_n, (long long)ly * fp_top.width + x, cam_t, d, inside, lc, wp, hc);
		else h = trace_kind<ACCEL, RAY_PRIMARY, COUNT>(sc, s_num0, cam_t, d, -1, inside, 0.f, lc, wp, hc, ((fp_top.origin_inside >> frame) & 1ull) != 0ull);
		const bool hit = inside && h.idx >= 0;
		if constexpr (COUNT) { ctr[CTR_PRIMARY] += inside; ctr[CTR_PXHIT] += hit; }
		float4 A = make_float4(0.f, 0.f, 0.f, 0.f);                          // colour, shininess of the triangle to be lit
		const DevScene sc_hit = scene_from_kernel_args(sc);
		if (hit) A = sc_hit.planes[PL_COL * sc_hit.stride + h.idx];
		bool bouncing = hit && A.w > 0.5f;                                   // Source.cpp:570
#ifdef RT_EXPERIMENT
		if (sc.debug & 8) bouncing = false;
#endif
		if constexpr (COUNT) ctr[CTR_PXMIRROR] += bouncing;
		bool shade = hit && !bouncing;                                       // Source.cpp:597
		vec3 hp = h.p;                                                       // what processLighting gets: hit.point, hit.cur
		int hidx = h.idx;
		float dim = 1.0f;
		// ---- mirrors: Source.cpp:561, 570-594 (skipped by the whole wave when no lane hit one)
		if (__builtin_amdgcn_ballot_w64(bouncing) != 0ull) {
			vec3 bo = h.p, bd = d;
			int bskip = h.idx, bounces = 0;
			const int max_bounces = frame_from_kernel_args(fp).max_bounces;
			{
				float4 N4 = make_float4(0.f, 0.f, 0.f, 0.f);
				if (bouncing) N4 = sc_hit.planes[PL_N * sc_hit.stride + h.idx];
				bd = normalize(reflect_off_face(d, xyz(N4)));                // Source.cpp:561, 572
			}
			while (__builtin_amdgcn_ballot_w64(bouncing) != 0ull) {
				if constexpr (COUNT) ctr[CTR_BOUNCE] += bouncing;
				const Hit g = trace_kind<ACCEL, RAY_BOUNCE, COUNT>(sc, s_num0, bo, bd, bskip, bouncing, 0.f, lc, wp);   // Source.cpp:573, 578
				const bool ghit = bouncing && g.idx >= 0;
				float4 GA = make_float4(0.f, 0.f, 0.f, 0.f);
				if (ghit) GA = sc_hit.planes[PL_COL * sc_hit.stride + g.idx];
				const bool cont = ghit && bounces < max_bounces && GA.w > 0.5f;      // Source.cpp:575
				if (ghit && !cont) {                                         // Source.cpp:582-586
					hp = g.p; hidx = g.idx; A = GA;
					dim = (float)(bounces * 1);
					shade = true;
				}                                                            // (a miss stays unlit: Source.cpp:590)
				if (__builtin_amdgcn_ballot_w64(cont) != 0ull) {
					float4 N4 = make_float4(0.f, 0.f, 0.f, 0.f);
					if (cont) N4 = sc_hit.planes[PL_N * sc_hit.stride + g.idx];
					if (cont) {                                              // Source.cpp:576-579
						bd = normalize(reflect_off_face(bd, xyz(N4)));
						bo = g.p;
						bskip = g.idx;
						bounces++;
					}
				}
				bouncing = cont;
			}
		}
		// ---- processLighting (Source.cpp:460-494), one light at a time for the whole wave
		vec3 color = make3(0.f, 0.f, 0.f);                                   // Source.cpp:462
		bool lit_any = false;
		if (__builtin_amdgcn_ballot_w64(shade) != 0ull) {
			const DevScene sc_lit = scene_from_kernel_args(sc);
			const c4ptr lights = c4ptr_of(sc_lit.lights);
			for (int li = 0; li < sc_lit.n_lights; li++) {
				const float4 L = lights[li];
				const vec3 ray = make3(L.x - hp.x, L.y - hp.y, L.z - hp.z);  // Source.cpp:470
				const float s = sqrtf(dot(ray, ray));                        // magnitude(ray), Source.cpp:26-28
				const float inv = 1.0f / s;                                  // glm::normalize: v * inversesqrt(dot(v,v))
				const vec3 ld = ray * inv;
				if constexpr (COUNT) ctr[CTR_SHADOW] += shade;
				const Hit sh = trace_shadow<ACCEL, COUNT>(sc, s_num0, li, make3(L.x, L.y, L.z), hp, ld, hidx, shade, s, lc, wp);   // Source.cpp:473
				if (shade && (!(sh.idx >= 0) || sh.t > s)) {                 // Source.cpp:476
					lit_any = true;
					const float f = 1.0f / (s * s) * L.w;                    // Source.cpp:488
					const float modify = div65025(f);                        // ... / 255.0 / 255.0f: binary64 divisions, rounded to binary32
					color.x = color.x + modify * A.x;                        // Source.cpp:489
					color.y = color.y + modify * A.y;
					color.z = color.z + modify * A.z;
				}
			}
		}
		if (inside) {
			vec3 c = color;
			c.x = c.x * dim; c.y = c.y * dim; c.z = c.z * dim;               // Source.cpp:586; dim == 1 (exact) on the matte and miss paths
			if (c.x > 1.0f) c.x = 1.0f;                                      // Source.cpp:602-610
			if (c.y > 1.0f) c.y = 1.0f;
			if (c.z > 1.0f) c.z = 1.0f;
			const FrameParams fp_end = frame_from_kernel_args(fp);
			store_pixel(fp_end, frame, (long long)ly * fp_end.width + x, c);        // (local rows)
		}
		if constexpr (COUNT) ctr[CTR_PXLIT] += lit_any;
	}
#ifdef RT_PROFILE
	if (!COUNT && fp.counters != nullptr && (threadIdx.x & 63) == 0)
		for (int k = 0; k < 10; k++) atomicAdd(&fp.counters[k], (unsigned long long)wp.v[k]);
#endif
	if constexpr (COUNT) {
		ctr[CTR_PLANE] = lc.plane; ctr[CTR_POINT] = lc.point; ctr[CTR_EDGE] = lc.edge; ctr[CTR_BOX] = lc.box;
		flush_counters<COUNT>(fp.counters, ctr);
	}
}

// ---- castRay for a batch of rays (tests, tools) ----------------------------------------------
template <int ACCEL>
__global__ void __launch_bounds__(256) cast_rays_kernel(DevScene sc, int n, const float* __restrict__ rays7, float* __restrict__ out6) {
	extern __shared__ float4 s_tri[];
	const int lds_stride = lds_stride_of(sc.n_tris);
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = r < n;
	vec3 o = make3(0.f, 0.f, 0.f), d = make3(0.f, 0.f, 1.f);
	int skip = -1;
	if (live) {
		const float* q = rays7 + (size_t)r * 7;
		o = make3(q[0], q[1], q[2]);
		d = make3(q[3], q[4], q[5]);
		const float sk = q[6];
		skip = (sk >= 0.0f && sk < (float)sc.n_tris) ? (int)sk : -1;
	}
	LaneCounters lc = { 0, 0, 0, 0 };
	WaveProf wp = {};
	const Hit h = trace<ACCEL, false>(sc, s_tri, lds_stride, nullptr, o, d, skip, live, false, 0.f, 0, false, lc, wp);
	if (live) {
		float* w = out6 + (size_t)r * 6;
		w[0] = h.idx >= 0 ? 1.0f : 0.0f;
		w[1] = (float)h.idx;
		w[2] = h.t;
		w[3] = h.p.x; w[4] = h.p.y; w[5] = h.p.z;
	}
}

// ---- host launchers ----------------------------------------------------------------------------

void set_pose(FrameParams& fp, int k, const float cam[3], const float trans16[16]) {
	float* p = fp.pose[k];
	for (int a = 0; a < 3; a++) {
		p[a] = cam[a];
		p[3 + a] = trans16[a];
		p[6 + a] = trans16[4 + a];
		p[9 + a] = trans16[8 + a] * 0.5f + trans16[12 + a] * 1.0f;       // Source.cpp:537-545: (m2 * 0.5 + m3 * 1), unfused
	}
}

DevScene dev_scene(const rt_scene* s, const DeviceState& d) {
	DevScene sc;
	sc.planes = (const float4*)d.planes;
	sc.lights = (const float4*)d.lights;
	sc.n_tris = (int)(s->tris.size() / RT_TRI_FLOATS);
	sc.stride = d.stride;
	sc.n_lights = (int)(s->lights.size() / 4);
	sc.n_clusters = d.n_clusters;
	sc.oct_clusters = d.oct_clusters; sc.oct_groups = d.oct_groups;
	sc.oct_cluster_pitch = d.oct_cluster_pitch; sc.oct_group_pitch = d.oct_group_pitch;
	sc.bvh_nodes = (const float4*)d.bvh_nodes;
	sc.bvh4 = (const uint4*)d.bvh4;
	sc.tri_aos = (const float4*)d.tri_aos;
	sc.bvh_root = d.bvh_root;
	sc.skip_count = d.skip_count;
	sc.skiplist = (const float4*)d.skiplist;
	sc.hoist = (const float4*)d.hoist;
	sc.hoist_count = d.hoist_count;
	sc.far_guard = d.far_guard;
	sc.box_within_guard = 1;
	for (int k = 0; k < 3; k++) if (!(fabsf(d.bb_lo[k]) < d.far_guard && fabsf(d.bb_hi[k]) < d.far_guard)) sc.box_within_guard = 0;
	sc.packet_scale = d.packet_scale;
	sc.traversal = s->traversal;
	sc.kernel_kind = s->kernel_kind;
	sc.per_lane_ok = d.per_lane_ok ? 1 : 0;
	sc.lgrid = (const LightGrid*)d.lgrid;
	sc.n_lgrid = d.n_lgrid;
	sc.debug = 0;
#ifdef RT_EXPERIMENT
	{ static const int dbg = [] { const char* e = getenv("RT355_GRID_DEBUG"); return e ? atoi(e) : 0; }(); sc.debug = dbg; }   // timing experiments: parts of grid_shadow switched off (wrong frames)
#endif
	for (int k = 0; k < 3; k++) { sc.bb_lo[k] = d.bb_lo[k]; sc.bb_hi[k] = d.bb_hi[k]; }
	return sc;
}

static bool use_packets_cols(const DevScene& sc, const float col0[3], const float col1[3], int width, int height);
bool use_packets(const DevScene& sc, const float trans16[16], int width, int height) {
	return use_packets_cols(sc, trans16, trans16 + 4, width, height);
}
static bool use_packets_cols(const DevScene& sc, const float col0[3], const float col1[3], int width, int height) {
	if (sc.skip_count <= 0) return false;
	if (!sc.per_lane_ok) return true;        // the 4-wide tree is too deep for the per-lane stack (rt_lbvh.hip tree_depths)
	if (sc.traversal != RT_TRAVERSAL_AUTO) return sc.traversal == RT_TRAVERSAL_PACKET;
#ifdef RT_EXPERIMENT
	static const int env_packet = [] { const char* e = getenv("RT355_PACKET"); return e ? atoi(e) : -1; }();
	if (env_packet >= 0) return env_packet != 0;
#endif
	const float du = sqrtf(col0[0] * col0[0] + col0[1] * col0[1] + col0[2] * col0[2]) / (float)(width > 0 ? width : 1);
	const float dv = sqrtf(col1[0] * col1[0] + col1[1] * col1[1] + col1[2] * col1[2]) / (float)(height > 0 ? height : 1);
	const float footprint = 8.0f * 0.5f * (du + dv) * sc.packet_scale;
	return footprint < 5.0f;
}

static int pick_tiles_per_wave(int accel, long long tiles) {
	// 8x8 tiles per wavefront for the STATE MACHINE (render_kernel: counting launches of brute-force scenes, RT_KERNEL_STATE_MACHINE,
	// RT_SHADE_GLSL); the lockstep kernels' run lengths are chosen in launch_render.  Measured: with two tiles a
	// lane that finishes its pixel early (a miss, a matte hit next to a 16-bounce mirror path) takes the next pixel instead of
	// idling — brute force gains 4 % on the room and stock scenes and the depth-16 mirror scene is at its best (373 us against
	// 386 with one tile, 412 with 16); longer runs only add imbalance at the tail of the launch.  The tree kernels are best with
	// one, and so are launches too small to fill the chip twice over.
	if (accel == RT_ACCEL_LBVH) return 1;
	return tiles >= 4 * 8192 ? 2 : 1;          // 8192 wave slots on 256 CUs
}

// scene_from_kernel_args() re-reads DevScene from offset 0 of the kernel-argument segment: every kernel whose traces call it must
// take the scene as its FIRST argument, by value.
template <typename F> struct first_kernel_arg;
template <typename A, typename... Rest> struct first_kernel_arg<void (*)(A, Rest...)> { typedef A type; };
#define RT_SCENE_IS_FIRST_ARG(kernel) static_assert(std::is_same<first_kernel_arg<decltype(&kernel)>::type, DevScene>::value, #kernel ": DevScene must be the first kernel argument (scene_from_kernel_args)")
template <typename F> struct second_kernel_arg;
template <typename A, typename B, typename... Rest> struct second_kernel_arg<void (*)(A, B, Rest...)> { typedef B type; };
// (frame_from_kernel_args / pose_from_kernel_args: FrameParams right behind it, as RenderKernelArgs lays them out)
static_assert(std::is_same<second_kernel_arg<decltype(&render_lockstep_kernel<0, false, false>)>::type, FrameParams>::value &&
              std::is_same<second_kernel_arg<decltype(&render_lockstep_kernel<3, true, false>)>::type, FrameParams>::value,
              "render_lockstep_kernel: FrameParams must be the second kernel argument (frame_from_kernel_args)");
RT_SCENE_IS_FIRST_ARG((render_kernel<0, false>));
RT_SCENE_IS_FIRST_ARG((render_kernel<3, false>));
RT_SCENE_IS_FIRST_ARG((render_lockstep_kernel<0, false, false>));
RT_SCENE_IS_FIRST_ARG((render_lockstep_kernel<3, true, false>));
RT_SCENE_IS_FIRST_ARG((cast_rays_kernel<0>));
RT_SCENE_IS_FIRST_ARG((cast_rays_kernel<3>));

int launch_render(const DevScene& sc, const FrameParams& fp_in, int accel, int render_flags, bool count, hipStream_t stream) {
	(void)render_flags;
	FrameParams fp = fp_in;
	if (fp.width <= 0 || fp.local_rows <= 0) return RT_OK;
	const int n_frames = fp.n_frames < 1 ? 1 : fp.n_frames;
	const long long tiles = (long long)((fp.width + RT_TILE_W - 1) / RT_TILE_W) * ((fp.local_rows + RT_TILE_H - 1) / RT_TILE_H);
	if (tiles * 64 > 0xf0000000LL) { set_error("rt_render: %dx%d needs more than 2^32 work items", fp.width, fp.local_rows); return RT_ERR_LIMIT; }
	// (the lockstep kernel walks its tiles one after the other, so longer runs only amortise the per-workgroup
	// set-up; brute force, measured (scripts/tiles_sweep.sh), us/frame at 1 | 2 | 4 | 8 | 16 tiles: room 800x800 x32 frames
	// 14.1 | 13.5 | 13.3 | 14.1, 1600x1600 - | - | 47.4 | 45.2, mirrors 3840x2160 255 | 242 | 236 | 233 | 239, room 7680x4320 653 | 613 | 585 | 567 | 573)
	// AUTO is the lockstep form: measured faster than, or equal to, the state machine on every scene tried, up to
	// depth 16 and with bounce depths that diverge inside a tile
	const bool lockstep_form = sc.kernel_kind != RT_KERNEL_STATE_MACHINE && !fp.shade_glsl;   // (RT_SHADE_GLSL lives in render_kernel only)
	const bool lockstep = !count && lockstep_form;
	int tpw = fp.tiles_per_wave > 0 ? fp.tiles_per_wave : (lockstep && accel != RT_ACCEL_LBVH ? (tiles * n_frames >= 128 * 8192 ? 2 * RT_LOCKSTEP_TILES : (tiles * n_frames >= 16 * 8192 ? RT_LOCKSTEP_TILES : (tiles * n_frames >= 4 * 8192 ? 2 : 1)))
	                                                                                           : pick_tiles_per_wave(accel, tiles * n_frames));
	// LBVH: walk the tree once per wavefront ("packet") when an 8x8 tile of rays is coherent against the
	// scene's triangles, per lane otherwise.  Footprint of a tile at the far side of the scene in units
	// of the mean triangle size; measured on the 100k-triangle scene (round 3: clustered tree, shadow rays through the light's
	// grid, so only primary and bounce rays differ), us/frame packets | per lane: 300x300 (footprint 12.3) 17.6 | 7.7, 400 (9.2)
	// 18.4 | 10.9, 500 (7.4) 20.4 | 15.5, 600 (6.1) 21.0 | 18.7, 800 (4.6) 26.4 | 29.0, 1000 (3.7) 34.3 | 41.6, 1200 (3.1) 44.4 | 55.9,
	// 1600 (2.3) 67.9 | 90.5, 2400 (1.5) 127.5 | 183.5, 3840x2160 (1.3) 172.1 | 244.7: packets below a footprint of 5.5.  Re-measured
	// with 64 frames per launch and workgroups of one wavefront: 300 11.4 | 5.2, 400 11.3 | 7.5, 500 11.9 | 10.3, 600 (6.1) 13.5 | 13.3,
	// 700 (5.3) 16.0 | 17.2, 800 18.9 | 21.3, 1000 24.9 | 30.3: packets below 6.  And once more after the per-lane walk lost its exit bound
	// (and with it its spilled registers): 500 11.9 | 9.65, 600 13.5 | 12.4, 700 (5.3) 16.2 | 16.0, 800 (4.6) 19.0 | 20.1, 900 22.0 | 24.0,
	// 1000 25.0 | 28.5: packets below 5.
	DevScene sc_sel = sc;
	if (accel == RT_ACCEL_LBVH) {
		bool packets = use_packets_cols(sc, fp.pose[0] + 3, fp.pose[0] + 6, fp.width, fp.height);
		// A SHORT launch walks per lane even where packets win at steady state: its wavefronts all start with cold caches, and the
		// packet walk's chains of dependent scalar loads then cost far more than the per-lane walk's vector loads (the scalar cache
		// serves few misses at a time).  Measured, 102k triangles, us per launch packets | per lane: one 800x800 frame 338 | 151, two
		// 282 | 149, four 305 | 193, eight 366 | 274 (16 and more: packets); one 1600x1600 frame 252 | 185, two 253 | 241, four 371 | 399.
		// (Counting launches keep the steady-state choice: they count for the batched launches that are timed.)
		if (packets && !count && sc.traversal == RT_TRAVERSAL_AUTO && sc.per_lane_ok && tiles * n_frames < 90000) packets = false;
		if (!packets) sc_sel.skip_count = 0;
	}
	// (tree kernels: one tile, except long launches of large frames, where the next tile down the column walks nearly the same nodes
	// while they are still in the caches — 102k triangles, us/frame at 1 | 2 | 3 | 4 | 6 tiles: 800x800 19.0 | 19.8, 1200x1200 32.5 | 32.1,
	// 1600x1600 50.3 | 49.1, 3840x2160 140.7 | 132.0 | 130.8 | 130.3 | 130.5, 7680x4320 541 | 492.5 | 483.5 | 479.5 | 476.6; round 2's kernels
	// had 3840x2160 at 214.6 | 222.7 and the rule began at 400k tiles a frame)
	const bool per_lane = accel == RT_ACCEL_LBVH && sc_sel.skip_count == 0;
	if (fp.tiles_per_wave <= 0 && lockstep && accel == RT_ACCEL_LBVH && !per_lane && tiles * n_frames >= 400000)
		tpw = (tiles >= 100000 && tiles * n_frames >= 1600000) ? 4 : (tiles >= 20000 ? 2 : 1);
	if (fp.tiles_per_wave <= 0 && lockstep && per_lane && tiles >= 400000) tpw = 2;      // (round 2's rule; its runs go along the row)
	// (the lockstep kernels run a wave's tiles down a column: whole columns of ceil(tile rows / tpw) bands; the per-lane walk's and the
	// state machine's go along rows)
	const long long tiles_x = (fp.width + RT_TILE_W - 1) / RT_TILE_W, tiles_y = (fp.local_rows + RT_TILE_H - 1) / RT_TILE_H;
	const bool column_runs = (lockstep || (count && lockstep_form && accel == RT_ACCEL_LBVH)) && !per_lane;   // (counting launches of the tree kernels: same instance family)
	const long long waves = column_runs ? tiles_x * ((tiles_y + tpw - 1) / tpw) : (tiles + tpw - 1) / tpw;
	// Workgroups of ONE wavefront for the tree kernels (no barrier in them; the per-lane walk's LDS stack is laid out for 64 lanes,
	// RT_STACK_PITCH): a workgroup's resources are released when its last wave ends, and the walks of four tiles differ in
	// length — measured, 102k triangles, packets, workgroups of 256 | 128 | 64 threads: 800x800 22.1 | 21.2 | 19.6 us/frame
	// (wavefronts resident per SIMD 5.2 of 8 with 256), 3840x2160 155.1 | 154.4 | 153.3.  Brute force fills a numerator table in LDS
	// per workgroup (one value per triangle) and gains about 1 % — two runs each, 256 | 64 threads: room 800x800 10.66, 10.73 | 10.53,
	// 10.55, stock 7.49, 7.53 | 7.41, 7.44, mirrors 4K 213.4, 214.3 | 212.2, 212.8, room 8K 518.6, 520.6 | 514.6, 516.6 — so scenes of up
	// to 256 triangles (a table of 1 KB) get one-wave workgroups too; larger forced-brute scenes keep 256.
#ifdef RT_EXPERIMENT
	static const int env_block = [] { const char* e = getenv("RT355_BLOCK_THREADS"); int v = e ? atoi(e) : 0; return (v == 64 || v == 128 || v == 256) ? v : 0; }();
	const int threads = (accel == RT_ACCEL_LBVH) ? 64 : (env_block ? env_block : (sc.n_tris <= 256 ? 64 : 256));
#else
	const int threads = (accel == RT_ACCEL_LBVH || sc.n_tris <= 256) ? 64 : 256;
#endif
	const int wpb = threads / 64;
	const long long blocks = (waves + wpb - 1) / wpb;
	if (blocks * n_frames > 0x7fffffffLL) { set_error("rt_render: frame too large (%lld workgroups)", blocks * n_frames); return RT_ERR_LIMIT; }
	fp.n_frames = n_frames;
	fp.origin_inside = 0ull;                // which frames' cameras lie inside the (padded) scene box: their primary rays need no exit bound
	for (int f = 0; f < n_frames && f < 64; f++) {
		bool in = true;
		for (int k = 0; k < 3; k++) in = in && fp.pose[f][k] > sc.bb_lo[k] && fp.pose[f][k] < sc.bb_hi[k];
		if (in) fp.origin_inside |= 1ull << f;
	}
	fp.blocks_per_frame = (unsigned int)blocks;
	dim3 grid((unsigned)(blocks * n_frames)), block((unsigned)threads);
	// XCD-aware workgroup order for multi-frame LBVH launches (rt_internal.h FrameParams::band_blocks).
	// Measured on the 100k-triangle scene, 32 frames per launch, us/frame without | with: 800x800 76 | 72
	// (L2 misses per wavefront 39 -> 28), 1000x1000 101 | 91, 1600x1600 168 | 161, 2400x2400 305 | 287;
	// wide and very large frames did not gain with round 2's kernels (3200x1800 295 | 294, 3840x2160 387 | 397, 7680x4320 1364 | 1365:
	// a band of full-width rows already spans most of the scene) and had a size limit; with round 3's (a third of the time per
	// frame, the same misses) they do — 3840x2160 153.2 | 149.2 and 153.6 | 150.0, 7680x4320 552.8 | 549.6 and 553.8 | 550.4, 800x800
	// 19.6 | 19.5 — so every size gets the bands.  RECTANGLES of tiles per XCD (4 x 2, nearer to square than bands of rows) were
	// slower: 3840x2160 157.7, 7680x4320 569.8, 800x800 19.7 (scripts/bands_sweep.sh, experiment build); so was an XCD that stays on
	// its band for 2 | 4 | 8 frames before moving on (its L2 would keep that part of the scene): 3840x2160 149.7 -> 151.9 | 158.8 |
	// 165.7, 7680x4320 551.6 -> 607.7 | 666.1 | 736.9 — bands differ in cost, and the rotation with every frame is what evens them out.
	fp.band_blocks = 0;
#ifdef RT_EXPERIMENT
	static const int env_bands = [] { const char* e = getenv("RT355_XCD_BANDS"); return e ? atoi(e) : -1; }();   // 0 off, 1 always
	const bool bands = env_bands >= 0 ? env_bands != 0 : true;
#else
	const bool bands = true;
#endif
#ifdef RT_EXPERIMENT
	static const int band_min_frames = [] { const char* e = getenv("RT355_BAND_MIN_FRAMES"); return e ? atoi(e) : 8; }();
#else
	const int band_min_frames = 8;
#endif
	if (accel == RT_ACCEL_LBVH && !count && n_frames >= band_min_frames && bands) {
		const long long band_blocks = (blocks + 7) / 8;
		if (band_blocks * 8 * n_frames <= 0x7fffffffLL) {
			fp.band_blocks = (unsigned int)band_blocks;
			grid = dim3((unsigned)(band_blocks * 8 * n_frames));
		}
	}
	// render_lockstep_kernel<ACCEL, PLAIN>: measured +1.5 % for packets; the brute-force and per-lane instances got 2-3 % SLOWER with it in
	// round 2 (register allocation).  Measured again at the end of round 3 for brute force (scenes with the numerator table), general |
	// plain: room 800x800 10.19 | 10.10 us/frame, stock 7.14 | 7.07, mirrors 3840x2160 204.5 | 204.0, room 7680x4320 482.4 | 478.8 —
	// so ACCEL 0 has the plain instance too, and so has the per-lane walk (400x400 6.95 | 6.91, 800x800 19.83 | 19.68, a lone 800x800 frame
	// 119.1 | 118.4); forced brute force on larger scenes keeps the general form.
	const bool plain = fp.uv_fast && fp.tile_stride == 1 && fp.tile_first == 0;
	if (accel == RT_ACCEL_LBVH) {
		const size_t lds = (size_t)RT_STACK_LDS * RT_STACK_PITCH * sizeof(int);   // [entry][lane]
		const size_t lds_h = 0;
		// (counting launches: in the kernel form and traversal the frame really gets)
		if (count && sc_sel.skip_count > 0 && lockstep_form) hipLaunchKernelGGL((render_lockstep_kernel<3, false, true>), grid, block, lds_h, stream, sc_sel, fp, tpw);
		else if (count && lockstep_form) hipLaunchKernelGGL((render_lockstep_kernel<2, false, true>), grid, block, lds + lds_h, stream, sc_sel, fp, tpw);
		else if (count && sc_sel.skip_count > 0) hipLaunchKernelGGL((render_kernel<3, true>), grid, block, 0, stream, sc_sel, fp, tpw);
		else if (count) hipLaunchKernelGGL((render_kernel<2, true>), grid, block, lds, stream, sc_sel, fp, tpw);
		else if (sc_sel.skip_count > 0 && lockstep && plain) hipLaunchKernelGGL((render_lockstep_kernel<3, true>), grid, block, lds_h, stream, sc_sel, fp, tpw);
		else if (sc_sel.skip_count > 0 && lockstep) hipLaunchKernelGGL((render_lockstep_kernel<3, false>), grid, block, lds_h, stream, sc_sel, fp, tpw);
		else if (sc_sel.skip_count > 0) hipLaunchKernelGGL((render_kernel<3, false>), grid, block, 0, stream, sc_sel, fp, tpw);
		else if (lockstep && plain) hipLaunchKernelGGL((render_lockstep_kernel<2, true>), grid, block, lds + lds_h, stream, sc_sel, fp, tpw);
		else if (lockstep) hipLaunchKernelGGL((render_lockstep_kernel<2, false>), grid, block, lds + lds_h, stream, sc_sel, fp, tpw);
		else hipLaunchKernelGGL((render_kernel<2, false>), grid, block, lds, stream, sc_sel, fp, tpw);
	} else if (sc.n_tris <= RT_LDS_MAX_TRIS) {
		const size_t lds = (size_t)lds_stride_of(sc.n_tris) * sizeof(float);
		if (count) hipLaunchKernelGGL((render_kernel<0, true>), grid, block, lds, stream, sc_sel, fp, tpw);
		else if (lockstep && plain) hipLaunchKernelGGL((render_lockstep_kernel<0, true>), grid, block, lds, stream, sc_sel, fp, tpw);
		else if (lockstep) hipLaunchKernelGGL((render_lockstep_kernel<0, false>), grid, block, lds, stream, sc_sel, fp, tpw);
		else hipLaunchKernelGGL((render_kernel<0, false>), grid, block, lds, stream, sc_sel, fp, tpw);
	} else {
		if (count) hipLaunchKernelGGL((render_kernel<1, true>), grid, block, 0, stream, sc_sel, fp, tpw);
		else if (lockstep) hipLaunchKernelGGL((render_lockstep_kernel<1, false>), grid, block, 0, stream, sc_sel, fp, tpw);
		else hipLaunchKernelGGL((render_kernel<1, false>), grid, block, 0, stream, sc_sel, fp, tpw);
	}
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return hip_fail(e, "render kernel launch");
	return RT_OK;
}

int launch_cast_rays(const DevScene& sc, int accel, int n, const float* d_rays7, float* d_out6, hipStream_t stream) {
	if (n <= 0) return RT_OK;
	const dim3 grid((unsigned)((n + 255) / 256)), block(256);
	if (accel == RT_ACCEL_LBVH) {
		DevScene sc_sel = sc;
		if (sc.traversal != RT_TRAVERSAL_PACKET && sc.per_lane_ok) sc_sel.skip_count = 0;   // ray batches are per lane unless packets are asked for
		if (sc_sel.skip_count > 0) hipLaunchKernelGGL((cast_rays_kernel<3>), grid, block, 0, stream, sc_sel, n, d_rays7, d_out6);
		else hipLaunchKernelGGL((cast_rays_kernel<2>), dim3((unsigned)((n + 63) / 64)), dim3(64), (size_t)RT_STACK_LDS * RT_STACK_PITCH * sizeof(int), stream, sc_sel, n, d_rays7, d_out6);   // (one wavefront per workgroup: RT_STACK_PITCH)
	} else if (sc.n_tris <= RT_LDS_MAX_TRIS) {
		const size_t lds = (size_t)lds_stride_of(sc.n_tris) * sizeof(float);
		hipLaunchKernelGGL((cast_rays_kernel<0>), grid, block, lds, stream, sc, n, d_rays7, d_out6);
	} else {
		hipLaunchKernelGGL((cast_rays_kernel<1>), grid, block, 0, stream, sc, n, d_rays7, d_out6);
	}
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return hip_fail(e, "cast_rays kernel launch");
	return RT_OK;
}

} // namespace rt
