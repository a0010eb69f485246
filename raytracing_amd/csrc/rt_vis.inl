// rt_vis.inl — primary visibility by rasterisation (device side + launcher), included by rt_kernels.hip.
//
// The primary rays of a frame (Source.cpp:534-548) all start at the camera: seen from there a triangle is a small patch
// of pixels, and only the rays of those pixels can hit it.  Walking the tree once per pixel (or once per 8x8 tile)
// asks the opposite question — which triangles does this ray meet — and at 800x800 a tile's packet walk tests each of
// its ~12 triangles on all 64 lanes although ~8 of them can hit it.  For LBVH scenes a pass of its own therefore
// resolves the primary hits of the tree's triangles the other way round, one lane per (frame, triangle):
//   * the triangle's padded box (the tree's leaf box: every hit point the reference can accept lies in it) is
//     projected onto the view plane — (u t, v t, t) = M^-1 (P - camera) with M the columns of Source.cpp:537-545 — and
//     the pixels whose sample lies in the projected box's bounding rectangle, widened by the rounding margin of
//     rt_lgrid.hip (m = 4e-6 S / D + 2e-5 in u, v units), are the candidates;
//   * each candidate pixel gets the reference's own test: its direction by primary_dir (the kernel's own function, same
//     bits), t = dot(p1 - camera, n) / dot(dir, n), eps < t < 1e9, the three edge terms (sign filter first);
//   * an accepted hit goes into the frame's visibility buffer with a 64-bit atomicMin of (bits of t : triangle index):
//     t > 0, so the smallest key is the smallest t and, among equal t, the lowest index — castRay's result
//     (Source.cpp:391: strict <, in index order).
// The render kernel then reads its pixel's entry instead of walking the tree, tests the scene-spanning triangles kept
// out of the tree and the frame's "list" against it (ties to the lower index) and goes on to shading; bounce rays
// still walk the tree.  What cannot be projected goes to that per-frame list, tested for every pixel: boxes nearer
// the camera than S/250 or wider than half their distance (a box with a corner behind the camera plane that is NOT
// that near cannot reach into the 90-degree view pyramid and is dropped; one entirely behind is dropped anyway), and
// boxes covering more than RT_VIS_MAX_PX pixels.  More than RT_VIS_LIST_MAX of them and the frame keeps the tree walk
// (the render kernel checks the count).  The pass only runs for whole frames, near-orthonormal view matrices
// (rt_camera_trans gives those; the pyramid argument needs them) and cameras within the commit's far_guard.
#pragma once

namespace rt {

#define RT_VIS_MAX_PX 16384

// One candidate pixel of triangle i: the reference's test (Source.cpp:383-396) for the primary ray through (x, y); an accepted
// hit goes into the frame's buffer `vf`.
struct VisTri { vec3 nn, p1, p2, p3; float num; float4 M0, M1, M2; };
template <bool COUNT>
__device__ __forceinline__ void vis_test_pixel(const DevScene& sc, const FrameParams& fp, int frame, int i, int x, int y, vec3 cam, const VisTri& T,
                                               unsigned long long* __restrict__ vf, unsigned int& c_plane, unsigned int& c_point, unsigned int& c_edge) {
	const vec3 d = primary_dir(x, y, fp, frame);                        // Source.cpp:534-545
	const float t = T.num / dot(d, T.nn);                               // Source.cpp:383
	const bool cand = t > RT_EPS_BELOW && t < 1e9f;                     // Source.cpp:385, 391 (closest.distance starts at 1e9)
	if (COUNT) { c_plane++; c_point += !(t < 0.0f); c_edge += cand; }
	if (!cand) return;
	const vec3 I = make3(cam.x + t * d.x, cam.y + t * d.y, cam.z + t * d.z);   // Source.cpp:389
	const vec3 w = I - T.p1;
	bool decided = false, pass = false;
	if (!COUNT) {                                                       // edge_filter(), one lane at a time
		const float w1 = (fabsf(w.x) + fabsf(w.y)) + fabsf(w.z);
		const float B0 = __builtin_fmaf(T.M0.w, w1, T.M2.w);
		const float Bd = fmaxf(B0, 1e-20f);
		const float s0 = __builtin_fmaf(T.M0.x, w.x, __builtin_fmaf(T.M0.y, w.y, T.M0.z * w.z));
		const float s1 = __builtin_fmaf(T.M1.x, w.x, __builtin_fmaf(T.M1.y, w.y, __builtin_fmaf(T.M1.z, w.z, -T.M1.w)));
		const float s2 = __builtin_fmaf(T.M2.x, w.x, __builtin_fmaf(T.M2.y, w.y, T.M2.z * w.z));
		const float least = fminf(fminf(s0, s1), s2);
		pass = least > Bd;
		decided = (pass || -least > Bd) && B0 < 1e20f;
	}
	if (!decided) {                                                     // the reference's own edge terms
		const float4 E0 = sc.planes[PL_E0 * sc.stride + i], E1 = sc.planes[PL_E1 * sc.stride + i], E2 = sc.planes[PL_E2 * sc.stride + i];
		pass = dot(T.nn, cross(xyz(E0), w)) > 0.0f;                     // Source.cpp:392
		if (pass) {
			if (COUNT) c_edge++;
			pass = dot(T.nn, cross(xyz(E1), I - T.p2)) > 0.0f;          // Source.cpp:393
			if (pass) {
				if (COUNT) c_edge++;
				pass = dot(T.nn, cross(xyz(E2), I - T.p3)) > 0.0f;      // Source.cpp:394
			}
		}
	}
	if (!pass) return;
	const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned int)i;
	unsigned long long* slot = vf + ((long long)y * fp.width + x);
	if (key < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, key);   // Source.cpp:391, 396
}

__device__ __forceinline__ VisTri vis_load_tri(const DevScene& sc, int i, vec3 cam) {
	VisTri T;
	T.nn = xyz(sc.planes[PL_N * sc.stride + i]);
	T.p1 = xyz(sc.planes[PL_P1 * sc.stride + i]);
	T.p2 = xyz(sc.planes[PL_P2 * sc.stride + i]);
	T.p3 = xyz(sc.planes[PL_P3 * sc.stride + i]);
	T.num = dot(T.p1 - cam, T.nn);                                      // Source.cpp:383, the part no pixel changes
	T.M0 = sc.planes[PL_M0 * sc.stride + i]; T.M1 = sc.planes[PL_M1 * sc.stride + i]; T.M2 = sc.planes[PL_M2 * sc.stride + i];
	return T;
}

// One lane per (frame, triangle).  Rectangles of up to RT_VIS_LANE_PX pixels are tested by their own lane; larger ones, one after
// the other, by all 64 lanes of the wavefront (a triangle near the camera covers thousands of pixels: one lane would keep its wave
// waiting for milliseconds).
#define RT_VIS_LANE_PX 24
template <bool COUNT>
__global__ void __launch_bounds__(256) vis_raster_kernel(DevScene sc, FrameParams fp, const unsigned char* __restrict__ skip,
                                                        unsigned long long* __restrict__ vis, int* __restrict__ count, int* __restrict__ list) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const int frame = blockIdx.y;
	unsigned int c_plane = 0, c_point = 0, c_edge = 0;
	const float* P = fp.pose[frame];
	const vec3 cam = make3(P[0], P[1], P[2]);
	unsigned long long* const vf = vis + (long long)frame * fp.vis_stride;
	bool raster = false;
	int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
	if (i < sc.n_tris && !skip[i]) {
		const vec3 m0 = make3(P[3], P[4], P[5]), m1 = make3(P[6], P[7], P[8]), mc = make3(P[9], P[10], P[11]);
		// rows of the inverse of [m0 m1 mc]
		vec3 r0 = cross(m1, mc), r1 = cross(mc, m0), r2 = cross(m0, m1);
		const float idet = 1.0f / dot(m0, r0);
		r0 = r0 * idet; r1 = r1 * idet; r2 = r2 * idet;
		const float4 A = sc.planes[PL_P1 * sc.stride + i], B = sc.planes[PL_P2 * sc.stride + i], C = sc.planes[PL_P3 * sc.stride + i];
		float lo[3], hi[3];
		lo[0] = fminf(A.x, fminf(B.x, C.x)); hi[0] = fmaxf(A.x, fmaxf(B.x, C.x));
		lo[1] = fminf(A.y, fminf(B.y, C.y)); hi[1] = fmaxf(A.y, fmaxf(B.y, C.y));
		lo[2] = fminf(A.z, fminf(B.z, C.z)); hi[2] = fmaxf(A.z, fmaxf(B.z, C.z));
		{   // rt_lbvh.hip tri_box_padded
			const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
			const float p = fp.vis_pad * 0.1f + 1e-3f * sqrtf(dx * dx + dy * dy + dz * dz);
			for (int k = 0; k < 3; k++) { lo[k] -= p; hi[k] += p; }
		}
		const float camv[3] = { cam.x, cam.y, cam.z };
		float d2 = 0.f, e2 = 0.f, big = 0.f;
		for (int k = 0; k < 3; k++) {
			const float dd = fmaxf(fmaxf(lo[k] - camv[k], camv[k] - hi[k]), 0.0f);
			d2 += dd * dd;
			e2 += (hi[k] - lo[k]) * (hi[k] - lo[k]);
			big = fmaxf(big, fmaxf(fabsf(camv[k]), fmaxf(fabsf(sc.bb_lo[k]), fabsf(sc.bb_hi[k]))));
		}
		const float D = sqrtf(d2), S = 4.0f * big;
		const float m = 4e-6f * S / D + 2e-5f;
		const bool near = !(D > 0.0f) || !(sqrtf(e2) <= 0.5f * D) || !(m <= 1e-3f);
		float tmin = __builtin_huge_valf(), tmax = -__builtin_huge_valf();
		float umin = tmin, umax = tmax, vmin = tmin, vmax = tmax;
		for (int c = 0; c < 8; c++) {
			const vec3 q = make3(((c & 1) ? hi[0] : lo[0]) - cam.x, ((c & 2) ? hi[1] : lo[1]) - cam.y, ((c & 4) ? hi[2] : lo[2]) - cam.z);
			const float t = dot(r2, q);
			tmin = fminf(tmin, t); tmax = fmaxf(tmax, t);
			const float it = 1.0f / t;
			const float u = dot(r0, q) * it, v = dot(r1, q) * it;
			umin = fminf(umin, u); umax = fmaxf(umax, u); vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
		}
		bool to_list = false;
		if (!(tmax > 0.0f)) { /* all of the box behind the camera: no hit point with t > 0 */ }
		else if (near) to_list = true;
		else if (!(tmin > 0.0f)) { /* a far box straddling the camera plane: outside the view pyramid (header) */ }
		else {
			const float W = (float)fp.width, H = (float)fp.height;
			// sample x has u = (x - W/2) / W: x = u W + W/2
			const float fx0 = ceilf((umin - m) * W + 0.5f * W - 0.01f), fx1 = floorf((umax + m) * W + 0.5f * W + 0.01f);
			const float fy0 = ceilf((vmin - m) * H + 0.5f * H - 0.01f), fy1 = floorf((vmax + m) * H + 0.5f * H + 0.01f);
			x0 = (int)fmaxf(fx0, 0.0f); x1 = (int)fminf(fx1, W - 1.0f);
			y0 = (int)fmaxf(fy0, 0.0f); y1 = (int)fminf(fy1, H - 1.0f);
			if (x1 >= x0 && y1 >= y0) {
				if ((long long)(x1 - x0 + 1) * (y1 - y0 + 1) > RT_VIS_MAX_PX) to_list = true;
				else raster = true;
			}
		}
		if (to_list) {
			const int k = atomicAdd(&count[frame], 1);
			if (k < RT_VIS_LIST_MAX) list[frame * RT_VIS_LIST_MAX + k] = i;
		}
	}
	const int npx = raster ? (x1 - x0 + 1) * (y1 - y0 + 1) : 0;
	if (npx > 0 && npx <= RT_VIS_LANE_PX) {
		const VisTri T = vis_load_tri(sc, i, cam);
		for (int y = y0; y <= y1; y++)
			for (int x = x0; x <= x1; x++) vis_test_pixel<COUNT>(sc, fp, frame, i, x, y, cam, T, vf, c_plane, c_point, c_edge);
	}
	// the larger rectangles of this wavefront, by all its lanes
	unsigned long long bigm = __builtin_amdgcn_ballot_w64(npx > RT_VIS_LANE_PX);
	const int lane = (int)(threadIdx.x & 63u);
	while (bigm) {
		const int src = __builtin_ctzll(bigm);
		bigm &= bigm - 1ull;
		const int bi = __builtin_amdgcn_readlane(i, src);
		const int bx0 = __builtin_amdgcn_readlane(x0, src), bx1 = __builtin_amdgcn_readlane(x1, src);
		const int by0 = __builtin_amdgcn_readlane(y0, src), by1 = __builtin_amdgcn_readlane(y1, src);
		const VisTri T = vis_load_tri(sc, bi, cam);
		const int w = bx1 - bx0 + 1;
		int x = bx0 + lane, y = by0;
		while (x > bx1) { x -= w; y++; }
		while (y <= by1) {
			vis_test_pixel<COUNT>(sc, fp, frame, bi, x, y, cam, T, vf, c_plane, c_point, c_edge);
			x += 64;
			while (x > bx1) { x -= w; y++; }
		}
	}
	if (COUNT && fp.counters != nullptr) {
		unsigned int v[3] = { c_plane, c_point, c_edge };
		const int slot[3] = { CTR_PLANE, CTR_POINT, CTR_EDGE };
		for (int k = 0; k < 3; k++) {
			unsigned int s = v[k];
			for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
			if ((threadIdx.x & 63) == 0 && s) atomicAdd(&fp.counters[slot[k]], (unsigned long long)s);
		}
	}
}

// castRay for the primary ray of a pixel whose tree hit the visibility pass has resolved (lockstep kernel): that hit, then the
// frame's list and the triangles kept out of the tree, ties to the lower index.  `pixel`: index into the frame (local rows).
template <int ACCEL, bool COUNT>
__device__ __forceinline__ Hit trace_primary_vis(const DevScene& sc_arg, const FrameParams& fp, int frame, int n_list, long long pixel, vec3 o, vec3 d, bool live,
                                                 LaneCounters& lc, WaveProf& wp, HoistCtx hc) {
	DevScene sc = sc_arg;
	if constexpr (ACCEL == 3) sc = scene_from_kernel_args(sc_arg);
	Hit h;
	h.t = 1e9f;
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	unsigned long long key = ~0ull;
	if (live) key = fp.vis[(long long)frame * fp.vis_stride + pixel];
	const vec3 ds = make3(fabsf(d.x) < 1e-30f ? 1e-30f : d.x, fabsf(d.y) < 1e-30f ? 1e-30f : d.y, fabsf(d.z) < 1e-30f ? 1e-30f : d.z);
	const vec3 inv = make3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
	const vec3 oi = make3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
	// (the launcher only runs the pass for cameras within far_guard: the boxes are trusted)
	const float ex = fminf(fminf(fmaxf(__builtin_fmaf(sc.bb_lo[0], inv.x, oi.x), __builtin_fmaf(sc.bb_hi[0], inv.x, oi.x)),
	                             fmaxf(__builtin_fmaf(sc.bb_lo[1], inv.y, oi.y), __builtin_fmaf(sc.bb_hi[1], inv.y, oi.y))),
	                       fmaxf(__builtin_fmaf(sc.bb_lo[2], inv.z, oi.z), __builtin_fmaf(sc.bb_hi[2], inv.z, oi.z)));
	float lim = fminf(fminf(ex + fabsf(ex) * 1e-4f, 1e9f), 1e9f);
	if (key != ~0ull) {
		h.t = __uint_as_float((unsigned int)(key >> 32));
		h.idx = (int)(unsigned int)key;
		h.p = make3(o.x + h.t * d.x, o.y + h.t * d.y, o.z + h.t * d.z);               // Source.cpp:389, as the pass computed it
		lim = h.t;
	}
	if (!live) lim = -1.0f;
	float lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	if (n_list > 0) {
		const c4ptr aos = c4ptr_of(sc.tri_aos);
		const float eps_lo = RT_EPS_BELOW * (1.0f - RT_PREFILTER_REL);
		for (int k = 0; k < n_list; k++) {
			const int i = fp.vis_list[frame * RT_VIS_LIST_MAX + k];
			packet_leaf<COUNT>(aos.offset_floats((unsigned int)i * (RT_AOS_F4 * 4u)), i, o, d, -1, false, eps_lo, h, lim, lim_hi, lc, wp);
		}
	}
#ifdef RT_EXPERIMENT
	if (sc.debug & 4) return h;
#endif
	if (sc.hoist_count > 0) {
		if (hc.by_mask) hoisted_by_mask<COUNT>(sc, hc.hm, o, d, -1, false, h, lim, lim_hi, lc, wp);
		else hoisted_tests<COUNT>(sc, o, d, inv, oi, -1, live, false, h, lim, lim_hi, lc, wp);
	}
	return h;
}

// ---- the hoisted triangles by direction, for primary rays -------------------------------------------------------------------
// The scene-spanning triangles kept out of the tree (DevScene::hoist: the walls of the reference's room) are what most primary rays
// end on, and testing them behind their six boxes was the largest single item of a frame of the 102 412-triangle scene (measured:
// 7.4 of 32 us at 800x800, 77 of 210 at 3840x2160).  Per frame of a launch this kernel marks, for each of RT_HOIST_CELLS x
// RT_HOIST_CELLS cells of the view, which hoisted triangles a ray through the cell can meet at all — a cone-against-box test as
// in rt_lgrid.hip hoisted_masks: the cell's rays are the cone t > 0, u0 <= a/t <= u1, v0 <= b/t <= v1 with (a, b, t) = M^-1 (P - camera),
// linear inequalities in P; a padded box all of whose corners break one of them cannot be met.  The render kernel then tests
// the one to three triangles of its pixel's cell (hoisted_by_mask) — no boxes.  Cells are widened by 2e-3 in u, v (far above the
// roundings involved); a box nearer the camera than S/250 is marked everywhere.
// One workgroup per frame: thread = (cell, one of RT_HOIST_GROUPS shares of the triangle list); most (cell, triangle) pairs end at
// the first check — all three corners outside one of the cone's planes — and only the rest are clipped.
#define RT_HOIST_GROUPS 4
__global__ void __launch_bounds__(RT_HOIST_CELLS * RT_HOIST_CELLS * RT_HOIST_GROUPS) hoist_mask_kernel(DevScene sc, FrameParams fp, unsigned int* __restrict__ masks) {
	__shared__ unsigned int s_mask[RT_HOIST_CELLS * RT_HOIST_CELLS];
	const int frame = blockIdx.x;
	const int cell = (int)threadIdx.x % (RT_HOIST_CELLS * RT_HOIST_CELLS), group = (int)threadIdx.x / (RT_HOIST_CELLS * RT_HOIST_CELLS);
	const int cu = cell % RT_HOIST_CELLS, cv = cell / RT_HOIST_CELLS;
	if (group == 0) s_mask[cell] = 0u;
	__syncthreads();
	const float* P = fp.pose[frame];
	const vec3 cam = make3(P[0], P[1], P[2]);
	const vec3 m0 = make3(P[3], P[4], P[5]), m1 = make3(P[6], P[7], P[8]), mc = make3(P[9], P[10], P[11]);
	vec3 r0 = cross(m1, mc), r1 = cross(mc, m0), r2 = cross(m0, m1);
	const float idet = 1.0f / dot(m0, r0);
	r0 = r0 * idet; r1 = r1 * idet; r2 = r2 * idet;
	const float m = 2e-3f, step = 1.0f / (float)RT_HOIST_CELLS;
	const float u0 = -0.5f + step * (float)cu - m, u1 = -0.5f + step * (float)(cu + 1) + m;
	const float v0 = -0.5f + step * (float)cv - m, v1 = -0.5f + step * (float)(cv + 1) + m;
	float big = 0.f;
	const float camv[3] = { cam.x, cam.y, cam.z };
	for (int k = 0; k < 3; k++) big = fmaxf(big, fmaxf(fabsf(camv[k]), fmaxf(fabsf(sc.bb_lo[k]), fabsf(sc.bb_hi[k]))));
	const float S = 4.0f * big;
	unsigned int mask = 0u;
	for (int j = group; j < sc.hoist_count && j < RT_HOIST_MAX; j += RT_HOIST_GROUPS) {
		const float4* rec = sc.hoist + (size_t)j * RT_AOS_F4;
		const float4 A = rec[1], B = rec[5], C = rec[6];
		float lo[3], hi[3];
		lo[0] = fminf(A.x, fminf(B.x, C.x)); hi[0] = fmaxf(A.x, fmaxf(B.x, C.x));
		lo[1] = fminf(A.y, fminf(B.y, C.y)); hi[1] = fmaxf(A.y, fmaxf(B.y, C.y));
		lo[2] = fminf(A.z, fminf(B.z, C.z)); hi[2] = fmaxf(A.z, fmaxf(B.z, C.z));
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		const float p = (fp.vis_pad * 0.1f + 1e-3f * sqrtf(dx * dx + dy * dy + dz * dz)) * 1.01f + 1e-6f * S;   // a little more than tri_box_padded
		float d2 = 0.f;
		for (int k = 0; k < 3; k++) {
			lo[k] -= p; hi[k] += p;
			const float dd = fmaxf(fmaxf(lo[k] - camv[k], camv[k] - hi[k]), 0.0f);
			d2 += dd * dd;
		}
		const float D = sqrtf(d2);
		// hit points lie within p of the triangle: seen from the camera that is an angle of at most p / D, about twice that in u, v
		const float mj = m + 3.0f * p / D;
		if (!(D * 250.0f > S) || !(mj < 0.05f)) { mask |= 1u << j; continue; }      // the camera is on or next to this triangle
		// The triangle in (a, b, t) = M^-1 (P - camera) against the cell's cone t >= 0, u0 <= a/t <= u1, v0 <= b/t <= v1 — two convex
		// cones from the origin, which meet unless a face plane of one has the other wholly outside (the separating-axis argument
		// for the convex polygons they cut out of the plane t = 1, where the cell's is bounded).  The triangle's cone is that of
		// its part with t >= 0: three or four rays after clipping at t = 0.  (A wall's BOX always has corners behind the camera,
		// which no corner test can tell from corners inside the cone: hence the triangle itself.)
		vec3 V[3];
		{
			const vec3 q0 = make3(A.x - cam.x, A.y - cam.y, A.z - cam.z), q1 = make3(B.x - cam.x, B.y - cam.y, B.z - cam.z), q2 = make3(C.x - cam.x, C.y - cam.y, C.z - cam.z);
			V[0] = make3(dot(r0, q0), dot(r1, q0), dot(r2, q0));
			V[1] = make3(dot(r0, q1), dot(r1, q1), dot(r2, q1));
			V[2] = make3(dot(r0, q2), dot(r1, q2), dot(r2, q2));
		}
		const float U0 = u0 - mj + m, U1 = u1 + mj - m, W0 = v0 - mj + m, W1 = v1 + mj - m;      // (u0, u1, v0, v1 carry m already)
		// (1) a face plane of the cell's cone with the whole triangle outside; or the triangle inside all of them
		bool out = V[0].z < 0.f && V[1].z < 0.f && V[2].z < 0.f, in = V[0].z >= 0.f && V[1].z >= 0.f && V[2].z >= 0.f;
		{
			float f[3];
			for (int k = 0; k < 3; k++) f[k] = V[k].x - U0 * V[k].z;
			out = out || (f[0] < 0.f && f[1] < 0.f && f[2] < 0.f); in = in && f[0] >= 0.f && f[1] >= 0.f && f[2] >= 0.f;
			for (int k = 0; k < 3; k++) f[k] = U1 * V[k].z - V[k].x;
			out = out || (f[0] < 0.f && f[1] < 0.f && f[2] < 0.f); in = in && f[0] >= 0.f && f[1] >= 0.f && f[2] >= 0.f;
			for (int k = 0; k < 3; k++) f[k] = V[k].y - W0 * V[k].z;
			out = out || (f[0] < 0.f && f[1] < 0.f && f[2] < 0.f); in = in && f[0] >= 0.f && f[1] >= 0.f && f[2] >= 0.f;
			for (int k = 0; k < 3; k++) f[k] = W1 * V[k].z - V[k].y;
			out = out || (f[0] < 0.f && f[1] < 0.f && f[2] < 0.f); in = in && f[0] >= 0.f && f[1] >= 0.f && f[2] >= 0.f;
		}
		if (out) continue;
		if (in) { mask |= 1u << j; continue; }
		// (2) the rays of the triangle's part in front of the camera
		vec3 Q[4];
		int n = 3;
		{
			const int n_in = (V[0].z >= 0.f) + (V[1].z >= 0.f) + (V[2].z >= 0.f);       // 1..3 here
			Q[0] = V[0]; Q[1] = V[1]; Q[2] = V[2]; Q[3] = V[0];
			if (n_in < 3) {
				// rotate so that the odd one out (the only vertex in front, or the only one behind) comes first
				const bool want = n_in == 1;
				const bool f0 = (V[0].z >= 0.f) == want, f1 = !f0 && (V[1].z >= 0.f) == want;
				const vec3 X = f0 ? V[0] : (f1 ? V[1] : V[2]), Y = f0 ? V[1] : (f1 ? V[2] : V[0]), Z = f0 ? V[2] : (f1 ? V[0] : V[1]);
				const float sy = X.z / (X.z - Y.z), sz = X.z / (X.z - Z.z);
				const vec3 XY = make3(X.x + sy * (Y.x - X.x), X.y + sy * (Y.y - X.y), 0.f), XZ = make3(X.x + sz * (Z.x - X.x), X.y + sz * (Z.y - X.y), 0.f);
				if (want) { Q[0] = X; Q[1] = XY; Q[2] = XZ; }
				else { Q[0] = Y; Q[1] = Z; Q[2] = XZ; Q[3] = XY; n = 4; }
			}
		}
		// (3) a face plane of the triangle's cone with the cell's four corner rays outside
		bool apart = false;
		auto face = [&](vec3 Pa, vec3 Pb, vec3 Pc) {
			const vec3 nr = cross(Pa, Pb);
			const float side = dot(nr, Pc), tol = 1e-4f * sqrtf(dot(nr, nr));
			if (!(fabsf(side) > tol * (fabsf(Pc.x) + fabsf(Pc.y) + fabsf(Pc.z)))) return;     // (a degenerate face decides nothing)
			const float sg = side > 0.f ? 1.f : -1.f;
			const float c00 = sg * (nr.x * U0 + nr.y * W0 + nr.z), c10 = sg * (nr.x * U1 + nr.y * W0 + nr.z);
			const float c01 = sg * (nr.x * U0 + nr.y * W1 + nr.z), c11 = sg * (nr.x * U1 + nr.y * W1 + nr.z);
			const float lim = -2.0f * tol;
			apart = apart || (c00 < lim && c10 < lim && c01 < lim && c11 < lim);
		};
		face(Q[0], Q[1], Q[2]);
		face(Q[1], Q[2], n == 4 ? Q[3] : Q[0]);
		if (n == 4) { face(Q[2], Q[3], Q[0]); face(Q[3], Q[0], Q[1]); }
		else face(Q[2], Q[0], Q[1]);
		if (!apart) mask |= 1u << j;
	}
	if (mask != 0u) atomicOr(&s_mask[cell], mask);
	__syncthreads();
	if (group == 0) masks[(size_t)frame * (RT_HOIST_CELLS * RT_HOIST_CELLS) + cell] = s_mask[cell];
}

int launch_hoist_masks(const DevScene& sc, const FrameParams& fp, hipStream_t stream) {
	const int nf = fp.n_frames < 1 ? 1 : fp.n_frames;
	hipLaunchKernelGGL(hoist_mask_kernel, dim3((unsigned int)nf), dim3(RT_HOIST_CELLS * RT_HOIST_CELLS * RT_HOIST_GROUPS), 0, stream, sc, fp, (unsigned int*)fp.hoist_mask);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return hip_fail(e, "hoisted-triangle mask launch");
	return RT_OK;
}

// Runs the pass for the frames of `fp` into fp.vis (set by the caller, with vis_stride, vis_count, vis_list): clears, rasterises.
int launch_vis_raster(const DevScene& sc, const FrameParams& fp, const unsigned char* d_skip, bool count, hipStream_t stream) {
	const int nf = fp.n_frames < 1 ? 1 : fp.n_frames;
	hipError_t e = hipMemsetAsync((void*)fp.vis, 0xff, (size_t)nf * (size_t)fp.vis_stride * sizeof(unsigned long long), stream);
	if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync (visibility buffer)");
	e = hipMemsetAsync((void*)fp.vis_count, 0, (size_t)nf * sizeof(int), stream);
	if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync (visibility list counts)");
	const dim3 grid((unsigned int)((sc.n_tris + 255) / 256), (unsigned int)nf), block(256);
	if (count) hipLaunchKernelGGL((vis_raster_kernel<true>), grid, block, 0, stream, sc, fp, d_skip, (unsigned long long*)fp.vis, (int*)fp.vis_count, (int*)fp.vis_list);
	else hipLaunchKernelGGL((vis_raster_kernel<false>), grid, block, 0, stream, sc, fp, d_skip, (unsigned long long*)fp.vis, (int*)fp.vis_count, (int*)fp.vis_list);
	e = hipGetLastError();
	if (e != hipSuccess) return hip_fail(e, "visibility pass launch");
	return RT_OK;
}

} // namespace rt
