// rt_lgrid.hip — per-light direction grids for shadow rays (built on the GPU at commit, LBVH commits only).
//
// processLighting (Source.cpp:460-494) casts, for every lit pixel and light, one ray from the hit point towards the
// light and only asks whether ANY triangle is accepted with t <= dist (Source.cpp:476).  All those rays are lines
// through ONE point, the light: as seen from the light a ray is a single direction, so a 2-D grid over directions — a
// cube map around the light, R x R cells per face — is an exact acceleration structure for them.  A cell lists the
// triangles whose padded box (rt_lbvh.hip tri_box_padded: every hit point the reference can accept lies inside it)
// is seen from the light within that cell, nearest first; a shadow ray reads the list of the one cell its direction
// falls into and stops at the first entry farther from the light than its own origin.  About half of all rays of a
// frame are shadow rays (102 412 triangles, measured: 49 % of the frame time at 800x800, 50 % at 3840x2160), and each
// costs a list of a few triangles here instead of a walk down the tree.
//
// The lists only ever skip work: the triangle test is the reference's own arithmetic (rt_kernels.hip grid_shadow),
// and a triangle is listed in every cell a ray that could be accepted on it may map to:
//   * hit points lie in the padded box B of their triangle (the tree relies on the same fact);
//   * the ray's direction as seen from the light, q = origin - light, differs from the direction of the computed hit
//     point I = o + t*d by the rounding of d = normalize(light - o) and of I: at most 3e-7 * S in position for
//     coordinates and distances below S/4 (S = 4 x the largest coordinate of scene box and light), i.e. 1e-6 * S / D
//     in face coordinates for a triangle D away from the light; the kernel's own cell arithmetic (v_rcp, one
//     multiply, the scaling) adds 1e-6.  Each triangle's cell rectangle is widened by m = 4e-6 * S / D + 1e-4;
//   * a face's rectangle is the exact projection of B (interval division: u = q_b / q_a is monotone in both), taken
//     only when all of B is in front of the face's plane.  For boxes whose diagonal is at most half their distance
//     from the light that loses nothing: such a box can only reach into a face's pyramid if all its corners are in
//     front of that plane.  Wider boxes, boxes within S/250 of the light (m would pass 1e-3) and boxes spanning more
//     than RT_LGRID_TRI_CELLS cells go to the light's short "always" list, tested for every ray of that light; more
//     than RT_LGRID_OVER_MAX of those and the light gets no grid (its shadow rays walk the tree as before);
//   * the scene-spanning triangles kept out of the tree (DevScene::hoist: walls, a ground plane) stay out of the cell
//     lists — seen from a light inside the room each covers a cube face or more — and get a coarse table of their own:
//     per 16 x 16 cell of a face a bit mask of the hoisted triangles that may be seen there (hoisted_masks below), so
//     a ray tests the one or two its direction can meet instead of six box tests per ray.
// Entries are (distance from the light to B, triangle), sorted per cell by that distance: an accepted hit with
// t in (0, dist] lies within dist of the light, so entries beyond dist * (1 + 1e-5) + 1e-6 * S end the list.
#include "rt_internal.h"
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace rt {

namespace {

typedef unsigned long long u64;

#ifndef RT_LGRID_CELL
#define RT_LGRID_CELL 0.3f          /* cell size in mean triangle sizes as seen from the light (102k triangles, 3840x2160 by packets: 0.6 231 us/frame, 0.4 219, 0.25 211, 0.15 209; entries 0.6M, 1.1M, 2.1M, 2.7M) */
#endif
#define RT_LGRID_R_MIN 16
#define RT_LGRID_R_MAX 1024
#define RT_LGRID_TRI_CELLS 4096      /* a triangle covering more cells than this goes to the "always" list */

struct GridParams {
	float L[3];
	float S;              // 4 x the largest coordinate of the padded scene box and the light
	int R;
	float half_R;
	float pad;            // the commit's pad (tri_box_padded)
};

__device__ __forceinline__ void tri_box_padded_g(const float4* __restrict__ planes, int stride, int i, float pad, float lo[3], float hi[3]) {
	// (the same box as rt_lbvh.hip tri_box_padded: the two must agree, the tree's leaves and the grid's entries stand for the same volume)
	const float4 a = planes[PL_P1 * stride + i], b = planes[PL_P2 * stride + i], c = planes[PL_P3 * stride + i];
	lo[0] = fminf(a.x, fminf(b.x, c.x)); hi[0] = fmaxf(a.x, fmaxf(b.x, c.x));
	lo[1] = fminf(a.y, fminf(b.y, c.y)); hi[1] = fmaxf(a.y, fmaxf(b.y, c.y));
	lo[2] = fminf(a.z, fminf(b.z, c.z)); hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
	const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
	const float p = pad * 0.1f + 1e-3f * sqrtf(dx * dx + dy * dy + dz * dz);
	for (int k = 0; k < 3; k++) { lo[k] -= p; hi[k] += p; }
}

struct Rect { int u0, u1, v0, v1; };

// The cells of face `f` that triangle box (lo, hi) may be seen in; false when none.  m: the widening in face coordinates.
__device__ __forceinline__ bool face_rect(const GridParams& g, const float lo[3], const float hi[3], int f, float m, Rect& r) {
	const int a = f >> 1, b = (a + 1) % 3, c = (a + 2) % 3;
	const bool neg = f & 1;
	const float qa0 = neg ? g.L[a] - hi[a] : lo[a] - g.L[a];       // nearest and farthest of the box along the face's axis
	const float qa1 = neg ? g.L[a] - lo[a] : hi[a] - g.L[a];
	if (!(qa0 > 0.0f)) return false;                             // part of the box is not in front of the face's plane (see the header)
	const float ia0 = 1.0f / qa0, ia1 = 1.0f / qa1;
	const float bl = lo[b] - g.L[b], bh = hi[b] - g.L[b], cl = lo[c] - g.L[c], ch = hi[c] - g.L[c];
	const float umin = fminf(bl * ia0, bl * ia1) - m, umax = fmaxf(bh * ia0, bh * ia1) + m;
	const float vmin = fminf(cl * ia0, cl * ia1) - m, vmax = fmaxf(ch * ia0, ch * ia1) + m;
	if (umin > 1.0f || umax < -1.0f || vmin > 1.0f || vmax < -1.0f) return false;
	const float top = (float)(g.R - 1);
	r.u0 = (int)fminf(fmaxf(floorf((umin + 1.0f) * g.half_R), 0.0f), top);
	r.u1 = (int)fminf(fmaxf(floorf((umax + 1.0f) * g.half_R), 0.0f), top);
	r.v0 = (int)fminf(fmaxf(floorf((vmin + 1.0f) * g.half_R), 0.0f), top);
	r.v1 = (int)fminf(fmaxf(floorf((vmax + 1.0f) * g.half_R), 0.0f), top);
	return true;
}

// -> distance from the light to the box, and whether the box is listed by cells (false: it goes to the "always" list)
__device__ __forceinline__ bool classify(const GridParams& g, const float lo[3], const float hi[3], float& D, float& m) {
	float d2 = 0.f, e2 = 0.f;
	for (int k = 0; k < 3; k++) {
		const float d = fmaxf(fmaxf(lo[k] - g.L[k], g.L[k] - hi[k]), 0.0f);
		d2 += d * d;
		e2 += (hi[k] - lo[k]) * (hi[k] - lo[k]);
	}
	D = sqrtf(d2);
	m = 4e-6f * g.S / D + 1e-4f;
	return D > 0.0f && sqrtf(e2) <= 0.5f * D && m <= 1e-3f;
}

// skip[i] != 0: triangle i is kept out of the tree (hoisted) and out of the grids
__global__ void __launch_bounds__(256) k_lg_count(const float4* __restrict__ planes, int stride, int n, const unsigned char* __restrict__ skip,
                                                 GridParams g, unsigned int* __restrict__ cell_count, unsigned int* __restrict__ tri_count,
                                                 int* __restrict__ over, int* __restrict__ n_over) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned int mine = 0;
	if (!skip[i]) {
		float lo[3], hi[3], D, m;
		tri_box_padded_g(planes, stride, i, g.pad, lo, hi);
		bool listed = classify(g, lo, hi, D, m);
		if (listed) {
			Rect r[6];
			bool on[6];
			unsigned int cells = 0;
			for (int f = 0; f < 6; f++) {
				on[f] = face_rect(g, lo, hi, f, m, r[f]);
				if (on[f]) cells += (unsigned int)(r[f].u1 - r[f].u0 + 1) * (unsigned int)(r[f].v1 - r[f].v0 + 1);
			}
			if (cells > RT_LGRID_TRI_CELLS) listed = false;
			else {
				mine = cells;
				for (int f = 0; f < 6; f++) {
					if (!on[f]) continue;
					for (int v = r[f].v0; v <= r[f].v1; v++)
						for (int u = r[f].u0; u <= r[f].u1; u++)
							atomicAdd(&cell_count[((unsigned int)f * g.R + v) * g.R + u], 1u);
				}
			}
		}
		if (!listed) {
			const int k = atomicAdd(n_over, 1);
			if (k < RT_LGRID_OVER_MAX) over[k] = i;
		}
	}
	tri_count[i] = mine;
}

__global__ void __launch_bounds__(256) k_lg_fill(const float4* __restrict__ planes, int stride, int n, const unsigned char* __restrict__ skip,
                                                GridParams g, const unsigned int* __restrict__ tri_start, u64* __restrict__ keys, int* __restrict__ vals) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || skip[i]) return;
	unsigned int at = tri_start[i];
	if (tri_start[i + 1] == at) return;                          // not listed by cells (k_lg_count decided)
	float lo[3], hi[3], D, m;
	tri_box_padded_g(planes, stride, i, g.pad, lo, hi);
	(void)classify(g, lo, hi, D, m);
	const u64 dbits = (u64)__float_as_uint(D);                   // D > 0: the bit pattern orders like the value
	for (int f = 0; f < 6; f++) {
		Rect r;
		if (!face_rect(g, lo, hi, f, m, r)) continue;
		for (int v = r.v0; v <= r.v1; v++)
			for (int u = r.u0; u <= r.u1; u++) {
				keys[at] = ((u64)(((unsigned int)f * g.R + v) * g.R + u) << 32) | dbits;
				vals[at] = i;
				at++;
			}
	}
}

__global__ void __launch_bounds__(256) k_lg_pack(const u64* __restrict__ keys, const int* __restrict__ vals, unsigned int n, int2* __restrict__ out) {
	const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = make_int2((int)(unsigned int)keys[i], vals[i]);
}

template <typename T>
int dev_alloc(T** p, size_t count) {
	hipError_t e = hipMalloc((void**)p, (count > 0 ? count : 1) * sizeof(T));
	if (e != hipSuccess) return hip_fail(e, "hipMalloc (shadow grid)");
	return RT_OK;
}

} // namespace

// Which hoisted triangles (kept out of the tree and out of the cell lists: walls, a ground plane) may be seen from the light in
// each coarse direction cell: bit j of mask[(face * HC + v) * HC + u].  On the host: the cell's directions are the cone
// s*q_a >= 0, u0 <= q_b / (s*q_a) <= u1, v0 <= q_c / (s*q_a) <= v1 — linear inequalities in q = point - light — and the TRIANGLE
// is clipped against them; what the reference can accept lies within the pad p of the triangle, so the cell is widened by the m
// of the cell lists' rectangles plus the angle p / D stands for.  A triangle nearer the light than S/250, or so near that the
// widening passes 0.1, is seen everywhere.
void hoisted_masks(const rt_scene* s, const DeviceState& d, const float L[3], float S, std::vector<unsigned int>& mask, float* hmind, int* hidx) {
	const int HC = RT_LGRID_HC;
	mask.assign((size_t)6 * HC * HC, 0u);
	const int nh = (int)d.hoisted.size();
	for (int j = 0; j < nh && j < RT_HOIST_MAX; j++) {
		const int t = d.hoisted[(size_t)j];
		hidx[j] = t;
		const float* v = s->tris.data() + (size_t)t * RT_TRI_FLOATS + 6;
		double lo[3], hi[3], e2 = 0.0;
		for (int k = 0; k < 3; k++) {
			lo[k] = fmin(v[k], fmin(v[3 + k], v[6 + k])); hi[k] = fmax(v[k], fmax(v[3 + k], v[6 + k]));
			e2 += (hi[k] - lo[k]) * (hi[k] - lo[k]);
		}
		const double p = ((double)d.pad * 0.1 + 1e-3 * sqrt(e2)) * 1.001 + 1e-6 * S;
		double d2 = 0.0;
		for (int k = 0; k < 3; k++) {
			lo[k] -= p; hi[k] += p;
			const double dd = fmax(fmax(lo[k] - L[k], L[k] - hi[k]), 0.0);
			d2 += dd * dd;
		}
		const double D = sqrt(d2);
		hmind[j] = (float)(D * (1.0 - 1e-6));
		if (!(D > 0.0) || !(4e-6 * S / D + 1e-4 <= 1e-3)) {           // the light is on or next to this triangle: every direction
			for (unsigned int& w : mask) w |= 1u << j;
			hmind[j] = 0.f;
			continue;
		}
		// hit points lie within p of the triangle: seen from the light an angle of at most p / D, at most twice that in face coordinates
		const double m = 4e-6 * S / D + 1e-4 + 3.0 * p / D;
		if (!(m < 0.1)) { for (unsigned int& w : mask) w |= 1u << j; hmind[j] = 0.f; continue; }
		// The triangle clipped against each cell's cone (Sutherland-Hodgman, five planes through the light): a wall's BOX always has
		// corners behind the face's plane, which no corner test can tell from corners inside the cone.
		for (int f = 0; f < 6; f++) {
			const int a = f >> 1, b = (a + 1) % 3, cc = (a + 2) % 3;
			const double sg = (f & 1) ? -1.0 : 1.0;
			double T[3][3];                                           // (q_b, q_c, s * q_a) of the three vertices
			bool front = false;
			for (int k = 0; k < 3; k++) {
				T[k][0] = (double)v[3 * k + b] - L[b]; T[k][1] = (double)v[3 * k + cc] - L[cc]; T[k][2] = sg * ((double)v[3 * k + a] - L[a]);
				front = front || T[k][2] > -p;
			}
			if (!front) continue;
			for (int cv = 0; cv < HC; cv++)
				for (int cu = 0; cu < HC; cu++) {
					const double u0 = -1.0 + 2.0 * cu / HC - m, u1 = -1.0 + 2.0 * (cu + 1) / HC + m;
					const double v0 = -1.0 + 2.0 * cv / HC - m, v1 = -1.0 + 2.0 * (cv + 1) / HC + m;
					const double pl[5][3] = { { 0, 0, 1 }, { 1, 0, -u0 }, { -1, 0, u1 }, { 0, 1, -v0 }, { 0, -1, v1 } };
					// (most cells end here: all three corners outside one plane — clipping would leave nothing — or inside all five)
					bool out = false, in = true;
					for (int c = 0; c < 5 && !out; c++) {
						double fv[3];
						for (int k = 0; k < 3; k++) fv[k] = pl[c][0] * T[k][0] + pl[c][1] * T[k][1] + pl[c][2] * T[k][2];
						out = fv[0] < 0.0 && fv[1] < 0.0 && fv[2] < 0.0;
						in = in && fv[0] >= 0.0 && fv[1] >= 0.0 && fv[2] >= 0.0;
					}
					if (out) continue;
					if (in) { mask[((size_t)f * HC + cv) * HC + cu] |= 1u << j; continue; }
					double P[10][3], Q[10][3];
					int n = 3;
					for (int k = 0; k < 3; k++) for (int e = 0; e < 3; e++) P[k][e] = T[k][e];
					for (int c = 0; c < 5 && n > 0; c++) {
						int k = 0;
						for (int vtx = 0; vtx < n; vtx++) {
							const int w = vtx + 1 == n ? 0 : vtx + 1;
							const double fc = pl[c][0] * P[vtx][0] + pl[c][1] * P[vtx][1] + pl[c][2] * P[vtx][2];
							const double fn = pl[c][0] * P[w][0] + pl[c][1] * P[w][1] + pl[c][2] * P[w][2];
							if (fc >= 0.0 && k < 10) { for (int e = 0; e < 3; e++) Q[k][e] = P[vtx][e]; k++; }
							if ((fc >= 0.0) != (fn >= 0.0) && k < 10) {
								const double sgm = fc / (fc - fn);
								for (int e = 0; e < 3; e++) Q[k][e] = P[vtx][e] + sgm * (P[w][e] - P[vtx][e]);
								k++;
							}
						}
						n = k;
						for (int vtx = 0; vtx < n; vtx++) for (int e = 0; e < 3; e++) P[vtx][e] = Q[vtx][e];
					}
					if (n > 0) mask[((size_t)f * HC + cv) * HC + cu] |= 1u << j;
				}
		}
	}
}

void lgrid_free(DeviceState& d) {
	for (void* p : d.lgrid_buffers) if (p) (void)hipFree(p);
	d.lgrid_buffers.clear();
	if (d.lgrid) (void)hipFree(d.lgrid);
	d.lgrid = nullptr;
	d.n_lgrid = 0;
	d.lgrid_info.clear();
	d.lgrid_ms = 0.f;
}

// One grid per light for the first RT_LGRID_MAX_LIGHTS lights of an LBVH commit.  A light whose grid cannot be had (too
// many wide or near triangles, lists beyond the entry budget) keeps R = 0 in its record: its shadow rays walk the tree.
int lgrid_build(rt_scene* s, DeviceState& d, hipStream_t stream) {
	lgrid_free(d);
	const int n = (int)(s->tris.size() / RT_TRI_FLOATS);
	const int nl = (int)(s->lights.size() / 4);
	if (!s->shadow_grid || d.accel != RT_ACCEL_LBVH || nl <= 0 || n < 2) return RT_OK;
	const int ng = nl < RT_LGRID_MAX_LIGHTS ? nl : RT_LGRID_MAX_LIGHTS;
	std::vector<LightGrid> table((size_t)ng);
	d.lgrid_info.assign((size_t)ng * 3, 0);
	for (LightGrid& t : table) { t.cell_start = nullptr; t.entries = nullptr; t.R = 0; t.half_R = 0.f; t.slack = 0.f; t.n_over = 0; for (int& o : t.over) o = 0; t.hmask = nullptr; for (float& x : t.hmind) x = 0.f; for (int& x : t.hidx) x = 0; }

	unsigned char* skip = nullptr;
	unsigned int *cell_count = nullptr, *tri_count = nullptr, *tri_start = nullptr;
	int *over = nullptr, *vals = nullptr, *vals_sorted = nullptr;
	u64 *keys = nullptr, *keys_sorted = nullptr;
	void* temp = nullptr;
	size_t temp_have = 0;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	int rc = RT_OK;
	auto cleanup = [&]() {
		if (skip) (void)hipFree(skip);
		if (cell_count) (void)hipFree(cell_count);
		if (tri_count) (void)hipFree(tri_count);
		if (tri_start) (void)hipFree(tri_start);
		if (over) (void)hipFree(over);
		if (vals) (void)hipFree(vals);
		if (vals_sorted) (void)hipFree(vals_sorted);
		if (keys) (void)hipFree(keys);
		if (keys_sorted) (void)hipFree(keys_sorted);
		if (temp) (void)hipFree(temp);
		if (e0) (void)hipEventDestroy(e0);
		if (e1) (void)hipEventDestroy(e1);
	};
#define LG_TRY(x) do { rc = (x); if (rc != RT_OK) { cleanup(); lgrid_free(d); return rc; } } while (0)
#define LG_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = hip_fail(e_, #x); cleanup(); lgrid_free(d); return rc; } } while (0)
	auto need_temp = [&](size_t bytes) -> int {
		if (bytes <= temp_have) return RT_OK;
		if (temp) { (void)hipFree(temp); temp = nullptr; temp_have = 0; }
		hipError_t e = hipMalloc(&temp, bytes);
		if (e != hipSuccess) return hip_fail(e, "hipMalloc (shadow grid scratch)");
		temp_have = bytes;
		return RT_OK;
	};

#ifdef RT_EXPERIMENT
	const bool times = std::getenv("RT355_LGRID_TIMES") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if (!times) return;
		(void)hipStreamSynchronize(stream);
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "lgrid %-22s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_last).count());
		t_last = now;
	};
#else
	auto lap = [](const char*) {};
#endif
	std::vector<unsigned char> h_skip((size_t)n, 0);
	for (int i : d.hoisted) h_skip[(size_t)i] = 1;
	LG_TRY(dev_alloc(&skip, (size_t)n));
	LG_HIP(hipMemcpy(skip, h_skip.data(), (size_t)n, hipMemcpyHostToDevice));
	LG_TRY(dev_alloc(&tri_count, (size_t)n + 1));
	LG_TRY(dev_alloc(&tri_start, (size_t)n + 1));
	LG_TRY(dev_alloc(&over, (size_t)RT_LGRID_OVER_MAX + 1));
	LG_HIP(hipEventCreate(&e0));
	LG_HIP(hipEventCreate(&e1));
	LG_HIP(hipEventRecord(e0, stream));
	const float4* planes = (const float4*)d.planes;
	const unsigned int blocks_n = (unsigned int)((n + 255) / 256);
	const size_t entry_budget = (size_t)n * 48 + (1u << 20);
	lap("set-up, allocations");

	for (int li = 0; li < ng; li++) {
		const float* Lp = s->lights.data() + (size_t)li * 4;
		GridParams g;
		float big = 0.f;
		for (int k = 0; k < 3; k++) {
			g.L[k] = Lp[k];
			big = fmaxf(big, fmaxf(fabsf(Lp[k]), fmaxf(fabsf(d.bb_lo[k]), fabsf(d.bb_hi[k]))));
		}
		if (!(big < 1e18f)) continue;                              // a light at infinity or NaN: no grid
		g.S = 4.0f * big;
		g.pad = d.pad;
		// resolution: cells of RT_LGRID_CELL mean triangle sizes as seen from the light
		// (a sample of at most ~4096 triangles, evenly spread over the list: the resolution only has to be about right)
		double sum = 0.0;
		long long cnt = 0;
		const int sample_step = n > 4096 ? n / 4096 : 1;
		for (int i = 0; i < n; i += sample_step) {
			if (h_skip[(size_t)i]) continue;
			const float* v = s->tris.data() + (size_t)i * RT_TRI_FLOATS + 6;
			double e2 = 0.0, d2 = 0.0;
			for (int k = 0; k < 3; k++) {
				const float lo = fminf(v[k], fminf(v[3 + k], v[6 + k])), hi = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
				e2 += (double)(hi - lo) * (double)(hi - lo);
				const double dd = fmax(fmax((double)lo - Lp[k], (double)Lp[k] - hi), 0.0);
				d2 += dd * dd;
			}
			if (d2 > 0.0 && e2 <= 0.25 * d2) { sum += sqrt(e2 / d2); cnt++; }
		}
		if (cnt == 0) continue;
		lap("mean angular size");
		float cell = RT_LGRID_CELL;
#ifdef RT_EXPERIMENT
		if (const char* e = std::getenv("RT355_LGRID_CELL")) cell = (float)atof(e);
#endif
		const double theta = sum / (double)cnt;
		int R = (int)ceil(1.5707963 / (theta * cell > 1e-6 ? theta * cell : 1e-6));
		R = R < RT_LGRID_R_MIN ? RT_LGRID_R_MIN : (R > RT_LGRID_R_MAX ? RT_LGRID_R_MAX : R);
		R = (R + 7) / 8 * 8;

		for (int attempt = 0; attempt < 4; attempt++, R = (R / 2 + 7) / 8 * 8) {
			if (R < RT_LGRID_R_MIN) break;
			g.R = R;
			g.half_R = 0.5f * (float)R;
			const size_t cells = (size_t)6 * R * R;
			if (cell_count) { (void)hipFree(cell_count); cell_count = nullptr; }
			LG_TRY(dev_alloc(&cell_count, cells + 1));
			int* cell_start = nullptr;
			LG_TRY(dev_alloc(&cell_start, cells + 1));
			d.lgrid_buffers.push_back(cell_start);
			LG_HIP(hipMemsetAsync(cell_count, 0, (cells + 1) * sizeof(unsigned int), stream));
			LG_HIP(hipMemsetAsync(over, 0, (RT_LGRID_OVER_MAX + 1) * sizeof(int), stream));
			LG_HIP(hipMemsetAsync(tri_count + n, 0, sizeof(unsigned int), stream));
			int* n_over = over + RT_LGRID_OVER_MAX;
			hipLaunchKernelGGL(k_lg_count, dim3(blocks_n), dim3(256), 0, stream, planes, d.stride, n, (const unsigned char*)skip, g, cell_count, tri_count, over, n_over);
			LG_HIP(hipGetLastError());
			size_t bytes = 0;
			LG_HIP(rocprim::exclusive_scan(nullptr, bytes, cell_count, (unsigned int*)cell_start, 0u, cells + 1, rocprim::plus<unsigned int>(), stream));
			LG_TRY(need_temp(bytes));
			LG_HIP(rocprim::exclusive_scan(temp, bytes, cell_count, (unsigned int*)cell_start, 0u, cells + 1, rocprim::plus<unsigned int>(), stream));
			LG_HIP(rocprim::exclusive_scan(nullptr, bytes, tri_count, tri_start, 0u, (size_t)n + 1, rocprim::plus<unsigned int>(), stream));
			LG_TRY(need_temp(bytes));
			LG_HIP(rocprim::exclusive_scan(temp, bytes, tri_count, tri_start, 0u, (size_t)n + 1, rocprim::plus<unsigned int>(), stream));
			unsigned int total = 0;
			int h_over[RT_LGRID_OVER_MAX + 1];
			LG_HIP(hipMemcpyAsync(&total, tri_start + n, sizeof total, hipMemcpyDeviceToHost, stream));
			LG_HIP(hipMemcpyAsync(h_over, over, sizeof h_over, hipMemcpyDeviceToHost, stream));
			LG_HIP(hipStreamSynchronize(stream));
			lap("count + scans");
			if (h_over[RT_LGRID_OVER_MAX] > RT_LGRID_OVER_MAX || (size_t)total > entry_budget) {
				d.lgrid_buffers.pop_back();
				(void)hipFree(cell_start);
				if (h_over[RT_LGRID_OVER_MAX] > RT_LGRID_OVER_MAX) break;             // too many wide or near triangles: no grid for this light
				continue;                                                              // lists too long at this resolution: a coarser grid
			}
			int2* entries = nullptr;
			LG_TRY(dev_alloc(&entries, (size_t)total));
			d.lgrid_buffers.push_back(entries);
			if (total > 0) {
				if (keys) { (void)hipFree(keys); keys = nullptr; }
				if (keys_sorted) { (void)hipFree(keys_sorted); keys_sorted = nullptr; }
				if (vals) { (void)hipFree(vals); vals = nullptr; }
				if (vals_sorted) { (void)hipFree(vals_sorted); vals_sorted = nullptr; }
				LG_TRY(dev_alloc(&keys, (size_t)total));
				LG_TRY(dev_alloc(&keys_sorted, (size_t)total));
				LG_TRY(dev_alloc(&vals, (size_t)total));
				LG_TRY(dev_alloc(&vals_sorted, (size_t)total));
				lap("entry allocations");
				hipLaunchKernelGGL(k_lg_fill, dim3(blocks_n), dim3(256), 0, stream, planes, d.stride, n, (const unsigned char*)skip, g, (const unsigned int*)tri_start, keys, vals);
				LG_HIP(hipGetLastError());
				unsigned int cell_bits = 1;
				while ((1ull << cell_bits) < cells) cell_bits++;
				LG_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys, keys_sorted, vals, vals_sorted, (size_t)total, 0, 32 + cell_bits, stream));
				LG_TRY(need_temp(bytes));
				LG_HIP(rocprim::radix_sort_pairs(temp, bytes, keys, keys_sorted, vals, vals_sorted, (size_t)total, 0, 32 + cell_bits, stream));
				hipLaunchKernelGGL(k_lg_pack, dim3((total + 255) / 256), dim3(256), 0, stream, (const u64*)keys_sorted, (const int*)vals_sorted, total, entries);
				LG_HIP(hipGetLastError());
			}
			lap("fill + sort + pack");
			LightGrid& t = table[(size_t)li];
			t.cell_start = cell_start;
			t.entries = entries;
			t.R = R;
			t.half_R = g.half_R;
			t.slack = 1e-6f * g.S;
			t.n_over = h_over[RT_LGRID_OVER_MAX];
			std::sort(h_over, h_over + t.n_over);
			for (int k = 0; k < t.n_over; k++) t.over[k] = h_over[k];
			{   // the triangles kept out of the tree, by direction
				std::vector<unsigned int> hm;
				hoisted_masks(s, d, g.L, g.S, hm, t.hmind, t.hidx);
				unsigned int* dm = nullptr;
				LG_TRY(dev_alloc(&dm, hm.size()));
				d.lgrid_buffers.push_back(dm);
				LG_HIP(hipMemcpy(dm, hm.data(), hm.size() * sizeof(unsigned int), hipMemcpyHostToDevice));
				t.hmask = dm;
			}
			lap("hoisted masks");
			d.lgrid_info[(size_t)li * 3] = R;
			d.lgrid_info[(size_t)li * 3 + 1] = (long long)total;
			d.lgrid_info[(size_t)li * 3 + 2] = t.n_over;
			break;
		}
	}
	LG_HIP(hipMalloc(&d.lgrid, (size_t)ng * sizeof(LightGrid)));
	LG_HIP(hipMemcpyAsync(d.lgrid, table.data(), (size_t)ng * sizeof(LightGrid), hipMemcpyHostToDevice, stream));
	LG_HIP(hipEventRecord(e1, stream));
	LG_HIP(hipEventSynchronize(e1));
	float ms = 0.f;
	LG_HIP(hipEventElapsedTime(&ms, e0, e1));
#undef LG_TRY
#undef LG_HIP
	cleanup();
	d.n_lgrid = ng;
	d.lgrid_ms = ms;
	return RT_OK;
}

} // namespace rt
