// rt_scene.cpp — host-side scene construction: the reference's start-up arithmetic
// (Triangle ctor Source.cpp:45-59, addTriangle Source.cpp:410-439, trans Source.cpp:1145-1147,
// loadModel Source.cpp:631-657) and the flatten-to-SoA step that precedes the upload.
//
// Compiled with -ffp-contract=off: these records are the inputs of every later comparison, so
// they must be the bits the reference's own constructors produce.  sinf/cosf come from the host
// libm exactly as they do for the reference (glm::rotate -> std::cos / std::sin).
#include "rt_internal.h"
#include <thread>

#include <cerrno>
#include <cstdio>
#include <cstring>

namespace rt {

namespace {

mat4 identity() {
	mat4 m;
	m.c[0] = make4(1, 0, 0, 0);
	m.c[1] = make4(0, 1, 0, 0);
	m.c[2] = make4(0, 0, 1, 0);
	m.c[3] = make4(0, 0, 0, 1);
	return m;
}

// glm::radians: degrees * 0.01745329251994329576923690768489 (as float)
float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// glm/ext/matrix_transform.inl:10-15
mat4 translate(const mat4& m, vec3 v) {
	mat4 r = m;
	r.c[3] = m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3];
	return r;
}

// glm/ext/matrix_transform.inl:18-46
mat4 rotate(const mat4& m, float angle, vec3 v) {
	const float a = angle;
	const float c = cosf(a);
	const float s = sinf(a);
	vec3 axis = normalize(v);
	vec3 temp = axis * (1.0f - c);
	float R[3][3];
	R[0][0] = c + temp.x * axis.x;
	R[0][1] = temp.x * axis.y + s * axis.z;
	R[0][2] = temp.x * axis.z - s * axis.y;
	R[1][0] = temp.y * axis.x - s * axis.z;
	R[1][1] = c + temp.y * axis.y;
	R[1][2] = temp.y * axis.z + s * axis.x;
	R[2][0] = temp.z * axis.x + s * axis.y;
	R[2][1] = temp.z * axis.y - s * axis.x;
	R[2][2] = c + temp.z * axis.z;
	mat4 r;
	for (int i = 0; i < 3; i++) r.c[i] = m.c[0] * R[i][0] + m.c[1] * R[i][1] + m.c[2] * R[i][2];
	r.c[3] = m.c[3];
	return r;
}

// glm/ext/matrix_transform.inl:78-86
mat4 scale(const mat4& m, vec3 v) {
	mat4 r;
	r.c[0] = m.c[0] * v.x;
	r.c[1] = m.c[1] * v.y;
	r.c[2] = m.c[2] * v.z;
	r.c[3] = m.c[3];
	return r;
}

// glm/detail/type_mat4x4.inl:630-648
mat4 matmul(const mat4& a, const mat4& b) {
	mat4 r;
	for (int i = 0; i < 4; i++) {
		vec4 bc = b.c[i];
		r.c[i] = a.c[0] * bc.x + a.c[1] * bc.y + a.c[2] * bc.z + a.c[3] * bc.w;
	}
	return r;
}

void put3(float* o, vec3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

} // namespace

void triangle_record(vec3 p1, vec3 p2, vec3 p3, vec3 color, float shin, float* o) {
	// Source.cpp:45-59
	vec3 n = normalize(cross(p1 - p2, p1 - p3));
	put3(o + 0, p1);           // p
	put3(o + 3, n);
	put3(o + 6, p1);
	put3(o + 9, p2);
	put3(o + 12, p3);
	put3(o + 15, p2 - p1);     // edge0
	put3(o + 18, p3 - p2);     // edge1
	put3(o + 21, p1 - p3);     // edge2
	put3(o + 24, color);
	o[27] = shin;
}

void quad_half_record(const float T[3], const float Rdeg[3], const float S[3], const float color[3],
                      float shin, float* out28) {
	// Source.cpp:410-439: base triangle, mod = translate * rotatex * rotatey * rotatez * scale
	vec4 p1 = make4(-1, -1, 0, 1), p2 = make4(-1, 1, 0, 1), p3 = make4(1, -1, 0, 1);
	const mat4 I = identity();
	mat4 t = translate(I, make3(T[0], T[1], T[2]));
	mat4 rx = rotate(I, radians(Rdeg[0]), make3(1, 0, 0));
	mat4 ry = rotate(I, radians(Rdeg[1]), make3(0, 1, 0));
	mat4 rz = rotate(I, radians(Rdeg[2]), make3(0, 0, 1));
	mat4 sc = scale(I, make3(S[0], S[1], S[2]));
	mat4 mod = matmul(matmul(matmul(matmul(t, rx), ry), rz), sc);
	p1 = mul(mod, p1);
	p2 = mul(mod, p2);
	p3 = mul(mod, p3);
	triangle_record(make3(p1.x, p1.y, p1.z), make3(p2.x, p2.y, p2.z), make3(p3.x, p3.y, p3.z),
	                make3(color[0], color[1], color[2]), shin, out28);
}

void camera_trans(const float rot_deg[3], float out16[16]) {
	// Source.cpp:1145-1147
	mat4 t = identity();
	t = rotate(t, radians(rot_deg[0]), make3(0.0f, 1.0f, 0.0f));
	t = rotate(t, radians(-rot_deg[1]), make3(1.0f, 0.0f, 0.0f));
	memcpy(out16, &t, sizeof(float) * 16);
}

// Bounding box of every triangle vertex, padded by 1e-3 of the scene's scale.  The render kernels
// use boxes only to skip work that cannot produce a hit (rt_kernels.hip), never to decide one.
// A scene with a non-finite vertex gets an infinite box (no pruning).  Returns the pad.
float scene_box(const std::vector<float>& tris, float lo[3], float hi[3]) {
	const float inf = __builtin_huge_valf();
	const size_t n = tris.size() / RT_TRI_FLOATS;
	bool finite = n > 0;
	float l[3] = { inf, inf, inf }, h[3] = { -inf, -inf, -inf };
	for (size_t i = 0; i < n && finite; i++) {
		const float* t = tris.data() + i * RT_TRI_FLOATS + 6;      // p1 p2 p3
		for (int v = 0; v < 9; v++) {
			const float c = t[v];
			if (!(c - c == 0.0f)) { finite = false; break; }
			if (c < l[v % 3]) l[v % 3] = c;
			if (c > h[v % 3]) h[v % 3] = c;
		}
	}
	if (!finite) {
		for (int k = 0; k < 3; k++) { lo[k] = -inf; hi[k] = inf; }
		return inf;
	}
	float scale = 1e-30f;
	for (int k = 0; k < 3; k++) {
		if (h[k] - l[k] > scale) scale = h[k] - l[k];
		if (fabsf(l[k]) > scale) scale = fabsf(l[k]);
		if (fabsf(h[k]) > scale) scale = fabsf(h[k]);
	}
	const float pad = scale * 1e-3f;
	for (int k = 0; k < 3; k++) { lo[k] = l[k] - pad; hi[k] = h[k] + pad; }
	return pad;
}

// The sign filter for the three edge terms of Source.cpp:392-394 (kernels: edge_signs in rt_kernels.hip).
// The reference evaluates  v_k = dot(n, cross(edge_k, I - p_k))  in binary32, 17 operations per edge, and only the
// SIGN of the result is used.  In real arithmetic  v_k = M_k . (I - p_k)  with  M_k = n x edge_k  (scalar triple
// product), and with w = I - p1:  v_0 = M_0 . w,  v_1 = M_1 . w - c_1  (c_1 = M_1 . (p2 - p1)),  v_2 = M_2 . w - c_2  where
// c_2 = M_2 . (p3 - p1) is zero up to the rounding of the stored edge_2 = fl(p1 - p3).  The kernels evaluate
// s_k = fma(M_k.x, w.x, fma(M_k.y, w.y, fma(M_k.z, w.z, -c_k)))  (c_0 = c_2 = 0) and trust its sign only when
// |s_k| > G * (|w.x| + |w.y| + |w.z|) + H; every other lane takes the reference's own 17 operations.
// Error budget, u = 2^-24, A = (|n.x|+|n.y|+|n.z|) * max_k (|e_k.x|+|e_k.y|+|e_k.z|), |w| the largest component of w:
//   the reference's value against the real one: 6 roundings on the longest path (w_k, product, cross, product, two sums)
//     of terms bounded by A * |w_k|, |w_k| <= |w| + |p_k - p1|                       <= 6 u A (|w| + E)
//   s_k against the real one: w = fl(I - p1) rounded per component (u A |w|), M_k rounded from binary64 (u A |w|), c_1
//     rounded (u A E), c_2 dropped — edge_2 = fl(p1 - p3) is rounded per component, so M_2 . (p3 - p1) can reach 2 u A E —
//     three fma roundings of partial sums bounded by A (|w| + E)                      <= 7 u A (|w| + E)
// with E = max(|p2 - p1|_1, |p3 - p1|_1).  13 u A (|w| + E) <= G |w|_1 + H for G = 16 u A, H = G E (both rounded up by
// 2^-10; a margin of 3 u A, not to be tightened on the strength of this count alone — tests/test_edge_filter_cpu.py pins the
// filter against exact rational arithmetic on needle triangles, where the rounding of edge_2 is largest) — no product may overflow or leave the normal range for this to hold, which the kernels make sure of by
// trusting the filter only for G |w|_1 + H < 1e20 (all partial results are then below 1e27) and raising the threshold to
// 1e-20 where it is smaller (what underflow can add is 20 orders of magnitude below that).  A degenerate triangle
// (n not finite) and one whose G is outside [1e-30, 1e30] get G = NaN: never trusted.
static void edge_filter_planes(const float* t, float* m0, float* m1, float* m2) {
	const float* nrm = t + 3; const float* p1 = t + 6; const float* p2 = t + 9; const float* p3 = t + 12;
	const float* e[3] = { t + 15, t + 18, t + 21 };
	float* m[3] = { m0, m1, m2 };
	double n1 = fabs((double)nrm[0]) + fabs((double)nrm[1]) + fabs((double)nrm[2]), emax = 0.0;
	double M[3][3];
	for (int k = 0; k < 3; k++) {
		const double ex = e[k][0], ey = e[k][1], ez = e[k][2], nx = nrm[0], ny = nrm[1], nz = nrm[2];
		M[k][0] = ny * ez - nz * ey;          // (n x e): products of two binary32 values are exact in binary64
		M[k][1] = nz * ex - nx * ez;
		M[k][2] = nx * ey - ny * ex;
		for (int c = 0; c < 3; c++) m[k][c] = (float)M[k][c];
		const double e1n = fabs(ex) + fabs(ey) + fabs(ez);
		if (e1n > emax) emax = e1n;
	}
	double d2[3], d3[3], E2 = 0.0, E3 = 0.0, c1 = 0.0;
	for (int c = 0; c < 3; c++) {
		d2[c] = (double)p2[c] - (double)p1[c];
		d3[c] = (double)p3[c] - (double)p1[c];
		E2 += fabs(d2[c]);
		E3 += fabs(d3[c]);
		c1 += M[1][c] * d2[c];
	}
	const double up = 1.0 + 1.0 / 1024.0;
	const double G = 16.0 * (1.0 / 16777216.0) * n1 * emax * up;
	const double H = G * (E2 > E3 ? E2 : E3) * up;
	m0[3] = (float)(G * up);                  // (the conversions round to nearest: the second 2^-10 covers them)
	m1[3] = (float)c1;
	m2[3] = (float)(H * up);
	// a filter that cannot be trusted for this triangle: G = NaN sends every lane to the reference's arithmetic
	// (G below 1e-30 — edges shorter than ~1e-24 — is switched off too: as a binary32 near or below the normal range it would no longer
	// be the bound to 2^-24, and with it the roundings of denormal M components, at most 2^-150 |w|, are always far inside G |w|_1)
	bool ok = G - G == 0.0 && H - H == 0.0 && c1 - c1 == 0.0 && fabs(c1) < 1e30 && G >= 1e-30 && G < 1e30;
	for (int k = 0; k < 3 && ok; k++) for (int c = 0; c < 3; c++) if (!(fabs(M[k][c]) < 1e30)) ok = false;
	if (!ok) m0[3] = __builtin_nanf("");
}

} // namespace rt

// Test hook (not part of include/rt355.h): the twelve filter constants of one 28-float triangle record — M0.xyz G | M1.xyz c1 |
// M2.xyz H — so that the CPU suite can hold the bound against exact rational arithmetic without a GPU.
extern "C" void rt_internal_edge_filter_planes(const float* tri28, float* out12) {
	rt::edge_filter_planes(tri28, out12, out12 + 4, out12 + 8);
}

namespace rt {

PlaneLayout flatten_planes(const std::vector<float>& tris, int stride, float pad, bool with_octant_records, std::vector<float>& out) {
	const int n = (int)(tris.size() / RT_TRI_FLOATS);
	const float inf = __builtin_huge_valf();
	out.assign((size_t)PL_COUNT * stride * 4, 0.0f);
	auto plane = [&](int k, int i) { return out.data() + ((size_t)k * stride + i) * 4; };
	// (per triangle and independent: large scenes are laid out by up to 8 host threads — 3.7 ms of a 12 ms commit at 100k triangles
	// on one)
	auto lay_out = [&](int i0, int i1) {
	for (int i = i0; i < i1; i++) {
		const float* t = tris.data() + (size_t)i * RT_TRI_FLOATS;
		const float* nrm = t + 3; const float* p1 = t + 6; const float* p2 = t + 9; const float* p3 = t + 12;
		const float* e0 = t + 15; const float* e1 = t + 18; const float* e2 = t + 21; const float* col = t + 24;
		float* q;
		q = plane(PL_N, i);  q[0] = nrm[0]; q[1] = nrm[1]; q[2] = nrm[2]; q[3] = t[27];
		q = plane(PL_P1, i); q[0] = p1[0]; q[1] = p1[1]; q[2] = p1[2]; q[3] = col[0];
		q = plane(PL_P2, i); q[0] = p2[0]; q[1] = p2[1]; q[2] = p2[2]; q[3] = col[1];
		q = plane(PL_P3, i); q[0] = p3[0]; q[1] = p3[1]; q[2] = p3[2]; q[3] = col[2];
		q = plane(PL_E0, i); q[0] = e0[0]; q[1] = e0[1]; q[2] = e0[2];
		q = plane(PL_E1, i); q[0] = e1[0]; q[1] = e1[1]; q[2] = e1[2];
		q = plane(PL_E2, i); q[0] = e2[0]; q[1] = e2[1]; q[2] = e2[2];
		q = plane(PL_COL, i); q[0] = col[0]; q[1] = col[1]; q[2] = col[2]; q[3] = t[27];
		edge_filter_planes(t, plane(PL_M0, i), plane(PL_M1, i), plane(PL_M2, i));
	}
	};
	{
		unsigned int hw = std::thread::hardware_concurrency();
		const int workers = n < 16384 ? 1 : (int)(hw < 1 ? 1 : (hw > 8 ? 8 : hw));
		if (workers <= 1) lay_out(0, n);
		else {
			std::vector<std::thread> pool;
			const int chunk = (n + workers - 1) / workers;
			for (int w = 0; w < workers; w++) {
				const int i0 = w * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
				if (i0 < i1) pool.emplace_back(lay_out, i0, i1);
			}
			for (std::thread& th : pool) th.join();
		}
	}
	// padded boxes of groups of RT_GROUP consecutive triangles
	for (int g = 0; g < stride; g++) {      // every slot past the last group is an empty box
		float l[3] = { inf, inf, inf }, h[3] = { -inf, -inf, -inf };
		bool finite = pad - pad == 0.0f;
		for (int i = g * RT_GROUP; i < (g + 1) * RT_GROUP && i < n; i++) {
			const float* v = tris.data() + (size_t)i * RT_TRI_FLOATS + 6;
			for (int c = 0; c < 9; c++) {
				if (!(v[c] - v[c] == 0.0f)) finite = false;
				if (v[c] < l[c % 3]) l[c % 3] = v[c];
				if (v[c] > h[c % 3]) h[c % 3] = v[c];
			}
		}
		float* ql = plane(PL_GLO, g);
		float* qh = plane(PL_GHI, g);
		const bool empty = g * RT_GROUP >= n;
		for (int k = 0; k < 3; k++) {
			ql[k] = empty ? inf : (finite ? l[k] - pad : -inf);
			qh[k] = empty ? -inf : (finite ? h[k] + pad : inf);
		}
	}
	// clusters: runs of consecutive groups whose common box stays small against the scene (half its
	// scale at most, RT_CLUSTER_MAX_GROUPS groups at most).  A small object's groups share one box
	// test; a wall-sized group stays alone (its cluster test would never cull anything).
	const int n_groups = (n + RT_GROUP - 1) / RT_GROUP;
	const float limit = pad * 500.0f;          // pad = 1e-3 * scene scale
	int n_clusters = 0;
	for (int g = 0; g < n_groups;) {
		float l[3], h[3];
		for (int k = 0; k < 3; k++) { l[k] = plane(PL_GLO, g)[k]; h[k] = plane(PL_GHI, g)[k]; }
		int cnt = 1;
		while (g + cnt < n_groups && cnt < RT_CLUSTER_MAX_GROUPS) {
			float l2[3], h2[3];
			bool small = true;
			for (int k = 0; k < 3; k++) {
				l2[k] = fminf(l[k], plane(PL_GLO, g + cnt)[k]);
				h2[k] = fmaxf(h[k], plane(PL_GHI, g + cnt)[k]);
				if (!(h2[k] - l2[k] <= limit)) small = false;
			}
			if (!small) break;
			for (int k = 0; k < 3; k++) { l[k] = l2[k]; h[k] = h2[k]; }
			cnt++;
		}
		float* ql = plane(PL_CLO, n_clusters);
		float* qh = plane(PL_CHI, n_clusters);
		for (int k = 0; k < 3; k++) { ql[k] = l[k]; qh[k] = h[k]; }
		memcpy(ql + 3, &g, sizeof(int));
		memcpy(qh + 3, &cnt, sizeof(int));
		n_clusters++;
		g += cnt;
	}
	// Octant-ordered copies (rt_internal.h DevScene): one 32-byte record per box and octant, so the
	// kernel fetches near and far faces with one scalar load and needs no per-axis min/max.
	// One spare entry per octant lets the kernel prefetch entry e+1 unconditionally.
	PlaneLayout L;
	L.n_clusters = n_clusters;
	if (!with_octant_records) {        // LBVH commits: the brute-force loop is never run on them
		L.oct_clusters = L.oct_groups = 0;
		L.oct_cluster_pitch = L.oct_group_pitch = 0;
		return L;
	}
	L.oct_cluster_pitch = n_clusters + 1;
	L.oct_group_pitch = n_groups + 1;
	L.oct_clusters = (long long)out.size();
	L.oct_groups = L.oct_clusters + 8ll * L.oct_cluster_pitch * 8;
	out.resize((size_t)(L.oct_groups + 8ll * L.oct_group_pitch * 8), 0.0f);
	for (int oct = 0; oct < 8; oct++) {
		auto put = [&](float* q, const float* lo, const float* hi) {
			for (int k = 0; k < 3; k++) {
				const bool neg = (oct >> k) & 1;            // bit k: the direction's component k is negative
				q[k] = neg ? hi[k] : lo[k];
				q[3 + k] = neg ? lo[k] : hi[k];
			}
		};
		for (int c = 0; c < n_clusters; c++) {
			float* q = out.data() + L.oct_clusters + ((size_t)oct * L.oct_cluster_pitch + c) * 8;
			put(q, plane(PL_CLO, c), plane(PL_CHI, c));
			q[6] = plane(PL_CLO, c)[3];
			q[7] = plane(PL_CHI, c)[3];
		}
		for (int g = 0; g < n_groups; g++)
			put(out.data() + L.oct_groups + ((size_t)oct * L.oct_group_pitch + g) * 8, plane(PL_GLO, g), plane(PL_GHI, g));
	}
	return L;
}

} // namespace rt

using namespace rt;

extern "C" {

rt_scene* rt_scene_create(void) {
	try { return new rt_scene(); } catch (...) { set_error("rt_scene_create: out of memory"); return nullptr; }
}

int rt_scene_clear(rt_scene* s) {
	if (!s) { set_error("rt_scene_clear: NULL scene"); return RT_ERR_INVALID; }
	s->tris.clear();
	s->lights.clear();
	s->generation++;
	for (auto& kv : s->dev) kv.second.valid = false;
	return RT_OK;
}

static void touched(rt_scene* s) {
	s->generation++;
	for (auto& kv : s->dev) kv.second.valid = false;
}

int rt_scene_add_triangle(rt_scene* s, const float p1[3], const float p2[3], const float p3[3],
                          const float color[3], float shininess) {
	if (!s || !p1 || !p2 || !p3 || !color) { set_error("rt_scene_add_triangle: NULL argument"); return RT_ERR_INVALID; }
	try {
		s->tris.resize(s->tris.size() + RT_TRI_FLOATS);
	} catch (...) { set_error("rt_scene_add_triangle: out of memory"); return RT_ERR_NOMEM; }
	triangle_record(make3(p1[0], p1[1], p1[2]), make3(p2[0], p2[1], p2[2]), make3(p3[0], p3[1], p3[2]),
	                make3(color[0], color[1], color[2]), shininess, s->tris.data() + s->tris.size() - RT_TRI_FLOATS);
	touched(s);
	return RT_OK;
}

int rt_scene_add_triangles(rt_scene* s, int n, const float* p1, const float* p2, const float* p3,
                           const float* color, const float* shininess) {
	if (!s || n < 0 || (n > 0 && (!p1 || !p2 || !p3 || !color || !shininess))) {
		set_error("rt_scene_add_triangles: NULL argument or negative count");
		return RT_ERR_INVALID;
	}
	size_t base = s->tris.size();
	try {
		s->tris.resize(base + (size_t)n * RT_TRI_FLOATS);
	} catch (...) { set_error("rt_scene_add_triangles: out of memory"); return RT_ERR_NOMEM; }
	for (int i = 0; i < n; i++) {
		const float *a = p1 + 3 * i, *b = p2 + 3 * i, *c = p3 + 3 * i, *col = color + 3 * i;
		triangle_record(make3(a[0], a[1], a[2]), make3(b[0], b[1], b[2]), make3(c[0], c[1], c[2]),
		                make3(col[0], col[1], col[2]), shininess[i], s->tris.data() + base + (size_t)i * RT_TRI_FLOATS);
	}
	touched(s);
	return RT_OK;
}

int rt_scene_add_quad_half(rt_scene* s, const float translate[3], const float rotate_deg[3],
                           const float scale[3], const float color[3], float shininess) {
	if (!s || !translate || !rotate_deg || !scale || !color) { set_error("rt_scene_add_quad_half: NULL argument"); return RT_ERR_INVALID; }
	try {
		s->tris.resize(s->tris.size() + RT_TRI_FLOATS);
	} catch (...) { set_error("rt_scene_add_quad_half: out of memory"); return RT_ERR_NOMEM; }
	quad_half_record(translate, rotate_deg, scale, color, shininess, s->tris.data() + s->tris.size() - RT_TRI_FLOATS);
	touched(s);
	return RT_OK;
}

int rt_scene_add_light(rt_scene* s, const float position[3], float intensity) {
	if (!s || !position) { set_error("rt_scene_add_light: NULL argument"); return RT_ERR_INVALID; }
	try {
		s->lights.insert(s->lights.end(), { position[0], position[1], position[2], intensity });
	} catch (...) { set_error("rt_scene_add_light: out of memory"); return RT_ERR_NOMEM; }
	touched(s);
	return RT_OK;
}

int rt_scene_load_rto(rt_scene* s, const char* path) {
	// Source.cpp:631-657 loadModel — same fscanf grammar; the reference's exit(1) paths become
	// error codes.  Like the reference, a short match (ret > 0 but < 12) still calls addTriangle
	// with whatever was parsed; unlike it, the unparsed fields are zero rather than uninitialised.
	if (!s || !path) { set_error("rt_scene_load_rto: NULL argument"); return RT_ERR_INVALID; }
	FILE* file = fopen(path, "rb");
	if (file == NULL) {
		set_error("rt_scene_load_rto: %s: %s", path, strerror(errno));   // reference: "File not found"
		return RT_ERR_IO;
	}
	int added = 0;
	while (true) {
		float a1 = 0, a2 = 0, a3 = 0, b1 = 0, b2 = 0, b3 = 0, c1 = 0, c2 = 0, c3 = 0;
		int d1 = 0, d2 = 0, d3 = 0;
		int ret = fscanf(file, "(%f, %f, %f), (%f, %f, %f), (%f, %f, %f), (%d, %d, %d); ",
		                 &a1, &a2, &a3, &b1, &b2, &b3, &c1, &c2, &c3, &d1, &d2, &d3);
		if (ret == -1) break;
		if (ret == 0) {
			fclose(file);
			set_error("rt_scene_load_rto: %s: malformed record after %d triangles", path, added);   // reference: "Error with loading a model"
			return RT_ERR_IO;
		}
		const float T[3] = { a1, a2, a3 }, R[3] = { b1, b2, b3 }, S[3] = { c1, c2, c3 };
		const float col[3] = { (float)d1, (float)d2, (float)d3 };
		int rc = rt_scene_add_quad_half(s, T, R, S, col, 0);
		if (rc != RT_OK) { fclose(file); return rc; }
		added++;
	}
	fclose(file);
	return added;
}

int rt_scene_triangle_count(const rt_scene* s) { return s ? (int)(s->tris.size() / RT_TRI_FLOATS) : RT_ERR_INVALID; }
int rt_scene_light_count(const rt_scene* s) { return s ? (int)(s->lights.size() / 4) : RT_ERR_INVALID; }

int rt_scene_get_triangles(const rt_scene* s, float* out28, int capacity) {
	if (!s || (capacity > 0 && !out28) || capacity < 0) { set_error("rt_scene_get_triangles: bad argument"); return RT_ERR_INVALID; }
	int n = (int)(s->tris.size() / RT_TRI_FLOATS);
	if (n > capacity) n = capacity;
	if (n > 0) memcpy(out28, s->tris.data(), (size_t)n * RT_TRI_FLOATS * sizeof(float));
	return n;
}

int rt_camera_trans(const float rot_deg[3], float out_trans16[16]) {
	if (!rot_deg || !out_trans16) { set_error("rt_camera_trans: NULL argument"); return RT_ERR_INVALID; }
	camera_trans(rot_deg, out_trans16);
	return RT_OK;
}

// ---- camera controller ----------------------------------------------------------------------

namespace {

vec3 ld(const float* p) { return make3(p[0], p[1], p[2]); }
void st(float* p, vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// Source.cpp:126-130 / 136-140 / 202-206
vec3 camera_direction(const float rot[3]) {
	const float rx = radians(rot[0]), ry = radians(rot[1]);
	return normalize(make3(sinf(rx) * cosf(ry), sinf(ry), cosf(rx) * cosf(ry)));
}

const float kMoveBaseSpeed = 1 / 128.0f;                                              // Source.cpp:108
const float kRotBaseSpeed = (float)((double)kMoveBaseSpeed * 3.14159265 * 2 * 2);     // Source.cpp:111 (double expression)

} // namespace

int rt_camera_init(rt_camera* cam, const float pos[3], const float rot_deg[3]) {
	if (!cam || !pos || !rot_deg) { set_error("rt_camera_init: NULL argument"); return RT_ERR_INVALID; }
	for (int k = 0; k < 3; k++) { cam->position[k] = pos[k]; cam->rotations[k] = rot_deg[k]; cam->right[k] = 0.f; cam->up[k] = 0.f; }
	st(cam->direction, camera_direction(cam->rotations));
	cam->move_speed = kMoveBaseSpeed;          // Source.cpp:109
	cam->rot_speed = kRotBaseSpeed;            // Source.cpp:112
	return RT_OK;
}

int rt_camera_rotate(rt_camera* cam, int l, int r, int u, int d) {
	if (!cam) { set_error("rt_camera_rotate: NULL argument"); return RT_ERR_INVALID; }
	vec3 rot = ld(cam->rotations);
	const float s = cam->rot_speed;
	if (l) rot = rot - make3(s, 0.f, 0.f);                 // Source.cpp:182-184
	if (r) rot = rot + make3(s, 0.f, 0.f);                 // Source.cpp:185-187
	if (u) rot = rot + make3(0.f, s, 0.f);                 // Source.cpp:188-190
	if (d) rot = rot - make3(0.f, s, 0.f);                 // Source.cpp:191-193
	if ((double)rot.y > 89.99) rot.y = (float)89.99;       // Source.cpp:195-197 (double literal)
	if ((double)rot.y < -89.99) rot.y = (float)-89.99;     // Source.cpp:198-200
	st(cam->rotations, rot);
	st(cam->direction, camera_direction(cam->rotations));  // Source.cpp:202-206
	return RT_OK;
}

int rt_camera_translate(rt_camera* cam, int l, int r, int u, int d, int f, int b) {
	if (!cam) { set_error("rt_camera_translate: NULL argument"); return RT_ERR_INVALID; }
	const vec3 world_up = make3(0.f, 1.f, 0.f);            // Source.cpp:107
	const vec3 dir = ld(cam->direction);
	const vec3 right = normalize(cross(world_up, dir));    // Source.cpp:149
	const vec3 up = normalize(cross(dir, right));          // Source.cpp:152
	vec3 p = ld(cam->position);
	const float s = cam->move_speed;
	if (l) p = p - right * s;                              // Source.cpp:158-160
	if (r) p = p + right * s;
	if (u) p = p + up * s;
	if (d) p = p - up * s;
	if (f) p = p + dir * s;
	if (b) p = p - dir * s;                                // Source.cpp:173-176
	st(cam->right, right);
	st(cam->up, up);
	st(cam->position, p);
	return RT_OK;
}

int rt_camera_frame_time(rt_camera* cam, float milis) {
	if (!cam) { set_error("rt_camera_frame_time: NULL argument"); return RT_ERR_INVALID; }
	cam->move_speed = kMoveBaseSpeed * milis;              // Source.cpp:1307
	cam->rot_speed = kRotBaseSpeed * milis;                // Source.cpp:1308
	return RT_OK;
}

int rt_camera_step(rt_camera* cam, unsigned int k, float milis) {
	if (!cam) { set_error("rt_camera_step: NULL argument"); return RT_ERR_INVALID; }
	rt_camera_rotate(cam, !!(k & RT_KEY_LOOK_LEFT), !!(k & RT_KEY_LOOK_RIGHT), !!(k & RT_KEY_LOOK_UP), !!(k & RT_KEY_LOOK_DOWN));   // Source.cpp:1294
	rt_camera_translate(cam, !!(k & RT_KEY_LEFT), !!(k & RT_KEY_RIGHT), !!(k & RT_KEY_UP), !!(k & RT_KEY_DOWN),
	                    !!(k & RT_KEY_FORWARD), !!(k & RT_KEY_BACKWARD));                                                             // Source.cpp:1295
	return rt_camera_frame_time(cam, milis);
}

int rt_camera_path(rt_camera* cam, int n_frames, const unsigned int* key_masks, const float* frame_ms,
                   float* out_pos3, float* out_trans16) {
	if (!cam || n_frames < 0 || (n_frames > 0 && (!key_masks || !frame_ms || !out_pos3 || !out_trans16))) {
		set_error("rt_camera_path: bad argument");
		return RT_ERR_INVALID;
	}
	for (int i = 0; i < n_frames; i++) {
		for (int k = 0; k < 3; k++) out_pos3[i * 3 + k] = cam->position[k];
		camera_trans(cam->rotations, out_trans16 + (size_t)i * 16);        // Source.cpp:1145-1147
		rt_camera_step(cam, key_masks[i], frame_ms[i]);
	}
	return RT_OK;
}

} // extern "C"
