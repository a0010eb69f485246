/* rt_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See rt_oracle.h.
 *
 * Every function restates one piece of /root/reference/RayTracing/src/Source.cpp ("Src") or of
 * the GLM 0.9.9.9 headers vendored next to it ("glm/" = RayTracing/Depend/include/glm/), in the
 * exact operation order those files use.  All arithmetic is IEEE binary32 with no contraction
 * (compile with -ffp-contract=off) except the two double-precision spots the reference has
 * (pixel -> u,v and the /255.0/255.0f light scale).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
typedef struct { v4 c[4]; } m4;          /* column-major: c[col] */

/* ---- glm vector helpers ------------------------------------------------------------------ */

static inline v3 v3sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 v3muls(v3 a, float s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }

/* glm/detail/func_geometric.inl:48-55  tmp = a*b; tmp.x + tmp.y + tmp.z */
static inline float dot3(v3 a, v3 b) {
	float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
	return tx + ty + tz;
}

/* glm/detail/func_geometric.inl:74-77 */
static inline v3 cross3(v3 x, v3 y) {
	v3 r = { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
	return r;
}

/* glm/detail/func_geometric.inl:88  v * inversesqrt(dot(v,v)); inversesqrt = 1/sqrt
 * (glm/detail/func_exponential.inl:136-139) */
static inline v3 normalize3(v3 v) {
	float inv = 1.0f / sqrtf(dot3(v, v));
	return v3muls(v, inv);
}

/* glm/detail/func_geometric.inl:108  I - N * dot(N, I) * 2 */
static inline v3 reflect3(v3 I, v3 N) {
	float d = dot3(N, I);
	v3 nd = v3muls(N, d);
	v3 nd2 = v3muls(nd, 2.0f);
	return v3sub(I, nd2);
}

static inline v4 v4muls(v4 a, float s) { v4 r = { a.x * s, a.y * s, a.z * s, a.w * s }; return r; }
static inline v4 v4add(v4 a, v4 b) { v4 r = { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; return r; }

static m4 m4identity(void) {
	m4 m;
	memset(&m, 0, sizeof m);
	m.c[0].x = 1.0f; m.c[1].y = 1.0f; m.c[2].z = 1.0f; m.c[3].w = 1.0f;
	return m;
}

/* glm/trigonometric: degrees * 0.01745329251994329576923690768489 (cast to float) */
static inline float radiansf(float deg) { return deg * 0.01745329251994329576923690768489f; }

/* glm/ext/matrix_transform.inl:10-15 */
static m4 m4translate(m4 m, v3 v) {
	m4 r = m;
	r.c[3] = v4add(v4add(v4add(v4muls(m.c[0], v.x), v4muls(m.c[1], v.y)), v4muls(m.c[2], v.z)), m.c[3]);
	return r;
}

/* glm/ext/matrix_transform.inl:18-46 */
static m4 m4rotate(m4 m, float angle, v3 v) {
	float a = angle;
	float c = cosf(a);
	float s = sinf(a);
	v3 axis = normalize3(v);
	v3 temp = v3muls(axis, 1.0f - c);           /* (T(1) - c) * axis */
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
	m4 r;
	for (int i = 0; i < 3; i++)
		r.c[i] = v4add(v4add(v4muls(m.c[0], R[i][0]), v4muls(m.c[1], R[i][1])), v4muls(m.c[2], R[i][2]));
	r.c[3] = m.c[3];
	return r;
}

/* glm/ext/matrix_transform.inl:78-86 */
static m4 m4scale(m4 m, v3 v) {
	m4 r;
	r.c[0] = v4muls(m.c[0], v.x);
	r.c[1] = v4muls(m.c[1], v.y);
	r.c[2] = v4muls(m.c[2], v.z);
	r.c[3] = m.c[3];
	return r;
}

/* glm/detail/type_mat4x4.inl:630-648 */
static m4 m4mul(m4 a, m4 b) {
	m4 r;
	for (int i = 0; i < 4; i++) {
		v4 bc = b.c[i];
		r.c[i] = v4add(v4add(v4add(v4muls(a.c[0], bc.x), v4muls(a.c[1], bc.y)), v4muls(a.c[2], bc.z)), v4muls(a.c[3], bc.w));
	}
	return r;
}

/* glm/detail/type_mat4x4.inl:561-572  (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
static v4 m4mulv(m4 m, v4 v) {
	v4 add0 = v4add(v4muls(m.c[0], v.x), v4muls(m.c[1], v.y));
	v4 add1 = v4add(v4muls(m.c[2], v.z), v4muls(m.c[3], v.w));
	return v4add(add0, add1);
}

/* ---- scene construction ------------------------------------------------------------------ */

static inline v3 ld3(const float* p) { v3 r = { p[0], p[1], p[2] }; return r; }
static inline void st3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

static void triangle_init(v3 p1, v3 p2, v3 p3, v3 color, float shininess, float* out) {
	/* Src:45-59 */
	v3 n = normalize3(cross3(v3sub(p1, p2), v3sub(p1, p3)));
	st3(out + 0, p1);                  /* p = p1 */
	st3(out + 3, n);
	st3(out + 6, p1);
	st3(out + 9, p2);
	st3(out + 12, p3);
	st3(out + 15, v3sub(p2, p1));      /* edge0 */
	st3(out + 18, v3sub(p3, p2));      /* edge1 */
	st3(out + 21, v3sub(p1, p3));      /* edge2 */
	st3(out + 24, color);
	out[27] = shininess;
}

void orc_triangle_init(const float p1[3], const float p2[3], const float p3[3],
                       const float color[3], float shininess, float out28[ORC_TRI_FLOATS]) {
	triangle_init(ld3(p1), ld3(p2), ld3(p3), ld3(color), shininess, out28);
}

void orc_quad_half(const float translate[3], const float rotate_deg[3], const float scale[3],
                   const float color[3], float shininess, float out28[ORC_TRI_FLOATS]) {
	/* Src:410-439 */
	v4 p1 = { -1, -1, 0, 1 }, p2 = { -1, 1, 0, 1 }, p3 = { 1, -1, 0, 1 };
	m4 I = m4identity();
	v3 ax = { 1, 0, 0 }, ay = { 0, 1, 0 }, az = { 0, 0, 1 };
	m4 T = m4translate(I, ld3(translate));
	m4 Rx = m4rotate(I, radiansf(rotate_deg[0]), ax);
	m4 Ry = m4rotate(I, radiansf(rotate_deg[1]), ay);
	m4 Rz = m4rotate(I, radiansf(rotate_deg[2]), az);
	m4 S = m4scale(I, ld3(scale));
	m4 mod = m4mul(m4mul(m4mul(m4mul(T, Rx), Ry), Rz), S);
	p1 = m4mulv(mod, p1);
	p2 = m4mulv(mod, p2);
	p3 = m4mulv(mod, p3);
	v3 a = { p1.x, p1.y, p1.z }, b = { p2.x, p2.y, p2.z }, c = { p3.x, p3.y, p3.z };
	triangle_init(a, b, c, ld3(color), shininess, out28);
}

void orc_camera_trans(const float rot_deg[3], float out16[16]) {
	/* Src:1145-1147 */
	v3 ay = { 0.0f, 1.0f, 0.0f }, ax = { 1.0f, 0.0f, 0.0f };
	m4 t = m4identity();
	t = m4rotate(t, radiansf(rot_deg[0]), ay);
	t = m4rotate(t, radiansf(-rot_deg[1]), ax);
	memcpy(out16, &t, sizeof t);
}

/* ---- Camera (Src:95-212) and the frame loop's tail (Src:1294-1295, 1307-1308) --------------- */

static v3 cam_direction(float rx_deg, float ry_deg) {
	/* Src:136-140 == 202-206 */
	float rx = radiansf(rx_deg), ry = radiansf(ry_deg);
	v3 d = { sinf(rx) * cosf(ry), sinf(ry), cosf(rx) * cosf(ry) };
	return normalize3(d);
}

void orc_camera_init(float c[ORC_CAM_FLOATS], const float pos[3], const float rot_deg[3]) {
	float move_base = 1 / 128.0f;                                   /* Src:108 */
	float rot_base = (float)(move_base * 3.14159265 * 2 * 2);       /* Src:111: float * double literal */
	memset(c, 0, sizeof(float) * ORC_CAM_FLOATS);
	memcpy(c + 0, pos, sizeof(float) * 3);                          /* Src:133 */
	memcpy(c + 3, rot_deg, sizeof(float) * 3);                      /* Src:134 */
	v3 d = cam_direction(rot_deg[0], rot_deg[1]);
	c[6] = d.x; c[7] = d.y; c[8] = d.z;
	c[15] = move_base;                                              /* Src:109 */
	c[16] = rot_base;                                               /* Src:112 */
}

void orc_camera_step(float c[ORC_CAM_FLOATS], const int k[10], float milis) {
	float move_base = 1 / 128.0f;
	float rot_base = (float)(move_base * 3.14159265 * 2 * 2);
	float rot_speed = c[16], move_speed = c[15];
	/* rotate(lleft, lright, lup, ldown)  Src:181-209: whole-vector += / -= */
	float rx = c[3], ry = c[4], rz = c[5];
	if (k[0]) { rx = rx - rot_speed; ry = ry - 0.0f; rz = rz - 0.0f; }
	if (k[1]) { rx = rx + rot_speed; ry = ry + 0.0f; rz = rz + 0.0f; }
	if (k[2]) { rx = rx + 0.0f; ry = ry + rot_speed; rz = rz + 0.0f; }
	if (k[3]) { rx = rx - 0.0f; ry = ry - rot_speed; rz = rz - 0.0f; }
	if (ry > 89.99) ry = 89.99;                                     /* Src:195-197: compared and assigned as double literals */
	if (ry < -89.99) ry = -89.99;                                   /* Src:198-200 */
	c[3] = rx; c[4] = ry; c[5] = rz;
	v3 dir = cam_direction(rx, ry);
	c[6] = dir.x; c[7] = dir.y; c[8] = dir.z;
	/* translate(left, right, up, down, forward, backward)  Src:144-179 */
	v3 world_up = { 0.0f, 1.0f, 0.0f };
	v3 right = normalize3(cross3(world_up, dir));                   /* Src:149 */
	v3 up = normalize3(cross3(dir, right));                         /* Src:152 */
	v3 p = { c[0], c[1], c[2] };
	if (k[4]) p = v3sub(p, v3muls(right, move_speed));
	if (k[5]) { v3 m = v3muls(right, move_speed); p.x += m.x; p.y += m.y; p.z += m.z; }
	if (k[6]) { v3 m = v3muls(up, move_speed); p.x += m.x; p.y += m.y; p.z += m.z; }
	if (k[7]) p = v3sub(p, v3muls(up, move_speed));
	if (k[8]) { v3 m = v3muls(dir, move_speed); p.x += m.x; p.y += m.y; p.z += m.z; }
	if (k[9]) p = v3sub(p, v3muls(dir, move_speed));
	c[0] = p.x; c[1] = p.y; c[2] = p.z;
	c[9] = right.x; c[10] = right.y; c[11] = right.z;
	c[12] = up.x; c[13] = up.y; c[14] = up.z;
	c[15] = move_base * milis;                                      /* Src:1307 */
	c[16] = rot_base * milis;                                       /* Src:1308 */
}

/* ---- castRay ----------------------------------------------------------------------------- */

typedef struct {
	v3 point;
	float distance;
	int cur;        /* index into the triangle array, -1 == NULL */
	int didHit;
} payload;

static payload cast_ray(const float* tris, int n, v3 orgin, v3 dir, int curr, orc_counters* ctr) {
	/* Src:277-405 (live lines 278, 301-305, 383-399, 403) */
	payload closest = { { 0, 0, 0 }, 1e9f, -1, 0 };
	uint64_t np = 0, npt = 0, ne = 0;
	for (int i = 0; i < n; i++) {
		if (i == curr) continue;                                   /* Src:302 */
		const float* T = tris + (size_t)i * ORC_TRI_FLOATS;
		v3 tn = ld3(T + 3), p1 = ld3(T + 6);
		np++;
		float t = dot3(v3sub(p1, orgin), tn) / dot3(dir, tn);      /* Src:383 */
		if (t < 0) continue;                                       /* Src:385 */
		npt++;
		v3 I = { orgin.x + t * dir.x, orgin.y + t * dir.y, orgin.z + t * dir.z };   /* Src:389 */
		if (t < closest.distance && (double)t > 0.00001) {         /* Src:391 */
			v3 p2 = ld3(T + 9), p3 = ld3(T + 12);
			v3 e0 = ld3(T + 15), e1 = ld3(T + 18), e2 = ld3(T + 21);
			ne++;
			if (dot3(tn, cross3(e0, v3sub(I, p1))) > 0.0) {        /* Src:392 */
				ne++;
				if (dot3(tn, cross3(e1, v3sub(I, p2))) > 0.0) {    /* Src:393 */
					ne++;
					if (dot3(tn, cross3(e2, v3sub(I, p3))) > 0.0) {/* Src:394 */
						closest.point = I;                         /* Src:396 */
						closest.distance = t;
						closest.cur = i;
						closest.didHit = 1;
					}
				}
			}
		}
	}
	if (ctr) { ctr->n_plane += np; ctr->n_point += npt; ctr->n_edge += ne; }
	return closest;
}

int orc_cast_ray(const float* tris28, int n_tris, const float origin[3], const float dir[3],
                 int skip, float out6[6], orc_counters* ctr) {
	payload p = cast_ray(tris28, n_tris, ld3(origin), ld3(dir), skip, ctr);
	out6[0] = p.didHit ? 1.0f : 0.0f;
	out6[1] = (float)p.cur;
	out6[2] = p.distance;
	st3(out6 + 3, p.point);
	return p.didHit;
}

/* ---- shading ----------------------------------------------------------------------------- */

/* Src:443-453 */
static inline v3 new_normal(v3 dir, v3 normal) {
	if (dot3(dir, normal) < 0) return normal;
	return v3muls(normal, -1.0f);
}

/* Src:26-28 */
static inline float magnitude(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

/* Src:460-494 (everything after `continue;` at Src:494 is unreachable) */
static v3 process_lighting(const float* tris, int n, const float* lights, int nl, payload hit,
                           orc_counters* ctr, int* any_lit) {
	v3 color = { 0, 0, 0 };
	v3 hc = ld3(tris + (size_t)hit.cur * ORC_TRI_FLOATS + 24);
	for (int li = 0; li < nl; li++) {
		v3 lpos = ld3(lights + 4 * li);
		float intensity = lights[4 * li + 3];
		v3 ray = v3sub(lpos, hit.point);
		v3 lightDir = normalize3(ray);                                     /* Src:472 */
		if (ctr) ctr->rays_shadow++;
		payload hit2 = cast_ray(tris, n, hit.point, lightDir, hit.cur, ctr);   /* Src:473 */
		float dist = magnitude(ray);                                       /* Src:474 */
		if (!hit2.didHit || hit2.distance > dist) {                        /* Src:476 */
			float dist_ = dist;                                            /* Src:487 (same value) */
			/* Src:488: float*float*float, then /255.0 (double) /255.0f (promoted), then to float */
			float f = 1.0f / (dist_ * dist_) * intensity;
			float modify = (float)((double)f / 255.0 / (double)255.0f);
			color.x = color.x + modify * hc.x;                             /* Src:489 */
			color.y = color.y + modify * hc.y;
			color.z = color.z + modify * hc.z;
			*any_lit = 1;
		}
	}
	return color;
}

static void pixel(const float* tris, int n, const float* lights, int nl, v3 cam, const m4* trans,
                  int x, int y, int width, int height, int maxBounces, float* data, orc_counters* ctr) {
	/* Src:530-628 */
	float v = (float)((y - height / 2.0) / (float)height);                 /* Src:534 */
	float u = (float)((x - width / 2.0) / (float)width);                   /* Src:535 */
	v4 in = { u, v, 0.5f, 1.0f };                                          /* Src:539, depth Src:118 */
	v4 pixel2 = m4mulv(*trans, in);
	v3 dir = { pixel2.x, pixel2.y, pixel2.z };
	v3 origin = cam;
	if (ctr) ctr->rays_primary++;
	payload hit = cast_ray(tris, n, origin, dir, -1, ctr);                 /* Src:548 */
	int index = (y * width * 3) + x * 3;                                   /* Src:550 */
	v3 color = { 0, 0, 0 };
	if (hit.didHit) {
		const float* HT = tris + (size_t)hit.cur * ORC_TRI_FLOATS;
		v3 normal = new_normal(dir, ld3(HT + 3));                          /* Src:561 */
		int lit = 0;
		if (ctr) ctr->px_hit++;
		if (HT[27] > .5) {                                                 /* Src:570 */
			int bounces = 0;
			if (ctr) { ctr->px_mirror++; ctr->rays_bounce++; }
			v3 newDir = normalize3(reflect3(dir, normal));                 /* Src:572 */
			payload hitN1 = cast_ray(tris, n, hit.point, newDir, hit.cur, ctr);   /* Src:573 */
			while (hitN1.didHit && bounces < maxBounces &&
			       tris[(size_t)hitN1.cur * ORC_TRI_FLOATS + 27] > .5) {   /* Src:575 */
				normal = new_normal(newDir, ld3(tris + (size_t)hitN1.cur * ORC_TRI_FLOATS + 3));
				newDir = normalize3(reflect3(newDir, normal));
				if (ctr) ctr->rays_bounce++;
				hitN1 = cast_ray(tris, n, hitN1.point, newDir, hitN1.cur, ctr);
				bounces++;
			}
			if (hitN1.didHit && bounces <= maxBounces) {                   /* Src:582 */
				color = process_lighting(tris, n, lights, nl, hitN1, ctr, &lit);
				float dim = (float)(bounces * 1);                          /* Src:585 */
				color.x = color.x * dim; color.y = color.y * dim; color.z = color.z * dim;
			} else {
				color.x = color.y = color.z = 0;
			}
		} else {
			color = process_lighting(tris, n, lights, nl, hit, ctr, &lit); /* Src:597 */
		}
		if (ctr && lit) ctr->px_lit++;
		if (color.x > 1.0) color.x = 1.0f;                                 /* Src:602-610 */
		if (color.y > 1.0) color.y = 1.0f;
		if (color.z > 1.0) color.z = 1.0f;
		data[index] = color.x; data[index + 1] = color.y; data[index + 2] = color.z;
	} else {
		data[index] = 0.0f; data[index + 1] = 0.0f; data[index + 2] = 0.0f;   /* Src:623-625 */
	}
}

void orc_render_rows(const float* tris28, int n_tris, const float* lights4, int n_lights,
                     const float cam_pos[3], const float trans16[16], int width, int height,
                     int max_bounces, int y0, int y1, int ystep, float* out_rgb, orc_counters* ctr) {
	m4 trans;
	memcpy(&trans, trans16, sizeof trans);
	v3 cam = ld3(cam_pos);
	if (ystep < 1) ystep = 1;
	for (int y = y0; y < y1 && y < height; y += ystep)                     /* Src:1248-1252 */
		for (int x = 0; x < width; x++)
			pixel(tris28, n_tris, lights4, n_lights, cam, &trans, x, y, width, height, max_bounces, out_rgb, ctr);
}

void orc_render(const float* tris28, int n_tris, const float* lights4, int n_lights,
                const float cam_pos[3], const float trans16[16], int width, int height,
                int max_bounces, float* out_rgb, orc_counters* ctr) {
	orc_render_rows(tris28, n_tris, lights4, n_lights, cam_pos, trans16, width, height, max_bounces,
	                0, height, 1, out_rgb, ctr);
}

typedef struct {
	const float* tris; int n; const float* lights; int nl; const float* cam; const float* trans;
	int w, h, depth, y0, ystep; float* out; orc_counters ctr; int want_ctr;
} mt_job;

static void* mt_main(void* arg) {
	mt_job* j = (mt_job*)arg;
	orc_render_rows(j->tris, j->n, j->lights, j->nl, j->cam, j->trans, j->w, j->h, j->depth,
	                j->y0, j->h, j->ystep, j->out, j->want_ctr ? &j->ctr : NULL);
	return NULL;
}

void orc_render_mt(const float* tris28, int n_tris, const float* lights4, int n_lights,
                   const float cam_pos[3], const float trans16[16], int width, int height,
                   int max_bounces, int n_threads, float* out_rgb, orc_counters* ctr) {
	if (n_threads < 1) n_threads = 1;
	if (n_threads > 1024) n_threads = 1024;
	mt_job* jobs = (mt_job*)calloc((size_t)n_threads, sizeof *jobs);
	pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof *th);
	for (int i = 0; i < n_threads; i++) {
		mt_job j = { tris28, n_tris, lights4, n_lights, cam_pos, trans16, width, height, max_bounces,
		             i, n_threads, out_rgb, { 0 }, ctr != NULL };
		jobs[i] = j;
		pthread_create(&th[i], NULL, mt_main, &jobs[i]);
	}
	for (int i = 0; i < n_threads; i++) {
		pthread_join(th[i], NULL);
		if (ctr) {
			const orc_counters* c = &jobs[i].ctr;
			ctr->rays_primary += c->rays_primary; ctr->rays_bounce += c->rays_bounce;
			ctr->rays_shadow += c->rays_shadow; ctr->n_plane += c->n_plane;
			ctr->n_point += c->n_point; ctr->n_edge += c->n_edge;
			ctr->px_hit += c->px_hit; ctr->px_mirror += c->px_mirror; ctr->px_lit += c->px_lit;
		}
	}
	free(jobs);
	free(th);
}

/* ---- the reference's OTHER renderer: RayTracing/src/Compute.shader ("CS"), the GL compute shader ----
 * PARITY UNPINNED: there is no GL in this image to run the shader, and GLSL leaves the precision of /, sqrt,
 * normalize and the summation order of mat4 * vec4 to the driver.  This restates the shader's written
 * operation order in IEEE binary32 with glm's orders for the built-ins (dot3, cross3, normalize3, m4mulv
 * above); it is what RT_SHADE_GLSL is checked against, nothing more. */

/* CS:140-175 castRay: the loop of Src:301-402 with an index for the skipped triangle and the literal .00001
 * as a binary32 constant (same decisions: for a binary32 t, t > 1e-5f  <=>  (double)t > 0.00001). */
static payload glsl_cast_ray(const float* tris, int n, v3 orgin, v3 dir, int index) {
	payload closest = { { 0, 0, 0 }, 1e9f, -1, 0 };
	for (int i = 0; i < n; i++) {
		if (i == index) continue;                                          /* CS:144 */
		const float* T = tris + (size_t)i * ORC_TRI_FLOATS;
		v3 nn = ld3(T + 3), p1 = ld3(T + 6), p2 = ld3(T + 9), p3 = ld3(T + 12);
		float t = dot3(v3sub(p1, orgin), nn) / dot3(dir, nn);              /* CS:152 */
		if (t < 0) continue;                                               /* CS:153 */
		v3 I = { orgin.x + t * dir.x, orgin.y + t * dir.y, orgin.z + t * dir.z };   /* CS:157 */
		if (t < closest.distance && t > .00001f) {                         /* CS:159 */
			if (dot3(nn, cross3(ld3(T + 15), v3sub(I, p1))) > 0.0f &&      /* CS:161-163 */
			    dot3(nn, cross3(ld3(T + 18), v3sub(I, p2))) > 0.0f &&
			    dot3(nn, cross3(ld3(T + 21), v3sub(I, p3))) > 0.0f) {
				closest.point = I;
				closest.distance = t;
				closest.cur = i;
				closest.didHit = 1;
			}
		}
	}
	return closest;
}

/* CS:224-345 main() per pixel, CS:181-219 processLighting.  out_rgb as orc_render (the shader's alpha = 1 is
 * dropped); nothing is clamped (imageStore into rgba32f, CS:7, 345). */
void orc_render_glsl(const float* tris28, int n_tris, const float* lights4, int n_lights,
                     const float cam_pos[3], const float trans16[16], int width, int height, float* out_rgb) {
	m4 trans;
	memcpy(&trans, trans16, sizeof trans);
	v3 cam = ld3(cam_pos);
	for (int y = 0; y < height; y++)
		for (int x = 0; x < width; x++) {
			float v = ((float)y - (float)height / 2.0f) / (float)height;   /* CS:235 (GLSL: all binary32) */
			float u = ((float)x - (float)width / 2.0f) / (float)width;     /* CS:236 */
			v4 in = { u, v, .5f, 1.0f };
			v4 pixel2 = m4mulv(trans, in);                                 /* CS:239 */
			v3 dir = { pixel2.x, pixel2.y, pixel2.z };
			payload hit = glsl_cast_ray(tris28, n_tris, cam, dir, -1);     /* CS:251 */
			v3 color = { 0, 0, 0 };                                        /* CS:253 */
			if (hit.didHit) {                                              /* CS:257 */
				float result = 0.0f;                                       /* CS:182: vec3 of three equal scalars */
				for (int i = 0; i < n_lights; i++) {                       /* CS:188 */
					v3 ray = v3sub(ld3(lights4 + 4 * i), hit.point);       /* CS:193 */
					v3 lightDir = normalize3(ray);                         /* CS:196 */
					payload hit2 = glsl_cast_ray(tris28, n_tris, hit.point, lightDir, hit.cur);   /* CS:197 */
					float dist = sqrtf(ray.x * ray.x + ray.y * ray.y + ray.z * ray.z);            /* CS:177-179, 198 */
					if (!hit2.didHit || hit2.distance > dist) {            /* CS:203 */
						float dist_ = dist;                                /* CS:213 */
						result = result + 1.0f / (dist_ * dist_) * lights4[4 * i + 3] / 255.0f;   /* CS:214 */
					}
				}
				const float* HC = tris28 + (size_t)hit.cur * ORC_TRI_FLOATS + 24;
				float lighting = 0.0f + result;                            /* CS:330: indirect (CS:272, 0) + direct */
				color.x = lighting * HC[0] / 255.0f;
				color.y = lighting * HC[1] / 255.0f;
				color.z = lighting * HC[2] / 255.0f;
			}
			float* px = out_rgb + ((size_t)y * (size_t)width + (size_t)x) * 3;
			px[0] = color.x; px[1] = color.y; px[2] = color.z;             /* CS:342-345 */
		}
}

void orc_quantize(const float* rgb, size_t n, uint8_t* out) {
	for (size_t i = 0; i < n; i++) {
		float c = rgb[i];
		if (!(c > 0.0f)) { out[i] = 0; continue; }
		if (c > 1.0f) c = 1.0f;
		out[i] = (uint8_t)(int)(c * 255.0f + 0.5f);
	}
}

double orc_flops(const orc_counters* c) {
	return 14.0 * (double)c->n_plane + 6.0 * (double)c->n_point + 17.0 * (double)c->n_edge +
	       32.0 * (double)c->rays_primary + 29.0 * (double)c->rays_bounce + 30.0 * (double)c->rays_shadow;
}
