/* rt_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-precision, unfused CPU restatement of the per-pixel render loop of
 * Expasito/RayTracing (reference file RayTracing/src/Source.cpp, "Src" below).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have something to check the HIP
 * path against and to time on the host.  Nothing under raytracing_amd/ may include, link, load or
 * call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py compares every function here, bit for bit,
 * with fixtures in tests/golden/ that were produced by the reference's own getPixelData /
 * castRay / addTriangle compiled from /root/reference (oracle/Makefile target `ref`,
 * tests/golden/make_golden.py).  Build with -O2 -ffp-contract=off (MSVC x64 /fp:precise, no FMA).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One triangle = 28 floats, the reference's `struct Triangle` flattened in declaration order
 * (Src:31-43): p[3] n[3] p1[3] p2[3] p3[3] edge0[3] edge1[3] edge2[3] color[3] shininess. */
#define ORC_TRI_FLOATS 28

/* Work counters, in the units SURVEY.md §8(d) prices. */
typedef struct {
	uint64_t rays_primary, rays_bounce, rays_shadow;
	uint64_t n_plane;   /* triangles visited (plane step, 14 FLOP each)              */
	uint64_t n_point;   /* of those, t >= 0 or NaN (point step, 6 FLOP each)         */
	uint64_t n_edge;    /* edge terms actually evaluated (17 FLOP each)              */
	uint64_t px_hit, px_mirror, px_lit;   /* pixels: primary hit / mirror branch / >=1 light unshadowed */
} orc_counters;

/* Src:45-59  Triangle::Triangle */
void orc_triangle_init(const float p1[3], const float p2[3], const float p3[3],
                       const float color[3], float shininess, float out28[ORC_TRI_FLOATS]);

/* Src:410-439  addTriangle (translate, rotate in degrees, scale) */
void orc_quad_half(const float translate[3], const float rotate_deg[3], const float scale[3],
                   const float color[3], float shininess, float out28[ORC_TRI_FLOATS]);

/* Src:1145-1147  per-frame `trans`; out is column-major like glm::value_ptr */
void orc_camera_trans(const float rot_deg[3], float out16[16]);

/* Src:95-212 Camera, Src:1294-1295 + 1307-1308 the frame loop's tail.  cam17 = position[3],
 * rotations[3] (degrees), direction[3], right[3], up[3], moveSpeed, rotSpeed. */
#define ORC_CAM_FLOATS 17
void orc_camera_init(float cam17[ORC_CAM_FLOATS], const float pos[3], const float rot_deg[3]);
/* keys10 = look left, right, up, down (Src:241), move left, right, up, down, forward, backward
 * (Src:237): rotate, translate, then the speeds from the frame time in ms. */
void orc_camera_step(float cam17[ORC_CAM_FLOATS], const int keys10[10], float milis);

/* Src:277-405  castRay.  skip < 0 == NULL.  out6 = {didHit, index|-1, distance, point xyz}.
 * Returns didHit. */
int orc_cast_ray(const float* tris28, int n_tris, const float origin[3], const float dir[3],
                 int skip, float out6[6], orc_counters* ctr);

/* Src:530-628 getPixelData for rows y = y0, y0+ystep, ... < y1 and all x (frame loop
 * Src:1248-1252).  out_rgb is the full frame, data[(y*w+x)*3+c].  ctr may be NULL. */
void orc_render_rows(const float* tris28, int n_tris, const float* lights4, int n_lights,
                     const float cam_pos[3], const float trans16[16], int width, int height,
                     int max_bounces, int y0, int y1, int ystep, float* out_rgb, orc_counters* ctr);

/* whole frame, one thread */
void orc_render(const float* tris28, int n_tris, const float* lights4, int n_lights,
                const float cam_pos[3], const float trans16[16], int width, int height,
                int max_bounces, float* out_rgb, orc_counters* ctr);

/* whole frame, rows interleaved over n_threads pthreads (stands in for the reference's disabled
 * std::execution::par loop, Src:1255-1264).  Same pixels, same bits. */
void orc_render_mt(const float* tris28, int n_tris, const float* lights4, int n_lights,
                   const float cam_pos[3], const float trans16[16], int width, int height,
                   int max_bounces, int n_threads, float* out_rgb, orc_counters* ctr);

/* Compute.shader:140-345, the reference's GL compute-shader renderer restated in IEEE binary32 (no mirrors,
 * binary32 light scale, colour applied after the sum, no clamp) — PARITY UNPINNED, see rt_oracle.c.
 * The checker of RT_SHADE_GLSL. */
void orc_render_glsl(const float* tris28, int n_tris, const float* lights4, int n_lights,
                     const float cam_pos[3], const float trans16[16], int width, int height, float* out_rgb);

/* The GL unorm rule both sides of every 8-bit comparison use (note after the table):
 * q = (int)(c*255.0f + 0.5f), c<0 or NaN -> 0. */
void orc_quantize(const float* rgb, size_t n, uint8_t* out);

/* 14*n_plane + 6*n_point + 17*n_edge + 32*primary + 29*bounce + 30*shadow */
double orc_flops(const orc_counters* c);

#ifdef __cplusplus
}
#endif
#endif
