// rt_internal.h — types shared by the host side (rt_api.cpp, rt_scene.cpp) and the kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/rt355.h"
#include "rt_vec.h"

namespace rt {

// ---- device scene: SoA of float4 planes in HBM -------------------------------------------------
// plane k of triangle i lives at planes[k * stride + i]  (stride = triangle count rounded up to 64
// so every plane starts 1 KiB-aligned and a wave's 64 x 16 B staging loads are fully coalesced).
//   plane 0  N  = (n.x,  n.y,  n.z,  shininess)         Source.cpp:34, 43
//   plane 1  P1 = (p1.x, p1.y, p1.z, color.r)           Source.cpp:36, 41
//   plane 2  P2 = (p2.x, p2.y, p2.z, color.g)
//   plane 3  P3 = (p3.x, p3.y, p3.z, color.b)
//   plane 4  E0 = (edge0.xyz, 0)                        Source.cpp:52
//   plane 5  E1 = (edge1.xyz, 0)                        Source.cpp:53
//   plane 6  E2 = (edge2.xyz, 0)                        Source.cpp:54
//   plane 7  GLO[g] = lower corner of the padded box of triangles 4g..4g+3 (entries g < stride/4)
//   plane 8  GHI[g] = upper corner            (an empty group has GLO = +inf, GHI = -inf)
//   plane 9  CLO[c] = (lower corner of cluster c, first group as int bits)   clusters = runs of
//   plane 10 CHI[c] = (upper corner,              group count as int bits)   consecutive groups
//   plane 11 COL = (color.r, color.g, color.b, shininess): what shading reads by hit index, in one load
// The `p` field of the reference record (Source.cpp:33, always == p1) is not uploaded.
enum { PL_N = 0, PL_P1, PL_P2, PL_P3, PL_E0, PL_E1, PL_E2, PL_GLO, PL_GHI, PL_CLO, PL_CHI, PL_COL, PL_M0, PL_M1, PL_M2, PL_COUNT };   // PL_M*: edge_filter_planes (rt_scene.cpp)
#define RT_CLUSTER_MAX_GROUPS 16
#define RT_BVH4_EMPTY 0x7ffffffe     /* unused child slot of a 4-wide node */
#define RT_AOS_F4 10                 /* float4 per triangle record of DevScene::tri_aos and DevScene::hoist */
#ifndef RT_GROUP
#define RT_GROUP 2
#endif

// Up to this many triangles the brute-force kernel keeps a per-workgroup LDS table of the primary
// rays' numerators (rt_kernels.hip, ACCEL 0); forced brute force on larger scenes runs without it
// (ACCEL 1).  The scene itself is read through the scalar cache in both.
#define RT_LDS_MAX_TRIS 1024


// LBVH: at most this many scene-spanning triangles are hoisted out of the tree (DevScene::hoist)
#define RT_HOIST_MAX 32

// Per-light direction grid for shadow rays (rt_lgrid.hip): one record per light in device memory (DevScene::lgrid).
#define RT_LGRID_MAX_LIGHTS 4
#define RT_LGRID_OVER_MAX 32
struct LightGrid {
	const int* cell_start;    // 6 * R * R + 1 offsets into `entries`; cell ((face * R + v) * R + u)
	const int2* entries;      // (bits of the distance from the light to the triangle's padded box, triangle index); per cell ascending by distance
	int R;                    // cells per face edge; 0: no grid for this light (its shadow rays walk the tree)
	float half_R;
	float slack;              // added to a ray's length before the distance cut-off of its list
	int n_over;               // triangles tested for every ray of this light (near the light, or wide as seen from it)
	int over[RT_LGRID_OVER_MAX];
	// The triangles kept out of the tree (DevScene::hoist), by direction: per coarse cell (RT_LGRID_HC x RT_LGRID_HC per cube face) a
	// bit mask of the hoisted triangles whose padded box may be seen in it, their distance from the light and their indices
	const unsigned int* hmask;   // 6 * RT_LGRID_HC * RT_LGRID_HC words
	float hmind[RT_HOIST_MAX];
	int hidx[RT_HOIST_MAX];
};
#define RT_LGRID_HC 16

struct DevScene {
	const float4* planes;   // PL_COUNT * stride float4
	const float4* lights;   // n_lights x (pos.xyz, intensity)   Source.cpp:69-73
	int n_tris;
	int stride;
	int n_lights;
	int n_clusters;
	// LBVH (null for brute force): see rt_lbvh.hip
	const float4* bvh_nodes;   // binary Karras tree, 4 float4 per node (kept for inspection)
	const uint4* bvh4;         // 4-wide quantised collapse of it, 64 B per node: what traversal walks
	const float4* tri_aos;     // RT_AOS_F4 float4 per triangle (N P1 M0 M1 M2 | P2 P3 E0 E1 E2): what a leaf test needs first, then what only the
	                           // reference's own edge arithmetic needs (rt_kernels.hip edge_filter)
	int bvh_root;
	int skip_count;            // inner nodes per octant of the packet records (0: none)
	// Wave-uniform ("packet") traversal of the same binary tree (rt_kernels.hip cast_ray_packet):
	// per octant and inner node one 64-byte record with BOTH children, the one nearer along the
	// octant first: (nearA.xy, farA.xy, nearA.z, farA.z, refA, refB | nearB.xy, farB.xy, nearB.z, farB.z, -, -): the
	// pairs a packed multiply-add takes (slab_hit_pk).
	const float4* skiplist;
	// Triangles whose box spans a large part of the scene (walls, a ground plane) are kept OUT of
	// the tree, where each would inflate every box on its path to the root, and are tested first
	// for every ray from this list (rt_lbvh.hip lbvh_build): hoist_count records of RT_AOS_F4 float4 as in
	// tri_aos, ascending by triangle index, then for each of the 8 octants hoist_count padded boxes
	// of 8 floats (near.xyz, far.xyz, -, triangle index as int bits) as in the octant records above.
	const float4* hoist;
	int hoist_count;
	// Octant-ordered copies of the group and cluster boxes (brute force): for each of the 8 sign
	// combinations of a ray direction, entry e is 8 floats (near.xyz, far.xyz, a, b) where near/far
	// are the box faces a ray of that octant meets first/last per axis; clusters carry a = first
	// group, b = group count as int bits.  Offsets in floats from `planes`; pitches in entries.
	long long oct_clusters, oct_groups;
	int oct_cluster_pitch, oct_group_pitch;
	// scene bounding box, padded (host: rt_api.cpp scene_box); +-inf when the scene has none
	float bb_lo[3], bb_hi[3];
	float far_guard;        // |origin| beyond this: the box tests are switched off for the ray
	int box_within_guard;   // 1: the whole scene box lies within far_guard — a ray that starts inside the box needs no check (rt_kernels.hip dev_scene)
	float packet_scale;     // scene diagonal / mean triangle box diagonal (launcher: is an 8x8 tile of rays coherent?)
	int traversal;          // rt_scene_set_traversal: RT_TRAVERSAL_AUTO | _PER_LANE | _PACKET (host side only)
	int kernel_kind;        // rt_scene_set_kernel (host side only)
	int per_lane_ok;        // 0: the 4-wide tree is too deep for the per-lane stack (RT_STACK_MAX); packets only
	const LightGrid* lgrid; // records for the first n_lgrid lights (LBVH commits; null / 0: none)
	int n_lgrid;
	int debug;              // RT_EXPERIMENT builds: switches for timing experiments (0 otherwise)
};

// Per-launch frame description.  Rows are addressed as LOCAL rows of this shard:
// local row ly -> tile ly / tile_rows -> global tile tile_first + that * tile_stride ->
// global row y = tile * tile_rows + ly % tile_rows.  A whole frame is tile_first 0, stride 1.
// One launch may render several frames of the same scene (a camera path: consecutive iterations
// of the frame loop, Source.cpp:1126, with known poses): workgroups [f * blocks_per_frame,
// (f+1) * blocks_per_frame) render frame f into out + f * frame_stride.
#define RT_MAX_FRAMES 64      /* 3 KB of poses in the kernel arguments (4 KB limit; DevScene + the rest of FrameParams are 0.35 KB) */
struct FrameParams {
	// per frame: camera.position (Source.cpp:545), then what Source.cpp:537-545 needs of trans
	// (Source.cpp:1145-1147, column-major): columns 0 and 1, and c = m2 * 0.5f + m3 * 1.0f, the part of
	// (trans * vec4(u, v, 0.5, 1)).xyz that no pixel changes (set_pose computes it on the host with the
	// same two roundings the reference makes per pixel)
	float pose[RT_MAX_FRAMES][12];
	int n_frames;
	unsigned int blocks_per_frame;  // set by the launcher
	// XCD-aware order of the workgroups of a multi-frame LBVH launch (0: off): the frame is cut into
	// 8 bands of band_blocks workgroups; workgroup b runs on XCD b % 8 (round-robin dispatch) and
	// renders, as the j-th workgroup of that XCD, band (xcd + frame) % 8 of frame j / band_blocks —
	// each XCD's L2 then sees the part of the scene behind one band at a time, and the rotation with
	// the frame number evens out bands of different cost over the frames of the launch.
	unsigned int band_blocks;
	long long frame_stride;         // 4-byte elements between consecutive frames in `out`
	unsigned long long origin_inside;   // bit f: frame f's camera lies inside the scene box (set by the launcher; rt_kernels.hip exit_bound_is_idle)
	int width, height;      // Source.cpp:887-888
	int max_bounces;        // Source.cpp:567
	int tile_rows, tile_first, tile_stride;
	int local_rows;         // rows this launch renders
	int tiles_per_wave;     // 8x8 pixel tiles per wavefront, 0 = let the launcher choose
	// u = (x - w/2) / w and v = (y - h/2) / h (Source.cpp:534-535) without the division sequence: with
	// rcp_w = RN(1/w), q = n * rcp_w, r = fma(-q, w, n), u = fma(r, rcp_w, q) is the correctly rounded
	// quotient for almost every numerator; the launcher checks EVERY column and row of the frame on the
	// host against the IEEE division and sets uv_fast only when all of them agree bit for bit.
	float rcp_w, rcp_h;
	int uv_fast;
	float* out;             // local_rows * width * 3 floats, or (rgba8) local_rows * width packed 8-bit pixels
	int rgba8;              // RT_OUT_RGBA8: store_pixel (rt_kernels.hip)
	int shade_glsl;         // RT_SHADE_GLSL: the compute shader's colour arithmetic (render_kernel only; the launcher picks it)
	unsigned long long* counters;   // rt_stats as 10 x u64, or null
	// Primary visibility of the launch's frames, resolved by the rasterising pass before the render kernel (rt_vis.inl), or null:
	// vis[frame * vis_stride + pixel] = (bits of t : triangle index) of the closest hit among the tree's triangles, ~0 for none
	const unsigned long long* vis;
	long long vis_stride;
	const int* vis_count;           // per frame: triangles the pass could not project (tested for every pixel); > RT_VIS_LIST_MAX: the frame walks the tree
	const int* vis_list;            // RT_VIS_LIST_MAX per frame
	float vis_pad;                  // the commit's pad (rt_lbvh.hip tri_box_padded)
	// Which hoisted triangles a primary ray can meet, per frame and coarse cell of the view (rt_vis.inl hoist_mask_kernel), or null:
	// hoist_mask[(frame * RT_HOIST_CELLS + cv) * RT_HOIST_CELLS + cu], cu = x * hoist_cell_x, cv = (global row) * hoist_cell_y
	const unsigned int* hoist_mask;
	float hoist_cell_x, hoist_cell_y;
};
#define RT_HOIST_CELLS 16
#define RT_VIS_LIST_MAX 32

// ---- host scene ------------------------------------------------------------------------------

struct DeviceState {
	int device = -1;
	bool valid = false;
	int accel = RT_ACCEL_BRUTE;
	void* planes = nullptr;
	void* lights = nullptr;
	void* bvh_nodes = nullptr;
	void* bvh4 = nullptr;
	void* tri_aos = nullptr;
	void* skiplist = nullptr;
	int skip_count = 0;
	void* hoist = nullptr;
	int hoist_count = 0;
	int bvh_node_count = 0;      // inner nodes of the binary tree (triangles in the tree - 1)
	int bvh_depth = 0;           // levels of inner nodes on the longest root-to-leaf path of the binary tree
	int bvh4_depth = 0;          // the same for the 4-wide collapse the per-lane traversal walks
	bool per_lane_ok = true;     // 3 * bvh4_depth fits the per-lane stack (rt_lbvh_trav.inl RT_STACK_MAX)
	std::vector<int> hoisted;    // triangle indices kept out of the tree, ascending
	float packet_scale = 0.f;
	void* counters = nullptr;
	int bvh_root = 0;
	int stride = 0;
	float build_ms = 0.f;
	void* tri_skip = nullptr;              // one byte per triangle: 1 = kept out of the tree (hoisted); LBVH commits
	// buffers of the primary-visibility pass (rt_vis.inl), one set per stream that launches on this device
	struct VisBuf { void* stream = nullptr; void* vis = nullptr; size_t vis_bytes = 0; void* meta = nullptr; void* hmask = nullptr; };
	std::vector<VisBuf> vis_bufs;
	// per-light shadow grids (rt_lgrid.hip)
	void* lgrid = nullptr;                 // n_lgrid LightGrid records
	int n_lgrid = 0;
	std::vector<void*> lgrid_buffers;      // cell offsets and entry lists the records point to
	std::vector<long long> lgrid_info;     // per light: R, entries, "always" triangles
	float lgrid_ms = 0.f;
	int built_by = -1;           // RT_BUILD_* the tree of this commit was really built by (PLOC falls back to FAST when it cannot finish)
	float bb_lo[3] = { 0, 0, 0 }, bb_hi[3] = { 0, 0, 0 };
	float far_guard = 0.f;
	float pad = 0.f;
	int n_clusters = 0;
	long long oct_clusters = 0, oct_groups = 0;
	int oct_cluster_pitch = 0, oct_group_pitch = 0;
	// scratch framebuffer for rt_render (host output) and the compact shard of rt_render_multi
	void* frame = nullptr;
	size_t frame_bytes = 0;
	// rt_render_multi: the whole frame on devices[0], where the shards are gathered before the one D2H copy
	void* gather = nullptr;
	size_t gather_bytes = 0;
	// rt_render_async: two device frames, a render and a copy stream, per slot "rendered" / "copied" events
	void* aframe[2] = { nullptr, nullptr };
	size_t aframe_bytes[2] = { 0, 0 };
	std::vector<hipStream_t> multi_streams;   // rt_render_multi: one per listing of this device, kept between calls (streams also key the per-stream scratch)
	hipStream_t render_stream = nullptr, render_stream2 = nullptr, copy_stream = nullptr;   // (slot 0 renders on the first, slot 1 on the second: the ramp and tail of consecutive frames overlap)
	hipEvent_t rendered[2] = { nullptr, nullptr }, copied[2] = { nullptr, nullptr };
	long long ticket_of_slot[2] = { -1, -1 };
};

} // namespace rt

struct rt_scene {
	std::vector<float> tris;      // RT_TRI_FLOATS per triangle, the reference's record
	std::vector<float> lights;    // 4 per light
	unsigned long long generation = 1;
	std::map<int, rt::DeviceState> dev;   // by HIP device ordinal
	int first_device = -1;
	int traversal = RT_TRAVERSAL_AUTO;    // rt_scene_set_traversal
	int tiles_per_wave = 0;               // rt_scene_set_tiles_per_wave (0: the launcher's rule)
	long long next_ticket = 0;            // rt_render_async
	int kernel_kind = RT_KERNEL_AUTO;     // rt_scene_set_kernel
	int build = RT_BUILD_PLOC;            // rt_scene_set_build (takes effect at the next LBVH commit)
	int shadow_grid = 1;                  // rt_scene_set_shadow_grid (takes effect at the next LBVH commit)
	int primary_raster = RT_PRIMARY_AUTO; // rt_scene_set_primary_raster
};

namespace rt {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
#define RT_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return rt::hip_fail(e_, #call); } while (0)

// rt_sah.cpp
void sah_topology(const std::vector<float>& tris, const std::vector<int>& hoisted, int nt, std::vector<unsigned long long>& keys,
                  std::vector<int>& child, std::vector<int>& parent_int, std::vector<int>& parent_leaf);

// rt_scene.cpp
void triangle_record(vec3 p1, vec3 p2, vec3 p3, vec3 color, float shin, float* out28);
void quad_half_record(const float T[3], const float Rdeg[3], const float S[3], const float color[3], float shin, float* out28);
void camera_trans(const float rot_deg[3], float out16[16]);
struct PlaneLayout { int n_clusters; long long oct_clusters, oct_groups; int oct_cluster_pitch, oct_group_pitch; };
PlaneLayout flatten_planes(const std::vector<float>& tris, int stride, float pad, bool with_octant_records, std::vector<float>& out);
float scene_box(const std::vector<float>& tris, float lo[3], float hi[3]);   // returns the pad it applied

// rt_kernels.hip
void set_pose(FrameParams& fp, int k, const float cam[3], const float trans16[16]);
DevScene dev_scene(const rt_scene* s, const DeviceState& d);
bool use_packets(const DevScene& sc, const float trans16[16], int width, int height);   // the launcher's choice for an LBVH scene
int launch_render(const DevScene& sc, const FrameParams& fp, int accel, int render_flags, bool count, hipStream_t stream);
int launch_cast_rays(const DevScene& sc, int accel, int n, const float* d_rays7, float* d_out6, hipStream_t stream);
int launch_scatter_shards(const void* shards, size_t pitch_bytes, int n, size_t row_bytes, int height, int tile_rows, void* frame, hipStream_t stream);   // rt_shards.hip
int launch_hoist_masks(const DevScene& sc, const FrameParams& fp, hipStream_t stream);   // rt_vis.inl
int launch_vis_raster(const DevScene& sc, const FrameParams& fp, const unsigned char* d_skip, bool count, hipStream_t stream);   // rt_vis.inl

// rt_lbvh.hip
int lbvh_build(rt_scene* s, DeviceState& d, hipStream_t stream);
void lbvh_free(DeviceState& d);

// rt_lgrid.hip
int lgrid_build(rt_scene* s, DeviceState& d, hipStream_t stream);
void lgrid_free(DeviceState& d);

} // namespace rt
