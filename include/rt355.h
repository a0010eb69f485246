/* rt355.h — C ABI of librt355.so: the MI355X render loop behind the reference's own entry points.
 *
 * The reference (Expasito/RayTracing, RayTracing/src/Source.cpp, "Src" below) has no plugin or
 * FFI layer: its API is free functions and globals in one translation unit.  The seam this library
 * fills is the CPU branch of the frame loop (Src:1244-1272): inputs are the `triangles` / `lights`
 * globals, `camera.position`, the per-frame `trans` matrix, `width`, `height`; the output is the
 * `float* data` buffer handed to glTexImage2D(GL_RGB, GL_FLOAT) at Src:1270.  Every entry point
 * below names the reference line it replaces.  All functions return RT_OK (0) or a negative
 * RT_ERR_* code and never exit, abort or throw across the boundary (the reference calls exit(1),
 * Src:634-651); rt_last_error() returns a thread-local description of the last failure.
 *
 * There is no CPU fallback: without a HIP device rt_scene_commit / rt_render* fail with
 * RT_ERR_DEVICE.  Scene construction (rt_scene_add_*) is host arithmetic by design — it is the
 * reference's start-up code, and its results are what gets uploaded.
 *
 * Pixel layout of every framebuffer: out[(y*width + x)*3 + c], float32, c in {R,G,B}, values in
 * [0,1], row y = 0 is the BOTTOM row of the displayed image (Src:550, 612-614, 1270).
 */
#ifndef RT355_H
#define RT355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK            0
#define RT_ERR_INVALID  -1   /* NULL / out-of-range argument                         */
#define RT_ERR_STATE    -2   /* scene not committed, or modified since commit        */
#define RT_ERR_DEVICE   -3   /* HIP runtime error, or no HIP device                   */
#define RT_ERR_NOMEM    -4
#define RT_ERR_IO       -5   /* file not found / unreadable / malformed              */
#define RT_ERR_LIMIT    -6   /* size beyond what the chosen path supports            */

/* rt_scene_commit flags: acceleration structure */
#define RT_ACCEL_AUTO    0   /* brute force up to RT_BRUTE_AUTO_MAX triangles, LBVH above */
#define RT_ACCEL_BRUTE   1   /* the reference's O(N) loop over every triangle (Src:301)   */
#define RT_ACCEL_LBVH    2   /* bounding-volume tree over the triangles (rt_scene_set_build)  */
#define RT_BRUTE_AUTO_MAX 40    /* measured at 800x800, us/frame brute | tree (SAH build): room (26 triangles) 17.8 | 23.5, 32: 12.2 | 14.5, 52: 14.1 | 13.0, 92: 17.8 | 13.6, 172: 20.4 | 13.8, 972: 36 | 18 */

/* rt_render* flags: the only mode is unfused IEEE binary32 in the reference's operation order —
 * bit-identical frames.  (A contracted, tolerance-bound build was measured 2-5 % faster and moved
 * the reference's diagonal seams by a pixel; it was dropped)  Any other value is
 * RT_ERR_INVALID. */
#define RT_RENDER_STRICT 0
/* Output format bit, OR-ed into render_flags of rt_render, rt_render_device, rt_render_frames_device,
 * rt_render_tiles_device and rt_render_async (RT_ERR_INVALID elsewhere): the output buffer takes ONE packed
 * 8-bit pixel per 4 bytes (R, G, B, 255 in memory order) instead of three floats — the pixel the
 * reference's viewer ends up displaying, i.e. the GL unorm conversion glTexImage2D applies to `data`
 * (Src:1270): q = (int)(c * 255.0f + 0.5f), NaN and negative -> 0.  The arithmetic up to the stored
 * colour is the same; a frame is a third of the bytes to copy to the host or to gather from other GPUs
 * (8K: 133 MB instead of 398 MB).  The `float*` output arguments then point at width*height 4-byte
 * pixels, and frame_stride_floats counts 4-byte pixels. */
#define RT_OUT_RGBA8 0x100
/* Shading bit, accepted wherever RT_OUT_RGBA8 is: the colour arithmetic of the reference's OTHER renderer, the GL
 * compute shader its `GPU` switch selects (Compute.shader:140-219, 224-345, host side Src:1168-1243), for cross-checking
 * frames against a machine that runs that shader.  Same primary rays and the same castRay; what differs:
 * no reflections — every hit is lit as matte, shininess is ignored (Compute.shader:257-330); per unshadowed light the
 * SCALAR (1.0 / (dist*dist) * intensity) / 255.0 is summed in binary32 (Compute.shader:214; Src:488 divides in binary64),
 * then colour = (sum * triangle colour) / 255.0 (Compute.shader:330); nothing is clamped to 1 (the shader stores into an
 * rgba32f image).  Evaluated here in IEEE binary32 in the shader's written order with glm's normalize; GLSL leaves the
 * precision of /, sqrt and normalize to the driver, so a real GL machine may differ in the last bits: this mode is
 * "parity unpinned" (no GL here to run the shader), checked against the oracle's restatement of the shader only.
 * max_bounces is ignored.  Runs the state-machine kernel: a checking mode, not a fast path. */
#define RT_SHADE_GLSL 0x200

/* One triangle record as the reference stores it (Src:31-43), 28 floats:
 * p[3] n[3] p1[3] p2[3] p3[3] edge0[3] edge1[3] edge2[3] color[3] shininess */
#define RT_TRI_FLOATS 28

typedef struct rt_scene rt_scene;

/* Work counters in the units SURVEY.md §8(d) prices (filled by rt_render_stats). */
typedef struct rt_stats {
	uint64_t rays_primary, rays_bounce, rays_shadow;
	uint64_t n_plane;    /* triangles visited by castRay (Src:383), 14 FLOP each           */
	uint64_t n_point;    /* of those, not rejected by t < 0 (Src:385-389), 6 FLOP each     */
	uint64_t n_edge;     /* edge terms evaluated with && short-circuit (Src:392-394), 17   */
	uint64_t n_box;      /* LBVH only: ray/AABB slab tests, 22 FLOP each                   */
	uint64_t px_hit, px_mirror, px_lit;
} rt_stats;

/* ---- scene (replaces the globals `triangles`, `lights`, Src:64-73) ------------------------ */

rt_scene* rt_scene_create(void);
void      rt_scene_destroy(rt_scene* s);
int       rt_scene_clear(rt_scene* s);

/* Triangle(p1,p2,p3,color,shininess) + triangles.push_back — Src:45-59.  color is 0..255. */
int rt_scene_add_triangle(rt_scene* s, const float p1[3], const float p2[3], const float p3[3],
                          const float color[3], float shininess);

/* n triangles at once; arrays are n x 3 (shininess: n).  Same arithmetic as the call above. */
int rt_scene_add_triangles(rt_scene* s, int n, const float* p1, const float* p2, const float* p3,
                           const float* color, const float* shininess);

/* addTriangle(translate, rotate_degrees, scale, color, shininess) — Src:410-439. */
int rt_scene_add_quad_half(rt_scene* s, const float translate[3], const float rotate_deg[3],
                           const float scale[3], const float color[3], float shininess);

/* lights.push_back({position, intensity}) — Src:69-73, 880. */
int rt_scene_add_light(rt_scene* s, const float position[3], float intensity);

/* loadModel(path) — Src:631-657: "(tx, ty, tz), (rx, ry, rz), (sx, sy, sz), (r, g, b); " records,
 * each becoming addTriangle(..., shininess 0).  Returns the number of triangles added (>= 0). */
int rt_scene_load_rto(rt_scene* s, const char* path);

int rt_scene_triangle_count(const rt_scene* s);
int rt_scene_light_count(const rt_scene* s);

/* Copy out up to `capacity` triangle records (RT_TRI_FLOATS floats each) in index order; returns
 * the number copied.  This is the reference's `triangles` vector, for inspection and tests. */
int rt_scene_get_triangles(const rt_scene* s, float* out28, int capacity);

/* Flatten to the SoA device buffer and upload to HIP device `device` (replaces the per-frame
 * visibleTriangles rebuild, Src:1153-1163); with RT_ACCEL_LBVH (or AUTO above the threshold)
 * also builds the LBVH on the GPU.  Must be called again after any rt_scene_add_* / clear.
 * A scene may be committed to several devices (one call per device). */
int rt_scene_commit(rt_scene* s, int device, int accel_flags);

/* Milliseconds the last commit on `device` spent building the LBVH on the GPU (0 for brute). */
int rt_scene_build_ms(const rt_scene* s, int device, float* out_ms);

/* Copy out the LBVH built by the last commit on `device` (tests, tools): up to `capacity` nodes of
 * 16 floats each — {left box lo xyz, hi xyz, right box lo xyz, hi xyz, left ref, right ref, 0, 0},
 * refs as int bits: >= 0 internal node index, < 0 leaf holding triangle ~ref; node 0 is the root.
 * Returns the number of nodes (triangles in the tree - 1), or 0 for a brute-force commit. */
int rt_scene_get_bvh(const rt_scene* s, int device, float* out_nodes16, int capacity);

/* The triangles the last LBVH commit on `device` kept OUT of the tree: those whose box has an
 * eighth or more of the scene box's surface area (walls, a ground plane; at most 32).  Every ray
 * tests them directly before it walks the tree, as castRay's loop would (Src:301-401); inside the
 * tree each would widen every box on its path to the root.  Copies up to `capacity` ascending
 * triangle indices; returns how many there are (0 for a brute-force commit). */
int rt_scene_get_hoisted(const rt_scene* s, int device, int* out_indices, int capacity);

/* The acceleration structure the last commit on `device` chose: RT_ACCEL_BRUTE or RT_ACCEL_LBVH. */
int rt_scene_accel(const rt_scene* s, int device);

/* Depth of the LBVH of the last commit on `device`: inner-node levels on the longest root-to-leaf
 * path of the binary Karras tree (at most 63: the keys have 63 bits) and of its 4-wide collapse.
 * The per-lane traversal keeps up to 3 entries per 4-wide level on its stack of RT_STACK_LIMIT
 * entries; a tree with 3 * wide_depth > RT_STACK_LIMIT (nested geometry graded towards a point) is
 * traversed by packets only, whose stack holds one entry per binary level.  Both 0 for brute force. */
#define RT_STACK_LIMIT 64
int rt_scene_bvh_depth(const rt_scene* s, int device, int* binary_depth, int* wide_depth);

/* ---- camera (Src:1145-1147) ---------------------------------------------------------------- */

/* trans = rotate(rotate(I, radians(rot.x), Y), radians(-rot.y), X), column-major 4x4. */
int rt_camera_trans(const float rot_deg[3], float out_trans16[16]);

/* ---- camera controller (Src:95-212, 237-241, 1294-1295, 1307-1308) -------------------------
 * The reference's `Camera` as plain data and its frame-loop tail as functions, so that a recorded
 * or scripted camera path (poses for rt_render_frames_device) moves exactly as the reference's
 * interactive camera would.  Host arithmetic (binary32, the reference's order; sinf/cosf from libm). */
typedef struct rt_camera {
	float position[3];     /* Src:98  */
	float rotations[3];    /* Src:100, degrees */
	float direction[3];    /* Src:102, from rotations (Src:126-130) */
	float right[3], up[3]; /* Src:103-104, set by rt_camera_translate (zero before) */
	float move_speed;      /* Src:109: moveBaseSpeed = 1/128 until the first frame time */
	float rot_speed;       /* Src:112: rotBaseSpeed = moveBaseSpeed*3.14159265*2*2 */
} rt_camera;

/* key flags of Src:237-241 as one mask */
#define RT_KEY_LEFT        0x001u  /* A      Src:246 */
#define RT_KEY_RIGHT       0x002u  /* D      Src:247 */
#define RT_KEY_UP          0x004u  /* SPACE  Src:248 */
#define RT_KEY_DOWN        0x008u  /* LSHIFT Src:249 */
#define RT_KEY_FORWARD     0x010u  /* W      Src:250 */
#define RT_KEY_BACKWARD    0x020u  /* S      Src:251 */
#define RT_KEY_LOOK_LEFT   0x040u  /* LEFT   Src:253 */
#define RT_KEY_LOOK_RIGHT  0x080u  /* RIGHT  Src:254 */
#define RT_KEY_LOOK_UP     0x100u  /* UP     Src:255 */
#define RT_KEY_LOOK_DOWN   0x200u  /* DOWN   Src:256 */

/* Camera(pos, rot) — Src:132-142 */
int rt_camera_init(rt_camera* cam, const float pos[3], const float rot_deg[3]);
/* Camera::rotate(l, r, u, d) — Src:181-209 (pitch clamped to +-89.99) */
int rt_camera_rotate(rt_camera* cam, int l, int r, int u, int d);
/* Camera::translate(l, r, u, d, f, b) — Src:144-179 */
int rt_camera_translate(rt_camera* cam, int l, int r, int u, int d, int f, int b);
/* moveSpeed = moveBaseSpeed*milis; rotSpeed = rotBaseSpeed*milis — Src:1307-1308 */
int rt_camera_frame_time(rt_camera* cam, float milis);
/* The tail of one frame-loop iteration: rotate, translate (Src:1294-1295), frame time (Src:1307-1308). */
int rt_camera_step(rt_camera* cam, unsigned int key_mask, float milis);
/* A scripted path: for frame i the pose the frame is drawn with (position -> out_pos3[i],
 * trans of Src:1145-1147 -> out_trans16[i]) is recorded, then rt_camera_step(key_masks[i],
 * frame_ms[i]) moves on.  The arrays are what rt_render_frames_device takes. */
int rt_camera_path(rt_camera* cam, int n_frames, const unsigned int* key_masks, const float* frame_ms,
                   float* out_pos3, float* out_trans16);

/* ---- render (replaces the getPixelData loop, Src:1248-1264 -> 530-628) --------------------- */

/* Whole frame into HOST memory, synchronous: launch on the scene's device, copy back.
 * `out_rgb` has width*height*3 floats.  max_bounces is the reference's maxBounces (Src:567: 1). */
int rt_render(rt_scene* s, const float cam_pos[3], const float trans16[16], int width, int height,
              int max_bounces, int render_flags, float* out_rgb);

/* Whole frame into DEVICE memory, asynchronous on `hip_stream` (a hipStream_t, NULL = default
 * stream) of device `device`.  The framebuffer stays resident in HBM for a viewer / interop. */
int rt_render_device(rt_scene* s, int device, const float cam_pos[3], const float trans16[16],
                     int width, int height, int max_bounces, int render_flags,
                     float* d_out_rgb, void* hip_stream);

/* Several frames of one committed scene — consecutive iterations of the reference's frame loop
 * (Src:1126) whose camera poses are known in advance, e.g. a recorded camera path — rendered by as
 * few launches as possible (up to RT_MAX_FRAMES_PER_LAUNCH frames each), asynchronously on
 * `hip_stream`.  cam_pos3 is n_frames x 3 and trans16 n_frames x 16 (host memory, read before the
 * call returns); frame f lands at d_out_rgb + f * frame_stride_floats, each in the usual layout.
 * A single 800x800 frame is ~10 wavefronts per SIMD, too few to fill an MI355X; batching frames
 * removes the per-launch ramp and tail.  Pixels are identical to rt_render's. */
#define RT_MAX_FRAMES_PER_LAUNCH 64
int rt_render_frames_device(rt_scene* s, int device, int n_frames, const float* cam_pos3, const float* trans16,
                            int width, int height, int max_bounces, int render_flags,
                            float* d_out_rgb, size_t frame_stride_floats, void* hip_stream);

/* Interleaved row tiles for multi-GPU sharding: renders tiles t = tile_first,
 * tile_first + tile_stride, ... of `tile_rows` rows each (tile t covers rows [t*tile_rows,
 * (t+1)*tile_rows)) into a COMPACT device buffer that holds only this shard's rows, in
 * increasing y.  rt_shard_rows() tells how many rows that is. */
int rt_render_tiles_device(rt_scene* s, int device, const float cam_pos[3], const float trans16[16],
                           int width, int height, int max_bounces, int render_flags,
                           int tile_rows, int tile_first, int tile_stride,
                           float* d_out_compact, void* hip_stream);
int rt_shard_rows(int height, int tile_rows, int tile_first, int tile_stride);
/* The same shard into HOST memory, synchronous (launch on the scene's device, copy back): for exchanges whose transport
 * takes host buffers (MPI or gloo without GPU-aware transport).  out_compact has rt_shard_rows() x width pixels. */
int rt_render_tiles(rt_scene* s, const float cam_pos[3], const float trans16[16], int width, int height, int max_bounces,
                    int render_flags, int tile_rows, int tile_first, int tile_stride, float* out_compact);

/* The exchange step's other half: after the gather, the rank that displays the frame holds every rank's
 * compact shard — shard r (tiles r, r + n_shards, ...: rt_render_tiles_device with tile_first = r, tile_stride =
 * n_shards) at d_shards + r * pitch_bytes — and needs the reference's framebuffer, row y at y * width pixels
 * (Src:550 `data[(y*w+x)*3+c]`, uploaded by Src:1270).  One kernel on `hip_stream` of `device` moves every row to its
 * place (16 bytes per lane where sizes and addresses allow).  render_flags says what a pixel is: 3 floats, or 4 bytes
 * with RT_OUT_RGBA8.  pitch_bytes: a multiple of 4, at least the size of shard 0 (the largest).  Asynchronous.
 * rt_scatter_shards_host is the same on host memory (one memcpy per tile). */
int rt_scatter_shards(int device, const void* d_shards, size_t pitch_bytes, int n_shards, int width, int height,
                      int tile_rows, int render_flags, void* d_frame, void* hip_stream);
int rt_scatter_shards_host(const void* shards, size_t pitch_bytes, int n_shards, int width, int height,
                           int tile_rows, int render_flags, void* frame);

/* The hand-off to the viewer (Src:1270 uploads `data` every frame) without stalling the render:
 * rt_render_async launches the frame on the scene's own render stream into one of two device frames
 * and queues its copy to `out_rgb` on a second stream, then returns; while that copy runs the next
 * rt_render_async already renders.  Returns a ticket >= 0; rt_render_wait(s, ticket) blocks until
 * that frame (and every earlier one) is in `out_rgb`.  At most two frames are in flight: a third
 * call first waits for the oldest.  `out_rgb` should come from rt_frame_alloc (pinned host memory:
 * the copy then runs at PCIe speed and truly asynchronously; it is what replaces the malloc of
 * Src:1035); pageable memory works but the copy is staged by the runtime.  Frames are rt_render's. */
int rt_render_async(rt_scene* s, const float cam_pos[3], const float trans16[16], int width, int height,
                    int max_bounces, int render_flags, float* out_rgb);
int rt_render_wait(rt_scene* s, int ticket);
/* Pinned host framebuffer of n_floats floats (replaces `data = malloc(...)`, Src:1035) / its release. */
float* rt_frame_alloc(size_t n_floats);
void   rt_frame_free(float* frame);

/* Same frame as rt_render, additionally counting work (slower instrumented kernel). */
int rt_render_stats(rt_scene* s, const float cam_pos[3], const float trans16[16], int width,
                    int height, int max_bounces, float* out_rgb /* may be NULL */, rt_stats* out);

/* One process, several GPUs: commits the scene to devices[0..n) if needed, renders interleaved
 * row tiles on each (tile t -> devices[t % n]), gathers the compact shards on devices[0] with ONE
 * peer-to-peer copy per device (xGMI), de-interleaves them there (rt_scatter_shards' kernel) and copies
 * the frame to host memory once.  render_flags may carry RT_OUT_RGBA8 (out_rgb then receives width*height
 * 4-byte pixels: a third of the bytes on every link) and RT_SHADE_GLSL.  A device may be listed more than
 * once (it renders those shards one after the other): the N-GPU tile assignment rehearsed on fewer GPUs. */
int rt_render_multi(rt_scene* s, const int* devices, int n_devices, const float cam_pos[3],
                    const float trans16[16], int width, int height, int max_bounces,
                    int render_flags, int tile_rows, float* out_rgb);

/* castRay for a batch of rays (Src:277-405), for tests and tools: rays7 = n x {origin xyz,
 * dir xyz, skip index (<0 = none)}; out6 = n x {didHit, index|-1, distance, point xyz}.
 * Host pointers; synchronous. */
int rt_cast_rays(rt_scene* s, int n_rays, const float* rays7, float* out6);

/* rt_cast_rays for callers that cannot allocate device memory (harnesses that forbid hipMalloc, serving
 * loops that must stay off the allocator, a process in which another thread is capturing a HIP graph — an
 * allocation then would invalidate the capture).  It has no counterpart in the reference.  Same
 * arguments and results as rt_cast_rays, but the device copies of `rays7` and `out6` live in a
 * buffer that is part of the library image (one per GPU, RT_NOALLOC_MAX_RAYS rays = 832 KiB, held
 * until the process ends) instead of being hipMalloc'ed per call: the call allocates and frees
 * nothing.  n_rays > RT_NOALLOC_MAX_RAYS is RT_ERR_LIMIT.  Its two copies are blocking ones on the null
 * stream, so the call itself is synchronous and cannot be recorded into a graph.  All calls on one GPU
 * share that GPU's buffer: the library lets them in one at a time (a mutex per GPU), so calls from several
 * host threads are safe and simply take turns; nothing in the buffer needs to start at zero — a
 * call overwrites the first n_rays slots of both halves and reads back only those. */
#define RT_NOALLOC_MAX_RAYS 16384
int rt_cast_rays_noalloc(rt_scene* s, int n_rays, const float* rays7, float* out6);

/* ---- misc ---------------------------------------------------------------------------------- */

int         rt_device_count(void);
/* What the device says it is: compute units and engine clock (kHz), for the roofline lines of bench.py (SURVEY 8d asks for
 * them next to the peak that is used).  No counterpart in the reference.  Either pointer may be NULL. */
int         rt_device_props(int device, int* compute_units, int* clock_khz);

/* Tuning knob of one scene: 8x8 pixel tiles each wavefront walks; 0 = the launcher's rule, at
 * most RT_TILES_PER_WAVE_MAX.  Pixels do not depend on it. */
#define RT_TILES_PER_WAVE_MAX 1024
int         rt_scene_set_tiles_per_wave(rt_scene* s, int tiles);
/* Tuning knob of one scene: which form of the render kernel its launches use.  Pixels do not depend
 * on it.  AUTO is the lockstep form (measured faster than, or equal to, the state machine on every scene
 * tried); the work-counting launches of rt_render_stats always run the state machine. */
#define RT_KERNEL_AUTO          0
#define RT_KERNEL_STATE_MACHINE 1   /* every lane its own stage, finished lanes refilled with new pixels (ballot + prefix ranks) */
#define RT_KERNEL_LOCKSTEP      2   /* the 64 pixels of a tile stay in step: primary ray, mirror loop, one shadow ray per light */
int         rt_scene_set_kernel(rt_scene* s, int kind);
/* Tuning knob of one scene: how the tree of an RT_ACCEL_LBVH commit is built (takes effect at the next
 * rt_scene_commit; committed trees of the scene become stale).  Pixels do not depend on it.
 * PLOC (the default): everything on the GPU — Morton codes and radix sort, then parallel locally-ordered
 * clustering over that order (each cluster merges with the neighbour, up to 8 places away, whose union with it has
 * the smallest surface area; Meister & Bittner 2018), boxes made on the way up.  A tree of binned-SAH quality (sum
 * of child-box areas over the root's, 102 412 triangles: 8.83; SAH 8.73; FAST 10.70) in about a millisecond.
 * FAST: Morton codes, radix sort and Karras' construction on the GPU, half a millisecond per 100k triangles; also
 * what a PLOC commit falls back to when clustering cannot finish (areas that overflow, a chain deeper than 63).
 * SAH: the topology by a binned surface-area heuristic on host threads, boxes and the derived records on the GPU;
 * about 15 ms per 100k triangles on 16 cores (round 2's default, kept for comparison). */
#define RT_BUILD_FAST 0
#define RT_BUILD_SAH  1
#define RT_BUILD_PLOC 2
int         rt_scene_set_build(rt_scene* s, int quality);
/* What really built the tree of the last LBVH commit on `device`: RT_BUILD_PLOC, _SAH or _FAST (a PLOC commit that
 * fell back reports FAST); RT_ERR_STATE for a brute-force commit.  No counterpart in the reference. */
int         rt_scene_build_kind(const rt_scene* s, int device);
/* Tuning knob of one scene: how the PRIMARY rays of an LBVH commit find their hit.  Pixels do not depend on it.  The primary rays of
 * a frame (Src:534-548) all start at the camera, so a triangle can only be hit by the rays of the few pixels it projects onto.
 * RASTER: before the render kernel a pass with one lane per (frame, triangle) projects each triangle of the tree onto the view plane,
 * runs the reference's own test (Src:383-396) for the pixels inside the projected box and keeps, per pixel, the smallest (t, index)
 * with a 64-bit atomic minimum; the render kernel reads its pixel's hit instead of walking the tree (bounce rays still walk it).
 * Applies to whole frames with a near-orthonormal view matrix (rt_camera_trans gives one) and a camera within the scene's
 * neighbourhood; other launches, and frames in which more than 32 triangles are too near the camera to project, walk the tree.
 * WALK: always the tree.  AUTO (the default) is WALK: measured on the 102 412-triangle scene the pass costs what the packet walk
 * it replaces costs (its atomic updates bound it), so it stays an option for scenes of very small triangles. */
#define RT_PRIMARY_AUTO   0
#define RT_PRIMARY_WALK   1
#define RT_PRIMARY_RASTER 2
int         rt_scene_set_primary_raster(rt_scene* s, int mode);
/* Tuning knob of one scene: direction grids for the shadow rays of LBVH commits (1, the default: on; 0: off; takes effect
 * at the next rt_scene_commit, committed trees of the scene become stale).  Pixels do not depend on it.  processLighting's
 * rays (Src:470-476) all end at a light, so as seen from the light each is ONE direction: per light (the first
 * RT_SHADOW_GRID_LIGHTS of the scene) the commit bins the triangles of the tree into a cube map of direction cells around
 * the light, nearest first, on the GPU; a shadow ray then tests the few triangles of its own cell that lie nearer the light
 * than its origin — with the reference's arithmetic — instead of walking the tree.  About half the rays of a frame are
 * shadow rays.  Lights beyond the limit, and lights crowded by geometry (more than 32 triangles within 1/250 of the scene
 * size of the light, or wider than half their distance from it), keep the tree walk. */
#define RT_SHADOW_GRID_LIGHTS 4
int         rt_scene_set_shadow_grid(rt_scene* s, int on);
/* The grid of light `light` from the last commit on `device` (tests, tools): cells per cube-face edge (0: this light has no
 * grid), list entries, triangles on the light's "always" list, and the GPU time of the commit's grid builds (all lights).
 * Any out pointer may be NULL.  No counterpart in the reference. */
int         rt_scene_shadow_grid(const rt_scene* s, int device, int light, int* cells_per_edge, long long* entries, int* always, float* build_ms);
/* Tuning knob of one scene: how its LBVH is traversed.  RT_TRAVERSAL_AUTO lets each launch choose
 * by the coherence of an 8x8 tile of rays against the scene's triangles; both give
 * the reference's bits.  A preference: a tree too deep for the per-lane stack (rt_scene_bvh_depth)
 * is walked by packets whatever is asked; rt_scene_traversal tells what a launch really gets. */
#define RT_TRAVERSAL_AUTO     0
#define RT_TRAVERSAL_PER_LANE 1   /* every lane walks the 4-wide tree with its own stack */
#define RT_TRAVERSAL_PACKET   2   /* one walk of the binary tree per wavefront and direction octant */
int         rt_scene_set_traversal(rt_scene* s, int mode);
/* Which traversal a frame of this size and pose gets on `device` in a launch long enough to run at steady state
 * (RT_TRAVERSAL_AUTO walks launches of fewer than 90 000 tiles in all — a lone 800x800 frame — per lane whatever
 * this says: their wavefronts all start cold): RT_TRAVERSAL_PER_LANE or RT_TRAVERSAL_PACKET for an LBVH commit,
 * 0 for a brute-force commit. */
int         rt_scene_traversal(const rt_scene* s, int device, const float trans16[16], int width, int height);
const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
