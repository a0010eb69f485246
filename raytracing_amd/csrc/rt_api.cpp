// rt_api.cpp — host side of the C ABI (include/rt355.h): device state, upload, launches, copies.
//
// This is where the reference's frame loop hands over (Source.cpp:1244-1272): instead of calling
// getPixelData W*H times it calls rt_render*, and gets back the same `float data[h*w*3]`.
// There is no CPU rendering path in this library: without a HIP device every entry point below
// returns RT_ERR_DEVICE.
#include "rt_internal.h"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cstdint>
#include <mutex>

#ifndef RT_BLIT_BLOCKS
#define RT_BLIT_BLOCKS 256
#endif

namespace rt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
	set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
	(void)hipGetLastError();   // clear the sticky last-error slot
	return e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE;
}

namespace {

// Makes `device` current for the scope and restores the caller's device afterwards.
struct DeviceGuard {
	int prev = -1;
	bool ok = false;
	int rc = RT_OK;
	explicit DeviceGuard(int device) {
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0) {
			(void)hipGetLastError();
			set_error("no HIP device available (hipGetDeviceCount: %s); this library has no CPU path", hipGetErrorString(e));
			rc = RT_ERR_DEVICE;
			return;
		}
		if (device < 0 || device >= n) {
			set_error("device %d out of range (have %d)", device, n);
			rc = RT_ERR_INVALID;
			return;
		}
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		e = hipSetDevice(device);
		if (e != hipSuccess) { rc = hip_fail(e, "hipSetDevice"); return; }
		ok = true;
	}
	~DeviceGuard() {
		if (ok && prev >= 0) (void)hipSetDevice(prev);
	}
};

void free_device_state(DeviceState& d) {
	if (d.device < 0) return;
	int prev = -1;
	(void)hipGetDevice(&prev);
	if (hipSetDevice(d.device) == hipSuccess) {
		// frames in flight (rt_render_async) finish before anything they use goes away
		if (d.render_stream) (void)hipStreamSynchronize(d.render_stream);
		if (d.render_stream2) (void)hipStreamSynchronize(d.render_stream2);
		if (d.copy_stream) (void)hipStreamSynchronize(d.copy_stream);
		for (hipStream_t ms : d.multi_streams) if (ms) { (void)hipStreamSynchronize(ms); (void)hipStreamDestroy(ms); }
		for (auto& b : d.vis_bufs) { if (b.vis) (void)hipFree(b.vis); if (b.meta) (void)hipFree(b.meta); if (b.hmask) (void)hipFree(b.hmask); }
		if (d.planes) (void)hipFree(d.planes);
		if (d.lights) (void)hipFree(d.lights);
		if (d.counters) (void)hipFree(d.counters);
		if (d.frame) (void)hipFree(d.frame);
		if (d.gather) (void)hipFree(d.gather);
		for (int k = 0; k < 2; k++) {
			if (d.aframe[k]) (void)hipFree(d.aframe[k]);
			if (d.rendered[k]) (void)hipEventDestroy(d.rendered[k]);
			if (d.copied[k]) (void)hipEventDestroy(d.copied[k]);
		}
		if (d.render_stream) (void)hipStreamDestroy(d.render_stream);
		if (d.render_stream2) (void)hipStreamDestroy(d.render_stream2);
		if (d.copy_stream) (void)hipStreamDestroy(d.copy_stream);
		lbvh_free(d);
		lgrid_free(d);
		if (d.tri_skip) (void)hipFree(d.tri_skip);
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	d = DeviceState();
}

int check_frame_args(const char* fn, const rt_scene* s, const float* cam, const float* trans, int w, int h, int depth) {
	if (!s || !cam || !trans) { set_error("%s: NULL argument", fn); return RT_ERR_INVALID; }
	if (w <= 0 || h <= 0) { set_error("%s: width and height must be positive (got %dx%d)", fn, w, h); return RT_ERR_INVALID; }
	if ((long long)w * h > 0x7fffffffLL / 3) { set_error("%s: %dx%d overflows the reference's int pixel index (Source.cpp:550)", fn, w, h); return RT_ERR_LIMIT; }
	if (depth < 0) { set_error("%s: max_bounces must be >= 0", fn); return RT_ERR_INVALID; }
	return RT_OK;
}

int check_flags(const char* fn, int render_flags, bool format_bits_ok = false) {
	if (format_bits_ok) render_flags &= ~(RT_OUT_RGBA8 | RT_SHADE_GLSL);
	if (render_flags != RT_RENDER_STRICT) { set_error("%s: unknown render flag 0x%x (RT_RENDER_STRICT, with RT_OUT_RGBA8 / RT_SHADE_GLSL where the header allows them)", fn, render_flags); return RT_ERR_INVALID; }
	return RT_OK;
}

DeviceState* committed(const char* fn, rt_scene* s, int device) {
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) {
		set_error("%s: scene is not committed on device %d (rt_scene_commit after the last rt_scene_add_*)", fn, device);
		return nullptr;
	}
	return &it->second;
}

// Source.cpp:534-535 for one dimension of n pixels: is the division-free form of FrameParams::rcp_w the
// IEEE quotient, bit for bit, for every numerator the frame has?
bool uv_division_free(int n) {
	if (n <= 0 || n >= (1 << 22)) return false;
	const float fn = (float)n, half = fn * 0.5f, c = 1.0f / fn;
	for (int x = 0; x < n; x++) {
		const float num = (float)x - half;
		const float want = num / fn;
		const float q = num * c;
		const float r = fmaf(-q, fn, num);
		const float got = fmaf(r, c, q);
		if (memcmp(&want, &got, sizeof want) != 0) return false;
	}
	return true;
}

// ... checked once per dimension and process; the table is shared by every scene and thread (render calls on different devices of
// one scene may run on different host threads: nothing of the scene is written here)
bool uv_division_free_cached(int n) {
	static std::mutex m;
	static std::map<int, bool> seen;
	{
		std::lock_guard<std::mutex> lock(m);
		auto it = seen.find(n);
		if (it != seen.end()) return it->second;
	}
	const bool ok = uv_division_free(n);
	std::lock_guard<std::mutex> lock(m);
	if (seen.size() >= 1024) seen.clear();
	seen[n] = ok;
	return ok;
}

void fill_frame(rt_scene* s, FrameParams& fp, const float cam[3], const float trans16[16], int w, int h, int depth) {
	fp.rcp_w = 1.0f / (float)w;
	fp.rcp_h = 1.0f / (float)h;
	fp.uv_fast = uv_division_free_cached(w) && uv_division_free_cached(h) ? 1 : 0;
	memset(fp.pose, 0, sizeof fp.pose);
	set_pose(fp, 0, cam, trans16);
	fp.n_frames = 1;
	fp.blocks_per_frame = 0;
	fp.frame_stride = 0;
	fp.width = w;
	fp.height = h;
	fp.max_bounces = depth;
	fp.tile_rows = h;
	fp.tile_first = 0;
	fp.tile_stride = 1;
	fp.local_rows = h;
	fp.tiles_per_wave = 0;
	fp.out = nullptr;
	fp.rgba8 = 0;
	fp.shade_glsl = 0;
	fp.counters = nullptr;
	fp.vis = nullptr;
	fp.vis_stride = 0;
	fp.vis_count = nullptr;
	fp.vis_list = nullptr;
	fp.vis_pad = 0.f;
	fp.hoist_mask = nullptr;
	fp.hoist_cell_x = fp.hoist_cell_y = 0.f;
}

// This stream's scratch buffers on the device (visibility buffer, per-frame lists, hoisted-triangle masks); null when a ninth
// stream asks (its launches then do without what the buffers are for).
DeviceState::VisBuf* stream_scratch(DeviceState& d, hipStream_t stream) {
	for (auto& b : d.vis_bufs) if (b.stream == (void*)stream) return &b;
	if (d.vis_bufs.size() >= 8) return nullptr;
	try { d.vis_bufs.emplace_back(); } catch (...) { return nullptr; }
	d.vis_bufs.back().stream = (void*)stream;
	return &d.vis_bufs.back();
}

// Per frame of the launch, which hoisted triangles a primary ray through each coarse cell of the view can meet (rt_vis.inl
// hoist_mask_kernel): LBVH commits with hoisted triangles, the lockstep kernel, cameras within far_guard, invertible view matrices.
int prepare_hoist_masks(rt_scene* s, DeviceState& d, const DevScene& sc, FrameParams& fp, hipStream_t stream) {
	if (d.accel != RT_ACCEL_LBVH || d.hoist_count <= 0 || fp.shade_glsl || s->kernel_kind == RT_KERNEL_STATE_MACHINE) return RT_OK;
	const int nf = fp.n_frames < 1 ? 1 : fp.n_frames;
	for (int f = 0; f < nf; f++) {
		const float* P = fp.pose[f];
		double big = 0.0, c[3][3], n2 = 0.0;
		for (int k = 0; k < 3; k++) {
			big = fmax(big, fabs((double)P[k]));
			c[0][k] = P[3 + k]; c[1][k] = P[6 + k]; c[2][k] = P[9 + k];
			n2 += c[0][k] * c[0][k] + c[1][k] * c[1][k] + c[2][k] * c[2][k];
		}
		if (!(big < (double)d.far_guard)) return RT_OK;
		const double det = c[0][0] * (c[1][1] * c[2][2] - c[1][2] * c[2][1]) - c[0][1] * (c[1][0] * c[2][2] - c[1][2] * c[2][0]) + c[0][2] * (c[1][0] * c[2][1] - c[1][1] * c[2][0]);
		if (!(fabs(det) > 1e-3 * pow(n2 / 3.0, 1.5)) || !(n2 < 1e12)) return RT_OK;     // (a view matrix that is close to singular: the boxes, as before)
	}
	DeviceState::VisBuf* vb = stream_scratch(d, stream);
	if (!vb) return RT_OK;
	if (!vb->hmask) RT_HIP(hipMalloc(&vb->hmask, (size_t)RT_MAX_FRAMES * RT_HOIST_CELLS * RT_HOIST_CELLS * sizeof(unsigned int)));
	fp.hoist_mask = (const unsigned int*)vb->hmask;
	fp.hoist_cell_x = (float)RT_HOIST_CELLS / (float)fp.width;
	fp.hoist_cell_y = (float)RT_HOIST_CELLS / (float)fp.height;
	fp.vis_pad = d.pad;
	int rc = launch_hoist_masks(sc, fp, stream);
#ifdef RT_EXPERIMENT
	if (rc == RT_OK && getenv("RT355_VIS_DEBUG")) {            // frame 0's masks: bits set per cell
		unsigned int h[RT_HOIST_CELLS * RT_HOIST_CELLS];
		(void)hipStreamSynchronize(stream);
		(void)hipMemcpy(h, vb->hmask, sizeof h, hipMemcpyDeviceToHost);
		fprintf(stderr, "hoist masks (%d hoisted):\n", d.hoist_count);
		for (int v = 0; v < RT_HOIST_CELLS; v++) {
			for (int u = 0; u < RT_HOIST_CELLS; u++) fprintf(stderr, " %2d", __builtin_popcount(h[v * RT_HOIST_CELLS + u]));
			fprintf(stderr, "\n");
		}
	}
#endif
	return rc;
}

// The primary-visibility pass (rt_vis.inl) for the frames of `fp`, where it applies: an LBVH commit, whole frames, the lockstep
// kernel, near-orthonormal view matrices, cameras within far_guard.  Sets fp.vis* and queues the pass on `stream`; leaves
// fp.vis null (the render kernel then walks the tree for primary rays as before) otherwise.
int prepare_visibility(rt_scene* s, DeviceState& d, const DevScene& sc, FrameParams& fp, bool count, hipStream_t stream) {
	// (AUTO is the walk: measured on the 102 412-triangle scene the pass and the packet walk it replaces cost the same — 800x800
	// 7.4 us/frame for the pass, bound by its ~0.5 M atomic updates at the 76 G/s this GPU sustains, against a walk of ~8; 3840x2160
	// 39 + 8 us clearing the buffer against ~46 — and the rest of the frame, the walls' tests and the shading, is unchanged)
	if (d.accel != RT_ACCEL_LBVH || !d.tri_skip || s->primary_raster != RT_PRIMARY_RASTER) return RT_OK;
	if (fp.shade_glsl || s->kernel_kind == RT_KERNEL_STATE_MACHINE) return RT_OK;
	if (!(fp.tile_stride == 1 && fp.tile_first == 0 && fp.local_rows == fp.height)) return RT_OK;
	const int nf = fp.n_frames < 1 ? 1 : fp.n_frames;
	for (int f = 0; f < nf; f++) {
		const float* P = fp.pose[f];
		double c0[3], c1[3], c2[3], big = 0.0;
		for (int k = 0; k < 3; k++) { c0[k] = P[3 + k]; c1[k] = P[6 + k]; c2[k] = 2.0 * P[9 + k]; big = fmax(big, fabs((double)P[k])); }
		if (!(big < (double)d.far_guard)) return RT_OK;
		auto dt = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
		const double tol = 1e-2;
		if (!(fabs(dt(c0, c0) - 1.0) < tol && fabs(dt(c1, c1) - 1.0) < tol && fabs(dt(c2, c2) - 1.0) < tol &&
		      fabs(dt(c0, c1)) < tol && fabs(dt(c0, c2)) < tol && fabs(dt(c1, c2)) < tol)) return RT_OK;
		const double det = c0[0] * (c1[1] * c2[2] - c1[2] * c2[1]) - c0[1] * (c1[0] * c2[2] - c1[2] * c2[0]) + c0[2] * (c1[0] * c2[1] - c1[1] * c2[0]);
		if (!(fabs(fabs(det) - 1.0) < 3.0 * tol)) return RT_OK;
	}
	const size_t pixels = (size_t)fp.width * (size_t)fp.height;
	const size_t bytes = (size_t)nf * pixels * sizeof(unsigned long long);
	if (bytes > ((size_t)8 << 30)) return RT_OK;
	DeviceState::VisBuf* vb = stream_scratch(d, stream);
	if (!vb) return RT_OK;                                         // (a ninth stream on this device: its launches walk the tree)
	if (!vb->meta) RT_HIP(hipMalloc(&vb->meta, (size_t)RT_MAX_FRAMES * (1 + RT_VIS_LIST_MAX) * sizeof(int)));
	if (vb->vis_bytes < bytes) {
		if (vb->vis) { RT_HIP(hipStreamSynchronize(stream)); (void)hipFree(vb->vis); vb->vis = nullptr; vb->vis_bytes = 0; }
		hipError_t e = hipMalloc(&vb->vis, bytes);
		if (e != hipSuccess) { (void)hipGetLastError(); return RT_OK; }     // no room for the buffer: the tree walk needs none
		vb->vis_bytes = bytes;
	}
	fp.vis = (const unsigned long long*)vb->vis;
	fp.vis_stride = (long long)pixels;
	fp.vis_count = (const int*)vb->meta;
	fp.vis_list = (const int*)vb->meta + RT_MAX_FRAMES;
	fp.vis_pad = d.pad;
	int rc = launch_vis_raster(sc, fp, (const unsigned char*)d.tri_skip, count, stream);
#ifdef RT_EXPERIMENT
	if (rc == RT_OK && getenv("RT355_VIS_DEBUG")) {            // how many triangles each frame could not project
		int h[RT_MAX_FRAMES];
		(void)hipStreamSynchronize(stream);
		(void)hipMemcpy(h, vb->meta, sizeof h, hipMemcpyDeviceToHost);
		fprintf(stderr, "vis lists:");
		for (int f = 0; f < nf; f++) fprintf(stderr, " %d", h[f]);
		fprintf(stderr, "\n");
	}
#endif
	return rc;
}

int ensure_frame(DeviceState& d, size_t bytes) {
	if (d.frame_bytes >= bytes) return RT_OK;
	if (d.frame) { (void)hipFree(d.frame); d.frame = nullptr; d.frame_bytes = 0; }
	RT_HIP(hipMalloc(&d.frame, bytes));
	d.frame_bytes = bytes;
	return RT_OK;
}

int ensure_buffer(void*& p, size_t& have, size_t bytes) {
	if (have >= bytes) return RT_OK;
	if (p) { (void)hipFree(p); p = nullptr; have = 0; }
	RT_HIP(hipMalloc(&p, bytes));
	have = bytes;
	return RT_OK;
}

// ---- rt_cast_rays_noalloc: staging that is part of the library image ---------------------------
// The device copies of the caller's rays and of the results live in this file-scope array instead
// of per-call hipMalloc'ed buffers: rays at float offset 0 (7 per ray), results behind them (6 per
// ray).  In host code the NAME `scratch` is only a stand-in for the symbol: each GPU the code
// object is loaded on holds its own copy at its own address, looked up once per device with
// hipGetSymbolAddress while that device is current; host code uses only the looked-up pointer.
#define RT_SCRATCH_FLOATS ((size_t)RT_NOALLOC_MAX_RAYS * 13)
static __device__ __align__(256) float scratch[RT_SCRATCH_FLOATS];
#define RT_MAX_DEVICES 64
std::mutex g_scratch_mutex;
float* g_scratch_ptr[RT_MAX_DEVICES];          // per HIP device ordinal; nullptr until looked up
std::mutex g_scratch_in_use[RT_MAX_DEVICES];   // one call at a time per GPU: every call on a GPU stages through that GPU's one array

// `device` must be current.
int scratch_for_device(int device, float** out) {
	if (device < 0 || device >= RT_MAX_DEVICES) { set_error("device %d beyond the %d the no-allocation path keeps a buffer for", device, RT_MAX_DEVICES); return RT_ERR_LIMIT; }
	std::lock_guard<std::mutex> lock(g_scratch_mutex);
	if (!g_scratch_ptr[device]) {
		void* p = nullptr;
		hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
		if (e != hipSuccess) return hip_fail(e, "hipGetSymbolAddress(scratch)");
		if (!p) { set_error("hipGetSymbolAddress(scratch) returned a null address on device %d", device); return RT_ERR_DEVICE; }
		g_scratch_ptr[device] = (float*)p;
	}
	*out = g_scratch_ptr[device];
	return RT_OK;
}

} // namespace

} // namespace rt

using namespace rt;

extern "C" {

void rt_scene_destroy(rt_scene* s) {
	if (!s) return;
	for (auto& kv : s->dev) free_device_state(kv.second);
	delete s;
}

int rt_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

int rt_device_props(int device, int* compute_units, int* clock_khz) {
	hipDeviceProp_t p;
	hipError_t e = hipGetDeviceProperties(&p, device);
	if (e != hipSuccess) { (void)hipGetLastError(); set_error("rt_device_props: device %d: %s", device, hipGetErrorString(e)); return RT_ERR_DEVICE; }
	if (compute_units) *compute_units = p.multiProcessorCount;
	if (clock_khz) *clock_khz = p.clockRate;
	return RT_OK;
}

const char* rt_last_error(void) { return g_err; }
const char* rt_version(void) { return "rt355 0.1 (gfx950)"; }

int rt_scene_traversal(const rt_scene* s, int device, const float trans16[16], int width, int height) {
	if (!s || !trans16) { set_error("rt_scene_traversal: NULL argument"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_traversal: not committed on device %d", device); return RT_ERR_STATE; }
	if (it->second.accel != RT_ACCEL_LBVH) return 0;
	return use_packets(dev_scene(s, it->second), trans16, width, height) ? RT_TRAVERSAL_PACKET : RT_TRAVERSAL_PER_LANE;
}

int rt_scene_set_traversal(rt_scene* s, int mode) {
	if (!s) { set_error("rt_scene_set_traversal: NULL scene"); return RT_ERR_INVALID; }
	if (mode < RT_TRAVERSAL_AUTO || mode > RT_TRAVERSAL_PACKET) { set_error("rt_scene_set_traversal: unknown mode %d", mode); return RT_ERR_INVALID; }
	s->traversal = mode;
	return RT_OK;
}

int rt_scene_set_tiles_per_wave(rt_scene* s, int tiles) {
	if (!s) { set_error("rt_scene_set_tiles_per_wave: NULL scene"); return RT_ERR_INVALID; }
	if (tiles < 0 || tiles > RT_TILES_PER_WAVE_MAX) { set_error("rt_scene_set_tiles_per_wave: %d is outside [0, %d]", tiles, RT_TILES_PER_WAVE_MAX); return RT_ERR_INVALID; }
	s->tiles_per_wave = tiles;
	return RT_OK;
}

int rt_scene_set_kernel(rt_scene* s, int kind) {
	if (!s) { set_error("rt_scene_set_kernel: NULL scene"); return RT_ERR_INVALID; }
	if (kind < RT_KERNEL_AUTO || kind > RT_KERNEL_LOCKSTEP) { set_error("rt_scene_set_kernel: unknown kind %d", kind); return RT_ERR_INVALID; }
	s->kernel_kind = kind;
	return RT_OK;
}

int rt_scene_set_build(rt_scene* s, int quality) {
	if (!s) { set_error("rt_scene_set_build: NULL scene"); return RT_ERR_INVALID; }
	if (quality != RT_BUILD_FAST && quality != RT_BUILD_SAH && quality != RT_BUILD_PLOC) { set_error("rt_scene_set_build: unknown quality %d", quality); return RT_ERR_INVALID; }
	if (s->build != quality) {
		s->build = quality;
		for (auto& kv : s->dev) if (kv.second.accel == RT_ACCEL_LBVH) kv.second.valid = false;      // trees are rebuilt at the next commit
	}
	return RT_OK;
}

int rt_scene_set_shadow_grid(rt_scene* s, int on) {
	if (!s) { set_error("rt_scene_set_shadow_grid: NULL scene"); return RT_ERR_INVALID; }
	if (on != 0 && on != 1) { set_error("rt_scene_set_shadow_grid: %d is neither 0 nor 1", on); return RT_ERR_INVALID; }
	if (s->shadow_grid != on) {
		s->shadow_grid = on;
		for (auto& kv : s->dev) if (kv.second.accel == RT_ACCEL_LBVH) kv.second.valid = false;      // grids are made (or dropped) at the next commit
	}
	return RT_OK;
}

int rt_scene_set_primary_raster(rt_scene* s, int mode) {
	if (!s) { set_error("rt_scene_set_primary_raster: NULL scene"); return RT_ERR_INVALID; }
	if (mode != RT_PRIMARY_AUTO && mode != RT_PRIMARY_WALK && mode != RT_PRIMARY_RASTER) { set_error("rt_scene_set_primary_raster: unknown mode %d", mode); return RT_ERR_INVALID; }
	s->primary_raster = mode;
	return RT_OK;
}

int rt_scene_shadow_grid(const rt_scene* s, int device, int light, int* cells_per_edge, long long* entries, int* always, float* build_ms) {
	if (!s || light < 0) { set_error("rt_scene_shadow_grid: bad argument"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_shadow_grid: not committed on device %d", device); return RT_ERR_STATE; }
	const DeviceState& d = it->second;
	const bool have = d.accel == RT_ACCEL_LBVH && light < d.n_lgrid && (size_t)light * 3 + 2 < d.lgrid_info.size();
	if (cells_per_edge) *cells_per_edge = have ? (int)d.lgrid_info[(size_t)light * 3] : 0;
	if (entries) *entries = have ? d.lgrid_info[(size_t)light * 3 + 1] : 0;
	if (always) *always = have ? (int)d.lgrid_info[(size_t)light * 3 + 2] : 0;
	if (build_ms) *build_ms = d.accel == RT_ACCEL_LBVH ? d.lgrid_ms : 0.f;
	return RT_OK;
}

int rt_scene_build_kind(const rt_scene* s, int device) {
	if (!s) { set_error("rt_scene_build_kind: NULL scene"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_build_kind: not committed on device %d", device); return RT_ERR_STATE; }
	if (it->second.accel != RT_ACCEL_LBVH) { set_error("rt_scene_build_kind: device %d holds a brute-force commit", device); return RT_ERR_STATE; }
	return it->second.built_by;
}

int rt_scene_bvh_depth(const rt_scene* s, int device, int* binary_depth, int* wide_depth) {
	if (!s || !binary_depth || !wide_depth) { set_error("rt_scene_bvh_depth: NULL argument"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_bvh_depth: not committed on device %d", device); return RT_ERR_STATE; }
	const bool tree = it->second.accel == RT_ACCEL_LBVH;
	*binary_depth = tree ? it->second.bvh_depth : 0;
	*wide_depth = tree ? it->second.bvh4_depth : 0;
	return RT_OK;
}

int rt_scene_commit(rt_scene* s, int device, int accel_flags) {
	if (!s) { set_error("rt_scene_commit: NULL scene"); return RT_ERR_INVALID; }
	if (accel_flags != RT_ACCEL_AUTO && accel_flags != RT_ACCEL_BRUTE && accel_flags != RT_ACCEL_LBVH) {
		set_error("rt_scene_commit: unknown accel flag %d", accel_flags);
		return RT_ERR_INVALID;
	}
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState& d = s->dev[device];
#ifdef RT_EXPERIMENT
	const bool times = getenv("RT355_LGRID_TIMES") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		if (!times) return;
		(void)hipDeviceSynchronize();
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "commit %-22s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_last).count());
		t_last = now;
	};
#else
	auto lap = [](const char*) {};
#endif
	if (d.device >= 0) free_device_state(d);
	lap("free previous state");
	d.device = device;
	const int n = (int)(s->tris.size() / RT_TRI_FLOATS);
	const int nl = (int)(s->lights.size() / 4);
	d.stride = ((n > 0 ? n : 1) + 63) / 64 * 64;
	std::vector<float> planes;
	const float pad = scene_box(s->tris, d.bb_lo, d.bb_hi);
	d.pad = pad;
	// Box tests are trusted only for origins within 1000 scene scales (beyond that the slab
	// arithmetic's rounding approaches the pad) and 1e6 absolute, in scenes no larger than 1e6
	// (keeps b * inv and o * inv finite for |inv| <= 1e30): rt_kernels.hip cast_ray_pruned.
	d.far_guard = pad * 1e6f < 1e6f ? pad * 1e6f : 1e6f;
	if (!(pad * 1e3f <= 1e6f)) d.far_guard = 0.f;
	// acceleration structure: decided before the upload, because only brute force needs the
	// octant-ordered box records (128 B per triangle)
	d.accel = accel_flags == RT_ACCEL_AUTO ? (n > RT_BRUTE_AUTO_MAX ? RT_ACCEL_LBVH : RT_ACCEL_BRUTE) : accel_flags;
	if (d.accel == RT_ACCEL_LBVH && n < 2) d.accel = RT_ACCEL_BRUTE;    // a tree needs two leaves
	bool box_finite = true;
	for (int k = 0; k < 3; k++) box_finite = box_finite && (d.bb_hi[k] - d.bb_lo[k] < __builtin_huge_valf());
	if (d.accel == RT_ACCEL_LBVH && accel_flags == RT_ACCEL_AUTO && !box_finite) d.accel = RT_ACCEL_BRUTE;   // a non-finite vertex
	try {
		const PlaneLayout L = flatten_planes(s->tris, d.stride, pad, d.accel == RT_ACCEL_BRUTE, planes);
		d.n_clusters = L.n_clusters;
		d.oct_clusters = L.oct_clusters; d.oct_groups = L.oct_groups;
		d.oct_cluster_pitch = L.oct_cluster_pitch; d.oct_group_pitch = L.oct_group_pitch;
	} catch (...) { set_error("rt_scene_commit: out of host memory"); return RT_ERR_NOMEM; }
	lap("scene box, planes (host)");
	// (the render kernels address the scene as a 64-bit base + 32-bit byte offset, rt_kernels.hip c4ptr)
	if (planes.size() * sizeof(float) >= 0xffffffffull) { set_error("rt_scene_commit: %d triangles need a scene buffer of 4 GiB or more", n); return RT_ERR_LIMIT; }
	RT_HIP(hipMalloc(&d.planes, planes.size() * sizeof(float)));
	RT_HIP(hipMemcpy(d.planes, planes.data(), planes.size() * sizeof(float), hipMemcpyHostToDevice));
	RT_HIP(hipMalloc(&d.lights, (size_t)(nl > 0 ? nl : 1) * 4 * sizeof(float)));
	if (nl > 0) RT_HIP(hipMemcpy(d.lights, s->lights.data(), (size_t)nl * 4 * sizeof(float), hipMemcpyHostToDevice));
	RT_HIP(hipMalloc(&d.counters, 32 * sizeof(unsigned long long)));
	lap("upload");
	d.build_ms = 0.f;
	{   // scene diagonal over the mean triangle box diagonal (rt_kernels.hip launch_render: packet or per-lane traversal)
		// (a sample of ~8192 triangles, evenly spread, for large scenes: the ratio only steers a choice between two correct traversals)
		double sum = 0.0;
		const int step = n > 16384 ? n / 8192 : 1;
		int counted = 0;
		for (int i = 0; i < n; i += step, counted++) {
			const float* v = s->tris.data() + (size_t)i * RT_TRI_FLOATS + 6;      // p1 p2 p3
			double dd = 0.0;
			for (int k = 0; k < 3; k++) {
				const float lo = fminf(v[k], fminf(v[3 + k], v[6 + k])), hi = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
				dd += (double)(hi - lo) * (double)(hi - lo);
			}
			sum += sqrt(dd);
		}
		double diag = 0.0;
		for (int k = 0; k < 3; k++) diag += (double)(d.bb_hi[k] - d.bb_lo[k]) * (double)(d.bb_hi[k] - d.bb_lo[k]);
		const double mean = counted > 0 ? sum / counted : 0.0;
		d.packet_scale = (mean > 0.0 && box_finite) ? (float)(sqrt(diag) / mean) : 3.0e38f;
	}
	if (d.accel == RT_ACCEL_LBVH) {
		// the tree's leaf boxes carry a tenth of the pad plus 1e-3 of the triangle's own size (rt_lbvh.hip
		// tri_box_padded): the slab arithmetic stays inside that for origins up to 30 scene scales away
		const float g = pad * 3e4f;
		if (d.far_guard > g) d.far_guard = g;
		lap("mean triangle size");
		int rc = lbvh_build(s, d, nullptr);
		if (rc != RT_OK) return rc;
		lap("lbvh_build");
		{   // which triangles the tree does not hold (the visibility pass leaves them to the render kernel)
			std::vector<unsigned char> skip((size_t)(n > 0 ? n : 1), 0);
			for (int i : d.hoisted) skip[(size_t)i] = 1;
			RT_HIP(hipMalloc(&d.tri_skip, skip.size()));
			RT_HIP(hipMemcpy(d.tri_skip, skip.data(), skip.size(), hipMemcpyHostToDevice));
		}
		rc = lgrid_build(s, d, nullptr);
		if (rc != RT_OK) return rc;
		lap("tri_skip + lgrid_build");
	}
	{   // set-up for rt_cast_rays_noalloc: this device's address of the library's staging array
		float* unused = nullptr;
		int rc = scratch_for_device(device, &unused);
		if (rc != RT_OK && rc != RT_ERR_LIMIT) return rc;
	}
	d.valid = true;
	if (s->first_device < 0 || !s->dev.count(s->first_device) || !s->dev[s->first_device].valid) s->first_device = device;
	return RT_OK;
}

int rt_scene_build_ms(const rt_scene* s, int device, float* out_ms) {
	if (!s || !out_ms) { set_error("rt_scene_build_ms: NULL argument"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_build_ms: not committed on device %d", device); return RT_ERR_STATE; }
	*out_ms = it->second.build_ms;
	return RT_OK;
}

int rt_scene_get_bvh(const rt_scene* s, int device, float* out_nodes16, int capacity) {
	if (!s || capacity < 0 || (capacity > 0 && !out_nodes16)) { set_error("rt_scene_get_bvh: bad argument"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_get_bvh: not committed on device %d", device); return RT_ERR_STATE; }
	const DeviceState& d = it->second;
	if (d.accel != RT_ACCEL_LBVH || !d.bvh_nodes) return 0;
	const int nodes = d.bvh_node_count;
	const int k = nodes < capacity ? nodes : capacity;
	if (k > 0) {
		DeviceGuard g(device);
		if (!g.ok) return g.rc;
		RT_HIP(hipMemcpy(out_nodes16, d.bvh_nodes, (size_t)k * 16 * sizeof(float), hipMemcpyDeviceToHost));
	}
	return nodes;
}

int rt_scene_get_hoisted(const rt_scene* s, int device, int* out_indices, int capacity) {
	if (!s || capacity < 0 || (capacity > 0 && !out_indices)) { set_error("rt_scene_get_hoisted: bad argument"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_get_hoisted: not committed on device %d", device); return RT_ERR_STATE; }
	const DeviceState& d = it->second;
	if (d.accel != RT_ACCEL_LBVH) return 0;
	const int nh = (int)d.hoisted.size();
	for (int j = 0; j < nh && j < capacity; j++) out_indices[j] = d.hoisted[j];
	return nh;
}

int rt_scene_accel(const rt_scene* s, int device) {
	if (!s) { set_error("rt_scene_accel: NULL scene"); return RT_ERR_INVALID; }
	auto it = s->dev.find(device);
	if (it == s->dev.end() || !it->second.valid) { set_error("rt_scene_accel: not committed on device %d", device); return RT_ERR_STATE; }
	return it->second.accel;
}

int rt_render_tiles_device(rt_scene* s, int device, const float cam_pos[3], const float trans16[16],
                           int width, int height, int max_bounces, int render_flags,
                           int tile_rows, int tile_first, int tile_stride,
                           float* d_out_compact, void* hip_stream) {
	int rc = check_frame_args("rt_render_tiles_device", s, cam_pos, trans16, width, height, max_bounces);
	if (rc != RT_OK) return rc;
	if (!d_out_compact) { set_error("rt_render_tiles_device: NULL output"); return RT_ERR_INVALID; }
	if (tile_rows <= 0 || tile_first < 0 || tile_stride <= 0) { set_error("rt_render_tiles_device: bad tile parameters"); return RT_ERR_INVALID; }
	if ((rc = check_flags("rt_render_tiles_device", render_flags, true)) != RT_OK) return rc;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_render_tiles_device", s, device);
	if (!d) return RT_ERR_STATE;
	FrameParams fp;
	fill_frame(s, fp, cam_pos, trans16, width, height, max_bounces);
	fp.tile_rows = tile_rows;
	fp.tile_first = tile_first;
	fp.tile_stride = tile_stride;
	fp.local_rows = rt_shard_rows(height, tile_rows, tile_first, tile_stride);
	fp.tiles_per_wave = s->tiles_per_wave;
	fp.out = d_out_compact;
	fp.rgba8 = (render_flags & RT_OUT_RGBA8) ? 1 : 0;
	fp.shade_glsl = (render_flags & RT_SHADE_GLSL) ? 1 : 0;
	const DevScene sc = dev_scene(s, *d);
	rc = prepare_hoist_masks(s, *d, sc, fp, (hipStream_t)hip_stream);
	if (rc != RT_OK) return rc;
	rc = prepare_visibility(s, *d, sc, fp, false, (hipStream_t)hip_stream);
	if (rc != RT_OK) return rc;
	return launch_render(sc, fp, d->accel, RT_RENDER_STRICT, false, (hipStream_t)hip_stream);
}

int rt_render_tiles(rt_scene* s, const float cam_pos[3], const float trans16[16], int width, int height, int max_bounces,
                    int render_flags, int tile_rows, int tile_first, int tile_stride, float* out_compact) {
	if (!s) { set_error("rt_render_tiles: NULL scene"); return RT_ERR_INVALID; }
	if (!out_compact) { set_error("rt_render_tiles: NULL output"); return RT_ERR_INVALID; }
	if (s->first_device < 0) { set_error("rt_render_tiles: scene is not committed to any device"); return RT_ERR_STATE; }
	if (tile_rows <= 0 || tile_first < 0 || tile_stride <= 0 || height <= 0 || width <= 0) { set_error("rt_render_tiles: bad tile parameters"); return RT_ERR_INVALID; }
	const int device = s->first_device;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_render_tiles", s, device);
	if (!d) return RT_ERR_STATE;
	const size_t bytes = (size_t)rt_shard_rows(height, tile_rows, tile_first, tile_stride) * (size_t)width * ((render_flags & RT_OUT_RGBA8) ? 4 : 3 * sizeof(float));
	if (bytes == 0) return check_frame_args("rt_render_tiles", s, cam_pos, trans16, width, height, max_bounces);
	int rc = ensure_frame(*d, bytes);
	if (rc != RT_OK) return rc;
	rc = rt_render_tiles_device(s, device, cam_pos, trans16, width, height, max_bounces, render_flags, tile_rows, tile_first, tile_stride, (float*)d->frame, nullptr);
	if (rc != RT_OK) return rc;
	RT_HIP(hipMemcpy(out_compact, d->frame, bytes, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_render_frames_device(rt_scene* s, int device, int n_frames, const float* cam_pos3, const float* trans16,
                            int width, int height, int max_bounces, int render_flags,
                            float* d_out_rgb, size_t frame_stride_floats, void* hip_stream) {
	if (n_frames < 0 || (n_frames > 0 && (!cam_pos3 || !trans16))) { set_error("rt_render_frames_device: bad frame list"); return RT_ERR_INVALID; }
	if (n_frames == 0) return RT_OK;
	int rc = check_frame_args("rt_render_frames_device", s, cam_pos3, trans16, width, height, max_bounces);
	if (rc != RT_OK) return rc;
	if (!d_out_rgb) { set_error("rt_render_frames_device: NULL output"); return RT_ERR_INVALID; }
	if ((rc = check_flags("rt_render_frames_device", render_flags, true)) != RT_OK) return rc;
	const bool rgba8 = (render_flags & RT_OUT_RGBA8) != 0;
	if (frame_stride_floats < (size_t)width * height * (rgba8 ? 1 : 3)) { set_error("rt_render_frames_device: frame stride smaller than a frame"); return RT_ERR_INVALID; }
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_render_frames_device", s, device);
	if (!d) return RT_ERR_STATE;
	const DevScene sc = dev_scene(s, *d);
	for (int f0 = 0; f0 < n_frames; f0 += RT_MAX_FRAMES) {
		const int nf = n_frames - f0 < RT_MAX_FRAMES ? n_frames - f0 : RT_MAX_FRAMES;
		FrameParams fp;
		fill_frame(s, fp, cam_pos3 + 3 * (size_t)f0, trans16 + 16 * (size_t)f0, width, height, max_bounces);
		for (int k = 0; k < nf; k++) {
			set_pose(fp, k, cam_pos3 + 3 * (size_t)(f0 + k), trans16 + 16 * (size_t)(f0 + k));
		}
		fp.n_frames = nf;
		fp.frame_stride = (long long)frame_stride_floats;
		fp.tiles_per_wave = s->tiles_per_wave;
		fp.out = d_out_rgb + (size_t)f0 * frame_stride_floats;
		fp.rgba8 = rgba8 ? 1 : 0;
		fp.shade_glsl = (render_flags & RT_SHADE_GLSL) ? 1 : 0;
		rc = prepare_hoist_masks(s, *d, sc, fp, (hipStream_t)hip_stream);
		if (rc != RT_OK) return rc;
		rc = prepare_visibility(s, *d, sc, fp, false, (hipStream_t)hip_stream);
		if (rc != RT_OK) return rc;
		rc = launch_render(sc, fp, d->accel, RT_RENDER_STRICT, false, (hipStream_t)hip_stream);
		if (rc != RT_OK) return rc;
	}
	return RT_OK;
}

int rt_shard_rows(int height, int tile_rows, int tile_first, int tile_stride) {
	if (height <= 0 || tile_rows <= 0 || tile_first < 0 || tile_stride <= 0) return 0;
	int rows = 0;
	for (long long t = tile_first; t * tile_rows < height; t += tile_stride) {
		long long r0 = t * tile_rows, r1 = r0 + tile_rows;
		if (r1 > height) r1 = height;
		rows += (int)(r1 - r0);
	}
	return rows;
}

int rt_render_device(rt_scene* s, int device, const float cam_pos[3], const float trans16[16],
                     int width, int height, int max_bounces, int render_flags,
                     float* d_out_rgb, void* hip_stream) {
	// one tile == the whole frame
	return rt_render_tiles_device(s, device, cam_pos, trans16, width, height, max_bounces, render_flags,
	                              height > 0 ? height : 1, 0, 1, d_out_rgb, hip_stream);
}

int rt_render(rt_scene* s, const float cam_pos[3], const float trans16[16], int width, int height,
              int max_bounces, int render_flags, float* out_rgb) {
	int rc = check_frame_args("rt_render", s, cam_pos, trans16, width, height, max_bounces);
	if (rc != RT_OK) return rc;
	if (!out_rgb) { set_error("rt_render: NULL output"); return RT_ERR_INVALID; }
	if (s->first_device < 0) { set_error("rt_render: scene is not committed to any device"); return RT_ERR_STATE; }
	const int device = s->first_device;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_render", s, device);
	if (!d) return RT_ERR_STATE;
	const size_t bytes = (size_t)width * height * ((render_flags & RT_OUT_RGBA8) ? 1 : 3) * sizeof(float);
	rc = ensure_frame(*d, bytes);
	if (rc != RT_OK) return rc;
	rc = rt_render_device(s, device, cam_pos, trans16, width, height, max_bounces, render_flags, (float*)d->frame, nullptr);
	if (rc != RT_OK) return rc;
	RT_HIP(hipMemcpy(out_rgb, d->frame, bytes, hipMemcpyDeviceToHost));
	return RT_OK;
}

int rt_render_stats(rt_scene* s, const float cam_pos[3], const float trans16[16], int width,
                    int height, int max_bounces, float* out_rgb, rt_stats* out) {
	int rc = check_frame_args("rt_render_stats", s, cam_pos, trans16, width, height, max_bounces);
	if (rc != RT_OK) return rc;
	if (!out) { set_error("rt_render_stats: NULL stats"); return RT_ERR_INVALID; }
	if (s->first_device < 0) { set_error("rt_render_stats: scene is not committed to any device"); return RT_ERR_STATE; }
	const int device = s->first_device;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_render_stats", s, device);
	if (!d) return RT_ERR_STATE;
	const size_t bytes = (size_t)width * height * 3 * sizeof(float);
	rc = ensure_frame(*d, bytes);
	if (rc != RT_OK) return rc;
	RT_HIP(hipMemsetAsync(d->counters, 0, 32 * sizeof(unsigned long long), nullptr));
	FrameParams fp;
	fill_frame(s, fp, cam_pos, trans16, width, height, max_bounces);
	fp.tiles_per_wave = s->tiles_per_wave;
	fp.out = (float*)d->frame;
	fp.counters = (unsigned long long*)d->counters;
#ifdef RT_PROFILE
	{
		const DevScene sc = dev_scene(s, *d);
		rc = prepare_hoist_masks(s, *d, sc, fp, nullptr);
		if (rc == RT_OK) rc = launch_render(sc, fp, d->accel, RT_RENDER_STRICT, false, nullptr);   // wave-level tuning counters
	}
#else
	{
		const DevScene sc = dev_scene(s, *d);
		rc = prepare_hoist_masks(s, *d, sc, fp, nullptr);
		if (rc == RT_OK) rc = prepare_visibility(s, *d, sc, fp, true, nullptr);             // (counted like the frame is rendered)
		if (rc == RT_OK) rc = launch_render(sc, fp, d->accel, RT_RENDER_STRICT, true, nullptr);
	}
#endif
	if (rc != RT_OK) return rc;
	unsigned long long c[32];
	RT_HIP(hipMemcpy(c, d->counters, sizeof c, hipMemcpyDeviceToHost));
	if (out_rgb) RT_HIP(hipMemcpy(out_rgb, d->frame, bytes, hipMemcpyDeviceToHost));
	out->rays_primary = c[0]; out->rays_bounce = c[1]; out->rays_shadow = c[2];
	out->n_plane = c[3]; out->n_point = c[4]; out->n_edge = c[5]; out->n_box = c[6];
	out->px_hit = c[7]; out->px_mirror = c[8]; out->px_lit = c[9];
#ifdef RT_PROFILE
	fprintf(stderr, "RT_PROFILE cycles: staging %llu setup %llu trace-rest %llu shade %llu | boxes %llu prefilter %llu exact %llu bounds %llu\n",
	        c[10], c[11], c[12], c[13], c[14], c[15], c[16], c[17]);
	fprintf(stderr, "RT_PROFILE lbvh: trips %llu | node trips %llu lanes %llu | leaf trips %llu lanes %llu | distinct nodes %llu, trips with 1 node %llu, with <=4 %llu\n",
	        c[20], c[21], c[22], c[23], c[24], c[25], c[26], c[27]);
#endif
	return RT_OK;
}

int rt_render_multi(rt_scene* s, const int* devices, int n_devices, const float cam_pos[3],
                    const float trans16[16], int width, int height, int max_bounces,
                    int render_flags, int tile_rows, float* out_rgb) {
	int rc = check_frame_args("rt_render_multi", s, cam_pos, trans16, width, height, max_bounces);
	if (rc != RT_OK) return rc;
	if (!devices || n_devices <= 0 || !out_rgb || tile_rows <= 0) { set_error("rt_render_multi: bad argument"); return RT_ERR_INVALID; }
	if ((rc = check_flags("rt_render_multi", render_flags, true)) != RT_OK) return rc;
	// (a device may be listed more than once: it then renders several shards, one after the other — a rehearsal of the N-GPU
	// tile assignment on fewer GPUs)
	// 1. commit where needed (keeps the acceleration choice of the first committed device)
	int accel = RT_ACCEL_AUTO;
	for (int k = 0; k < n_devices; k++) {
		auto it = s->dev.find(devices[k]);
		if (it != s->dev.end() && it->second.valid) { accel = it->second.accel; break; }
	}
	for (int k = 0; k < n_devices; k++) {
		auto it = s->dev.find(devices[k]);
		if (it == s->dev.end() || !it->second.valid) {
			rc = rt_scene_commit(s, devices[k], accel);
			if (rc != RT_OK) return rc;
		}
	}
	// 2. every device renders its interleaved row tiles into a compact buffer: devices[0] straight into its piece of the
	//    gather area on devices[0] (shard k at k * pitch, then the frame), the others into their scratch frames
	const bool rgba8 = (render_flags & RT_OUT_RGBA8) != 0;
	const size_t row_bytes = (size_t)width * (rgba8 ? 4 : 3 * sizeof(float));
	const size_t frame_bytes = row_bytes * (size_t)height;
	size_t pitch = (size_t)rt_shard_rows(height, tile_rows, 0, n_devices) * row_bytes;       // (shard 0 is the largest)
	pitch = (pitch + 255) & ~(size_t)255;
	const size_t frame_at = pitch * (size_t)n_devices;
	const int root = devices[0];
	std::vector<hipStream_t> streams((size_t)n_devices, nullptr);
	std::vector<hipEvent_t> landed((size_t)n_devices, nullptr);
	std::vector<void*> extra((size_t)n_devices, nullptr);          // shard buffers of a non-root device's second, third ... listing
	auto cleanup = [&]() {
		for (int k = 0; k < n_devices; k++) {
			if (landed[k] || extra[k]) (void)hipSetDevice(devices[k]);
			if (extra[k]) (void)hipFree(extra[k]);
			if (landed[k]) (void)hipEventDestroy(landed[k]);          // (the streams stay with their devices' states)
		}
	};
	int prev = -1;
	(void)hipGetDevice(&prev);
	if (hipSetDevice(root) != hipSuccess) { set_error("rt_render_multi: hipSetDevice(%d) failed", root); (void)hipGetLastError(); return RT_ERR_DEVICE; }
	{
		DeviceState& d0 = s->dev[root];
		rc = ensure_buffer(d0.gather, d0.gather_bytes, frame_at + frame_bytes);
	}
	for (int k = 1; k < n_devices && rc == RT_OK; k++) {
		// let devices[k] write into devices[0]'s memory directly (xGMI); already enabled is fine
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, devices[k], root) == hipSuccess && can && hipSetDevice(devices[k]) == hipSuccess)
			(void)hipDeviceEnablePeerAccess(root, 0);        // (without it the peer copy below is staged through the host)
		(void)hipGetLastError();
	}
	char* const area = (char*)s->dev[root].gather;
	for (int k = 0; k < n_devices && rc == RT_OK; k++) {
		DeviceState& d = s->dev[devices[k]];
		const int rows = rt_shard_rows(height, tile_rows, k, n_devices);
		if (hipSetDevice(devices[k]) != hipSuccess) { rc = RT_ERR_DEVICE; set_error("rt_render_multi: hipSetDevice(%d) failed", devices[k]); break; }
		{   // this listing's stream on its device: made once, kept (a new stream per call would also use up the per-stream scratch slots)
			size_t nth = 0;
			for (int a = 0; a < k; a++) nth += devices[a] == devices[k];
			try { if (d.multi_streams.size() <= nth) d.multi_streams.resize(nth + 1, nullptr); } catch (...) { rc = RT_ERR_NOMEM; set_error("rt_render_multi: out of host memory"); break; }
			if (!d.multi_streams[nth] && hipStreamCreateWithFlags(&d.multi_streams[nth], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); rc = RT_ERR_DEVICE; set_error("rt_render_multi: stream creation failed"); break; }
			streams[k] = d.multi_streams[nth];
		}
		if (hipEventCreateWithFlags(&landed[k], hipEventDisableTiming) != hipSuccess) { rc = RT_ERR_DEVICE; set_error("rt_render_multi: event creation failed"); break; }
		if (rows > 0) {
			float* target = (float*)(area + (size_t)k * pitch);          // devices[0] renders where the shard belongs
			if (devices[k] != root) {
				bool first = true;
				for (int a = 0; a < k; a++) first = first && devices[a] != devices[k];
				if (first) {
					rc = ensure_frame(d, (size_t)rows * row_bytes);
					if (rc != RT_OK) break;
					target = (float*)d.frame;
				} else {
					hipError_t e = hipMalloc(&extra[k], (size_t)rows * row_bytes);
					if (e != hipSuccess) { rc = hip_fail(e, "rt_render_multi shard buffer"); break; }
					target = (float*)extra[k];
				}
			}
			rc = rt_render_tiles_device(s, devices[k], cam_pos, trans16, width, height, max_bounces, render_flags,
			                            tile_rows, k, n_devices, target, streams[k]);
			if (rc != RT_OK) break;
			// 3. the exchange step: ONE peer copy per device, queued behind its render (7 concurrent xGMI transfers into GPU 0 on a full node)
			if (devices[k] != root) {
				hipError_t e = hipMemcpyPeerAsync(area + (size_t)k * pitch, root, target, devices[k], (size_t)rows * row_bytes, streams[k]);
				if (e != hipSuccess) { rc = hip_fail(e, "rt_render_multi gather"); break; }
			}
		}
		hipError_t e = hipEventRecord(landed[k], streams[k]);
		if (e != hipSuccess) { rc = hip_fail(e, "rt_render_multi event"); break; }
	}
	// 4. devices[0] de-interleaves the shards into the frame (one kernel) and copies it to the caller once
	if (rc == RT_OK) {
		(void)hipSetDevice(root);
		for (int k = 1; k < n_devices && rc == RT_OK; k++) {
			hipError_t e = hipStreamWaitEvent(streams[0], landed[k], 0);
			if (e != hipSuccess) rc = hip_fail(e, "rt_render_multi wait");
		}
		if (rc == RT_OK) rc = launch_scatter_shards(area, pitch, n_devices, row_bytes, height, tile_rows, area + frame_at, streams[0]);
		if (rc == RT_OK) {
			hipError_t e = hipMemcpyAsync(out_rgb, area + frame_at, frame_bytes, hipMemcpyDeviceToHost, streams[0]);
			if (e != hipSuccess) rc = hip_fail(e, "rt_render_multi copy to host");
		}
	}
	for (int k = 0; k < n_devices; k++) {
		if (!streams[k]) continue;
		(void)hipSetDevice(devices[k]);
		hipError_t e = hipStreamSynchronize(streams[k]);
		if (e != hipSuccess && rc == RT_OK) rc = hip_fail(e, "rt_render_multi sync");
	}
	cleanup();
	if (prev >= 0) (void)hipSetDevice(prev);
	return rc;
}

// The de-interleave on its own, for exchanges that run outside this library (torch.distributed / RCCL, MPI): SURVEY 8e.
static int check_scatter(const char* fn, const void* shards, size_t pitch_bytes, int n_shards, int width, int height, int tile_rows, int render_flags, const void* frame, size_t* row_bytes) {
	if (!shards || !frame || n_shards <= 0 || width <= 0 || height <= 0 || tile_rows <= 0) { set_error("%s: bad argument", fn); return RT_ERR_INVALID; }
	int rc = check_flags(fn, render_flags, true);
	if (rc != RT_OK) return rc;
	*row_bytes = (size_t)width * ((render_flags & RT_OUT_RGBA8) ? 4 : 3 * sizeof(float));
	if (pitch_bytes % 4 != 0 || pitch_bytes < (size_t)rt_shard_rows(height, tile_rows, 0, n_shards) * *row_bytes) {
		set_error("%s: pitch %zu is smaller than the largest shard (%d rows of %zu bytes) or not a multiple of 4", fn, pitch_bytes, rt_shard_rows(height, tile_rows, 0, n_shards), *row_bytes);
		return RT_ERR_INVALID;
	}
	return RT_OK;
}

int rt_scatter_shards(int device, const void* d_shards, size_t pitch_bytes, int n_shards, int width, int height, int tile_rows,
                      int render_flags, void* d_frame, void* hip_stream) {
	size_t row_bytes = 0;
	int rc = check_scatter("rt_scatter_shards", d_shards, pitch_bytes, n_shards, width, height, tile_rows, render_flags, d_frame, &row_bytes);
	if (rc != RT_OK) return rc;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	return launch_scatter_shards(d_shards, pitch_bytes, n_shards, row_bytes, height, tile_rows, d_frame, (hipStream_t)hip_stream);
}

int rt_scatter_shards_host(const void* shards, size_t pitch_bytes, int n_shards, int width, int height, int tile_rows,
                           int render_flags, void* frame) {
	size_t row_bytes = 0;
	int rc = check_scatter("rt_scatter_shards_host", shards, pitch_bytes, n_shards, width, height, tile_rows, render_flags, frame, &row_bytes);
	if (rc != RT_OK) return rc;
	// tile t = rows [t * tile_rows, ...) is rank (t % n)'s (t / n)-th tile: whole tiles move in one memcpy each
	for (long long t = 0; t * tile_rows < height; t++) {
		const long long r0 = t * tile_rows, r1 = std::min<long long>(r0 + tile_rows, height);
		const char* src = (const char*)shards + (size_t)(t % n_shards) * pitch_bytes + (size_t)(t / n_shards) * (size_t)tile_rows * row_bytes;
		memcpy((char*)frame + (size_t)r0 * row_bytes, src, (size_t)(r1 - r0) * row_bytes);
	}
	return RT_OK;
}

// ---- the hand-off to the viewer, pipelined (Source.cpp:1270) -------------------------------------

// Frame copy into pinned host memory by the shader cores (16-byte coalesced stores over PCIe) instead of a copy
// engine: measured per 7.68 MB frame, pipelined, 0.29 ms with hipMemcpyAsync (26 GB/s) against 0.174 ms with this
// kernel on 256 workgroups (44 GB/s; 64 workgroups 0.198, 1024 the same as 256).
__global__ void __launch_bounds__(256) blit_frame_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n_floats) {
	const size_t n4 = n_floats / 4;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
		((float4*)dst)[i] = ((const float4*)src)[i];
	for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_floats; i += stride)
		dst[i] = src[i];
}

float* rt_frame_alloc(size_t n_floats) {
	void* p = nullptr;
	hipError_t e = hipHostMalloc(&p, (n_floats > 0 ? n_floats : 1) * sizeof(float), hipHostMallocDefault);
	if (e != hipSuccess) { (void)hip_fail(e, "hipHostMalloc (rt_frame_alloc)"); return nullptr; }
	return (float*)p;
}

void rt_frame_free(float* frame) {
	if (frame) { (void)hipHostFree(frame); (void)hipGetLastError(); }
}

int rt_render_async(rt_scene* s, const float cam_pos[3], const float trans16[16], int width, int height,
                    int max_bounces, int render_flags, float* out_rgb) {
	int rc = check_frame_args("rt_render_async", s, cam_pos, trans16, width, height, max_bounces);
	if (rc != RT_OK) return rc;
	if (!out_rgb) { set_error("rt_render_async: NULL output"); return RT_ERR_INVALID; }
	if ((rc = check_flags("rt_render_async", render_flags, true)) != RT_OK) return rc;
	if (s->first_device < 0) { set_error("rt_render_async: scene is not committed to any device"); return RT_ERR_STATE; }
	const int device = s->first_device;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_render_async", s, device);
	if (!d) return RT_ERR_STATE;
	if (s->next_ticket >= 0x7fffffffLL) { set_error("rt_render_async: ticket counter exhausted"); return RT_ERR_LIMIT; }
	if (!d->render_stream) {
		RT_HIP(hipStreamCreateWithFlags(&d->render_stream, hipStreamNonBlocking));
		RT_HIP(hipStreamCreateWithFlags(&d->render_stream2, hipStreamNonBlocking));
		RT_HIP(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
		for (int k = 0; k < 2; k++) {
			RT_HIP(hipEventCreateWithFlags(&d->rendered[k], hipEventDisableTiming));
			RT_HIP(hipEventCreateWithFlags(&d->copied[k], hipEventDisableTiming));
		}
	}
	const long long ticket = s->next_ticket;
	const int slot = (int)(ticket & 1);
	const size_t bytes = (size_t)width * height * ((render_flags & RT_OUT_RGBA8) ? 1 : 3) * sizeof(float);
	// Each slot renders on a stream of its own: a lone 800x800 frame is ~10 wavefronts per SIMD, and on ONE stream frame f+1 cannot
	// start before the last wavefront of f has ended — measured 23.6 us per frame back to back against 12.6 with the two
	// alternating (scripts/single_frame.py).  Copies stay in ticket order on the one copy stream.
	const hipStream_t rstream = slot ? d->render_stream2 : d->render_stream;
	if (d->ticket_of_slot[slot] >= 0) {
		// the device frame of this slot is free once its copy (two frames ago) has finished
		if (d->aframe_bytes[slot] < bytes) RT_HIP(hipEventSynchronize(d->copied[slot]));   // about to be reallocated
		else RT_HIP(hipStreamWaitEvent(rstream, d->copied[slot], 0));
	}
	rc = ensure_buffer(d->aframe[slot], d->aframe_bytes[slot], bytes);
	if (rc != RT_OK) return rc;
	rc = rt_render_device(s, device, cam_pos, trans16, width, height, max_bounces, render_flags, (float*)d->aframe[slot], rstream);
	if (rc != RT_OK) return rc;
	RT_HIP(hipEventRecord(d->rendered[slot], rstream));
	RT_HIP(hipStreamWaitEvent(d->copy_stream, d->rendered[slot], 0));
	{
		// a pinned destination (rt_frame_alloc) is written by a copy kernel; pageable memory goes through the runtime
		hipPointerAttribute_t attr;
		float* dev_view = nullptr;
		if (hipPointerGetAttributes(&attr, out_rgb) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer && ((uintptr_t)attr.devicePointer & 15u) == 0)
			dev_view = (float*)attr.devicePointer;
		(void)hipGetLastError();
		if (dev_view) {
			hipLaunchKernelGGL(blit_frame_kernel, dim3(RT_BLIT_BLOCKS), dim3(256), 0, d->copy_stream, (const float*)d->aframe[slot], dev_view, bytes / sizeof(float));
			RT_HIP(hipGetLastError());
		} else {
			RT_HIP(hipMemcpyAsync(out_rgb, d->aframe[slot], bytes, hipMemcpyDeviceToHost, d->copy_stream));
		}
	}
	RT_HIP(hipEventRecord(d->copied[slot], d->copy_stream));
	d->ticket_of_slot[slot] = ticket;
	s->next_ticket = ticket + 1;
	return (int)ticket;
}

int rt_render_wait(rt_scene* s, int ticket) {
	if (!s) { set_error("rt_render_wait: NULL scene"); return RT_ERR_INVALID; }
	if (ticket < 0 || ticket >= s->next_ticket) { set_error("rt_render_wait: ticket %d was not issued", ticket); return RT_ERR_INVALID; }
	if (s->first_device < 0) { set_error("rt_render_wait: scene is not committed to any device"); return RT_ERR_STATE; }
	auto it = s->dev.find(s->first_device);
	if (it == s->dev.end() || !it->second.copy_stream) { set_error("rt_render_wait: no frame in flight (the scene was committed again?)"); return RT_ERR_STATE; }
	DeviceState& d = it->second;
	DeviceGuard g(d.device);
	if (!g.ok) return g.rc;
	// copies complete in ticket order on the one copy stream: waiting for the slot that holds this
	// ticket (or a later one that re-used it) covers it and every earlier frame
	const int slot = ticket & 1;
	if (d.ticket_of_slot[slot] >= ticket) RT_HIP(hipEventSynchronize(d.copied[slot]));
	return RT_OK;
}

int rt_cast_rays(rt_scene* s, int n_rays, const float* rays7, float* out6) {
	if (!s || n_rays < 0 || (n_rays > 0 && (!rays7 || !out6))) { set_error("rt_cast_rays: bad argument"); return RT_ERR_INVALID; }
	if (s->first_device < 0) { set_error("rt_cast_rays: scene is not committed to any device"); return RT_ERR_STATE; }
	if (n_rays == 0) return RT_OK;
	const int device = s->first_device;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_cast_rays", s, device);
	if (!d) return RT_ERR_STATE;
	float *dr = nullptr, *dout = nullptr;
	RT_HIP(hipMalloc((void**)&dr, (size_t)n_rays * 7 * sizeof(float)));
	hipError_t e = hipMalloc((void**)&dout, (size_t)n_rays * 6 * sizeof(float));
	if (e != hipSuccess) { (void)hipFree(dr); return hip_fail(e, "hipMalloc"); }
	int rc = RT_OK;
	e = hipMemcpy(dr, rays7, (size_t)n_rays * 7 * sizeof(float), hipMemcpyHostToDevice);
	if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy rays");
	if (rc == RT_OK) rc = launch_cast_rays(dev_scene(s, *d), d->accel, n_rays, dr, dout, nullptr);
	if (rc == RT_OK) {
		e = hipMemcpy(out6, dout, (size_t)n_rays * 6 * sizeof(float), hipMemcpyDeviceToHost);
		if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy hits");
	}
	(void)hipFree(dr);
	(void)hipFree(dout);
	return rc;
}

int rt_cast_rays_noalloc(rt_scene* s, int n_rays, const float* rays7, float* out6) {
	if (!s || n_rays < 0 || (n_rays > 0 && (!rays7 || !out6))) { set_error("rt_cast_rays_noalloc: bad argument"); return RT_ERR_INVALID; }
	if (n_rays > RT_NOALLOC_MAX_RAYS) {
		set_error("rt_cast_rays_noalloc: %d rays exceed the %d the library's own staging buffer holds (split the batch, or use rt_cast_rays)", n_rays, RT_NOALLOC_MAX_RAYS);
		return RT_ERR_LIMIT;
	}
	if (s->first_device < 0) { set_error("rt_cast_rays_noalloc: scene is not committed to any device"); return RT_ERR_STATE; }
	if (n_rays == 0) return RT_OK;
	const int device = s->first_device;
	DeviceGuard g(device);
	if (!g.ok) return g.rc;
	DeviceState* d = committed("rt_cast_rays_noalloc", s, device);
	if (!d) return RT_ERR_STATE;
	float* base = nullptr;                         // this device's copy of `scratch` (looked up at commit, else here)
	int rc = scratch_for_device(device, &base);
	if (rc != RT_OK) return rc;
	float* const dr = base;                                              // n_rays x 7
	float* const dout = base + (size_t)RT_NOALLOC_MAX_RAYS * 7;          // n_rays x 6
	// (the copies are blocking and on the null stream: the call itself cannot be recorded into a HIP graph — it serves callers that
	// must not ALLOCATE; calls from several host threads take turns on a GPU's array)
	std::lock_guard<std::mutex> turn(g_scratch_in_use[device]);
	RT_HIP(hipMemcpy(dr, rays7, (size_t)n_rays * 7 * sizeof(float), hipMemcpyHostToDevice));
	rc = launch_cast_rays(dev_scene(s, *d), d->accel, n_rays, dr, dout, nullptr);
	if (rc != RT_OK) return rc;
	RT_HIP(hipMemcpy(out6, dout, (size_t)n_rays * 6 * sizeof(float), hipMemcpyDeviceToHost));
	return RT_OK;
}

} // extern "C"
