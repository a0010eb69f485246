// rt_headless.cpp — headless stand-in for the reference's main() (Source.cpp:803-1314) without the
// GL viewer: builds a scene through the C ABI exactly as main() does through addTriangle /
// lights.push_back (Source.cpp:815-851, 880), renders frames with rt_render (which replaces the
// getPixelData loop, Source.cpp:1248-1264), prints the reference's per-frame "ms / FPS" line
// (Source.cpp:1297-1301) and dumps the framebuffer for comparison:
//
//   Image.txt / .ppm (P3)  "P3\nW H\n255\n" then one "R G B" triple per line — the layout of the
//                          stale RayTracing/src/Image.txt / Image.ppm the reference tree carries
//   .ppm (P6)              binary, same pixels
//   .f32                   the raw float data[h*w*3] (row 0 = bottom, Source.cpp:550)
//
// Image files are written top row first (data row h-1 first) so that a viewer shows what the GL
// quad shows.  8-bit values use the GL unorm rule q = (int)(c*255.0f + 0.5f).
//
//   rt_headless [--scene stock|FILE] [--size W H] [--depth D] [--camera px py pz rx ry rz]
//               [--frames N] [--async] [--rgba8] [--glsl] [--lbvh|--brute] [--device K]
//               [--txt Image.txt] [--ppm out.ppm] [--raw out.f32] [--rto model.rto]
//               [--walk SCRIPT]
//
// --walk drives the camera as the reference's frame loop does (Source.cpp:1294-1295, 1307-1308):
// SCRIPT has one line per frame, `frame ms ll lr lu ld l r u d f b` (frame time in ms, then the look
// and move key flags of Source.cpp:237-241; the format oracle/_ref's `camera` mode reads); each frame
// is drawn with the current pose, then rt_camera_step moves on.  The last frame drawn is dumped.
//
// FILE uses the text format of oracle/SCENE_FORMAT.md.  No CPU rendering happens here.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt355.h"

namespace {

struct Options {
	std::string scene = "stock";
	int width = 800, height = 800;     // Source.cpp:887-888
	bool size_set = false;
	int depth = 1;                     // Source.cpp:567
	bool depth_set = false;
	float cam[6] = { -5, 5, 5, 135, -45, 0 };   // Source.cpp:810
	bool cam_set = false;
	int frames = 1;
	bool glsl = false;          // --glsl: RT_SHADE_GLSL, the colour arithmetic of the reference's compute shader (rt355.h)
	bool rgba8 = false;         // --rgba8: frames as RT_OUT_RGBA8 pixels (the 8 bits GL makes of `data`); images are written from them, --raw dumps them
	bool pipelined = false;     // --async: rt_render_async into two pinned frames (the hand-off of Source.cpp:1270 overlapped with the next render)
	int accel = RT_ACCEL_AUTO;
	int device = 0;
	std::string txt, ppm, raw, rto, walk;
};

int fail(const char* what) {
	fprintf(stderr, "rt_headless: %s: %s\n", what, rt_last_error());
	return 1;
}

// Source.cpp:815-851, 880: the stock scene, call for call.
int build_stock(rt_scene* s) {
	struct Wall { float t[3]; float r[3]; float col[3]; };
	const Wall room[6] = {
		{ { 0, 0, 10 }, { 0, 0, 0 }, { 255, 0, 0 } },   { { 0, 0, -10 }, { 0, 0, 0 }, { 255, 0, 128 } },
		{ { 10, 0, 0 }, { 0, 90, 0 }, { 0, 255, 0 } },  { { -10, 0, 0 }, { 0, 90, 0 }, { 0, 255, 128 } },
		{ { 0, -10, 0 }, { 90, 0, 0 }, { 0, 0, 255 } }, { { 0, 10, 0 }, { 90, 0, 0 }, { 255, 0, 255 } },
	};
	const Wall cube[6] = {
		{ { 0, 0, 1 }, { 0, 0, 0 }, { 255, 255, 255 } },  { { 0, 0, -1 }, { 0, 0, 0 }, { 255, 255, 255 } },
		{ { 1, 0, 0 }, { 0, 90, 0 }, { 255, 255, 255 } }, { { -1, 0, 0 }, { 0, 90, 0 }, { 255, 255, 255 } },
		{ { 0, -1, 0 }, { 90, 0, 0 }, { 255, 255, 255 } }, { { 0, 1, 0 }, { 90, 0, 0 }, { 255, 255, 255 } },
	};
	for (int pass = 0; pass < 2; pass++) {
		const Wall* w = pass ? cube : room;
		const float sc = pass ? 1.0f : 10.0f;
		const float S[3] = { sc, sc, sc };
		for (int i = 0; i < 6; i++) {
			float r2[3] = { w[i].r[0], w[i].r[1], w[i].r[2] + 180.0f };
			if (rt_scene_add_quad_half(s, w[i].t, w[i].r, S, w[i].col, 0) != RT_OK) return -1;
			if (rt_scene_add_quad_half(s, w[i].t, r2, S, w[i].col, 0) != RT_OK) return -1;
		}
	}
	const float lp[3] = { 0, 8, 0 };
	return rt_scene_add_light(s, lp, 4096);
}

int build_from_file(rt_scene* s, Options& o) {
	FILE* f = fopen(o.scene.c_str(), "r");
	if (!f) { fprintf(stderr, "rt_headless: cannot open %s\n", o.scene.c_str()); return -1; }
	char line[2048];
	int lineno = 0;
	while (fgets(line, sizeof line, f)) {
		lineno++;
		char* p = line;
		while (*p == ' ' || *p == '\t') p++;
		if (*p == '#' || *p == '\n' || *p == 0) continue;
		char kw[32];
		int off = 0;
		if (sscanf(p, "%31s%n", kw, &off) != 1) continue;
		p += off;
		float v[16];
		int n = 0;
		while (n < 16) {
			char* end;
			float x = strtof(p, &end);
			if (end == p) break;
			v[n++] = x;
			p = end;
		}
		int rc = RT_OK;
		if (!strcmp(kw, "quadhalf") && n == 13) rc = rt_scene_add_quad_half(s, v, v + 3, v + 6, v + 9, v[12]);
		else if (!strcmp(kw, "tri") && n == 13) rc = rt_scene_add_triangle(s, v, v + 3, v + 6, v + 9, v[12]);
		else if (!strcmp(kw, "light") && n == 4) rc = rt_scene_add_light(s, v, v[3]);
		else if (!strcmp(kw, "camera") && n == 6) { if (!o.cam_set) memcpy(o.cam, v, sizeof o.cam); }
		else if (!strcmp(kw, "size") && n == 2) { if (!o.size_set) { o.width = (int)v[0]; o.height = (int)v[1]; } }
		else if (!strcmp(kw, "depth") && n == 1) { if (!o.depth_set) o.depth = (int)v[0]; }
		else { fprintf(stderr, "rt_headless: %s:%d: bad line\n", o.scene.c_str(), lineno); fclose(f); return -1; }
		if (rc != RT_OK) { fclose(f); return -1; }
	}
	fclose(f);
	return 0;
}

unsigned char q8(float c) {
	if (!(c > 0.0f)) return 0;
	if (c > 1.0f) c = 1.0f;
	return (unsigned char)(int)(c * 255.0f + 0.5f);
}

// `data` holds float RGB, or (rgba8) one packed R, G, B, 255 pixel per float slot as RT_OUT_RGBA8 frames do
bool write_image(const std::string& path, const std::vector<float>& data, int w, int h, bool text, bool rgba8) {
	FILE* f = fopen(path.c_str(), text ? "w" : "wb");
	if (!f) return false;
	fprintf(f, "%s\n%d %d\n255\n", text ? "P3" : "P6", w, h);
	const unsigned char* bytes = (const unsigned char*)data.data();
	for (int y = h - 1; y >= 0; y--)
		for (int x = 0; x < w; x++) {
			unsigned char rgb[3];
			if (rgba8) memcpy(rgb, bytes + ((size_t)y * w + x) * 4, 3);
			else { const float* p = &data[((size_t)y * w + x) * 3]; rgb[0] = q8(p[0]); rgb[1] = q8(p[1]); rgb[2] = q8(p[2]); }
			if (text) fprintf(f, "%d %d %d\n", rgb[0], rgb[1], rgb[2]);
			else fwrite(rgb, 1, 3, f);
		}
	return fclose(f) == 0;
}

} // namespace

int main(int argc, char** argv) {
	Options o;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto need = [&](int k) { if (i + k >= argc) { fprintf(stderr, "rt_headless: %s needs %d value(s)\n", a.c_str(), k); exit(2); } };
		if (a == "--scene") { need(1); o.scene = argv[++i]; }
		else if (a == "--size") { need(2); o.width = atoi(argv[++i]); o.height = atoi(argv[++i]); o.size_set = true; }
		else if (a == "--depth") { need(1); o.depth = atoi(argv[++i]); o.depth_set = true; }
		else if (a == "--camera") { need(6); for (int k = 0; k < 6; k++) o.cam[k] = strtof(argv[++i], nullptr); o.cam_set = true; }
		else if (a == "--frames") { need(1); o.frames = atoi(argv[++i]); }
		else if (a == "--async") o.pipelined = true;
		else if (a == "--rgba8") o.rgba8 = true;
		else if (a == "--glsl") o.glsl = true;
		else if (a == "--lbvh") o.accel = RT_ACCEL_LBVH;
		else if (a == "--brute") o.accel = RT_ACCEL_BRUTE;
		else if (a == "--device") { need(1); o.device = atoi(argv[++i]); }
		else if (a == "--txt") { need(1); o.txt = argv[++i]; }
		else if (a == "--ppm") { need(1); o.ppm = argv[++i]; }
		else if (a == "--raw") { need(1); o.raw = argv[++i]; }
		else if (a == "--rto") { need(1); o.rto = argv[++i]; }
		else if (a == "--walk") { need(1); o.walk = argv[++i]; }
		else { fprintf(stderr, "rt_headless: unknown option %s (see the header of rt_headless.cpp)\n", a.c_str()); return 2; }
	}
	auto bad_size = [](int w, int h) { return w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL / 3; };
	if (bad_size(o.width, o.height)) { fprintf(stderr, "rt_headless: frame size %dx%d is not usable (positive, width*height*3 < 2^31)\n", o.width, o.height); return 2; }
	rt_scene* s = rt_scene_create();
	if (!s) return fail("rt_scene_create");
	int rc = (o.scene == "stock") ? build_stock(s) : build_from_file(s, o);
	if (rc != 0) { fail("scene"); rt_scene_destroy(s); return 1; }
	if (bad_size(o.width, o.height)) {      // (a `size` line of the scene file)
		fprintf(stderr, "rt_headless: frame size %dx%d is not usable (positive, width*height*3 < 2^31)\n", o.width, o.height);
		rt_scene_destroy(s);
		return 2;
	}
	if (!o.rto.empty() && rt_scene_load_rto(s, o.rto.c_str()) < 0) { fail("loadModel"); rt_scene_destroy(s); return 1; }
	if (rt_scene_commit(s, o.device, o.accel) != RT_OK) { fail("rt_scene_commit"); rt_scene_destroy(s); return 1; }

	// key script (optional): one entry per frame
	std::vector<unsigned int> keys;
	std::vector<float> key_ms;
	if (!o.walk.empty()) {
		FILE* f = fopen(o.walk.c_str(), "r");
		if (!f) { fprintf(stderr, "rt_headless: cannot open %s\n", o.walk.c_str()); rt_scene_destroy(s); return 1; }
		char line[512];
		while (fgets(line, sizeof line, f)) {
			float ms;
			int k[10];
			if (sscanf(line, "frame %f %d %d %d %d %d %d %d %d %d %d", &ms, &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &k[7], &k[8], &k[9]) != 11) continue;
			const unsigned int bit[10] = { RT_KEY_LOOK_LEFT, RT_KEY_LOOK_RIGHT, RT_KEY_LOOK_UP, RT_KEY_LOOK_DOWN, RT_KEY_LEFT, RT_KEY_RIGHT,
			                               RT_KEY_UP, RT_KEY_DOWN, RT_KEY_FORWARD, RT_KEY_BACKWARD };
			unsigned int m = 0;
			for (int j = 0; j < 10; j++) if (k[j]) m |= bit[j];
			keys.push_back(m);
			key_ms.push_back(ms);
		}
		fclose(f);
		if (keys.empty()) { fprintf(stderr, "rt_headless: %s holds no `frame` line\n", o.walk.c_str()); rt_scene_destroy(s); return 1; }
		o.frames = (int)keys.size();
	}

	rt_camera camera;
	rt_camera_init(&camera, o.cam, o.cam + 3);                          // Source.cpp:810
	float trans[16];
	const size_t n_floats = (size_t)o.width * o.height * (o.rgba8 ? 1 : 3);      // RT_OUT_RGBA8: one 4-byte pixel instead of three floats
	const int flags = RT_RENDER_STRICT | (o.rgba8 ? RT_OUT_RGBA8 : 0) | (o.glsl ? RT_SHADE_GLSL : 0);
	std::vector<float> data;                                            // Source.cpp:1035
	float* pinned[2] = { nullptr, nullptr };
	try {
		data.resize(n_floats);
	} catch (...) { fprintf(stderr, "rt_headless: out of memory for a %dx%d frame\n", o.width, o.height); rt_scene_destroy(s); return 1; }
	if (o.pipelined) {
		pinned[0] = rt_frame_alloc(n_floats);
		pinned[1] = rt_frame_alloc(n_floats);
		if (!pinned[0] || !pinned[1]) { fail("rt_frame_alloc"); rt_frame_free(pinned[0]); rt_frame_free(pinned[1]); rt_scene_destroy(s); return 1; }
	}
	int ticket[2] = { -1, -1 };
	float build_ms = 0;
	rt_scene_build_ms(s, o.device, &build_ms);
	printf("triangles %d lights %d accel %s build %.3f ms\n", rt_scene_triangle_count(s), rt_scene_light_count(s),
	       rt_scene_accel(s, o.device) == RT_ACCEL_LBVH ? "lbvh" : "brute", build_ms);
	for (int f = 0; f < o.frames; f++) {
		auto t0 = std::chrono::steady_clock::now();                     // Source.cpp:1129
		rt_camera_trans(camera.rotations, trans);                       // Source.cpp:1145-1147
		if (o.pipelined) {
			// frame f renders and copies while the caller still holds frame f-1 (what a viewer would be uploading)
			ticket[f & 1] = rt_render_async(s, camera.position, trans, o.width, o.height, o.depth, flags, pinned[f & 1]);
			if (ticket[f & 1] < 0) { fail("rt_render_async"); rt_scene_destroy(s); return 1; }
			if (f > 0 && rt_render_wait(s, ticket[(f - 1) & 1]) != RT_OK) { fail("rt_render_wait"); rt_scene_destroy(s); return 1; }
		} else if (rt_render(s, camera.position, trans, o.width, o.height, o.depth, flags, data.data()) != RT_OK) { fail("rt_render"); rt_scene_destroy(s); return 1; }
		if (!keys.empty()) rt_camera_step(&camera, keys[f], key_ms[f]); // Source.cpp:1294-1295, 1307-1308 (scripted frame time)
		auto t1 = std::chrono::steady_clock::now();
		double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
		printf("%f ms %f FPS\n", ms, 1000.0 / ms);                      // Source.cpp:1301
	}
	if (o.pipelined && o.frames > 0) {
		const int last = (o.frames - 1) & 1;
		if (rt_render_wait(s, ticket[last]) != RT_OK) { fail("rt_render_wait"); rt_scene_destroy(s); return 1; }
		memcpy(data.data(), pinned[last], n_floats * sizeof(float));
	}
	rt_frame_free(pinned[0]);
	rt_frame_free(pinned[1]);
	if (!keys.empty())
		printf("camera after the walk: position %.9g %.9g %.9g rotations %.9g %.9g %.9g\n", camera.position[0], camera.position[1],
		       camera.position[2], camera.rotations[0], camera.rotations[1], camera.rotations[2]);
	bool ok = true;
	if (!o.txt.empty()) ok = write_image(o.txt, data, o.width, o.height, true, o.rgba8) && ok;
	if (!o.ppm.empty()) ok = write_image(o.ppm, data, o.width, o.height, false, o.rgba8) && ok;
	if (!o.raw.empty()) {
		FILE* f = fopen(o.raw.c_str(), "wb");
		ok = f && fwrite(data.data(), sizeof(float), data.size(), f) == data.size() && ok;
		if (f) fclose(f);
	}
	rt_scene_destroy(s);
	if (!ok) { fprintf(stderr, "rt_headless: could not write an output file\n"); return 1; }
	return 0;
}
