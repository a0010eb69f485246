// ref_driver_tail.cpp — TEST INFRASTRUCTURE, not product code.
//
// This file is never compiled on its own.  oracle/Makefile streams the reference's
// translation unit (/root/reference/RayTracing/src/Source.cpp, read where it lies, with the one
// MSVC-only token `Triangle& const t` fixed by sed in the pipe) into g++ and appends this file
// to the same stream, so everything below sees the reference's own `Triangle`, `Light`,
// `Camera`, `triangles`, `visibleTriangles`, `lights`, `addTriangle`, `castRay` and
// `getPixelData` (Source.cpp:31-73, 95-212, 277-405, 410-439, 530-628).  No reference source is
// copied into the repository and nothing stands in for GLFW: the reference's `main` (renamed
// `reference_main` by -Dmain=…) is simply never called, and the GLFW symbols it alone needs are
// left unresolved by the linker (--unresolved-symbols=ignore-all).
//
// What the driver does: it reads a scene description (format in oracle/SCENE_FORMAT.md), turns
// every line into a call of the reference's OWN entry points, renders with the reference's OWN
// getPixelData exactly as the frame loop does (Source.cpp:1145-1163, serial form 1248-1252), and
// dumps raw little-endian float32 files that become tests/golden/ fixtures.
//
//   ref_render render <scene.txt> <out_prefix> [repeat]
//        -> <out_prefix>.tris.f32   N x 28 floats (p,n,p1,p2,p3,edge0,edge1,edge2,color,shininess)
//           <out_prefix>.trans.f32  16 floats, column-major (glm::value_ptr order)
//           <out_prefix>.frame.f32  H*W*3 floats, data[(y*W+x)*3+c]
//           stdout: one JSON line with N, W, H, ms per frame (min over `repeat` frames)
//   ref_render rays <scene.txt> <rays.f32> <out.f32>
//        rays.f32: K x 7 floats (origin xyz, dir xyz, skip index as float, <0 = NULL)
//        out.f32 : K x 6 floats (didHit, index of hit triangle or -1, distance, point xyz)
//
//   ref_render camera <script.txt> <out.f32>
//        script: `start px py pz rx ry rz`, then one line per frame `frame ms ll lr lu ld l r u d f b`
//        (frame time in ms as the loop measures it, Source.cpp:1300; the look and move key flags of
//        Source.cpp:237-241).  Per frame the reference's own Camera does what the frame loop does
//        after drawing: rotate, translate (Source.cpp:1294-1295), then the speeds for the next
//        frame (Source.cpp:1307-1308).
//        out.f32: per frame 33 floats = position, rotations, direction, right, up, moveSpeed,
//        rotSpeed after the update, and the `trans` the NEXT frame is drawn with (Source.cpp:1145-1147)
//   ref_render rto <scene.txt> <model.rto> <out_prefix>
//        as `render`, but after the scene file's own lines the reference's OWN loadModel
//        (Source.cpp:631-657) appends the records of <model.rto> (e.g. the reference's sample
//        RayTracing/src/cube.rto) to `triangles`; same three output files.
//
// When built with -DORACLE_DEPTH_PARAM the Makefile's sed additionally rewrites the literal
// `int maxBounces = 1;` (Source.cpp:567) to read g_oracle_max_bounces, set by a `depth` line of
// the scene file; the plain build keeps the literal and rejects depth != 1.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>

#ifdef ORACLE_DEPTH_PARAM
// declared in oracle/ref_prelude.h (force-included before the reference TU)
#else
static int g_oracle_max_bounces = 1;
#endif

namespace oracle_driver {

struct Scene {
	int width = 100, height = 100;
	glm::vec3 camPos{ -5, 5, 5 };
	glm::vec3 camRot{ 135, -45, 0 };
	int depth = 1;
};

static bool parseScene(const char* path, Scene& sc) {
	FILE* f = fopen(path, "r");
	if (!f) { fprintf(stderr, "ref_render: cannot open %s: %s\n", path, strerror(errno)); return false; }
	triangles.clear();
	lights.clear();
	char line[1024];
	int lineno = 0;
	while (fgets(line, sizeof line, f)) {
		lineno++;
		char* s = line;
		while (*s == ' ' || *s == '\t') s++;
		if (*s == '#' || *s == '\n' || *s == 0) continue;
		char kw[32];
		int off = 0;
		if (sscanf(s, "%31s%n", kw, &off) != 1) continue;
		s += off;
		float v[16];
		int n = 0;
		while (n < 16) {
			char* end;
			float x = strtof(s, &end);
			if (end == s) break;
			v[n++] = x;
			s = end;
		}
		if (!strcmp(kw, "quadhalf") && n == 13) {
			// the reference's own scene entry point, Source.cpp:410
			addTriangle({ v[0], v[1], v[2] }, { v[3], v[4], v[5] }, { v[6], v[7], v[8] }, { v[9], v[10], v[11] }, v[12]);
		} else if (!strcmp(kw, "tri") && n == 13) {
			// the reference's own Triangle constructor, Source.cpp:45
			triangles.push_back(Triangle({ v[0], v[1], v[2] }, { v[3], v[4], v[5] }, { v[6], v[7], v[8] }, { v[9], v[10], v[11] }, v[12]));
		} else if (!strcmp(kw, "light") && n == 4) {
			lights.push_back({ { v[0], v[1], v[2] }, v[3] });   // Source.cpp:880
		} else if (!strcmp(kw, "camera") && n == 6) {
			sc.camPos = { v[0], v[1], v[2] };
			sc.camRot = { v[3], v[4], v[5] };
		} else if (!strcmp(kw, "size") && n == 2) {
			sc.width = (int)v[0];
			sc.height = (int)v[1];
		} else if (!strcmp(kw, "depth") && n == 1) {
			sc.depth = (int)v[0];
		} else {
			fprintf(stderr, "ref_render: %s:%d: bad line (%s, %d numbers)\n", path, lineno, kw, n);
			fclose(f);
			return false;
		}
	}
	fclose(f);
	return true;
}

static bool dump(const char* prefix, const char* suffix, const float* p, size_t n) {
	char path[2048];
	snprintf(path, sizeof path, "%s%s", prefix, suffix);
	FILE* f = fopen(path, "wb");
	if (!f) { fprintf(stderr, "ref_render: cannot write %s\n", path); return false; }
	size_t w = fwrite(p, sizeof(float), n, f);
	fclose(f);
	return w == n;
}

static void flattenTriangles(std::vector<float>& out) {
	for (Triangle& t : triangles) {
		const glm::vec3* fields[9] = { &t.p, &t.n, &t.p1, &t.p2, &t.p3, &t.edge0, &t.edge1, &t.edge2, &t.color };
		for (const glm::vec3* v : fields) { out.push_back(v->x); out.push_back(v->y); out.push_back(v->z); }
		out.push_back(t.shininess);
	}
}

static void loadVisible() {
	// Source.cpp:1153-1163: the per-frame copy castRay iterates
	visibleTriangles.clear();
	for (Triangle& t : triangles) visibleTriangles.push_back(t);
}

static int doRender(const char* scenePath, const char* prefix, int repeat, const char* rtoPath = NULL) {
	Scene sc;
	if (!parseScene(scenePath, sc)) return 2;
	if (rtoPath) {
		FILE* probe = fopen(rtoPath, "rb");       // loadModel calls exit(1) on a missing file (Source.cpp:634-637)
		if (!probe) { fprintf(stderr, "ref_render: cannot open %s\n", rtoPath); return 2; }
		fclose(probe);
		loadModel(rtoPath);                       // Source.cpp:631
	}
#ifndef ORACLE_DEPTH_PARAM
	if (sc.depth != 1) { fprintf(stderr, "ref_render: this build keeps the reference's literal maxBounces = 1\n"); return 3; }
#endif
	g_oracle_max_bounces = sc.depth;

	Camera camera(sc.camPos, sc.camRot);                       // Source.cpp:810
	glm::mat4 trans = glm::mat4(1.0f);                         // Source.cpp:1145-1147
	trans = glm::rotate(trans, glm::radians((float)camera.rotations.x), glm::vec3(0.0f, 1.0f, 0.0f));
	trans = glm::rotate(trans, glm::radians(-(float)camera.rotations.y), glm::vec3(1.0f, 0.0f, 0.0f));
	loadVisible();

	const int width = sc.width, height = sc.height;
	float* data = (float*)malloc(sizeof(float) * width * height * 3);   // Source.cpp:1035
	double best = 1e30;
	for (int r = 0; r < repeat; r++) {
		auto t0 = std::chrono::steady_clock::now();
		for (int y = 0; y < height; y++) {                     // Source.cpp:1248-1252
			for (int x = 0; x < width; x++) {
				getPixelData(x, y, width, height, data, trans, &camera);
			}
		}
		auto t1 = std::chrono::steady_clock::now();
		double ms = (t1 - t0).count() / 1000000.0;             // Source.cpp:1300
		if (ms < best) best = ms;
	}

	std::vector<float> flat;
	flattenTriangles(flat);
	bool ok = dump(prefix, ".tris.f32", flat.data(), flat.size());
	ok = dump(prefix, ".trans.f32", glm::value_ptr(trans), 16) && ok;
	ok = dump(prefix, ".frame.f32", data, (size_t)width * height * 3) && ok;
	printf("{\"triangles\": %zu, \"lights\": %zu, \"width\": %d, \"height\": %d, \"depth\": %d, \"ms_per_frame\": %.4f, \"sizeof_Triangle\": %zu}\n",
		triangles.size(), lights.size(), width, height, sc.depth, best, sizeof(Triangle));
	free(data);
	return ok ? 0 : 4;
}

static int doRays(const char* scenePath, const char* raysPath, const char* outPath) {
	Scene sc;
	if (!parseScene(scenePath, sc)) return 2;
	loadVisible();
	FILE* f = fopen(raysPath, "rb");
	if (!f) { fprintf(stderr, "ref_render: cannot open %s\n", raysPath); return 2; }
	std::vector<float> rays;
	float rec[7];
	while (fread(rec, sizeof(float), 7, f) == 7) rays.insert(rays.end(), rec, rec + 7);
	fclose(f);
	std::vector<float> out;
	for (size_t i = 0; i + 7 <= rays.size(); i += 7) {
		int skip = (int)rays[i + 6];
		Triangle* curr = (skip >= 0 && (size_t)skip < visibleTriangles.size()) ? &visibleTriangles[skip] : NULL;
		PayLoad p = castRay({ rays[i], rays[i + 1], rays[i + 2] }, { rays[i + 3], rays[i + 4], rays[i + 5] }, curr);   // Source.cpp:277
		out.push_back(p.didHit ? 1.0f : 0.0f);
		out.push_back(p.didHit ? (float)(p.cur - &visibleTriangles[0]) : -1.0f);
		out.push_back(p.distance);
		out.push_back(p.point.x); out.push_back(p.point.y); out.push_back(p.point.z);
	}
	return dump(outPath, "", out.data(), out.size()) ? 0 : 4;
}

static int doCamera(const char* scriptPath, const char* outPath) {
	FILE* f = fopen(scriptPath, "r");
	if (!f) { fprintf(stderr, "ref_render: cannot open %s\n", scriptPath); return 2; }
	Camera camera({ 0, 0, 0 }, { 0, 0, 0 });
	bool started = false;
	std::vector<float> out;
	char line[1024];
	while (fgets(line, sizeof line, f)) {
		float v[11];
		if (sscanf(line, "start %f %f %f %f %f %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6) {
			camera = Camera({ v[0], v[1], v[2] }, { v[3], v[4], v[5] });            // Source.cpp:810
			started = true;
		} else if (sscanf(line, "frame %f %f %f %f %f %f %f %f %f %f %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &v[8],
		                  &v[9], &v[10]) == 11 && started) {
			float milis = v[0];                                                     // Source.cpp:1075, 1300
			camera.rotate(v[1] != 0, v[2] != 0, v[3] != 0, v[4] != 0);             // Source.cpp:1294
			camera.translate(v[5] != 0, v[6] != 0, v[7] != 0, v[8] != 0, v[9] != 0, v[10] != 0);   // Source.cpp:1295
			camera.moveSpeed = camera.moveBaseSpeed * milis;                        // Source.cpp:1307
			camera.rotSpeed = camera.rotBaseSpeed * milis;                          // Source.cpp:1308
			glm::mat4 trans = glm::mat4(1.0f);                                      // Source.cpp:1145-1147
			trans = glm::rotate(trans, glm::radians((float)camera.rotations.x), glm::vec3(0.0f, 1.0f, 0.0f));
			trans = glm::rotate(trans, glm::radians(-(float)camera.rotations.y), glm::vec3(1.0f, 0.0f, 0.0f));
			const glm::vec3* fields[5] = { &camera.position, &camera.rotations, &camera.direction, &camera.right, &camera.up };
			for (const glm::vec3* p : fields) { out.push_back(p->x); out.push_back(p->y); out.push_back(p->z); }
			out.push_back(camera.moveSpeed);
			out.push_back(camera.rotSpeed);
			for (int k = 0; k < 16; k++) out.push_back(glm::value_ptr(trans)[k]);
		}
	}
	fclose(f);
	return dump(outPath, "", out.data(), out.size()) ? 0 : 4;
}

} // namespace oracle_driver

#undef main
int main(int argc, char** argv) {
	if (argc >= 4 && !strcmp(argv[1], "render")) {
		int repeat = argc >= 5 ? atoi(argv[4]) : 1;
		return oracle_driver::doRender(argv[2], argv[3], repeat < 1 ? 1 : repeat);
	}
	if (argc >= 5 && !strcmp(argv[1], "rto")) {
		return oracle_driver::doRender(argv[2], argv[4], 1, argv[3]);
	}
	if (argc >= 5 && !strcmp(argv[1], "rays")) {
		return oracle_driver::doRays(argv[2], argv[3], argv[4]);
	}
	if (argc >= 4 && !strcmp(argv[1], "camera")) {
		return oracle_driver::doCamera(argv[2], argv[3]);
	}
	fprintf(stderr, "usage: ref_render render <scene.txt> <out_prefix> [repeat]\n       ref_render rays <scene.txt> <rays.f32> <out.f32>\n       ref_render rto <scene.txt> <model.rto> <out_prefix>\n");
	return 1;
}
