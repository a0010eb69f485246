"""LBVH at 1M triangles: build time, frame time, and castRay parity against brute force."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import api, scenes
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p1, p2, p3, col, sh = scenes.spheres_arrays(ns, 4)
W = H = 800
with api.Scene() as sc:
    room = scenes.stock(8, 8)
    for op in room.ops[:12]:
        if op[0] == "quad": pass
    sc.addTriangles(p1, p2, p3, col, sh)
    sc.addLight((0, 8, 0), 4096)
    print("triangles", sc.n_triangles, flush=True)
    t0 = time.time(); sc.commit(0, api.ACCEL_LBVH); print("commit %.3f s, build %.3f ms" % (time.time() - t0, sc.build_ms(0)))
    t0 = time.time(); sc.commit(0, api.ACCEL_LBVH); print("re-commit %.3f s, build %.3f ms" % (time.time() - t0, sc.build_ms(0)), flush=True)
    cam = api.Camera((-9, 9, 9), (135, -35, 0))
    NF = 8
    out = torch.empty((NF, H, W, 3), dtype=torch.float32, device="cuda:0")
    cams = np.tile(np.asarray(cam.position, np.float32), (NF, 1)); trs = np.tile(cam.trans().reshape(1, 16), (NF, 1))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc.render_frames_device(0, cams, trs, W, H, 1, api.STRICT, out.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / NF
    print("LBVH %.1f us/frame at %dx%d, lit pixels %d" % (dt * 1e6, W, H, int((out[0].sum(dim=2) > 0).sum())), flush=True)
    rng = np.random.default_rng(1)
    rays = np.zeros((4096, 7), np.float32)
    rays[:, :3] = rng.uniform(-9, 9, (4096, 3)); rays[:, 3:6] = rng.normal(size=(4096, 3)); rays[:, 6] = -1
    a = sc.cast_rays(rays)
    sc.commit(0, api.ACCEL_BRUTE)
    t0 = time.time(); b = sc.cast_rays(rays); print("brute castRay of 4096 rays: %.2f s" % (time.time() - t0))
    print("castRay LBVH == brute:", bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), "hits", int(a[:, 0].sum()))
