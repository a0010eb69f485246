"""Per-pixel overhead of the brute-force kernel: frames of scenes with almost no geometry."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import api, scenes
W = H = 800
NF = 16
out = torch.empty((NF, H, W, 3), dtype=torch.float32, device="cuda:0")
def run(name, build):
    with api.Scene() as sc:
        build(sc)
        sc.commit(0, api.ACCEL_BRUTE)
        cam = api.Camera((-5, 5, 5), (135, -45, 0))
        cams = np.tile(np.asarray(cam.position, np.float32), (NF, 1))
        trs = np.tile(np.asarray(cam.trans(), np.float32).reshape(1, 16), (NF, 1))
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8):
                sc.render_frames_device(0, cams, trs, W, H, 1, api.STRICT, out.data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (8 * NF)
        print("%-28s %d triangles  %.2f us/frame" % (name, sc.n_triangles, dt * 1e6), flush=True)
def behind(sc):
    sc.Triangle((100, 100, 100), (101, 100, 100), (100, 101, 100), (255, 0, 0), 0); sc.addLight((0, 8, 0), 4096)
def floor(sc):
    sc.addTriangle((0, -10, 0), (90, 0, 0), (10, 10, 10), (255, 255, 255), 0); sc.addLight((0, 8, 0), 4096)
def floor_nolight(sc):
    sc.addTriangle((0, -10, 0), (90, 0, 0), (10, 10, 10), (255, 255, 255), 0)
run("all rays miss", behind)
run("floor quad, no light", floor_nolight)
run("floor quad, one light", floor)
