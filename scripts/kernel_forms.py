"""us/frame of the two forms of the brute-force render kernel (rt_scene_set_kernel) on scenes whose bounce depth
diverges inside an 8x8 tile: random soups with a quarter of the triangles mirrors, depth 8 and 16."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import api, scenes
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_parity import _soup

def bench(desc, kind, frames=16, reps=5):
    with api.Scene.from_desc(desc) as sc:
        sc.set_kernel(kind)
        sc.commit(0, api.ACCEL_BRUTE)
        cams, trs = api.camera_sweep(frames, desc.cam_pos, desc.cam_rot)
        n = desc.width * desc.height * 3
        buf = torch.empty(frames * n, dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        sc.render_frames_device(0, cams, trs, desc.width, desc.height, desc.depth, api.STRICT, buf.data_ptr(), n, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            sc.render_frames_device(0, cams, trs, desc.width, desc.height, desc.depth, api.STRICT, buf.data_ptr(), n, st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * frames) * 1e6

for seed, n, depth in [(1, 40, 8), (3, 120, 16), (7, 100, 8), (9, 24, 16)]:
    desc = _soup(seed, n, 10.0, w=800, h=800, depth=depth, lights=2, cam=((0.0, 0.0, 14.0), (180.0, 0.0, 0.0)))
    a = bench(desc, api.KERNEL_STATE_MACHINE)
    b = bench(desc, api.KERNEL_LOCKSTEP)
    print("soup seed %d, %d triangles, depth %d: state machine %.1f us/frame, lockstep %.1f" % (seed, n, depth, a, b))
# the facing mirrors under a checker of matte patches: neighbouring pixels stop after different numbers of bounces
for n, size, depth in ((16, (800, 800), 8), (24, (1600, 900), 16), (32, (1920, 1080), 16)):
    d = scenes.mirror_checker(size[0], size[1], depth, n)
    print("mirror_checker n=%d %dx%d depth %d (%d triangles): state machine %.1f us/frame, lockstep %.1f" % (
        n, size[0], size[1], depth, d.n_triangles, bench(d, api.KERNEL_STATE_MACHINE), bench(d, api.KERNEL_LOCKSTEP)))
for d in (scenes.mirrors(1920, 1080, 16), scenes.mirrors(3840, 2160, 16), scenes.room(800, 800)):
    print(d.name, "state machine %.1f, lockstep %.1f" % (bench(d, api.KERNEL_STATE_MACHINE), bench(d, api.KERNEL_LOCKSTEP)))
