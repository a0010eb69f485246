"""Brute force vs LBVH over scene size: us/frame at 800x800 (16 frames per launch, one pose)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import api, scenes
W = H = 800
NF = 32
out = torch.empty((NF, H, W, 3), dtype=torch.float32, device="cuda:0")
for (ns, sub) in [tuple(map(int, a.split(","))) for a in sys.argv[1:]] or [(2, 1), (4, 1), (6, 1), (8, 1), (12, 1), (6, 2), (20, 1), (10, 2), (20, 2), (20, 3)]:
    desc = scenes.spheres(ns, sub, W, H)
    cams = np.tile(np.asarray(desc.cam_pos, np.float32), (NF, 1))
    trs = np.tile(api.camera_trans(desc.cam_rot).astype(np.float32).reshape(1, 16), (NF, 1))
    row = []
    ref = None
    for accel in (api.ACCEL_BRUTE, api.ACCEL_LBVH):
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, accel)
            n = sc.n_triangles
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(4):
                    sc.render_frames_device(0, cams, trs, W, H, desc.depth, api.STRICT, out.data_ptr())
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (4 * NF)
            img = out[0].cpu().numpy().view(np.uint32)
            if ref is None: ref = img
            same = bool((img == ref).all())
            row.append("%s %.1f us%s" % ("brute" if accel == api.ACCEL_BRUTE else "lbvh", dt * 1e6, "" if same else " MISMATCH"))
    print(n, "triangles:", " | ".join(row), flush=True)
