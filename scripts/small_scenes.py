"""Brute force vs LBVH (auto traversal) on the small BASELINE scenes."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import api, scenes
NF = 32
for name, desc in [("room800", scenes.room(800, 800)), ("stock800", scenes.stock(800, 800)), ("mirrors4k", scenes.mirrors(3840, 2160, 16))]:
    W, H = desc.width, desc.height
    out = torch.empty((NF, H, W, 3), dtype=torch.float32, device="cuda:0")
    poses = scenes.camera_sweep(NF, desc.cam_pos, desc.cam_rot)
    cams = np.array([p for p, _ in poses], np.float32); trs = np.stack([api.camera_trans(r) for _, r in poses]).astype(np.float32)
    row = []; ref = None
    for accel in (api.ACCEL_BRUTE, api.ACCEL_LBVH):
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, accel)
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(4):
                    sc.render_frames_device(0, cams, trs, W, H, desc.depth, api.STRICT, out.data_ptr())
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (4 * NF)
            img = out[3].cpu().numpy().view(np.uint32)
            if ref is None: ref = img
            row.append("%s %.1f us%s" % ("brute" if accel == api.ACCEL_BRUTE else "lbvh", dt * 1e6, "" if (img == ref).all() else " MISMATCH"))
    print(name, sc.n_triangles if False else "", " | ".join(row), flush=True)
    del out
