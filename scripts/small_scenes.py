"""Brute force vs the tree (both builds, both traversals) on the small BASELINE scenes: us/frame, 32 frames per launch."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import api, scenes
NF = 32
for name, desc in [("room800", scenes.room(800, 800)), ("stock800", scenes.stock(800, 800)), ("mirrors4k", scenes.mirrors(3840, 2160, 16))]:
    W, H = desc.width, desc.height
    out = torch.empty((NF, H, W, 3), dtype=torch.float32, device="cuda:0")
    cams, trs = api.camera_sweep(NF, desc.cam_pos, desc.cam_rot)
    row = []; ref = None
    for accel, build, trav, label in ((api.ACCEL_BRUTE, api.BUILD_SAH, api.TRAVERSAL_AUTO, "brute"),
                                      (api.ACCEL_LBVH, api.BUILD_SAH, api.TRAVERSAL_PACKET, "sah packets"), (api.ACCEL_LBVH, api.BUILD_SAH, api.TRAVERSAL_PER_LANE, "sah per lane"),
                                      (api.ACCEL_LBVH, api.BUILD_FAST, api.TRAVERSAL_PACKET, "morton packets")):
        with api.Scene.from_desc(desc) as sc:
            sc.set_build(build); sc.set_traversal(trav)
            sc.commit(0, accel)
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(4):
                    sc.render_frames_device(0, cams, trs, W, H, desc.depth, api.STRICT, out.data_ptr())
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (4 * NF)
            img = out[3].cpu().numpy().view(np.uint32)
            if ref is None: ref = img
            row.append("%s %.1f us%s" % (label, dt * 1e6, "" if (img == ref).all() else " MISMATCH"))
            n = sc.n_triangles
    print(name, n, "triangles:", " | ".join(row), flush=True)
    del out
