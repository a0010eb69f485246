"""How much of a frame of the 102k-triangle scene is shadow rays?  The same launches with and without the light
(no light: no shadow rays, frames black), per traversal and size — sizes the gain a per-light structure could bring."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import dataclasses
import numpy as np
import torch
from raytracing_amd import api, scenes


def timed(sc, cams, trs, w, h, depth, n=4):
    fpl = len(cams)
    buf = torch.empty(fpl * w * h * 3, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        sc.render_frames_device(0, cams, trs, w, h, depth, api.STRICT, buf.data_ptr(), w * h * 3, st)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n):
        sc.render_frames_device(0, cams, trs, w, h, depth, api.STRICT, buf.data_ptr(), w * h * 3, st)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n / fpl * 1e3


for (w, h, fpl) in ((800, 800, 32), (3840, 2160, 16)):
    base = scenes.spheres(20, 4, w, h)
    dark = dataclasses.replace(base, ops=[op for op in base.ops if op[0] != "light"])
    cams, trs = api.camera_sweep(fpl, base.cam_pos, base.cam_rot)
    for name, desc in (("lit", base), ("no light", dark)):
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, api.ACCEL_LBVH)
            for mode, mname in ((api.TRAVERSAL_PER_LANE, "per lane"), (api.TRAVERSAL_PACKET, "packets")):
                sc.set_traversal(mode)
                print("%dx%d %-9s %-9s %8.1f us/frame" % (w, h, name, mname, timed(sc, cams, trs, w, h, desc.depth)), flush=True)
