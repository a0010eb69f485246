"""Lone 800x800 frames of a brute-force scene, back to back, against tiles per wavefront: us/frame on one stream and on two."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from raytracing_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "room"
desc = scenes.get(name).with_size(800, 800)
cams, trs = api.camera_sweep(64, desc.cam_pos, desc.cam_rot)
W, H, D = desc.width, desc.height, desc.depth
n = 1500
with api.Scene.from_desc(desc) as sc:
    sc.commit(0, api.ACCEL_BRUTE)
    bufs = [torch.empty(H * W * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
    s0 = torch.cuda.current_stream()
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    for tpw in (0, 1, 2, 3, 4, 6):
        sc.set_tiles_per_wave(tpw)
        out = []
        for label, streams in (("one stream", [s0, s0]), ("two streams", side)):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for timed in (False, True):
                torch.cuda.synchronize()
                if timed:
                    ev0.record(s0)
                    for st in side:
                        st.wait_stream(s0)
                for i in range(n):
                    sc.render_device(0, cams[i % 64], trs[i % 64], W, H, D, api.STRICT, bufs[i & 1].data_ptr(), streams[i & 1].cuda_stream)
                if timed:
                    for st in side:
                        s0.wait_stream(st)
                    ev1.record(s0)
            torch.cuda.synchronize()
            out.append("%s %.1f" % (label, ev0.elapsed_time(ev1) / n * 1e3))
        print(name, "tiles per wave", tpw, "us/frame:", ", ".join(out), flush=True)
