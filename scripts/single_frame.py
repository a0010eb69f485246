"""One frame per launch, back to back (the reference's call pattern, Source.cpp:1126-1310): us per frame on one stream and
alternating between two streams; run under rocprofv3 --kernel-trace to see the kernel's own duration."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from raytracing_amd import api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "room"
desc = scenes.spheres(20, 4, 800, 800) if name == "spheres" else scenes.get(name).with_size(800, 800)
accel = api.ACCEL_LBVH if name == "spheres" else api.ACCEL_BRUTE
cams, trs = api.camera_sweep(64, desc.cam_pos, desc.cam_rot)
W, H, D = desc.width, desc.height, desc.depth
n = 2000
with api.Scene.from_desc(desc) as sc:
    sc.commit(0, accel)
    bufs = [torch.empty(H * W * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
    s0 = torch.cuda.current_stream()
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    for label, streams in (("one stream", [s0, s0]), ("two streams", side)) * 3:
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
        print("%s %s: %.1f us/frame" % (name, label, ev0.elapsed_time(ev1) / n * 1e3), flush=True)
