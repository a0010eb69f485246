"""us per launch of rt_render_tiles_device for shard 0 of 2 (interleaved 16-row tiles) of a 7680x4320 frame of the 102k-triangle scene:
the kernel instance multi-GPU ranks run (render_lockstep_kernel<3, false>)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from raytracing_amd import api, scenes

desc = scenes.spheres(20, 4, 7680, 4320)
cams, trs = api.camera_sweep(64, desc.cam_pos, desc.cam_rot)
W, H, D = desc.width, desc.height, desc.depth
rows = api.shard_rows(H, 16, 0, 2)
buf = torch.empty(rows * W * 3, dtype=torch.float32, device="cuda")
s0 = torch.cuda.current_stream()
with api.Scene.from_desc(desc) as sc:
    sc.commit(0, api.ACCEL_LBVH)
    for rep in range(3):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for timed in (False, True):
            torch.cuda.synchronize()
            if timed:
                ev0.record(s0)
            for i in range(64):
                sc.render_tiles_device(0, cams[i], trs[i], W, H, D, api.STRICT, 16, 0, 2, buf.data_ptr(), s0.cuda_stream)
            if timed:
                ev1.record(s0)
        torch.cuda.synchronize()
        print("shard 0 of 2, 7680x4320: %.1f us per launch" % (ev0.elapsed_time(ev1) / 64 * 1e3), flush=True)
