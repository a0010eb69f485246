"""Single-frame launches of the 102k-triangle scene, 800x800: us/frame by shadow grid on/off, traversal, build; one stream."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from raytracing_amd import api, scenes

desc = scenes.spheres(20, 4, int(sys.argv[1]) if len(sys.argv) > 2 else 800, int(sys.argv[2]) if len(sys.argv) > 2 else 800)
cams, trs = api.camera_sweep(64, desc.cam_pos, desc.cam_rot)
W, H, D = desc.width, desc.height, desc.depth
n = 400 if W * H < 2000000 else 100
buf = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
s0 = torch.cuda.current_stream()


def run(sc, label):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for timed in (False, True):
        torch.cuda.synchronize()
        if timed:
            ev0.record(s0)
        for i in range(n):
            sc.render_device(0, cams[i % 64], trs[i % 64], W, H, D, api.STRICT, buf.data_ptr(), s0.cuda_stream)
        if timed:
            ev1.record(s0)
    torch.cuda.synchronize()
    print("%-40s %.1f us/frame" % (label, ev0.elapsed_time(ev1) / n * 1e3), flush=True)


with api.Scene.from_desc(desc) as sc:
    for grid in (True,):
        sc.set_shadow_grid(grid)
        sc.commit(0, api.ACCEL_LBVH)
        for trav, tn in ((api.TRAVERSAL_PACKET, "packet"), (api.TRAVERSAL_PER_LANE, "per lane")):
            sc.set_traversal(trav)
            for tpw in (0,):
                sc.set_tiles_per_wave(tpw)
                run(sc, "grid %s %s tpw %d" % (grid, tn, tpw))
