"""Launch shapes by the hundred (not collected by pytest; run by hand on a GPU box):
    python tests/fuzz_launch_shapes.py [n_cases] [first_seed]
random frame sizes (ragged tile counts), numbers of frames per launch (below and above the 8 that switch the XCD bands on),
runs of 0-7 tiles per wavefront, both traversals and brute force, float and 8-bit frames: every frame of a multi-frame launch
must be, bit for bit, the frame a single-frame launch with the launcher's own choices gives for its pose.  (The single-frame
launches are what tests/fuzz_vs_oracle.py holds against the oracle.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raytracing_amd import api, scenes

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(8, 700)), int(rng.integers(8, 500))
    nf = int(rng.choice([2, 7, 8, 9, 13, 16, 24]))
    tpw = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 7]))
    kind = int(rng.integers(0, 4))          # 0 brute, 1 per lane, 2 packets, 3 packets by the state machine
    rgba8 = bool(rng.integers(0, 2))
    desc = scenes.spheres(int(rng.integers(2, 5)), 2, w, h, depth=int(rng.integers(1, 4))) if kind else scenes.room(w, h)
    cams, trs = api.camera_sweep(nf, desc.cam_pos, desc.cam_rot)
    mode = api.STRICT | (api.OUT_RGBA8 if rgba8 else 0)
    elems = w * h * (1 if rgba8 else 3)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH if kind else api.ACCEL_BRUTE)
        if kind:
            sc.set_traversal(api.TRAVERSAL_PER_LANE if kind == 1 else api.TRAVERSAL_PACKET)
            sc.set_kernel(api.KERNEL_STATE_MACHINE if kind == 3 else api.KERNEL_AUTO)
        one = torch.empty(elems, dtype=torch.int32, device="cuda:0")
        singles = []
        sc.set_tiles_per_wave(0)
        for i in range(nf):
            sc.render_device(0, cams[i], trs[i], w, h, desc.depth, mode, one.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            singles.append(one.clone())
        sc.set_tiles_per_wave(tpw)
        buf = torch.full((nf * elems,), 0x7fc00001, dtype=torch.int32, device="cuda:0")
        sc.render_frames_device(0, cams, trs, w, h, desc.depth, mode, buf.data_ptr(), elems, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for i in range(nf):
            if not torch.equal(buf[i * elems:(i + 1) * elems], singles[i]):
                bad += 1
                print("seed", seed, "size", (w, h), "frames", nf, "tiles per wave", tpw, "kind", kind, "rgba8", rgba8, "frame", i, "differs", flush=True)
                break
    if (seed - first) % 20 == 0:
        print("seed", seed, "ok so far, mismatches", bad, flush=True)
print("done:", n_cases, "launch shapes, ", bad, "mismatches")
sys.exit(1 if bad else 0)
