"""One-off differential fuzz: brute force vs LBVH per lane vs LBVH packets on random soups (frames must agree bit for bit)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from raytracing_amd import api
from test_gpu_parity import _soup
bad = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(1000, 1000 + n_seeds):
    rng = np.random.default_rng(seed)
    desc = _soup(seed, int(rng.integers(2, 1500)), float(10 ** rng.uniform(-3, 4)), w=int(rng.integers(1, 160)), h=int(rng.integers(1, 120)),
                 depth=int(rng.integers(0, 8)), lights=int(rng.integers(0, 5)), needles=bool(seed % 3 == 0))
    frames = []
    with api.Scene.from_desc(desc) as sc:
        tr = api.camera_trans(desc.cam_rot)
        sc.commit(0, api.ACCEL_BRUTE)
        frames.append(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth))
        sc.commit(0, api.ACCEL_LBVH)
        for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            api.set_traversal(mode)
            frames.append(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth))
        api.set_traversal(api.TRAVERSAL_AUTO)
    a = frames[0].view(np.uint32)
    for k, f in enumerate(frames[1:]):
        if not np.array_equal(a, f.view(np.uint32)):
            bad += 1
            print("seed", seed, "path", k + 1, "differs in", int((a != f.view(np.uint32)).any(axis=2).sum()), "pixels", flush=True)
    if seed % 50 == 0:
        print("seed", seed, "ok so far, mismatches", bad, flush=True)
print("done:", n_seeds, "scenes,", bad, "mismatches")
sys.exit(1 if bad else 0)
