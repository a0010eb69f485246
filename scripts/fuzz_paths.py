"""One-off differential fuzz: brute force vs LBVH per lane vs LBVH packets on random soups (frames must agree bit for bit)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from raytracing_amd import api
from test_gpu_parity import _soup
from raytracing_amd import scenes


def _walls_and_crumbs(seed, n, extent, big, **kw):
    """Many small triangles and a few scene-sized ones: the scenes lbvh_build hoists triangles from."""
    rng = np.random.default_rng(seed)
    sc = scenes.SceneDesc("crumbs%d" % seed, width=kw["w"], height=kw["h"], depth=kw["depth"])
    order = rng.permutation(n + big)
    for i in order:
        c = rng.uniform(-extent, extent, 3)
        size = extent * (rng.uniform(0.7, 3.0) if i < big else 10.0 ** rng.uniform(-3.0, -1.3))
        a, b, d = (c + rng.normal(size=3) * size for _ in range(3))
        sc.tri(a, b, d, rng.integers(0, 256, 3), 1.0 if rng.random() < 0.25 else 0.0)
    for _ in range(kw["lights"]):
        sc.light(rng.uniform(-extent, extent, 3), float(10.0 ** rng.uniform(1, 4)))
    sc.cam_pos, sc.cam_rot = tuple(map(float, rng.uniform(-extent, extent, 3))), tuple(map(float, rng.uniform(-180, 180, 3)))
    return sc


bad = 0
hoisted_scenes = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for seed in range(1000, 1000 + n_seeds):
    rng = np.random.default_rng(seed)
    kw = dict(w=int(rng.integers(1, 160)), h=int(rng.integers(1, 120)), depth=int(rng.integers(0, 8)), lights=int(rng.integers(0, 5)))
    if seed % 2:
        desc = _walls_and_crumbs(seed, int(rng.integers(2, 1500)), float(10 ** rng.uniform(-3, 4)), int(rng.integers(1, 40)), **kw)
    else:
        desc = _soup(seed, int(rng.integers(2, 1500)), float(10 ** rng.uniform(-3, 4)), needles=bool(seed % 3 == 0), **kw)
    frames = []
    with api.Scene.from_desc(desc) as sc:
        tr = api.camera_trans(desc.cam_rot)
        sc.commit(0, api.ACCEL_BRUTE)
        frames.append(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth))
        sc.commit(0, api.ACCEL_LBVH)
        hoisted_scenes += sc.hoisted(0).size > 0
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
print("done:", n_seeds, "scenes,", hoisted_scenes, "with hoisted triangles,", bad, "mismatches")
sys.exit(1 if bad else 0)
