"""Differential fuzz of the render paths: random triangle soups (with needles and slivers) and 'walls and
crumbs' scenes (a few scene-sized triangles over many tiny ones — the ones lbvh_build hoists), random frame
sizes, bounce depths, light counts and extents from 1e-3 to 1e4.  Every path of the library — brute force and
the LBVH (both tree builds) walked per lane and by packets, each in the lockstep and the state-machine kernel — must give the
oracle's frame bit for bit (the oracle restates Source.cpp:642-780; tests/test_oracle_golden.py pins it).
scripts/fuzz_paths.py runs the same cases by the thousand without the oracle."""
import numpy as np
import pytest

from raytracing_amd import api, scenes
from test_gpu_parity import _soup

pytestmark = pytest.mark.gpu


def walls_and_crumbs(seed, n, extent, big, w, h, depth, lights):
    rng = np.random.default_rng(seed)
    sc = scenes.SceneDesc("crumbs%d" % seed, width=w, height=h, depth=depth)
    order = rng.permutation(n + big)
    for i in order:
        c = rng.uniform(-extent, extent, 3)
        size = extent * (rng.uniform(0.7, 3.0) if i < big else 10.0 ** rng.uniform(-3.0, -1.3))
        a, b, d = (c + rng.normal(size=3) * size for _ in range(3))
        sc.tri(a, b, d, rng.integers(0, 256, 3), 1.0 if rng.random() < 0.25 else 0.0)
    for _ in range(lights):
        sc.light(rng.uniform(-extent, extent, 3), float(10.0 ** rng.uniform(1, 4)))
    sc.cam_pos, sc.cam_rot = tuple(map(float, rng.uniform(-extent, extent, 3))), tuple(map(float, rng.uniform(-180, 180, 3)))
    return sc


def fuzz_case(seed, max_tris=1500, max_w=160, max_h=120):
    rng = np.random.default_rng(seed)
    kw = dict(w=int(rng.integers(1, max_w)), h=int(rng.integers(1, max_h)), depth=int(rng.integers(0, 8)), lights=int(rng.integers(0, 5)))
    if seed % 2:
        return walls_and_crumbs(seed, int(rng.integers(2, max_tris)), float(10 ** rng.uniform(-3, 4)), int(rng.integers(1, 40)), **kw)
    return _soup(seed, int(rng.integers(2, max_tris)), float(10 ** rng.uniform(-3, 4)), needles=bool(seed % 3 == 0), **kw)


def render_all_paths(desc):
    """[(label, frame)] of one scene through every path of the library."""
    frames = []
    with api.Scene.from_desc(desc) as sc:
        tr = api.camera_trans(desc.cam_rot)
        sc.commit(0, api.ACCEL_BRUTE)
        for kind, kname in ((api.KERNEL_LOCKSTEP, "lockstep"), (api.KERNEL_STATE_MACHINE, "state machine")):
            sc.set_kernel(kind)
            frames.append(("brute/" + kname, sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)))
        for build, bname in ((api.BUILD_PLOC, "ploc"), (api.BUILD_SAH, "sah"), (api.BUILD_FAST, "morton")):
            sc.set_build(build)
            sc.commit(0, api.ACCEL_LBVH)
            hoisted = int(sc.hoisted(0).size)
            for mode, mname in ((api.TRAVERSAL_PER_LANE, "per lane"), (api.TRAVERSAL_PACKET, "packets")):
                sc.set_traversal(mode)
                for kind, kname in ((api.KERNEL_LOCKSTEP, "lockstep"), (api.KERNEL_STATE_MACHINE, "state machine")):
                    sc.set_kernel(kind)
                    frames.append(("lbvh %s %s/%s" % (bname, mname, kname), sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)))
    return frames, hoisted


def test_every_path_equals_the_oracle_on_random_scenes(oracle):
    hoisted_scenes = 0
    for seed in range(7000, 7120):
        desc = fuzz_case(seed, max_tris=600, max_w=96, max_h=72)
        want = oracle.render_desc(desc).view(np.uint32)
        frames, hoisted = render_all_paths(desc)
        hoisted_scenes += hoisted > 0
        for label, f in frames:
            diff = int((want != f.view(np.uint32)).any(axis=2).sum())
            assert diff == 0, "seed %d (%d triangles, %dx%d, depth %d) %s: %d pixels differ from the oracle" % (
                seed, desc.n_triangles, desc.width, desc.height, desc.depth, label, diff)
    assert hoisted_scenes > 40
