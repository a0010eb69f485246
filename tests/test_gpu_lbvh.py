"""the GPU-built LBVH.  Structure (Karras invariants) and, above all, results:
traversal must return what the reference's brute-force castRay loop returns — same t, same
triangle (lowest index on ties), same bits — so every golden frame is re-rendered through the
tree, and random scenes are compared with the brute-force kernel and the oracle."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases, load_golden
from raytracing_amd import api, scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _check_tree(nodes, tris, hoisted=()):
    """2N-1 nodes, every leaf reachable exactly once, child boxes inside the parent's slot for
    them, every leaf box containing its triangle; the hoisted triangles (kept out of the tree,
    rt_scene_get_hoisted) are in no leaf."""
    n = tris.shape[0]
    assert nodes.shape == (n - len(hoisted) - 1, 16)
    refs = nodes[:, 12:14].copy().view(np.int32)
    seen_leaf = np.zeros(n, dtype=np.int32)
    seen_node = np.zeros(nodes.shape[0], dtype=np.int32)
    seen_node[0] = 1
    stack = [(0, None)]
    max_depth = 0
    while stack:
        node, depth = stack.pop()
        depth = (depth or 0) + 1
        max_depth = max(max_depth, depth)
        for side in (0, 1):
            box_lo = nodes[node, side * 6: side * 6 + 3]
            box_hi = nodes[node, side * 6 + 3: side * 6 + 6]
            assert np.all(box_lo <= box_hi)
            r = int(refs[node, side])
            if r < 0:
                t = ~r
                assert 0 <= t < n
                seen_leaf[t] += 1
                v = tris[t, 6:15].reshape(3, 3)
                assert np.all(v.min(axis=0) >= box_lo) and np.all(v.max(axis=0) <= box_hi)
            else:
                assert 0 < r < nodes.shape[0]
                seen_node[r] += 1
                child_lo = np.minimum(nodes[r, 0:3], nodes[r, 6:9])
                child_hi = np.maximum(nodes[r, 3:6], nodes[r, 9:12])
                assert np.all(child_lo >= box_lo) and np.all(child_hi <= box_hi)   # parent box == union of children
                assert np.array_equal(child_lo, box_lo) and np.array_equal(child_hi, box_hi)
                stack.append((r, depth))
    in_tree = np.ones(n, dtype=np.int32)
    in_tree[np.asarray(hoisted, dtype=np.int64)] = 0
    assert np.array_equal(seen_leaf, in_tree) and np.all(seen_node == 1)
    return max_depth


BUILDS = [api.BUILD_PLOC, api.BUILD_FAST, api.BUILD_SAH]
BUILD_IDS = ["ploc", "karras", "host-sah"]


@pytest.mark.parametrize("build", BUILDS, ids=BUILD_IDS)
def test_tree_invariants(build):
    """Every build: 24 triangles (the clustering finishes in its one-workgroup tail), 972 (the same), 2572 and 7692
    (two-launch passes first, whose workgroups own 224 clusters each and look 8 places past their ends)."""
    for desc in (scenes.stock(8, 8), scenes.spheres(3, 2, 8, 8), scenes.spheres(2, 3, 8, 8, seed=7), scenes.spheres(6, 3, 8, 8)):
        with api.Scene.from_desc(desc) as sc:
            sc.set_build(build)
            sc.commit(0, api.ACCEL_LBVH)
            assert sc.accel(0) == api.ACCEL_LBVH
            assert sc.build_kind(0) == build
            depth = _check_tree(sc.bvh(0), sc.triangles(), sc.hoisted(0))
            assert depth <= 62 and depth == sc.bvh_depth(0)[0]
            assert sc.build_ms(0) > 0
            if build != api.BUILD_SAH:
                first = sc.bvh(0).copy()                                    # the GPU builds are deterministic: same nodes, same numbering
                sc.set_build(api.BUILD_SAH).set_build(build)
                sc.commit(0, api.ACCEL_LBVH)
                assert np.array_equal(first.view(np.uint32), sc.bvh(0).view(np.uint32))


def sc_build_before(sc):
    return getattr(sc, "_last_build", api.BUILD_PLOC)


def test_clustered_tree_is_the_default_and_cheaper_to_cross_than_the_karras_tree():
    """rt_scene_set_build: the GPU's clustered (PLOC) tree is what a commit builds unless told otherwise; on tessellated
    spheres its sum of child-box areas (the expected number of box tests of a random ray, up to a constant) is well below
    the Morton/Karras tree's and within a few per cent of the host's binned-SAH tree; changing the knob makes the
    committed tree stale."""
    desc = scenes.spheres(6, 3, 8, 8)
    cost = {}
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        assert sc.build_kind(0) == api.BUILD_PLOC
        default_nodes = sc.bvh(0).copy()
        for build in (api.BUILD_PLOC, api.BUILD_SAH, api.BUILD_FAST):
            sc.set_build(build)
            if build != sc_build_before(sc):
                with pytest.raises(api.RtError):                            # the committed tree is stale now
                    sc.render((0, 0, 0), api.camera_trans((0, 0, 0)), 8, 8, 1)
            sc.commit(0, api.ACCEL_LBVH)
            sc._last_build = build
            nodes = sc.bvh(0)
            _check_tree(nodes, sc.triangles(), sc.hoisted(0))
            e = np.concatenate([nodes[:, 3:6] - nodes[:, 0:3], nodes[:, 9:12] - nodes[:, 6:9]]).astype(np.float64)
            cost[build] = float((e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]).sum())
            if build == api.BUILD_PLOC:
                assert np.array_equal(nodes.view(np.uint32), default_nodes.view(np.uint32))
        assert cost[api.BUILD_SAH] < 0.95 * cost[api.BUILD_FAST], cost
        assert cost[api.BUILD_PLOC] < 0.95 * cost[api.BUILD_FAST], cost
        assert cost[api.BUILD_PLOC] < 1.06 * cost[api.BUILD_SAH], cost
        with pytest.raises(api.RtError):
            sc.set_build(7)


def _box_area(tris):
    v = tris[:, 6:15].reshape(-1, 3, 3)
    e = (v.max(axis=1) - v.min(axis=1)).astype(np.float64)
    return e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]


def test_scene_spanning_triangles_are_hoisted(oracle):
    """The room's 12 wall triangles around tessellated spheres stay out of the tree (each would
    widen every box on its path to the root); frames are the same bits with and without them in
    the tree, for both traversals, and equal the oracle's."""
    desc = scenes.spheres(3, 2, 96, 64, depth=2)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        tris = sc.triangles()
        h = sc.hoisted(0)
        assert h.tolist() == list(range(12))                      # _room() comes first in the scene
        v = tris[:, 6:15].reshape(-1, 3, 3)
        ext = (v.max(axis=(0, 1)) - v.min(axis=(0, 1))).astype(np.float64)
        scene_area = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0]
        big = _box_area(tris) * 8.0 >= scene_area
        assert np.flatnonzero(big).tolist() == h.tolist()
        _check_tree(sc.bvh(0), tris, h)
        want = oracle.render_desc(desc)
        for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            sc.set_traversal(mode)
            got = sc.render(desc.cam_pos, api.camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth)
            assert np.array_equal(_bits(got), _bits(want))
        sc.set_traversal(api.TRAVERSAL_AUTO)
        # rays that graze walls and spheres at equal t keep the lowest index
        rays = np.array([[0, 0, 0, 0, -1, 0, -1], [0, 0, 0, 1, 0, 0, -1], [0, 0, 0, 0, 1, 0, 3]], np.float32)
        sc.commit(0, api.ACCEL_BRUTE)
        ref = sc.cast_rays(rays)
        sc.commit(0, api.ACCEL_LBVH)
        assert np.array_equal(_bits(sc.cast_rays(rays)), _bits(ref))


def test_frame_batches_through_the_tree_equal_single_frames():
    """Launches of 8 or more frames of an LBVH scene run their workgroups in the XCD-aware band order
    (rt_kernels.hip launch_render): every frame of the batch must still be the frame rt_render gives
    for its pose, with both traversals, for frame sizes whose last band is padded."""
    import torch
    for (w, h) in ((120, 88), (67, 45), (200, 40), (40, 200)):           # 165, 54, 125 and 125 tiles in 8 bands: the last band is padded in each
        desc = scenes.spheres(3, 2, w, h, depth=2)
        cams, trs = api.camera_sweep(19, desc.cam_pos, desc.cam_rot)
        n = w * h * 3
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, api.ACCEL_LBVH)
            for mode, kind, tpw in ((api.TRAVERSAL_PER_LANE, api.KERNEL_AUTO, 0), (api.TRAVERSAL_PACKET, api.KERNEL_LOCKSTEP, 0),
                                    (api.TRAVERSAL_PACKET, api.KERNEL_LOCKSTEP, 3), (api.TRAVERSAL_PER_LANE, api.KERNEL_LOCKSTEP, 2),
                                    (api.TRAVERSAL_PACKET, api.KERNEL_STATE_MACHINE, 0)):
                sc.set_traversal(mode)
                sc.set_kernel(kind)
                sc.set_tiles_per_wave(0)
                singles = [sc.render(cams[i], trs[i], w, h, desc.depth) for i in range(19)]
                sc.set_tiles_per_wave(tpw)                               # (runs of tiles that cross a band's edge)
                buf = torch.full((19 * n,), -1.0, dtype=torch.float32, device="cuda:0")
                sc.render_frames_device(0, cams, trs, w, h, desc.depth, api.STRICT, buf.data_ptr(), n,
                                        torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                got = buf.cpu().numpy().reshape(19, h, w, 3)
                for i in range(19):
                    assert np.array_equal(_bits(got[i]), _bits(singles[i])), (w, h, mode, kind, tpw, i)
        assert not np.array_equal(singles[0], singles[18])


def test_long_launches_of_large_frames_equal_single_frames():
    """The launch shape bench.py times at 3840x2160 — 16 frames of the 102 412-triangle scene in one launch: XCD bands, four tiles
    per wavefront down a column (launch_render: 129 600 tiles a frame, 2 M in the launch) — against the same poses rendered one
    frame per launch (one tile per wavefront, no bands): the same bits."""
    import torch
    w, h = 3840, 2160
    desc = scenes.spheres(20, 4, w, h)
    cams, trs = api.camera_sweep(16, desc.cam_pos, desc.cam_rot)
    n = w * h * 3
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        assert sc.traversal(0, trs[0], w, h) == api.TRAVERSAL_PACKET
        buf = torch.full((16 * n,), -1.0, dtype=torch.float32, device="cuda:0")
        sc.render_frames_device(0, cams, trs, w, h, desc.depth, api.STRICT, buf.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        one = torch.empty(n, dtype=torch.float32, device="cuda:0")
        for i in (0, 5, 10, 15):
            sc.render_device(0, cams[i], trs[i], w, h, desc.depth, api.STRICT, one.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(buf[i * n:(i + 1) * n].view(torch.int32), one.view(torch.int32)), i
        del buf, one


def test_hoisting_limits():
    """Nothing is hoisted from a scene made of large triangles (more than 32 qualify) or from a
    small one where they would be more than a quarter; an all-walls scene still builds."""
    with api.Scene.from_desc(scenes.stock(8, 8)) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        assert sc.hoisted(0).size == 0                              # 12 of 24 triangles qualify: too large a share
    rng = np.random.default_rng(5)
    with api.Scene() as sc:
        for _ in range(200):
            p = rng.uniform(-5, 5, (3, 3))
            sc.Triangle(p[0], p[1], p[2], (9, 9, 9), 0)
        sc.addLight((0, 8, 0), 4096)
        sc.commit(0, api.ACCEL_LBVH)
        assert sc.hoisted(0).size == 0
        _check_tree(sc.bvh(0), sc.triangles())


def test_duplicate_centroids_still_build():
    """Many equal Morton codes: index-augmented keys keep the tree well formed."""
    with api.Scene() as sc:
        for i in range(300):
            sc.Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), (i % 255, 3, 4), 0)     # 300 identical triangles
        sc.addLight((0, 8, 0), 4096)
        sc.commit(0, api.ACCEL_LBVH)
        depth = _check_tree(sc.bvh(0), sc.triangles(), sc.hoisted(0))
        assert depth <= 12
        # all hit at the same t: the lowest index must win (Source.cpp:391 strict <)
        out = sc.cast_rays(np.array([[0.2, 0.2, 1, 0, 0, -1, -1], [0.2, 0.2, 1, 0, 0, -1, 0], [0.2, 0.2, 1, 0, 0, -1, 5]], np.float32))
        assert out[:, 1].tolist() == [0, 1, 0]


@pytest.mark.parametrize("case", [c for c in golden_cases() if c not in ("empty_16x8",)])
def test_golden_frames_through_the_tree(case):
    g = load_golden(case)
    d = g["desc"]
    if d.n_triangles < 2:
        pytest.skip("a tree needs two leaves")
    for traversal in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):      # (degenerate_48: a NaN-normal leaf in both walks)
        frame = api.render_desc(d, 0, api.ACCEL_LBVH, traversal=traversal)
        assert hashlib.sha256(frame.tobytes()).hexdigest() == g["sha256"], traversal


def test_cast_rays_known_answers_through_the_tree():
    z = np.load(os.path.join(GOLDEN, "rays_stock.npz"), allow_pickle=False)
    desc = scenes.SceneDesc.from_text(bytes(z["scene"]).decode())
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        got = sc.cast_rays(z["rays"])
    want = z["out"]
    assert np.array_equal(got[:, 0], want[:, 0])
    assert np.array_equal(got[:, 1], want[:, 1])
    hit = want[:, 0] > 0
    assert np.array_equal(_bits(got[hit]), _bits(want[hit]))


def test_lbvh_equals_brute_on_sphere_scenes(oracle):
    """Tessellated spheres with mirrors (config 4's generator at a size the brute-force kernel and
    the oracle finish quickly): LBVH frame == brute-force frame == oracle frame."""
    desc = scenes.spheres(8, 2, 256, 192, depth=3)          # 8 x 320 + 12 triangles
    a = api.render_desc(desc, 0, api.ACCEL_LBVH)
    b = api.render_desc(desc, 0, api.ACCEL_BRUTE)
    assert np.array_equal(_bits(a), _bits(b))
    want = oracle.render_desc(desc, threads=8)
    assert np.array_equal(_bits(a), _bits(want))


_FULL = {}


def _full_size_scene():
    """BASELINE config 4's 102 412 triangles, built once per test process (the Python-side scene list takes seconds)."""
    if "desc" not in _FULL:
        _FULL["desc"] = scenes.spheres(20, 4, 800, 800)
    return _FULL["desc"]


def test_lbvh_100k_rays_equal_brute():
    """BASELINE config 4 at full size (102 412 triangles): castRay through the tree of EVERY build — the GPU's
    clustered tree, the GPU's Karras tree, the host's SAH tree — against the brute-force kernel on 4096 rays (the
    brute-force walk over 100k triangles is the expensive side), by the per-lane walk and by packets."""
    desc = _full_size_scene()
    rng = np.random.default_rng(5)
    rays = np.zeros((4096, 7), np.float32)
    rays[:, 0:3] = rng.uniform(-9.5, 9.5, (4096, 3))
    rays[:, 3:6] = rng.normal(size=(4096, 3))
    rays[:, 6] = -1
    with api.Scene.from_desc(desc) as sc:
        assert sc.n_triangles == 102412
        sc.commit(0, api.ACCEL_BRUTE)
        b = sc.cast_rays(rays)
        for build in BUILDS:
            sc.set_build(build)
            sc.commit(0, api.ACCEL_LBVH)
            assert sc.build_kind(0) == build
            for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
                sc.set_traversal(mode)
                assert np.array_equal(_bits(sc.cast_rays(rays)), _bits(b)), (build, mode)
    assert 0.5 < b[:, 0].mean() <= 1.0


def test_gpu_builds_of_the_full_size_scene_are_fast():
    """north_star's build (Morton sort + tree on the GPU) at 102 412 triangles: the clustered tree within 2 ms of GPU
    time, a sound tree of SAH-like quality; the Karras tree within 1 ms.  (Second commit: the first one in a process
    also loads the kernels' code objects.)"""
    desc = _full_size_scene()
    cost = {}
    with api.Scene.from_desc(desc) as sc:
        for build, limit in ((api.BUILD_PLOC, 2.0), (api.BUILD_FAST, 1.0)):
            sc.set_build(build)
            sc.commit(0, api.ACCEL_LBVH)
            ms = 1e9
            for _ in range(2):
                sc.commit(0, api.ACCEL_LBVH)
                ms = min(ms, sc.build_ms(0))
            assert sc.build_kind(0) == build
            assert 0 < ms <= limit, (build, ms)
            nodes = sc.bvh(0)
            e = np.concatenate([nodes[:, 3:6] - nodes[:, 0:3], nodes[:, 9:12] - nodes[:, 6:9]]).astype(np.float64)
            cost[build] = float((e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]).sum())
            if build == api.BUILD_PLOC:
                assert _check_tree(nodes, sc.triangles(), sc.hoisted(0)) == sc.bvh_depth(0)[0]
    assert cost[api.BUILD_PLOC] < 0.9 * cost[api.BUILD_FAST], cost


def test_auto_picks_lbvh_above_threshold():
    with api.Scene.from_desc(scenes.spheres(4, 2, 8, 8)) as sc:       # 1292 triangles (> RT_BRUTE_AUTO_MAX = 40)
        sc.commit(0, api.ACCEL_AUTO)
        assert sc.accel(0) == api.ACCEL_LBVH
    with api.Scene.from_desc(scenes.stock(8, 8)) as sc:
        sc.commit(0, api.ACCEL_AUTO)
        assert sc.accel(0) == api.ACCEL_BRUTE


# ---- deep trees: the traversal stacks -------------------------------------

def _graded_chain(bits, per, dups=0, seed=0):
    """Geometry graded towards a corner so that the Karras tree is a long chain: one cluster of `per`
    nearly identical triangles per Morton bit b = 3*level + axis (box centre with only that bit set
    among the leading ones), sizes halving with the level; `dups` exact copies of the innermost
    triangle add index-bit levels below the chain.  Emulated on the CPU when the test was written:
    (54, 3) -> binary depth 55, 4-wide depth 19; (51, 4, 256) -> 57 and 23."""
    rng = np.random.default_rng(seed)
    tris = []
    for b in range(bits):
        level, axis = divmod(b, 3)
        cell = 2.0 ** -(level + 1)
        c = np.full(3, 0.5 * cell)
        c[axis] = 1.5 * cell
        base = c + 0.45 * cell * rng.uniform(-1, 1, (3, 3))
        for _ in range(per):
            t = base + 1e-3 * cell * rng.normal(size=(3, 3))
            t += c - 0.5 * (t.min(axis=0) + t.max(axis=0))           # the box centre decides the Morton code
            tris.append(t)
    tris += [tris[-1].copy() for _ in range(dups)]
    desc = scenes.SceneDesc("graded_%d_%d_%d" % (bits, per, dups), width=96, height=64, depth=3,
                            cam_pos=(1.2, 1.1, 1.3), cam_rot=(225.0, -35.26, 0.0))
    rng2 = np.random.default_rng(seed + 1)
    for t in np.asarray(tris, np.float32):
        desc.tri(t[0], t[1], t[2], rng2.integers(30, 256, 3), 1.0 if rng2.random() < 0.2 else 0.0)
    desc.light((1.5, 2.0, 0.5), 64.0)
    desc.light((0.01, 0.3, 0.02), 1.0)
    return desc, np.asarray(tris, np.float32)


def _rays_into_the_corner(tris, n_random, seed):
    """Rays from outside aimed at every cluster down to the innermost (they cross all the nested boxes
    on the way), plus rays from the inside out and random ones."""
    rng = np.random.default_rng(seed)
    centres = tris.mean(axis=1)
    rays = []
    for c in centres[:: max(1, len(centres) // 400)]:
        o = np.array([0.9, 1.0, 1.1]) + 0.05 * rng.normal(size=3)
        rays.append([*o, *(c - o), -1.0])
        rays.append([*(c * 0.5), *(rng.normal(size=3)), -1.0])
    for _ in range(n_random):
        o = rng.uniform(0, 1.2, 3) * 10.0 ** rng.uniform(-6, 0)
        rays.append([*o, *rng.normal(size=3), float(rng.integers(-1, len(tris)))])
    return np.asarray(rays, np.float32)


@pytest.mark.parametrize("build", BUILDS, ids=BUILD_IDS)
def test_deep_tree_within_the_per_lane_stack(oracle, build):
    """A 4-wide tree ~19 levels deep (Morton/Karras build; the SAH build of the same scene is whatever it
    is, at most 63 binary levels): the per-lane walk keeps up to 3 entries per level, far past the 16
    entries it holds in LDS and close to the RT_STACK_LIMIT of 64.  Frames and ray batches by the
    per-lane walk, by packets and by brute force are the oracle's."""
    desc, tris = _graded_chain(54, 3)
    rays = _rays_into_the_corner(tris, 1500, 3)
    otris, _ = oracle.build_scene(desc)
    want_rays = np.stack([oracle.cast_ray(otris, r[0:3], r[3:6], int(r[6])) for r in rays])
    want = oracle.render_desc(desc, threads=4)
    assert (want.max(axis=2) > 0).sum() >= 20
    with api.Scene.from_desc(desc) as sc:
        sc.set_build(build)
        sc.commit(0, api.ACCEL_LBVH)
        d2, d4 = sc.bvh_depth(0)
        if build == api.BUILD_FAST:
            assert 40 <= d2 <= 63 and 16 < 3 * d4 <= api.STACK_LIMIT, (d2, d4)
        assert 1 <= d2 <= 63, (d2, d4)
        assert _check_tree(sc.bvh(0), sc.triangles(), sc.hoisted(0)) == d2
        tr = api.camera_trans(desc.cam_rot)
        for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            sc.set_traversal(mode)
            assert sc.traversal(0, tr, desc.width, desc.height) == (mode if 3 * d4 <= api.STACK_LIMIT else api.TRAVERSAL_PACKET)
            got = sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)
            assert np.array_equal(_bits(got), _bits(want)), mode
            out = sc.cast_rays(rays)
            assert np.array_equal(out[:, :2], want_rays[:, :2]), mode
            hit = want_rays[:, 0] > 0
            assert np.array_equal(_bits(out[hit]), _bits(want_rays[hit])), mode
        assert 0.05 < (want_rays[:, 0] > 0).mean() < 1.0


def test_tree_too_deep_for_the_per_lane_stack_goes_by_packets(oracle):
    """3 * (4-wide depth) > RT_STACK_LIMIT: the per-lane walk could overflow its stack, so the commit
    marks the tree and every launch walks it by packets (one stack entry per binary level, at most
    63) whatever traversal is asked for — results stay the oracle's, nothing is dropped silently."""
    desc, tris = _graded_chain(51, 4, dups=256)
    rays = _rays_into_the_corner(tris, 1500, 5)
    otris, _ = oracle.build_scene(desc)
    want_rays = np.stack([oracle.cast_ray(otris, r[0:3], r[3:6], int(r[6])) for r in rays])
    want = oracle.render_desc(desc, threads=4)
    with api.Scene.from_desc(desc) as sc:
        sc.set_build(api.BUILD_FAST)             # (the Morton/Karras tree of this scene is the deep one)
        sc.commit(0, api.ACCEL_LBVH)
        d2, d4 = sc.bvh_depth(0)
        assert d2 <= 63 and 3 * d4 > api.STACK_LIMIT, (d2, d4)
        tr = api.camera_trans(desc.cam_rot)
        for mode in (api.TRAVERSAL_AUTO, api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            sc.set_traversal(mode)
            assert sc.traversal(0, tr, desc.width, desc.height) == api.TRAVERSAL_PACKET
            got = sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)
            assert np.array_equal(_bits(got), _bits(want)), mode
            out = sc.cast_rays(rays)
            assert np.array_equal(out[:, :2], want_rays[:, :2]), mode
        sc.commit(0, api.ACCEL_BRUTE)
        assert sc.bvh_depth(0) == (0, 0)
        assert np.array_equal(_bits(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)), _bits(want))


@pytest.mark.parametrize("build", BUILDS, ids=BUILD_IDS)
def test_config4_full_size_all_traversals(oracle, build):
    """BASELINE config 4 at full size — 102 412 triangles, 800x800, through the tree of every build (the GPU's
    clustered tree is north_star's and the default): packets (with the walls hoisted and the XCD band order of a
    frame batch) == per-lane walk == forced brute force, bit for bit, and a 64x64 frame of the same scene against
    the oracle (the reference's loop over 100k triangles)."""
    import torch
    desc = _full_size_scene()
    tr = api.camera_trans(desc.cam_rot)
    with api.Scene.from_desc(desc) as sc:
        assert sc.n_triangles == 102412
        sc.set_build(build)
        sc.commit(0, api.ACCEL_LBVH)
        assert sc.build_kind(0) == build
        assert sc.hoisted(0).tolist() == list(range(12))
        d2, d4 = sc.bvh_depth(0)
        assert 3 * d4 <= api.STACK_LIMIT and d2 <= 63, (d2, d4)
        frames = {}
        for mode in (api.TRAVERSAL_PACKET, api.TRAVERSAL_PER_LANE):
            sc.set_traversal(mode)
            frames[mode] = sc.render(desc.cam_pos, tr, 800, 800, desc.depth)
        assert np.array_equal(_bits(frames[api.TRAVERSAL_PACKET]), _bits(frames[api.TRAVERSAL_PER_LANE]))
        # a batch of 8 equal poses runs in the band order: every copy is the single frame
        sc.set_traversal(api.TRAVERSAL_PACKET)
        n = 800 * 800 * 3
        buf = torch.empty(8 * n, dtype=torch.float32, device="cuda:0")
        sc.render_frames_device(0, np.tile(np.asarray(desc.cam_pos, np.float32), (8, 1)), np.tile(tr, (8, 1)), 800, 800, desc.depth,
                                api.STRICT, buf.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        batch = buf.cpu().numpy().reshape(8, 800, 800, 3)
        for k in (0, 3, 7):
            assert np.array_equal(_bits(batch[k]), _bits(frames[api.TRAVERSAL_PACKET])), k
        small = {}
        for mode in (api.TRAVERSAL_PACKET, api.TRAVERSAL_PER_LANE):
            sc.set_traversal(mode)
            small[mode] = sc.render(desc.cam_pos, tr, 64, 64, desc.depth)
        if "brute" not in _FULL:
            sc.commit(0, api.ACCEL_BRUTE)
            _FULL["brute"] = sc.render(desc.cam_pos, tr, 800, 800, desc.depth)
        brute = _FULL["brute"]
        assert np.array_equal(_bits(brute), _bits(frames[api.TRAVERSAL_PACKET]))
    if "oracle64" not in _FULL:
        _FULL["oracle64"] = oracle.render_desc(desc.with_size(64, 64), threads=16)
    want = _FULL["oracle64"]
    for mode, got in small.items():
        assert np.array_equal(_bits(got), _bits(want)), mode
    assert (brute.max(axis=2) > 0).mean() > 0.5


def test_counting_kernel_follows_the_traversal_in_use():
    """rt_render_stats on an LBVH scene counts inside the traversal the frame really gets (packets or
    per lane): same frame bits, same ray and pixel counts from both, each with its own box/plane work."""
    desc = scenes.spheres(6, 3, 320, 200, depth=2)
    tr = api.camera_trans(desc.cam_rot)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        plain = sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)
        stats = {}
        for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            sc.set_traversal(mode)
            frame, st = sc.render_stats(desc.cam_pos, tr, desc.width, desc.height, desc.depth)
            assert np.array_equal(_bits(frame), _bits(plain)), mode
            stats[mode] = st.as_dict()
    a, b = stats[api.TRAVERSAL_PER_LANE], stats[api.TRAVERSAL_PACKET]
    for k in ("rays_primary", "rays_bounce", "rays_shadow", "px_hit", "px_mirror", "px_lit"):
        assert a[k] == b[k] > 0, k
    assert a["rays_primary"] == desc.width * desc.height
    for k in ("n_box", "n_plane", "n_point", "n_edge"):
        assert a[k] > 0 and b[k] > 0, k
    assert b["n_box"] != a["n_box"]                   # a wave's walk visits the union of its rays' boxes


def test_counting_launches_with_runs_of_tiles_cover_the_frame():
    """rt_render_stats with rt_scene_set_tiles_per_wave on an LBVH scene: the counting instances of the tree kernels lay their
    waves out as the frame's kernels do (packets: runs down a column; per lane: along the row) — a launcher that sized the grid
    for the other form would leave tiles out.  Same frame, same ray counts, whatever the run length."""
    desc = scenes.spheres(6, 3, 328, 200, depth=2)      # 41 x 25 tiles: neither count is a multiple of the run lengths
    tr = api.camera_trans(desc.cam_rot)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            sc.set_traversal(mode)
            sc.set_tiles_per_wave(0)
            want, st0 = sc.render_stats(desc.cam_pos, tr, desc.width, desc.height, desc.depth)
            for tpw in (2, 3, 7):
                sc.set_tiles_per_wave(tpw)
                got, st = sc.render_stats(desc.cam_pos, tr, desc.width, desc.height, desc.depth)
                assert np.array_equal(_bits(got), _bits(want)), (mode, tpw)
                a, b = st0.as_dict(), st.as_dict()
                for k in ("rays_primary", "rays_bounce", "rays_shadow", "px_hit", "px_mirror", "px_lit"):
                    assert a[k] == b[k], (mode, tpw, k)
                assert np.array_equal(_bits(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)), _bits(want)), (mode, tpw)
