"""Per-light direction grids for shadow rays (raytracing_amd/csrc/rt_lgrid.hip, rt_lgrid_trav.inl): frames with the
grids are the frames without them and the oracle's, bit for bit — processLighting's question (Source.cpp:473-476) must
get the reference's answer whichever structure finds the blocking triangle.  Scenes aim at the grid's edge cases: lights
inside, on and far outside the geometry, lights crowded by triangles (the "always" list, and past its limit the fall-back
to the tree), more lights than grids, needles, directions on cube-face edges and corners."""
import numpy as np
import pytest

from raytracing_amd import api, scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _all_paths(sc, desc):
    tr = api.camera_trans(desc.cam_rot)
    out = []
    for mode, mname in ((api.TRAVERSAL_PER_LANE, "per lane"), (api.TRAVERSAL_PACKET, "packets")):
        sc.set_traversal(mode)
        for kind, kname in ((api.KERNEL_LOCKSTEP, "lockstep"), (api.KERNEL_STATE_MACHINE, "state machine")):
            sc.set_kernel(kind)
            out.append((mname + "/" + kname, sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)))
    return out


def _check(desc, oracle, want_grids=None, threads=4):
    want = oracle.render_desc(desc, threads=threads)
    with api.Scene.from_desc(desc) as sc:
        for on in (True, False):
            sc.set_shadow_grid(on)
            sc.commit(0, api.ACCEL_LBVH)
            grids = [sc.shadow_grid(0, li)["cells_per_edge"] for li in range(desc.n_lights)]
            if not on:
                assert not any(grids)
            elif want_grids is not None:
                assert all(w is None or (g > 0) == w for g, w in zip(grids, want_grids)), grids
            for label, f in _all_paths(sc, desc):
                diff = int((_bits(want) != _bits(f)).any(axis=2).sum())
                assert diff == 0, "%s grid %s %s: %d pixels differ from the oracle" % (desc.name, on, label, diff)
    return want


def test_sphere_scene_with_and_without_the_grid(oracle):
    desc = scenes.spheres(8, 2, 192, 128, depth=3)
    want = _check(desc, oracle, want_grids=[True], threads=8)
    assert (want.max(axis=2) > 0).mean() > 0.5
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        g = sc.shadow_grid(0, 0)
        assert g["cells_per_edge"] >= 16 and g["entries"] > 2560 and g["always"] == 0 and g["build_ms"] > 0
        assert sc.shadow_grid(0, 1)["cells_per_edge"] == 0          # no such light
        sc.commit(0, api.ACCEL_BRUTE)
        assert sc.shadow_grid(0, 0) == {"cells_per_edge": 0, "entries": 0, "always": 0, "build_ms": 0.0}


def _sphere_soup(name, lights, n_spheres=5, subdiv=2, seed=11, w=112, h=80, depth=2, extra=()):
    desc = scenes.SceneDesc(name, width=w, height=h, depth=depth)
    scenes._room(desc)
    p1, p2, p3, col, sh = scenes.spheres_arrays(n_spheres, subdiv, seed)
    for i in range(len(p1)):
        desc.tri(p1[i], p2[i], p3[i], col[i], sh[i])
    for t in extra:
        desc.tri(*t)
    for pos, inten in lights:
        desc.light(pos, inten)
    return desc


def test_more_lights_than_grids(oracle):
    """Six lights: the first RT_SHADOW_GRID_LIGHTS get grids, the others walk the tree; the state machine's lanes sit at
    different lights at the same time."""
    rng = np.random.default_rng(3)
    lights = [(tuple(rng.uniform(-8, 8, 3)), float(rng.uniform(200, 2000))) for _ in range(6)]
    # (a random light may sit so close to a sphere that it is refused a grid: either way is right for the first four)
    _check(_sphere_soup("six_lights", lights), oracle, want_grids=[None] * 4 + [False] * 2)
    lights = [((0.0, 8.0, 0.0), 4096.0), ((5.0, 8.0, 5.0), 900.0), ((-9.0, -9.0, 9.0), 2000.0), ((9.5, 0.0, -9.5), 1500.0), ((0.0, -9.0, 0.0), 700.0)]
    _check(_sphere_soup("five_lights", lights, seed=0x5EED), oracle, want_grids=[None] * 4 + [False])


def test_light_positions_at_the_grids_edge_cases(oracle):
    """A light in the middle of a sphere (everything dark), on the cube-map seams of the geometry (sphere centres exactly
    along the diagonals and axes of the light), far outside the room, and AT a triangle's vertex."""
    p1, p2, p3, _, _ = scenes.spheres_arrays(5, 2, 11)
    centre = 0.5 * (p1[:320].min(axis=0) + p1[:320].max(axis=0))
    cases = {
        "inside_sphere": [(tuple(centre), 900.0)],
        "far_outside": [((300.0, 400.0, -250.0), 4.0e6)],
        "at_a_vertex": [(tuple(p1[400]), 500.0), ((0.0, 8.0, 0.0), 4096.0)],
        "on_a_wall": [((0.0, 10.0, 0.0), 4096.0), ((10.0, 10.0, 10.0), 9000.0)],
    }
    for name, lights in cases.items():
        _check(_sphere_soup(name, lights), oracle)
    # geometry exactly on the diagonals and axes through the light: directions on the edges and corners of the cube map
    L = np.array([1.0, 2.0, -1.0])
    extra = []
    for dirn in ((1, 1, 1), (1, -1, 1), (-1, 1, 0), (0, 1, 1), (1, 0, 0), (0, -1, 0), (-1, -1, -1), (1, 1, 0)):
        c = L + 3.0 * np.asarray(dirn, float)
        e1 = np.cross(dirn, (0.3, 0.5, 0.7)); e1 /= np.linalg.norm(e1)
        e2 = np.cross(dirn, e1); e2 /= np.linalg.norm(e2)
        extra.append((c - 0.4 * e1 - 0.4 * e2, c + 0.4 * e1 - 0.4 * e2, c + 0.4 * e2, (200, 180, 90), 0.0))
    _check(_sphere_soup("seams", [(tuple(L), 3000.0)], extra=extra), oracle)


def test_crowded_light_uses_the_always_list_and_then_the_tree(oracle):
    """Triangles within 1/250 of the scene size of the light, or wider than half their distance from it, cannot be binned
    by direction: up to 32 of them are tested for every ray of the light; with more the light keeps the tree walk."""
    L = np.array([2.0, 3.0, 1.0])
    rng = np.random.default_rng(9)

    def crowd(k):
        out = []
        for _ in range(k):
            c = L + rng.normal(size=3) * 0.08
            out.append((c + rng.normal(size=3) * 0.05, c + rng.normal(size=3) * 0.05, c + rng.normal(size=3) * 0.05, (250, 250, 250), 0.0))
        return out

    desc = _sphere_soup("crowd_12", [(tuple(L), 2500.0)], extra=crowd(12))
    _check(desc, oracle, want_grids=[True])
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        assert 1 <= sc.shadow_grid(0, 0)["always"] <= 12
    _check(_sphere_soup("crowd_60", [(tuple(L), 2500.0)], extra=crowd(60)), oracle, want_grids=[False])


def test_random_soups_with_needles_and_many_lights(oracle):
    """The fuzz generator's soups (needles, scales 1e-3..1e4, up to 4 lights, cameras outside) through every LBVH path with
    the grids on — tests/test_gpu_fuzz.py runs them too; here with larger frames and every scene lit."""
    from test_gpu_parity import _soup
    from test_gpu_fuzz import walls_and_crumbs
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        kw = dict(w=int(rng.integers(24, 120)), h=int(rng.integers(24, 90)), depth=int(rng.integers(0, 4)), lights=int(rng.integers(1, 5)))
        if seed % 2:
            desc = walls_and_crumbs(seed, int(rng.integers(50, 900)), float(10 ** rng.uniform(-3, 4)), int(rng.integers(1, 40)), **kw)
        else:
            desc = _soup(seed, int(rng.integers(50, 900)), float(10 ** rng.uniform(-3, 4)), needles=bool(seed % 3 == 0), **kw)
        want = oracle.render_desc(desc)
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, api.ACCEL_LBVH)
            for label, f in _all_paths(sc, desc):
                diff = int((_bits(want) != _bits(f)).any(axis=2).sum())
                assert diff == 0, "seed %d %s: %d pixels differ" % (seed, label, diff)


def test_full_size_scene_grid_equals_tree_walk():
    """BASELINE config 4 (102 412 triangles) at 800x800: the frame with the light's grid is the frame without it, by the
    per-lane walk and by packets; and the work counters see the difference (no box tests for shadow rays)."""
    from test_gpu_lbvh import _full_size_scene
    desc = _full_size_scene()
    tr = api.camera_trans(desc.cam_rot)
    frames, boxes = {}, {}
    with api.Scene.from_desc(desc) as sc:
        for on in (True, False):
            sc.set_shadow_grid(on)
            sc.commit(0, api.ACCEL_LBVH)
            if on:
                g = sc.shadow_grid(0, 0)
                assert g["cells_per_edge"] >= 128 and g["entries"] > 102400 and g["always"] == 0
            for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
                sc.set_traversal(mode)
                frames[(on, mode)] = sc.render(desc.cam_pos, tr, 800, 800, desc.depth)
            f, st = sc.render_stats(desc.cam_pos, tr, 800, 800, desc.depth)
            assert np.array_equal(_bits(f), _bits(frames[(on, api.TRAVERSAL_PACKET)]))
            boxes[on] = st.n_box
            assert st.rays_shadow > 300000
    first = frames[(True, api.TRAVERSAL_PER_LANE)]
    for k, f in frames.items():
        assert np.array_equal(_bits(f), _bits(first)), k
    assert boxes[True] < 0.8 * boxes[False]
