"""The edge-term sign filter (rt_kernels.hip edge_filter, rt_scene.cpp edge_filter_planes): the kernels decide the three
comparisons of Source.cpp:392-394 from M_k . (I - p1) where that is provably the reference's sign and fall back to the
reference's own 51 operations for hit points within the rounding bound of an edge line.  Rays aimed AT the edge lines and
vertices of triangles of every size and shape — the hits the filter must hand over — have to come out as the oracle's
castRay decides them, bit for bit, through brute force, both tree walks and frames."""
import numpy as np
import pytest

from raytracing_amd import api, scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _triangles(rng, n, scale, needle):
    """n triangles around the origin: sizes `scale` * 10^U(-2, 0), some of them needles (aspect up to 1e4)."""
    tris = []
    for _ in range(n):
        c = rng.uniform(-4, 4, 3) * scale
        s = scale * 10.0 ** rng.uniform(-2, 0)
        a, b = rng.normal(size=3), rng.normal(size=3)
        a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
        if needle and rng.random() < 0.4:
            b = a + b * 10.0 ** rng.uniform(-4, -1)
        tris.append((c, c + a * s, c + b * s))
    return tris


def _edge_rays(rng, tris, per_tri, scale):
    """Rays through points ON the edge lines (and a few ulps either side), the vertices, and the shared edges of neighbours."""
    rays = []
    for (p1, p2, p3) in tris:
        P = [np.float32(p1), np.float32(p2), np.float32(p3)]
        for _ in range(per_tri):
            k = rng.integers(0, 3)
            a, b = P[k].astype(np.float64), P[(k + 1) % 3].astype(np.float64)
            u = rng.choice([0.0, 1.0, 0.5, rng.random(), rng.random()])
            target = a + u * (b - a)
            if rng.random() < 0.5:                                  # nudge by a few units in the last place of the largest coordinate
                target = target + rng.integers(-3, 4, 3) * np.spacing(np.float32(np.abs(target).max() + 1e-30)).astype(np.float64)
            o = target + rng.normal(size=3) * scale * 10.0 ** rng.uniform(-1, 1)
            d = target - o
            if rng.random() < 0.5:
                d = d / np.linalg.norm(d)
            rays.append(np.concatenate([o, d, [-1.0]]))
    return np.asarray(rays, dtype=np.float32)


def _check(oracle, sc, tris28, rays, what):
    got = sc.cast_rays(rays)
    n_hit = 0
    for r, g in zip(rays, got):
        w = oracle.cast_ray(tris28, r[:3], r[3:6], int(r[6]))
        assert g[0] == w[0] and g[1] == w[1], (what, r, g, w)
        if w[0] > 0:
            n_hit += 1
            assert np.array_equal(_bits(g), _bits(w)), (what, r, g, w)
    return n_hit


@pytest.mark.parametrize("scale", [1e-4, 1.0, 3e3])
def test_rays_aimed_at_edges_and_vertices(oracle, scale):
    rng = np.random.default_rng(int(scale * 7) + 11)
    tris = _triangles(rng, 60, scale, needle=True)
    # a fan around a shared vertex and a strip with shared edges: neighbours meet exactly on their common edge
    c = np.zeros(3)
    ring = [np.array([np.cos(t), np.sin(t), 0.3 * np.sin(3 * t)]) * scale for t in np.linspace(0, 2 * np.pi, 9)[:-1]]
    tris += [(c, ring[k], ring[(k + 1) % 8]) for k in range(8)]
    desc = scenes.SceneDesc("edges", width=32, height=32, depth=1)
    for (p1, p2, p3) in tris:
        desc.tri(p1, p2, p3, (200, 100, 50), 0.0)
    desc.light((0.0, 0.0, 5.0 * scale), 10.0)
    tris28, _ = oracle.build_scene(desc)
    rays = _edge_rays(rng, tris, 12, scale)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        hits = _check(oracle, sc, tris28, rays, "brute")
        sc.commit(0, api.ACCEL_LBVH)
        for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            sc.set_traversal(mode)
            assert _check(oracle, sc, tris28, rays, mode) == hits
    assert 0.2 * len(rays) < hits < 0.98 * len(rays)            # on the fence: both outcomes occur


def test_degenerate_and_huge_triangles_take_the_reference_arithmetic(oracle):
    """Zero-area triangles (normal not finite), coordinates near the ends of the binary32 range: the filter must stand
    aside (its bound is NaN or beyond its guard), frames and rays equal the oracle's."""
    desc = scenes.SceneDesc("odd", width=48, height=40, depth=1)
    desc.tri((-3, -3, 6), (3, -3, 6), (0, 4, 6), (255, 128, 3), 0.0)
    desc.tri((1, 1, 5), (1, 1, 5), (2, 2, 5), (10, 255, 10), 0.0)                    # zero area
    desc.tri((-1e18, -1e18, 7), (1e18, -1e18, 7), (0, 1e18, 7), (10, 20, 255), 0.0)  # huge
    desc.tri((0.5, 0.5, 4), (0.5 + 1e-30, 0.5, 4), (0.5, 0.5 + 1e-30, 4), (255, 255, 0), 0.0)   # vanishing
    desc.light((0.5, 0.25, 2.0), 400.0)
    desc.cam_pos, desc.cam_rot = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
    want = oracle.render_desc(desc)
    tr = api.camera_trans(desc.cam_rot)
    tris28, _ = oracle.build_scene(desc)
    rng = np.random.default_rng(3)
    rays = np.concatenate([rng.normal(size=(400, 3)) * 2, rng.normal(size=(400, 3)) * np.array([1, 1, 0.2]) + np.array([0, 0, 1.0]), -np.ones((400, 1))], axis=1).astype(np.float32)
    rays[::7, 3:6] *= 1e20
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        assert np.array_equal(_bits(sc.render(desc.cam_pos, tr, 48, 40, 1)), _bits(want))
        _check(oracle, sc, tris28, rays, "brute")
    assert want.any()
