"""The host half of the RT_BUILD_SAH tree build (raytracing_amd/csrc/rt_sah.cpp), without a GPU: the topology it
hands to the refit kernel — leaf order, child references in depth-first preorder, parent links — is a binary
tree over exactly the triangles that are not hoisted, no deeper than the packet stack allows (63), for ordinary
scenes, for scenes whose sizes grade over 15 decades, and for triangles that all share one centre."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import api, scenes


def _topology(sc, hoisted=()):
    n = sc.n_triangles
    h = np.asarray(sorted(hoisted), dtype=np.int32)
    nt = n - len(h)
    keys = np.zeros(nt, np.uint64)
    child = np.zeros(2 * (nt - 1), np.int32)
    pint = np.zeros(nt, np.int32)
    pleaf = np.zeros(nt, np.int32)
    f = api.lib().rt_internal_sah_topology
    f.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 4
    f.restype = C.c_int
    rc = f(sc._h, h.ctypes.data, len(h), keys.ctypes.data, child.ctypes.data, pint.ctypes.data, pleaf.ctypes.data)
    assert rc == 0
    return keys, child.reshape(-1, 2), pint, pleaf


def _check(sc, hoisted=()):
    n = sc.n_triangles
    keys, child, pint, pleaf = _topology(sc, hoisted)
    nt = n - len(hoisted)
    assert sorted(keys.tolist()) == sorted(set(range(n)) - set(hoisted))       # leaf order: a permutation of the tree's triangles
    seen = np.zeros(nt - 1, np.int32)
    leaves = []
    depth = 0
    stack = [(0, 1, -1)]
    while stack:
        node, d, link = stack.pop()
        seen[node] += 1
        depth = max(depth, d)
        assert pint[node] == link                                               # (parent << 1) | side, -1 at the root
        for side in (1, 0):                                                     # (left subtree is popped first: leaves in order)
            r = int(child[node, side])
            if r >= 0:
                assert node < r < nt - 1                                        # preorder: children after their parent
                stack.append((r, d + 1, (node << 1) | side))
            else:
                leaves.append((~r, (node << 1) | side))
    assert np.all(seen == 1) and len(leaves) == nt
    # leaf position p holds triangle keys[p] and links back to the node and side that refer to it
    by_tri = {t: link for t, link in leaves}
    for p in range(nt):
        assert pleaf[p] == by_tri[int(keys[p])]
    return depth


def test_ordinary_scenes():
    for desc, hoisted in ((scenes.stock(8, 8), ()), (scenes.spheres(3, 2, 8, 8), range(12)), (scenes.spheres(20, 3, 8, 8), range(12))):
        with api.Scene.from_desc(desc) as sc:
            depth = _check(sc, tuple(hoisted))
            assert depth <= 2 * int(np.ceil(np.log2(sc.n_triangles))) + 2


def test_threads_give_the_same_tree():
    """Large subtrees are built on separate threads; the result does not depend on who built what."""
    desc = scenes.spheres(20, 4, 8, 8)                                          # 102 412 triangles: above the threading threshold
    with api.Scene.from_desc(desc) as sc:
        a = _topology(sc, tuple(range(12)))
        b = _topology(sc, tuple(range(12)))
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_graded_sizes_and_coincident_centres_stay_within_the_stack():
    rng = np.random.default_rng(5)
    with api.Scene() as sc:
        for i in range(400):                                                    # sizes from 1e-9 to 1e6 around one corner
            s = 10.0 ** (-9 + 15 * i / 399)
            a = rng.uniform(0, 1, 3) * s
            sc.Triangle(a, a + rng.uniform(0, 1, 3) * s, a + rng.uniform(0, 1, 3) * s, (200, 200, 200))
        assert _check(sc) <= 63
    with api.Scene() as sc:
        for i in range(300):                                                    # concentric: every centre is the origin
            s = 1.0 + i
            sc.Triangle((-s, -s, 0), (s, s, 0), (0, 0, 0), (200, 200, 200))
            sc.Triangle((-s, -s, 0), (s, s, 0), (-s, s, 0), (200, 200, 200)) if i % 50 == 0 else None
        assert _check(sc) <= int(np.ceil(np.log2(sc.n_triangles))) + 1          # median splits of the order as it stands
    with api.Scene() as sc:
        sc.Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 2, 3))
        sc.Triangle((0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 2, 3))
        assert _check(sc) == 1
