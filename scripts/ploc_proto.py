"""CPU prototype of the PLOC tree build (Meister & Bittner 2018: parallel locally-ordered clustering over the
Morton order) — the figure of merit (sum of child-box areas over the root's: expected box tests per random ray)
for the 102k-triangle scene at several search radii, next to the host SAH build's, without a GPU.  The GPU
kernels in raytracing_amd/csrc/rt_ploc.hip follow this file step by step."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_amd import api, scenes


def area(lo, hi):
    e = hi - lo
    return e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]


def morton_order(tlo, thi):
    c = 0.5 * (tlo + thi)
    lo, hi = tlo.min(0), thi.max(0)
    u = np.clip((c - lo) / (hi - lo) * 2097152.0, 0, 2097151).astype(np.uint64)
    def expand(v):
        v = (v | (v << np.uint64(32))) & np.uint64(0x001f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x001f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    code = (expand(u[:, 0]) << np.uint64(2)) | (expand(u[:, 1]) << np.uint64(1)) | expand(u[:, 2])
    return np.argsort(code, kind="stable")


def ploc(tlo, thi, radius):
    """-> (sum of child areas / root area, iterations, depth)"""
    order = morton_order(tlo, thi)
    lo, hi = tlo[order].copy(), thi[order].copy()
    n = len(lo)
    depth = np.zeros(n, np.int32)
    total = 0.0
    it = 0
    while n > 1:
        it += 1
        best = np.full(n, np.inf)
        nn = np.full(n, -1, np.int64)
        for d in range(1, min(radius, n - 1) + 1):
            a = area(np.minimum(lo[:-d], lo[d:]), np.maximum(hi[:-d], hi[d:]))
            # cluster i with neighbour i + d: ties go to the lower neighbour index — the left neighbour was seen first
            idx = np.arange(n - d)
            better = a < best[idx]
            best[idx[better]] = a[better]; nn[idx[better]] = idx[better] + d
            # cluster i + d with neighbour i (a lower index than anything recorded at distance < d on the right, so <=)
            j = idx + d
            better = (a < best[j]) | ((a == best[j]) & (idx < nn[j]))
            best[j[better]] = a[better]; nn[j[better]] = idx[better]
        i = np.arange(n)
        mutual = nn[nn] == i
        first = mutual & (i < nn)
        gone = mutual & (i > nn)
        p = i[first]; q = nn[first]
        total += (area(lo[p], hi[p]) + area(lo[q], hi[q])).sum()
        nlo, nhi = np.minimum(lo[p], lo[q]), np.maximum(hi[p], hi[q])
        lo[p], hi[p] = nlo, nhi
        depth[p] = np.maximum(depth[p], depth[q]) + 1
        keep = ~gone
        lo, hi, depth = lo[keep], hi[keep], depth[keep]
        n = len(lo)
    return total / area(lo[0], hi[0]), it, int(depth[0])


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "big"
    desc = scenes.spheres(20, 4, 8, 8) if which == "big" else scenes.spheres(6, 3, 8, 8)
    with api.Scene.from_desc(desc) as sc:
        T = sc.triangles()[12:]
    P = T[:, 6:15].reshape(-1, 3, 3).astype(np.float64)
    tlo, thi = P.min(1), P.max(1)
    for r in (4, 8, 16, 32, 64):
        t = time.perf_counter()
        c, it, depth = ploc(tlo, thi, r)
        print("radius %3d: cost %.3f, %d iterations, depth %d (%.1f s)" % (r, c, it, depth, time.perf_counter() - t))
