"""The bound behind the edge-term sign filter (raytracing_amd/csrc/rt_scene.cpp edge_filter_planes, rt_kernels.hip
edge_filter), without a GPU: for triangles of many sizes and shapes and hit points on, next to and far from their edge
lines, the kernels' expression s_k — evaluated here with exactly rounded binary32 fused multiply-adds — must have the
sign of the reference's dot(n, cross(edge_k, I - p_k)) (Source.cpp:392-394, evaluated in binary32 in glm's order)
whenever |s_k| exceeds the bound B = max(G |w|_1 + H, 1e-20) the kernels compare it with."""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest

from raytracing_amd import api

f32 = np.float32


def rn32(x: Fraction) -> np.float32:
    """Fraction -> nearest binary32, ties to even (normal and subnormal range; the tests stay far from overflow)."""
    if x == 0:
        return f32(0.0)
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    q = a / ulp
    n = q.numerator // q.denominator
    r = q - n
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and n % 2 == 1):
        n += 1
    v = float(n * ulp)                       # exactly representable: n < 2^25, a power-of-two ulp
    return f32(-v if x < 0 else v)


def fma(a, b, c) -> np.float32:
    return rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def planes(tri28):
    f = api.lib().rt_internal_edge_filter_planes
    f.argtypes = [C.c_void_p, C.c_void_p]
    f.restype = None
    t = np.ascontiguousarray(tri28, dtype=np.float32)
    out = np.zeros(12, np.float32)
    f(t.ctypes.data, out.ctypes.data)
    return out


def reference_edge(n, e, I, p):
    """dot(n, cross(e, I - p)) in binary32, glm's order (func_geometric.inl:52-53, 74-77)."""
    w = [f32(I[k] - p[k]) for k in range(3)]
    c = [f32(f32(e[1] * w[2]) - f32(w[1] * e[2])), f32(f32(e[2] * w[0]) - f32(w[2] * e[0])), f32(f32(e[0] * w[1]) - f32(w[0] * e[1]))]
    return f32(f32(f32(n[0] * c[0]) + f32(n[1] * c[1])) + f32(n[2] * c[2]))


def filter_terms(m, I, p1):
    w = [f32(I[k] - p1[k]) for k in range(3)]
    w1 = f32(f32(abs(w[0]) + abs(w[1])) + abs(w[2]))
    B0 = fma(m[3], w1, m[11])
    B = max(B0, f32(1e-20))
    s0 = fma(m[0], w[0], fma(m[1], w[1], f32(m[2] * w[2])))
    s1 = fma(m[4], w[0], fma(m[5], w[1], fma(m[6], w[2], -m[7])))
    s2 = fma(m[8], w[0], fma(m[9], w[1], f32(m[10] * w[2])))
    return (s0, s1, s2), B0, B


def _cases(oracle, rng, n_tris, scale, needle):
    for _ in range(n_tris):
        c = rng.uniform(-4, 4, 3) * scale
        s = scale * 10.0 ** rng.uniform(-3, 0)
        a, b = rng.normal(size=3), rng.normal(size=3)
        a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
        if needle and rng.random() < 0.5:
            b = a + b * 10.0 ** rng.uniform(-5, -1)
        yield oracle.triangle_init(c, c + a * s, c + b * s, (1, 2, 3), 0.0), s


@pytest.mark.parametrize("scale", [1e-5, 1.0, 1e4])
def test_sign_of_the_filter_term_is_the_reference_sign_beyond_the_bound(oracle, scale):
    rng = np.random.default_rng(int(np.log10(scale)) + 40)
    decided = undecided = 0
    with np.errstate(over="ignore", invalid="ignore"):
        for T, size in _cases(oracle, rng, 40, scale, needle=True):
            n, P, E = T[3:6], [T[6:9], T[9:12], T[12:15]], [T[15:18], T[18:21], T[21:24]]
            m = planes(T)
            if not np.isfinite(n).all():                        # a needle so thin that its binary32 normal is not finite: filter off
                assert np.isnan(m[3])
                continue
            assert np.isfinite(m).all()
            for _ in range(30):
                k = rng.integers(0, 3)
                a, b = P[k].astype(np.float64), P[(k + 1) % 3].astype(np.float64)
                u = rng.choice([0.0, 1.0, 0.5, rng.random(), rng.uniform(-1, 2)])
                I = a + u * (b - a)
                kind = rng.integers(0, 4)
                if kind == 1:                                   # a few ulps off the edge line
                    I = I + rng.integers(-4, 5, 3) * np.spacing(f32(np.abs(I).max() + 1e-30)).astype(np.float64)
                elif kind == 2:                                 # a little inside or outside
                    I = I + rng.normal(size=3) * size * 10.0 ** rng.uniform(-7, -2)
                elif kind == 3:                                 # anywhere near the triangle, or far from it
                    I = I + rng.normal(size=3) * size * 10.0 ** rng.uniform(-1, 3)
                I = I.astype(np.float32)
                ref = [reference_edge(n, E[j], I, P[j]) for j in range(3)]
                s, B0, B = filter_terms(m, I, P[0])
                assert B0 < f32(1e20)
                least = min(s)
                if least > B:
                    assert all(r > 0 for r in ref), (T, I, s, B, ref)
                    decided += 1
                elif -least > B:
                    assert not all(r > 0 for r in ref), (T, I, s, B, ref)
                    decided += 1
                else:
                    undecided += 1
                # and term by term (stronger than what the kernels use): beyond the bound the signs agree
                for j in range(3):
                    if abs(s[j]) > B:
                        assert (s[j] > 0) == (ref[j] > 0), (j, T, I, s, B, ref)
    assert decided > 400 and undecided > 50                     # both sides of the bound were visited


def test_filter_is_off_where_its_bound_would_not_hold(oracle):
    """Zero-area triangles (normal NaN), edges below 1e-24 and beyond 1e30: G = NaN, so no lane is ever decided by the filter."""
    for p1, p2, p3 in [((1, 1, 5), (1, 1, 5), (2, 2, 5)), ((0.5, 0.5, 4), (0.5 + 1e-30, 0.5, 4), (0.5, 0.5 + 1e-30, 4)),
                       ((-1e33, -1e33, 7), (1e33, -1e33, 7), (0, 1e33, 7))]:
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            m = planes(oracle.triangle_init(p1, p2, p3, (1, 2, 3), 0.0))
        assert np.isnan(m[3]), (p1, m)
    m = planes(oracle.triangle_init((-1, -1, 0), (-1, 1, 0), (1, -1, 0), (1, 2, 3), 0.0))
    assert np.isfinite(m).all() and 1e-7 < m[3] < 1e-5 and m[11] > 0
