"""rt_cast_rays_noalloc (include/rt355.h): castRay for a batch of rays with the device staging inside
the library image instead of per-call hipMalloc — for callers that cannot allocate.  Checked against
rt_cast_rays bit for bit, on two calls in a row of which the second is smaller (the buffer keeps the
first call's contents), through brute force and through the LBVH, and at and beyond the size limit."""
import numpy as np
import pytest

from raytracing_amd import api, scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _rays(n, seed, extent=9.5):
    rng = np.random.default_rng(seed)
    r = np.zeros((n, 7), np.float32)
    r[:, 0:3] = rng.uniform(-extent, extent, (n, 3))
    r[:, 3:6] = rng.normal(size=(n, 3))
    r[:, 6] = rng.integers(-1, 24, n)
    return r


@pytest.mark.parametrize("accel", [api.ACCEL_BRUTE, api.ACCEL_LBVH])
def test_noalloc_cast_rays_equal_cast_rays(accel):
    desc = scenes.stock(8, 8) if accel == api.ACCEL_BRUTE else scenes.spheres(4, 2, 8, 8)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, accel)
        big, small = _rays(5000, 1), _rays(777, 2)
        want_big, want_small = sc.cast_rays(big), sc.cast_rays(small)
        got_big = sc.cast_rays_noalloc(big)
        got_small = sc.cast_rays_noalloc(small)             # second call, smaller: stale slots behind it must not leak
        assert np.array_equal(_bits(got_big), _bits(want_big))
        assert np.array_equal(_bits(got_small), _bits(want_small))
        assert 0.3 < want_big[:, 0].mean() <= 1.0
        # the limit itself, then one ray more
        full = _rays(api.NOALLOC_MAX_RAYS, 3)
        assert np.array_equal(_bits(sc.cast_rays_noalloc(full)), _bits(sc.cast_rays(full)))
        with pytest.raises(api.RtError) as e:
            sc.cast_rays_noalloc(_rays(api.NOALLOC_MAX_RAYS + 1, 4))
        assert e.value.code == -6 and "16384" in str(e.value)
        assert sc.cast_rays_noalloc(np.zeros((0, 7), np.float32)).shape == (0, 6)
        # a second scene on the same GPU shares the buffer: sequential calls do not disturb each other
        with api.Scene.from_desc(scenes.room(8, 8)) as other:
            other.commit(0, api.ACCEL_BRUTE)
            w2 = other.cast_rays(small)
            assert np.array_equal(_bits(other.cast_rays_noalloc(small)), _bits(w2))
        assert np.array_equal(_bits(sc.cast_rays_noalloc(small)), _bits(want_small))
