"""The C-ABI library on a machine without a GPU: it loads, exports everything include/rt355.h
declares, builds scenes with the reference's bits (host arithmetic: Triangle ctor Source.cpp:45-59,
addTriangle Source.cpp:410-439, trans Source.cpp:1145-1147, loadModel Source.cpp:631-657) and fails
loudly — never falls back — when asked to render without a device."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_golden, load_rto_golden, rto_cases
from raytracing_amd import api, scenes


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.SYMBOLS), (declared ^ set(api.SYMBOLS))
    L = api.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.rt_version()


@pytest.mark.parametrize("case", golden_cases())
def test_scene_records_bit_exact(case):
    g = load_golden(case)
    with api.Scene.from_desc(g["desc"]) as sc:
        assert sc.n_triangles == g["tris"].shape[0]
        assert sc.n_lights == g["desc"].n_lights
        assert np.array_equal(_bits(sc.triangles()), _bits(g["tris"]))
    assert np.array_equal(_bits(api.camera_trans(g["desc"].cam_rot)), _bits(g["trans"]))


def test_bulk_add_equals_single_add():
    p1, p2, p3, col, sh = scenes.spheres_arrays(1, 1)
    with api.Scene() as a, api.Scene() as b:
        a.addTriangles(p1, p2, p3, col, sh)
        for i in range(len(p1)):
            b.Triangle(p1[i], p2[i], p3[i], col[i], sh[i])
        assert np.array_equal(_bits(a.triangles()), _bits(b.triangles()))


def test_error_codes_not_exits():
    L = api.lib()
    assert L.rt_scene_clear(None) == -1
    assert b"NULL" in L.rt_last_error()
    assert L.rt_scene_triangle_count(None) == -1
    with api.Scene() as sc:
        sc.addTriangle((0, 0, 10), (0, 0, 0), (10, 10, 10), (255, 0, 0), 0)
        with pytest.raises(api.RtError) as e:
            sc.commit(0, accel=7)
        assert e.value.code == -1
        # rendering before any commit is a state error, whatever the machine
        with pytest.raises(api.RtError) as e:
            sc.render((0, 0, 0), np.eye(4, dtype=np.float32).ravel(), 8, 8)
        assert e.value.code == -2
        with pytest.raises(api.RtError) as e:
            sc.render((0, 0, 0), np.eye(4, dtype=np.float32).ravel(), 0, 8)
        assert e.value.code == -1
        with pytest.raises(api.RtError):
            sc.loadModel("/nonexistent/file.rto")


def test_no_cpu_fallback_without_device():
    if api.device_count() > 0:
        pytest.skip("a HIP device is present")
    with api.Scene.from_desc(scenes.stock(8, 8)) as sc:
        with pytest.raises(api.RtError) as e:
            sc.commit(0)
        assert e.value.code == -3 and "no CPU path" in str(e.value)


def test_shard_rows_partition():
    for h, tr, n in [(800, 16, 8), (2160, 16, 8), (100, 7, 3), (5, 16, 4), (4320, 8, 5)]:
        rows = [api.shard_rows(h, tr, k, n) for k in range(n)]
        assert sum(rows) == h
        assert max(rows) - min(rows) <= tr
    assert api.shard_rows(0, 16, 0, 1) == 0 and api.shard_rows(10, 0, 0, 1) == 0


def test_rto_loader(tmp_path):
    """Source.cpp:631-657: '(t), (r), (s), (r,g,b); ' records -> addTriangle(..., shininess 0)."""
    text = "(0, 0, 1), (0, 0, 0), (1, 1, 1), (255, 255, 255); \n(0, 0, 1), (0, 0, 180), (1, 1, 1), (255, 255, 255); \n"
    f = tmp_path / "cube.rto"
    f.write_text(text)
    with api.Scene() as a, api.Scene() as b:
        assert a.loadModel(str(f)) == 2
        b.addTriangle((0, 0, 1), (0, 0, 0), (1, 1, 1), (255, 255, 255), 0)
        b.addTriangle((0, 0, 1), (0, 0, 180), (1, 1, 1), (255, 255, 255), 0)
        assert np.array_equal(_bits(a.triangles()), _bits(b.triangles()))
    bad = tmp_path / "bad.rto"
    bad.write_text("garbage")
    with api.Scene() as a:
        with pytest.raises(api.RtError) as e:
            a.loadModel(str(bad))
        assert e.value.code == -5


@pytest.mark.parametrize("case", rto_cases())
def test_rto_loader_equals_reference_loadModel(case, tmp_path):
    """rt_scene_load_rto against the reference's OWN loadModel (Source.cpp:631-657) run on the same
    bytes — among them the reference's sample RayTracing/src/cube.rto, written without blanks after
    the commas: the appended Triangle records are the reference's, bit for bit."""
    g = load_rto_golden(case)
    f = tmp_path / "model.rto"
    f.write_bytes(g["rto"])
    with api.Scene.from_desc(g["desc"]) as sc:
        before = sc.n_triangles
        added = sc.loadModel(str(f))
        assert added == g["tris"].shape[0] - before > 0
        got = sc.triangles()
    assert np.array_equal(_bits(got), _bits(g["tris"]))
    assert np.all(got[before:, 27] == 0)                      # loadModel always passes shininess 0 (Source.cpp:653)


def test_quantize_matches_oracle(oracle):
    x = np.array([0.0, 1.0, 0.5, 0.25, 1.5, -0.1, np.nan, 0.999, 1 / 255.0, 0.5 / 255.0], dtype=np.float32)
    assert np.array_equal(api.quantize(x), oracle.quantize(x))
    r = np.random.default_rng(1).random(10000, dtype=np.float32) * 1.2 - 0.1
    assert np.array_equal(api.quantize(r), oracle.quantize(r))


def test_frame_diff_and_imgdiff_tool(tmp_path):
    """the comparison north_star asks for (<= 1 LSB on >= 99.9 % of pixels, flips counted)."""
    import json
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    a = rng.uniform(0.1, 1.0, (40, 50, 3)).astype(np.float32)
    b = a.copy()
    r = api.frame_diff(a, b)
    assert r["bit_identical"] and r["differing"] == 0 and r["flips"] == 0 and r["meets_north_star"]
    b[3, 4] = 0.0                       # one hit -> miss flip
    b[5, 6, 1] += 2.5 / 255.0           # one pixel 2-3 LSB off
    b[7, 8, 2] += 1.0 / 255.0           # one pixel about 1 LSB off
    r = api.frame_diff(a, b)
    assert not r["bit_identical"] and r["flips"] == 1 and r["over_1_lsb"] == 2 and r["differing"] == 3
    assert r["pixels"] == 2000 and r["meets_north_star"] == (1 - 2 / 2000 >= 0.999)
    # the command-line tool on a raw dump and a P6 image of the same frame
    raw = tmp_path / "a.f32"
    a.tofile(raw)
    ppm = tmp_path / "b.ppm"
    with open(ppm, "wb") as f:
        f.write(b"P6\n50 40\n255\n" + api.quantize(b)[::-1].tobytes())
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "imgdiff.py"), str(raw), str(ppm), "--size", "50", "40"],
                         capture_output=True, text=True)
    out = json.loads(res.stdout)
    assert out["flips"] == 1 and out["over_1_lsb"] == 2
    assert res.returncode == (0 if out["meets_north_star"] else 1)
