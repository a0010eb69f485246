"""Camera controller: the reference's Camera (Source.cpp:95-212) and the tail of
its frame loop (rotate, translate: Source.cpp:1294-1295; speeds from the frame time:
Source.cpp:1307-1308).  tests/golden/camera_path.npz holds what the REFERENCE's own class did for a
600-frame key script (made by tests/golden/make_golden.py from oracle/_ref); the oracle's
restatement and the library's host code must both reproduce it bit for bit."""
import os

import numpy as np
import pytest

from conftest import ROOT
from raytracing_amd import api, scenes


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(ROOT, "tests", "golden", "camera_path.npz"))
    return {k: g[k] for k in g.files}


def _mask(k10):
    bits = [api.KEY_LOOK_LEFT, api.KEY_LOOK_RIGHT, api.KEY_LOOK_UP, api.KEY_LOOK_DOWN, api.KEY_LEFT, api.KEY_RIGHT,
            api.KEY_UP, api.KEY_DOWN, api.KEY_FORWARD, api.KEY_BACKWARD]
    return sum(b for b, on in zip(bits, k10) if on)


def test_golden_covers_the_cases(golden):
    out, keys = golden["out"], golden["keys"]
    assert out.shape == (600, 33) and keys.shape == (600, 10)
    assert (out[:, 4] == np.float32(89.99)).any() and (out[:, 4] == np.float32(-89.99)).any()   # both clamps
    assert (keys.sum(axis=1) == 0).any() and (keys.sum(axis=1) == 10).any()                      # idle, everything at once
    assert np.isfinite(out).all()


def test_oracle_camera_equals_reference(golden, oracle):
    got = oracle.camera_run(golden["start"], golden["ms"], golden["keys"])
    assert np.array_equal(_bits(got), _bits(golden["out"]))


def test_library_camera_equals_reference(golden):
    cam = api.Camera(golden["start"][:3], golden["start"][3:])
    for i in range(len(golden["ms"])):
        cam.step(_mask(golden["keys"][i]), golden["ms"][i])
        assert np.array_equal(_bits(cam.state()), _bits(golden["out"][i, :17])), i
        assert np.array_equal(_bits(cam.trans()), _bits(golden["out"][i, 17:])), i


def test_separate_calls_equal_step(golden):
    a = api.Camera(golden["start"][:3], golden["start"][3:])
    b = api.Camera(golden["start"][:3], golden["start"][3:])
    for i in range(200):
        k = golden["keys"][i]
        a.step(_mask(k), golden["ms"][i])
        b.rotate(*[bool(x) for x in k[:4]]).translate(*[bool(x) for x in k[4:]]).frame_time(golden["ms"][i])
        assert np.array_equal(_bits(a.state()), _bits(b.state()))


def test_path_records_the_pose_each_frame_is_drawn_with(golden):
    cam = api.Camera(golden["start"][:3], golden["start"][3:])
    masks = [_mask(k) for k in golden["keys"]]
    pos, tr = cam.path(masks, golden["ms"])
    assert np.array_equal(_bits(pos[0]), _bits(golden["start"][:3]))
    assert np.array_equal(_bits(tr[0]), _bits(api.camera_trans(golden["start"][3:])))
    assert np.array_equal(_bits(pos[1:]), _bits(golden["out"][:-1, 0:3]))
    assert np.array_equal(_bits(tr[1:]), _bits(golden["out"][:-1, 17:]))
    assert np.array_equal(_bits(cam.state()), _bits(golden["out"][-1, :17]))


def test_bench_sweep_is_the_pinned_controller(oracle):
    """api.camera_sweep (the 64-pose path of bench.py and BASELINE config 5) is rt_camera_path with look
    right + forward held down and 16 ms frames: the same bits as the oracle's restatement of the
    reference's Camera, which the fixture above pins to the reference's own class."""
    start = np.array([-5, 5, 5, 135, -45, 0], np.float32)
    pos, tr = api.camera_sweep(64, start[:3], start[3:])
    keys = np.zeros((64, 10), np.int32)
    keys[:, 1] = 1          # look right
    keys[:, 8] = 1          # move forward
    out = oracle.camera_run(start, np.full(64, 16.0, np.float32), keys)
    assert np.array_equal(_bits(pos[0]), _bits(start[:3]))
    assert np.array_equal(_bits(tr[0]), _bits(oracle.camera_trans(start[3:])))
    assert np.array_equal(_bits(pos[1:]), _bits(out[:-1, 0:3]))
    assert np.array_equal(_bits(tr[1:]), _bits(out[:-1, 17:]))
    assert len({tuple(p) for p in pos}) == 64 and not np.array_equal(tr[0], tr[63])


def test_null_arguments_are_errors():
    L = api.lib()
    assert L.rt_camera_rotate(None, 0, 0, 0, 0) == -1
    assert L.rt_camera_translate(None, 0, 0, 0, 0, 0, 0) == -1
    assert L.rt_camera_step(None, 0, 1.0) == -1
    assert L.rt_camera_frame_time(None, 1.0) == -1
    assert L.rt_camera_init(None, None, None) == -1
    assert b"NULL" in L.rt_last_error()
