"""The C++ headless harness (raytracing_amd/host/rt_headless, links only the C ABI): the stand-in
for the reference's main() renders a scene file along a key-scripted camera walk
(Source.cpp:1126-1310 without the viewer) and dumps the last frame; it must equal the oracle's
frame for the pose the reference's camera has at that point."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from raytracing_amd import api, scenes

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "raytracing_amd", "host", "rt_headless")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_headless_walk_equals_oracle(tmp_path, oracle):
    assert os.path.exists(EXE), "build raytracing_amd/host first (__graft_entry__.build)"
    desc = scenes.room(96, 64)
    scene = tmp_path / "scene.txt"
    scene.write_text(desc.to_text())
    frames = [(16.6, [0, 0, 0, 0, 0, 0, 0, 0, 1, 0]), (33.3, [1, 0, 0, 0, 0, 0, 0, 0, 1, 0]), (250.0, [0, 0, 1, 0, 0, 1, 1, 0, 0, 0]),
              (100.0, [0, 1, 0, 1, 1, 0, 0, 0, 0, 1]), (16.6, [0] * 10)]
    walk = tmp_path / "walk.txt"
    walk.write_text("".join("frame %.9g %s\n" % (ms, " ".join(map(str, k))) for ms, k in frames))
    raw, txt = tmp_path / "out.f32", tmp_path / "Image.txt"
    res = subprocess.run([EXE, "--scene", str(scene), "--walk", str(walk), "--raw", str(raw), "--txt", str(txt)],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert res.stdout.count(" FPS") == len(frames)
    got = np.fromfile(raw, dtype=np.float32).reshape(desc.height, desc.width, 3)

    # the pose of the LAST frame drawn: four steps from the scene's camera
    cam = api.Camera(desc.cam_pos, desc.cam_rot)
    bits = [api.KEY_LOOK_LEFT, api.KEY_LOOK_RIGHT, api.KEY_LOOK_UP, api.KEY_LOOK_DOWN, api.KEY_LEFT, api.KEY_RIGHT,
            api.KEY_UP, api.KEY_DOWN, api.KEY_FORWARD, api.KEY_BACKWARD]
    for ms, k in frames[:-1]:
        cam.step(sum(b for b, on in zip(bits, k) if on), ms)
    tris, lights = oracle.build_scene(desc)
    want = oracle.render(tris, lights, cam.position, cam.trans(), desc.width, desc.height, desc.depth)
    assert np.array_equal(_bits(got), _bits(want))
    assert not np.array_equal(_bits(want), _bits(oracle.render_desc(desc)))      # the camera really moved

    # the same walk through the pipelined hand-off (rt_render_async into two pinned frames) and with a loaded model
    raw2 = tmp_path / "out_async.f32"
    res = subprocess.run([EXE, "--scene", str(scene), "--walk", str(walk), "--raw", str(raw2), "--async"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert np.array_equal(_bits(np.fromfile(raw2, dtype=np.float32).reshape(desc.height, desc.width, 3)), _bits(want))
    # the same walk with 8-bit frames (RT_OUT_RGBA8) through the pipelined hand-off: the P6 image is the quantised oracle frame
    ppm8 = tmp_path / "out8.ppm"
    res = subprocess.run([EXE, "--scene", str(scene), "--walk", str(walk), "--ppm", str(ppm8), "--async", "--rgba8"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    blob = ppm8.read_bytes()
    head = ("P6\n%d %d\n255\n" % (desc.width, desc.height)).encode()
    assert blob.startswith(head)
    px8 = np.frombuffer(blob[len(head):], dtype=np.uint8).reshape(desc.height, desc.width, 3)
    assert np.array_equal(px8[::-1], oracle.quantize(want).reshape(desc.height, desc.width, 3))
    # the same walk as the reference's compute shader would colour it (RT_SHADE_GLSL; its restatement is unpinned)
    raw3 = tmp_path / "out_glsl.f32"
    res = subprocess.run([EXE, "--scene", str(scene), "--walk", str(walk), "--raw", str(raw3), "--glsl"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    want_glsl = oracle.render_glsl(tris, lights, cam.position, cam.trans(), desc.width, desc.height)
    assert np.array_equal(_bits(np.fromfile(raw3, dtype=np.float32).reshape(desc.height, desc.width, 3)), _bits(want_glsl))
    assert not np.array_equal(_bits(want_glsl), _bits(want))
    bad = subprocess.run([EXE, "--scene", str(scene), "--size", "-5", "64"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "not usable" in bad.stderr                     # (used to abort on the allocation)

    # Image.txt: P3, top row first, GL unorm quantisation
    tok = txt.read_text().split()
    assert tok[:4] == ["P3", str(desc.width), str(desc.height), "255"]
    px = np.array(tok[4:], dtype=np.int64).reshape(desc.height, desc.width, 3)
    assert np.array_equal(px[::-1], oracle.quantize(want).astype(np.int64))


def test_bench_row_tile_mode_two_ranks_on_one_gpu():
    """The N > 1 default of bench.py — BASELINE config 5: interleaved 16-row tiles of 7680x4320 frames,
    rendered by rt_render_tiles_device on every rank and gathered on rank 0 by the pipelined exchange —
    rehearsed with two ranks sharing this one GPU (gloo carries the shards; on a node it is RCCL): the
    line must carry the strong-scaling fields and the gathered frame must be the single-GPU frame."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "row tiles x2"
    assert d["config"]["width"] == 7680 and d["config"]["height"] == 4320
    assert d["gather_check"].startswith("bit-identical") and d["gather_check"].endswith("= unorm8 of the binary32 frame")
    assert d["exchange"]["format"] == "rgba8" and 60 < d["exchange"]["shard_MB"] < 70          # 7680 x 2160 rows x 4 B
    assert d["exchange_other_format"]["format"] == "f32" and d["exchange_other_format"]["shard_MB"] > 190
    assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel_ms"] > 0
    assert d["single_gpu"]["ms_per_frame"] > 0 and d["value"] > 0


def test_device_props_are_what_the_roofline_assumes():
    """rt_device_props: the compute units and engine clock bench.py records next to the FP32 VALU peak it uses."""
    import ctypes as C
    cus, khz = C.c_int(0), C.c_int(0)
    assert api.lib().rt_device_props(0, C.byref(cus), C.byref(khz)) == 0
    assert cus.value > 0 and khz.value > 0
    assert api.lib().rt_device_props(0, None, None) == 0
    assert api.lib().rt_device_props(1 << 20, C.byref(cus), C.byref(khz)) != 0
