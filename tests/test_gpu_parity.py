"""Parity tests proper: the HIP render loop, called through the C ABI, against
(1) the committed fixtures the reference's own CPU path produced and (2) the oracle restatement
on the same inputs.  STRICT mode promises the reference's bits, so every comparison here is
bit-exact; the north_star tolerance (<= 1 LSB per 8-bit channel on >= 99.9 % of pixels, flips
counted) is what FAST mode is held to in test_gpu_fast.py."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases, load_golden
from raytracing_amd import api, scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _render(desc, accel=api.ACCEL_BRUTE, mode=api.STRICT):
    return api.render_desc(desc, 0, accel, mode)


@pytest.mark.parametrize("case", golden_cases())
def test_golden_frames_bit_exact(case):
    """getPixelData / castRay / processLighting vs the reference-made fixture."""
    g = load_golden(case)
    frame = _render(g["desc"])
    assert frame.shape == (g["desc"].height, g["desc"].width, 3)
    assert hashlib.sha256(frame.tobytes()).hexdigest() == g["sha256"]
    if g["frame"] is not None:
        assert np.array_equal(_bits(frame), _bits(g["frame"]))
    if g["q8"] is not None:
        assert np.array_equal(api.quantize(frame), g["q8"])


def test_cast_rays_known_answers():
    """castRay (Source.cpp:383-399) on the hand-made + random rays the reference answered."""
    z = np.load(os.path.join(GOLDEN, "rays_stock.npz"), allow_pickle=False)
    desc = scenes.SceneDesc.from_text(bytes(z["scene"]).decode())
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        got = sc.cast_rays(z["rays"])
    want = z["out"]
    assert np.array_equal(got[:, 0], want[:, 0]), "hit/miss decisions differ"
    assert np.array_equal(got[:, 1], want[:, 1]), "hit triangle index differs"
    hit = want[:, 0] > 0
    assert np.array_equal(_bits(got[hit]), _bits(want[hit]))
    miss = ~hit
    assert np.all(got[miss, 2] == np.float32(1e9))


@pytest.mark.parametrize("name,w,h", [("room", 800, 800), ("stock", 800, 800), ("room", 400, 400)])
def test_baseline_configs_vs_oracle(oracle, name, w, h):
    """BASELINE configs 1/2 (and the unmodified stock scene) at full size, against the oracle
    run here on the same inputs."""
    desc = scenes.get(name, w=w, h=h)
    want = oracle.render_desc(desc, threads=8)
    got = _render(desc)
    assert np.array_equal(_bits(got), _bits(want))


def test_mirrors_depth16_vs_oracle(oracle):
    """BASELINE config 3 at 1/4 size (the oracle needs seconds, not minutes): bounce loop trip
    counts 0..16, the dim = bounces quirk (Source.cpp:585) and black mirrors-to-matte."""
    desc = scenes.mirrors(960, 540, 16)
    want = oracle.render_desc(desc, threads=8)
    got = _render(desc)
    assert np.array_equal(_bits(got), _bits(want))
    assert (got.max(axis=2) > 0).mean() > 0.2


def test_work_counters_equal_oracle(oracle):
    """rt_render_stats counts rays and plane/point/edge steps with the reference's per-ray
    short-circuit semantics; the oracle counts the same on the CPU."""
    for desc in (scenes.room(160, 120), scenes.mirrors(128, 72, 16), scenes.two_lights(64, 64)):
        want_frame, want = oracle.render_desc(desc, counters=True)
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, api.ACCEL_BRUTE)
            frame, st = sc.render_stats(desc.cam_pos, api.camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth)
        assert np.array_equal(_bits(frame), _bits(want_frame))
        got = st.as_dict()
        assert got.pop("n_box") == 0
        assert got == want.as_dict()
        assert st.flops == oracle.flops(want)


def test_row_tiles_equal_full_frame():
    """Interleaved row tiles: shards rendered separately reassemble to the frame."""
    import torch
    desc = scenes.room(200, 150)
    trans = api.camera_trans(desc.cam_rot)
    full = _render(desc)
    for tile_rows, n in [(16, 3), (7, 4), (150, 2), (1, 5)]:
        out = np.zeros_like(full)
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, api.ACCEL_BRUTE)
            for k in range(n):
                rows = api.shard_rows(desc.height, tile_rows, k, n)
                if rows == 0:
                    continue
                buf = torch.zeros(rows * desc.width * 3, dtype=torch.float32, device="cuda:0")
                sc.render_tiles_device(0, desc.cam_pos, trans, desc.width, desc.height, desc.depth, api.STRICT,
                                       tile_rows, k, n, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                shard = buf.cpu().numpy().reshape(rows, desc.width, 3)
                r = 0
                t = k
                while t * tile_rows < desc.height:
                    r0, r1 = t * tile_rows, min((t + 1) * tile_rows, desc.height)
                    out[r0:r1] = shard[r:r + (r1 - r0)]
                    r += r1 - r0
                    t += n
        assert np.array_equal(_bits(out), _bits(full)), (tile_rows, n)


def test_render_multi_single_device():
    desc = scenes.stock(120, 90)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        a = sc.render(desc.cam_pos, api.camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth)
        b = sc.render_multi([0], desc.cam_pos, api.camera_trans(desc.cam_rot), desc.width, desc.height, desc.depth, tile_rows=16)
    assert np.array_equal(_bits(a), _bits(b))


def test_tiles_per_wave_does_not_change_pixels():
    desc = scenes.mirrors(192, 108, 16)
    base = _render(desc)
    try:
        for tpw in (1, 2, 5, 64):
            api.lib().rt_set_tiles_per_wave(tpw)
            assert np.array_equal(_bits(_render(desc)), _bits(base)), tpw
    finally:
        api.lib().rt_set_tiles_per_wave(0)


def test_recommit_after_edit():
    desc = scenes.stock(64, 64)
    trans = api.camera_trans(desc.cam_rot)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0)
        a = sc.render(desc.cam_pos, trans, 64, 64)
        sc.addTriangle((0, 6, 0), (90, 0, 0), (1, 1, 1), (0, 0, 255), 0)
        with pytest.raises(api.RtError) as e:
            sc.render(desc.cam_pos, trans, 64, 64)
        assert e.value.code == -2
        sc.commit(0)
        b = sc.render(desc.cam_pos, trans, 64, 64)
    assert not np.array_equal(a, b)


def test_frame_batch_equals_single_frames():
    """rt_render_frames_device: several camera poses in one launch give the same bits as one
    rt_render per pose (20 poses = two launches of up to RT_MAX_FRAMES_PER_LAUNCH)."""
    import torch
    desc = scenes.room(120, 88)
    poses = scenes.camera_sweep(20, desc.cam_pos, desc.cam_rot)
    cams = np.array([p for p, _ in poses], np.float32)
    trs = np.stack([api.camera_trans(r) for _, r in poses])
    n = desc.width * desc.height * 3
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        singles = [sc.render(cams[i], trs[i], desc.width, desc.height, desc.depth) for i in range(20)]
        for count, stride in [(1, n), (3, n + 64), (20, n)]:
            buf = torch.full((count * stride,), -1.0, dtype=torch.float32, device="cuda:0")
            sc.render_frames_device(0, cams[:count], trs[:count], desc.width, desc.height, desc.depth, api.STRICT,
                                    buf.data_ptr(), stride, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = buf.cpu().numpy().reshape(count, stride)
            for i in range(count):
                assert np.array_equal(_bits(got[i, :n].reshape(desc.height, desc.width, 3)), _bits(singles[i])), (count, i)
                assert np.all(got[i, n:] == -1.0)
    assert not np.array_equal(singles[0], singles[19])
