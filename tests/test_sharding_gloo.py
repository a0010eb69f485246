"""Multi-process path on CPU (gloo, world_size 2 and 3): the sharding/gather logic of
raytracing_amd/sharding.py with the ORACLE standing in for the per-rank render (there is no GPU
here; on the GPU box the same functions move rt_render_tiles_device output over RCCL).
SURVEY.md §4 T5: the N-rank interleaved-row frame equals the 1-rank frame byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from raytracing_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, tile_rows, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from raytracing_amd import scenes
    desc = scenes.room(width, height)
    tris, lights = pyoracle.build_scene(desc)
    trans = pyoracle.camera_trans(desc.cam_rot)
    # this rank's tiles, rendered by the checker, packed compactly like rt_render_tiles_device does
    parts = []
    for r0, r1 in sharding.shard_tiles(height, tile_rows, rank, world):
        f = pyoracle.render(tris, lights, desc.cam_pos, trans, width, height, desc.depth, rows=(r0, r1, 1))
        parts.append(f[r0:r1].reshape(-1))
    shard = torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.float32))
    assert shard.numel() == sharding.shard_rows(height, tile_rows, rank, world) * width * 3
    full = torch.zeros(height * width * 3) if rank == 0 else None
    dist.barrier()
    got = sharding.gather_frame(dist, shard, full, rank, world, height, width, tile_rows)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        np.save(out_path, got.numpy().reshape(height, width, 3))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,width,tile_rows", [(2, 48, 40, 16), (2, 50, 24, 7), (3, 37, 16, 4)])
def test_row_tile_gather_equals_single_rank(oracle, tmp_path, world, height, width, tile_rows):
    from raytracing_amd import scenes
    out_path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, tile_rows, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    want = oracle.render_desc(scenes.room(width, height))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _pipeline_worker(rank, world, port, height, width, tile_rows, depth, n_frames, out_path, rgba8=False, collective=True):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from raytracing_amd import api, scenes
    desc = scenes.room(width, height)
    tris, lights = pyoracle.build_scene(desc)
    cams, trs = api.camera_sweep(n_frames, desc.cam_pos, desc.cam_rot)
    if rgba8:     # OUT_RGBA8 frames travel as one 32-bit element per pixel
        ex = sharding.TileExchange(dist, rank, world, height, width, tile_rows, lambda n: torch.zeros(n, dtype=torch.int32), depth=depth, channels=1)
    else:
        ex = sharding.TileExchange(dist, rank, world, height, width, tile_rows, lambda n: torch.zeros(n), depth=depth, collective=collective)
    frames = {}

    def keep(done):
        for f, full in done:
            assert f not in frames
            frames[f] = full.numpy().reshape(height, width, 1 if rgba8 else 3).copy()    # the buffer is reused `depth` frames later

    for f in range(n_frames):
        buf = ex.shard(f)
        parts = []
        for r0, r1 in sharding.shard_tiles(height, tile_rows, rank, world):
            img = pyoracle.render(tris, lights, cams[f], trs[f], width, height, desc.depth, rows=(r0, r1, 1))
            part = img[r0:r1]
            if rgba8:
                q = pyoracle.quantize(part).reshape(r1 - r0, width, 3)
                part = np.concatenate([q, np.full((r1 - r0, width, 1), 255, np.uint8)], axis=2).view(np.int32)
            parts.append(part.reshape(-1))
        if parts:
            buf.copy_(torch.from_numpy(np.concatenate(parts)))
        keep(ex.submit(f))
    keep(ex.drain())
    if rank == 0:
        assert sorted(frames) == list(range(n_frames))
        np.save(out_path, np.stack([frames[f] for f in range(n_frames)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,depth,collective", [(2, 2, True), (3, 2, True), (2, 1, True), (2, 3, True), (3, 2, False)])
def test_pipelined_tile_exchange_equals_single_rank(oracle, tmp_path, world, depth, collective):
    """TileExchange: shards of frame f+1 are produced while frame f is in flight; every completed
    frame equals the single-rank frame of its pose, in order, whatever the pipeline depth — by one gather per
    frame, and by point-to-point sends (the fallback)."""
    from raytracing_amd import api, scenes
    height, width, tile_rows, n_frames = 40, 24, 8, 5
    out_path = str(tmp_path / "frames.npy")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), height, width, tile_rows, depth, n_frames, out_path, False, collective), nprocs=world, join=True)
    got = np.load(out_path)
    desc = scenes.room(width, height)
    tris, lights = oracle.build_scene(desc)
    cams, trs = api.camera_sweep(n_frames, desc.cam_pos, desc.cam_rot)
    for f in range(n_frames):
        want = oracle.render(tris, lights, cams[f], trs[f], width, height, desc.depth)
        assert np.array_equal(got[f].view(np.uint32), want.view(np.uint32)), f


def test_pipelined_exchange_of_8bit_frames(oracle, tmp_path):
    """The same exchange with RT_OUT_RGBA8 frames (one 32-bit element per pixel, a third of the bytes): every
    gathered frame is the quantised single-rank frame."""
    from raytracing_amd import api, scenes
    height, width, tile_rows, n_frames, world = 40, 24, 8, 4, 3
    out_path = str(tmp_path / "frames8.npy")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), height, width, tile_rows, 2, n_frames, out_path, True), nprocs=world, join=True)
    got = np.load(out_path)
    desc = scenes.room(width, height)
    tris, lights = oracle.build_scene(desc)
    cams, trs = api.camera_sweep(n_frames, desc.cam_pos, desc.cam_rot)
    for f in range(n_frames):
        want = oracle.quantize(oracle.render(tris, lights, cams[f], trs[f], width, height, desc.depth)).reshape(height, width, 3)
        g = got[f].view(np.uint8).reshape(height, width, 4)
        assert np.array_equal(g[..., :3], want) and (g[..., 3] == 255).all(), f


def test_tile_partition_matches_c_abi():
    from raytracing_amd import api
    for h, tr, n in [(800, 16, 8), (4320, 16, 8), (100, 7, 3), (5, 16, 4)]:
        rows = [sharding.shard_rows(h, tr, k, n) for k in range(n)]
        assert rows == [api.shard_rows(h, tr, k, n) for k in range(n)]
        covered = sorted(r for k in range(n) for t in sharding.shard_tiles(h, tr, k, n) for r in range(*t))
        assert covered == list(range(h))


def test_scatter_numpy_all_shapes():
    rng = np.random.default_rng(0)
    for h, w, tr, n in [(64, 5, 8, 4), (70, 3, 8, 4), (9, 2, 16, 2), (33, 4, 1, 5)]:
        want = rng.random((h, w * 3)).astype(np.float32)
        full = np.zeros(h * w * 3, np.float32)
        for k in range(n):
            tiles = sharding.shard_tiles(h, tr, k, n)
            shard = np.concatenate([want[a:b].reshape(-1) for a, b in tiles]) if tiles else np.zeros(0, np.float32)
            sharding.scatter_shard_into_frame(full, shard, h, w, tr, k, n)
        assert np.array_equal(full.reshape(h, w * 3), want)


def test_scatter_all_shapes_at_once():
    """scatter_all_into_frame: the shards of all ranks, padded to a common pitch as the gather leaves them, land
    in the frame in two or three copies — whole groups, ragged tail, short last tile; numpy and torch alike."""
    rng = np.random.default_rng(1)
    for h, w, tr, n, ch in [(64, 5, 8, 4, 3), (70, 3, 8, 4, 3), (9, 2, 16, 2, 1), (33, 4, 1, 5, 3), (4320, 3, 16, 8, 1),
                            (40, 24, 8, 3, 3), (100, 3, 7, 3, 1), (5, 2, 16, 4, 3)]:
        want = rng.integers(0, 1 << 30, (h, w * ch)).astype(np.int32)
        pitch = max(sharding.shard_rows(h, tr, k, n) for k in range(n)) * w * ch
        recv = np.full(n * pitch, -1, np.int32)
        for k in range(n):
            tiles = sharding.shard_tiles(h, tr, k, n)
            sh = np.concatenate([want[a:b].reshape(-1) for a, b in tiles]) if tiles else np.zeros(0, np.int32)
            recv[k * pitch: k * pitch + sh.size] = sh
        full = np.zeros(h * w * ch, np.int32)
        sharding.scatter_all_into_frame(full, recv, h, w, tr, n, ch, pitch)
        assert np.array_equal(full.reshape(h, w * ch), want), (h, w, tr, n, ch)
        tfull = torch.zeros(h * w * ch, dtype=torch.int32)
        sharding.scatter_all_into_frame(tfull, torch.from_numpy(recv), h, w, tr, n, ch, pitch)
        assert np.array_equal(tfull.numpy().reshape(h, w * ch), want), (h, w, tr, n, ch)
        # the library's own de-interleave (rt_scatter_shards_host: what TileExchange calls for host buffers)
        for make in (lambda a: a, torch.from_numpy):
            nfull = np.zeros(h * w * ch, np.int32)
            sharding.scatter_all_native(make(nfull), make(recv), h, w, tr, n, ch, pitch)
            assert np.array_equal(nfull.reshape(h, w * ch), want), (h, w, tr, n, ch)


def test_scatter_host_rejects_bad_arguments():
    from raytracing_amd import api
    a = np.zeros(64, np.int32)
    with pytest.raises(api.RtError):
        api.scatter_shards_host(a.ctypes.data, 4, 2, 4, 8, 2, api.STRICT | api.OUT_RGBA8, a.ctypes.data)      # pitch below the largest shard
    with pytest.raises(api.RtError):
        api.scatter_shards_host(a.ctypes.data, 64, 0, 4, 8, 2, api.STRICT, a.ctypes.data)
    with pytest.raises(api.RtError):
        api.scatter_shards_host(0, 64, 2, 4, 8, 2, api.STRICT, a.ctypes.data)
    with pytest.raises(api.RtError):
        api.scatter_shards_host(a.ctypes.data, 64, 2, 4, 8, 2, 0x400, a.ctypes.data)


def test_frames_round_robin():
    a = sharding.frames_for_rank(5, 4, 1)
    assert a == [1, 5, 9, 13, 17]
    seen = sorted(f for r in range(8) for f in sharding.frames_for_rank(8, 8, r))
    assert seen == list(range(64))
