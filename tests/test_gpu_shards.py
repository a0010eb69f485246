"""The exchange step's device side behind the C ABI: rt_scatter_shards against the tensor-operation form it
replaces, rt_render_tiles (host shards), and rt_render_multi's tile assignment rehearsed on ONE GPU by listing it several
times — the gathered frame is the frame one launch renders, bit for bit, for float and 8-bit pixels."""
import numpy as np
import pytest
import torch

from raytracing_amd import api, scenes, sharding

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("h,w,tr,n,ch", [(64, 5, 8, 4, 3), (70, 3, 8, 4, 3), (9, 2, 16, 2, 1), (33, 4, 1, 5, 3), (4320, 12, 16, 8, 1),
                                          (40, 24, 8, 3, 3), (100, 3, 7, 3, 1), (5, 2, 16, 4, 3), (2160, 3840, 16, 8, 1), (801, 800, 16, 7, 3)])
def test_scatter_kernel_equals_tensor_ops(h, w, tr, n, ch):
    """Wide (16-byte) and narrow (4-byte) rows, ragged heights, more ranks than tiles, padded pitches; a frame-sized case."""
    g = torch.Generator(device="cpu").manual_seed(h * 131 + w)
    rows = [sharding.shard_rows(h, tr, k, n) for k in range(n)]
    for pad in (0, 3):
        pitch = max(rows) * w * ch + pad
        recv = torch.randint(0, 1 << 30, (n * pitch,), dtype=torch.int32, generator=g).cuda()
        want = torch.full((h * w * ch,), -1, dtype=torch.int32, device="cuda")
        sharding.scatter_all_into_frame(want, recv, h, w, tr, n, ch, pitch)
        got = torch.full((h * w * ch,), -2, dtype=torch.int32, device="cuda")
        sharding.scatter_all_native(got, recv, h, w, tr, n, ch, pitch)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (h, w, tr, n, ch, pad)
    with pytest.raises(api.RtError):
        api.scatter_shards(0, recv.data_ptr(), 4, n, w, h, tr, api.STRICT | (api.OUT_RGBA8 if ch == 1 else 0), got.data_ptr())


@pytest.mark.parametrize("name,W,H,tile_rows,devices", [("room", 200, 150, 16, [0, 0]), ("room", 97, 61, 8, [0, 0, 0]), ("stock", 64, 40, 16, [0] * 5),
                                                       ("mirrors", 160, 90, 4, [0] * 8), ("room", 33, 7, 16, [0, 0, 0])])
def test_render_multi_tile_assignment_on_one_gpu(oracle, name, W, H, tile_rows, devices):
    desc = scenes.get(name).with_size(W, H) if name != "mirrors" else scenes.mirrors(W, H, 6)
    tr = api.camera_trans(desc.cam_rot)
    want = oracle.render_desc(desc, threads=4)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        whole = sc.render(desc.cam_pos, tr, W, H, desc.depth)
        assert np.array_equal(_bits(whole), _bits(want))
        multi = sc.render_multi(devices, desc.cam_pos, tr, W, H, desc.depth, tile_rows=tile_rows)
        assert np.array_equal(_bits(multi), _bits(whole))
        q = sc.render(desc.cam_pos, tr, W, H, desc.depth, api.STRICT | api.OUT_RGBA8)
        m8 = sc.render_multi(devices, desc.cam_pos, tr, W, H, desc.depth, api.STRICT | api.OUT_RGBA8, tile_rows=tile_rows)
        assert m8.dtype == np.uint8 and np.array_equal(m8, q)
        # the same shards one by one into host memory (rt_render_tiles), de-interleaved on the host
        n = len(devices)
        for mode, ch, dtype, ref in ((api.STRICT, 3, np.float32, whole), (api.STRICT | api.OUT_RGBA8, 1, np.uint32, q)):
            rows = [api.shard_rows(H, tile_rows, k, n) for k in range(n)]
            pitch = max(rows) * W * ch
            recv = np.zeros(n * pitch, np.uint32)
            for k in range(n):
                sh = sc.render_tiles(desc.cam_pos, tr, W, H, desc.depth, mode, tile_rows, k, n)
                recv[k * pitch: k * pitch + rows[k] * W * ch] = sh.reshape(-1).view(np.uint32)
            full = np.zeros(H * W * ch, np.uint32)
            sharding.scatter_all_native(full, recv, H, W, tile_rows, n, ch, pitch)
            assert np.array_equal(full, np.ascontiguousarray(ref).reshape(-1).view(np.uint32)), (mode,)


def test_render_multi_tree_scene_with_repeated_device():
    """The tree kernels through the same path: 102 412 triangles would be slow to check on the host, so a sphere soup; the shards
    of a repeated device render on streams of their own (each its own hoisted-triangle masks)."""
    desc = scenes.spheres(6, 2, 160, 120, depth=2)
    tr = api.camera_trans(desc.cam_rot)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        whole = sc.render(desc.cam_pos, tr, 160, 120, desc.depth)
        for devices, tile_rows in (([0, 0], 16), ([0] * 4, 8), ([0] * 7, 16)):
            multi = sc.render_multi(devices, desc.cam_pos, tr, 160, 120, desc.depth, tile_rows=tile_rows)
            assert np.array_equal(_bits(multi), _bits(whole)), (devices, tile_rows)
        # many calls: the listings' streams (and their scratch) are kept between calls, and survive a re-commit
        for k in range(12):
            if k == 6:
                sc.commit(0, api.ACCEL_LBVH)
            multi = sc.render_multi([0, 0, 0], desc.cam_pos, tr, 160, 120, desc.depth, tile_rows=8)
            assert np.array_equal(_bits(multi), _bits(whole)), k
