"""RT_OUT_RGBA8: the render entry points can store the 8-bit pixel the reference's viewer ends up displaying
(glTexImage2D converts `data` with the GL unorm rule, Source.cpp:1270; SURVEY 8a fixes it as
q = (int)(c * 255.0f + 0.5f)) instead of three floats — a third of the bytes to hand off or to gather.  Every
such frame must be exactly the oracle's quantisation of the oracle's binary32 frame: R, G, B, 255 per pixel."""
import numpy as np
import pytest

from raytracing_amd import api, scenes, sharding

pytestmark = pytest.mark.gpu
RGBA8 = api.STRICT | api.OUT_RGBA8


def _want(oracle, frame):
    q = oracle.quantize(frame).reshape(frame.shape[0], frame.shape[1], 3)
    return np.concatenate([q, np.full(q.shape[:2] + (1,), 255, np.uint8)], axis=2)


@pytest.mark.parametrize("name,size,depth", [("room", (200, 150), 1), ("stock", (160, 120), 1), ("mirrors", (128, 96), 16)])
def test_rgba8_frames_are_the_quantised_oracle_frames(oracle, name, size, depth):
    desc = (scenes.mirrors(size[0], size[1], depth) if name == "mirrors" else scenes.get(name).with_size(*size))
    want = _want(oracle, oracle.render_desc(desc))
    tr = api.camera_trans(desc.cam_rot)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_BRUTE)
        for kind in (api.KERNEL_LOCKSTEP, api.KERNEL_STATE_MACHINE):
            sc.set_kernel(kind)
            got = sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth, RGBA8)
            assert got.dtype == np.uint8 and got.shape == want.shape
            assert np.array_equal(got, want), kind
        assert np.array_equal(got[..., :3], api.quantize(sc.render(desc.cam_pos, tr, desc.width, desc.height, desc.depth)))
    assert (want[..., :3].max(axis=2) > 0).mean() > 0.2


def test_rgba8_through_the_tree_tiles_batches_and_the_async_hand_off(oracle):
    import torch
    desc = scenes.spheres(3, 2, 96, 64, depth=2)
    want = _want(oracle, oracle.render_desc(desc))
    tr = api.camera_trans(desc.cam_rot)
    W, H = desc.width, desc.height
    with api.Scene.from_desc(desc) as sc:
        sc.commit(0, api.ACCEL_LBVH)
        for mode in (api.TRAVERSAL_PER_LANE, api.TRAVERSAL_PACKET):
            sc.set_traversal(mode)
            assert np.array_equal(sc.render(desc.cam_pos, tr, W, H, desc.depth, RGBA8), want), mode
        # interleaved row tiles of 8-bit pixels, gathered by the sharding helper with one element per pixel
        full = torch.zeros(H * W, dtype=torch.int32, device="cuda:0")
        for rank in range(3):
            rows = api.shard_rows(H, 16, rank, 3)
            shard = torch.empty(rows * W, dtype=torch.int32, device="cuda:0")
            sc.render_tiles_device(0, desc.cam_pos, tr, W, H, desc.depth, RGBA8, 16, rank, 3, shard.data_ptr())
            torch.cuda.synchronize()
            sharding.scatter_shard_into_frame(full, shard, H, W, 16, rank, 3, channels=1)
        assert np.array_equal(full.cpu().numpy().view(np.uint8).reshape(H, W, 4), want)
        # a batch of frames: the stride counts 4-byte pixels
        cams = np.tile(np.asarray(desc.cam_pos, np.float32), (3, 1))
        trs = np.tile(tr.astype(np.float32).reshape(1, 16), (3, 1))
        buf = torch.zeros(3 * (H * W + 5), dtype=torch.int32, device="cuda:0")
        sc.render_frames_device(0, cams, trs, W, H, desc.depth, RGBA8, buf.data_ptr(), H * W + 5)
        torch.cuda.synchronize()
        b = buf.cpu().numpy().reshape(3, H * W + 5)
        for k in range(3):
            assert np.array_equal(b[k, :H * W].view(np.uint8).reshape(H, W, 4), want), k
            assert not b[k, H * W:].any()
        with pytest.raises(api.RtError):
            sc.render_frames_device(0, cams, trs, W, H, desc.depth, RGBA8, buf.data_ptr(), H * W - 1)
        # the pipelined hand-off into a pinned 8-bit frame, and into a pageable one
        hf = api.HostFrame(W, H, rgba8=True)
        sc.render_wait(sc.render_async(desc.cam_pos, tr, W, H, desc.depth, hf))
        assert hf.array.dtype == np.uint8 and np.array_equal(hf.array, want)
        page = np.zeros((H, W, 4), np.uint8)
        sc.render_wait(sc.render_async(desc.cam_pos, tr, W, H, desc.depth, page))
        assert np.array_equal(page, want)
        hf.close()
        # the in-process multi-GPU call gathers 8-bit shards as well (a third of the bytes on every link)
        assert np.array_equal(sc.render_multi([0, 0, 0], desc.cam_pos, tr, W, H, desc.depth, RGBA8, tile_rows=8), want)
        with pytest.raises(api.RtError):
            sc.render(desc.cam_pos, tr, W, H, desc.depth, 0x400)


def test_unorm8_edge_values(oracle):
    """Colours at and around the quantisation steps, above 1 and exactly 0: a one-triangle scene lit so that the
    pixel values sweep the range (intensities from nothing to far beyond saturation)."""
    for k, intensity in enumerate((0.0, 1e-3, 0.37, 2.0, 17.5, 254.6, 1e4, 3e38)):
        desc = scenes.SceneDesc("unorm%d" % k, width=48, height=32, depth=1)
        desc.tri((-30, -30, -6), (30, -30, -6), (0, 40, -6), (255, 128, 3), 0.0)
        desc.light((0.5, 0.25, 2.0), intensity)
        desc.cam_pos, desc.cam_rot = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
        f = oracle.render_desc(desc)
        want = _want(oracle, f)
        with api.Scene.from_desc(desc) as sc:
            sc.commit(0, api.ACCEL_BRUTE)
            got = sc.render(desc.cam_pos, api.camera_trans(desc.cam_rot), 48, 32, 1, RGBA8)
        assert np.array_equal(got, want), intensity
