"""Multi-GPU sharding of the render loop: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two ways to shard, both without touching the per-pixel arithmetic:

* frames   — frames of a camera path are independent units; rank r renders frames r, r+N, ...
             No data-path collective (bench.py default, weak scaling).
* row tiles — one frame split into interleaved tiles of `tile_rows` rows, tile t -> rank t % N
             (balances mirror/matte/miss regions); every rank renders into a COMPACT buffer that
             holds only its rows (rt_render_tiles_device) and ONE exchange step per frame gathers
             the shards to rank 0, which de-interleaves them into the reference's framebuffer
             layout data[(y*w+x)*3+c] (Source.cpp:550).  Used for the large-resolution configs.

Nothing here computes pixels.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def frames_for_rank(steps: int, world: int, rank: int, n_poses: int = 64) -> List[int]:
    """Pose index rendered by `rank` at each of its `steps` steps: frame (step*world + rank) % n_poses."""
    return [(s * world + rank) % n_poses for s in range(steps)]


def shard_tiles(height: int, tile_rows: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Global row ranges [r0, r1) of the tiles `rank` owns, in increasing y — the order in which
    rt_render_tiles_device lays them out in its compact buffer."""
    out = []
    t = rank
    while t * tile_rows < height:
        out.append((t * tile_rows, min((t + 1) * tile_rows, height)))
        t += world
    return out


def shard_rows(height: int, tile_rows: int, rank: int, world: int) -> int:
    return sum(r1 - r0 for r0, r1 in shard_tiles(height, tile_rows, rank, world))


def scatter_shard_into_frame(full, shard, height: int, width: int, tile_rows: int, rank: int, world: int,
                             channels: int = 3) -> None:
    """De-interleave one compact shard (rows*width*channels) into the full frame (height*width*channels).
    channels: elements per pixel — 3 for float32 RGB frames, 1 for OUT_RGBA8 frames held as one 32-bit
    element per pixel.  Works on torch tensors (CPU or GPU) and numpy arrays alike."""
    fv = full.reshape(height, width * channels)
    rows = shard_rows(height, tile_rows, rank, world)
    sv = shard.reshape(rows, width * channels)
    n_full = height // (tile_rows * world)          # tile groups in which every rank has a whole tile
    if n_full > 0:
        body = fv[: n_full * tile_rows * world].reshape(n_full, world, tile_rows, width * channels)
        body[:, rank] = sv[: n_full * tile_rows].reshape(n_full, tile_rows, width * channels)
    r = n_full * tile_rows
    for r0, r1 in shard_tiles(height, tile_rows, rank, world)[n_full:]:
        fv[r0:r1] = sv[r: r + (r1 - r0)]
        r += r1 - r0


def gather_frame(dist, shard, full, rank: int, world: int, height: int, width: int, tile_rows: int,
                 recv_bufs: Optional[list] = None, channels: int = 3):
    """The exchange step: every rank's compact shard goes to rank 0 (point-to-point sends — on the
    GPU box 7 concurrent xGMI transfers into GPU 0), which scatters them into `full`.  Returns
    `full` on rank 0, None elsewhere.  `recv_bufs` lets the caller reuse receive buffers."""
    if world == 1:
        scatter_shard_into_frame(full, shard, height, width, tile_rows, 0, 1, channels)
        return full
    if rank != 0:
        dist.send(shard, dst=0)
        return None
    bufs = recv_bufs if recv_bufs is not None else [None] * world
    reqs = []
    for k in range(1, world):
        if bufs[k] is None:
            bufs[k] = shard.new_empty(shard_rows(height, tile_rows, k, world) * width * channels)
        reqs.append(dist.irecv(bufs[k], src=k))
    scatter_shard_into_frame(full, shard, height, width, tile_rows, 0, world, channels)
    for q in reqs:
        q.wait()
    for k in range(1, world):
        scatter_shard_into_frame(full, bufs[k], height, width, tile_rows, k, world, channels)
    return full


def scatter_all_into_frame(full, recv, height: int, width: int, tile_rows: int, world: int, channels: int, pitch: int) -> None:
    """De-interleave the shards of ALL ranks at once: `recv` holds rank r's compact shard at elements
    [r*pitch, r*pitch + its size) (what a gather of equally padded shards leaves on rank 0).  The tile groups in
    which every rank has a whole tile move in ONE strided copy, the ragged tail in one more (plus one for a last,
    shorter tile) — two or three device operations per frame instead of one or two per rank."""
    e = tile_rows * width * channels                     # elements of a whole tile
    n_tiles = (height + tile_rows - 1) // tile_rows
    g_full = (height // tile_rows) // world              # groups of `world` whole tiles
    rv = recv.reshape(-1)[: world * pitch].reshape(world, pitch)
    fv = full.reshape(-1)
    if g_full > 0:
        fv[: g_full * world * e].reshape(g_full, world, e)[...] = rv[:, : g_full * e].reshape(world, g_full, e).swapaxes(0, 1)
    tail = n_tiles - g_full * world                      # tiles of the last, incomplete group: ranks 0 .. tail-1, local tile g_full
    if tail > 0:
        last_rows = height - (n_tiles - 1) * tile_rows   # rows of the very last tile (== tile_rows unless the height is ragged)
        whole = tail if last_rows == tile_rows else tail - 1
        base = g_full * world * e
        if whole > 0:
            fv[base: base + whole * e].reshape(whole, e)[...] = rv[:whole, g_full * e: (g_full + 1) * e]
        if whole < tail:
            n = last_rows * width * channels
            fv[base + whole * e: base + whole * e + n] = rv[whole, g_full * e: g_full * e + n]


def scatter_all_native(full, recv, height: int, width: int, tile_rows: int, world: int, channels: int, pitch: int, device: int = 0) -> None:
    """:func:`scatter_all_into_frame` by the library (rt_scatter_shards: one kernel on the current stream for GPU tensors;
    rt_scatter_shards_host: one memcpy per tile for host tensors and numpy arrays) — no tensor operations in the interpreter.
    `channels` and `pitch` count elements: 3 float32 per pixel, or 1 32-bit element per OUT_RGBA8 pixel."""
    from . import api
    mode = api.STRICT | (api.OUT_RGBA8 if channels == 1 else 0)
    if hasattr(full, "data_ptr"):
        esize, fptr, rptr, on_gpu = full.element_size(), full.data_ptr(), recv.data_ptr(), bool(full.is_cuda)
    else:
        esize, fptr, rptr, on_gpu = full.itemsize, full.ctypes.data, recv.ctypes.data, False
    if esize != 4:
        raise ValueError("frames are held as 32-bit elements (float32 RGB or one packed RGBA8 pixel)")
    if on_gpu:
        import torch
        api.scatter_shards(full.device.index if full.device.index is not None else device, rptr, pitch * 4, world, width, height,
                           tile_rows, mode, fptr, torch.cuda.current_stream(full.device).cuda_stream)
    else:
        api.scatter_shards_host(rptr, pitch * 4, world, width, height, tile_rows, mode, fptr)


class TileExchange:
    """Pipelined form of :func:`gather_frame` for a sequence of frames: `depth` shard buffers per
    rank, so that frame f+1 is rendered while frame f is still on its way to rank 0.

    Per frame, every rank calls::

        buf = ex.shard(f)          # compact buffer to render this rank's tiles of frame f into
        ... rt_render_tiles_device(..., buf) ...
        done = ex.submit(f)        # start the exchange of frame f

    `submit` on rank 0 returns the list of (frame index, full frame) that became complete with this
    call (frames complete in order, one call late at most `depth - 1` frames); other ranks get [].
    `drain()` completes what is still in flight.  `depth == 1` is the serial gather.

    The exchange of a frame is ONE non-blocking `gather` to rank 0 (RCCL runs it as grouped sends and
    receives: the 7 transfers into GPU 0 use their xGMI links concurrently): every rank's shard buffer is padded
    to the size of the largest shard so that the pieces are equal, and rank 0 de-interleaves all of them with
    :func:`scatter_all_native` — per frame one collective call and ONE kernel of the library on rank 0 (rt_scatter_shards),
    whatever the number of ranks; :func:`scatter_all_into_frame` is the same thing in tensor operations, kept as the tests' check.
    """

    def __init__(self, dist, rank: int, world: int, height: int, width: int, tile_rows: int, alloc, depth: int = 2,
                 channels: int = 3, collective: bool = True):
        """`alloc(n)` returns a tensor of n elements on the device the exchange runs on: float32 with
        channels = 3 (RGB frames), a 32-bit integer type with channels = 1 (OUT_RGBA8 frames: a third of the bytes).
        collective=False exchanges by point-to-point sends and receives instead of the gather (also the fallback
        when the backend has no gather)."""
        self.dist, self.rank, self.world = dist, rank, world
        self.h, self.w, self.tile_rows, self.depth, self.ch = height, width, tile_rows, max(1, depth), channels
        self.n = shard_rows(height, tile_rows, rank, world) * width * channels
        self.pitch = max(shard_rows(height, tile_rows, k, world) for k in range(world)) * width * channels
        self._shards = [alloc(self.pitch) for _ in range(self.depth)]
        self._work = [None] * self.depth                     # the gather that reads shard buffer k
        self._use_gather = bool(collective)
        if rank == 0:
            self._full = [alloc(height * width * channels) for _ in range(self.depth)]
            self._recv = [alloc(world * self.pitch) for _ in range(self.depth)] if world > 1 else None
            self._pending = []                               # [(frame, slot)] in frame order

    def shard(self, f: int):
        """This rank's compact buffer for frame f (its first shard_rows * width * channels elements are the shard)."""
        k = f % self.depth
        if self._work[k] is not None:                        # the buffer is free once the gather that reads it is done
            self._work[k].wait()
            self._work[k] = None
        return self._shards[k][: self.n]                     # (rank 0: slot k's frame was completed by an earlier submit)

    def submit(self, f: int):
        k = f % self.depth
        if self.world == 1:
            scatter_all_native(self._full[k], self._shards[k], self.h, self.w, self.tile_rows, 1, self.ch, self.pitch)
            return [(f, self._full[k])]
        pieces = None
        if self.rank == 0:
            pieces = [self._recv[k][r * self.pitch: (r + 1) * self.pitch] for r in range(self.world)]
        if self._use_gather:
            try:
                self._work[k] = self.dist.gather(self._shards[k], pieces, dst=0, async_op=True)
            except (RuntimeError, NotImplementedError):      # a backend without gather: every rank lands here alike
                self._use_gather = False
        if not self._use_gather:
            self._work[k] = self._point_to_point(k, pieces)
        if self.rank != 0:
            return []
        self._pending.append((f, k))
        return self._complete(keep=self.depth - 1)

    def _point_to_point(self, k: int, pieces):
        """The same exchange as non-blocking sends to rank 0 and one receive per peer there."""
        if self.rank != 0:
            return self.dist.isend(self._shards[k], dst=0)
        reqs = [self.dist.irecv(pieces[src], src=src) for src in range(1, self.world)]
        pieces[0][...] = self._shards[k]

        class _All:
            def wait(_self):
                for q in reqs:
                    q.wait()
        return _All()

    def _complete(self, keep: int):
        done = []
        while self.rank == 0 and len(self._pending) > keep:
            f, k = self._pending.pop(0)
            if self._work[k] is not None:
                self._work[k].wait()
                self._work[k] = None
            scatter_all_native(self._full[k], self._recv[k], self.h, self.w, self.tile_rows, self.world, self.ch, self.pitch)
            done.append((f, self._full[k]))
        return done

    def drain(self):
        done = self._complete(keep=0) if self.rank == 0 and self.world > 1 else []
        for k in range(self.depth):
            if self._work[k] is not None:
                self._work[k].wait()
                self._work[k] = None
        return done
