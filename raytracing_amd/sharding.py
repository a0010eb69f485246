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
    """

    def __init__(self, dist, rank: int, world: int, height: int, width: int, tile_rows: int, alloc, depth: int = 2,
                 channels: int = 3):
        """`alloc(n)` returns a tensor of n elements on the device the exchange runs on: float32 with
        channels = 3 (RGB frames), a 32-bit integer type with channels = 1 (OUT_RGBA8 frames: a third of the bytes)."""
        self.dist, self.rank, self.world = dist, rank, world
        self.h, self.w, self.tile_rows, self.depth, self.ch = height, width, tile_rows, max(1, depth), channels
        n = shard_rows(height, tile_rows, rank, world) * width * channels
        self._shards = [alloc(n) for _ in range(self.depth)]
        self._sends = [None] * self.depth                    # rank != 0: the send using shard buffer k
        if rank == 0:
            self._full = [alloc(height * width * channels) for _ in range(self.depth)]
            self._recv = [[alloc(shard_rows(height, tile_rows, k, world) * width * channels) if k else None
                           for k in range(world)] for _ in range(self.depth)]
            self._pending = []                               # [(frame, slot, [requests])] in frame order

    def shard(self, f: int):
        k = f % self.depth
        if self._sends[k] is not None:                       # the buffer is free once its last send is done
            self._sends[k].wait()
            self._sends[k] = None
        return self._shards[k]                              # (rank 0: slot k's frame was completed by an earlier submit)

    def submit(self, f: int):
        k = f % self.depth
        if self.world == 1:
            scatter_shard_into_frame(self._full[k], self._shards[k], self.h, self.w, self.tile_rows, 0, 1, self.ch)
            return [(f, self._full[k])]
        if self.rank != 0:
            self._sends[k] = self.dist.isend(self._shards[k], dst=0)
            return []
        reqs = [self.dist.irecv(self._recv[k][src], src=src) for src in range(1, self.world)]
        scatter_shard_into_frame(self._full[k], self._shards[k], self.h, self.w, self.tile_rows, 0, self.world, self.ch)
        self._pending.append((f, k, reqs))
        return self._complete(keep=self.depth - 1)

    def _complete(self, keep: int):
        done = []
        while self.rank == 0 and len(self._pending) > keep:
            f, k, reqs = self._pending.pop(0)
            for q in reqs:
                q.wait()
            for src in range(1, self.world):
                scatter_shard_into_frame(self._full[k], self._recv[k][src], self.h, self.w, self.tile_rows, src, self.world, self.ch)
            done.append((f, self._full[k]))
        return done

    def drain(self):
        for k in range(self.depth):
            if self._sends[k] is not None:
                self._sends[k].wait()
                self._sends[k] = None
        return self._complete(keep=0) if self.rank == 0 else []
