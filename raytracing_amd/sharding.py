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


def scatter_shard_into_frame(full, shard, height: int, width: int, tile_rows: int, rank: int, world: int) -> None:
    """De-interleave one compact shard (rows*width*3) into the full frame (height*width*3).
    Works on torch tensors (CPU or GPU) and numpy arrays alike."""
    fv = full.reshape(height, width * 3)
    rows = shard_rows(height, tile_rows, rank, world)
    sv = shard.reshape(rows, width * 3)
    n_full = height // (tile_rows * world)          # tile groups in which every rank has a whole tile
    if n_full > 0:
        body = fv[: n_full * tile_rows * world].reshape(n_full, world, tile_rows, width * 3)
        body[:, rank] = sv[: n_full * tile_rows].reshape(n_full, tile_rows, width * 3)
    r = n_full * tile_rows
    for r0, r1 in shard_tiles(height, tile_rows, rank, world)[n_full:]:
        fv[r0:r1] = sv[r: r + (r1 - r0)]
        r += r1 - r0


def gather_frame(dist, shard, full, rank: int, world: int, height: int, width: int, tile_rows: int,
                 recv_bufs: Optional[list] = None):
    """The exchange step: every rank's compact shard goes to rank 0 (point-to-point sends — on the
    GPU box 7 concurrent xGMI transfers into GPU 0), which scatters them into `full`.  Returns
    `full` on rank 0, None elsewhere.  `recv_bufs` lets the caller reuse receive buffers."""
    if world == 1:
        scatter_shard_into_frame(full, shard, height, width, tile_rows, 0, 1)
        return full
    if rank != 0:
        dist.send(shard, dst=0)
        return None
    bufs = recv_bufs if recv_bufs is not None else [None] * world
    reqs = []
    for k in range(1, world):
        if bufs[k] is None:
            bufs[k] = shard.new_empty(shard_rows(height, tile_rows, k, world) * width * 3)
        reqs.append(dist.irecv(bufs[k], src=k))
    scatter_shard_into_frame(full, shard, height, width, tile_rows, 0, world)
    for q in reqs:
        q.wait()
    for k in range(1, world):
        scatter_shard_into_frame(full, bufs[k], height, width, tile_rows, k, world)
    return full
