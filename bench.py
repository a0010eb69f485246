#!/usr/bin/env python3
"""bench.py — the headline measurement: Mrays/s (primary + bounce + shadow) and ms/frame of the
per-pixel render loop at 800x800 on 1..8 MI355X (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one frame: one pass of the render loop (rt_render_device, include/rt355.h) over the
BASELINE config-2 workload — the README-style room (26 triangles, two mirror walls, blue occluder,
one light), 800x800, reference depth 1 — into a framebuffer that stays resident in HBM.  The scene
is already committed (resident) when the timed region starts.  Successive frames walk the
deterministic 64-pose camera path of config 5 (scenes.camera_sweep); with N ranks the frames are
dealt round-robin to the ranks (frames are independent units: no data-path collective, weak
scaling — every rank renders K frames).  `value` = rays of all frames of all ranks / wall time of
the slowest rank.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      FP32 VALU roofline of the render kernel: algorithmic FLOPs (SURVEY §8d pricing,
                counted by rt_render_stats on the same frames) / mean kernel duration from HIP
                events on the launch stream / 157.3 TFLOP/s
  cpu_baseline  the reference's own getPixelData loop (oracle/_ref/ref_render, built from
                /root/reference in the build container) or, when that binary is absent, the C
                restatement in oracle/ — one host thread, same scene and size, a bounded number
                of frames.
Other workloads (BASELINE configs 3-5) are selected with --workload; they are not what the driver
runs.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="room800",
                    choices=["room800", "stock800", "mirrors4k", "spheres800", "spheres4k", "sweep8k", "sweep8k_spheres"])
    ap.add_argument("--mode", default="strict", choices=["strict", "fast"])
    ap.add_argument("--tiles-per-wave", type=int, default=0)
    ap.add_argument("--frames-per-launch", type=int, default=32,
                    help="frames of the camera path rendered by one launch (rt_render_frames_device); 1 = one launch per frame")
    ap.add_argument("--size", type=int, nargs=2, default=None, help="override the workload's frame size (experiments)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo lets several ranks share one GPU to rehearse the N>1 path")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_workload(name):
    from raytracing_amd import scenes
    if name == "room800":
        return scenes.room(800, 800), "brute", "README room scene (26 triangles, 2 mirror walls, occluder, 1 light) 800x800 depth 1"
    if name == "stock800":
        return scenes.stock(800, 800), "brute", "unmodified v1 scene (24 triangles, 1 light) 800x800 depth 1"
    if name == "mirrors4k":
        return scenes.mirrors(3840, 2160, 16), "brute", "facing mirrors 3840x2160 depth 16"
    if name == "spheres800":
        return scenes.spheres(20, 4, 800, 800), "lbvh", "20 icospheres x 5120 + room = 102412 triangles, 800x800, LBVH"
    if name == "spheres4k":
        return scenes.spheres(20, 4, 3840, 2160), "lbvh", "102412 triangles, 3840x2160, LBVH"
    if name == "sweep8k":
        return scenes.room(7680, 4320), "brute", "README room scene 7680x4320, 64-pose sweep, interleaved row tiles + gather"
    if name == "sweep8k_spheres":
        return scenes.spheres(20, 4, 7680, 4320), "lbvh", "102412 triangles 7680x4320, 64-pose sweep, LBVH, interleaved row tiles + gather"
    raise ValueError(name)


def cpu_baseline(desc, seconds):
    """One host thread on the same scene/size.  Prefers the reference's own code."""
    from oracle import pyoracle
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_render" if desc.depth == 1 else "ref_render_depth")
    pyoracle.build()
    # one counted frame: rays per frame for this pose (identical to the GPU's counters) and a
    # first per-frame time to size the sample
    t0 = time.perf_counter()
    _, ctr = pyoracle.render_desc(desc, counters=True)
    one = time.perf_counter() - t0
    rays = ctr.rays
    frames = max(1, min(64, int(seconds / max(one, 1e-3))))
    if os.access(ref, os.X_OK):
        with tempfile.TemporaryDirectory() as tmp:
            sp = os.path.join(tmp, "scene.txt")
            with open(sp, "w") as f:
                f.write(desc.to_text())
            t0 = time.perf_counter()
            res = subprocess.run([ref, "render", sp, os.path.join(tmp, "o"), str(frames)],
                                 capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if res.returncode == 0:
                info = json.loads(res.stdout.strip().splitlines()[-1])
                ms = float(info["ms_per_frame"])           # min over `frames` frames, as Source.cpp:1300 measures
                return {"value": rays / ms / 1e3, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                        "ms_per_frame": ms, "rays_per_frame": rays,
                        "sample": "%d frames of the same %dx%d scene by the reference's getPixelData loop "
                                  "(oracle/_ref, g++ -O2 -ffp-contract=off), best frame, %.1f s of CPU" % (frames, desc.width, desc.height, wall)}
    best = one
    t_all = time.perf_counter()
    for _ in range(frames):
        t0 = time.perf_counter()
        pyoracle.render_desc(desc)
        best = min(best, time.perf_counter() - t0)
    wall = time.perf_counter() - t_all
    return {"value": rays / best / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "ms_per_frame": best * 1e3,
            "rays_per_frame": rays,
            "sample": "%d frames of the same %dx%d scene by oracle/rt_oracle.c (gcc -O2 -ffp-contract=off), "
                      "best frame, %.1f s of CPU" % (frames, desc.width, desc.height, wall)}


def cpu_threaded(desc, threads):
    from oracle import pyoracle
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        pyoracle.render_desc(desc, threads=threads)
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    from raytracing_amd import api, scenes, sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render loop has no CPU path")
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist = None
    on_gpu = args.backend == "nccl"          # where the tensors handed to torch.distributed live
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")

    desc, accel_name, label = make_workload(args.workload)
    if args.size:
        desc = desc.with_size(args.size[0], args.size[1])
        label += " [size overridden to %dx%d]" % tuple(args.size)
    accel = api.ACCEL_LBVH if accel_name == "lbvh" else api.ACCEL_BRUTE
    mode = api.FAST if args.mode == "fast" else api.STRICT
    if args.tiles_per_wave:
        api.lib().rt_set_tiles_per_wave(args.tiles_per_wave)

    scene = api.Scene.from_desc(desc)
    scene.commit(dev, accel)
    W, H, D = desc.width, desc.height, desc.depth
    poses = scenes.camera_sweep(64, desc.cam_pos, desc.cam_rot)
    trans = [api.camera_trans(r) for _, r in poses]

    # work per pose (rays and algorithmic FLOPs), counted by the instrumented kernel outside the
    # timed region; identical to the oracle's counters (tests/test_gpu_parity.py)
    tile_mode = args.workload.startswith("sweep8k") and world > 1
    if tile_mode:
        my_frames = [s % 64 for s in range(args.steps)]         # every rank works on the same frame
    else:
        my_frames = sharding.frames_for_rank(args.steps, world, rank)
    need = sorted(set(my_frames))
    rays_of, flops_of = {}, {}
    for f in need:
        _, st = scene.render_stats(poses[f][0], trans[f], W, H, D, want_frame=False)
        rays_of[f], flops_of[f] = st.rays, st.flops

    if tile_mode:
        tile_rows = 16
        rows = api.shard_rows(H, tile_rows, rank, world)
        fb = torch.empty(rows * W * 3, dtype=torch.float32, device="cuda")
        # two shard buffers per rank: frame f+1 renders while frame f is on its way to rank 0
        exchange = sharding.TileExchange(dist, rank, world, H, W, tile_rows,
                                         lambda n: torch.empty(n, dtype=torch.float32, device="cuda" if on_gpu else "cpu"), depth=2)
    else:
        fpl = max(1, min(32, args.frames_per_launch))
        fb = torch.empty(fpl * H * W * 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    # pre-marshalled arguments: the timed loop is then one ctypes call per frame
    import ctypes as C
    L = api.lib()
    fptr = C.POINTER(C.c_float)
    cam_c = [np.ascontiguousarray(p, dtype=np.float32) for p, _ in poses]
    cam_p = [a.ctypes.data_as(fptr) for a in cam_c]
    trans_p = [a.ctypes.data_as(fptr) for a in trans]
    fb_ptr, stream_p, handle = C.c_void_p(fb.data_ptr()), C.c_void_p(stream), scene._h

    batch_cache = {}

    def launch_frames(frames):
        """One call = one launch for up to 32 frames of the path (frames stay resident in `fb`).
        The marshalled pose arrays of a batch are built once (warm-up) and reused."""
        key = tuple(frames)
        if key not in batch_cache:
            cams = np.ascontiguousarray(np.stack([cam_c[f] for f in frames]), dtype=np.float32)
            trs = np.ascontiguousarray(np.stack([trans[f] for f in frames]), dtype=np.float32)
            batch_cache[key] = (cams, trs, cams.ctypes.data_as(fptr), trs.ctypes.data_as(fptr), C.c_size_t(H * W * 3))
        _, _, cp, tp, stride = batch_cache[key]
        rc = L.rt_render_frames_device(handle, dev, len(frames), cp, tp, W, H, D, mode, fb_ptr, stride, stream_p)
        if rc != 0:
            raise api.RtError(rc, L.rt_last_error().decode())

    def run_steps(frames):
        """`len(frames)` steps (one step = one frame)."""
        if tile_mode:
            # row tiles: render this rank's interleaved tiles, then the one exchange step (RCCL), pipelined
            for i, f in enumerate(frames):
                buf = exchange.shard(i)
                if on_gpu:
                    scene.render_tiles_device(dev, poses[f][0], trans[f], W, H, D, mode, tile_rows, rank, world, buf.data_ptr(), stream)
                else:   # gloo rehearsal: the shard is staged through host memory; RCCL sends it GPU to GPU
                    scene.render_tiles_device(dev, poses[f][0], trans[f], W, H, D, mode, tile_rows, rank, world, fb.data_ptr(), stream)
                    buf.copy_(fb)
                exchange.submit(i)
            exchange.drain()
        else:
            for i in range(0, len(frames), fpl):
                launch_frames(frames[i:i + fpl])

    run_steps([my_frames[s % len(my_frames)] for s in range(args.warmup)])
    if not tile_mode:                       # marshal every batch of the timed loop ahead of it
        for i in range(0, len(my_frames), fpl):
            key = tuple(my_frames[i:i + fpl])
            if key not in batch_cache:
                launch_frames(list(key))
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(my_frames)
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n_launches = args.steps if tile_mode else (args.steps + fpl - 1) // fpl
    kernel_ms = ev0.elapsed_time(ev1) / n_launches          # HIP events on the launch stream: mean launch duration

    my_rays = float(sum(rays_of[f] for f in my_frames))
    my_flops = float(sum(flops_of[f] for f in my_frames))
    if dist:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        if not tile_mode:
            r = torch.tensor([my_rays, my_flops], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
            my_rays, my_flops = float(r[0].item()), float(r[1].item())
    total_rays, total_flops = my_rays, my_flops

    if rank == 0:
        frames_total = args.steps if tile_mode else args.steps * world
        flops_per_launch = sum(flops_of[f] for f in my_frames) / n_launches
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12 if not tile_mode else None
        out = {
            "metric": "Mrays/s (primary+bounce+shadow) at 800x800" if args.workload.endswith("800") else "Mrays/s (primary+bounce+shadow)",
            "value": total_rays / wall / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "ms_per_frame": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if tile_mode else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label, "accel": accel_name, "traversal": {0: None, 1: "per-lane", 2: "packet"}[scene.traversal(dev, trans[0], W, H)], "mode": args.mode, "width": W, "height": H, "max_bounces": D,
                       "triangles": scene.n_triangles, "lights": scene.n_lights,
                       "frames_per_launch": (fpl if not tile_mode else 1),
                       "frames": ("64-pose camera sweep, frame (step*N + rank) mod 64, %d frames per launch" % fpl) if not tile_mode else "64-pose sweep, interleaved 16-row tiles, gather to rank 0",
                       "rays_per_frame_mean": total_rays / frames_total, "parallelism": "frames x%d" % world if not tile_mode else "row tiles x%d" % world},
        }
        if achieved is not None:
            out["roofline"] = {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": read_traffic(args.workload),
                               "kernel_ms": kernel_ms,
                               "flops_per_launch": flops_per_launch, "frames_per_launch": fpl, "launches": n_launches,
                               "note": "FP32 VALU roofline (SURVEY 8d): nothing here is HBM- or MFMA-shaped; "
                                       "algorithmic FLOPs = 14/plane + 6/point + 17/edge term + 22/box + 32|29|30 per ray"}
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(desc, args.cpu_seconds)
            out["cpu_baseline"] = cb
            threads = min(16, os.cpu_count() or 1)
            mt = cpu_threaded(desc, threads)
            out["cpu_threaded"] = {"value": cb["rays_per_frame"] / mt / 1e6, "unit": "Mrays/s",
                                   "cores": threads, "kind": "port", "ms_per_frame": mt * 1e3,
                                   "sample": "row-interleaved pthreads over the same frame (stands in for the reference's disabled std::execution::par loop)"}
            out["gpu_over_cpu_1thread"] = out["value"] / cb["value"]
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()


def read_traffic(workload):
    """HBM bytes per launch from the PMC passes kept under profiles/ (collected by
    tools/profile.sh in separate rocprofv3 --pmc runs); None when no such pass exists."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
