#!/usr/bin/env python3
"""bench.py — the headline measurement: Mrays/s (primary + bounce + shadow) and ms/frame of the
per-pixel render loop at 800x800 on 1..8 MI355X (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One GPU (the default): the workload is BASELINE config 2 — the README-style room (26 triangles, two
mirror walls, blue occluder, one light), 800x800, reference depth 1.  A *step* is ONE LAUNCH of the
render loop over a batch of consecutive frames (64 at 800x800: the whole path) of the deterministic camera path of config 5
(rt_render_frames_device, include/rt355.h; the poses come from the reference's own controller,
api.camera_sweep -> rt_camera_path), into framebuffers that stay resident in HBM; the scene is
committed before the timed region starts.  K steps = 64*K frames at 800x800.  `value` = rays of all frames /
wall time.  The line also carries, measured outside the timed region of the headline:
  configs       the other BASELINE configs on this GPU (stock 800x800, facing mirrors 3840x2160 depth
                16, 102 412 triangles through the LBVH at 800x800 and 3840x2160 with the build time,
                the room at 7680x4320), each with ms/frame, Mrays/s, kernel ms and roofline fraction
  hand_off      one frame per launch and the end-to-end cost of the viewer hand-off (Source.cpp:1270):
                rt_render into pageable memory, and rt_render_async into pinned double buffers
  roofline      FP32 VALU roofline of the render kernel: algorithmic FLOPs (SURVEY §8d pricing,
                counted by rt_render_stats in the same traversal on the same poses) / mean kernel
                duration from HIP events on the launch stream / 157.3 TFLOP/s
  cpu_baseline  the reference's own getPixelData loop (oracle/_ref/ref_render, built from
                /root/reference in the build container) or, when that binary is absent, the C
                restatement in oracle/ — one host thread, same scene and size, a bounded sample.

Several GPUs (--gpus N > 1, one rank per GPU): the default workload is BASELINE config 5, the one
north_star partitions — 7680x4320 frames of the same camera path, each split into interleaved
16-row tiles (tile t -> rank t mod N, rt_render_tiles_device), one exchange step per frame that
gathers the shards on rank 0 over RCCL (sharding.TileExchange, pipelined over two shard buffers).  A
step is one frame; total work is fixed ("scaling": "strong").  `--workload room800 --gpus N` keeps the
frame-parallel mode instead (rank r renders batches r, r+N, ...; no collective; weak scaling).
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
from raytracing_amd.srchash import source_hash  # noqa: E402
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
N_POSES = 64

# name -> (scene builder, acceleration, label, frames per launch)
WORKLOADS = {
    "room800": (lambda s: s.room(800, 800), "brute",
                "README room scene (26 triangles, 2 mirror walls, occluder, 1 light) 800x800 depth 1", 64),
    "stock800": (lambda s: s.stock(800, 800), "brute", "unmodified v1 scene (24 triangles, 1 light) 800x800 depth 1", 64),
    "mirrors4k": (lambda s: s.mirrors(3840, 2160, 16), "brute", "facing mirrors 3840x2160 depth 16", 64),
    "spheres800": (lambda s: s.spheres(20, 4, 800, 800), "lbvh", "20 icospheres x 5120 + room = 102412 triangles, 800x800, LBVH", 64),
    "spheres4k": (lambda s: s.spheres(20, 4, 3840, 2160), "lbvh", "102412 triangles, 3840x2160, LBVH", 64),
    "sweep8k": (lambda s: s.room(7680, 4320), "brute", "README room scene 7680x4320, 64-pose camera path", 32),
    "sweep8k_spheres": (lambda s: s.spheres(20, 4, 7680, 4320), "lbvh", "102412 triangles 7680x4320, 64-pose camera path, LBVH", 32),
}

SPHERE_SIZES = {"spheres800": (800, 800), "spheres4k": (3840, 2160), "sweep8k_spheres": (7680, 4320)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: room800 on one GPU, sweep8k (row tiles + gather) on several")
    ap.add_argument("--tiles-per-wave", type=int, default=0)
    ap.add_argument("--frames-per-launch", type=int, default=0, help="frames per launch (= per step); 0: the workload's own (64 at 800x800 and 4K: 6.4 GB of frames at 4K; 32 at 8K: 12.7 GB)")
    ap.add_argument("--size", type=int, nargs=2, default=None, help="override the workload's frame size (experiments)")
    ap.add_argument("--traversal", default="auto", choices=["auto", "per-lane", "packet"])
    ap.add_argument("--kernel", default="auto", choices=["auto", "state-machine", "lockstep"], help="form of the brute-force render kernel (rt_scene_set_kernel)")
    ap.add_argument("--build", default="ploc", choices=["ploc", "fast", "sah"],
                    help="how LBVH commits build their tree (rt_scene_set_build): ploc = Morton sort + locally-ordered clustering on the GPU (default), "
                         "fast = Morton sort + Karras on the GPU, sah = binned SAH on host threads")
    ap.add_argument("--exchange", default="rgba8", choices=["rgba8", "f32"],
                    help="pixel format of the frames the row-tile mode renders and gathers (RT_OUT_RGBA8 or binary32 RGB)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo lets several ranks share one GPU to rehearse the N>1 path")
    ap.add_argument("--settle-ms", type=float, default=40.0, help="untimed launches for this long after the W warm-up steps, so that the timed steps run at settled clocks (0: none)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the other configs and the hand-off numbers")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


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


def gpu_same_pose(api, torch, desc, dev, accel, frames=32, launches=4):
    """The GPU on exactly the frame the CPU legs render — desc's own pose, `frames` copies of it per launch (the camera path of
    the headline number visits other poses with other ray counts) -> ms per frame."""
    cams = np.ascontiguousarray(np.tile(np.asarray(desc.cam_pos, dtype=np.float32), (frames, 1)))
    trs = np.ascontiguousarray(np.tile(np.asarray(api.camera_trans(desc.cam_rot), dtype=np.float32).reshape(1, 16), (frames, 1)))
    n = desc.width * desc.height * 3
    buf = torch.empty(frames * n, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with api.Scene.from_desc(desc) as sc:
        sc.commit(dev, accel)
        for _ in range(2):
            sc.render_frames_device(dev, cams, trs, desc.width, desc.height, desc.depth, api.STRICT, buf.data_ptr(), n, st)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(launches):
            sc.render_frames_device(dev, cams, trs, desc.width, desc.height, desc.depth, api.STRICT, buf.data_ptr(), n, st)
        ev1.record()
        torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / (launches * frames)


def cpu_gpu_row(api, torch, desc, dev, accels, seconds, what):
    """One BASELINE.md section-3 row: the reference's loop on one host thread and the GPU on the same frame (same pose, same size)."""
    cb = cpu_baseline(desc, seconds)
    row = {"what": what, "width": desc.width, "height": desc.height, "triangles": desc.n_triangles,
           "cpu_baseline": cb}
    for label, accel in accels:
        ms = gpu_same_pose(api, torch, desc, dev, accel)
        row["gpu_%s_ms_per_frame" % label] = ms
        row["gpu_%s_Mrays_per_s" % label] = cb["rays_per_frame"] / ms / 1e3
        row["gpu_%s_over_cpu_1thread" % label] = cb["rays_per_frame"] / ms / 1e3 / cb["value"]
    return row


def cpu_threaded(desc, threads):
    from oracle import pyoracle
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        pyoracle.render_desc(desc, threads=threads)
        best = min(best, time.perf_counter() - t0)
    return best


class Workload:
    """One committed scene + its camera path + pre-marshalled launch arguments."""

    _descs = {}     # scene descriptions by workload (102 412 triangles take seconds to list in Python; several configs share them)

    def __init__(self, api, torch, name, dev, args, frames_per_launch=0, build=None):
        from raytracing_amd import scenes
        import ctypes as C
        make, accel_name, label, fpl = WORKLOADS[name]
        self.api, self.torch, self.C, self.name, self.dev = api, torch, C, name, dev
        if name not in Workload._descs:
            if name in SPHERE_SIZES:       # one list of 102 412 triangles for all its frame sizes
                if "spheres" not in Workload._descs:
                    Workload._descs["spheres"] = make(scenes)
                Workload._descs[name] = Workload._descs["spheres"].with_size(*SPHERE_SIZES[name])
            else:
                Workload._descs[name] = make(scenes)
        desc = Workload._descs[name]
        if args.size:
            desc = desc.with_size(args.size[0], args.size[1])
            label += " [size overridden to %dx%d]" % tuple(args.size)
        self.desc, self.accel_name, self.label = desc, accel_name, label
        self.fpl = max(1, min(64, frames_per_launch or args.frames_per_launch or fpl))
        self.W, self.H, self.D = desc.width, desc.height, desc.depth
        self.scene = api.Scene.from_desc(desc)
        self.scene.set_traversal({"auto": api.TRAVERSAL_AUTO, "per-lane": api.TRAVERSAL_PER_LANE, "packet": api.TRAVERSAL_PACKET}[args.traversal])
        self.scene.set_kernel({"auto": api.KERNEL_AUTO, "state-machine": api.KERNEL_STATE_MACHINE, "lockstep": api.KERNEL_LOCKSTEP}[args.kernel])
        kinds = {"fast": api.BUILD_FAST, "sah": api.BUILD_SAH, "ploc": api.BUILD_PLOC}
        self.scene.set_build(kinds[build or args.build])
        self.build = None
        if args.tiles_per_wave:
            self.scene.set_tiles_per_wave(args.tiles_per_wave)
        self.scene.commit(dev, api.ACCEL_LBVH if accel_name == "lbvh" else api.ACCEL_BRUTE)
        if accel_name == "lbvh":
            # what REALLY built the tree (a clustered build that cannot finish falls back to Karras'), and the build time of a
            # second commit: the first one in a process also loads the build kernels' code objects
            self.build_ms_first = self.scene.build_ms(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.scene.commit(dev, api.ACCEL_LBVH)
            torch.cuda.synchronize()
            self.commit_wall_ms = (time.perf_counter() - t0) * 1e3     # the whole rt_scene_commit: upload, tree, records, shadow grids
            self.build_ms = self.scene.build_ms(dev)
            try:
                g = [self.scene.shadow_grid(dev, k) for k in range(min(4, self.scene.n_lights))]
            except api.RtError:
                g = []
            self.grid_ms = sum(x["build_ms"] for x in g[:1]) if g else 0.0     # (build_ms is the time of all the commit's grids together)
            self.grids = [{k: x[k] for k in ("cells_per_edge", "entries", "always")} for x in g]
            self.build = {v: k for k, v in kinds.items()}[self.scene.build_kind(dev)]
        self.cams, self.trans = api.camera_sweep(N_POSES, desc.cam_pos, desc.cam_rot)     # (64,3), (64,16): rt_camera_path
        self.rays_of, self.flops_of = {}, {}
        self.fb = None
        self.stream = torch.cuda.current_stream().cuda_stream
        self._batches = {}

    def count(self, frames):
        """Rays and algorithmic FLOPs per pose, counted by the instrumented kernel in the traversal the
        frame really gets (outside every timed region; brute force: identical to the oracle's counters)."""
        for f in sorted(set(frames)):
            if f not in self.rays_of:
                _, st = self.scene.render_stats(self.cams[f], self.trans[f], self.W, self.H, self.D, want_frame=False)
                self.rays_of[f], self.flops_of[f] = st.rays, st.flops

    def launch(self, frames):
        """One launch for up to 32 frames of the path; the frames stay resident in HBM."""
        api, C = self.api, self.C
        if self.fb is None:
            self.fb = self.torch.empty(self.fpl * self.H * self.W * 3, dtype=self.torch.float32, device="cuda")
        _, _, cp, tp = self.prepare(frames)
        rc = api.lib().rt_render_frames_device(self.scene._h, self.dev, len(frames), cp, tp, self.W, self.H, self.D, api.STRICT,
                                               C.c_void_p(self.fb.data_ptr()), C.c_size_t(self.H * self.W * 3), C.c_void_p(self.stream))
        if rc != 0:
            raise api.RtError(rc, api.lib().rt_last_error().decode())

    def prepare(self, frames):
        """Marshalled pose arrays of one batch (built once, ahead of the timed loop)."""
        key = tuple(frames)
        if key not in self._batches:
            fptr = self.C.POINTER(self.C.c_float)
            cams = np.ascontiguousarray(self.cams[list(frames)], dtype=np.float32)
            trs = np.ascontiguousarray(self.trans[list(frames)], dtype=np.float32)
            self._batches[key] = (cams, trs, cams.ctypes.data_as(fptr), trs.ctypes.data_as(fptr))
        return self._batches[key]

    def batch(self, step, world=1, rank=0):
        """Poses of step `step` of rank `rank`: consecutive frames of the 64-pose path."""
        first = ((step * world + rank) * self.fpl) % N_POSES
        return [(first + k) % N_POSES for k in range(self.fpl)]

    def traversal(self):
        return {0: None, 1: "per-lane", 2: "packet"}[self.scene.traversal(self.dev, self.trans[0], self.W, self.H)]

    def timed_launches(self, batches):
        """-> (wall seconds, mean kernel ms per launch from HIP events on the launch stream)."""
        torch = self.torch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for b in batches:
            self.launch(b)
        ev1.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / max(1, len(batches))

    def close(self):
        self.fb = None
        self.scene.close()


_DEVICE = {}


def device_info(torch, dev):
    """What the box says it is (record CU count and clock next to the peak used): the FP32 VALU peak recomputed
    from them, CUs x 256 FLOP/clk x the reported engine clock, is informative; `peak` stays the guide's 157.3."""
    if not _DEVICE:
        try:
            p = torch.cuda.get_device_properties(dev)
            import ctypes
            from raytracing_amd import api as _api
            cus, clk = ctypes.c_int(0), ctypes.c_int(0)
            khz = clk.value if _api.lib().rt_device_props(int(dev), ctypes.byref(cus), ctypes.byref(clk)) == 0 else 0
            _DEVICE.update({"name": p.name, "gcn_arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
                            "clock_mhz": khz / 1e3 if khz else None,
                            "fp32_valu_peak_from_these": p.multi_processor_count * 256 * khz * 1e3 / 1e12 if khz else None})
        except Exception as e:                       # (informative only)
            _DEVICE.update({"error": str(e)})
    return dict(_DEVICE)


# cycles per wave64 VALU instruction per SIMD, measured on this GPU (scripts/ubench/valu_forms.hip): FP32 add/mul/fma from
# registers 2.5, the same with an SGPR operand 4.1 (the PMC mix cannot tell the two apart: low and high estimate); v_rcp/v_sqrt 8.2;
# compares, selects, min/max, conversions, integer and lane ops 4.1
CPI_ARITH_LOW, CPI_ARITH_HIGH, CPI_TRANS, CPI_OTHER = 2.5, 4.1, 8.2, 4.1
XGMI_LINK_GBPS = 153.0        # per link and direction; 7 links per GPU, point to point (MI355X_MICROARCH.md)
HBM_GBPS = 8000.0
SIMDS = 1024
CLOCK_GHZ = 2.4


def stamped(name, workload, key):
    """profiles/<name>[workload][key] if that entry was measured on the sources this run uses (raytracing_amd/srchash.py), else None."""
    e = read_profile_json(name).get(workload)
    if not isinstance(e, dict) or e.get("source_hash") != source_hash():
        return None
    return e if key is None else e.get(key)


def issue_use_of(workload, kernel_ms, fpl, share=1.0):
    """Share of the SIMDs' vector issue slots the dominant kernel fills: VALU instructions per launch (rocprofv3 --pmc pass kept under
    profiles/, same sources) priced at the measured cycles per instruction, over SIMDs x the kernel's cycles in THIS run."""
    e = stamped("issue.json", workload, None)
    if not e or not kernel_ms:
        return None
    k = fpl / float(e.get("frames_per_launch") or fpl) * share    # (the profiled launch may have held another number of frames; share: the part of a frame this launch renders)
    arith, trans = e["valu_fp32_arith"] * k, e["valu_trans"] * k
    other = e["valu"] * k - arith - trans
    slots = SIMDS * kernel_ms * 1e-3 * CLOCK_GHZ * 1e9
    # low: every non-transcendental instruction at the fastest rate measured (2.5: register-only arithmetic, moves, logic); high: FP32
    # arithmetic with an SGPR operand and everything else at 4.1 — capped at 1: the kernel cannot fill more slots than there are, so
    # a raw value above 1 says the mix is cheaper than the high prices (cycles_per_valu_if_issue_bound is the price at which it
    # would be exactly 1)
    low = ((arith + other) * CPI_ARITH_LOW + trans * CPI_TRANS) / slots
    high_raw = (arith * CPI_ARITH_HIGH + trans * CPI_TRANS + other * CPI_OTHER) / slots
    return {"low": min(1.0, low), "high": min(1.0, high_raw), "high_uncapped": high_raw, "valu_per_wave": e["valu"] / e["waves"], "salu_per_wave": e["salu"] / e["waves"],
            "cycles_per_valu_if_issue_bound": slots / (e["valu"] * k),
            "waves_resident_per_simd": e["wave_cycles_x4"] * 4.0 / (e["gui_active_all_xcds"] / 8.0 * SIMDS) if e.get("gui_active_all_xcds") else None,
            "profile_round": e.get("round"), "kernel": e.get("kernel")}


def roofline_of(flops_per_launch, kernel_ms, fpl, launches, workload, share=1.0):
    """share < 1 (row-tile mode): this launch renders that part of ONE frame; profiles/ holds whole-frame launches of the same kernel family,
    so traffic and the instruction counts behind valu_issue_use are scaled by it (profile_scaled_by says so)."""
    achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
    pmc = stamped("traffic.json", workload, "bytes_per_launch")
    e_fpl = stamped("traffic.json", workload, "frames_per_launch") or stamped("issue.json", workload, "frames_per_launch")
    scaled = None
    if share != 1.0:
        scaled = (fpl / float(e_fpl) if e_fpl else None)
        scaled = scaled * share if scaled else None
        pmc = pmc * scaled if (pmc is not None and scaled) else None
    issued = stamped("issued.json", workload, "tflops")
    stale = read_profile_json("issue.json").get(workload) is not None and stamped("issue.json", workload, None) is None
    return {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": pmc, "issued_tflops": issued,
            "valu_issue_use": issue_use_of(workload, kernel_ms, fpl, share), "profile_source_hash": source_hash(), "profile_stale": stale, "profile_scaled_by": scaled,
            "kernel_ms": kernel_ms, "flops_per_launch": flops_per_launch, "frames_per_launch": fpl, "launches": launches,
            "note": "FP32 VALU roofline (SURVEY 8d): nothing here is HBM- or MFMA-shaped; algorithmic FLOPs = 14/plane + 6/point + "
                    "17/edge term + 22/box + 32|29|30 per ray, counted in the traversal that is timed (brute force: the reference's own loop over every triangle — "
                    "the kernel's box tests and sign filters skip part of that work, so frac can pass 1 on a small scene); valu_issue_use = share of the SIMDs' "
                    "vector issue slots the kernel fills in this run (instruction counts of the PMC pass under profiles/ x measured cycles "
                    "per instruction / (1024 SIMDs x kernel cycles)); issued_tflops = VALU FLOPs the kernel really executes per second; "
                    "traffic, issued_tflops and valu_issue_use are null when profiles/ was measured on other sources (profile_stale)"}


def side_config(api, torch, name, dev, args, launches=6, build=None):
    """One of the other BASELINE configs, measured on its own: `launches` timed launches after two warm-up launches."""
    w = Workload(api, torch, name, dev, args, build=build)
    try:
        batches = [w.batch(s) for s in range(launches)]
        w.count([f for b in batches for f in b])
        w.launch(batches[0])
        w.launch(batches[1])
        t_settle = time.perf_counter()                 # (settled clocks, as for the headline: run_frames_mode)
        k = 0
        while args.settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            w.launch(batches[k % len(batches)])
            torch.cuda.synchronize()
            k += 1
        wall, kernel_ms = w.timed_launches(batches)
        frames = launches * w.fpl
        rays = float(sum(w.rays_of[f] for b in batches for f in b))
        flops = float(sum(w.flops_of[f] for b in batches for f in b))
        out = {"workload": name, "label": w.label, "accel": w.accel_name, "traversal": w.traversal(), "width": w.W, "height": w.H,
               "max_bounces": w.D, "triangles": w.scene.n_triangles, "frames_per_launch": w.fpl, "launches": launches,
               "ms_per_frame": wall / frames * 1e3, "Mrays_per_s": rays / wall / 1e6, "rays_per_frame": rays / frames,
               "kernel_ms": kernel_ms, "kernel_ms_per_frame": kernel_ms / w.fpl,
               "roofline_frac": flops / launches / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
               "flops_per_ray": flops / rays}
        if build is None:
            out["valu_issue_use"] = issue_use_of(name, kernel_ms, w.fpl)    # (profiles/ holds the default build's kernels)
        if w.accel_name == "lbvh":
            out["build"] = w.build
            out["build_where"] = "host threads (topology) + GPU (boxes, records)" if w.build == "sah" else "GPU"
            out["build_ms"] = w.build_ms
            out["build_ms_first_commit"] = w.build_ms_first
            out["shadow_grid_build_ms"] = w.grid_ms
            out["shadow_grids"] = w.grids
            out["commit_wall_ms"] = w.commit_wall_ms
            out["bvh_depth"] = list(w.scene.bvh_depth(dev))
        return out
    finally:
        w.close()
        torch.cuda.empty_cache()


def hand_off(api, torch, w):
    """What one frame costs a viewer that wants it in host memory (Source.cpp:1270 uploads `data` every
    frame): a lone launch, rt_render into pageable memory, rt_render_async into two pinned buffers."""
    sc, W, H, D = w.scene, w.W, w.H, w.D
    n = 40
    bufs = [torch.empty(H * W * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
    s0 = torch.cuda.current_stream()
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    single = {}
    # back-to-back single-frame launches: on ONE stream a frame cannot start before the last wavefront of the one before has ended;
    # on two alternating streams (what rt_render_async does with its two slots) the ramp and tail of neighbours overlap
    for label, streams, reps in (("one_stream", [s0, s0], 10), ("two_streams", side, 10)):
        for timed in (False, True):
            torch.cuda.synchronize()
            if timed:
                ev0.record(s0)
                for st in side:
                    st.wait_stream(s0)
            for i in range(n * reps):
                sc.render_device(w.dev, w.cams[i % N_POSES], w.trans[i % N_POSES], W, H, D, api.STRICT, bufs[i & 1].data_ptr(), streams[i & 1].cuda_stream)
            if timed:
                for st in side:
                    s0.wait_stream(st)
                ev1.record(s0)
        torch.cuda.synchronize()
        single[label] = ev0.elapsed_time(ev1) / (n * reps) * 1e3
    single_us = single["two_streams"]
    del bufs
    sc.render(w.cams[0], w.trans[0], W, H, D)
    t0 = time.perf_counter()
    for i in range(n):
        sc.render(w.cams[i % N_POSES], w.trans[i % N_POSES], W, H, D)
    sync_ms = (time.perf_counter() - t0) / n * 1e3
    async_ms = {}
    for rgba8 in (False, True):
        bufs = [api.HostFrame(W, H, rgba8=rgba8), api.HostFrame(W, H, rgba8=rgba8)]
        tickets = [sc.render_async(w.cams[0], w.trans[0], W, H, D, bufs[0]), None]
        sc.render_wait(tickets[0])
        t0 = time.perf_counter()
        for i in range(n):
            tickets[i & 1] = sc.render_async(w.cams[i % N_POSES], w.trans[i % N_POSES], W, H, D, bufs[i & 1])
            if i > 0:
                sc.render_wait(tickets[(i - 1) & 1])
        sc.render_wait(tickets[(n - 1) & 1])
        async_ms[rgba8] = (time.perf_counter() - t0) / n * 1e3
        for b in bufs:
            b.close()
    mb = H * W * 12 / 1e6
    return {"single_frame_launch_us": single_us, "single_frame_launch_one_stream_us": single["one_stream"], "end_to_end_ms": sync_ms, "end_to_end_async_ms": async_ms[False],
            "frame_MB": mb, "async_GBps": mb / async_ms[False],
            "end_to_end_async_rgba8_ms": async_ms[True], "frame_rgba8_MB": mb / 3,
            "note": "one %dx%d frame per launch (HIP events, 10 x %d back-to-back launches, alternating between two streams as rt_render_async's two slots "
                    "do; _one_stream_us: all on one stream); end_to_end_ms = rt_render incl. the D2H copy into "
                    "pageable memory; end_to_end_async_ms = rt_render_async into two pinned frames, frame f copying while f+1 renders "
                    "(wall per frame); end_to_end_async_rgba8_ms = the same with RT_OUT_RGBA8 frames (the 8-bit pixels the viewer displays)" % (W, H, n)}


def read_profile_json(name):
    """Per-workload numbers from the PMC passes kept under profiles/ (collected by tools/profile.sh in
    separate rocprofv3 --pmc runs): traffic.json = HBM bytes per launch, issued.json = issued VALU TFLOP/s."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
            return d if isinstance(d, dict) else {}
    except (OSError, ValueError):
        return {}


def run_frames_mode(args, api, torch, dist, rank, world, dev, on_gpu, name):
    w = Workload(api, torch, name, dev, args)
    fpl = w.fpl
    mine = [w.batch(s, world, rank) for s in range(args.steps)]
    w.count([f for b in mine for f in b])
    for b in mine:                          # marshal every batch of the timed loop ahead of it
        w.prepare(b)
    for s in range(args.warmup):
        w.launch(w.batch(s, world, rank))
    # Settling (untimed, like the warm-up it extends): the engine clock of an MI355X keeps rising for tens of milliseconds after
    # work starts — in a kernel trace of this loop the same batch ran 5 % faster at the 25th launch than at the 1st — and W = 5
    # launches of 0.4 ms end long before that.  The timed region then measures the rate a renderer runs at, not the ramp.
    t_settle = time.perf_counter()
    k = 0
    while args.settle_ms > 0 and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        w.launch(mine[k % len(mine)])
        k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for b in mine:
        w.launch(b)
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream: mean launch duration
    my_rays = float(sum(w.rays_of[f] for b in mine for f in b))
    my_flops = float(sum(w.flops_of[f] for b in mine for f in b))
    total_rays = my_rays
    if dist:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        r = torch.tensor([my_rays], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = float(r[0].item())
    out = None
    if rank == 0:
        frames_total = args.steps * fpl * world
        out = {
            "metric": "Mrays/s (primary+bounce+shadow) at 800x800" if name.endswith("800") else "Mrays/s (primary+bounce+shadow)",
            "value": total_rays / wall / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_ms": args.settle_ms,
            "ms_per_step": wall / args.steps * 1e3, "frames_per_step": fpl, "ms_per_frame": wall / (args.steps * fpl) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w.label, "accel": w.accel_name, "traversal": w.traversal(), "build": w.build, "width": w.W, "height": w.H,
                       "max_bounces": w.D, "triangles": w.scene.n_triangles, "lights": w.scene.n_lights, "frames_per_step": fpl,
                       "frames": "64-pose camera path (rt_camera_path: look right + forward, 16 ms frames); step s of rank r = one launch "
                                 "over the %d consecutive poses from ((s*N + r)*%d) mod 64" % (fpl, fpl),
                       "rays_per_frame_mean": total_rays / frames_total, "parallelism": "frames x%d" % world},
            "roofline": roofline_of(my_flops / args.steps, kernel_ms, fpl, args.steps, name),
            "device": device_info(torch, dev),
        }
        if world == 1 and not args.no_configs:
            out["hand_off"] = hand_off(api, torch, w)
    w_desc = w.desc
    w.close()
    torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        if not args.no_configs:
            # (config 4 through the tree of every build: the GPU-built ones are north_star's, the host's SAH tree is the yardstick)
            out["configs"] = [side_config(api, torch, n, dev, args, build=b) for n, b in
                              (("stock800", None), ("mirrors4k", None), ("spheres800", None), ("spheres800", "fast"), ("spheres800", "sah"),
                               ("spheres4k", None), ("spheres4k", "fast"), ("spheres4k", "sah"), ("sweep8k", None), ("sweep8k_spheres", None)) if n != name]
        if not args.no_cpu:
            cb = cpu_baseline(w_desc, args.cpu_seconds)
            out["cpu_baseline"] = cb
            threads = min(16, os.cpu_count() or 1)
            mt = cpu_threaded(w_desc, threads)
            out["cpu_threaded"] = {"value": cb["rays_per_frame"] / mt / 1e6, "unit": "Mrays/s",
                                   "cores": threads, "kind": "port", "ms_per_frame": mt * 1e3,
                                   "sample": "row-interleaved pthreads over the same frame (stands in for the reference's disabled std::execution::par loop)"}
            # (the ratio on ONE frame: the CPU legs render the scene's own pose, so the GPU is timed on that pose too — the
            # headline value averages a 64-pose path with other ray counts)
            accel = api.ACCEL_LBVH if WORKLOADS[name][1] == "lbvh" else api.ACCEL_BRUTE
            same_ms = gpu_same_pose(api, torch, w_desc, dev, accel)
            out["gpu_same_pose"] = {"ms_per_frame": same_ms, "Mrays_per_s": cb["rays_per_frame"] / same_ms / 1e3,
                                    "note": "32 copies per launch of the pose the cpu_baseline renders"}
            out["gpu_over_cpu_1thread"] = out["gpu_same_pose"]["Mrays_per_s"] / cb["value"]
            if not args.no_configs:
                from raytracing_amd import scenes
                rows = []
                if True:
                    rows.append(cpu_gpu_row(api, torch, scenes.room(400, 400), dev, (("brute", api.ACCEL_BRUTE),), min(args.cpu_seconds, 6.0),
                                            "BASELINE config 1: the reference's room at 400x400 (Source.cpp:69-70 as shipped)"))
                big = Workload._descs.get("spheres") or WORKLOADS["spheres800"][0](scenes)
                rows.append(cpu_gpu_row(api, torch, big.with_size(64, 64), dev, (("lbvh", api.ACCEL_LBVH), ("brute", api.ACCEL_BRUTE)), 1.0,
                                        "BASELINE config 4 at reduced size: 102 412 triangles, 64x64, the reference's loop over every triangle "
                                        "(one frame is ~10 s of one host core; 800x800 would be ~25 min)"))
                out["cpu_rows"] = rows
    return out


def run_tile_mode(args, api, torch, dist, rank, world, dev, on_gpu, name):
    """BASELINE config 5: every rank renders its interleaved 16-row tiles of the SAME frame, one exchange
    step per frame gathers them on rank 0 (pipelined: frame f+1 renders while f is in flight).  The frames
    travel as packed 8-bit pixels (RT_OUT_RGBA8: what the reference's viewer displays, a third of the bytes,
    SURVEY 8e) unless --exchange f32; the other format is measured on a few frames as a secondary figure."""
    from raytracing_amd import sharding
    w = Workload(api, torch, name, dev, args, frames_per_launch=1)
    W, H, D, sc = w.W, w.H, w.D, w.scene
    frames = [s % N_POSES for s in range(args.steps)]
    w.count(frames)
    tile_rows = 16
    rows = api.shard_rows(H, tile_rows, rank, world)
    where = "cuda" if on_gpu else "cpu"

    def leg(fmt, frames, warmup, check):
        """Timed exchange of `frames` in format `fmt` -> (wall s, this rank's mean kernel ms, gather check or None)."""
        rgba8 = fmt == "rgba8"
        ch, dtype, mode = (1, torch.int32, api.STRICT | api.OUT_RGBA8) if rgba8 else (3, torch.float32, api.STRICT)
        ex = sharding.TileExchange(dist, rank, world, H, W, tile_rows, lambda n: torch.empty(n, dtype=dtype, device=where), depth=2, channels=ch)
        # (gloo rehearsal: the shards go through host memory — rt_render_tiles renders and copies back inside the library)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in frames]

        def run(fs, timed):
            done = []
            for i, f in enumerate(fs):
                buf = ex.shard(i)
                if timed:
                    evs[i][0].record()
                if on_gpu:
                    sc.render_tiles_device(dev, w.cams[f], w.trans[f], W, H, D, mode, tile_rows, rank, world, buf.data_ptr(), w.stream)
                else:
                    sc.render_tiles(w.cams[f], w.trans[f], W, H, D, mode, tile_rows, rank, world, out=buf.data_ptr())
                if timed:
                    evs[i][1].record()
                done += ex.submit(i)
            done += ex.drain()
            return done

        run([frames[s % len(frames)] for s in range(max(1, warmup))], False)    # (at least one untimed exchange: the first one sets up the communicator)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(frames, True)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)      # this rank's shard kernel, HIP events on the launch stream
        verdict = None
        if check:
            # outside the timed region: one more frame through the same exchange, compared on rank 0 with the frame
            # one GPU renders in one piece — the gathered framebuffer must be that frame, bit for bit
            got = run([frames[-1]], False)
            if rank == 0:
                whole = torch.empty(H * W * ch, dtype=dtype, device="cuda")
                sc.render_device(dev, w.cams[frames[-1]], w.trans[frames[-1]], W, H, D, mode, whole.data_ptr(), w.stream)
                torch.cuda.synchronize()
                same = torch.equal(got[-1][1].to("cuda").view(torch.int32), whole.view(torch.int32))
                verdict = "bit-identical to the single-GPU frame" if same else "DIFFERS from the single-GPU frame"
                if rgba8 and same:
                    # and the 8-bit frame is the GL unorm conversion of the binary32 frame
                    f32 = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
                    sc.render_device(dev, w.cams[frames[-1]], w.trans[frames[-1]], W, H, D, api.STRICT, f32.data_ptr(), w.stream)
                    torch.cuda.synchronize()
                    c = torch.where(f32 > 0, f32, torch.zeros_like(f32)).clamp_(max=1.0)
                    q = (c * 255.0 + 0.5).to(torch.int32).view(H * W, 3)
                    want = q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16) | torch.tensor(-16777216, dtype=torch.int32, device="cuda")
                    verdict += "; = unorm8 of the binary32 frame" if torch.equal(want, whole) else "; DIFFERS from unorm8 of the binary32 frame"
                    del f32
                del whole
        t = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=where)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        del ex
        return float(t[0].item()), kernel_ms, float(t[1].item()), verdict

    # the same frames by ONE GPU in one piece (rank 0, outside the timed region): the N=1 point of this workload.  The other ranks
    # wait at the barriers: in the one-GPU rehearsal (gloo) they share rank 0's GPU and their counting launches would be timed with it.
    single = None
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        whole = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
        k = min(8, len(frames))
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in frames[:k]:
                sc.render_device(dev, w.cams[f], w.trans[f], W, H, D, api.STRICT, whole.data_ptr(), w.stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter() - t0
        single = {"ms_per_frame": t1 / k * 1e3, "value": sum(w.rays_of[f] for f in frames[:k]) / t1 / 1e6, "frames": k,
                  "note": "whole binary32 frames on rank 0's GPU alone, one frame per launch, frames resident in HBM, no exchange"}
        del whole
    torch.cuda.synchronize()
    dist.barrier()
    fmt = args.exchange
    other = "f32" if fmt == "rgba8" else "rgba8"
    wall, kernel_ms, kernel_ms_max, check = leg(fmt, frames, args.warmup, True)
    k2 = min(8, len(frames))
    wall2, _, kmax2, _ = leg(other, frames[:k2], min(2, args.warmup), False)
    bytes_px = {"f32": 12, "rgba8": 4}
    out = None
    if rank == 0:
        rays = float(sum(w.rays_of[f] for f in frames))
        flops = float(sum(w.flops_of[f] for f in frames))
        share = rows / H
        out = {
            "metric": "Mrays/s (primary+bounce+shadow)", "value": rays / wall / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "frames_per_step": 1,
            "ms_per_frame": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": w.label + ", interleaved 16-row tiles + gather to rank 0", "accel": w.accel_name, "traversal": w.traversal(),
                       "build": w.build, "width": W, "height": H, "max_bounces": D, "triangles": sc.n_triangles, "lights": sc.n_lights,
                       "frames": "64-pose camera path (rt_camera_path), step s = pose s mod 64, every rank renders its tiles of it",
                       "rays_per_frame_mean": rays / args.steps, "parallelism": "row tiles x%d" % world,
                       "output": ("packed 8-bit RGBA pixels (RT_OUT_RGBA8: the GL unorm conversion of the binary32 colour, what the reference's viewer displays)"
                                  if fmt == "rgba8" else "binary32 RGB (the reference's `data`)"),
                       "exchange": "one non-blocking gather of the compact %s shards to rank 0 per frame (%s; shards padded to a common size), two shard buffers per rank" % (fmt, args.backend)},
            "roofline": roofline_of(flops / args.steps * share, kernel_ms, 1, args.steps, name, share),
            "device": device_info(torch, dev),
            "exchange": {"format": fmt, "kernel_ms_rank0": kernel_ms, "kernel_ms_max_over_ranks": kernel_ms_max,
                         "non_render_ms_per_frame": wall / args.steps * 1e3 - kernel_ms_max,
                         "shard_MB": rows * W * bytes_px[fmt] / 1e6,
                         # the bound the links set: every other rank's shard crosses ITS OWN xGMI link into rank 0's GPU (7 links x
                         # ~153 GB/s per GPU, MI355X_MICROARCH.md), concurrently — one shard over one link; and the render at its best
                         "model_ms": (rows * W * bytes_px[fmt]) / XGMI_LINK_GBPS / 1e6 if world > 1 else 0.0,
                         "scatter_model_ms": 2.0 * H * W * bytes_px[fmt] / HBM_GBPS / 1e6,
                         "render_ms_ideal": single["ms_per_frame"] / world if single else None,
                         "de_interleave": "rt_scatter_shards (one kernel of the library on rank 0 per frame)" if on_gpu
                                          else "rt_scatter_shards_host (one memcpy per tile inside the library)",
                         "note": "roofline is rank 0's shard kernel: the frame's counted FLOPs x its share of the rows (%d of %d, interleaved "
                                 "tiles); non_render_ms_per_frame = wall per frame minus the slowest rank's kernel = exchange + "
                                 "de-interleave not hidden behind rendering; model_ms = shard bytes / 153 GB/s (one xGMI link per peer, "
                                 "all peers at once); scatter_model_ms = frame bytes read + written at 8 TB/s; render_ms_ideal = the "
                                 "single-GPU frame / ranks" % (rows, H)},
            "exchange_other_format": {"format": other, "frames": k2, "ms_per_frame": wall2 / k2 * 1e3,
                                      "value": float(sum(w.rays_of[f] for f in frames[:k2])) / wall2 / 1e6,
                                      "kernel_ms_max_over_ranks": kmax2, "shard_MB": rows * W * bytes_px[other] / 1e6},
            "single_gpu": single,
            "gather_check": check,
        }
        if check is not None and "DIFFERS" in check:
            raise SystemExit("bench.py: the gathered frame differs from the single-GPU frame")
    w.close()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    from raytracing_amd import api

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

    name = args.workload or ("room800" if world == 1 else "sweep8k")
    tile_mode = world > 1 and name.startswith("sweep8k")
    if tile_mode:
        out = run_tile_mode(args, api, torch, dist, rank, world, dev, on_gpu, name)
        if not args.no_configs and name != "sweep8k_spheres":
            # the heavier partition of the same frame size (102 412 triangles through the tree): it shards better than the room
            steps, warm = args.steps, args.warmup
            args.steps, args.warmup = min(8, steps), min(2, warm)
            side = run_tile_mode(args, api, torch, dist, rank, world, dev, on_gpu, "sweep8k_spheres")
            args.steps, args.warmup = steps, warm
            if rank == 0:
                out["configs"] = [{"workload": "sweep8k_spheres", "label": side["config"]["workload"], "n_gpus": world, "steps": side["steps"],
                                   "value": side["value"], "unit": side["unit"], "ms_per_frame": side["ms_per_frame"], "scaling": "strong",
                                   "build": side["config"]["build"], "traversal": side["config"]["traversal"],
                                   "roofline_frac": side["roofline"]["frac"], "exchange": side["exchange"], "single_gpu": side["single_gpu"],
                                   "gather_check": side["gather_check"]}]
    else:
        out = run_frames_mode(args, api, torch, dist, rank, world, dev, on_gpu, name)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
