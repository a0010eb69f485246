#!/bin/bash
# tools/profile.sh <round-tag> ["<workloads>"] — run on the GPU box (gpurun).  For the headline workload (room800) and the
# tree workloads (spheres800, spheres4k) — or the bench.py workloads named in the second argument — it collects, each in a run of its own:
#   (1) rocprofv3 --kernel-trace --stats of the bench command            -> <tag>_<w>_kernel_stats.csv
#   (2) HBM traffic: FETCH_SIZE and WRITE_SIZE in separate --pmc passes   -> traffic.json (bytes per launch)
#   (3) SQ instruction and wait counters                                  -> <tag>_<w>_pmc.json
#   (4) the VALU instruction mix (two passes)                             -> issued.json (issued VALU TFLOP/s)
# Summaries go to gpurun_out/profiles_<tag>/; copy what should be judged into profiles/.
tag=${1:-r02}
workloads=${2:-"room800 spheres800 spheres4k"}
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
for w in $workloads; do
  args="--steps 8 --warmup 2 --no-cpu --no-configs --workload $w"
  # (the kernel trace runs the default step counts, so that its average is over launches at steady clocks like bench.py's own)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -o $w -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-configs --workload $w > $out/${tag}_${w}_bench_under_rocprof.json 2> $out/trace_$w.err || exit 1
  cp $out/trace_$w/${w}_kernel_stats.csv $out/${tag}_${w}_kernel_stats.csv
  i=0
  for pass in "FETCH_SIZE" "WRITE_SIZE" \
              "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
              "SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM" \
              "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
              "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_${w}_$i -o p -- python3 bench.py $args > $out/pmc_${w}_$i.json 2> $out/pmc_${w}_$i.err || exit 1
  done
done
python3 - $out $tag $workloads <<'PY'
import csv, sys, json, collections, glob, os
sys.path.insert(0, os.getcwd())
from raytracing_amd.srchash import source_hash
out, tag = sys.argv[1], sys.argv[2]
stamp = source_hash()                      # (what bench.py compares with the sources it runs: a profile of other code is not quoted)
traffic, issued, issue = {}, {}, {}
for w in sys.argv[3:]:
    # the dominant kernel of the run: the render kernel with the largest total time in the kernel trace
    # (room800: render_lockstep_kernel<0>; the tree scenes: <2> walked per lane or <3> by packets, as the launcher chose)
    rows = [r for r in csv.DictReader(open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, w)))) if "render_lockstep_kernel" in r["Name"] and ", true>(" not in r["Name"]]   # (not the counting instance bench.py runs once per pose, outside the timed region)
    kern = max(rows, key=lambda r: float(r["TotalDurationNs"]))["Name"]
    kern = kern[kern.index("render_lockstep_kernel"):].split("(")[0]
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_%s_*" % w, "p_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    summ = {k: sum(v) / len(v) for k, v in acc.items()}
    summ["_kernel"] = kern
    summ["_launches_sampled"] = {k: len(v) for k, v in acc.items()}
    # kernel time of the un-counted bench run under the kernel trace (same command)
    ns = None
    for r in csv.DictReader(open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, w)))):
        if kern in r["Name"]:
            ns = float(r["AverageNs"])
    summ["kernel_trace_average_ns"] = ns
    # FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half the bytes of a
    # wide coalesced read stream (uncalibrated for other widths); WRITE_SIZE is exact for 16-B stores.
    fetch_kib, write_kib = summ.get("FETCH_SIZE", 0.0), summ.get("WRITE_SIZE", 0.0)
    summ["hbm_bytes_per_launch"] = {"fetch_raw": fetch_kib * 1024, "fetch_x2_gfx950": 2 * fetch_kib * 1024, "write": write_kib * 1024,
                                    "total_corrected": (2 * fetch_kib + write_kib) * 1024}
    traffic[w] = {"bytes_per_launch": summ["hbm_bytes_per_launch"]["total_corrected"], "write_bytes_per_launch": write_kib * 1024,
                  "source_hash": stamp, "round": tag, "kernel": kern}
    summ["source_hash"] = stamp
    wv = summ.get("SQ_WAVES", 1.0)
    summ["per_wave"] = {k: summ[k] / wv for k in summ if k.startswith("SQ_INSTS")}
    # issued VALU FLOPs: one per lane for add/mul/transcendental, two for fma, 64 lanes per wave instruction (an upper
    # bound: masked-off lanes are counted) — the secondary figure SURVEY 8d asks for beside the algorithmic one
    fl = 64.0 * (summ.get("SQ_INSTS_VALU_ADD_F32", 0) + summ.get("SQ_INSTS_VALU_MUL_F32", 0) + 2 * summ.get("SQ_INSTS_VALU_FMA_F32", 0)
                 + summ.get("SQ_INSTS_VALU_TRANS_F32", 0) + summ.get("SQ_INSTS_VALU_ADD_F64", 0) + summ.get("SQ_INSTS_VALU_MUL_F64", 0)
                 + 2 * summ.get("SQ_INSTS_VALU_FMA_F64", 0))
    summ["issued_valu_flops_per_launch"] = fl
    if ns:
        summ["issued_valu_tflops"] = fl / (ns * 1e-9) / 1e12
        issued[w] = {"tflops": summ["issued_valu_tflops"], "source_hash": stamp, "round": tag, "kernel": kern}
    # Vector issue slots: VALU instructions per launch by class, priced by bench.py with the issue rates
    # measured on this GPU (scripts/ubench/valu_forms.hip: FP32 add/mul/fma 2.5 cycles per wave64 instruction from registers,
    # 4.1 with an SGPR operand; transcendental 8.2; everything else 4.1) against SIMDs x the kernel's cycles of the LIVE run.
    arith = summ.get("SQ_INSTS_VALU_ADD_F32", 0) + summ.get("SQ_INSTS_VALU_MUL_F32", 0) + summ.get("SQ_INSTS_VALU_FMA_F32", 0)
    trans = summ.get("SQ_INSTS_VALU_TRANS_F32", 0)
    try:
        fpl = json.loads(open(os.path.join(out, "pmc_%s_3.json" % w)).read().strip().splitlines()[-1])["config"]["frames_per_step"]
    except Exception:
        fpl = None
    issue[w] = {"frames_per_launch": fpl, "valu": summ.get("SQ_INSTS_VALU", 0), "valu_fp32_arith": arith, "valu_trans": trans, "salu": summ.get("SQ_INSTS_SALU", 0),
                "smem": summ.get("SQ_INSTS_SMEM", 0), "waves": wv, "wave_cycles_x4": summ.get("SQ_WAVE_CYCLES", 0),
                "gui_active_all_xcds": summ.get("GRBM_GUI_ACTIVE", 0), "kernel_ns_under_trace": ns,
                "source_hash": stamp, "round": tag, "kernel": kern}
    summ["issue"] = issue[w]
    json.dump(summ, open(os.path.join(out, "%s_%s_pmc.json" % (tag, w)), "w"), indent=1)
    print(w, json.dumps(summ, indent=1))
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
json.dump(issued, open(os.path.join(out, "issued.json"), "w"), indent=1)
json.dump(issue, open(os.path.join(out, "issue.json"), "w"), indent=1)
PY
for w in $workloads; do cut -c1-200 $out/${tag}_${w}_kernel_stats.csv | head -8; done
