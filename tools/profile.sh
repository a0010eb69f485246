#!/bin/bash
# tools/profile.sh <round-tag> — run on the GPU box (gpurun): collects, for the default bench
# command, (1) the rocprofv3 kernel-trace summary, (2) HBM traffic counters in separate --pmc
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), (3) SQ instruction counters.
# Writes summaries under gpurun_out/profiles_<tag>/; copy what should be judged into profiles/.
tag=${1:-r01}
out=gpurun_out/profiles_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 4 --warmup 1 --no-cpu --no-configs"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o room800 -- python3 bench.py $args > $out/bench_under_trace.json 2> $out/trace.err || exit 1
cp $out/trace/room800_kernel_stats.csv $out/${tag}_room800_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs > /dev/null 2> $out/pmc_$name.err || exit 1
done
python3 - $out $tag <<'PY'
import csv, sys, json, collections, glob, os
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "render_kernel<0, false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: sum(v) / len(v) for k, v in acc.items()}
summ["_launches_sampled"] = {k: len(v) for k, v in acc.items()}
# FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half the bytes of a
# wide coalesced read stream (uncalibrated for other widths); WRITE_SIZE is exact for 16-B stores.
fetch_kib, write_kib = summ.get("FETCH_SIZE", 0.0), summ.get("WRITE_SIZE", 0.0)
summ["hbm_bytes_per_launch"] = {"fetch_raw": fetch_kib * 1024, "fetch_x2_gfx950": 2 * fetch_kib * 1024, "write": write_kib * 1024,
                                "total_corrected": (2 * fetch_kib + write_kib) * 1024}
json.dump(summ, open(os.path.join(out, tag + "_room800_pmc.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
cat $out/${tag}_room800_kernel_stats.csv

# ---- LBVH workload (config 4): kernel trace + SQ counters ----------------------------------------
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_lbvh -o spheres800 -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload spheres800 > $out/${tag}_spheres800_bench_under_rocprof.json 2> $out/trace_lbvh.err || exit 1
cp $out/trace_lbvh/spheres800_kernel_stats.csv $out/${tag}_spheres800_kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/pmc_lbvh -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload spheres800 > /dev/null 2> $out/pmc_lbvh.err || exit 1
python3 - $out $tag <<'PY'
import csv, sys, json, collections, os
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(out, "pmc_lbvh", "p_counter_collection.csv"))):
    if "render_kernel<2, false>" in r["Kernel_Name"] or "render_kernel<3, false>" in r["Kernel_Name"]:   # per-lane or packet traversal
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: sum(v) / len(v) for k, v in acc.items()}
w = summ.get("SQ_WAVES", 1.0)
summ["per_wave"] = {k: summ[k] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if k in summ}
json.dump(summ, open(os.path.join(out, tag + "_spheres800_pmc.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
cat $out/${tag}_spheres800_kernel_stats.csv
