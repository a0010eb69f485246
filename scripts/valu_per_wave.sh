#!/bin/bash
# VALU / SALU instructions per wave of a bench workload's render kernel under the experiment build's RT355_GRID_DEBUG switches
export TMPDIR=/tmp RT355_LIB=$PWD/build/librt355_exp.so
wl=${1:-spheres4k}; shift
for dbg in ${@:-0}; do
  export RT355_GRID_DEBUG=$dbg
  rm -rf gpurun_out/vw_$dbg
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/vw_$dbg -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-configs --workload $wl > /dev/null 2>&1 || exit 1
  python3 - $dbg gpurun_out/vw_$dbg/p_counter_collection.csv <<'PY'
import csv, sys, collections
dbg, f = sys.argv[1:3]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "render_lockstep_kernel" in r["Kernel_Name"] and ", true>(" not in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m["SQ_WAVES"]
print("debug", dbg, " ".join("%s/wave %.0f" % (k[3:], m[k] / w) for k in sorted(m) if k != "SQ_WAVES"))
PY
done
