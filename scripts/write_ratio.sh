#!/bin/bash
# WRITE_SIZE / frame bytes of a bench workload's render kernel (rocprofv3 --pmc WRITE_SIZE), for the tiles-per-wave values given
export TMPDIR=/tmp
wl=${1:-sweep8k}; shift
for t in ${@:-0}; do
  rm -rf gpurun_out/wr_$t
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/wr_$t -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $wl --tiles-per-wave $t $EXTRA > gpurun_out/wr_$t.json 2>/dev/null || exit 1
  python3 - $t gpurun_out/wr_$t/p_counter_collection.csv gpurun_out/wr_$t.json <<'PY'
import csv, sys, json
t, f, j = sys.argv[1:4]
d = json.loads(open(j).read().strip().splitlines()[-1])
c = d["config"]
frame = c["width"] * c["height"] * 12 * c["frames_per_step"]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "render_lockstep_kernel" in r["Kernel_Name"] and ", true>(" not in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
print("tpw", t, "per launch", [round(x * 1024 / frame, 3) for x in v], "launches", len(v), "write/frame bytes %.4f" % (sum(v) / len(v) * 1024 / frame), "us/frame %.1f" % (d["ms_per_frame"] * 1e3))
PY
done
