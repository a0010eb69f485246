#!/bin/bash
# us/frame of bench workloads against tiles per wave (0 = the launcher's own choice)
for wl in ${WLS:-spheres800 spheres4k}; do
  for t in ${TPWS:-0 1 2 3 4 6 8}; do
    r=$(timeout -k 10 120 python bench.py --workload $wl --no-cpu --no-configs --steps 10 --warmup 3 --tiles-per-wave $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_frame']*1e3,1))") || exit 1
    echo "$wl tpw=$t us/frame $r"
  done
done
