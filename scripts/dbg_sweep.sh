#!/bin/bash
# us/frame of bench workloads under the experiment build's RT355_GRID_DEBUG switches (attribution of frame time; outputs differ by design)
export RT355_LIB=$PWD/build/librt355_exp.so
for wl in ${WLS:-spheres800 spheres4k}; do
  for dbg in ${DBGS:-0 1 2 3 8}; do
    export RT355_GRID_DEBUG=$dbg
    r=$(timeout -k 10 120 python bench.py --workload $wl --no-cpu --no-configs --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_frame']*1e3,1))") || exit 1
    echo "$wl debug=$dbg us/frame $r"
  done
done
