#!/bin/bash
# tree kernels: us/frame against tiles per wave at frame sizes between 800x800 and 3840x2160 (102k triangles)
for sz in ${SIZES:-1200 1600 2000 2400}; do
  for t in ${TPWS:-1 2 3}; do
    r=$(timeout -k 10 120 python bench.py --workload spheres800 --size $sz $sz --no-cpu --no-configs --steps 8 --warmup 3 --tiles-per-wave $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_frame']*1e3,1), d['config'].get('frames_per_step'))") || exit 1
    echo "spheres ${sz}x${sz} tpw=$t us/frame, frames per launch: $r"
  done
done
