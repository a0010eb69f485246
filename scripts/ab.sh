#!/bin/bash
# scripts/ab.sh "<workload> ..." <tag> [<tag> ...]  — interleaved A/B of library builds (build/librt355_<tag>.so,
# made by scripts/prof_build.sh; tag "cur" = the in-tree library) on one box: us/frame per workload and build, 2 rounds.
wls=$1; shift
for round in 1 2; do
  for w in $wls; do
    for v in "$@"; do
      lib=$PWD/build/librt355_$v.so; [ "$v" = cur ] && lib=$PWD/raytracing_amd/librt355.so
      RT355_LIB=$lib timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --workload $w ${AB_ARGS} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $v', round(d['ms_per_frame']*1000,2), 'us/frame', round(d['value']), 'Mrays/s  kernel', round(d['roofline']['kernel_ms']*1000/d['roofline']['frames_per_launch'],2), 'frac', round(d['roofline']['frac'],3), d['config'].get('traversal'))" || exit 1
    done
  done
done
