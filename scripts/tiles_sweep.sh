# brute-force workloads over tiles per wavefront (round 2; scripts/tpw_sweep.sh is the general form)
for w in room800 mirrors4k sweep8k; do for t in 1 2 4 8; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --workload $w --tiles-per-wave $t | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w tpw=$t', round(d['ms_per_frame']*1000,2), 'us/frame')" || exit 1
done; done
