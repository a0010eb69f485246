for w in mirrors4k sweep8k; do for t in 8 12 16 32; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --workload $w --tiles-per-wave $t | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w tpw=$t', round(d['ms_per_frame']*1000,2), 'us/frame')" || exit 1
done; done
for s in "1600 1600" "2400 2400"; do for t in 4 8; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --workload room800 --size $s --tiles-per-wave $t | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('room $s tpw=$t', round(d['ms_per_frame']*1000,2), 'us/frame')" || exit 1
done; done
