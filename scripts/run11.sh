export TMPDIR=/tmp
timeout -k 10 500 python -u -m pytest tests/test_gpu_lbvh.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for v in lw4 lw5 lw6 lw8; do for w in spheres800 spheres4k; do RT355_LIB=$PWD/build/librt355_$v.so timeout -k 10 200 python bench.py --steps 32 --warmup 16 --no-cpu --workload $w | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $w', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us/frame frac', round(d['roofline']['frac'],4))" || exit 1; done; done
timeout -k 10 100 python scripts/lbvh_stats.py | tail -2
