mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t4.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/t4.log
[ $rc -eq 0 ] || exit 1
for b in 64 256; do for t in 1; do RT355_BLOCK_THREADS=$b timeout -k 10 60 python bench.py --steps 512 --warmup 32 --no-cpu --tiles-per-wave $t | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('block', $b, 'tpw', $t, round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))" || exit 1; done; done
timeout -k 10 120 python bench.py --steps 64 --warmup 8 --no-cpu --workload mirrors4k | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mirrors4k', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))"
