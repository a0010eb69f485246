export TMPDIR=/tmp
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for f in 1 2 4 8 16; do timeout -k 10 100 python bench.py --steps 512 --warmup 64 --no-cpu --frames-per-launch $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fpl $f', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us/frame  frac', round(d['roofline']['frac'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))" || exit 1; done
