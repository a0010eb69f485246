mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t5.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/t5.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do timeout -k 10 60 python bench.py --steps 512 --warmup 64 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('room800', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))" || exit 1; done
timeout -k 10 120 python bench.py --steps 64 --warmup 8 --no-cpu --workload mirrors4k | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mirrors4k', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))"
bash scripts/pmc.sh grp | grep -A12 "false" | grep -E "VALU/wave|SQ_WAIT_INST_ANY|SQ_BUSY"
