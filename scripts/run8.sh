mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t8.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t8.log
[ $rc -eq 0 ] || exit 1
for w in room800 stock800 mirrors4k; do timeout -k 10 60 python bench.py --steps 512 --warmup 64 --no-cpu --workload $w | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))" || exit 1; done
bash scripts/pmc.sh cur | grep -A12 "false" | grep -E "VALU/wave"
RT355_LIB=$PWD/build/librt355_prof.so timeout -k 10 60 python scripts/waveprof.py room
