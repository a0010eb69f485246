# usage: ab2.sh <workload> variants...   timing then VALU/wave per variant
mkdir -p gpurun_out; export TMPDIR=/tmp
wl=$1; shift
export RT355_LIB=$PWD/build/librt355_$1.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for rep in 1 2; do for v in "$@"; do
  export RT355_LIB=$PWD/build/librt355_$v.so
  timeout -k 10 60 python bench.py --steps 1024 --warmup 128 --no-cpu --workload $wl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $v', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))" || exit 1
done; done
for v in "$@"; do
  export RT355_LIB=$PWD/build/librt355_$v.so
  echo "== $v"; bash scripts/pmc.sh $v --workload $wl | grep -A12 "false" | grep -E "VALU/wave"
done
