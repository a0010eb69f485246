# A/B several builds of librt355 in one box: timing (un-profiled) then VALU counts (profiled)
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "$@"; do
  export RT355_LIB=$PWD/build/librt355_$v.so
  timeout -k 10 60 python bench.py --steps 512 --warmup 32 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))" || exit 1
done
for v in "$@"; do
  export RT355_LIB=$PWD/build/librt355_$v.so
  echo "== $v"; bash scripts/pmc.sh $v | grep -A12 "Lb0\|false" | grep -E "VALU/wave|SQ_WAIT_INST_ANY|SQ_BUSY"
done
