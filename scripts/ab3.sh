# usage: ab3.sh size variants...
sz=$1; shift
for rep in 1 2; do for v in "$@"; do
  RT355_LIB=$PWD/build/librt355_$v.so timeout -k 10 100 python bench.py --steps 64 --warmup 8 --no-cpu --size $sz $sz | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('room $sz $v', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))" || exit 1
done; done
