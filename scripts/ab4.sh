# usage: ab4.sh "<bench args>" variants...
args=$1; shift
for rep in 1 2; do for v in "$@"; do
  RT355_LIB=$PWD/build/librt355_$v.so timeout -k 10 200 python bench.py --steps 32 --warmup 16 --no-cpu $args | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us/frame frac', round(d['roofline']['frac'],4))" || exit 1
done; done
