# usage: ab6.sh variants...  (all GPU tests on the first, then brute-force timings, interleaved)
RT355_LIB=$PWD/build/librt355_$1.so timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 100 --timeout-method thread 2>&1 | tail -1
for w in room800 stock800 mirrors4k; do for rep in 1 2; do for v in "$@"; do
  RT355_LIB=$PWD/build/librt355_$v.so timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $v', round(d['value']), 'Mrays/s', round(d['ms_per_frame']*1000,2), 'us/frame', 'kernel', round(d['roofline']['kernel_ms']*1000/d['roofline']['frames_per_launch'],2))" || exit 1
done; done; done
