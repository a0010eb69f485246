# usage: ab5.sh variants...  (LBVH tests on the first, then 800 and 4k timings, interleaved)
RT355_LIB=$PWD/build/librt355_$1.so timeout -k 10 300 python -u -m pytest tests/test_gpu_lbvh.py tests/test_gpu_parity.py -m gpu -x -q --timeout 100 --timeout-method thread -k "lbvh or tree or soup or far_origin" 2>&1 | tail -1
for w in spheres800 spheres4k; do for rep in 1 2; do for v in "$@"; do
  RT355_LIB=$PWD/build/librt355_$v.so timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $v', round(d['value']), 'Mrays/s', round(d['ms_per_frame']*1000,1), 'us/frame')" || exit 1
done; done; done
