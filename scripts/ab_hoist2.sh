# usage: ab_hoist2.sh variants...   spheres800 / spheres4k, packets, with and without hoisting, interleaved
for w in spheres800 spheres4k; do for rep in 1 2; do for v in "$@"; do for nh in 0 1; do
  if [ $nh = 1 ]; then export RT355_NO_HOIST=1; else unset RT355_NO_HOIST; fi
  RT355_LIB=$PWD/build/librt355_$v.so timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-configs --workload $w | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w $v nohoist=$nh', d['config'].get('traversal'), round(d['value']), 'Mrays/s', round(d['ms_per_frame']*1000,1), 'us/frame')" || exit 1
done; done; done; done
