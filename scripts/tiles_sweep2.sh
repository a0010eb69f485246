# us/frame over tiles per wavefront for the tree scenes: bash scripts/tiles_sweep2.sh
for w in spheres800 spheres4k sweep8k_spheres; do for t in 1 2 4; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --workload $w --tiles-per-wave $t | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w tpw=$t', round(d['ms_per_frame']*1000,2), 'us/frame', d['config'].get('traversal'))" || exit 1
done; done
