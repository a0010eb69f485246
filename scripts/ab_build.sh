# usage: bash scripts/ab_build.sh "<lbvh workloads>"  — us/frame, build time and tree depth for both tree builds
for w in $1; do for tr in packet per-lane; do for b in fast sah; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-configs --workload $w --build $b --traversal $tr | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w $tr $b', round(d['ms_per_frame']*1000,2), 'us/frame', round(d['value']), 'Mrays/s frac', round(d['roofline']['frac'],3), d['config'].get('traversal'))" || exit 1
done; done; done
