# usage: bash scripts/ab_lib.sh <other librt355.so> "<workloads>" [extra bench args]  — us/frame of the tree's library against another build of it
# (the other library is swapped in and out of raytracing_amd/; both runs in one call, alternating)
cur=raytracing_amd/librt355.so
cp $cur /tmp/ab_cur.so || exit 1
for w in $2; do for rep in 1 2; do for which in other cur; do
  if [ $which = other ]; then cp $1 $cur; else cp /tmp/ab_cur.so $cur; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-configs --workload $w $3 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w $which', round(d['ms_per_frame']*1000,2), 'us/frame frac', round(d['roofline']['frac'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config'].get('traversal'))" || { cp /tmp/ab_cur.so $cur; exit 1; }
done; done; done
cp /tmp/ab_cur.so $cur
