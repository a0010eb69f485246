# us/frame of the three tree workloads of bench.py with the library in the tree (quick check after a kernel change)
cd /root/repo
for wl in spheres800 spheres4k sweep8k_spheres; do
  timeout -k 10 120 python3 bench.py --workload $wl --steps 12 --warmup 4 --no-configs --no-cpu > gpurun_out/prod_${wl}.json 2>gpurun_out/prod_err.log || exit 1
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/prod_${wl}.json') if l.startswith('{')][-1])
print('$wl', round(d['ms_per_frame']*1e3,1), 'us/frame')"
done
