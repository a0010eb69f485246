# A/B of two library builds on the kernels a change to the tile loop touches besides the packet kernel: the shard instance (multi-GPU ranks) and the per-lane walk
cd /root/repo
for lib in build/librt355_row.so raytracing_amd/librt355.so; do
  echo "== $lib"
  RT355_LIB=$lib timeout -k 10 100 python3 scripts/shard_kernel_time.py 2>&1 | grep "us per" || exit 1
  for a in "--size 400 400" "--traversal per-lane" "--size 1600 1600 --traversal per-lane"; do
    RT355_LIB=$lib timeout -k 10 120 python3 bench.py --workload spheres800 $a --steps 12 --warmup 4 --no-configs --no-cpu > gpurun_out/ab.json 2>gpurun_out/ab_err.log || exit 1
    python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{')][-1])
print('spheres $a', d['config']['traversal'], round(d['ms_per_frame']*1e3,2), 'us/frame')"
  done
done
