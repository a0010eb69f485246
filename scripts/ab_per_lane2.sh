# A/B of two library builds on what the per-lane walk serves: 400x400 and 800x800 per lane, 3840x2160 by packets (control), lone 800x800 frames
cd /root/repo
for i in 1 2; do
for lib in ${PREV:-build/librt355_prev.so} raytracing_amd/librt355.so; do
  for a in "--size 400 400" "--traversal per-lane" "--workload spheres4k"; do
    w="--workload spheres800"; case "$a" in *workload*) w="";; esac
    RT355_LIB=$lib timeout -k 10 120 python3 bench.py $w $a --steps 12 --warmup 4 --no-configs --no-cpu > gpurun_out/ab.json 2>gpurun_out/ab_err.log || exit 1
    python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{')][-1])
print('$lib spheres $a', d['config']['traversal'], round(d['ms_per_frame']*1e3,2), 'us/frame')"
  done
  RT355_LIB=$lib timeout -k 10 100 python3 scripts/single_frame_tree.py 2>&1 | grep "per lane" | sed "s|^|$lib lone frame |"
done
done
