# A/B of two builds of the library over bench workloads:  LIBS="raytracing_amd/librt355.so build/librt355_x.so" WLS="..." bash scripts/ab_libs.sh
cd /root/repo
for wl in ${WLS:-spheres800 spheres4k sweep8k_spheres room800 sweep8k}; do
 for lib in ${LIBS}; do
  RT355_LIB=$lib timeout -k 10 120 python3 bench.py --workload $wl --steps 12 --warmup 4 --no-configs --no-cpu > gpurun_out/ab.json 2>gpurun_out/ab_err.log || exit 1
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{')][-1])
print('$wl $lib', round(d['ms_per_frame']*1e3,2), 'us/frame')"
 done
done
