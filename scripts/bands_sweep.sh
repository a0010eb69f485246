# XCD-aware workgroup order of the tree kernels: off | 8 bands of rows (RT_EXPERIMENT build: RT355_XCD_BANDS)
cd /root/repo
export RT355_LIB=build/librt355_exp.so
for wl in ${WLS:-spheres800 spheres4k sweep8k_spheres}; do
 for cfg in "0 0" "1 1"; do
  set -- $cfg
  RT355_XCD_BANDS=$1 RT355_BAND_COLS=$2 timeout -k 10 120 python3 bench.py --workload $wl --steps 12 --warmup 4 --no-configs --no-cpu > gpurun_out/bands_${wl}_$1_$2.json 2>gpurun_out/bands_err.log || exit 1
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/bands_${wl}_$1_$2.json') if l.startswith('{')][-1])
print('$wl bands=$1 cols=$2 (0: auto)', round(d['ms_per_frame']*1e3,1), 'us/frame')"
 done
done
