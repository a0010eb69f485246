# us/frame against frames per launch and tiles per wave at 3840x2160 and 7680x4320 (the launch shape)
cd /root/repo
run() { timeout -k 10 150 python3 bench.py --workload $1 --steps 8 --warmup 3 --no-configs --no-cpu $2 > gpurun_out/fpl.json 2>gpurun_out/fpl_err.log || exit 1
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/fpl.json') if l.startswith('{')][-1])
print('$1 $2', round(d['ms_per_frame']*1e3,1), 'us/frame')"; }
run spheres4k ""
run spheres4k "--tiles-per-wave 2"
run spheres4k "--frames-per-launch 64"
run sweep8k_spheres ""
run sweep8k_spheres "--frames-per-launch 16"
run sweep8k_spheres "--frames-per-launch 32"
run sweep8k ""
run sweep8k "--frames-per-launch 16"
run mirrors4k ""
run mirrors4k "--frames-per-launch 64"
