# LBVH tests, then spheres800 / spheres4k with the hoisting of scene-spanning triangles on and off, packets and per lane, interleaved (round 2 study)
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_lbvh.py tests/test_gpu_packet.py -x -q --timeout 120 --timeout-method thread > gpurun_out/hoist_tests.log 2>&1 || { tail -30 gpurun_out/hoist_tests.log; exit 1; }
tail -3 gpurun_out/hoist_tests.log
for rep in 1 2; do
for w in spheres800 spheres4k; do
for pk in 0 1; do
for nh in 1 0; do
  if [ $nh = 1 ]; then export RT355_NO_HOIST=1; else unset RT355_NO_HOIST; fi
  RT355_PACKET=$pk timeout -k 10 120 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu --no-configs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w packet=$pk nohoist=$nh', round(d['ms_per_frame']*1000,1),'us', round(d['value']), d['unit'], 'frac', round(d['roofline']['frac'],4))"
done; done; done; done
