mkdir -p gpurun_out
for w in spheres800 spheres4k mirrors4k stock800; do timeout -k 10 200 python bench.py --steps 32 --warmup 4 --no-cpu --workload $w | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4), d['config']['triangles'])" || exit 1; done
timeout -k 10 60 ./raytracing_amd/host/rt_headless --scene stock --size 64 64 | head -3
