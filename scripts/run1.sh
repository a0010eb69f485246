mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t2.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t2.log
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" && \
timeout -k 10 200 python bench.py --steps 256 --warmup 16 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?"; cat gpurun_out/bench1.json; tail -3 gpurun_out/bench1.err
for t in 1 2 4; do timeout -k 10 60 python bench.py --steps 256 --warmup 16 --no-cpu --tiles-per-wave $t | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tpw', $t, d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
timeout -k 10 60 ./raytracing_amd/host/rt_headless --frames 3 --ppm gpurun_out/stock800.ppm
