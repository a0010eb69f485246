mkdir -p gpurun_out/prof2; export TMPDIR=/tmp
for i in 1 2 3; do timeout -k 10 60 python bench.py --steps 1024 --warmup 128 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('room800', round(d['value']), 'Mrays/s', round(d['ms_per_step']*1000,1), 'us  frac', round(d['roofline']['frac'],4))" || exit 1; done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -o room800 -- python3 bench.py --steps 512 --warmup 32 --no-cpu > gpurun_out/prof2/bench.json 2> gpurun_out/prof2/err.log; echo rc=$?
cat gpurun_out/prof2/room800_kernel_stats.csv
