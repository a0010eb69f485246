mkdir -p gpurun_out/prof1
export TMPDIR=/tmp
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -o room800 -- python3 bench.py --steps 512 --warmup 32 --no-cpu > gpurun_out/prof1/bench.json 2> gpurun_out/prof1/err.log; echo rc=$?
ls -R gpurun_out/prof1 | head -20
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1); cat $f
