export TMPDIR=/tmp
# N=2 rehearsal on ONE GPU (gloo): frames mode and row-tile mode
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 64 --warmup 16 --backend gloo | tail -1 | cut -c1-700 || exit 1
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 4 --warmup 1 --backend gloo --workload sweep8k | tail -1 | cut -c1-700 || exit 1
# the default run, as the driver does it
time timeout -k 10 600 python bench.py > gpurun_out/bench_default.json && cat gpurun_out/bench_default.json
