mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_lbvh.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t6.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/t6.log
