set -o pipefail
D=gpurun_out/r3f; mkdir -p $D
timeout -k 10 700 python -u -m pytest tests/test_stream_budget_gpu.py tests/test_dist_gpu.py -m gpu -x -q -s --timeout 400 --timeout-method thread > $D/dist.log 2>&1; echo rc=$?
grep -v "^$" $D/dist.log | tail -40
