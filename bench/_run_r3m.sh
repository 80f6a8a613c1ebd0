set -o pipefail
D=gpurun_out/r3m; mkdir -p $D
timeout -k 10 600 python -u -m pytest tests/test_hip_ops.py tests/test_conv_regime_gpu.py tests/test_deterministic_gpu.py tests/test_bn_fold.py -m gpu -x -q --timeout 300 --timeout-method thread > $D/tests.log 2>&1; rc=$?; echo rc=$rc; tail -15 $D/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --batch 1024 --steps 30 --warmup 10 > $D/bench_1024.json 2> $D/bench.err; echo rc=$?
timeout -k 10 300 python bench.py --batch 3072 --steps 8 --warmup 4 > $D/bench_3072.json 2> $D/bench2.err; echo rc=$?
grep -o '"ms_per_step": [0-9.]*' $D/bench_1024.json $D/bench_3072.json
