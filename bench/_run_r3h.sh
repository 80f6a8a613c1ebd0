set -o pipefail
D=gpurun_out/r3h; mkdir -p $D
timeout -k 10 300 python -u -m pytest tests/test_hip_ops.py tests/test_conv_regime_gpu.py tests/test_deterministic_gpu.py -m gpu -x -q -k "wgrad" --timeout 120 --timeout-method thread > $D/wgrad_tests.log 2>&1; rc=$?; echo rc=$rc; tail -5 $D/wgrad_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench/conv_layers.py --help > $D/help.txt 2>&1; head -30 $D/help.txt
