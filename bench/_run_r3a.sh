set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 600 python -u -m pytest tests/test_hip_ops.py tests/test_conv_regime_gpu.py tests/test_bn_fold.py tests/test_engine_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3a/tests.log 2>&1 || { tail -30 gpurun_out/r3a/tests.log; exit 1; }
tail -3 gpurun_out/r3a/tests.log
timeout -k 10 200 python bench/conv_layers.py --batch 1024 --miopen 0 --ops fprop,dgrad > gpurun_out/r3a/layers_new.txt 2>&1 && \
PDT_HIP_LIB=bench/ab_libs/libpdt_hip_base.so timeout -k 10 200 python bench/conv_layers.py --batch 1024 --miopen 0 --ops fprop,dgrad > gpurun_out/r3a/layers_base.txt 2>&1 && \
bench/ab.sh 3 "PDT_HIP_LIB=bench/ab_libs/libpdt_hip_base.so" "PDT_DUMMY=1" > gpurun_out/r3a/ab.txt 2>&1
tail -4 gpurun_out/r3a/ab.txt
