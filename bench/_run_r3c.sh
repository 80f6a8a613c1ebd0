set -o pipefail
D=gpurun_out/r3c; mkdir -p $D
timeout -k 10 600 python -u -m pytest tests/test_hip_ops.py tests/test_conv_regime_gpu.py tests/test_bn_fold.py tests/test_engine_gpu.py tests/test_deterministic_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $D/tests.log 2>&1 || { tail -40 $D/tests.log; exit 1; }
tail -3 $D/tests.log
PDT_HIP_LIB=bench/ab_libs/libpdt_hip_diag.so timeout -k 10 300 python bench/diag/onek_ablate.py --variants 0,27 > $D/onek_ablate.txt 2>&1 && cat $D/onek_ablate.txt && \
PDT_DMA_PREFETCH=0 PDT_HIP_LIB=bench/ab_libs/libpdt_hip_diag.so timeout -k 10 300 python bench/diag/onek_ablate.py --variants 0,27 > $D/onek_ablate_nopf.txt 2>&1 && cat $D/onek_ablate_nopf.txt && \
bench/ab.sh 3 "PDT_DMA_PREFETCH=0" "PDT_DMA_PREFETCH=1" > $D/ab.txt 2>&1
tail -4 $D/ab.txt
