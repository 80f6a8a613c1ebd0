set -o pipefail
D=gpurun_out/r3e; mkdir -p $D
timeout -k 10 900 python -u -m pytest tests/test_engine_oracle_gpu.py tests/test_xgmi_gpu.py -m gpu -q -s --timeout 300 --timeout-method thread > $D/oracle.log 2>&1; echo rc=$?
grep -v "^$" $D/oracle.log | tail -40
