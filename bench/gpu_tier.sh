# The GPU tier as one call: every GPU test, then the flagship bench, a BASELINE configuration by file, and the forced-collectives run
# (gpurun --timeout 1190 -- bash bench/gpu_tier.sh).  Each step runs under its own time limit.
set -o pipefail
D=gpurun_out/gpu_tier; mkdir -p $D
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $D/gpu_tests.log 2>&1; rc=$?; echo rc=$rc
tail -5 $D/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 10 > $D/bench.json 2> $D/bench.err; echo bench rc=$?; cat $D/bench.json
timeout -k 10 300 python bench.py --cfg-filepath config/test-sync.yml --steps 5 --warmup 3 > $D/bench_cfg.json 2> $D/bench_cfg.err; echo cfg rc=$?; cat $D/bench_cfg.json
PDT_FORCE_COLLECTIVES=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 3 --batch 64 > $D/bench_rccl.json 2> $D/bench_rccl.err; echo rccl rc=$?; cat $D/bench_rccl.json
