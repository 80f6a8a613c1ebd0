set -o pipefail
D=gpurun_out/r3k; mkdir -p $D
PDT_RCCL_LOG_DIR=$D/rccl PDT_FORCE_COLLECTIVES=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 3 --batch 64 > $D/bench_rccl.json 2> $D/bench_rccl.err; echo rccl rc=$?; grep -o '"rccl": .*' $D/bench_rccl.json | cut -c1-2500
bash bench/batch_sweep.sh > $D/batch_sweep.txt 2>&1; echo sweep rc=$?; cat $D/batch_sweep.txt
