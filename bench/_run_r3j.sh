set -o pipefail
D=gpurun_out/r3j; mkdir -p $D
PDT_FORCE_COLLECTIVES=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 3 --batch 64 > $D/bench_rccl.json 2> $D/bench_rccl.err; echo rccl rc=$?; grep -o '"rccl": .*' $D/bench_rccl.json | cut -c1-1500
timeout -k 10 600 python bench/loader_throughput.py --images 2048 --workers 16 --batch 256 --seconds 5 > $D/loader.txt 2>&1; echo loader rc=$?; cat $D/loader.txt
