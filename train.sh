#!/usr/bin/env bash
# Example launch: one node, one process per visible MI355X, SyncBN ResNet-50 recipe.
# (flag-driven spawn mode; for torchrun use:
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
#       train_distributed.py --cfg-filepath config/ResNet50.yml --log-dir run/r50 --file-name-cfg ResNet50)
export HSA_ENABLE_IPC_MODE_LEGACY=0
python train_distributed.py \
    --num-nodes 1 \
    --rank 0 \
    --multiprocessing \
    --dist-url tcp://127.0.0.1:9001 \
    --log-dir run/distributed-with-syncbn \
    --file-name-cfg ResNet50 \
    --cfg-filepath config/ResNet50.yml \
    --seed 1029
