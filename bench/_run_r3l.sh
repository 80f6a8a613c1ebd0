set -o pipefail
D=gpurun_out/r3l; mkdir -p $D
R50_BATCHES="4096" R101_BATCHES="3072" R152_BATCHES="2048 3072" bash bench/batch_sweep.sh > $D/batch_sweep2.txt 2>&1; echo sweep rc=$?; cat $D/batch_sweep2.txt
