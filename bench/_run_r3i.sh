set -o pipefail
D=gpurun_out/r3i; mkdir -p $D
PDT_WGRAD_ROW3=0 timeout -k 10 200 python bench/conv_layers.py --batch 1024 --ops wgrad --layers a2,b6,c6,d6 --miopen 0 > $D/wgrad_old.txt 2>&1; echo rc=$?
timeout -k 10 200 python bench/conv_layers.py --batch 1024 --ops wgrad --layers a2,b6,c6,d6 --miopen 0 > $D/wgrad_new.txt 2>&1; echo rc=$?
cat $D/wgrad_old.txt $D/wgrad_new.txt
PDT_WGRAD_ROW3=0 timeout -k 10 300 python bench.py --steps 30 --warmup 10 > $D/bench_old.json 2> $D/bench_old.err; echo rc=$?
timeout -k 10 300 python bench.py --steps 30 --warmup 10 > $D/bench_new.json 2> $D/bench_new.err; echo rc=$?
PDT_WGRAD_ROW3=0 timeout -k 10 300 python bench.py --steps 30 --warmup 10 > $D/bench_old2.json 2> $D/bench_old2.err; echo rc=$?
timeout -k 10 300 python bench.py --steps 30 --warmup 10 > $D/bench_new2.json 2> $D/bench_new2.err; echo rc=$?
grep -o '"ms_per_step": [0-9.]*' $D/bench_old.json $D/bench_new.json $D/bench_old2.json $D/bench_new2.json
