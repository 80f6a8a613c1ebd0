# per-GPU batch sweep on one MI355X: model, batch, images/s, ms/step, peak GB of HBM in use by the process
run() {
  line=$(timeout -k 10 280 python bench.py --model $1 --batch $2 --steps $3 --warmup 4 2>/tmp/sweep_err.txt | grep '^{' | tail -1)
  if [ -z "$line" ]; then echo "$1 $2 FAILED: $(grep -v '^$' /tmp/sweep_err.txt | tail -3 | tr '\n' ' ' | cut -c1-400)"; return 0; fi
  python -c "import json,sys; j=json.loads(sys.argv[1]); print(sys.argv[2], sys.argv[3], j['value'], j['ms_per_step'], j['config']['peak_mem_gb'], j['config'].get('step_replay'))" "$line" $1 $2
}
for b in ${R50_BATCHES:-256 512 1024 1536 2048 3072 4096}; do run ResNet50 $b 10; done
for b in ${R101_BATCHES:-256 512 1024 2048 3072}; do run ResNet101 $b 8; done
for b in ${R152_BATCHES:-512 1024 2048}; do run ResNet152 $b 6; done
