for b in 384 512 640 768 1024; do
  line=$(timeout -k 10 170 python bench.py --batch $b --steps 12 --warmup 5 2>/dev/null | grep '^{' | tail -1)
  python -c "import json,sys; j=json.loads(sys.argv[1]); print(sys.argv[2], j['value'], j['ms_per_step'], j['config']['peak_mem_gb'])" "$line" $b
done
