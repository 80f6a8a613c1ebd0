#!/bin/bash
# Throughput of every model family on one MI355X:  bench/model_family.sh [OUT_FILE] [BATCH]
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/model_family.txt}"
BATCH="${2:-1024}"
cd "$ROOT"
: > "$OUT"
for m in ResNet18 ResNet34 ResNet50 ResNet101 ResNet152; do
  line=$(timeout -k 10 170 python bench.py --model "$m" --batch "$BATCH" --steps 20 --warmup 6 2>/dev/null | grep '^{' | tail -1)
  python - "$m" "$BATCH" "$line" <<'PY' | tee -a "$OUT"
import json, sys
m, b, line = sys.argv[1:4]
try:
    j = json.loads(line)
    print(f"{m:10s} batch {b:>4s}  {j['value']:9.0f} images/s  {j['ms_per_step']:8.2f} ms/step  peak {j['config']['peak_mem_gb']:.1f} GB  replay {j['config']['step_replay']}")
except Exception as e:
    print(f"{m:10s} batch {b:>4s}  failed ({e})")
PY
done
