#!/bin/bash
# Interleaved A/B of the flagship bench on ONE box (boxes differ by several per cent, runs on one box by ~0.3 %):
#   bench/ab.sh ROUNDS "ENV_A" "ENV_B" ["ENV_C" ...]      e.g.  bench/ab.sh 3 "PDT_CONV_DMA=1" "PDT_CONV_DMA=0"
# prints ms/step of every run and the per-variant median.
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"
rounds=$1; shift
declare -A all
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    ms=$(env $v timeout -k 10 150 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])" 2>/dev/null || echo nan)
    echo "round $r  [$v]  $ms"
    all["$v"]+="$ms "
  done
done
for v in "$@"; do
  python - "$v" ${all["$v"]} <<'PY'
import sys, statistics
v, xs = sys.argv[1], [float(x) for x in sys.argv[2:] if x != "nan"]
print(f"median [{v}]: {statistics.median(xs):.3f} ms/step  (min {min(xs):.3f}, n={len(xs)})")
PY
done
