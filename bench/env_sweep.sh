#!/bin/bash
# Same-box sweep of measurement switches on the flagship step: bench/env_sweep.sh OUT.txt "VAR=a VAR=b ..." (one variable setting per
# run, a plain run before, between and after).
OUT="${1:?out file}"; shift
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$ROOT"
: > "$OUT"
one() { ms=$(env $1 timeout -k 10 200 python bench.py --steps ${STEPS:-30} --warmup 10 ${BENCH_ARGS:-} 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'); echo "$1 $ms" | tee -a "$OUT"; }
one PDT_X=0
for v in "$@"; do one "$v"; done
one PDT_X=0
