#!/bin/bash
# Per-dispatch roofline of the flagship step: bench/roofline.sh OUT.txt [BATCH]   (see bench/roofline.py)
set -uo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/roofline.txt}"
BATCH="${2:-1024}"
cd "$ROOT"
TMP=$(mktemp -d /tmp/roofline.XXXX)
timeout -k 10 200 python bench/roofline.py dump "$TMP/oplist.json" --batch "$BATCH" | tail -1
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$TMP/prof" -- python3 bench.py --batch "$BATCH" --steps 6 --warmup 5 > "$TMP/prof.log" 2>&1 || { tail -5 "$TMP/prof.log"; exit 1; }
f=$(find "$TMP/prof" -name "*kernel_trace.csv" | head -1)
python bench/roofline.py join "$TMP/oplist.json" "$f" --steps 4 --top 40 --all > "$OUT"
head -32 "$OUT"
rm -rf "$TMP"
