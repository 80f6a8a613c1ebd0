#!/bin/bash
# Hardware-counter passes over a short program (default: the per-layer conv benchmark).
#   bench/rocprof_pmc.sh [OUT_DIR] [program args...]
# rocprofv3 does not split counters over passes, so every group below is one run of its own and
# stays within one block's slots (8 SQ, 4 TCC with FETCH_SIZE=3/WRITE_SIZE=2, 4 TCP, 2 TA/TD/GRBM).
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/prof_pmc}"
shift || true
if [ "$#" -eq 0 ]; then set -- bench/conv_layers.py --layers a2,b6,c6,d6,a3,c5 --ops fprop,dgrad,wgrad; fi
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
PASSES=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
  "FETCH_SIZE"
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for counters in "${PASSES[@]}"; do
  i=$((i + 1))
  # shellcheck disable=SC2086
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/pass$i" -- python3 "$@"
done
python3 bench/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
