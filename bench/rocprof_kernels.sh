#!/bin/bash
# Per-kernel time of the flagship training step.
#   bench/rocprof_kernels.sh [OUT_DIR] [bench.py args...]
# Writes rocprofv3's kernel trace + statistics under OUT_DIR (default gpurun_out/prof_kernels);
# copy the *_kernel_stats.csv you want to keep into profiles/.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/prof_kernels}"
shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
# the program itself goes right after "--": the profiler's preloaded library has already
# initialised the GPU, so no wrapper that would replace the process may sit in between
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --steps 10 --warmup 5 "$@"
find "$OUT" -name '*kernel_stats.csv' -print
