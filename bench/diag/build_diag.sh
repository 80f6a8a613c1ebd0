#!/bin/bash
# Ablation build of the kernel library (conv_dma.hip with -DPDT_DMA_DIAG): bench/ab_libs/libpdt_hip_diag.so
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
CSRC="$ROOT/pytorch_distributed_training_amd/ops/csrc"
TMP=$(mktemp -d /tmp/pdt_diag.XXXX)
mkdir -p "$ROOT/bench/ab_libs"
objs=()
for src in "$CSRC"/*.hip; do
  obj="$TMP/$(basename "$src" .hip).o"
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -DPDT_DMA_DIAG -I "$CSRC" -c "$src" -o "$obj" &
  objs+=("$obj")
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o "$ROOT/bench/ab_libs/libpdt_hip_diag.so" "${objs[@]}"
rm -rf "$TMP"
echo "$ROOT/bench/ab_libs/libpdt_hip_diag.so"
