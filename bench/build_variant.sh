#!/bin/bash
# Build the kernel library of another git revision next to the working tree's, for same-box A/B runs:
#   bench/build_variant.sh <git-ref> <name>      -> bench/ab_libs/libpdt_hip_<name>.so
# and select it per run with PDT_HIP_LIB=bench/ab_libs/libpdt_hip_<name>.so (bench/ab.sh takes env assignments).
# The library has a plain C ABI; a variant only makes sense while that ABI is the one ops/hip.py declares.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF="$1"; NAME="$2"
CSRC=pytorch_distributed_training_amd/ops/csrc
TMP=$(mktemp -d /tmp/pdt_variant.XXXX)
mkdir -p "$ROOT/bench/ab_libs" "$TMP/csrc" "$TMP/obj"
cd "$ROOT"
for f in $(git ls-tree --name-only "$REF" "$CSRC/"); do git show "$REF:$f" > "$TMP/csrc/$(basename "$f")"; done
objs=()
for src in "$TMP"/csrc/*.hip; do
  obj="$TMP/obj/$(basename "$src" .hip).o"
  hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -I "$TMP/csrc" -c "$src" -o "$obj" &
  objs+=("$obj")
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o "$ROOT/bench/ab_libs/libpdt_hip_$NAME.so" "${objs[@]}"
rm -rf "$TMP"
echo "$ROOT/bench/ab_libs/libpdt_hip_$NAME.so"
