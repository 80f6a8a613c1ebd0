#!/bin/bash
# Regenerate the per-round evidence from the tree as it is:  bench/regen_profiles.sh SUFFIX [OUT_DIR]
# One script, one suffix: bench JSONs (batch 256 / 512 / 1024 / 1536), model families, roofline, rocprofv3 kernel
# statistics, kernel resources from the compiler, hardware counters over the step, forced one-rank collectives.
# Every GPU step runs under its own time limit; the script stops at the first failure.
set -uo pipefail
SUF="${1:?suffix, e.g. r3}"
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${2:-$ROOT/gpurun_out/regen_$SUF}"
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
step() { echo "== $1"; }
# REGEN_PARTS: which parts to run (default: all); a GPU call is limited to 20 minutes, two calls cover everything
PARTS="${REGEN_PARTS:-bench family roofline stats resources pmc forced}"
want() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }

if want bench; then
step "bench JSONs"
for b in 256 512 1024 1536; do
  timeout -k 10 280 python bench.py --batch $b --steps 40 --warmup 10 > "$OUT/bench_${SUF}_native_b$b.json" 2> "$OUT/bench_b$b.err" || exit 1
  grep -o '"value": [0-9.]*, "unit": "images/s".\{0,80\}"ms_per_step": [0-9.]*' "$OUT/bench_${SUF}_native_b$b.json"
done
timeout -k 10 280 python bench.py --steps 300 --warmup 10 > "$OUT/bench_${SUF}_native_default_300steps.json" 2> "$OUT/bench_300.err" || exit 1

fi
if want family; then
step "model families"
bash bench/model_family.sh "$OUT/model_family_throughput_$SUF.txt" 1024 || exit 1

fi
if want roofline; then
step "roofline (batch 1024)"
bash bench/roofline.sh "$OUT/roofline_$SUF.txt" 1024 > "$OUT/roofline_head.txt" 2>&1 || { tail -5 "$OUT/roofline_head.txt"; exit 1; }
head -34 "$OUT/roofline_$SUF.txt"

fi
if want stats; then
step "kernel statistics (rocprofv3 --kernel-trace --stats, batch 1024)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kernels" -- python3 bench.py --batch 1024 --steps 10 --warmup 5 > "$OUT/prof_kernels.log" 2>&1 || { tail -5 "$OUT/prof_kernels.log"; exit 1; }
f=$(find "$OUT/prof_kernels" -name '*kernel_stats.csv' | head -1)
cp "$f" "$OUT/kernel_stats_${SUF}_resnet50_b1024.csv"
rm -rf "$OUT/prof_kernels"
head -12 "$OUT/kernel_stats_${SUF}_resnet50_b1024.csv"

fi
if want resources; then
step "kernel resources (compiler remarks of this build)"
python - "$OUT/kernel_resources_$SUF.txt" <<'PY' || exit 1
import json, sys
from pytorch_distributed_training_amd.ops import build
per_file = json.load(open(build.RESOURCES))
res = {k: v for kernels in per_file.values() for k, v in kernels.items()}
names = sorted(res)
pretty = build.demangle(names)
with open(sys.argv[1], "w") as f:
    f.write("kernel resources of the shipped library (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n")
    f.write(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'LDS':>7} {'scratch':>7} {'occ':>4}  kernel\n")
    for n, p in sorted(zip(names, pretty), key=lambda t: t[1]):
        r = res[n]
        f.write(f"{r.get('vgprs', 0):5d} {r.get('agprs', 0):5d} {r.get('sgprs', 0):5d} {r.get('lds', 0):7d} {r.get('scratch', 0):7d} "
                f"{r.get('occupancy', 0):4d}  {p[:150]}\n")
print(open(sys.argv[1]).read()[:1500])
PY

fi
if want pmc; then
step "hardware counters over the step (three passes, batch 1024)"
i=0
for counters in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  # shellcheck disable=SC2086
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$OUT/pmc/pass$i" -- python3 bench.py --batch 1024 --steps 3 --warmup 3 > "$OUT/pmc_pass$i.log" 2>&1 || { tail -5 "$OUT/pmc_pass$i.log"; exit 1; }
done
python3 bench/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_step_${SUF}_resnet50_b1024.txt" 2>&1 || exit 1
rm -rf "$OUT/pmc"
head -40 "$OUT/pmc_step_${SUF}_resnet50_b1024.txt"

fi
if want forced; then
step "forced one-rank collectives"
bash bench/forced_collectives.sh "$OUT/forced_collectives_$SUF.txt" || exit 1
fi
echo "regenerated under $OUT"
