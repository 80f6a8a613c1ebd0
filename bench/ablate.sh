#!/bin/bash
# What each co-running part of the step costs inside the step: the flagship bench with parts switched off
# (results of the ablated runs are wrong on purpose; only their times mean anything).
#   bench/ablate.sh [OUT_FILE] [bench.py args...]
set -uo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/ablation.txt}"
shift || true
cd "$ROOT"
run() {  # label, env assignments...
    label="$1"; shift
    line=$(env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 6 "${EXTRA[@]}" 2>/dev/null | tail -1)
    ms=$(python -c "import json,sys; print(json.loads(sys.argv[1])['ms_per_step'])" "$line" 2>/dev/null || echo failed)
    printf "%-58s %s ms/step\n" "$label" "$ms" | tee -a "$OUT"
}
EXTRA=("$@")
: > "$OUT"
run "full step" PDT_ABLATE=
run "weight gradients dropped (critical stream alone)" PDT_ABLATE=wgrad
run "weight gradients on the critical stream (serial)" PDT_WGRAD_STREAM=0
run "optimizer after backward (whole model, one kernel)" PDT_OPT_IN_BACKWARD=0
run "every other weight gradient dropped" PDT_ABLATE=wgrad_half
run "BatchNorm-backward elementwise passes dropped" PDT_ABLATE=elemt
run "BatchNorm-forward elementwise passes (no residual) dropped" PDT_ABLATE=apply
run "full step again" PDT_ABLATE=
