#!/bin/bash
# What the distributed machinery costs when nothing has to travel: the flagship bench on ONE GPU, plain vs with every
# collective really issued through a one-rank RCCL group (PDT_FORCE_COLLECTIVES=1: SyncBN statistics exchange per layer
# and direction, every gradient bucket on the communication stream, per-bucket optimizer behind it), interleaved on one
# box.  Also sweeps the bucket size.    bench/forced_collectives.sh [OUT_FILE]
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/forced_collectives.txt}"
cd "$ROOT"
: > "$OUT"
run() {  # label, launcher(0/1), extra bench args..., env via ENVV
  label="$1"; launcher="$2"; shift 2
  if [ "$launcher" = 1 ]; then
    cmd=(python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 1 --steps 20 --warmup 6 "$@")
  else
    cmd=(python bench.py --gpus 1 --steps 20 --warmup 6 "$@")
  fi
  line=$(env $ENVV timeout -k 10 170 "${cmd[@]}" 2>/dev/null | grep '^{' | tail -1)
  python - "$label" "$line" <<'PY' | tee -a "$OUT"
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    j = json.loads(line); c = j["config"]
    print(f"{label:58s} {j['ms_per_step']:7.3f} ms/step  exposed comm {c['exposed_comm_ms_per_step']:.3f} ms  buckets/step {c['grad_buckets_reduced_per_step']}  syncbn {c['syncbn_comm']}  {c['grad_allreduce_mb_per_step']:.0f} MB/step")
except Exception as e:
    print(f"{label:58s} failed ({e})")
PY
}
for r in 1 2; do
  ENVV="PDT_X=0" run "plain (no process group)" 0
  ENVV="PDT_FORCE_COLLECTIVES=1 PDT_SYNCBN_COMM=xgmi" run "forced, SyncBN via in-kernel xGMI exchange" 1
  ENVV="PDT_FORCE_COLLECTIVES=1 PDT_SYNCBN_COMM=rccl" run "forced, SyncBN via RCCL all-reduce" 1
  ENVV="PDT_FORCE_COLLECTIVES=1 PDT_SYNCBN_COMM=xgmi PDT_OPT_IN_BACKWARD=0" run "forced, xGMI, optimizer after backward" 1
done
for cap in 8 25 64 128; do
  ENVV="PDT_FORCE_COLLECTIVES=1 PDT_SYNCBN_COMM=xgmi" run "forced, xGMI, bucket_cap_mb $cap" 1 --bucket-cap-mb $cap
done
