"""Headline benchmark: ResNet-50 bf16 training throughput (images/s, whole job) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

For N > 1 there is one rank per GPU.  Either a launcher starts them, e.g.
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N --steps K --warmup W`` (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* are read from the environment), or -- from a plain shell, with none of those set --
this script starts the N ranks itself: the parent makes no GPU call, runs N fresh
interpreters of this file with the rendezvous variables set (the way upstream's trainer
spawns its workers, reference ``train_distributed.py:125-135``, ``train.sh:6-14``), relays
rank 0's JSON line and exits non-zero if any rank failed.  Collectives go over RCCL.

A step is the full training iteration of the trainer: host-to-device copy of a synthetic
uint8 ImageNet-shaped batch (prefetched on a copy stream), fused normalise/layout kernel,
forward, loss, backward with SyncBatchNorm (N > 1) and bucketed gradient all-reduce
overlapped on a side stream, fused SGD step.  Weak scaling: the per-GPU batch is fixed.
Three set-up steps in which the engine records its step program (``--setup-steps``, reported in the JSON line),
W un-timed warm-up steps, then exactly K timed steps bracketed by barrier +
``torch.cuda.synchronize()``; the time is the MAX over ranks; rank 0 prints one JSON line.

``--cfg-filepath config/ResNet50.yml`` (or ``config/test-sync.yml``, ...) takes model, optimizer, SyncBN switch, bucket
size and the batch from a trainer config the way the trainer reads it (the YAML's ``batch_size`` is per NODE and is
divided by the number of GPUs, reference ``train_distributed.py:191-195``; ``per_gpu_batch_size`` wins when present);
the dataset is replaced by synthetic data of the same shape.  Command-line values given explicitly override the file.

With more than one rank (or forced one-rank collectives) the communication library's own start-up log is captured
(``NCCL_DEBUG=INFO``, subsystems INIT/GRAPH/ENV: no per-call output inside the timed region) and rank 0 reports what RCCL
chose -- ranks, channels, rings, transports -- under ``config.rccl``; ``--rccl-tuning-log 1`` adds the per-collective
algorithm / protocol choices (it logs every call, so it is off by default).  A run whose step recording failed
(``step_replay: eager(failed)``) still prints its line and then exits with status 4: an eager step costs several times the
host time and must not pass for the real thing.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytorch_distributed_training_amd.config import validate_cfg  # noqa: E402
from pytorch_distributed_training_amd.data.loader import DevicePrefetcher  # noqa: E402
from pytorch_distributed_training_amd.engine.engines import build_engine  # noqa: E402
from pytorch_distributed_training_amd.parallel import collectives, dist_env  # noqa: E402


def synthetic_batches(batch: int, image_size: int, n_classes: int, fmt: str, device: torch.device, mode: str,
                      pool: int = 4, seed: int = 0):
    """Endless generator of (img, label) batches with ImageNet's shapes.

    ``mode='h2d'``: batches live in pinned host memory and are copied every step (what a real
    loader delivers); ``mode='device'``: batches are resident on the GPU.
    """
    g = torch.Generator().manual_seed(seed)
    imgs, labels = [], []
    for _ in range(pool):
        if fmt == "uint8_nhwc":
            img = torch.randint(0, 256, (batch, image_size, image_size, 3), generator=g, dtype=torch.uint8)
        else:
            img = torch.randn(batch, 3, image_size, image_size, generator=g)
        lab = torch.randint(0, n_classes, (batch,), generator=g, dtype=torch.int64)
        if mode == "device":
            img, lab = img.to(device), lab.to(device)
        elif device.type == "cuda":
            img, lab = img.pin_memory(), lab.pin_memory()
        imgs.append(img)
        labels.append(lab)
    i = 0
    while True:
        yield imgs[i % pool], labels[i % pool]
        i += 1


def _free_port() -> int:
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(n_ranks: int, argv, timeout_s: float = 3000.0, script: str = None) -> int:
    """Start ``n_ranks`` fresh interpreters of this script, one per GPU (ranks share GPUs round
    robin when fewer are visible).  Called before this process has touched the GPU; it never
    does.  Rank 0's stdout is this process's stdout, the other ranks write to stderr."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL and the xGMI exchange need it
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(_free_port()))
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n_ranks)
    procs = []
    for rank in range(n_ranks):
        renv = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__), *argv], env=renv,
                                      stdout=None if rank == 0 else sys.stderr))
    deadline = time.monotonic() + timeout_s
    codes = [None] * n_ranks
    try:
        while any(c is None for c in codes):
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    codes[i] = pr.poll()
            failed = [c for c in codes if c not in (None, 0)]
            if failed or time.monotonic() > deadline:
                break
            time.sleep(0.05)
    finally:
        # fail fast: one dead rank leaves the others blocked in a collective
        for i, pr in enumerate(procs):
            if codes[i] is None:
                pr.terminate()
        for i, pr in enumerate(procs):
            if codes[i] is None:
                try:
                    codes[i] = pr.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    codes[i] = pr.wait()
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def rccl_log_setup(tuning: bool) -> str:
    """Point the communication library's debug log of THIS process at a file of its own (before the process group
    exists: the variables are read when the communicator is created).  Returns the file's path."""
    base = os.environ.get("PDT_RCCL_LOG_DIR") or tempfile.mkdtemp(prefix="pdt_rccl_")
    os.makedirs(base, exist_ok=True)
    path = os.path.join(base, f"rccl_rank{os.environ.get('RANK', '0')}.log")
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):       # (boxes ship with VERSION: header only)
        os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,ENV" + (",TUNING" if tuning else ""))
    os.environ["NCCL_DEBUG_FILE"] = path
    return path


def parse_rccl_log(text: str) -> dict:
    """What the library said it set up: enough to tell a ring over one xGMI link from a multi-channel mesh without
    opening the log.  Every field is best effort (the wording differs between releases); ``lines`` is the size of the
    log, ``excerpt`` its first matching lines verbatim."""
    out: dict = {"lines": text.count("\n")}
    m = re.search(r"(?:RCCL|NCCL) version\s*:?\s*([^\s:][^\s]*)", text)
    if m:
        out["version"] = m.group(1)
    m = re.search(r"nranks (\d+)", text)
    if m:
        out["nranks"] = int(m.group(1))
    m = re.search(r"(\d+) coll channels, (\d+) (?:collnet|nvls) channels.*?(\d+) p2p channels(?:, (\d+) p2p channels per peer)?", text)
    if m:
        out["coll_channels"], out["p2p_channels"] = int(m.group(1)), int(m.group(3))
        if m.group(4):
            out["p2p_channels_per_peer"] = int(m.group(4))
    rings = re.findall(r"(?m)Channel (\d+)/(\d+) :((?: +\d+)+) *$", text)
    if rings:
        out["channels"] = int(rings[0][1])
        out["rings"] = sorted({" ".join(r[2].split()) for r in rings})[:8]
    trees = re.findall(r"Trees ((?:\[\d+\] [-\d/>]+ ?)+)", text)
    if trees:
        out["trees"] = trees[0].strip()[:200]
    transports = sorted(set(re.findall(r"via ([A-Za-z0-9/_]+(?:/[A-Za-z0-9_]+)*)", text)))
    if transports:
        out["transports"] = transports[:8]
    env = sorted(set(re.findall(r"(NCCL_[A-Z0-9_]+|RCCL_[A-Z0-9_]+) set by environment to ([^\s.]+)", text)))
    if env:
        out["env"] = {k: v for k, v in env if k != "NCCL_DEBUG_FILE"}
    tuning = re.findall(r"(AllReduce|Broadcast|AllGather|ReduceScatter): (\d+) Bytes -> Algo (\d+) proto (\d+)", text)
    if tuning:
        algo = {"0": "tree", "1": "ring", "2": "collnet_direct", "3": "collnet_chain"}
        proto = {"0": "LL", "1": "LL128", "2": "simple"}
        seen = {}
        for coll, nbytes, a, pr in tuning:
            seen[(coll, int(nbytes))] = f"{algo.get(a, a)}/{proto.get(pr, pr)}"
        out["choices"] = [{"collective": c, "bytes": b, "algo_proto": v} for (c, b), v in sorted(seen.items())][:24]
    keep = [ln for ln in text.splitlines() if re.search(r"Channel \d+/|coll channels|Trees|via |version|nranks", ln)]
    out["excerpt"] = [ln[-160:] for ln in keep[:6]]
    return out


def cfg_from_file(path: str, args, explicit: set) -> dict:
    """Trainer YAML -> the benchmark's config: the file decides what the command line did not spell out."""
    from pytorch_distributed_training_amd.config import get_cfg
    cfg = get_cfg(path)
    tr = cfg["training"]
    gpus = max(args.gpus, 1)
    if "batch" not in explicit:
        args.batch = int(tr.get("per_gpu_batch_size") or max(int(tr["batch_size"]) // gpus, 1))
    if "model" not in explicit:
        args.model = cfg["model"]["name"]
    if "sync_bn" not in explicit:
        args.sync_bn = int(bool(tr.get("sync_bn", False)))
    if "bucket_cap_mb" not in explicit:
        args.bucket_cap_mb = float(tr.get("bucket_cap_mb", 25.0))
    if "engine" not in explicit:
        args.engine = tr.get("engine", "native")
    cfg["dataset"] = {"name": "synthetic", "n_classes": int(cfg["dataset"].get("n_classes", 1000)),
                      "image_size": args.image_size, "input_format": args.input_format}
    tr.update({"train_iters": args.steps + args.warmup, "batch_size": args.batch * gpus, "per_gpu_batch_size": args.batch,
               "sync_bn": bool(args.sync_bn), "precision": "bf16", "bucket_cap_mb": args.bucket_cap_mb,
               "engine": args.engine})
    for key in ("resume", "checkpoint_dir", "checkpoint_interval"):
        tr.pop(key, None)
    return validate_cfg(cfg)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", type=str, default="ResNet50")
    ap.add_argument("--batch", type=int, default=1536,
                    help="per-GPU batch size (sized for 288 GB of HBM: 1536 uses 60 GB and is where ResNet-50's "
                         "throughput saturates -- +1.6 %% over 1024, flat from there to 4096 / 159 GB, "
                         "profiles/batch_sweep_r3.txt; per image it needs a sixth of the SyncBN exchanges and "
                         "gradient all-reduces of batch 256)")
    ap.add_argument("--engine", type=str, default="native", choices=("native", "torch"))
    ap.add_argument("--input-format", type=str, default="uint8_nhwc", choices=("uint8_nhwc", "float_nchw"))
    ap.add_argument("--data-mode", type=str, default="h2d", choices=("h2d", "device"))
    ap.add_argument("--sync-bn", type=int, default=1, help="SyncBatchNorm when more than one GPU")
    ap.add_argument("--bucket-cap-mb", type=float, default=25.0)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--backend", type=str, default="nccl")
    ap.add_argument("--setup-steps", type=int, default=3,
                    help="untimed steps before the warm-up in which the engine records its step (see main)")
    ap.add_argument("--cfg-filepath", type=str, default="",
                    help="trainer YAML (config/ResNet50.yml, config/test-sync.yml, ...): model, optimizer, SyncBN, "
                         "bucket size and batch come from it unless given on the command line; data stays synthetic")
    ap.add_argument("--rccl-tuning-log", type=int, default=0,
                    help="1: also capture RCCL's per-collective algorithm / protocol choices (logs every call)")
    args = ap.parse_args()
    explicit = {a.dest for a in ap._actions if any(o in sys.argv[1:] or any(x.startswith(o + "=") for x in sys.argv[1:])
                                                    for o in a.option_strings)}

    hang_dump = float(os.environ.get("PDT_HANG_DUMP_S", "0") or 0)
    if hang_dump > 0:                 # debug aid: every thread's Python stack if the run is still going then
        import faulthandler
        faulthandler.dump_traceback_later(hang_dump, exit=False)

    if args.gpus > 1 and not dist_env.env_launch_detected():
        return self_launch(args.gpus, sys.argv[1:])          # nothing here has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rccl_log = None
    if args.backend == "nccl" and (world > 1 or os.environ.get("PDT_FORCE_COLLECTIVES") == "1") \
            and os.environ.get("PDT_RCCL_LOG", "1") != "0":
        rccl_log = rccl_log_setup(bool(args.rccl_tuning_log))
    if world > 1 or args.gpus > 1 or dist_env.env_launch_detected():
        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
            return 2
        info = dist_env.init_from_env(args.backend, device_type="cuda")
    else:
        info = dist_env.init_distributed(args.backend, None, 1, 0, 0, 1, device_type="cuda")
    device = info.device
    if device.type != "cuda":
        print("bench.py: no GPU visible", file=sys.stderr)
        return 2

    cfg = cfg_from_file(args.cfg_filepath, args, explicit) if args.cfg_filepath else validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 1000, "image_size": args.image_size,
                    "input_format": args.input_format},
        "training": {
            "optimizer": {"name": "SGD", "lr": 0.1, "weight_decay": 1.0e-4, "momentum": 0.9},
            "lr_schedule": {"name": "multi_step", "milestones": [150000, 300000], "gamma": 0.1},
            "train_iters": args.steps + args.warmup, "batch_size": args.batch * max(args.gpus, 1),
            "per_gpu_batch_size": args.batch, "sync_bn": bool(args.sync_bn), "precision": "bf16",
            "bucket_cap_mb": args.bucket_cap_mb, "engine": args.engine,
        },
        "model": {"name": args.model},
    })
    torch.manual_seed(1029)
    engine = build_engine(cfg, device)
    n_classes = int(cfg["dataset"]["n_classes"])
    source = synthetic_batches(args.batch, args.image_size, n_classes, args.input_format, device, args.data_mode,
                               seed=info.rank)
    batches = DevicePrefetcher(source, device) if args.data_mode == "h2d" else source

    def step():
        img, label = next(batches)
        return engine.train_step(img, label)

    loss = None
    # Set-up, before the warm-up proper: the engine runs the first step of a batch shape eagerly (it fills its lazily
    # built caches), records the second as a command list and replays that from the third on -- what a tracing compiler
    # would do at build time.  Untimed, counted separately from the W warm-up steps and reported as "setup_steps".
    for _ in range(args.setup_steps):
        loss = step()
    for _ in range(args.warmup):
        loss = step()
    if not engine.check_health():                 # collective; a failed exchange would have poisoned the warm-up
        print("bench.py: communication failure during warm-up", file=sys.stderr)
        return 3
    native = args.engine == "native"
    if native:
        engine.model.reducer.start_timing()
    if info.distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    enqueue = time.perf_counter() - t0          # host time to submit the steps (GPU still running)
    if info.distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if info.distributed:
        collectives.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    exposed = engine.model.reducer.exposed_ms() if native else None
    healthy = engine.check_health()
    final_loss = float(loss) if loss is not None else float("nan")
    global_batch = args.batch * info.world_size
    value = global_batch * args.steps / elapsed
    replay_mode = engine.model.active_replay_mode() if native else None
    rccl = None
    if rccl_log and info.rank == 0:
        try:
            # the library writes its log through a buffered C stream of this process: push it out before reading
            import ctypes
            ctypes.CDLL(None).fflush(None)
            texts = [open(f, errors="replace").read() for f in sorted(glob.glob(rccl_log + "*"))]
            rccl = parse_rccl_log("\n".join(texts)) if texts else {"lines": 0}
            rccl["log"] = rccl_log
        except OSError as exc:
            rccl = {"error": str(exc)}
    opt = cfg["training"]["optimizer"]
    if info.rank == 0:
        print(json.dumps({
            "metric": f"train images/sec (whole node) {args.model} bf16",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": info.world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": f"synthetic ImageNet-shaped ({args.input_format}, {args.data_mode}), random-init weights",
            "config": {
                "model": args.model, "global_batch": global_batch, "per_gpu_batch": args.batch,
                "setup_steps": args.setup_steps,
                "image_size": args.image_size, "seq_len": None,
                "parallelism": f"dp{info.world_size}" + ("+syncbn" if (args.sync_bn and info.world_size > 1) else ""),
                "engine": args.engine,
                "optimizer": f"{opt['name']} momentum {opt.get('momentum', 0)} wd {opt.get('weight_decay', 0):g}",
                "cfg_file": args.cfg_filepath or None, "final_loss": round(final_loss, 4),
                "peak_mem_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                "rccl_world": dist.get_world_size() if dist.is_initialized() else 1,
                "backend": dist.get_backend() if dist.is_initialized() else None,
                "grad_buckets": len(engine.model.buckets) if native else None,
                "grad_buckets_reduced_per_step": (
                    engine.model.reducer.stats["launched"] // max(engine.model.reducer.stats["steps"], 1)
                    if native else None),
                "grad_allreduce_mb_per_step": (
                    round(engine.model.reducer.stats["bytes"] / max(engine.model.reducer.stats["steps"], 1) / 2 ** 20, 2)
                    if native else None),
                # time per step the compute stream stood still at the end of backward waiting for the
                # communication stream (last all-reduce + the per-bucket optimizer updates not yet done)
                "exposed_comm_ms_per_step": round(exposed, 4) if exposed is not None else None,
                "optimizer_in_backward": engine.model.optimizer_in_backward if native else None,
                "comm_healthy": bool(healthy),
                # payload reads the in-kernel exchange had to repeat on this rank (checksum not matching yet)
                "xgmi_payload_rereads": (engine.model._xgmi.rereads() if native and engine.model._xgmi is not None else None),
                "host_submit_ms_per_step": round(1e3 * enqueue / args.steps, 3),
                "syncbn_comm": engine.model.syncbn_comm if (args.engine == "native" and engine.model.sync_bn) else None,
                "step_replay": replay_mode,
                "sync_bn_active": bool(engine.model.sync_bn) if args.engine == "native" else None,
                "rccl": rccl,
            },
        }), flush=True)
    if dist.is_initialized():
        dist_env.shutdown()
    if not healthy:
        return 3
    if replay_mode == "eager(failed)":
        print("bench.py: the step recording failed and the timed steps ran eagerly (see the warning above): the number is "
              "not the engine's", file=sys.stderr)
        return 4
    return 0


if __name__ == "__main__":
    sys.exit(main())
