"""Command-line entry point: ``python -m pytorch_distributed_training_amd.cli ...``.

Accepts upstream's nine flags with the same names and defaults (reference
``train_distributed.py:38-49``; example invocation in ``train.sh:6-14``) plus four
additions: ``--set a.b=c`` config overrides, ``--nproc-per-node`` (process count when it
should not be derived from the GPU count, e.g. CPU/gloo runs), ``--ops-backend`` and
``--eval-only`` (one validation pass over the checkpoint in ``training.resume``).

Failure handling differs from upstream on purpose: upstream's handler removes the whole
log directory on any exception (``shutil.rmtree(os.path.join(log_dir), "tf-board-logs")``,
``train_distributed.py:82`` -- the second positional argument is ``ignore_errors``); here
the traceback is logged at CRITICAL, logs are kept and the process exits non-zero.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import traceback
from functools import partial

import torch

from .config import get_cfg
from .launch import START_METHOD, launch
from .logging_utils import MultiProcessLoggerListener, get_train_logger
from .parallel import dist_env
from .utils.determinism import make_deterministic


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MI355X-native distributed ImageNet training")
    p.add_argument("--num-nodes", default=-1, type=int, help="number of nodes for distributed training")
    p.add_argument("--rank", default=-1, type=int, help="node rank for distributed training")
    p.add_argument("--dist-url", default="tcp://127.0.0.1:9876", type=str)
    p.add_argument("--dist-backend", default="nccl", type=str, help="distributed backend (nccl = RCCL, gloo = CPU)")
    p.add_argument("--seed", default=None, type=int, help="seed for initializing training")
    p.add_argument("--multiprocessing", action="store_true", help="one process per GPU on this node")
    p.add_argument("--file-name-cfg", type=str, default="train")
    p.add_argument("--log-dir", type=str, default="run/default")
    p.add_argument("--cfg-filepath", type=str, required=True)
    # additions
    p.add_argument("--set", dest="overrides", action="append", default=[], metavar="a.b=value",
                   help="override a config key (repeatable)")
    p.add_argument("--nproc-per-node", type=int, default=None, help="processes per node (default: GPU count)")
    p.add_argument("--ops-backend", type=str, default="auto", choices=("auto", "hip", "ref"))
    p.add_argument("--eval-only", action="store_true",
                   help="evaluate the checkpoint named by training.resume on the validation split and exit")
    return p


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)

    if args.seed is not None:
        print("Set seed:", args.seed)
        make_deterministic(args.seed)

    rank_env = int(os.environ.get("RANK", "0")) if dist_env.env_launch_detected() else 0
    file_name = args.file_name_cfg if rank_env == 0 else f"{args.file_name_cfg}.rank{rank_env}"
    logger_listener = MultiProcessLoggerListener(
        partial(get_train_logger, logdir=args.log_dir, filename=file_name), START_METHOD)
    logger = logger_listener.get_logger("Runner")
    status = 0
    try:
        cfg = get_cfg(args.cfg_filepath, args.overrides + (["runtime.eval_only=true"] if args.eval_only else []))
        logger.info("Starting distributed runner")
        launch(cfg, args.num_nodes, args.rank, args.dist_url, args.dist_backend, args.multiprocessing, args.seed,
               args.log_dir, args.file_name_cfg, logger_listener.queue, args.nproc_per_node, args.ops_backend,
               logger)
    except Exception as e:  # fail fast, keep the logs
        tb = traceback.format_exc()
        logger.critical("While running, exception:\n%s\nTraceback:\n%s", str(e), str(tb))
        time.sleep(0.5)
        status = 1
    finally:
        logger_listener.stop()
    return status


if __name__ == "__main__":
    sys.exit(main())
