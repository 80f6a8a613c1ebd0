"""Per-rank training worker: setup, iteration loop, logging, validation, checkpoints.

Mirrors the behaviour of upstream's ``Runner.worker / train_iter / validate``
(reference ``train_distributed.py:137-331``) with its documented quirks either kept or
fixed:

kept   iteration-based loop (``while iter < train_iters``), scheduler stepped after every
       iteration, validation when ``(iter != 0 and (iter+1) % val_interval == 0) or iter ==
       train_iters-1``, per-node ``batch_size`` / ``num_workers`` divided by the node's
       process count, one optimizer group (weight decay on every parameter), the train
       log line ``"Iter [i/N] Lr: [...] Loss: x"`` and the scalar tags ``loss/train``,
       ``lr_group/{gid}``, ``eval/Acc@1``, ``eval/Acc@5``, ``eval/loss``.
fixed  the device is selected before anything touches it; the sampler's epoch advances;
       the ``validation`` config section is honoured; validation sums exact
       correct/seen/loss counters on device and all-reduces ONCE at the end instead of
       three all-reduces per batch on padded shards; the progress bar runs on rank 0
       only; the log directory is never deleted on failure; periodic checkpoints and
       ``resume`` exist.
"""
from __future__ import annotations

import itertools
import math
import os
import time
from typing import Any, Callable, Dict, Optional

import torch
import torch.distributed as dist

from . import checkpoint as ckpt_io
from .config import validation_loader_settings
from .data import DevicePrefetcher, GpuAugment, build_loader, get_dataset, make_iter_dataloader
from .engine.engines import build_engine
from .logging_utils import worker_logger
from .optim.schedulers import get_scheduler
from .parallel import collectives as coll
from .parallel import dist_env
from .utils.determinism import make_deterministic
from .utils.metrics import TopKCounter
from .utils.profiling import ThroughputMeter, enable_ranges, peak_memory_gb, trace_range


def is_val_iter(it: int, val_interval: int, train_iters: int) -> bool:
    """Upstream's validation schedule (reference ``train_distributed.py:255-259``)."""
    p1 = it != 0
    p2 = (it + 1) % val_interval == 0
    p3 = it == train_iters - 1
    return (p1 and p2) or p3


def per_process_batch(cfg: Dict[str, Any], procs_per_node: int, distributed: bool) -> int:
    tr = cfg["training"]
    if tr.get("per_gpu_batch_size"):
        return int(tr["per_gpu_batch_size"])
    if distributed:
        return max(int(tr["batch_size"] / procs_per_node), 1)      # train_distributed.py:194
    return int(tr["batch_size"])


def per_process_workers(n_workers: int, procs_per_node: int, distributed: bool) -> int:
    if distributed:
        return int((n_workers + procs_per_node - 1) / procs_per_node)  # train_distributed.py:195
    return int(n_workers)


class Trainer:
    def __init__(self, cfg: Dict[str, Any], info: dist_env.DistInfo, logger_queue=None,
                 tb_writer_constructor: Optional[Callable[[], Any]] = None, seed: Optional[int] = None,
                 log_dir: Optional[str] = None, ops_backend: str = "auto") -> None:
        self.cfg = cfg
        self.info = info
        self.device = info.device
        self.rank = info.rank
        self.world = info.world_size
        self.seed = seed
        self.log_dir = log_dir
        self.iter = 0
        tr = cfg["training"]

        self.logger = worker_logger(logger_queue, f"worker_rank_{self.rank}")
        self.tb_writer = tb_writer_constructor() if (self.rank == 0 and tb_writer_constructor) else None
        self.logger.info("Use device: %s for training, current rank: %d / world %d", self.device, self.rank, self.world)
        enable_ranges(cfg["runtime"].get("profile", False))

        # ---- data -----------------------------------------------------------------------
        ds_cfg = cfg["dataset"]
        fmt = ds_cfg.get("input_format", "float_nchw")
        common = dict(n_classes=ds_cfg["n_classes"], image_size=ds_cfg.get("image_size", 224), fmt=fmt,
                      seed=seed or 0)
        extra = {"cpu_augment": ds_cfg.get("cpu_augment", False)} if ds_cfg["name"].lower() == "packed" else {}
        train_ds = get_dataset(ds_cfg["name"], ds_cfg["root"], "train", length=ds_cfg.get("train_size"), **common,
                               **extra)
        val_ds = get_dataset(ds_cfg["name"], ds_cfg["root"], "val", length=ds_cfg.get("val_size"), **common)
        # packed records are cropped / resized / flipped on the GPU, on the copy stream behind their upload
        self.gpu_augment = None
        if getattr(train_ds, "gpu_augment", False):
            self.gpu_augment = GpuAugment(ds_cfg.get("image_size", 224), seed or 0, info.rank)
            self._record_side = train_ds.side

        self.batch_size = per_process_batch(cfg, info.procs_per_node, info.distributed)
        n_workers = per_process_workers(tr["num_workers"], info.procs_per_node, info.distributed)
        vset = validation_loader_settings(cfg)
        self.val_batch_size = (max(int(vset["batch_size"] / info.procs_per_node), 1) if info.distributed
                               else int(vset["batch_size"]))
        if tr.get("per_gpu_batch_size"):
            self.val_batch_size = int(tr["per_gpu_batch_size"])
        val_workers = per_process_workers(vset["num_workers"], info.procs_per_node, info.distributed)
        self.logger.info("batch_size: %d, workers: %d", self.batch_size, n_workers)

        self.train_loader = build_loader(train_ds, self.batch_size, n_workers, True, self.world, self.rank,
                                         seed=seed or 0)
        self.val_loader = build_loader(val_ds, self.val_batch_size, val_workers, False, self.world, self.rank)
        self.logger.info("Load dataset done\nTraining: %d imgs, %d batchs\nEval: %d imgs, %d batchs",
                         len(train_ds), len(self.train_loader), len(val_ds), len(self.val_loader))

        # ---- model / optimizer / schedule ------------------------------------------------
        self.engine = build_engine(cfg, self.device, ops_backend=ops_backend, logger=self.logger)
        self.logger.info("Engine: %s", self.engine.describe())
        self.logger.info("Loaded optimizer:\n%s", self.engine.optimizer)
        self.scheduler = get_scheduler(self.engine.optimizer, tr["lr_schedule"], train_iters=tr["train_iters"])

        self.ckpt_dir = tr.get("checkpoint_dir") or (os.path.join(log_dir, "checkpoints") if log_dir else None)
        self._warned_stale = False
        self.ckpt_interval = tr.get("checkpoint_interval") or tr["val_interval"]
        resume = tr.get("resume")
        if resume == "auto":
            resume = ckpt_io.latest_checkpoint(self.ckpt_dir)
        if resume:
            self._resume(resume)
        self.meter = ThroughputMeter(self.device)
        self.last_val: Dict[str, float] = {}
        self.history = []          # (iter, loss) pairs logged on this rank (tests read it)
        self._replay_mode_logged = None

    # ------------------------------------------------------------------------------ state
    def _resume(self, path: str) -> None:
        ck = ckpt_io.load_checkpoint(path)
        self.engine.load_model_state_dict(ck["model"])
        if ck.get("optimizer"):
            self.engine.load_optimizer_state_dict(ck["optimizer"])
        if ck.get("scheduler"):
            self.scheduler.load_state_dict(ck["scheduler"])
        self.iter = int(ck.get("iter", 0))
        ckpt_io.restore_rng(ck, self.logger)
        self.logger.info("Resumed from %s at iteration %d", path, self.iter)

    def save_checkpoint(self, tag: Optional[str] = None) -> Optional[str]:
        if self.ckpt_dir is None:
            return None
        if not self.engine.check_health():  # collective: never persist a replica a failed exchange has poisoned
            raise RuntimeError(f"communication failure detected before the checkpoint of iteration {self.iter + 1}")
        self.engine.prepare_validation()   # collective on every rank (buffer sync without SyncBN)
        path = None
        if self.rank == 0:
            path = os.path.join(self.ckpt_dir, f"{tag or 'iter_%d' % (self.iter + 1)}.pth")
            ckpt_io.save_checkpoint(path, self.engine.model_state_dict(), self.engine.optimizer_state_dict(),
                                    self.scheduler.state_dict(), self.iter + 1, self.cfg)
            self.logger.info("Saved checkpoint %s", path)
            keep = int(self.cfg["training"].get("keep_last_checkpoints", 3) or 0)
            if tag is None and keep > 0:
                for old in ckpt_io.prune_checkpoints(self.ckpt_dir, keep, upto=self.iter + 1):
                    self.logger.info("Removed old checkpoint %s", old)
                stale = ckpt_io.stale_checkpoints(self.ckpt_dir, self.iter + 1)
                if stale and not self._warned_stale:
                    self._warned_stale = True
                    self.logger.warning("checkpoint directory %s holds %d checkpoint(s) from beyond iteration %d (another "
                                        "run's, e.g. %s): they are left alone, and 'resume: auto' would pick the newest of "
                                        "them", self.ckpt_dir, len(stale), self.iter + 1, os.path.basename(stale[-1]))
        if self.world > 1:
            # the other ranks wait here, on the host, rather than inside the next step's first
            # statistics exchange on the GPU
            dist.barrier()
        return path

    # ------------------------------------------------------------------------------ loop
    def run(self) -> None:
        tr = self.cfg["training"]
        train_iters = tr["train_iters"]
        per_epoch = max(len(self.train_loader), 1)
        if self.gpu_augment is not None and self.iter > self.gpu_augment.batches:
            # resumed run: continue the crop stream where the checkpointed run left it
            self.gpu_augment.skip(self.iter - self.gpu_augment.batches, self.batch_size, self._record_side)
        batches = DevicePrefetcher(
            make_iter_dataloader(self.train_loader, self.iter // per_epoch, self.iter % per_epoch), self.device,
            transform=self.gpu_augment)
        debug_every = self.cfg["runtime"].get("debug_sync_check", 0)
        fault_iter = int(os.environ.get("FAULT_INJECT_ITER", "-1"))
        fault_rank = int(os.environ.get("FAULT_INJECT_RANK", "0"))
        self.meter.start()
        while self.iter < train_iters:
            t0 = time.perf_counter()
            img, label = next(batches)
            wait = time.perf_counter() - t0
            if self.iter == fault_iter and self.rank == fault_rank:
                raise RuntimeError(f"injected fault at iteration {self.iter} on rank {self.rank}")
            self.train_iter(img, label, wait)
            if debug_every and (self.iter + 1) % debug_every == 0:
                ok = dist_env.check_replicas_in_sync(self.engine.flat_parameters())
                if not ok:
                    raise RuntimeError(f"replica parameters diverged at iteration {self.iter}")
            if is_val_iter(self.iter, tr["val_interval"], train_iters):
                self.validate()
                self.meter.start()
            if self.ckpt_dir and ((self.iter + 1) % self.ckpt_interval == 0 or self.iter == train_iters - 1):
                self.save_checkpoint()
            self.iter += 1
        if self.tb_writer is not None:
            self.tb_writer.flush()

    def train_iter(self, img: torch.Tensor, label: torch.Tensor, data_wait_s: float = 0.0) -> None:
        tr = self.cfg["training"]
        with trace_range("train_iter"):
            loss = self.engine.train_step(img, label)
        self.meter.add_step(int(label.shape[0]), data_wait_s)
        if self.iter % tr["print_interval"] == 0:
            loss_val = loss.detach().clone().float()
            if self.world > 1:
                coll.all_reduce(loss_val)
            loss_val = loss_val.item() / self.world
            lrs = self.scheduler.get_last_lr()
            mode = getattr(getattr(self.engine, "model", None), "active_replay_mode", None)
            if mode is not None and mode() != self._replay_mode_logged:
                self._replay_mode_logged = mode()
                self.logger.info("Step replay mode: %s", self._replay_mode_logged)
            self.history.append((self.iter, loss_val))
            if self.rank == 0:
                perf = self.meter.read()
                self.logger.info(
                    "Iter [{:d}/{:d}] Lr: {} Loss: {:.4f}  ({:.1f} img/s/rank, {:.1f} img/s total, "
                    "step {:.1f} ms, data-wait {:.2f} ms, mem {:.1f} GB)".format(
                        self.iter, tr["train_iters"], lrs, loss_val, perf["images_per_s"],
                        perf["images_per_s"] * self.world, perf["step_ms"], perf["data_wait_ms"],
                        peak_memory_gb(self.device)))
                if self.tb_writer is not None:
                    self.tb_writer.add_scalar("loss/train", loss_val, self.iter)
                    for gid, lr in enumerate(lrs):
                        self.tb_writer.add_scalar("lr_group/{}".format(gid), lr, self.iter)
                    self.tb_writer.add_scalar("perf/images_per_s", perf["images_per_s"] * self.world, self.iter)
                    self.tb_writer.add_scalar("perf/step_ms", perf["step_ms"], self.iter)
                    self.tb_writer.add_scalar("perf/data_wait_ms", perf["data_wait_ms"], self.iter)
            if not math.isfinite(loss_val):
                raise FloatingPointError(f"non-finite training loss at iteration {self.iter}")
        self.scheduler.step()

    def validate(self) -> Dict[str, float]:
        if self.rank == 0:
            self.logger.info("Start valuation")
        self.engine.prepare_validation()
        counter = TopKCounter(self.device)
        max_batches = self.cfg["training"].get("max_val_batches")
        loader = self.val_loader
        if self.rank == 0:
            try:
                import tqdm

                loader = tqdm.tqdm(self.val_loader, disable=None)
            except Exception:
                pass
        if max_batches is not None:
            loader = itertools.islice(loader, int(max_batches))
        with torch.no_grad(), trace_range("validate"):
            # the next batch's host-to-device copy runs on the copy stream under this batch's forward
            for img, label in DevicePrefetcher(loader, self.device):
                counter.add_tensor(self.engine.eval_step(img, label))
        counter.all_reduce()
        res = counter.result()
        self.last_val = res
        if self.rank == 0:
            self.logger.info("Acc@1: %.4f, Acc@5: %.4f, Loss: %.5f", res["acc1"], res["acc5"], res["loss"])
            if self.tb_writer is not None:
                self.tb_writer.add_scalar("eval/Acc@1", res["acc1"], self.iter)
                self.tb_writer.add_scalar("eval/Acc@5", res["acc5"], self.iter)
                self.tb_writer.add_scalar("eval/loss", res["loss"], self.iter)
        return res


def run_worker(cfg: Dict[str, Any], info: dist_env.DistInfo, logger_queue=None, tb_writer_constructor=None,
               seed: Optional[int] = None, log_dir: Optional[str] = None, ops_backend: str = "auto") -> Trainer:
    if seed is not None:
        make_deterministic(seed, strict=cfg["runtime"].get("deterministic", False))
    trainer = Trainer(cfg, info, logger_queue, tb_writer_constructor, seed, log_dir, ops_backend)
    if cfg["runtime"].get("eval_only", False):
        trainer.validate()            # checkpoint evaluation: weights come from training.resume
        if trainer.tb_writer is not None:
            trainer.tb_writer.flush()
        return trainer
    trainer.run()
    return trainer
