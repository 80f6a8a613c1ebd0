"""Training engines behind one interface.

``NativeEngine``  the MI355X path: :class:`.native_resnet.NativeResNet` (explicit program
                  over HIP kernels, flat buffers, bucketed RCCL reducer, fused SGD).
``TorchEngine``   the stock-PyTorch path the upstream trainer uses -- eager module,
                  ``SyncBatchNorm.convert_sync_batchnorm``, ``DistributedDataParallel``,
                  ``torch.optim.SGD`` (reference ``train_distributed.py:196-207``) -- plus
                  channels-last + bf16 autocast on GPU.  It exists as the A/B baseline for
                  benchmarks and as an independent cross-check in tests.

Interface: ``train_step(img, label) -> loss``, ``eval_step(img, label) -> [loss_sum, c1,
c5, n]``, ``optimizer`` (exposes ``param_groups``), ``model_state_dict()`` /
``load_model_state_dict()`` in the torchvision schema, ``optimizer_state_dict()`` /
``load_optimizer_state_dict()``, ``prepare_validation()``.
"""
from __future__ import annotations

import copy
from typing import Any, Dict

import torch
import torch.distributed as dist

from ..models.resnet import get_model, get_spec
from ..ops import get_ops, ref as ref_ops
from ..optim.sgd import FlatSGD, get_optimizer
from ..utils.profiling import trace_range
from .native_resnet import NativeResNet


def _optimizer_kwargs(cfg: Dict[str, Any]) -> Dict[str, Any]:
    kw = copy.deepcopy(cfg["training"]["optimizer"])
    kw.pop("name")
    return kw


class NativeEngine:
    name = "native"

    def __init__(self, cfg: Dict[str, Any], device: torch.device, ops_backend: str = "auto", group=None,
                 logger=None) -> None:
        tr = cfg["training"]
        rt = cfg.get("runtime", {})
        self.device = torch.device(device)
        self.ops = get_ops(self.device, ops_backend)
        # runtime.deterministic: bit-identical GPU runs (ordered reductions instead of float atomics);
        # set before anything sizes an accumulator
        self.deterministic = bool(rt.get("deterministic", False))
        if hasattr(self.ops, "set_deterministic"):
            self.ops.set_deterministic(self.deterministic)
        precision = tr.get("precision", "bf16")
        dtype = torch.bfloat16 if (precision == "bf16" and self.device.type == "cuda") else torch.float32
        spec = get_spec(cfg["model"]["name"], cfg["dataset"]["n_classes"])
        self.model = NativeResNet(
            spec, self.device, self.ops, dtype, sync_bn=bool(tr.get("sync_bn", False)), group=group,
            bucket_cap_mb=tr.get("bucket_cap_mb", 25.0), first_bucket_mb=tr.get("first_bucket_mb", 1.0),
            overlap_comm=tr.get("overlap_comm", True),
            hip_graphs=cfg.get("runtime", {}).get("hip_graphs", True),
            syncbn_overlap=tr.get("syncbn_overlap", False),
            syncbn_comm=tr.get("syncbn_comm", "auto"),
            step_replay=rt.get("step_replay", "list"),
            grad_allreduce_dtype=tr.get("grad_allreduce_dtype", "fp32"),
            # A wait inside the exchange is a kernel spinning on the GPU: it gets a bound of its own, far shorter than a
            # library collective's (a dead peer must not hold every surviving GPU for half an hour before the status word
            # flips).  The places where a rank is legitimately late by more than seconds -- rank 0 writing a checkpoint,
            # the start of validation -- are fenced by host-side barriers in the trainer, not waited out in here.
            xgmi_timeout_s=float(rt.get("xgmi_timeout_s") or min(60.0, float(rt.get("collective_timeout_s", 1800)))),
            logger=logger,
        )
        # Same-seed initialisation on every rank (upstream seeds every worker identically,
        # train_distributed.py:141-142); the eager module is only the initialiser here.
        init = get_model(cfg["model"]["name"], cfg["dataset"]["n_classes"])
        self.model.load_state_dict(init.state_dict())
        del init
        self.model.broadcast_state(0)
        opt_cls = get_optimizer(tr["optimizer"])
        self.optimizer: FlatSGD = opt_cls(self.model.params, self.ops, no_wd_bn=tr.get("no_wd_bn", False),
                                          **_optimizer_kwargs(cfg))
        self.world = self.model.world
        import os
        env = os.environ.get("PDT_OPT_IN_BACKWARD")          # measurement override of the config key
        want = tr.get("optimizer_in_backward", True) if env is None else env == "1"
        # The per-bucket update runs while backward is still going, which is only sound when backward never reads what
        # the update writes.  The HIP data-gradient kernels read the TRANSPOSED weight copies (refreshed after the
        # step); the PyTorch reference backend reads the forward weights -- the very slices the update rewrites -- so
        # with it on a GPU (where the hook runs inside backward, on the communication stream or in line) the update
        # waits for the end of backward.  On the CPU the reducer runs the hooks from finish(), after backward.
        safe = not getattr(self.ops, "dgrad_reads_forward_weights", True) or self.device.type != "cuda"
        if want and not safe and logger is not None:
            logger.info("optimizer_in_backward: the %s backend's data gradients read the forward weights; the update "
                        "runs after backward", getattr(self.ops, "name", "?"))
        if want and safe:
            self.model.attach_optimizer(self.optimizer)

    def train_step(self, img: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        loss = self.model.forward_backward(img, label)
        # a timed-out statistics exchange has poisoned this replica with NaN: end the job now
        # (a host-memory read, no synchronisation)
        self.model.raise_on_comm_failure()
        with trace_range("optimizer"):
            if self.model.optimizer_in_backward:
                self.optimizer.finish_step()      # the update itself ran bucket by bucket inside backward
            else:
                self.optimizer.step(grad_scale=self.model.reducer.grad_scale)
            self.model.refresh_derived_weights()
        return loss

    def check_health(self) -> bool:
        """Collective + synchronising: no rank has seen a communication failure."""
        return self.model.check_comm_health()

    def eval_step(self, img: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        return self.model.eval_step(img, label)

    def prepare_validation(self) -> None:
        if not self.model.sync_bn:
            self.model.sync_buffers(0)

    def model_state_dict(self) -> Dict[str, torch.Tensor]:
        return self.model.state_dict()

    def load_model_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        self.model.load_state_dict(sd)

    def optimizer_state_dict(self) -> Dict[str, Any]:
        return self.optimizer.state_dict()

    def load_optimizer_state_dict(self, sd: Dict[str, Any]) -> None:
        self.optimizer.load_state_dict(sd)

    def flat_parameters(self) -> torch.Tensor:
        return self.model.params.p

    def describe(self) -> str:
        m = self.model
        return (f"NativeEngine(ops={self.ops.name}, dtype={m.dtype}, params={m.params.num_parameters()}, "
                f"buckets={len(m.buckets)}, sync_bn={m.sync_bn}, syncbn_comm={m.syncbn_comm if m.sync_bn else None}, "
                f"world={m.world}, step_replay={m.step_replay}, optimizer_in_backward={m.optimizer_in_backward}, "
                f"deterministic={self.deterministic})")


class TorchEngine:
    name = "torch"

    def __init__(self, cfg: Dict[str, Any], device: torch.device, group=None, **_) -> None:  # noqa: D401
        tr = cfg["training"]
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda:
            # what upstream switches on right after seeding (train_distributed.py:54): let the
            # convolution library autotune its algorithm per shape (MIOpen find mode on ROCm)
            torch.backends.cudnn.benchmark = True
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        model = get_model(cfg["model"]["name"], cfg["dataset"]["n_classes"]).to(self.device)
        if self.cuda:
            model = model.to(memory_format=torch.channels_last)
        self.bf16 = self.cuda and tr.get("precision", "bf16") == "bf16"
        if self.world > 1:
            if tr.get("sync_bn", False) and self.cuda:
                model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
            self.ddp = torch.nn.parallel.DistributedDataParallel(
                model, device_ids=[self.device.index] if self.cuda else None,
                bucket_cap_mb=tr.get("bucket_cap_mb", 25.0))
            self.module = self.ddp.module
            self.net = self.ddp
        else:
            self.module = model
            self.net = model
        kw = _optimizer_kwargs(cfg)
        self.optimizer = torch.optim.SGD(self.net.parameters(), **kw)
        self.loss_fn = torch.nn.CrossEntropyLoss()
        self.mean = torch.tensor((0.485, 0.456, 0.406), device=self.device).view(1, 3, 1, 1)
        self.std = torch.tensor((0.229, 0.224, 0.225), device=self.device).view(1, 3, 1, 1)

    def _prep(self, img: torch.Tensor) -> torch.Tensor:
        if img.dtype == torch.uint8:      # uint8 NHWC -> normalised float NCHW
            img = img.permute(0, 3, 1, 2).float().div_(255.0)
            img = (img - self.mean) / self.std
        if self.cuda:
            img = img.contiguous(memory_format=torch.channels_last)
        return img

    def train_step(self, img: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        self.net.train()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
            pred = self.net(self._prep(img))
            loss = self.loss_fn(pred.float(), label)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    @torch.no_grad()
    def eval_step(self, img: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        self.net.eval()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
            pred = self.module(self._prep(img))
        return ref_ops.ce_metrics(pred, label)

    def prepare_validation(self) -> None:
        pass

    def check_health(self) -> bool:
        return True

    def model_state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone() for k, v in self.module.state_dict().items()}

    def load_model_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        self.module.load_state_dict(sd)

    def optimizer_state_dict(self) -> Dict[str, Any]:
        return self.optimizer.state_dict()

    def load_optimizer_state_dict(self, sd: Dict[str, Any]) -> None:
        self.optimizer.load_state_dict(sd)

    def flat_parameters(self) -> torch.Tensor:
        return torch.cat([p.detach().reshape(-1).float() for p in self.module.parameters()])

    def describe(self) -> str:
        return f"TorchEngine(bf16={self.bf16}, world={self.world})"


def build_engine(cfg: Dict[str, Any], device: torch.device, group=None, ops_backend: str = "auto", logger=None):
    kind = cfg["training"].get("engine", "native")
    if kind == "native":
        return NativeEngine(cfg, device, ops_backend=ops_backend, group=group, logger=logger)
    if kind == "torch":
        return TorchEngine(cfg, device, group=group)
    raise ValueError(f"unknown engine '{kind}'")
