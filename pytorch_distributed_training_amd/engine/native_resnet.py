"""Native ResNet training engine: an explicit forward / backward program over fused ops.

What upstream does implicitly through ``DistributedDataParallel(model)(img)``,
``loss.backward()`` and ``optimizer.step()`` (reference ``train_distributed.py:196-198,
267-278``) -- eager NCHW fp32 modules, autograd, ``SyncBatchNorm`` with two blocking
collectives per layer, per-parameter gradient tensors copied in and out of DDP buckets --
is written out here as a static program for one GPU per process:

* activations are NHWC in the compute dtype (bf16 on MI355X), parameters live in flat
  fp32 buffers (:mod:`.params`) with a bf16 mirror, BatchNorm statistics stay fp32;
* forward of a conv+BN unit is ``conv (per-channel moments come out of the conv
  epilogue) -> [SyncBN: ONE all-reduce of 2C floats] -> finalize -> fused
  scale/shift(+residual)(+ReLU)``;
* backward is hand-derived: ``fused ReLU-mask + BN reduce -> [SyncBN: ONE all-reduce of
  2C floats] -> fused BN elementwise -> weight-gradient (accumulates into the flat
  gradient buffer) -> data-gradient``.  The data-gradient epilogue of a block's first conv
  also adds the other residual path and applies the ReLU backward of the block INPUT, so
  the gradient that reaches the previous block is already masked and its (largest)
  BatchNorm backward passes read no mask tensor.  No autograd graph is recorded;
* after each layer's gradients are final the bucketed reducer
  (:mod:`..parallel.reducer`) is told how far the flat gradient buffer is complete, so
  RCCL all-reduces overlap the rest of backward on a side stream;
* on the GPU the weight-gradient kernels run on a second stream (they are off the critical
  path), the SyncBN statistics travel through a one-kernel xGMI exchange fused into the
  finalize kernels (:mod:`..parallel.xgmi_allreduce`) when it passed its self-test, and the
  whole step is recorded once and replayed (:mod:`.graphs`): a command list of the backend's
  own launches by default, HIP graphs on request.

The ops come from an op backend (``ops.hip`` on MI355X, ``ops.ref`` on CPU), so the same
program is exercised by the CPU/gloo tests and checked against the eager model
(:class:`..models.resnet.ResNet`) with autograd.
"""
from __future__ import annotations

import hashlib
import logging
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..models.resnet import BlockSpec, ConvBNSpec, ResNetSpec
from ..ops import ref as _ref
from ..parallel import collectives as coll
from ..parallel import streams
from ..parallel.reducer import GradReducer
from ..utils.profiling import trace_range
from .params import FlatParams, make_buckets

MASK_NONE, MASK_TENSOR, MASK_RECOMPUTE = _ref.MASK_NONE, _ref.MASK_TENSOR, _ref.MASK_RECOMPUTE

_log = logging.getLogger(__name__)


class ReplicaMismatchError(RuntimeError):
    """The ranks of a job were not built alike (model, class count, bucket layout)."""


class CommFailure(RuntimeError):
    """The in-kernel xGMI statistics exchange timed out: the step's results are poisoned (NaN)."""


class StepReplayError(RuntimeError):
    """Recording a step for replay failed at a point from which the step cannot be repeated."""


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def backward_param_order(spec: ResNetSpec) -> List[str]:
    """Parameter keys in the order backward finalises their gradients."""
    order: List[str] = ["fc.weight", "fc.bias"]

    def unit_keys(u: ConvBNSpec) -> List[str]:
        return [f"{u.bn}.weight", f"{u.bn}.bias", f"{u.conv}.weight"]

    for b in reversed(spec.blocks):
        for u in reversed(b.units):
            order += unit_keys(u)
        if b.downsample is not None:
            order += unit_keys(b.downsample)
    order += unit_keys(spec.stem)
    return order


class _UnitCtx:
    """Tensors a conv+BN unit keeps from forward for its backward."""

    __slots__ = ("x_in", "y", "scale", "shift", "mean", "invstd", "count", "gram", "mom", "y_bytes")

    def __init__(self) -> None:
        self.x_in = self.y = self.scale = self.shift = self.mean = self.invstd = None
        self.count = 0
        # a unit whose output was never materialised (fused block tail): the Gram matrix and column means of its
        # input, kept for the folded backward, and the size its output would have had
        self.gram = self.mom = None
        self.y_bytes = 0


class NativeResNet:
    def __init__(self, spec: ResNetSpec, device: torch.device, ops, compute_dtype: torch.dtype = torch.bfloat16,
                 sync_bn: bool = False, group=None, bucket_cap_mb: float = 25.0, first_bucket_mb: float = 1.0,
                 overlap_comm: bool = True, bn_momentum: float = 0.1, bn_eps: float = 1e-5,
                 hip_graphs: bool = False, syncbn_overlap: bool = False, syncbn_comm: str = "auto",
                 step_replay: str = "list", grad_allreduce_dtype: str = "fp32", xgmi_timeout_s: float = 60.0,
                 logger: Optional[logging.Logger] = None) -> None:
        self.spec = spec
        self.device = torch.device(device)
        self.ops = ops
        self.dtype = compute_dtype
        self.group = group
        self.log = logger or _log
        initialized = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if initialized else 1
        forced = initialized and os.environ.get("PDT_FORCE_COLLECTIVES") == "1"   # debug: see GradReducer
        self.sync_bn = bool(sync_bn) and (self.world > 1 or forced)
        self.syncbn_overlap = bool(syncbn_overlap)
        self.bn_momentum = bn_momentum
        self.bn_eps = bn_eps
        self.training = True

        schema = spec.state_dict_schema()
        order = backward_param_order(spec)
        assert sorted(order) == sorted(spec.parameter_keys())
        self.params = FlatParams([(k, schema[k][0]) for k in order], self.device, compute_dtype)
        self.buckets = make_buckets(self.params, bucket_cap_mb, first_bucket_mb)
        if grad_allreduce_dtype not in ("fp32", "bf16"):
            raise ValueError(f"grad_allreduce_dtype must be fp32 or bf16, not {grad_allreduce_dtype!r}")
        self.reducer = GradReducer(self.params.g, self.buckets, group=group, overlap=overlap_comm,
                                   comm_dtype=torch.bfloat16 if grad_allreduce_dtype == "bf16" else None)
        self._optimizer = None           # set by attach_optimizer: its update then runs per gradient bucket

        # BatchNorm running statistics: flat fp32 buffers, one slot per BN in forward order.
        self.units = spec.conv_bn_units()
        self._bn_off: Dict[str, Tuple[int, int]] = {}
        off = 0
        for u in self.units:
            self._bn_off[u.bn] = (off, u.cout)
            off += u.cout
        self.running_mean = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.running_var = torch.ones(off, dtype=torch.float32, device=self.device)
        self.num_batches_tracked: Dict[str, int] = {u.bn: 0 for u in self.units}

        self.mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=self.device)

        # Data-gradient weight copies [Cin, R, S, Cout] (only for backends that want them).
        self._needs_wt = bool(getattr(ops, "needs_wt", False))
        self.wt: Optional[torch.Tensor] = None
        if self._needs_wt:
            self.wt = torch.zeros(self.params.total, dtype=compute_dtype, device=self.device)
        self._stem_w = None
        self._fc_aux = None
        self._tw_table = None            # (table, rows, blocks) of the batched dgrad-weight transpose
        self._eval_affine: Optional[Dict[str, Tuple[torch.Tensor, torch.Tensor]]] = None
        self._ctx: Dict[str, _UnitCtx] = {}
        self._saved: Dict[str, object] = {}
        self._deferred = None            # weight-gradient closure held back under SyncBN (see _wgrad)
        self._wgrad_stream = None        # second stream for the weight-gradient kernels (GPU only)
        self._wgrad_keep: List[tuple] = []
        self._fork_event = self._join_event = None
        # SyncBN statistics exchange: the one-kernel xGMI all-reduce when it is available and has
        # passed its self-test on this machine, else the process group's all-reduce.
        self._xgmi = None
        self.syncbn_comm = "rccl"
        if self.sync_bn and self.device.type == "cuda" and getattr(ops, "name", "") == "hip" \
                and os.environ.get("PDT_SYNCBN_COMM", syncbn_comm) in ("auto", "xgmi"):
            from ..parallel.xgmi_allreduce import XgmiAllReduce
            x = XgmiAllReduce.create(group, self.device)
            if x is not None and not x.self_test(production_timeout_s=xgmi_timeout_s):
                x.close()
                x = None
            self._xgmi = x
            self.syncbn_comm = "xgmi" if x is not None else "rccl"
        # One-GPU rehearsal of a multi-rank job (several ranks share the card over a gloo group): the gradient buckets can
        # go through a second instance of the exchange instead of gloo's host staging -- a kernel on the communication
        # stream that ends only when the peer's matching kernel runs, which is how an RCCL all-reduce behaves (RCCL itself
        # refuses two ranks on one device: "Duplicate GPU detected").  tests/test_stream_budget_gpu.py drives it.
        self._bucket_xgmi = None
        if os.environ.get("PDT_BUCKETS_OVER_XGMI") == "1" and self.device.type == "cuda" and dist.is_initialized() \
                and getattr(ops, "name", "") == "hip":
            from ..parallel.xgmi_allreduce import XgmiAllReduce
            x = XgmiAllReduce.create(group, self.device)
            if x is None or not x.self_test(production_timeout_s=xgmi_timeout_s):
                raise RuntimeError("PDT_BUCKETS_OVER_XGMI=1: the exchange is not available between these ranks")
            self._bucket_xgmi = x
            coll.set_device_rendezvous(x, group)
        # Fused statistics path: the conv leaves its statistics in the accumulator and ONE kernel
        # does fold + rank exchange + finalize (needs the in-kernel exchange when SyncBN is on).
        self._fused_stats = bool(getattr(ops, "fused_finalize", False)) and self.device.type == "cuda" \
            and (not self.sync_bn or self._xgmi is not None) and os.environ.get("PDT_FUSED_STATS", "1") == "1"
        # Producer -> consumer pairs around activations larger than this run in batch slices (the
        # MI355X Infinity Cache holds 256 MB; a slice plus what streams past it must fit).
        self._slice_bytes = 0
        if self.device.type == "cuda" and getattr(ops, "batch_chunks", False):
            self._slice_bytes = int(float(os.environ.get("PDT_SLICE_MB", "224")) * 2 ** 20)
        # BatchNorm-backward reductions accumulated by the data-gradient epilogue that produces
        # the incoming gradient (saves the reduction pass over g and y).
        self._fused_bwd_stats = bool(getattr(ops, "fused_bwd_stats", False)) and self.device.type == "cuda" \
            and os.environ.get("PDT_FUSED_BWD_STATS", "1") == "1"
        self._fused_pool_bwd = os.environ.get("PDT_FUSED_POOL_BWD", "1") == "1"
        # ... and its backward sums come out of the first block's data-gradient epilogue (no pass over the 112x112
        # tensor for them): the pooling keeps the winners' values and the pooled tensor's packed sign mask
        self._pool_winners = self._fused_pool_bwd and self._fused_bwd_stats and bool(getattr(ops, "pool_winners", False)) \
            and os.environ.get("PDT_POOL_WINNERS", "1") == "1"
        # ReLU backward of a block output from a packed bit mask written by the pass that produced
        # the tensor, instead of re-reading the tensor for its sign in the data-gradient epilogue.
        self._relu_bits = bool(getattr(ops, "relu_bits", False)) and os.environ.get("PDT_RELU_BITS", "1") == "1"
        self._relu_bits_only = bool(getattr(ops, "bwd_stats_need_bits", False))
        self._half_addend = bool(getattr(ops, "half_addend", False)) and os.environ.get("PDT_HALF_ADDEND", "1") == "1"
        # BatchNorm backward of a bottleneck block's last unit folded into the gradient products of its 1x1
        # convolution: the widest gradient tensor of the block is never written (see _folded_tail_backward).
        self._bn_fold = bool(getattr(ops, "bn_fold", False)) and os.environ.get("PDT_BN_FOLD", "1") == "1"
        # ... where the tensor it avoids is large: the fold costs a few small kernels and a C x C Gram product per
        # layer.  At batch 512 it wins clearly on the 56x56 and 28x28 stages (822 / 411 MB avoided; stand-alone
        # 933 -> 589 and 474 -> 376 us per block, bench/fold_layers.py), is even on the 14x14 stage (205 MB: 5 % slower
        # stand-alone, 0.1 ms faster inside the step, where the elementwise pass shares the chip with the weight
        # gradients) and loses on the 7x7 stage (103 MB).
        self._bn_fold_min_bytes = int(float(os.environ.get("PDT_BN_FOLD_MIN_MB", "150")) * 2 ** 20) \
            if self.device.type == "cuda" else 0
        # ... and where it is not too wide: per byte avoided, the extra multiplies (Gram matrix, second operand) grow
        # with the channel count.  The 7x7 stage (2048 channels) loses at any batch: 61.75 ms/step folded against
        # 61.47 at batch 1024, where its tensors are as large as the 14x14 stage's at 512.  (0: no limit.)
        self._bn_fold_max_cout = int(os.environ.get("PDT_BN_FOLD_MAX_COUT", "1024")) if self.device.type == "cuda" else 0
        # Folded units take their backward sum s1 from the (unscaled) weight-gradient product, so the kernel that
        # produces their incoming gradient only adds up the gradient itself and reads no BatchNorm input for them;
        # with that, a stride-1 projection shortcut is folded as well (PDT_BN_FOLD_PRODUCT=0: sums as before).
        self._bn_fold_product = self._bn_fold and os.environ.get("PDT_BN_FOLD_PRODUCT", "1") == "1"
        self._colsum_pending: set = set()     # accumulators a forward pass filled and no backward has collected yet
        # Block tails in ONE kernel in forward: the batch statistics of a 1x1 convolution's output follow from the
        # Gram matrix of its input (which the folded backward wants anyway), so BatchNorm is folded into the weights
        # before the convolution runs and "+ shortcut, ReLU" is its epilogue (see _fused_tail_forward).
        self._fwd_fuse = self._bn_fold_product and os.environ.get("PDT_FWD_FUSE", "1") == "1" \
            and hasattr(ops, "conv_fprop_fused_tail")
        # Step replay (see .graphs): the second step of a batch signature is recorded, later ones
        # replay the record.  "list": a command list of the backend's own launches (default);
        # "graph": HIP graphs separated by host actions.  PDT_HIP_GRAPHS: 0 = off, 1 = on,
        # force = graphs whatever the number of segments; PDT_STEP_REPLAY overrides the mode.
        env = os.environ.get("PDT_HIP_GRAPHS")
        on = (bool(hip_graphs) if env is None else env in ("1", "force")) and self.device.type == "cuda" \
            and getattr(ops, "name", "") == "hip"
        self._force_graphs = env == "force"
        mode = os.environ.get("PDT_STEP_REPLAY", "graph" if self._force_graphs else step_replay)
        if mode not in ("list", "graph", "off"):
            raise ValueError(f"step_replay must be list, graph or off, not {mode!r}")
        self.step_replay = mode if on else "off"
        self.hip_graphs = self.step_replay != "off"
        # Weight gradients on a second stream: pays with stream launches (eager or "list"), not
        # inside HIP graphs, whose parallel branches launch slowly.
        want_ws = os.environ.get("PDT_WGRAD_STREAM", "1" if self.step_replay == "list" else "0") == "1"
        if want_ws and self.device.type == "cuda" and getattr(ops, "name", "") == "hip":
            # PDT_WGRAD_PRIO: stream priority of the weight-gradient stream (HIP: lower value = served first)
            prio = os.environ.get("PDT_WGRAD_PRIO")
            self._wgrad_stream = streams.get("wgrad", self.device, int(prio) if prio else None)
            self._fork_event, self._join_event = ops.new_event(), ops.new_event()
        # With an overlapped reducer the communication stream waits for the weight-gradient stream at every bucket;
        # otherwise the main stream joins it there itself.  PDT_WGRAD_JOIN=1 forces the joins (A/B measurements).
        self._reducer_waits_wgrad = (self._wgrad_stream is not None and os.environ.get("PDT_WGRAD_JOIN", "0") != "1"
                                     and self.reducer.add_producer_stream(self._wgrad_stream))
        self._rec = None                 # program being recorded (StepProgram / ListProgram), else None
        self.replay_failures = 0
        self._plans: Dict[tuple, object] = {}
        self._eager_seen: Dict[tuple, int] = {}
        self.refresh_derived_weights()

    # ------------------------------------------------------------------ parameters / state
    def rm(self, bn: str) -> torch.Tensor:
        o, c = self._bn_off[bn]
        return self.running_mean[o:o + c]

    def rv(self, bn: str) -> torch.Tensor:
        o, c = self._bn_off[bn]
        return self.running_var[o:o + c]

    def _wt_view(self, key: str) -> Optional[torch.Tensor]:
        if self.wt is None:
            return None
        e = self.params.entries[key]
        co, r, s, ci = e.shape
        return self.wt[e.offset:e.offset + e.numel].view(ci, r, s, co)

    def refresh_derived_weights(self) -> None:
        """Rebuild every weight image derived from the compute mirror (after an optimizer
        step or a state load): dgrad copies, the packed stem filter, the padded classifier."""
        P, ops = self.params, self.ops
        if self._needs_wt:
            batched = getattr(ops, "transpose_weights_batched", None)
            if batched is not None and self.device.type == "cuda":
                if self._tw_table is None:
                    ents = []
                    for u in self.units[1:]:
                        e = P.entries[f"{u.conv}.weight"]
                        co, r, sdim, ci = e.shape
                        ents.append((e.offset, co, r * sdim, ci))
                    self._tw_table = ops.make_transpose_table(ents, self.device)
                batched(P.pc, self.wt, *self._tw_table)
            else:
                for u in self.units[1:]:
                    key = f"{u.conv}.weight"
                    ops.transpose_weights(P.compute(key), self._wt_view(key))
        # refreshed in place: captured step graphs hold these addresses
        self._stem_w = ops.prepare_stem_weight(P.compute("conv1.weight"), out=self._stem_w)
        prep_fc = getattr(ops, "prepare_fc", None)
        self._fc_aux = prep_fc(P.compute("fc.weight"), out=self._fc_aux) if prep_fc is not None else None
        self._eval_affine = None

    def train(self, mode: bool = True) -> "NativeResNet":
        self.training = mode
        return self

    def eval(self) -> "NativeResNet":
        return self.train(False)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Model state in the torchvision ResNet schema (fp32, OIHW, no ``module.`` prefix)."""
        exported = self.params.export("p")
        out: Dict[str, torch.Tensor] = {}
        for key, (shape, dtype) in self.spec.state_dict_schema().items():
            if key in exported:
                out[key] = exported[key]
            elif key.endswith("running_mean"):
                out[key] = self.rm(key[: -len(".running_mean")]).clone()
            elif key.endswith("running_var"):
                out[key] = self.rv(key[: -len(".running_var")]).clone()
            else:
                bn = key[: -len(".num_batches_tracked")]
                out[key] = torch.tensor(self.num_batches_tracked[bn], dtype=torch.int64)
        return out

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True) -> None:
        state = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
        schema = self.spec.state_dict_schema()
        if strict:
            missing = [k for k in schema if k not in state and not k.endswith("num_batches_tracked")]
            unexpected = [k for k in state if k not in schema]
            if missing or unexpected:
                raise KeyError(f"state dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
        self.params.load_from(state, "p", strict=strict)
        for u in self.units:
            if f"{u.bn}.running_mean" in state:
                self.rm(u.bn).copy_(state[f"{u.bn}.running_mean"].to(torch.float32))
                self.rv(u.bn).copy_(state[f"{u.bn}.running_var"].to(torch.float32))
            nbt = state.get(f"{u.bn}.num_batches_tracked")
            if nbt is not None:
                self.num_batches_tracked[u.bn] = int(nbt)
        self.refresh_derived_weights()

    def attach_optimizer(self, optimizer) -> None:
        """Run ``optimizer``'s update per gradient bucket, enqueued right behind the bucket's
        all-reduce on the communication stream, instead of as one pass over the whole model
        after backward: when backward ends only the last small bucket and its slice of the
        update remain.  Backward never reads what the update writes (master weights, momentum,
        the forward bf16 mirror): data gradients use the transposed weight copies, which are
        refreshed after the step.  ``optimizer.finish_step()`` is then the caller's only duty."""
        self._optimizer = optimizer
        red = self.reducer
        if optimizer is None:
            red.set_bucket_hook(None)
        else:
            red.set_bucket_hook(lambda lo, hi: optimizer.step_range(lo, hi, red.grad_scale))
        self._plans.clear()              # a recorded step has the old arrangement baked in
        self._eager_seen.clear()

    @property
    def optimizer_in_backward(self) -> bool:
        return self._optimizer is not None

    def layout_fingerprint(self) -> Tuple[str, int, str, str]:
        """What every rank of a job must agree on before flat buffers are broadcast over each
        other: model, element count, per-entry shapes (in flat order), gradient bucket layout."""
        P = self.params
        shapes = hashlib.sha256(repr([(k, P.entries[k].shape, P.entries[k].offset) for k in P.order]).encode())
        buckets = hashlib.sha256(repr(self.buckets).encode())
        return (self.spec.name if hasattr(self.spec, "name") else type(self.spec).__name__, int(P.total),
                shapes.hexdigest(), buckets.hexdigest())

    def verify_replicas(self) -> None:
        """Collective: raise :class:`ReplicaMismatchError` on EVERY rank unless all ranks hold
        the same parameter layout (DDP's ``_verify_param_shape_across_processes``,
        ``torch/nn/parallel/distributed.py:862``, behind reference ``train_distributed.py:198``).
        Broadcasting or reducing flat buffers of different layouts corrupts silently or hangs."""
        if self.world <= 1:
            return
        mine = self.layout_fingerprint() + (int(self.running_mean.numel()),)
        seen = [None] * self.world
        dist.all_gather_object(seen, mine, group=self.group)
        if any(s != seen[0] for s in seen):
            rows = "\n".join(f"  rank {r}: model={s[0]} elements={s[1]} shapes={s[2][:12]} buckets={s[3][:12]} "
                             f"bn={s[4]}" for r, s in enumerate(seen))
            raise ReplicaMismatchError("ranks were built with different models or bucket layouts:\n" + rows)

    def broadcast_state(self, src: int = 0) -> None:
        """Initial replica synchronisation: a layout check, then ONE flat broadcast each for
        parameters and for BN buffers (DDP does this at construction,
        ``torch/nn/parallel/distributed.py:862-871``)."""
        self.verify_replicas()
        if self.world > 1:
            coll.broadcast(self.params.p, src=src, group=self.group)
            coll.broadcast(self.params.m, src=src, group=self.group)
            self.sync_buffers(src)
            self.params.refresh_compute()
            self.refresh_derived_weights()

    def sync_buffers(self, src: int = 0) -> None:
        """Rank ``src``'s running statistics win (what DDP's per-forward buffer broadcast gives
        upstream without SyncBN).  With SyncBN every rank already holds identical values."""
        if self.world > 1:
            coll.broadcast(self.running_mean, src=src, group=self.group)
            coll.broadcast(self.running_var, src=src, group=self.group)
            self._eval_affine = None

    # ------------------------------------------------------------------ BN helpers
    def _host(self, fn) -> None:
        """A host-side action of the step (collective, stream/event call): run it now; while a
        step program is being captured it also ends the current graph and is recorded, so the
        replay issues it at the same point between two graphs."""
        if self._rec is None:
            fn()
        else:
            self._rec.host(fn)

    def _stat_all_reduce(self, t: torch.Tensor) -> None:
        """SUM a per-layer statistics vector over the ranks (in place)."""
        x = self._xgmi
        if x is not None and x.supports(t):
            x.all_reduce(t)                  # an ordinary kernel on the compute stream
        else:
            group = self.group
            self._host(lambda: coll.all_reduce(t, group=group))

    def comm_failed(self) -> bool:
        """Has an in-kernel statistics exchange of THIS rank timed out?  Reads a pinned host word
        the kernel writes: no synchronisation, cheap enough for every step; it may lag the GPU
        by the steps still queued."""
        return any(x is not None and x.peek() != 0 for x in (self._xgmi, self._bucket_xgmi))

    def raise_on_comm_failure(self) -> None:
        """Fail fast: a timed-out exchange produced NaN statistics (by construction, never stale
        memory), so the replica is unusable; the job must end and resume from a checkpoint."""
        if self.comm_failed():
            raise CommFailure("an xGMI exchange timed out or failed its checksum (a peer rank is gone or stalled beyond "
                              "runtime.xgmi_timeout_s); this replica's state is poisoned -- resume from the last "
                              "checkpoint")

    def check_comm_health(self) -> bool:
        """Collective and synchronising: True on every rank iff no rank has seen a timeout.  Run
        before anything is persisted (a checkpoint must not capture poisoned weights)."""
        return all(x.healthy() for x in (self._xgmi, self._bucket_xgmi) if x is not None)

    def _bn_train(self, u: ConvBNSpec, ctx: _UnitCtx, stats, n_local: int) -> None:
        """``stats``: what the conv returned -- a moments tensor, or (fused backends) the pending
        accumulator, which one kernel folds, exchanges over the ranks and finalizes."""
        P, ops = self.params, self.ops
        if self.sync_bn:
            inv_world, count = 1.0 / self.world, n_local * self.world
        else:
            inv_world, count = 1.0, n_local
        args = (P.param(f"{u.bn}.weight"), P.param(f"{u.bn}.bias"), self.rm(u.bn), self.rv(u.bn),
                self.bn_momentum, self.bn_eps, count)
        if self._fused_stats and u.cout <= ops.FUSED_MAX_C:
            ctx.scale, ctx.shift, ctx.mean, ctx.invstd = ops.bn_finalize_fused(
                stats, self._xgmi if self.sync_bn else None, inv_world, *args)
        else:
            moments = ops.stats_to_moments(stats) if hasattr(ops, "stats_to_moments") else stats
            if self.sync_bn:
                self._stat_all_reduce(moments)
            ctx.scale, ctx.shift, ctx.mean, ctx.invstd = ops.bn_finalize(moments, inv_world, *args)
        ctx.count = count

    def _bn_backward(self, u: ConvBNSpec, ctx: _UnitCtx, g: torch.Tensor, mask_src, mask_mode: int,
                     prereduced: bool = False, pool_idx=None) -> torch.Tensor:
        """Gradient w.r.t. the raw conv output; accumulates dgamma / dbeta into the flat buffer.
        ``prereduced``: the data-gradient kernel that produced ``g`` already accumulated this
        layer's sums (see :meth:`_bstats`), the reduction pass over ``g`` and ``y`` is skipped.
        ``pool_idx``: ``g`` is the gradient of the max pooling behind this layer's ReLU (the stem);
        both passes gather the dense gradient from it instead of reading a materialised one."""
        ekw = {"pool_idx": pool_idx} if pool_idx is not None else {}
        sums = self._bn_backward_sums(u, ctx, g, mask_src, mask_mode, prereduced, pool_idx)
        return self.ops.bn_bwd_elemt(g, ctx.y, ctx.mean, ctx.invstd, ctx.scale, ctx.shift, sums, 1.0 / ctx.count,
                                     mask_src, mask_mode, **ekw)

    def _bn_backward_sums(self, u: ConvBNSpec, ctx: _UnitCtx, g: torch.Tensor, mask_src, mask_mode: int,
                          prereduced: bool = False, pool_idx=None) -> torch.Tensor:
        """First half of :meth:`_bn_backward`: the (globally reduced) sums the elementwise pass needs;
        dgamma / dbeta are accumulated here."""
        P, ops = self.params, self.ops
        kw = {"prereduced": True} if prereduced else {}
        if pool_idx is not None:
            kw["pool_idx"] = pool_idx
        if self.sync_bn and self._xgmi is not None and self._fused_stats:
            return ops.bn_bwd_reduce(g, ctx.y, ctx.mean, ctx.invstd, ctx.scale, ctx.shift, mask_src, mask_mode,
                                     P.grad(f"{u.bn}.weight"), P.grad(f"{u.bn}.bias"), comm=self._xgmi, **kw)
        sums = ops.bn_bwd_reduce(g, ctx.y, ctx.mean, ctx.invstd, ctx.scale, ctx.shift, mask_src, mask_mode,
                                 P.grad(f"{u.bn}.weight"), P.grad(f"{u.bn}.bias"), **kw)
        self._allreduce_bwd_sums(sums)
        return sums

    def _allreduce_bwd_sums(self, sums: torch.Tensor) -> None:
        """SyncBN: sum the local ``[2, C]`` backward sums over the ranks, in place."""
        if self.sync_bn:
            # The all-reduce sits on the critical path (dgrad needs its result).  With
            # ``syncbn_overlap`` the weight gradient of the unit processed just before, which is
            # independent of it, is enqueued between the start and the wait of an asynchronous
            # collective.  That costs two cross-stream hops and ~4x the host time of a
            # synchronous call (which RCCL runs on the compute stream itself), so it is opt-in.
            group = self.group
            if self._xgmi is not None and self._xgmi.supports(sums):
                self._xgmi.all_reduce(sums)
            elif self.syncbn_overlap and not coll.is_host_staged(sums, group):
                pending = []
                self._host(lambda: pending.append(dist.all_reduce(sums, group=group, async_op=True)))
                self._run_deferred()
                self._host(lambda: pending.pop().wait())
            else:
                self._host(lambda: coll.all_reduce(sums, group=group))

    def _wgrad(self, fn) -> None:
        """Run a weight-gradient closure now, or (SyncBN) hold it back until the next statistics
        all-reduce is in flight.  Closures run in submission order, so gradient buckets still
        complete in the order the reducer expects."""
        if self.sync_bn and self.syncbn_overlap and self._xgmi is None:
            self._run_deferred()
            self._deferred = fn
        else:
            fn()

    def _run_deferred(self) -> None:
        fn, self._deferred = self._deferred, None
        if fn is not None:
            fn()

    def _affine_eval(self, u: ConvBNSpec) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._eval_affine is None:
            self._eval_affine = {}
        if u.bn not in self._eval_affine:
            P = self.params
            self._eval_affine[u.bn] = self.ops.bn_eval_affine(
                P.param(f"{u.bn}.weight"), P.param(f"{u.bn}.bias"), self.rm(u.bn), self.rv(u.bn), self.bn_eps)
        return self._eval_affine[u.bn]

    # ------------------------------------------------------------------ forward
    def _unit_conv_bn(self, u: ConvBNSpec, x: torch.Tensor, train: bool, pre=None):
        """conv + BatchNorm statistics/finalize of one unit: ``(y, scale, shift)``, nothing applied yet.
        ``pre = (y, stats)``: the convolution has already been run (in batch slices, see
        :meth:`_apply_then_convs`)."""
        ops = self.ops
        if pre is not None:
            y, moments = pre
        else:
            w = self.params.compute(f"{u.conv}.weight")
            kw = {"defer_stats": True} if (train and self._fused_stats) else {}
            y, moments = ops.conv_fprop(x, w, u.ksize, u.ksize, u.stride, u.pad, self.rm(u.bn) if train else None,
                                        **kw)
        if train:
            ctx = _UnitCtx()
            ctx.x_in, ctx.y = x, y
            self._bn_train(u, ctx, moments, y.numel() // y.shape[-1])
            self._ctx[u.bn] = ctx
            return y, ctx.scale, ctx.shift
        scale, shift = self._affine_eval(u)
        return y, scale, shift

    def _unit_forward(self, u: ConvBNSpec, x: torch.Tensor, residual: Optional[torch.Tensor], relu: bool,
                      train: bool, pre=None, colsum_for: Optional[ConvBNSpec] = None) -> torch.Tensor:
        """``colsum_for``: the unit that consumes the result, when its BatchNorm backward is going to be folded
        (it wants the per-channel sums of this tensor, which the pass that writes it can add up for free)."""
        y, scale, shift = self._unit_conv_bn(u, x, train, pre)
        if colsum_for is not None and train and self._will_fold(colsum_for, y):
            slot = colsum_for.bn
            if slot in self._colsum_pending:      # a training forward whose backward never ran: drop its sums
                self.ops.colsum_take(slot, y.shape[-1], self.device)
            self._colsum_pending.add(slot)
            self._saved[f"colsum:{slot}"] = True
            return self.ops.bn_apply(y, scale, shift, residual, relu, colsum_slot=slot)
        return self.ops.bn_apply(y, scale, shift, residual, relu)

    def _will_fold(self, u: ConvBNSpec, x_in: torch.Tensor) -> bool:
        """:meth:`_can_fold` before the unit has run: from its input (1x1 / stride 1: the output has the input's
        pixels and ``u.cout`` channels)."""
        return self._can_fold(u, (x_in.numel() // x_in.shape[-1]) * u.cout * x_in.element_size())

    # -- cache-sized batch slices ---------------------------------------------------------------
    def _n_slices(self, t: torch.Tensor) -> int:
        """Number of batch slices for a producer -> consumer pair around tensor ``t``: 1 unless the
        tensor is larger than the memory-side cache, then the smallest divisor of the batch that
        makes a slice fit."""
        if self._slice_bytes <= 0:
            return 1
        nbytes = t.numel() * t.element_size()
        if nbytes <= self._slice_bytes:
            return 1
        B = t.shape[0]
        for n in range(2, 9):
            if B % n == 0 and nbytes // n <= self._slice_bytes:
                return n
        return 1

    def _apply_then_convs(self, tail, users: List[ConvBNSpec], train: bool):
        """Block tail (BatchNorm apply + shortcut + ReLU) and the convolutions that read its result,
        run slice by slice over the batch: a 400 MB activation written in full and then re-read in
        full goes to HBM and back, a 200 MB slice is still in the Infinity Cache when the
        convolution reads it.  BatchNorm statistics of the convolutions accumulate over the
        slices in the usual accumulator.  Returns the block output and ``{bn: (y, stats)}``."""
        ops = self.ops
        kind, y, scale, shift, extra = tail
        n = self._n_slices(y)
        B = y.shape[0]
        step = B // n
        out = torch.empty_like(y)
        bits = self._new_bits(out, train)
        ys, stats = {}, {}
        for c in range(n):
            sl = slice(c * step, (c + 1) * step)
            bkw = {"bits_out": bits[sl]} if bits is not None else {}
            if kind == "apply2":
                y2, sc2, sh2 = extra
                ops.bn_apply2(y[sl], scale, shift, y2[sl], sc2, sh2, True, out=out[sl], **bkw)
            else:
                ops.bn_apply(y[sl], scale, shift, extra[sl] if extra is not None else None, True, out=out[sl], **bkw)
            for k, u in enumerate(users):
                w = self.params.compute(f"{u.conv}.weight")
                Ho = (y.shape[1] + 2 * u.pad - u.ksize) // u.stride + 1
                Wo = (y.shape[2] + 2 * u.pad - u.ksize) // u.stride + 1
                if u.bn not in ys:
                    ys[u.bn] = torch.empty(B, Ho, Wo, u.cout, dtype=y.dtype, device=y.device)
                _, st = ops.conv_fprop(out[sl], w, u.ksize, u.ksize, u.stride, u.pad,
                                       self.rm(u.bn) if train else None, defer_stats=train, out=ys[u.bn][sl],
                                       acc_slot="stats" if k == 0 else f"stats{k}")
                if st is not None:
                    st.count = B * Ho * Wo          # the accumulator ends up holding every slice
                    stats[u.bn] = st
        return out, {bn: (ys[bn], stats.get(bn)) for bn in ys}

    def _block_forward(self, b: BlockSpec, x, train: bool):
        """``x``: the block input, or the previous block's un-applied tail (a tuple, see below) when
        that tensor is worth slicing.  Returns this block's output the same way."""
        ops = self.ops
        pre = {}
        fuse_tail = train and self._can_fuse_tail(b, tuple(x[1].shape if isinstance(x, tuple) else x.shape))
        if isinstance(x, tuple):
            # (a fused tail runs the projection shortcut itself, as a second operand of the last convolution)
            users = [b.units[0]] + ([b.downsample] if (b.downsample is not None and not fuse_tail) else [])
            x, pre = self._apply_then_convs(x, users, train)
        shortcut = None                    # (y, scale, shift) of a projection shortcut, applied in the tail
        identity = x
        if b.downsample is not None and not fuse_tail:
            shortcut = self._unit_conv_bn(b.downsample, x, train, pre.get(b.downsample.bn))
            if not hasattr(ops, "bn_apply2"):
                identity = ops.bn_apply(*shortcut, None, False)
                shortcut = None
        out = x
        last = len(b.units) - 1
        for i, u in enumerate(b.units):
            p = pre.get(u.bn) if i == 0 else None
            if i < last:
                out = self._unit_forward(u, out, None, True, train, p,
                                         colsum_for=b.units[last] if (i == last - 1 and last > 0) else None)
                continue
            if fuse_tail:
                return self._fused_tail_forward(b, out, x)
            y, scale, shift = self._unit_conv_bn(u, out, train, p)
            tail = ("apply2", y, scale, shift, shortcut) if shortcut is not None else \
                ("apply", y, scale, shift, identity)
            if self._can_slice(train) and self._n_slices(y) > 1 and b is not self.spec.blocks[-1]:
                return tail                 # applied by the next block, slice by slice with its convs
            return self._apply_tail(tail, train)
        return out

    def _can_fuse_tail(self, b: BlockSpec, shape) -> bool:
        """The block's tail (last 1x1 convolution + BatchNorm [+ projection shortcut + its BatchNorm] + add + ReLU)
        runs as one kernel: a bottleneck block whose last unit -- and shortcut, if it has one -- will be folded in
        backward (the two go together: without the convolution outputs there is nothing else backward could use)."""
        if not self._fwd_fuse or len(b.units) < 2:
            return False
        u, d = b.units[-1], b.downsample
        B, H, W, _ = shape
        s = 1
        for unit in b.units:
            s *= unit.stride
        Ho, Wo = H // s, W // s
        nbytes = B * Ho * Wo * u.cout * torch.empty(0, dtype=self.dtype).element_size()
        if not self._can_fold(u, nbytes):
            return False
        if d is None:
            return s == 1
        if d.stride == 1:
            return s == 1 and self._can_fold(d, nbytes)
        return d.stride == 2 and s == 2 and H % 2 == 0 and W % 2 == 0 and self._half_addend \
            and self._can_fold(d, nbytes, strided=True)

    def _fused_tail_forward(self, b: BlockSpec, a: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """``relu(bn3(conv3(a)) + shortcut)`` without ``conv3(a)`` (or the projection's output) in memory.

        ``y = a . W^T`` is a pointwise product, so ``sum_m y = W . colsum(a)`` and ``sum_m y^2 = W (a^T a) W^T``: the
        batch statistics are known BEFORE the convolution runs.  BatchNorm's scale is folded into the weights, its
        shift becomes the accumulators' start value, the identity is the epilogue's addend (a projection shortcut is
        a second operand with its own folded weights), ReLU and the packed mask close the epilogue.  The Gram matrix
        and column means are kept: they are what the folded backward of these units needs instead of ``y``."""
        ops, P = self.ops, self.params
        u, d = b.units[-1], b.downsample
        Bn, H, W, _ = a.shape
        rows = Bn * H * W

        def unit_stats(unit: ConvBNSpec, inp: torch.Tensor, slot: Optional[str]) -> _UnitCtx:
            gram, mom = ops.gram_and_means(inp, slot)
            st = ops.bn_stats_from_gram(P.compute(f"{unit.conv}.weight"), gram, mom, rows, self.rm(unit.bn))
            ctx = _UnitCtx()
            ctx.x_in, ctx.gram, ctx.mom = inp, gram, mom
            ctx.y_bytes = rows * unit.cout * inp.element_size()
            self._bn_train(unit, ctx, st, rows)
            self._ctx[unit.bn] = ctx
            return ctx

        slot = u.bn if self._saved.pop(f"colsum:{u.bn}", False) else None
        self._colsum_pending.discard(u.bn)
        ctx = unit_stats(u, a, slot)
        w2 = sc2 = sh2 = xs = None
        if d is not None:
            xs = ops.subsample2(x) if d.stride == 2 else x          # the pixels the projection reads, dense
            cd = unit_stats(d, xs, None)
            w2, sc2, sh2 = P.compute(f"{d.conv}.weight"), cd.scale, cd.shift
        wf, shift = ops.bn_fuse_fprop_weights(P.compute(f"{u.conv}.weight"), ctx.scale, ctx.shift, w2, sc2, sh2)
        out = torch.empty(Bn, H, W, u.cout, dtype=a.dtype, device=a.device)
        bits = self._new_bits(out, True)
        ops.conv_fprop_fused_tail(a, wf, shift, addend=x if d is None else None, in2=xs, out=out,
                                  **({"bits_out": bits} if bits is not None else {}))
        return out

    def _can_slice(self, train: bool) -> bool:
        return self._slice_bytes > 0 and (self._fused_stats or not train)

    def _new_bits(self, out: torch.Tensor, train: bool):
        """Packed ReLU mask of a block output (one byte per 8 channels), attached to the tensor: the
        data-gradient kernel that applies this ReLU's backward reads it instead of the tensor."""
        if not (train and self._relu_bits):
            return None
        bits = self.ops.new_relu_bits(out)
        out._pdt_relu_bits = bits
        return bits

    def _apply_tail(self, tail, train: bool = False) -> torch.Tensor:
        kind, y, scale, shift, extra = tail
        bits = self.ops.new_relu_bits(y) if (train and self._relu_bits) else None
        kw = {"bits_out": bits} if bits is not None else {}
        if kind == "apply2":
            out = self.ops.bn_apply2(y, scale, shift, *extra, True, **kw)
        else:
            out = self.ops.bn_apply(y, scale, shift, extra, True, **kw)
        if bits is not None:
            out._pdt_relu_bits = bits
        return out

    def forward(self, img: torch.Tensor, train: Optional[bool] = None) -> torch.Tensor:
        """Logits ``[B, num_classes]`` for a loader batch (float NCHW or uint8 NHWC)."""
        train = self.training if train is None else train
        ops = self.ops
        self._ctx.clear()
        self._saved.clear()
        if train and self._rec is None:
            self._count_training_forward()
        with trace_range("forward"):
            x0 = ops.input_transform(img, self.mean, self.std, self.dtype)
            stem = self.spec.stem
            kw = {"defer_stats": True} if (train and self._fused_stats) else {}
            y, moments = ops.stem_fprop(x0, self._stem_w, self.rm(stem.bn) if train else None, **kw)
            if train:
                ctx = _UnitCtx()
                ctx.x_in, ctx.y = x0, y
                self._bn_train(stem, ctx, moments, y.numel() // y.shape[-1])
                self._ctx[stem.bn] = ctx
                scale, shift = ctx.scale, ctx.shift
            else:
                scale, shift = self._affine_eval(stem)
            if train and self._pool_winners:
                # keep y at the pooling winners and the pooled tensor's sign mask: the first block's data gradient
                # then adds up the stem BatchNorm's backward sums itself (see backward)
                x, idx, self._saved["pool_ywin"] = ops.bn_relu_maxpool_fwd(y, scale, shift, winners=True)
            else:
                x, idx = ops.bn_relu_maxpool_fwd(y, scale, shift)
            if train:
                self._saved["stem_shape"] = tuple(y.shape)
                self._saved["pool_idx"] = idx
            for b in self.spec.blocks:
                x = self._block_forward(b, x, train)
            if isinstance(x, tuple):
                x = self._apply_tail(x, train)
            pooled = ops.avgpool_fwd(x)
            if train:
                self._saved["final_hw"] = (x.shape[1], x.shape[2])
                self._saved["final_out"] = x
                self._saved["pooled"] = pooled
            P = self.params
            logits = ops.linear_fwd(pooled, P.compute("fc.weight"), P.param("fc.bias"))
        return logits

    # ------------------------------------------------------------------ backward
    def _conv_wgrad_fn(self, u: ConvBNSpec, x: torch.Tensor, dy: torch.Tensor, final: bool = True):
        P, ops, red = self.params, self.ops, self.reducer
        key = f"{u.conv}.weight"

        def run() -> None:
            end = P.end_offset(key)
            ws = self._wgrad_stream
            if ws is None:
                ops.conv_wgrad(x, dy, u.ksize, u.ksize, u.stride, u.pad, P.grad(key))
            else:
                # The weight gradient is off the critical path (only the optimiser and the bucket
                # all-reduce wait for it): on a second stream it fills the slots the data-gradient
                # kernels leave idle in their last, partial round of tiles.
                cur = torch.cuda.current_stream(self.device).cuda_stream
                ops.stream_wait_stream(ws.cuda_stream, cur, self._fork_event)
                with torch.cuda.stream(ws):
                    ops.conv_wgrad(x, dy, u.ksize, u.ksize, u.stride, u.pad, P.grad(key))
                self._wgrad_keep.append((x, dy))       # their memory must outlive the side-stream kernel
                if final and red.completes_bucket(end) and not self._reducer_waits_wgrad:
                    self._join_wgrad()
            if final:                                  # (a batch slice: the gradient is not complete yet)
                self._mark_ready(end)
        return run

    def _join_wgrad(self) -> None:
        if self._wgrad_stream is not None and self._wgrad_keep:
            cur = torch.cuda.current_stream(self.device).cuda_stream
            self.ops.stream_wait_stream(cur, self._wgrad_stream.cuda_stream, self._join_event)
            self._wgrad_keep.clear()

    def _mark_ready(self, end_offset: int) -> None:
        red = self.reducer
        if red.completes_bucket(end_offset):        # otherwise nothing would be launched
            self._host(lambda: red.mark_ready(end_offset))

    def _bstats(self, u: Optional[ConvBNSpec], mask_mode: int):
        """``bwd_stats`` argument for the data-gradient kernel whose result feeds ``u``'s BatchNorm
        backward, or None when the backend cannot fuse that reduction."""
        if u is None or not self._fused_bwd_stats:
            return None
        ctx = self._ctx[u.bn]
        return {"bwd_stats": (ctx.y, ctx.mean, ctx.scale, ctx.shift, mask_mode)}

    def _can_slice_bwd(self) -> bool:
        # (the deferred weight-gradient closures of the RCCL overlap mode are whole-layer units)
        return self._slice_bytes > 0 and not (self.sync_bn and self.syncbn_overlap and self._xgmi is None)

    @staticmethod
    def _ysize(ctx: _UnitCtx):
        """The unit's output for :meth:`_can_fold`: the tensor, or its size when it was never materialised."""
        return ctx.y if ctx.y is not None else ctx.y_bytes

    def _can_fold(self, u: ConvBNSpec, y, strided: bool = False) -> bool:
        """The unit's BatchNorm backward is folded into its convolution's gradient products: a 1x1 / stride 1
        convolution with channel counts the kernels take (every bottleneck block's last unit) whose output
        ``y`` is large enough for the fold to pay.  ``strided``: a stride-2 one (a projection shortcut), for
        which the caller subsamples the input first."""
        if not self._bn_fold or u.ksize != 1 or u.stride != (2 if strided else 1) or u.pad != 0:
            return False
        nbytes = y if isinstance(y, int) else y.numel() * y.element_size()
        if nbytes < self._bn_fold_min_bytes or (self._bn_fold_max_cout and u.cout > self._bn_fold_max_cout):
            return False
        if getattr(self.ops, "name", "") != "hip":
            return True
        return u.cin % 64 == 0 and u.cout % 64 == 0 and u.cin <= self.ops.BN_FOLD_MAX_C

    def _folded_wgrad_fn(self, u: ConvBNSpec, x: torch.Tensor, g: torch.Tensor, fold, prod=None, gram=None, mom=None):
        """Like :meth:`_conv_wgrad_fn`, for a unit whose BatchNorm backward is folded (``g`` is the gradient
        w.r.t. the BatchNorm output; ``prod``: the unscaled product ``g^T x`` when it exists already)."""
        P, ops, red = self.params, self.ops, self.reducer
        key = f"{u.conv}.weight"
        kw = {"prod": prod} if prod is not None else {}
        if gram is not None:
            kw.update(gram=gram, mom=mom)           # kept from the forward pass (fused block tail)
        elif self._saved.pop(f"colsum:{u.bn}", False):
            kw["amean_slot"] = u.bn                 # the forward pass left the column sums of x behind
            self._colsum_pending.discard(u.bn)

        def run() -> None:
            end = P.end_offset(key)
            ws = self._wgrad_stream
            if ws is None:
                ops.conv_wgrad_folded(x, g, fold, P.compute(key), P.grad(key), **kw)
            else:
                cur = torch.cuda.current_stream(self.device).cuda_stream
                ops.stream_wait_stream(ws.cuda_stream, cur, self._fork_event)
                with torch.cuda.stream(ws):
                    ops.conv_wgrad_folded(x, g, fold, P.compute(key), P.grad(key), **kw)
                self._wgrad_keep.append((x, g, prod, gram, mom, *fold.tensors()))   # must outlive the side-stream kernels
                if red.completes_bucket(end) and not self._reducer_waits_wgrad:
                    self._join_wgrad()
            self._mark_ready(end)
        return run

    def _folded_tail_backward(self, u: ConvBNSpec, prev_u: Optional[ConvBNSpec], ctx: _UnitCtx, g: torch.Tensor,
                              prereduced: bool, s0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Backward of a bottleneck block's last unit (1x1 convolution + BatchNorm) WITHOUT the gradient w.r.t.
        the convolution output: that tensor has 4x the channels of the unit's input, and the elementwise pass
        that writes it plus the two kernels that read it back are a quarter of the block's backward traffic.
        The convolution output is itself a product with the weights, so the BatchNorm-backward terms move into
        small per-step matrices (``ops.bn_fold_prepare``) and both gradient products read ``g`` and the unit's
        INPUT instead (``ops.conv_dgrad_folded`` / ``ops.conv_wgrad_folded``)."""
        P, ops = self.params, self.ops
        key = f"{u.conv}.weight"
        prod = None
        if s0 is None:
            sums = self._bn_backward_sums(u, ctx, g, None, MASK_NONE, prereduced)
        else:
            # ``s0 = sum g`` came with g; ``sum g * xhat`` follows from the product g^T x, which the weight gradient
            # needs anyway: formed here, on the critical stream, scaled and added by the correction kernel later
            prod = ops.zeros_f32(P.grad(key).shape, self.device)
            ops.conv_wgrad(ctx.x_in, g, 1, 1, 1, 0, prod)
            sums = ops.bn_fold_sums(prod, P.compute(key), s0, ctx.mean, ctx.invstd, P.grad(f"{u.bn}.weight"),
                                    P.grad(f"{u.bn}.bias"))
            self._allreduce_bwd_sums(sums)
        fold = ops.bn_fold_prepare(sums, ctx.scale, ctx.invstd, ctx.mean, 1.0 / ctx.count, P.compute(key),
                                   self._wt_view(key))
        self._wgrad(self._folded_wgrad_fn(u, ctx.x_in, g, fold, prod, ctx.gram, ctx.mom))
        return ops.conv_dgrad_folded(g, ctx.x_in, fold,
                                     **((self._bstats(prev_u, MASK_RECOMPUTE) or {}) if prev_u is not None else {}))

    def _sliced_tail_backward(self, u: ConvBNSpec, prev_u: ConvBNSpec, ctx: _UnitCtx, g: torch.Tensor,
                              prereduced: bool) -> torch.Tensor:
        """Backward of a block's last unit in batch slices: the BatchNorm elementwise pass writes a
        slice of dy and the data- and weight-gradient kernels of the unit's convolution consume it
        while it is still in the memory-side cache (dy of a first-stage block is 800 MB at batch
        512).  Sums of the consuming layers accumulate over the slices."""
        P, ops = self.params, self.ops
        sums = self._bn_backward_sums(u, ctx, g, None, MASK_NONE, prereduced)
        n = self._n_slices(ctx.y)
        B = g.shape[0]
        step = B // n
        key = f"{u.conv}.weight"
        x_in = ctx.x_in
        dy = torch.empty_like(ctx.y)
        dx = torch.empty_like(x_in)
        bst = self._bstats(prev_u, MASK_RECOMPUTE)
        for c in range(n):
            sl = slice(c * step, (c + 1) * step)
            ops.bn_bwd_elemt(g[sl], ctx.y[sl], ctx.mean, ctx.invstd, ctx.scale, ctx.shift, sums, 1.0 / ctx.count,
                             None, MASK_NONE, out=dy[sl])
            self._wgrad(self._conv_wgrad_fn(u, x_in[sl], dy[sl], final=(c == n - 1)))
            kw = {}
            if bst is not None:
                by, bmean, bscale, bshift, bmode = bst["bwd_stats"]
                kw = {"bwd_stats": (by[sl], bmean, bscale, bshift, bmode)}
            ops.conv_dgrad(dy[sl], P.compute(key), self._wt_view(key), tuple(x_in[sl].shape), u.ksize, u.ksize,
                           u.stride, u.pad, out=dx[sl], **kw)
        return dx

    def _block_backward(self, b: BlockSpec, g: torch.Tensor, g_prereduced=False,
                        prev_last: Optional[ConvBNSpec] = None) -> Tuple[torch.Tensor, object]:
        """``g``: gradient w.r.t. the block's pre-ReLU sum, i.e. ALREADY multiplied by the mask of
        the block's output ReLU (the producer of ``g`` fuses that).  Returns the gradient w.r.t.
        the block input, likewise already masked by the input tensor's own ReLU.

        Second result: whether that final kernel also accumulated the backward sums of
        ``prev_last``'s BatchNorm (what the caller passes as the next ``g_prereduced``).

        ``g_prereduced``: the kernel that produced ``g`` also accumulated the backward sums of this
        block's last BatchNorm (True), or only the per-channel sums of ``g`` itself (``"s0"``: the last unit is
        folded and takes the rest from its weight-gradient product).  ``prev_last``: the last unit of the block
        that produced this block's input -- the final data-gradient kernel here does the same for it."""
        P, ops, red = self.params, self.ops, self.reducer
        last = len(b.units) - 1
        x_block = self._ctx[b.units[0].bn].x_in
        cur = g
        dy_first = None
        s0 = None
        if g_prereduced == "s0":
            s0 = ops.grad_sums(g.shape[-1], self.device)
            g_prereduced = False
        for i in range(last, -1, -1):
            u = b.units[i]
            ctx = self._ctx.pop(u.bn)
            if i == last and i > 0 and (ctx.y is None or self._can_fold(u, ctx.y)):
                if ctx.y is None and s0 is None:
                    # a fused tail has no convolution output for the reduction pass over (g, y): s0 from g alone,
                    # the rest from the weight-gradient product
                    s0 = ops.channel_sums(cur)
                cur = self._folded_tail_backward(u, b.units[i - 1], ctx, cur, g_prereduced, s0)
                continue
            if i == last and i > 0 and self._can_slice_bwd() and self._n_slices(ctx.y) > 1:
                cur = self._sliced_tail_backward(u, b.units[i - 1], ctx, cur, g_prereduced)
                continue
            if i == last:
                dy = self._bn_backward(u, ctx, cur, None, MASK_NONE, prereduced=g_prereduced)
            else:
                dy = self._bn_backward(u, ctx, cur, None, MASK_RECOMPUTE, prereduced=self._fused_bwd_stats)
            key = f"{u.conv}.weight"
            self._wgrad(self._conv_wgrad_fn(u, ctx.x_in, dy))
            if i > 0:
                cur = ops.conv_dgrad(dy, P.compute(key), self._wt_view(key), tuple(ctx.x_in.shape),
                                     u.ksize, u.ksize, u.stride, u.pad,
                                     **(self._bstats(b.units[i - 1], MASK_RECOMPUTE) or {}))
            else:
                dy_first = dy
        u0 = b.units[0]
        key0 = f"{u0.conv}.weight"
        half = False
        if b.downsample is not None:
            d = b.downsample
            ctx = self._ctx.pop(d.bn)
            Bx, Hx, Wx, Cx = x_block.shape
            if ctx.y is None:
                # fused tail: the shortcut's input was subsampled in forward already
                other = self._folded_tail_backward(d, None, ctx, g, False, s0)
                dyd = None
                half = d.stride == 2
            elif s0 is not None and self._can_fold(d, ctx.y):
                # (same g as the block's last BatchNorm: the same s0)
                other = self._folded_tail_backward(d, None, ctx, g, False, s0)
                dyd = None
            elif s0 is not None and self._half_addend and Hx % 2 == 0 and Wx % 2 == 0 \
                    and self._can_fold(d, ctx.y, strided=True):
                # a stride-2 shortcut reads the even pixels only: on those, made dense once, it is a stride-1 unit,
                # and its data gradient stays at half resolution like the unfolded one below
                ctx.x_in = ops.subsample2(x_block)
                other = self._folded_tail_backward(d, None, ctx, g, False, s0)
                dyd = None
                half = True
            else:
                dyd = self._bn_backward(d, ctx, g, None, MASK_NONE)
                keyd = f"{d.conv}.weight"
                self._wgrad(self._conv_wgrad_fn(d, x_block, dyd))
            if dyd is None:
                pass
            elif self._half_addend and d.ksize == 1 and d.stride == 2 and d.pad == 0 and Hx % 2 == 0 and Wx % 2 == 0:
                # A stride-2 1x1 shortcut touches the even input pixels only: its data gradient is
                # kept at half resolution (a plain 1x1 product) and the kernel below adds it at
                # those pixels, instead of writing and re-reading a tensor that is 3/4 zeros.
                other = ops.conv_dgrad(dyd, P.compute(keyd), self._wt_view(keyd), (Bx, Hx // 2, Wx // 2, Cx),
                                       1, 1, 1, 0)
                half = True
            else:
                other = ops.conv_dgrad(dyd, P.compute(keyd), self._wt_view(keyd), tuple(x_block.shape),
                                       d.ksize, d.ksize, d.stride, d.pad)
        else:
            other = g                       # identity path
        # x_block is a post-ReLU tensor (previous block output, or the max-pooled stem activation,
        # where a zero means the whole pooling window was zero): apply its ReLU backward here.
        bits = getattr(x_block, "_pdt_relu_bits", None)
        mkw = {"relu_bits": bits} if bits is not None else {"relu_mask": x_block}
        if prev_last is not None and self._bn_fold_product \
                and self._can_fold(prev_last, self._ysize(self._ctx[prev_last.bn])):
            skw, pre = {"out_sums": True}, "s0"      # that unit is folded: all it wants with g is sum g
        elif prev_last is None and bits is not None and "pool_ywin" in self._saved:
            # The first block: its input gradient belongs to the max-pooled stem activation.  Every pooling window
            # hands its gradient to one element of the stem BatchNorm's output, so that layer's backward sums are
            # sums over the windows -- of this gradient (masked here) times the winner's value.
            c0 = self._ctx[self.spec.stem.bn]
            skw = {"bwd_stats": (self._saved.pop("pool_ywin"), c0.mean, c0.scale, c0.shift, MASK_NONE)}
            pre = True
        else:
            # the fused backward sums of the previous block's last BatchNorm need the packed mask
            skw = (self._bstats(prev_last, MASK_NONE) or {}) if (bits is not None or not self._relu_bits_only) else {}
            pre = bool(skw)
        g_in = ops.conv_dgrad(dy_first, P.compute(key0), self._wt_view(key0), tuple(x_block.shape),
                              u0.ksize, u0.ksize, u0.stride, u0.pad, addend=other, **mkw, **skw,
                              **({"addend_half": True} if half else {}))
        return g_in, pre

    def backward(self, dlogits: torch.Tensor) -> None:
        """Back-propagate ``dlogits``; gradients accumulate into the flat gradient buffer."""
        P, ops, red = self.params, self.ops, self.reducer
        if red.enabled:
            self._host(red.begin_step)
        with trace_range("backward"):
            pooled = self._saved.pop("pooled")
            dpooled = ops.linear_bwd(pooled, dlogits, P.compute("fc.weight"), self._fc_aux,
                                     P.grad("fc.weight"), P.grad("fc.bias"))
            self._mark_ready(P.end_offset("fc.bias"))
            H, W = self._saved.pop("final_hw")
            g = ops.avgpool_bwd(dpooled, H, W, relu_mask=self._saved.pop("final_out"))
            blocks = self.spec.blocks
            pre = False
            for j in range(len(blocks) - 1, -1, -1):
                prev_last = blocks[j - 1].units[-1] if j > 0 else None
                g, pre = self._block_backward(blocks[j], g, pre, prev_last)
            stem = self.spec.stem
            ctx = self._ctx.pop(stem.bn)
            pool_idx, stem_shape = self._saved.pop("pool_idx"), self._saved.pop("stem_shape")
            if self._fused_pool_bwd:
                # the dense 112x112 gradient of the pooling is never materialised
                dy = self._bn_backward(stem, ctx, g, None, MASK_RECOMPUTE, prereduced=pre is True, pool_idx=pool_idx)
            else:
                da = ops.maxpool_bwd(g, pool_idx, stem_shape)
                dy = self._bn_backward(stem, ctx, da, None, MASK_RECOMPUTE)
            self._run_deferred()
            self._join_wgrad()
            ops.stem_wgrad(ctx.x_in, dy, P.grad("conv1.weight"))
            self._mark_ready(P.end_offset("conv1.weight"))
        if red.enabled:
            with trace_range("allreduce_wait"):
                self._host(red.finish)

    # ------------------------------------------------------------------ steps
    def _count_training_forward(self) -> None:
        for bn in self.num_batches_tracked:
            self.num_batches_tracked[bn] += 1

    def _forward_backward_eager(self, img: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        logits = self.forward(img, train=True)
        loss, dlogits = self.ops.softmax_ce(logits, label, 1.0)
        self.backward(dlogits)
        return loss

    def forward_backward(self, img: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        """One forward + backward; returns the local mean loss (0-dim fp32 device tensor).

        With step replay the second step of a batch signature is recorded (command list) or
        captured (HIP graphs) and every later one replays it; signatures beyond ``MAX_PLANS``
        (the short last batch of an epoch is the second) run eagerly."""
        if not self.hip_graphs:
            return self._forward_backward_eager(img, label)
        key = (tuple(img.shape), img.dtype, tuple(label.shape), label.dtype)
        plan = self._plans.get(key)
        if plan is None:
            seen = self._eager_seen.get(key, 0)
            self._eager_seen[key] = seen + 1
            if seen == 0 or len(self._plans) >= self.MAX_PLANS:
                # the first step of a signature runs eagerly: it fills every lazily built cache
                # (accumulators, tap tables, occupancy queries) that must not happen in a capture
                return self._forward_backward_eager(img, label)
            if self.step_replay == "list":
                plan = self._record_plan(img, label)
                self._plans[key] = plan
                if plan is not None:                 # recording ran the step for real
                    self._count_training_forward()
                    self.log.info("step replay: recorded a command list for batch %s (%d library calls, %d host "
                                  "actions)", tuple(img.shape), plan[3].num_calls, plan[3].num_host_actions)
                    return plan[2].clone()
            else:
                plan = self._capture_plan(img, label)
                self._plans[key] = plan
            if plan is None:                         # capture failed or not worthwhile: stay eager
                self.hip_graphs = False
                return self._forward_backward_eager(img, label)
        static_img, static_label, loss, program = plan
        static_img.copy_(img, non_blocking=True)
        static_label.copy_(label, non_blocking=True)
        self._count_training_forward()
        with trace_range("step_replay"):
            program.replay()
        return loss.clone()

    def _record_plan(self, img: torch.Tensor, label: torch.Tensor):
        """Run one step for real while the backend records every call it makes."""
        from .graphs import ListProgram
        program = ListProgram(self.device, self.ops)
        static_img, static_label = img.clone(), label.clone()
        # The eager step before this one left its activations cached in the allocator's default pool, and the recording
        # allocates the same amount again from its private pool: hand the cached blocks back to the driver first, or a
        # batch sized for more than half of the HBM runs out of memory right here.
        torch.cuda.synchronize(self.device)
        torch.cuda.empty_cache()
        try:
            with torch.cuda.use_mem_pool(program.pool):
                self._rec = program
                program.begin()
                try:
                    loss = self._forward_backward_eager(static_img, static_label)
                finally:
                    self._rec = None
                    program.finish()
        except Exception as exc:                     # noqa: BLE001 -- fall back to the eager path where that is sound
            self.replay_failures += 1
            self._ctx.clear()
            self._saved.clear()
            self._deferred = None
            self._wgrad_keep.clear()
            if self.optimizer_in_backward:
                # The step died half way: the buckets launched before the exception have ALREADY been updated (weights,
                # momentum, zeroed gradients), the others hold partial gradient sums.  Re-running the batch would update
                # the first ones twice, and under data parallelism only this rank would do so.  There is no sound way on
                # from here: end the job (the trainer keeps logs and checkpoints; resume from the last one).
                raise StepReplayError(f"step recording failed mid-step with the per-bucket optimizer active "
                                      f"({type(exc).__name__}: {exc}); the replica has taken a partial update") from exc
            self.log.warning("step recording failed (%s: %s): this batch signature runs EAGERLY from now on "
                             "(several times the host cost per step)", type(exc).__name__, exc)
            self.params.g.zero_()                    # partial gradient sums of the aborted attempt
            return None
        return static_img, static_label, loss, program

    MAX_PLANS = 2
    MAX_HOST_ACTIONS = 16

    def active_replay_mode(self) -> str:
        """``list`` / ``graph`` when at least one batch signature is being replayed, ``off``
        when replay is disabled, ``pending`` before the first recording, ``eager(failed)``
        when recording failed."""
        if not self.hip_graphs:
            return "eager(failed)" if self.replay_failures else "off"
        if any(p is not None for p in self._plans.values()):
            return self.step_replay
        return "eager(failed)" if self.replay_failures else "pending"

    def _capture_plan(self, img: torch.Tensor, label: torch.Tensor):
        from .graphs import StepProgram
        static_img, static_label = img.clone(), label.clone()
        cur = torch.cuda.current_stream(self.device)
        side = streams.get("capture", self.device)
        side.wait_stream(cur)
        program = StepProgram(self.device)
        loss = None
        try:
            with torch.cuda.stream(side):
                self._rec = program
                program.begin()
                try:
                    loss = self._forward_backward_eager(static_img, static_label)
                    program.finish()
                finally:
                    self._rec = None
                    program.abort()
        except Exception as exc:                     # noqa: BLE001 -- fall back to the eager path
            self.log.warning("step-graph capture failed (%s: %s): this batch signature runs EAGERLY from now on",
                             type(exc).__name__, exc)
            self.replay_failures += 1
            self._ctx.clear()
            self._saved.clear()
            self._deferred = None
            cur.wait_stream(side)
            return None
        cur.wait_stream(side)
        if program.num_host_actions > self.MAX_HOST_ACTIONS and not self._force_graphs:
            # Every graph boundary costs ~15 us of GPU idle time: a step cut into a hundred
            # segments by per-layer collectives runs faster eagerly.
            return None
        return static_img, static_label, loss, program

    @torch.no_grad()
    def eval_step(self, img: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
        """``[loss_sum, correct@1, correct@5, n]`` for a validation batch."""
        logits = self.forward(img, train=False)
        return self.ops.ce_metrics(logits, label)
