"""HIP op backend: ctypes bindings to the gfx950 kernel library.

Same function contract as :mod:`.ref` (which documents the math); tensors are NHWC bf16
on a GPU, per-channel vectors fp32.  Kernels are enqueued on the CURRENT torch stream.
The library must have been built (``python -m pytorch_distributed_training_amd.ops.build``);
importing this module without it raises -- there is no silent fall-back.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import build as _build
from .ref import MASK_NONE, MASK_RECOMPUTE, MASK_TENSOR  # noqa: F401  (re-exported)

name = "hip"
# conv_dgrad reads the transposed copies ``wt`` only: the forward weights may be updated while backward still runs
dgrad_reads_forward_weights = False
needs_wt = True
NREP = 16

_LIB_PATH = os.environ.get("PDT_HIP_LIB") or _build.LIB      # PDT_HIP_LIB: e.g. the host-sanitized debug build
if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} is missing: build the gfx950 kernels first with "
        "`python -m pytorch_distributed_training_amd.ops.build`"
    )
_real = ctypes.CDLL(_LIB_PATH)

_vp, _i, _ll, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float


def _sig(fn_name: str, *argtypes) -> None:
    fn = getattr(_real, fn_name)
    fn.argtypes = list(argtypes)
    fn.restype = ctypes.c_int


_sig("pdt_sgd_step", _vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _i, _vp)
_sig("pdt_bn_apply", _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp)
_sig("pdt_bn_apply_ex", _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp)
_sig("pdt_bn_apply2", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp)
_sig("pdt_bn_bwd_elemt", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _ll, _i, _i, _vp)
_sig("pdt_input_pad", _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _vp)
_sig("pdt_crop_resize_flip", _vp, _vp, _vp, _i, _i, _i, _vp)
_sig("pdt_maxpool_fwd", _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp)
_sig("pdt_bn_relu_maxpool_fwd", _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp)
_sig("pdt_maxpool_bwd", _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp)
_sig("pdt_bn_relu_maxpool_fwd_winners", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp)
_sig("pdt_pool_bn_bwd_reduce_accum", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp)
_sig("pdt_pool_bn_bwd_elemt", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _vp)
_sig("pdt_avgpool_fwd", _vp, _vp, _i, _i, _i, _vp)
_sig("pdt_avgpool_bwd", _vp, _vp, _vp, _i, _i, _i, _vp)
_sig("pdt_transpose_w", _vp, _vp, _i, _i, _i, _i, _vp)
_sig("pdt_stem_pack", _vp, _vp, _i, _vp)
_sig("pdt_stem_unpack_add", _vp, _vp, _i, _vp)
_sig("pdt_bn_stats_accum", _vp, _vp, _vp, _ll, _i, _vp)
_sig("pdt_bn_moments", _vp, _vp, _i, _f, _vp)
_sig("pdt_bn_finalize", _vp, _f, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _i, _vp)
_sig("pdt_bn_eval_affine", _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp)
_sig("pdt_bn_bwd_reduce_accum", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp)
_sig("pdt_bn_bwd_sums", _vp, _vp, _vp, _vp, _vp, _i, _vp)
_sig("pdt_softmax_ce", _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp)
_sig("pdt_ce_metrics", _vp, _vp, _vp, _i, _i, _i, _vp)
_sig("pdt_colsum_add", _vp, _vp, _i, _i, _i, _vp)
_sig("pdt_bn_finalize_fused", _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _i, _vp)
_sig("pdt_bn_bwd_sums_fused", _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp)
_sig("pdt_car_create", _i, _i, _i, ctypes.POINTER(ctypes.c_void_p))
_sig("pdt_car_ipc_handle", _vp, _vp)
_sig("pdt_car_open", _vp, _vp)
_sig("pdt_car_set_timeout_ms", _vp, _i)
_sig("pdt_car_corrupt_checksum", _vp, _i)
_sig("pdt_car_allreduce", _vp, _vp, _i, _vp)
_sig("pdt_car_allreduce_chunks", _vp, _vp, _ll, _vp)
_sig("pdt_car_query", _vp, _vp)
_sig("pdt_car_peek", _vp)
_sig("pdt_car_destroy", _vp)
_sig("pdt_conv_igemm", _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
     _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp)
_sig("pdt_conv_igemm_bstats", _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i,
     _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp)
_sig("pdt_conv_dma", _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
     _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp)
_sig("pdt_conv_wgrad", _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _ll, _vp)
_sig("pdt_conv_wgrad_ex", _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _ll,
     _vp, _vp)
_sig("pdt_conv_dma_ex", _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
     _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp)
_sig("pdt_bnfold_fwd_stats", _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp)
_sig("pdt_bnfold_scale_weights", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp)
_sig("pdt_bnfold_prepare", _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp)
_sig("pdt_bnfold_wgrad", _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp)
_sig("pdt_bnfold_sums", _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp)
_sig("pdt_set_deterministic", _i)
_sig("pdt_accumulator_rows")


_sig("pdt_transpose_w_batched", _vp, _vp, _vp, _i, _i, _vp)
_sig("pdt_zero_f32", _vp, _ll, _vp)
_sig("pdt_subsample2", _vp, _vp, _i, _i, _i, _i, _vp)
_sig("pdt_event_create", ctypes.POINTER(ctypes.c_void_p))
_sig("pdt_event_destroy", _vp)
_sig("pdt_stream_wait_stream", _vp, _vp, _vp)


class HipKernelError(RuntimeError):
    pass


# ---- call recording -------------------------------------------------------------------------
# Every kernel launch of this backend is one call into the library with plain arguments (raw
# pointers, integers, floats, a stream handle).  While ``_recording`` is a list, each call is
# appended to it as ``(function, args)`` besides being executed; replaying the list re-issues the
# same launches without any of the Python above it (see engine/graphs.ListProgram).  The
# pointers stay valid because the recorded step allocates from a memory pool of its own.
_recording = None


class _LibProxy:
    def __init__(self, lib) -> None:
        self._lib = lib

    def __getattr__(self, fn_name: str):
        fn = getattr(self._lib, fn_name)

        def call(*args, _fn=fn):
            rec = _recording
            if rec is not None:
                rec.append((_fn, args))
            return _fn(*args)

        setattr(self, fn_name, call)       # next lookup finds the wrapper directly
        return call


_lib = _LibProxy(_real)


def begin_recording(items: list) -> None:
    global _recording
    _recording = items


def end_recording() -> None:
    global _recording
    _recording = None


def new_event() -> int:
    ev = ctypes.c_void_p()
    _check(_real.pdt_event_create(ctypes.byref(ev)), "pdt_event_create")
    return ev.value


def stream_wait_stream(waiter: int, waited: int, event: int) -> None:
    """``waiter`` (raw stream handle) waits for everything enqueued so far on ``waited``."""
    _check(_lib.pdt_stream_wait_stream(waiter, waited, event), "pdt_stream_wait_stream")


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipKernelError(f"{what} failed with status {rc}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    # hundreds of launches per step: the raw-handle query is ~10x cheaper than building a
    # torch.cuda.Stream object each time
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _bf16c(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.bfloat16 or not t.is_cuda:
        raise TypeError(f"{what}: expected a bf16 GPU tensor, got {t.dtype} on {t.device}")
    return t if t.is_contiguous() else t.contiguous()


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{what}: expected an fp32 GPU tensor, got {t.dtype} on {t.device}")
    if not t.is_contiguous():
        raise ValueError(f"{what}: must be contiguous")
    return t


# --------------------------------------------------------------------------- small caches
_acc_cache: Dict[Tuple[int, int, str], torch.Tensor] = {}

# ---- deterministic mode ---------------------------------------------------------------------
# Float atomics make the low bits of BatchNorm statistics, weight gradients and the loss depend on the
# order in which workgroups finish.  ``set_deterministic(True)`` switches the library to single-writer
# accumulator rows folded in index order, weight gradients through per-slice slabs added in slice order,
# and ordered sums for the loss and the classifier bias: two runs on the same inputs then give the same
# bits (``runtime.deterministic``), a few percent slower.
_deterministic = False
_wgrad_ws: Dict[Tuple[int, int], torch.Tensor] = {}
WGRAD_WORKSPACE_BYTES = 128 << 20


def set_deterministic(on: bool) -> None:
    global _deterministic, NREP
    torch.cuda.synchronize()
    _check(_real.pdt_set_deterministic(int(bool(on))), "pdt_set_deterministic")
    _deterministic = bool(on)
    NREP = int(_real.pdt_accumulator_rows())
    _acc_cache.clear()         # the accumulators are sized by the number of replica rows
    if not on:
        _wgrad_ws.clear()


def is_deterministic() -> bool:
    return _deterministic


def _wgrad_workspace(device: torch.device, stream: int):
    """Slab workspace of the deterministic weight-gradient kernel: one per stream, because launches on
    different streams may run at the same time."""
    if not _deterministic:
        return None, 0
    key = (device.index or 0, stream)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = torch.empty(WGRAD_WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
        _wgrad_ws[key] = ws
    return ws.data_ptr(), ws.numel() * 4


def _accumulator(device: torch.device, C: int, kind: str, per_stream: bool = False) -> torch.Tensor:
    """Persistent zero-initialised ``[NREP, 2, C]`` accumulator (kernels re-zero it after use).

    ``per_stream``: one buffer per issuing stream.  For the kinds that are filled AND consumed inside one call (the
    "fold" scratch sums): such a call may come from the weight-gradient stream while another one with the same channel
    count runs on the main stream, and two streams adding into and re-zeroing one buffer would mix their sums.  Kinds
    that hand sums from one call to a later one (statistics, column sums) stay keyed by name: their producer and
    consumer may sit on different streams, ordered by the engine's events."""
    key = (device.index or 0, C, kind, int(_stream() or 0) if per_stream else None)
    acc = _acc_cache.get(key)
    if acc is None:
        acc = torch.zeros(NREP * 2 * C, dtype=torch.float32, device=device)
        _acc_cache[key] = acc
    return acc


class _Taps:
    __slots__ = ("T", "offy", "offx", "widx", "_keep")

    def __init__(self, offy, offx, widx) -> None:
        self.T = len(offy)
        oy = np.ascontiguousarray(np.asarray(offy, dtype=np.int8))
        ox = np.ascontiguousarray(np.asarray(offx, dtype=np.int8))
        wi = np.ascontiguousarray(np.asarray(widx, dtype=np.uint16))
        self._keep = (oy, ox, wi)
        self.offy = oy.ctypes.data if self.T else None
        self.offx = ox.ctypes.data if self.T else None
        self.widx = wi.ctypes.data if self.T else None


_tap_cache: Dict[tuple, object] = {}


def _fprop_taps(R: int, S: int, pad: int) -> _Taps:
    key = ("f", R, S, pad)
    if key not in _tap_cache:
        oy, ox, wi = [], [], []
        for r in range(R):
            for s in range(S):
                oy.append(r - pad); ox.append(s - pad); wi.append(r * S + s)
        _tap_cache[key] = _Taps(oy, ox, wi)
    return _tap_cache[key]


def _dgrad_taps(R: int, S: int, pad: int, stride: int, ph: int, pw: int) -> _Taps:
    key = ("d", R, S, pad, stride, ph, pw)
    if key not in _tap_cache:
        oy, ox, wi = [], [], []
        for r in range(R):
            if (ph + pad - r) % stride:
                continue
            for s in range(S):
                if (pw + pad - s) % stride:
                    continue
                oy.append((ph + pad - r) // stride); ox.append((pw + pad - s) // stride); wi.append(r * S + s)
        _tap_cache[key] = _Taps(oy, ox, wi)
    return _tap_cache[key]


def _stem_taps() -> _Taps:
    key = ("stem",)
    if key not in _tap_cache:
        _tap_cache[key] = _Taps(list(range(7)), [0] * 7, list(range(7)))
    return _tap_cache[key]


# The conv kernels address their operands through buffer descriptors with 32-bit byte offsets (out-of-range taps read
# zeros for free).  The streaming kernels (conv_dma.hip, conv_wgrad.hip) base the descriptor at the tile / slice they
# are working on, so ONE launch covers tensors of any size.  The register-staged fallback kernel (conv_igemm.hip: stem
# without LDS-DMA, classifier, odd channel counts) keeps a whole-tensor descriptor and therefore sees at most 2 GB of
# its gathered operand: beyond that its launches go out over batch ranges (rows of different images are independent,
# statistics accumulate across the launches).
MAX_OPERAND_BYTES = 0x70000000


def _batch_parts(B: int, nbytes: int):
    n = -(-nbytes // MAX_OPERAND_BYTES)
    if n <= 1 or B <= 1:
        return None
    step = -(-B // n)
    return [(b, min(b + step, B)) for b in range(0, B, step)]


def _part(t, b0: int, b1: int):
    return None if t is None else t[b0:b1]


def _igemm(inp, wgt, out, M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS, oy0, ox0, N, ldo, wrow, taps: _Taps,
           addend=None, mask=None, bias=None, stat_acc=None, stat_shift=None, tile: int = 0, bstat=None,
           mask_bits: bool = False, addend_half: bool = False, sums_only: bool = False) -> None:
    dma = (_USE_CONV_DMA and tile <= 0 and bias is None and ldo == N and N % 64 == 0 and Ci % 64 == 0 and taps.T >= 1
           and pix_stride % 8 == 0 and (mask is None or mask_bits) and (_DMA_TAPS or taps.T == 1)
           and N * wrow * 2 < 0x7fff0000)
    parts = None if dma else _batch_parts(inp.shape[0], inp.numel() * inp.element_size())
    if parts is not None:
        B = inp.shape[0]
        if out.shape[0] != B or M % B:
            raise ValueError("conv: cannot split this launch over the batch")
        for b0, b1 in parts:
            sub = None if bstat is None else (bstat[0], bstat[1][b0:b1], *bstat[2:])
            _igemm(inp[b0:b1], wgt, out[b0:b1], (M // B) * (b1 - b0), Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS,
                   oy0, ox0, N, ldo, wrow, taps, addend=_part(addend, b0, b1), mask=_part(mask, b0, b1), bias=bias,
                   stat_acc=stat_acc, stat_shift=stat_shift, tile=tile, bstat=sub, mask_bits=mask_bits,
                   addend_half=addend_half, sums_only=sums_only)
        return
    if dma:
        acc, by, bmean, bscale, bshift = bstat if bstat is not None else (stat_acc, None, None, None, None)
        rc = _lib.pdt_conv_dma(_p(inp), _p(wgt), _p(out), _p(addend), int(addend_half), _p(mask), _p(acc), _p(stat_shift),
                               _p(by), _p(bmean), _p(bscale), _p(bshift), M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS,
                               oy0, ox0, N, wrow, taps.T, taps.offy, taps.offx, taps.widx, -tile or _DMA_TILE, int(sums_only),
                               _stream())
        _check(rc, "pdt_conv_dma")
        return
    tile = max(tile, 0)
    if bstat is not None:        # (acc, y, mean, scale-or-None, shift-or-None): BN-backward sums of the consumer
        acc, by, bmean, bscale, bshift = bstat
        rc = _lib.pdt_conv_igemm_bstats(_p(inp), _p(wgt), _p(out), _p(addend), int(addend_half), _p(mask), int(mask_bits), _p(acc), _p(by), _p(bmean),
                                        _p(bscale), _p(bshift), M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS, oy0,
                                        ox0, N, ldo, wrow, taps.T, taps.offy, taps.offx, taps.widx, tile, _stream())
        _check(rc, "pdt_conv_igemm_bstats")
        return
    rc = _lib.pdt_conv_igemm(_p(inp), _p(wgt), _p(out), _p(addend), int(addend_half), _p(mask), int(mask_bits), _p(bias), _p(stat_acc), _p(stat_shift),
                             M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, Ho, Wo, OS, oy0, ox0, N, ldo, wrow, taps.T,
                             taps.offy, taps.offx, taps.widx, tile, _stream())
    _check(rc, "pdt_conv_igemm")


def _wgrad(dy, inp, dw, M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, N, ldy, wrow, taps: _Taps, tile: int = 0,
           target_blocks: int = 0, atomic_budget_mb: int = 0, row_scale=None, part_ok: bool = True) -> None:
    stream = _stream()
    ws, ws_bytes = _wgrad_workspace(dw.device, stream)
    rc = _lib.pdt_conv_wgrad_ex(_p(dy), _p(inp), _p(dw), M, Hi, Wi, pix_stride, Ci, Hg, Wg, IS, N, ldy, wrow, taps.T,
                                taps.offy, taps.offx, taps.widx, tile, target_blocks, atomic_budget_mb, ws, ws_bytes,
                                _p(row_scale), stream)
    if rc == -8 and part_ok:
        # deterministic mode: more slices than the workspace has slabs are needed to keep every slice within a
        # descriptor's reach -- the one case that still goes out over batch ranges
        parts = _batch_parts(inp.shape[0], max(inp.numel(), dy.numel()) * 2)
        B = inp.shape[0]
        if parts is not None and dy.shape[0] == B and M % B == 0:
            for b0, b1 in parts:
                _wgrad(dy[b0:b1], inp[b0:b1], dw, (M // B) * (b1 - b0), Hi, Wi, pix_stride, Ci, Hg, Wg, IS, N, ldy, wrow,
                       taps, tile=tile, target_blocks=target_blocks, atomic_budget_mb=atomic_budget_mb,
                       row_scale=row_scale, part_ok=False)
            return
    _check(rc, "pdt_conv_wgrad")


# Convolutions go to the streaming LDS-DMA kernel (csrc/conv_dma.hip) when its shape rules hold (channel counts in
# multiples of 64, dense output rows, no bias, ReLU mask in packed form): everything in a ResNet but the stem and the
# classifier.  PDT_CONV_DMA=0 keeps every convolution on the register-staged implicit-GEMM kernel (A/B measurements);
# a positive ``tile`` argument selects a tile of that kernel, a negative one a tile of the DMA kernel.
_USE_CONV_DMA = os.environ.get("PDT_CONV_DMA", "1") == "1"
_DMA_TILE = int(os.environ.get("PDT_DMA_TILE", "0"))          # measurement override of the DMA kernel's tile choice
_DMA_TAPS = os.environ.get("PDT_CONV_DMA_TAPS", "1") == "1"   # 0: only single-tap (1x1) launches use the DMA kernel
_STEM_DMA = os.environ.get("PDT_STEM_DMA", "1") == "1"        # 0: the stem convolution stays on the register-staged kernel


class PendingStats:
    """Per-channel statistics still sitting in the replicated accumulator a conv epilogue filled:
    consumed either by :func:`bn_finalize_fused` (one kernel: moments, rank exchange, finalize)
    or turned into a moments tensor by :func:`stats_to_moments`."""

    __slots__ = ("acc", "C", "count")

    def __init__(self, acc: torch.Tensor, C: int, count: int) -> None:
        self.acc, self.C, self.count = acc, C, count


fused_finalize = True       # this backend offers bn_finalize_fused / defer_stats
FUSED_MAX_C = 2048


def stats_to_moments(stats) -> torch.Tensor:
    if isinstance(stats, PendingStats):
        return _moments_from_acc(stats.acc, stats.C, stats.count)
    return stats


def _comm_handle(comm):
    return None if comm is None else comm._h


def bn_finalize_fused(stats: PendingStats, comm, inv_world: float, gamma, beta, running_mean, running_var,
                      momentum: float, eps: float, count: int):
    """``bn_moments`` + (all-reduce over the ranks of ``comm``, an ``XgmiAllReduce``) + ``bn_finalize``
    in one launch.  ``count`` is the GLOBAL element count; returns ``(scale, shift, mean, invstd)``."""
    C = stats.C
    out = torch.empty(4, C, dtype=torch.float32, device=gamma.device)
    unbias = float(count) / float(max(count - 1, 1))
    rc = _lib.pdt_bn_finalize_fused(_comm_handle(comm), _p(stats.acc), 1.0 / float(stats.count), inv_world, _p(gamma),
                                    _p(beta), _p(running_mean), _p(running_var), momentum, eps, unbias, _p(out[0]),
                                    _p(out[1]), _p(out[2]), _p(out[3]), C, _stream())
    _check(rc, "pdt_bn_finalize_fused")
    if comm is not None:
        comm.calls += 1
    return out[0], out[1], out[2], out[3]


def _moments_from_acc(acc: torch.Tensor, C: int, count: int) -> torch.Tensor:
    moments = torch.empty(2, C, dtype=torch.float32, device=acc.device)
    _check(_lib.pdt_bn_moments(_p(acc), _p(moments), C, 1.0 / float(count), _stream()), "pdt_bn_moments")
    return moments


# --------------------------------------------------------------------------- input / stem
def _check_boxes(src: torch.Tensor, boxes: torch.Tensor, what: str) -> None:
    """The kernel indexes the stored image with the box as given: refuse anything that leaves it."""
    if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[1] != src.shape[2] or src.shape[3] != 3:
        raise TypeError(f"{what}: expected stored images [B, S, S, 3] uint8, got {tuple(src.shape)} {src.dtype}")
    if boxes.dtype != torch.int32 or tuple(boxes.shape) != (src.shape[0], 5):
        raise TypeError(f"{what}: boxes must be int32 [B, 5] = (x, y, w, h, flip)")
    S = src.shape[1]
    b = boxes.cpu() if boxes.is_cuda else boxes
    ok = ((b[:, 0] >= 0) & (b[:, 1] >= 0) & (b[:, 2] >= 1) & (b[:, 3] >= 1) & (b[:, 0] + b[:, 2] <= S)
          & (b[:, 1] + b[:, 3] <= S) & (b[:, 4] >= 0) & (b[:, 4] <= 1))
    if not bool(ok.all()):
        raise ValueError(f"{what}: a crop box leaves the {S} x {S} stored image")


def crop_resize_flip(src: torch.Tensor, boxes: torch.Tensor, out_size: int) -> torch.Tensor:
    """RandomResizedCrop + flip of a stored uint8 batch on the GPU; ``boxes`` is a HOST int32 ``[B, 5]`` tensor
    (validated here, then copied to the device on the current stream)."""
    if not src.is_cuda:
        raise TypeError("crop_resize_flip: the stored images must already be on the GPU")
    if boxes.is_cuda:
        raise TypeError("crop_resize_flip: boxes are chosen (and checked) on the host")
    _check_boxes(src, boxes, "crop_resize_flip")
    src = src.contiguous()
    B, S = src.shape[0], src.shape[1]
    dbox = boxes.contiguous().to(src.device, non_blocking=True)
    out = torch.empty(B, out_size, out_size, 3, dtype=torch.uint8, device=src.device)
    _check(_lib.pdt_crop_resize_flip(_p(src), _p(dbox), _p(out), B, S, out_size, _stream()), "pdt_crop_resize_flip")
    return out


def input_transform(img: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Loader batch (fp32 NCHW normalised, or uint8 NHWC raw) -> zero-haloed NHWC4 bf16 image
    ``[B, H+6, Wp, 4]`` with the picture at offset (3, 3)."""
    if not img.is_cuda:
        raise TypeError("input_transform: the batch must already be on the GPU")
    img = img.contiguous()
    if img.dtype == torch.uint8:
        B, H, W, C = img.shape
        is_u8 = 1
    elif img.dtype == torch.float32:
        B, C, H, W = img.shape
        is_u8 = 0
    else:
        raise TypeError(f"input_transform: unsupported image dtype {img.dtype}")
    if C != 3:
        raise ValueError("input_transform: expected 3 colour channels")
    Hp, Wp = H + 6, (W + 8 + 1) // 2 * 2
    out = torch.empty(B, Hp, Wp, 4, dtype=torch.bfloat16, device=img.device)
    m, s = _host_mean_std(mean, std)
    rc = _lib.pdt_input_pad(_p(img), _p(out), B, H, W, Hp, Wp, is_u8, m[0], m[1], m[2], s[0], s[1], s[2], _stream())
    _check(rc, "pdt_input_pad")
    out._pdt_hw = (H, W)
    return out


_ms_cache: Dict[int, tuple] = {}


def _host_mean_std(mean: torch.Tensor, std: torch.Tensor):
    key = mean.data_ptr()
    if key not in _ms_cache:
        _ms_cache[key] = (tuple(float(v) for v in mean.tolist()), tuple(float(v) for v in std.tolist()))
    return _ms_cache[key]


def prepare_stem_weight(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[Co, 7, 7, 3]`` -> packed ``[Co, 7, 32]`` (k = s*4 + c, zero padded).  ``out``: a previous
    result to refresh in place (keeps its address, which captured graphs rely on)."""
    w = _bf16c(w, "prepare_stem_weight")
    Co = w.shape[0]
    wp = out if out is not None else torch.empty(Co, 7, 32, dtype=torch.bfloat16, device=w.device)
    _check(_lib.pdt_stem_pack(_p(w), _p(wp), Co, _stream()), "pdt_stem_pack")
    return wp


def _stem_geometry(x0: torch.Tensor):
    B, Hp, Wp, _ = x0.shape
    H, W = getattr(x0, "_pdt_hw", (Hp - 6, Wp - 8))
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    return B, Hp, Wp, Ho, Wo


def stem_fprop(x0: torch.Tensor, wp: torch.Tensor, stat_shift: Optional[torch.Tensor], defer_stats: bool = False):
    B, Hp, Wp, Ho, Wo = _stem_geometry(x0)
    Co = wp.shape[0]
    y = torch.empty(B, Ho, Wo, Co, dtype=torch.bfloat16, device=x0.device)
    M = B * Ho * Wo
    acc = _accumulator(x0.device, Co, "stats") if stat_shift is not None else None
    taps = _stem_taps()
    if _USE_CONV_DMA and _STEM_DMA and Co % 64 == 0 and x0.numel() * 2 < 0x7fff0000:
        # the LDS-DMA kernel with K slices of 32 (one tap = one slice of the packed image)
        rc = _lib.pdt_conv_dma(_p(x0), _p(wp), _p(y), None, 0, None, _p(acc), _p(stat_shift), None, None, None, None, M,
                               Hp, Wp, 4, 32, Ho, Wo, 2, Ho, Wo, 1, 0, 0, Co, 7 * 32, taps.T, taps.offy, taps.offx,
                               taps.widx, 5, 0, _stream())
        _check(rc, "pdt_conv_dma (stem)")
    else:
        _igemm(x0, wp, y, M, Hp, Wp, 4, 32, Ho, Wo, 2, Ho, Wo, 1, 0, 0, Co, Co, 7 * 32, taps,
               stat_acc=acc, stat_shift=stat_shift)
    if acc is None:
        return y, None
    return y, (PendingStats(acc, Co, M) if defer_stats else _moments_from_acc(acc, Co, M))


def stem_wgrad(x0: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor) -> None:
    B, Hp, Wp, Ho, Wo = _stem_geometry(x0)
    dy = _bf16c(dy, "stem_wgrad dy")
    Co = dy.shape[-1]
    scratch = torch.empty(Co, 7, 32, dtype=torch.float32, device=dy.device)
    _check(_lib.pdt_zero_f32(_p(scratch), scratch.numel(), _stream()), "pdt_zero_f32")
    _wgrad(dy, x0, scratch, B * Ho * Wo, Hp, Wp, 4, 32, Ho, Wo, 2, Co, Co, 7 * 32, _stem_taps())
    _check(_lib.pdt_stem_unpack_add(_p(scratch), _p(dw), Co, _stream()), "pdt_stem_unpack_add")



def _out_like(out: Optional[torch.Tensor], shape, device, what: str) -> torch.Tensor:
    """``out`` (a caller-provided bf16 destination, e.g. a batch slice of a larger tensor) or a new tensor."""
    if out is None:
        return torch.empty(*shape, dtype=torch.bfloat16, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.bfloat16 or not out.is_contiguous() or not out.is_cuda:
        raise ValueError(f"{what}: out must be a contiguous bf16 GPU tensor of shape {tuple(shape)}")
    return out

# Tile selection is the launchers' own (shape and grid-fill heuristics in conv_dma.hip / conv_igemm.hip /
# conv_wgrad.hip): swept against every explicit tile per layer at batch 256, 512 and 1024 it is within 1.5 % of the
# per-shape best (profiles/tile_table_r2.txt), so there is no per-shape table to keep in step with the kernels -- what
# upstream leaves to cudnn.benchmark autotuning is a fixed rule here.  ``tile=`` stays as a measurement override.


# --------------------------------------------------------------------------- convolution
def conv_fprop(x: torch.Tensor, w: torch.Tensor, R: int, S: int, stride: int, pad: int,
               stat_shift: Optional[torch.Tensor] = None, tile: int = 0, defer_stats: bool = False,
               out: Optional[torch.Tensor] = None, acc_slot: str = "stats"):
    """``acc_slot``: which statistics accumulator to use -- two convolutions whose deferred
    statistics are pending at the same time (interleaved batch slices) need distinct ones."""
    x = _bf16c(x, "conv_fprop x")
    w = _bf16c(w, "conv_fprop w")
    B, Hi, Wi, Ci = x.shape
    Co = w.shape[0]
    Ho, Wo = (Hi + 2 * pad - R) // stride + 1, (Wi + 2 * pad - S) // stride + 1
    y = _out_like(out, (B, Ho, Wo, Co), x.device, "conv_fprop")
    M = B * Ho * Wo
    acc = _accumulator(x.device, Co, acc_slot) if stat_shift is not None else None
    _igemm(x, w, y, M, Hi, Wi, Ci, Ci, Ho, Wo, stride, Ho, Wo, 1, 0, 0, Co, Co, R * S * Ci, _fprop_taps(R, S, pad),
           stat_acc=acc, stat_shift=stat_shift, tile=tile)
    if acc is None:
        return y, None
    return y, (PendingStats(acc, Co, M) if defer_stats else _moments_from_acc(acc, Co, M))


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, wt: torch.Tensor, in_shape, R: int, S: int, stride: int, pad: int,
               addend: Optional[torch.Tensor] = None, relu_mask: Optional[torch.Tensor] = None, tile: int = 0,
               bwd_stats=None, out: Optional[torch.Tensor] = None, relu_bits: Optional[torch.Tensor] = None,
               addend_half: bool = False, out_sums: bool = False):
    """``bwd_stats = (y, mean, scale, shift, mask_mode)`` of the BatchNorm layer that consumes the
    result ``g``: the epilogue also accumulates that layer's backward sums (what
    :func:`bn_bwd_reduce` would compute from ``g`` and ``y`` in a pass of its own); pass
    ``prereduced=True`` to :func:`bn_bwd_reduce` afterwards.  ``out_sums``: the epilogue adds up the result
    itself per channel instead (no other tensor is read); :func:`grad_sums` collects the sums."""
    dy = _bf16c(dy, "conv_dgrad dy")
    if wt is None:
        raise ValueError("conv_dgrad: the hip backend needs the transposed weight copy wt[Cin,R,S,Cout]")
    B, Hi, Wi, Ci = in_shape
    _, Ho, Wo, Co = dy.shape
    dx = _out_like(out, (B, Hi, Wi, Ci), dy.device, "conv_dgrad")
    if addend is not None:
        addend = _bf16c(addend, "conv_dgrad addend")
        want = (B, Hi // 2, Wi // 2, Ci) if addend_half else (B, Hi, Wi, Ci)
        if tuple(addend.shape) != want or (addend_half and (Hi % 2 or Wi % 2)):
            raise ValueError(f"conv_dgrad: addend shape {tuple(addend.shape)}, expected {want}")
    mask_src = None
    bits = relu_bits is not None
    if bits:            # packed ReLU mask of the tensor the gradient belongs to (see bn_apply bits_out)
        _check_bits(relu_bits, (B, Hi, Wi, Ci), "conv_dgrad relu_bits")
        mask_src = relu_bits
    elif relu_mask is not None:
        mask_src = _bf16c(relu_mask, "conv_dgrad relu_mask")
    bstat = None
    if bwd_stats is not None:
        by, bmean, bscale, bshift, bmode = bwd_stats
        by = _bf16c(by, "conv_dgrad bwd_stats y")
        if tuple(by.shape) != (B, Hi, Wi, Ci) or bmode not in (MASK_NONE, MASK_RECOMPUTE):
            raise ValueError("conv_dgrad: bwd_stats tensor/mode does not match the gradient being produced")
        recompute = bmode == MASK_RECOMPUTE
        bstat = (_accumulator(dy.device, Ci, "bwd"), by, bmean, bscale if recompute else None,
                 bshift if recompute else None)
    if bstat is not None and mask_src is not None and not bits:
        raise ValueError("conv_dgrad: with bwd_stats the ReLU mask must be given in packed form (relu_bits)")
    gacc = None
    if out_sums:
        if bstat is not None:
            raise ValueError("conv_dgrad: out_sums and bwd_stats share the epilogue's statistics path")
        gacc = _accumulator(dy.device, Ci, "gsum")
    if stride == 1:
        _igemm(dy, wt, dx, B * Hi * Wi, Ho, Wo, Co, Co, Hi, Wi, 1, Hi, Wi, 1, 0, 0, Ci, Ci, R * S * Co,
               _dgrad_taps(R, S, pad, 1, 0, 0), addend=addend, mask=mask_src, tile=tile, bstat=bstat, mask_bits=bits,
               addend_half=addend_half, stat_acc=gacc, sums_only=gacc is not None)
        return dx
    for ph in range(stride):
        for pw in range(stride):
            Hg = (Hi - ph + stride - 1) // stride
            Wg = (Wi - pw + stride - 1) // stride
            if Hg <= 0 or Wg <= 0:
                continue
            _igemm(dy, wt, dx, B * Hg * Wg, Ho, Wo, Co, Co, Hg, Wg, 1, Hi, Wi, stride, ph, pw, Ci, Ci, R * S * Co,
                   _dgrad_taps(R, S, pad, stride, ph, pw), addend=addend, mask=mask_src, tile=tile, bstat=bstat,
                   mask_bits=bits, addend_half=addend_half, stat_acc=gacc, sums_only=gacc is not None)
    return dx


def grad_sums(C: int, device: torch.device) -> torch.Tensor:
    """``[C]``: per-channel sums of the last gradient produced with ``conv_dgrad(out_sums=True)`` (collects and
    re-zeroes the accumulator)."""
    out = torch.empty(2, C, dtype=torch.float32, device=device)
    _check(_lib.pdt_bn_moments(_p(_accumulator(device, C, "gsum")), _p(out), C, 1.0, _stream()), "pdt_bn_moments")
    return out[0]


def colsum_take(slot: str, C: int, device: torch.device) -> torch.Tensor:
    """``[C]`` per-channel sums left by ``bn_apply(colsum_slot=slot)``; collects and re-zeroes the accumulator."""
    out = torch.empty(2, C, dtype=torch.float32, device=device)
    _check(_lib.pdt_bn_moments(_p(_accumulator(device, C, "colsum:" + slot)), _p(out), C, 1.0, _stream()),
           "pdt_bn_moments")
    return out[0]


def subsample2(x: torch.Tensor) -> torch.Tensor:
    """``x[:, ::2, ::2]`` of an NHWC tensor with even height and width, dense."""
    x = _bf16c(x, "subsample2")
    B, H, W, C = x.shape
    out = torch.empty(B, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
    _check(_lib.pdt_subsample2(_p(x), _p(out), B, H, W, C, _stream()), "pdt_subsample2")
    return out


def zeros_f32(shape, device: torch.device) -> torch.Tensor:
    """A zeroed fp32 tensor through the library's own kernel (so that a recorded step zeroes it again)."""
    t = torch.empty(shape, dtype=torch.float32, device=device)
    _check(_lib.pdt_zero_f32(_p(t), t.numel(), _stream()), "pdt_zero_f32")
    return t


# Measurement aid (bench/ablate.sh): PDT_ABLATE=wgrad drops every weight-gradient launch, so the step
# time that remains is what the critical stream costs without its co-runner.  Results are wrong.
_ABLATE = set(os.environ.get("PDT_ABLATE", "").split(","))
_ABLATE_WGRAD = "wgrad" in _ABLATE
_ABLATE_WGRAD_HALF = "wgrad_half" in _ABLATE        # every other launch: what a 2x faster kernel would buy in the step
_ABLATE_ELEMT = "elemt" in _ABLATE                  # no BatchNorm-backward elementwise pass (bound on fusing it away)
_ABLATE_APPLY = "apply" in _ABLATE                  # no BatchNorm-forward elementwise pass
_wgrad_calls = 0


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, R: int, S: int, stride: int, pad: int, dw: torch.Tensor,
               tile: int = 0, target_blocks: int = 0, atomic_budget_mb: int = 0) -> None:
    if _ABLATE_WGRAD:
        return
    if _ABLATE_WGRAD_HALF:
        global _wgrad_calls
        _wgrad_calls += 1
        if _wgrad_calls & 1:
            return
    x = _bf16c(x, "conv_wgrad x")
    dy = _bf16c(dy, "conv_wgrad dy")
    _f32c(dw, "conv_wgrad dw")
    B, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    _wgrad(dy, x, dw, B * Ho * Wo, Hi, Wi, Ci, Ci, Ho, Wo, stride, Co, Co, R * S * Ci, _fprop_taps(R, S, pad),
           tile=tile, target_blocks=target_blocks, atomic_budget_mb=atomic_budget_mb)


def transpose_weights(w: torch.Tensor, wt: torch.Tensor) -> None:
    Co, R, S, Ci = w.shape
    _check(_lib.pdt_transpose_w(_p(w), _p(wt), Co, R * S, Ci, Co, _stream()), "pdt_transpose_w")


TW_PER_BLOCK = 2048


def make_transpose_table(entries, device: torch.device):
    """``entries``: (flat element offset, Co, T, Ci) per conv weight -> (int32 device table, rows, blocks)."""
    rows, first = [], 0
    for off, Co, T, Ci in entries:
        tiled = int(Co % 64 == 0 and Ci % 64 == 0 and off % 8 == 0)      # 64 x 64 tiles through LDS, 16-byte rows
        nb = T * (Co // 64) * (Ci // 64) if tiled else (Co * T * Ci + TW_PER_BLOCK - 1) // TW_PER_BLOCK
        rows.append([off, Co, T, Ci, first, nb, tiled])
        first += nb
    table = torch.tensor(rows, dtype=torch.int32, device=device).contiguous()
    return table, len(rows), first


def transpose_weights_batched(w_flat: torch.Tensor, wt_flat: torch.Tensor, table: torch.Tensor, n: int, blocks: int) -> None:
    """Every ``[Co, T, Ci] -> [Ci, T, Co]`` copy of the flat weight mirror in one launch."""
    if w_flat.dtype != torch.bfloat16 or wt_flat.dtype != torch.bfloat16 or w_flat.numel() != wt_flat.numel():
        raise TypeError("transpose_weights_batched: expected two bf16 flat buffers of one size")
    _check(_lib.pdt_transpose_w_batched(_p(w_flat), _p(wt_flat), _p(table), n, blocks, _stream()),
           "pdt_transpose_w_batched")


# --------------------------------------------------------------------------- batch norm
def bn_stats(y: torch.Tensor, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    y = _bf16c(y, "bn_stats")
    C = y.shape[-1]
    rows = y.numel() // C
    acc = _accumulator(y.device, C, "stats")
    _check(_lib.pdt_bn_stats_accum(_p(y), _p(shift), _p(acc), rows, C, _stream()), "pdt_bn_stats_accum")
    return _moments_from_acc(acc, C, rows)


def bn_finalize(moments, inv_world: float, gamma, beta, running_mean, running_var, momentum: float, eps: float,
                count: int):
    C = gamma.numel()
    out = torch.empty(4, C, dtype=torch.float32, device=gamma.device)
    unbias = float(count) / float(max(count - 1, 1))
    rc = _lib.pdt_bn_finalize(_p(moments), inv_world, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                              momentum, eps, unbias, _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), C, _stream())
    _check(rc, "pdt_bn_finalize")
    return out[0], out[1], out[2], out[3]


def bn_eval_affine(gamma, beta, running_mean, running_var, eps: float):
    C = gamma.numel()
    out = torch.empty(2, C, dtype=torch.float32, device=gamma.device)
    rc = _lib.pdt_bn_eval_affine(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(out[0]), _p(out[1]),
                                 C, _stream())
    _check(rc, "pdt_bn_eval_affine")
    return out[0], out[1]


relu_bits = True             # bn_apply/bn_apply2(bits_out=...) and conv_dgrad(relu_bits=...)
bwd_stats_need_bits = True   # conv_dgrad(bwd_stats=...) takes its ReLU mask in packed form only
half_addend = True           # conv_dgrad(addend_half=True): addend given at the even output pixels only


def new_relu_bits(like: torch.Tensor) -> torch.Tensor:
    """Destination for the packed ReLU mask of a tensor shaped like ``like``: one byte per 8 channels."""
    return torch.empty(*like.shape[:-1], like.shape[-1] // 8, dtype=torch.uint8, device=like.device)


def _check_bits(bits, shape, what: str) -> None:
    want = (*shape[:-1], shape[-1] // 8)
    if bits.dtype != torch.uint8 or tuple(bits.shape) != want or not bits.is_contiguous() or not bits.is_cuda:
        raise ValueError(f"{what}: expected a contiguous uint8 GPU tensor of shape {want}")


def bn_apply(y, scale, shift, residual, relu: bool, out: Optional[torch.Tensor] = None,
             bits_out: Optional[torch.Tensor] = None, colsum_slot: Optional[str] = None):
    """``bits_out``: also receives ``[out > 0]`` packed one byte per 8 channels -- what the ReLU
    backward of this tensor needs, at 1/16 of the tensor's size.  ``colsum_slot``: the per-channel sums of the
    result are left in the accumulator of that name (for ``conv_wgrad_folded(amean_slot=...)`` of the layer that
    consumes this tensor); ignored in deterministic mode."""
    y = _bf16c(y, "bn_apply y")
    C = y.shape[-1]
    out = _out_like(out, y.shape, y.device, "bn_apply")
    if residual is not None:
        residual = _bf16c(residual, "bn_apply residual")
    if bits_out is not None:
        _check_bits(bits_out, y.shape, "bn_apply bits_out")
    if _ABLATE_APPLY and residual is None:
        return out
    acc = _accumulator(y.device, C, "colsum:" + colsum_slot) if (colsum_slot and not _deterministic) else None
    rc = _lib.pdt_bn_apply_ex(_p(y), _p(scale), _p(shift), _p(residual), _p(out), _p(bits_out), y.numel() // C, C,
                              int(bool(relu)), _p(acc), _stream())
    _check(rc, "pdt_bn_apply")
    return out


def bn_apply2(y, scale, shift, y2, scale2, shift2, relu: bool, out: Optional[torch.Tensor] = None,
              bits_out: Optional[torch.Tensor] = None):
    """``relu(y*scale + shift + y2*scale2 + shift2)``: main path and projection shortcut of a block
    normalised, added and activated in one pass."""
    y = _bf16c(y, "bn_apply2 y")
    y2 = _bf16c(y2, "bn_apply2 y2")
    if y.shape != y2.shape:
        raise ValueError("bn_apply2: the two tensors must have one shape")
    C = y.shape[-1]
    out = _out_like(out, y.shape, y.device, "bn_apply2")
    if bits_out is not None:
        _check_bits(bits_out, y.shape, "bn_apply2 bits_out")
    rc = _lib.pdt_bn_apply2(_p(y), _p(scale), _p(shift), _p(y2), _p(scale2), _p(shift2), _p(out), _p(bits_out),
                            y.numel() // C, C, int(bool(relu)), _stream())
    _check(rc, "pdt_bn_apply2")
    return out


batch_chunks = True          # ops take out= destinations, so a batch can be processed in cache-sized slices
fused_bwd_stats = True       # conv_dgrad(bwd_stats=...) + bn_bwd_reduce(prereduced=True)


def _pool_geometry(dout, y, what: str):
    """Shapes of the pooled gradient ``dout`` against the pooling's input ``y`` (3x3 / s2 / p1)."""
    B, H, W, C = y.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    if tuple(dout.shape) != (B, Ho, Wo, C):
        raise ValueError(f"{what}: pooled gradient {tuple(dout.shape)} does not belong to an input {tuple(y.shape)}")
    return B, H, W, C, Ho, Wo


def bn_bwd_reduce(g, y, mean, invstd, scale, shift, mask_src, mask_mode: int, dgamma=None, dbeta=None, comm=None,
                  prereduced: bool = False, pool_idx=None):
    """``comm`` (an ``XgmiAllReduce``): the returned sums are already summed over its ranks (the
    exchange runs inside the kernel that folds the accumulator replicas).  ``prereduced``: the
    accumulator already holds this layer's sums (the data-gradient epilogue that produced ``g``
    put them there), only the fold remains.  ``pool_idx``: ``g`` is the gradient of the max
    pooling that follows this layer's ReLU; the dense gradient is gathered on the fly."""
    g = _bf16c(g, "bn_bwd_reduce g")
    y = _bf16c(y, "bn_bwd_reduce y")
    if mask_mode == MASK_TENSOR:
        mask_src = _bf16c(mask_src, "bn_bwd_reduce mask")
    C = y.shape[-1]
    rows = y.numel() // C
    acc = _accumulator(y.device, C, "bwd")
    if pool_idx is not None and not prereduced:
        if mask_mode != MASK_RECOMPUTE:
            raise ValueError("bn_bwd_reduce: pool_idx goes with the recomputed ReLU mask")
        B, H, W, _, Ho, Wo = _pool_geometry(g, y, "bn_bwd_reduce")
        rc = _lib.pdt_pool_bn_bwd_reduce_accum(_p(g), _p(pool_idx), _p(y), _p(mean), _p(scale), _p(shift), _p(acc),
                                               B, H, W, C, Ho, Wo, _stream())
        _check(rc, "pdt_pool_bn_bwd_reduce_accum")
    elif not prereduced:
        rc = _lib.pdt_bn_bwd_reduce_accum(_p(g), _p(y), _p(mask_src) if mask_mode == MASK_TENSOR else None, _p(mean),
                                          _p(scale), _p(shift), _p(acc), rows, C, mask_mode, _stream())
        _check(rc, "pdt_bn_bwd_reduce_accum")
    sums = torch.empty(2, C, dtype=torch.float32, device=y.device)
    if comm is not None and 2 * C <= 2 * FUSED_MAX_C:
        rc = _lib.pdt_bn_bwd_sums_fused(_comm_handle(comm), _p(acc), _p(invstd), _p(sums), _p(dgamma), _p(dbeta), C,
                                        _stream())
        _check(rc, "pdt_bn_bwd_sums_fused")
        comm.calls += 1
        return sums
    _check(_lib.pdt_bn_bwd_sums(_p(acc), _p(invstd), _p(sums), _p(dgamma), _p(dbeta), C, _stream()), "pdt_bn_bwd_sums")
    if comm is not None:
        comm.all_reduce(sums.view(-1))
    return sums


def bn_bwd_elemt(g, y, mean, invstd, scale, shift, sums, inv_count: float, mask_src, mask_mode: int,
                 out: Optional[torch.Tensor] = None, pool_idx=None):
    g = _bf16c(g, "bn_bwd_elemt g")
    y = _bf16c(y, "bn_bwd_elemt y")
    C = y.shape[-1]
    dx = _out_like(out, y.shape, y.device, "bn_bwd_elemt")
    if pool_idx is not None:
        if mask_mode != MASK_RECOMPUTE:
            raise ValueError("bn_bwd_elemt: pool_idx goes with the recomputed ReLU mask")
        B, H, W, _, Ho, Wo = _pool_geometry(g, y, "bn_bwd_elemt")
        rc = _lib.pdt_pool_bn_bwd_elemt(_p(g), _p(pool_idx), _p(y), _p(mean), _p(invstd), _p(scale), _p(shift),
                                        _p(sums), inv_count, _p(dx), B, H, W, C, Ho, Wo, _stream())
        _check(rc, "pdt_pool_bn_bwd_elemt")
        return dx
    if _ABLATE_ELEMT:
        return dx
    rc = _lib.pdt_bn_bwd_elemt(_p(g), _p(y), _p(mask_src) if mask_mode == MASK_TENSOR else None, _p(mean), _p(invstd),
                               _p(scale), _p(shift), _p(sums), inv_count, _p(dx), y.numel() // C, C, mask_mode,
                               _stream())
    _check(rc, "pdt_bn_bwd_elemt")
    return dx



# --------------------------------------------------------------------------- folded BatchNorm backward
bn_fold = True               # bn_fold_prepare / conv_dgrad_folded / conv_wgrad_folded (csrc/bn_fold.hip)
BN_FOLD_MAX_C = 512          # accumulator start values of the data-gradient kernel live in LDS


class BnFold:
    """What :func:`bn_fold_prepare` hands to the two folded gradient products of one layer."""

    __slots__ = ("wfold", "bias", "coef", "C", "N")

    def __init__(self, wfold, bias, coef, C: int, N: int) -> None:
        self.wfold, self.bias, self.coef, self.C, self.N = wfold, bias, coef, C, N

    def tensors(self):
        return self.wfold, self.bias, self.coef


def bn_fold_prepare(sums, scale, invstd, mean, inv_count: float, w: torch.Tensor, wt: torch.Tensor) -> BnFold:
    """Fold the backward of the BatchNorm behind a 1x1 / stride 1 convolution ``y = a . w^T`` into that
    convolution's two gradient products (math: :func:`..ref.bn_fold_prepare`).  ``w``: ``[N, 1, 1, C]``,
    ``wt``: its data-gradient copy ``[C, 1, 1, N]``; ``sums``: the reduced ``[2, N]`` backward sums."""
    wt = _bf16c(wt, "bn_fold_prepare wt")
    C, N = wt.shape[0], wt.shape[-1]
    if wt.numel() != C * N or C % 64 or N % 64 or C > BN_FOLD_MAX_C:
        raise ValueError(f"bn_fold_prepare: needs a 1x1 filter with channel counts in multiples of 64 and at most "
                         f"{BN_FOLD_MAX_C} input channels, got {tuple(wt.shape)}")
    dev = wt.device
    wfold = torch.empty(C, N + C, dtype=torch.bfloat16, device=dev)
    bias = torch.empty(C, dtype=torch.float32, device=dev)
    coef = torch.empty(3, N, dtype=torch.float32, device=dev)
    rc = _lib.pdt_bnfold_prepare(_p(_f32c(sums, "bn_fold_prepare sums")), _p(scale), _p(invstd), _p(mean), inv_count,
                                 _p(wt), _p(wfold), _p(bias), _p(coef), C, N, _stream())
    _check(rc, "pdt_bnfold_prepare")
    return BnFold(wfold, bias, coef, C, N)


def bn_fold_sums(prod: torch.Tensor, w: torch.Tensor, s0: torch.Tensor, mean, invstd, dgamma=None, dbeta=None):
    """The ``[2, N]`` backward sums of the BatchNorm behind ``y = a . w^T`` from ``s0 = sum g`` and the unscaled
    weight-gradient product ``prod = g^T a`` (``[N, 1, 1, C]`` fp32): ``sum g y = sum_c w[n, c] prod[n, c]``.
    Also ``dbeta += s0``, ``dgamma += sum g xhat`` like :func:`bn_bwd_reduce`."""
    w = _bf16c(w, "bn_fold_sums w")
    _f32c(prod, "bn_fold_sums prod")
    N, C = w.shape[0], w.shape[-1]
    if w.numel() != N * C or prod.numel() != N * C or C % 8:
        raise ValueError("bn_fold_sums: w / prod must be [N, 1, 1, C] with C a multiple of 8")
    sums = torch.empty(2, N, dtype=torch.float32, device=w.device)
    rc = _lib.pdt_bnfold_sums(_p(prod), _p(w), _p(s0), _p(mean), _p(invstd), _p(sums), _p(dgamma), _p(dbeta), C, N,
                              _stream())
    _check(rc, "pdt_bnfold_sums")
    return sums


def conv_dgrad_folded(g: torch.Tensor, a_in: torch.Tensor, fold: BnFold, bwd_stats=None,
                      out: Optional[torch.Tensor] = None, tile: int = 0) -> torch.Tensor:
    """Data gradient of ``conv1x1 -> BatchNorm`` w.r.t. the convolution input ``a_in`` from ``g``, the gradient
    w.r.t. the BatchNorm OUTPUT: ``g . (A o w) + a_in . S + bias`` in one kernel (``bwd_stats`` as in
    :func:`conv_dgrad`)."""
    g = _bf16c(g, "conv_dgrad_folded g")
    a_in = _bf16c(a_in, "conv_dgrad_folded a_in")
    B, H, W, N = g.shape
    C = a_in.shape[-1]
    if tuple(a_in.shape[:3]) != (B, H, W) or (C, N) != (fold.C, fold.N):
        raise ValueError("conv_dgrad_folded: tensor shapes do not belong to the folded layer")
    dx = _out_like(out, (B, H, W, C), g.device, "conv_dgrad_folded")
    acc = by = bmean = bscale = bshift = None
    if bwd_stats is not None:
        by, bmean, bscale, bshift, bmode = bwd_stats
        by = _bf16c(by, "conv_dgrad_folded bwd_stats y")
        if tuple(by.shape) != (B, H, W, C) or bmode not in (MASK_NONE, MASK_RECOMPUTE):
            raise ValueError("conv_dgrad_folded: bwd_stats tensor/mode does not match the gradient being produced")
        if bmode != MASK_RECOMPUTE:
            bscale = bshift = None
        acc = _accumulator(g.device, C, "bwd")
    taps = _one_tap()
    for b0, b1 in [(0, B)]:           # (one launch whatever the size: the kernel rebases its descriptors per tile)
        M = (b1 - b0) * H * W
        rc = _lib.pdt_conv_dma_ex(_p(g[b0:b1]), _p(fold.wfold), _p(dx[b0:b1]), None, 0, None, _p(acc), None,
                                  _p(by[b0:b1]) if by is not None else None, _p(bmean), _p(bscale), _p(bshift), M, H, W, N,
                                  N, H, W, 1, H, W, 1, 0, 0, C, N + C, 1, taps.offy, taps.offx, taps.widx, tile,
                                  _p(a_in[b0:b1]), C, C, _p(fold.bias), 0, None, 0, _stream())
        _check(rc, "pdt_conv_dma_ex")
    return dx


def gram_and_means(a: torch.Tensor, colsum_slot: Optional[str] = None):
    """``(G, mom)`` of an NHWC tensor seen as ``[rows, C]``: the Gram matrix ``a^T a`` (``[C, C]`` fp32, an ordinary
    weight-gradient launch with ``a`` on both sides) and ``mom[0]`` = the column means (``mom[1]``: scratch).
    ``colsum_slot``: the column sums were left in that accumulator by ``bn_apply(colsum_slot=...)``."""
    a = _bf16c(a, "gram_and_means")
    B, H, W, C = a.shape
    rows = B * H * W
    dev = a.device
    if colsum_slot and not _deterministic:
        acc = _accumulator(dev, C, "colsum:" + colsum_slot)
    else:
        acc = _accumulator(dev, C, "fold", per_stream=True)
        _check(_lib.pdt_bn_stats_accum(_p(a), None, _p(acc), rows, C, _stream()), "pdt_bn_stats_accum")
    mom = torch.empty(2, C, dtype=torch.float32, device=dev)
    _check(_lib.pdt_bn_moments(_p(acc), _p(mom), C, 1.0 / rows, _stream()), "pdt_bn_moments")
    gram = torch.empty(C, C, dtype=torch.float32, device=dev)
    _check(_lib.pdt_zero_f32(_p(gram), C * C, _stream()), "pdt_zero_f32")
    _wgrad(a, a, gram, rows, H, W, C, C, H, W, 1, C, C, C, _one_tap())
    return gram, mom


def bn_stats_from_gram(w: torch.Tensor, gram: torch.Tensor, mom: torch.Tensor, rows: int,
                       stat_shift: Optional[torch.Tensor], acc_slot: str = "stats") -> "PendingStats":
    """Batch statistics of ``y = a . w^T`` (1x1 convolution, ``w``: ``[N, 1, 1, C]``) WITHOUT ``y``: from the Gram
    matrix and column means of ``a`` (:func:`gram_and_means`).  Returns what ``conv_fprop(defer_stats=True)`` would:
    the shifted sums in the accumulator, to be consumed by :func:`bn_finalize_fused` / :func:`stats_to_moments`."""
    w = _bf16c(w, "bn_stats_from_gram w")
    N, C = w.shape[0], w.shape[-1]
    if w.numel() != N * C or C % 32 or N % 32 or tuple(gram.shape) != (C, C):
        raise ValueError("bn_stats_from_gram: needs a 1x1 filter with channel counts in multiples of 32")
    acc = _accumulator(w.device, N, acc_slot)
    rc = _lib.pdt_bnfold_fwd_stats(_p(w), _p(_f32c(gram, "gram")), _p(mom), float(rows), _p(stat_shift), _p(acc), C, N,
                                   _stream())
    _check(rc, "pdt_bnfold_fwd_stats")
    return PendingStats(acc, N, rows)


def bn_fuse_fprop_weights(w, scale, shift, w2=None, scale2=None, shift2=None):
    """``([scale o w | scale2 o w2], shift + shift2)``: the weights and accumulator start values with which
    ``conv(a, .) [+ conv(x, .)]`` IS ``bn(conv(a, w)) [+ bn2(conv(x, w2))]`` (``w2`` optional; both 1x1)."""
    w = _bf16c(w, "bn_fuse_fprop_weights w")
    N, C = w.shape[0], w.shape[-1]
    C2 = 0
    if w2 is not None:
        w2 = _bf16c(w2, "bn_fuse_fprop_weights w2")
        C2 = w2.shape[-1]
        if w2.shape[0] != N or w2.numel() != N * C2:
            raise ValueError("bn_fuse_fprop_weights: the two filters must have the same output channels")
    if w.numel() != N * C:
        raise ValueError("bn_fuse_fprop_weights: 1x1 filters only")
    wf = torch.empty(N, C + C2, dtype=torch.bfloat16, device=w.device)
    sh = torch.empty(N, dtype=torch.float32, device=w.device)
    rc = _lib.pdt_bnfold_scale_weights(_p(w), _p(scale), _p(shift), _p(w2), _p(scale2), _p(shift2), _p(wf), _p(sh), C, C2,
                                       N, _stream())
    _check(rc, "pdt_bnfold_scale_weights")
    return wf, sh


def conv_fprop_fused_tail(a: torch.Tensor, wf: torch.Tensor, shift: torch.Tensor, addend: Optional[torch.Tensor] = None,
                          in2: Optional[torch.Tensor] = None, bits_out: Optional[torch.Tensor] = None,
                          out: Optional[torch.Tensor] = None, tile: int = 0) -> torch.Tensor:
    """``relu(a . wf[:, :C]^T + in2 . wf[:, C:]^T + shift + addend)`` in one kernel: with the weights of
    :func:`bn_fuse_fprop_weights` the whole tail of a residual block (1x1 convolution, BatchNorm, projection or
    identity shortcut, ReLU, packed ReLU mask) without the convolution outputs in memory."""
    a = _bf16c(a, "conv_fprop_fused_tail a")
    wf = _bf16c(wf, "conv_fprop_fused_tail wf")
    B, H, W, C = a.shape
    N = wf.shape[0]
    C2 = wf.shape[1] - C
    if (in2 is None) != (C2 == 0) or C % 64 or N % 64 or C2 % 64:
        raise ValueError("conv_fprop_fused_tail: weight columns do not match the operands")
    if in2 is not None:
        in2 = _bf16c(in2, "conv_fprop_fused_tail in2")
        if tuple(in2.shape) != (B, H, W, C2):
            raise ValueError("conv_fprop_fused_tail: the second operand must have the first one's pixels")
    y = _out_like(out, (B, H, W, N), a.device, "conv_fprop_fused_tail")
    if addend is not None:
        addend = _bf16c(addend, "conv_fprop_fused_tail addend")
        if tuple(addend.shape) != (B, H, W, N):
            raise ValueError("conv_fprop_fused_tail: addend shape")
    if bits_out is not None:
        _check_bits(bits_out, (B, H, W, N), "conv_fprop_fused_tail bits_out")
    taps = _one_tap()
    for b0, b1 in [(0, B)]:           # (one launch whatever the size: the kernel rebases its descriptors per tile)
        M = (b1 - b0) * H * W
        rc = _lib.pdt_conv_dma_ex(_p(a[b0:b1]), _p(wf), _p(y[b0:b1]), _p(_part(addend, b0, b1)), 0, None, None, None, None,
                                  None, None, None, M, H, W, C, C, H, W, 1, H, W, 1, 0, 0, N, C + C2, 1, taps.offy,
                                  taps.offx, taps.widx, tile, _p(_part(in2, b0, b1)), C2, C2, _p(shift), 1,
                                  _p(_part(bits_out, b0, b1)), 0, _stream())
        _check(rc, "pdt_conv_dma_ex")
    return y


def channel_sums(g: torch.Tensor) -> torch.Tensor:
    """``[C]`` per-channel sums of an NHWC tensor."""
    g = _bf16c(g, "channel_sums")
    C = g.shape[-1]
    rows = g.numel() // C
    acc = _accumulator(g.device, C, "fold", per_stream=True)
    _check(_lib.pdt_bn_stats_accum(_p(g), None, _p(acc), rows, C, _stream()), "pdt_bn_stats_accum")
    out = torch.empty(2, C, dtype=torch.float32, device=g.device)
    _check(_lib.pdt_bn_moments(_p(acc), _p(out), C, 1.0, _stream()), "pdt_bn_moments")
    return out[0]


def conv_wgrad_folded(a_in: torch.Tensor, g: torch.Tensor, fold: BnFold, w: torch.Tensor, dw: torch.Tensor,
                      prod: Optional[torch.Tensor] = None, amean_slot: Optional[str] = None, gram=None,
                      mom=None) -> None:
    """``dw += (A o g + B o y + K)^T . a_in`` without the tensor in parentheses and without ``y``: the scaled
    product ``A o (g^T a)``, the Gram matrix ``a^T a`` (an ordinary weight-gradient launch with ``a`` on both
    sides), the column means of ``a``, and the small correction kernel.  ``prod``: the unscaled ``g^T a`` is
    already there (see :func:`bn_fold_sums`); the correction kernel scales and adds it.  ``amean_slot``: the
    column sums of ``a_in`` were left in that accumulator by ``bn_apply(colsum_slot=...)``.  ``gram`` / ``mom``:
    the Gram matrix and column means exist already (:func:`gram_and_means`, e.g. from the forward pass)."""
    if _ABLATE_WGRAD:
        return
    a_in = _bf16c(a_in, "conv_wgrad_folded a_in")
    g = _bf16c(g, "conv_wgrad_folded g")
    w = _bf16c(w, "conv_wgrad_folded w")
    _f32c(dw, "conv_wgrad_folded dw")
    B, H, W, C = a_in.shape
    N = g.shape[-1]
    if (C, N) != (fold.C, fold.N) or w.numel() != N * C or dw.numel() != N * C:
        raise ValueError("conv_wgrad_folded: tensor shapes do not belong to the folded layer")
    rows = B * H * W
    dev = a_in.device
    taps = _one_tap()
    if gram is None or mom is None:
        gram, mom = gram_and_means(a_in, amean_slot)
    if prod is None:
        # A o (g^T a) straight into the gradient buffer, then the correction
        _wgrad(g, a_in, dw, rows, H, W, C, C, H, W, 1, N, N, C, taps, row_scale=fold.coef[0])
    else:
        _f32c(prod, "conv_wgrad_folded prod")
        if prod.numel() != N * C:
            raise ValueError("conv_wgrad_folded: prod must be [N, 1, 1, C]")
    rc = _lib.pdt_bnfold_wgrad(_p(dw), _p(w), _p(gram), _p(mom), float(rows), _p(fold.coef), _p(prod), C, N, _stream())
    _check(rc, "pdt_bnfold_wgrad")

# --------------------------------------------------------------------------- pooling
def maxpool_fwd(x):
    x = _bf16c(x, "maxpool_fwd")
    B, H, W, C = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16, device=x.device)
    idx = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=x.device)
    _check(_lib.pdt_maxpool_fwd(_p(x), _p(out), _p(idx), B, H, W, C, Ho, Wo, _stream()), "pdt_maxpool_fwd")
    return out, idx


pool_winners = True          # bn_relu_maxpool_fwd(winners=True)


def bn_relu_maxpool_fwd(y, scale, shift, winners: bool = False):
    """``maxpool3x3s2(relu(y*scale + shift))`` without materialising the normalised tensor.  ``winners``: also
    returns ``y`` at every window's winner (``[B, Ho, Wo, C]``) and attaches the packed sign mask of the pooled
    result to it -- what the kernel producing the pooled gradient needs to apply this ReLU's backward and to add up
    the BatchNorm-backward sums of ``y``'s layer on the way (``conv_dgrad(relu_bits=, bwd_stats=)``)."""
    y = _bf16c(y, "bn_relu_maxpool_fwd")
    B, H, W, C = y.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16, device=y.device)
    idx = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=y.device)
    if winners:
        ywin = torch.empty_like(out)
        bits = new_relu_bits(out)
        rc = _lib.pdt_bn_relu_maxpool_fwd_winners(_p(y), _p(scale), _p(shift), _p(out), _p(idx), _p(ywin), _p(bits),
                                                  B, H, W, C, Ho, Wo, _stream())
        _check(rc, "pdt_bn_relu_maxpool_fwd_winners")
        out._pdt_relu_bits = bits
        return out, idx, ywin
    rc = _lib.pdt_bn_relu_maxpool_fwd(_p(y), _p(scale), _p(shift), _p(out), _p(idx), B, H, W, C, Ho, Wo, _stream())
    _check(rc, "pdt_bn_relu_maxpool_fwd")
    return out, idx


def maxpool_bwd(dout, idx, in_shape):
    dout = _bf16c(dout, "maxpool_bwd")
    B, H, W, C = in_shape
    _, Ho, Wo, _ = dout.shape
    dx = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dout.device)
    _check(_lib.pdt_maxpool_bwd(_p(dout), _p(idx), _p(dx), B, H, W, C, Ho, Wo, _stream()), "pdt_maxpool_bwd")
    return dx


def avgpool_fwd(x):
    x = _bf16c(x, "avgpool_fwd")
    B, H, W, C = x.shape
    out = torch.empty(B, C, dtype=torch.bfloat16, device=x.device)
    _check(_lib.pdt_avgpool_fwd(_p(x), _p(out), B, H * W, C, _stream()), "pdt_avgpool_fwd")
    return out


def avgpool_bwd(dp, H: int, W: int, relu_mask=None):
    dp = _bf16c(dp, "avgpool_bwd")
    B, C = dp.shape
    dx = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dp.device)
    if relu_mask is not None:
        relu_mask = _bf16c(relu_mask, "avgpool_bwd relu_mask")
    _check(_lib.pdt_avgpool_bwd(_p(dp), _p(relu_mask), _p(dx), B, H * W, C, _stream()), "pdt_avgpool_bwd")
    return dx


# --------------------------------------------------------------------------- classifier + loss
_ONE_TAP = None


def _one_tap() -> _Taps:
    global _ONE_TAP
    if _ONE_TAP is None:
        _ONE_TAP = _Taps([0], [0], [0])
    return _ONE_TAP


def _npad(n: int) -> int:
    return (n + 63) // 64 * 64


def prepare_fc(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``w[N, K]`` -> zero-padded transposed copy ``wt[K, Npad]`` for the classifier's data gradient
    (``out``: refresh a previous result in place)."""
    w = _bf16c(w, "prepare_fc")
    N, K = w.shape
    wt = out if out is not None else torch.empty(K, _npad(N), dtype=torch.bfloat16, device=w.device)
    _check(_lib.pdt_transpose_w(_p(w), _p(wt), N, 1, K, _npad(N), _stream()), "pdt_transpose_w")
    return wt


def linear_fwd(a, w, bias):
    a = _bf16c(a, "linear_fwd a")
    w = _bf16c(w, "linear_fwd w")
    B, K = a.shape
    N = w.shape[0]
    ld = _npad(N)
    out = torch.empty(B, ld, dtype=torch.bfloat16, device=a.device)
    _igemm(a, w, out, B, 1, 1, K, K, 1, 1, 1, 1, 1, 1, 0, 0, N, ld, K, _one_tap(), bias=bias)
    return out[:, :N]


def linear_bwd(a, dlogits, w, wt, dw, dbias):
    a = _bf16c(a, "linear_bwd a")
    B, K = a.shape
    N = w.shape[0]
    ld = dlogits.stride(0)
    if dlogits.dtype != torch.bfloat16 or dlogits.stride(1) != 1 or ld % 64 != 0 or wt is None or wt.shape[1] != ld:
        raise ValueError("linear_bwd: dlogits must be the padded bf16 buffer produced by softmax_ce")
    da = torch.empty(B, K, dtype=torch.bfloat16, device=a.device)
    _igemm(dlogits, wt, da, B, 1, 1, ld, ld, 1, 1, 1, 1, 1, 1, 0, 0, K, K, ld, _one_tap())
    _wgrad(dlogits, a, dw, B, 1, 1, K, K, 1, 1, 1, N, ld, K, _one_tap())
    _check(_lib.pdt_colsum_add(_p(dlogits), _p(dbias), B, N, ld, _stream()), "pdt_colsum_add")
    return da


def softmax_ce(logits, labels, grad_scale: float = 1.0):
    if logits.dtype != torch.bfloat16 or logits.stride(1) != 1:
        raise TypeError("softmax_ce: logits must be bf16 with unit column stride")
    B, N = logits.shape
    ld_out = _npad(N)
    dlog = torch.empty(B, ld_out, dtype=torch.bfloat16, device=logits.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    labels = labels if labels.dtype == torch.int64 else labels.long()
    row_loss = torch.empty(B, dtype=torch.float32, device=logits.device) if _deterministic else None
    rc = _lib.pdt_softmax_ce(_p(logits), _p(labels.contiguous()), _p(dlog), _p(loss), _p(row_loss), B, N,
                             logits.stride(0), ld_out, grad_scale, _stream())
    _check(rc, "pdt_softmax_ce")
    return loss[0], dlog[:, :N]


def ce_metrics(logits, labels):
    if logits.dtype != torch.bfloat16 or logits.stride(1) != 1:
        raise TypeError("ce_metrics: logits must be bf16 with unit column stride")
    B, N = logits.shape
    out = torch.empty(4, dtype=torch.float32, device=logits.device)
    labels = labels if labels.dtype == torch.int64 else labels.long()
    rc = _lib.pdt_ce_metrics(_p(logits), _p(labels.contiguous()), _p(out), B, N, logits.stride(0), _stream())
    _check(rc, "pdt_ce_metrics")
    return out


# --------------------------------------------------------------------------- optimizer
def sgd_step(p, g, m, pc, lr: float, momentum: float, weight_decay: float, grad_scale: float, nesterov: bool = False,
             dampening: float = 0.0, first_step: bool = False) -> None:
    n = p.numel()
    rc = _lib.pdt_sgd_step(_p(p), _p(g), _p(m), _p(pc), n, lr, momentum, weight_decay, grad_scale, dampening,
                           int(bool(nesterov)), int(bool(first_step)), _stream())
    _check(rc, "pdt_sgd_step")
