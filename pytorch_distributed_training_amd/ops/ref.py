"""Pure-PyTorch fp32 reference implementation of every engine op.

This backend has two jobs: it is the numerical oracle the HIP kernels are tested
against (same signatures, same tensor layouts), and it is what the engine runs on when
there is no GPU (CPU / gloo plumbing tests).  Every function up-casts its inputs to fp32,
computes with stock ``torch`` ops and rounds the result to the activation dtype.

Layouts: activations are NHWC ``[B, H, W, C]`` contiguous; convolution weights are OHWI
``[Cout, R, S, Cin]``; per-channel vectors are fp32.

The ops correspond to the kernels PyTorch dispatches implicitly for the upstream trainer's
``model(img)`` / ``loss.backward()`` / ``optimizer.step()`` (reference
``train_distributed.py:274-277``): convolution forward / data-gradient / weight-gradient,
(Sync)BatchNorm statistics / apply / backward-reduce / backward-elementwise
(``torch/nn/modules/_functions.py:39-179``), max/avg pooling, the classifier GEMM,
softmax cross-entropy, SGD with momentum (``torch/optim/sgd.py:322-367``) and top-k.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

name = "ref"
# conv_dgrad reads the forward weights ``w`` (an update of them must wait until backward is over)
dgrad_reads_forward_weights = True

MASK_NONE = 0       # dy = g
MASK_TENSOR = 1     # dy = g * (mask_src > 0)
MASK_RECOMPUTE = 2  # dy = g * (y * scale + shift > 0)


def _nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1).contiguous()


def _w_oihw(w: torch.Tensor) -> torch.Tensor:
    return w.permute(0, 3, 1, 2)


def channel_moments(y32: torch.Tensor, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[2, C]``: per-channel SHIFTED moments ``E[y - s]`` and ``E[(y - s)^2]``.

    ``s`` is a per-channel shift (the BatchNorm running mean; zero when ``None``).  Shifted
    moments are what ranks exchange for SyncBatchNorm: running means are identical on all
    ranks and per-rank counts are equal, so the global moments are the plain average of the
    local ones -- ONE sum all-reduce of ``2C`` floats -- and the variance
    ``E[(y-s)^2] - E[y-s]^2`` only suffers cancellation if the batch mean is many standard
    deviations away from the running mean.
    """
    flat = y32.reshape(-1, y32.shape[-1])
    if shift is not None:
        flat = flat - shift
    var, mean = torch.var_mean(flat, dim=0, unbiased=False)
    return torch.stack([mean, var + mean * mean])


# ---------------------------------------------------------------------------- input
def input_transform(img: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Loader batch -> stem input.

    ``img`` is either float NCHW, already normalised (the upstream loader contract,
    reference ``train_distributed.py:272``), or uint8 NHWC raw pixels, normalised here
    with ``(x/255 - mean) / std``.  Returns NHWC ``[B, H, W, 3]`` in ``dtype``.
    """
    if img.dtype == torch.uint8:
        x = img.float().div_(255.0)
        x = (x - mean.view(1, 1, 1, 3)) / std.view(1, 1, 1, 3)
        return x.to(dtype).contiguous()
    return _nhwc(img.float()).to(dtype)


def crop_resize_flip(src: torch.Tensor, boxes: torch.Tensor, out_size: int) -> torch.Tensor:
    """``src [B, S, S, 3]`` uint8, ``boxes [B, 5]`` int32 ``(x, y, w, h, flip)`` -> ``[B, out, out, 3]`` uint8:
    bilinear resize (half-pixel centres, no antialiasing) of each box, then the horizontal flip."""
    outs = []
    for img, (x, y, w, h, flip) in zip(src, boxes.tolist()):
        crop = img[y:y + h, x:x + w].permute(2, 0, 1).unsqueeze(0).float()
        r = F.interpolate(crop, size=(out_size, out_size), mode="bilinear", align_corners=False)
        r = (r + 0.5).floor().clamp_(0, 255).to(torch.uint8)[0].permute(1, 2, 0)
        outs.append(r.flip(1) if flip else r)
    return torch.stack(outs).contiguous()


def prepare_stem_weight(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Compute-side stem weights from the ``[64, 7, 7, 3]`` compute-dtype view (identity here)."""
    return w


def stem_fprop(x0: torch.Tensor, w: torch.Tensor, stat_shift: Optional[torch.Tensor] = None):
    return conv_fprop(x0, w, 7, 7, 2, 3, stat_shift)


def stem_wgrad(x0: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor) -> None:
    conv_wgrad(x0, dy, 7, 7, 2, 3, dw)


# ---------------------------------------------------------------------------- convolution
def conv_fprop(x: torch.Tensor, w: torch.Tensor, R: int, S: int, stride: int, pad: int,
               stat_shift: Optional[torch.Tensor] = None, **_) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """``y = conv(x, w)``; when ``stat_shift`` is given also the shifted per-channel moments of
    the result (see :func:`channel_moments`).  Here they are taken from the fp32 result; a
    backend may take them from the rounded ``y`` it stores instead (the HIP kernel does: that is
    what a separate statistics pass over the output tensor would see)."""
    y32 = F.conv2d(_nchw(x.float()), _w_oihw(w.float()), None, stride, pad)
    y32 = _nhwc(y32)
    stats = channel_moments(y32, stat_shift) if stat_shift is not None else None
    return y32.to(x.dtype), stats


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, wt: Optional[torch.Tensor], in_shape, R: int, S: int,
               stride: int, pad: int, addend: Optional[torch.Tensor] = None,
               relu_mask: Optional[torch.Tensor] = None, relu_bits: Optional[torch.Tensor] = None,
               addend_half: bool = False, out_sums: bool = False, **_) -> torch.Tensor:
    """``dx = (conv_transpose(dy, w) [+ addend]) [* (relu_mask > 0)]`` ; ``in_shape = (B, Hi, Wi, Ci)``.
    ``relu_bits``: the same mask in packed form (see :func:`bn_apply`).

    ``addend`` accumulates the other gradient path of a residual block; ``relu_mask`` (the
    post-ReLU tensor this gradient belongs to) applies that ReLU's backward right here, so the
    consumers of ``dx`` -- the BatchNorm backward passes of the previous block -- need no mask.
    The sum is rounded to the activation dtype before the mask, like a separate add would."""
    B, Hi, Wi, Ci = in_shape
    dx = torch.nn.grad.conv2d_input((B, Ci, Hi, Wi), _w_oihw(w.float()), _nchw(dy.float()), stride, pad)
    dx = _nhwc(dx)
    dx = dx.to(dy.dtype)
    if addend is not None:
        if addend_half:          # given at the output pixels with two even coordinates only, zero elsewhere
            full = torch.zeros_like(dx)
            full[:, ::2, ::2] = addend
            addend = full
        dx = (dx.float() + addend.float()).to(dy.dtype)
    if relu_bits is not None:
        dx = dx * _unpack_bits(relu_bits)
    elif relu_mask is not None:
        dx = dx * (relu_mask > 0)
    if out_sums:         # per-channel sums of the stored result, collected by grad_sums
        _grad_sums[Ci] = dx.float().reshape(-1, Ci).sum(0)
    return dx


_grad_sums = {}


def grad_sums(C: int, device=None) -> torch.Tensor:
    """``[C]`` sums left by the last ``conv_dgrad(out_sums=True)`` with ``C`` output channels."""
    return _grad_sums.pop(C)


def colsum_take(slot: str, C: int, device=None):
    """(fused backends: the sums ``bn_apply(colsum_slot=slot)`` left behind; nothing is kept here)"""
    return None


def subsample2(x: torch.Tensor) -> torch.Tensor:
    """``x[:, ::2, ::2]`` of an NHWC tensor, dense: the pixels a stride-2 1x1 convolution reads."""
    return x[:, ::2, ::2].contiguous()


def zeros_f32(shape, device) -> torch.Tensor:
    return torch.zeros(shape, dtype=torch.float32, device=device)


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, R: int, S: int, stride: int, pad: int, dw: torch.Tensor,
               **_) -> None:
    """``dw[Cout, R, S, Cin] += x (*) dy`` (fp32 accumulate into the flat gradient buffer)."""
    Co, Ci = dy.shape[-1], x.shape[-1]
    g = torch.nn.grad.conv2d_weight(_nchw(x.float()), (Co, Ci, R, S), _nchw(dy.float()), stride, pad)
    dw.add_(g.permute(0, 2, 3, 1))


def transpose_weights(w: torch.Tensor, wt: Optional[torch.Tensor]) -> None:
    """Fill the data-gradient weight copy ``wt[Cin, R, S, Cout]`` (unused by this backend)."""
    if wt is not None:
        wt.copy_(w.permute(3, 1, 2, 0))


# ---------------------------------------------------------------------------- batch norm
def bn_stats(y: torch.Tensor, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Stand-alone shifted per-channel moments of an NHWC tensor."""
    return channel_moments(y.float(), shift)


def bn_finalize(moments: torch.Tensor, inv_world: float, gamma: torch.Tensor, beta: torch.Tensor,
                running_mean: torch.Tensor, running_var: torch.Tensor, momentum: float, eps: float, count: int):
    """Summed moments -> ``(scale, shift, mean, invstd)``; updates running stats in place.

    ``moments`` is the (all-reduced) SUM over ranks of the local moments shifted by
    ``running_mean`` (its value BEFORE this update) and ``inv_world = 1 / world_size`` turns
    it into the global moments.  ``scale = gamma * invstd``, ``shift = beta - mean * scale``.
    The running variance uses the unbiased estimate with the GLOBAL element count
    ``count``, as ``SyncBatchNorm`` does (``torch/nn/modules/_functions.py:106-115``).
    """
    m0 = moments[0] * inv_world
    mean = running_mean + m0
    var = (moments[1] * inv_world - m0 * m0).clamp_min(0.0)
    invstd = torch.rsqrt(var + eps)
    scale = gamma * invstd
    shift = beta - mean * scale
    unbiased = var * (count / max(count - 1, 1))
    running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
    running_var.mul_(1.0 - momentum).add_(unbiased, alpha=momentum)
    return scale, shift, mean, invstd


def bn_eval_affine(gamma, beta, running_mean, running_var, eps: float):
    invstd = torch.rsqrt(running_var + eps)
    scale = gamma * invstd
    return scale, beta - running_mean * scale


relu_bits = True             # bn_apply/bn_apply2(bits_out=...) and conv_dgrad(relu_bits=...)
half_addend = True           # conv_dgrad(addend_half=True)
_BIT_WEIGHTS = (1, 2, 4, 8, 16, 32, 64, 128)


def new_relu_bits(like: torch.Tensor) -> torch.Tensor:
    """Destination for the packed ReLU mask of a tensor shaped like ``like``: one byte per 8 channels."""
    return torch.empty(*like.shape[:-1], like.shape[-1] // 8, dtype=torch.uint8, device=like.device)


def _pack_bits(out: torch.Tensor, bits_out: torch.Tensor) -> None:
    w = torch.tensor(_BIT_WEIGHTS, dtype=torch.int32, device=out.device)
    on = (out > 0).to(torch.int32).reshape(*out.shape[:-1], out.shape[-1] // 8, 8)
    bits_out.copy_((on * w).sum(-1).to(torch.uint8))


def _unpack_bits(bits: torch.Tensor) -> torch.Tensor:
    w = torch.tensor(_BIT_WEIGHTS, dtype=torch.int32, device=bits.device)
    on = (bits.to(torch.int32).unsqueeze(-1) & w) != 0
    return on.reshape(*bits.shape[:-1], bits.shape[-1] * 8)


def bn_apply(y: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, residual: Optional[torch.Tensor],
             relu: bool, bits_out: Optional[torch.Tensor] = None, colsum_slot: Optional[str] = None) -> torch.Tensor:
    """``bits_out``: also receives ``[out > 0]``, bit k of byte j = channel ``8j + k``.  ``colsum_slot``: a fused
    backend leaves the per-channel sums of the result under that name for :func:`conv_wgrad_folded`; this backend
    recomputes them there."""
    out = y.float() * scale + shift
    if residual is not None:
        out = out + residual.float()
    if relu:
        out = torch.relu(out)
    out = out.to(y.dtype)
    if bits_out is not None:
        _pack_bits(out, bits_out)
    return out


def _masked(g, y32, scale, shift, mask_src, mask_mode):
    dy = g.float()
    if mask_mode == MASK_TENSOR:
        dy = dy * (mask_src.float() > 0)
    elif mask_mode == MASK_RECOMPUTE:
        dy = dy * ((y32 * scale + shift) > 0)
    return dy


def bn_apply2(y, scale, shift, y2, scale2, shift2, relu: bool, bits_out: Optional[torch.Tensor] = None):
    """``relu(y*scale + shift + y2*scale2 + shift2)`` -- a block's main path and its projection
    shortcut normalised, added and activated in one pass (one rounding to the compute dtype)."""
    out = y.float() * scale + shift + y2.float() * scale2 + shift2
    if relu:
        out = torch.relu(out)
    out = out.to(y.dtype)
    if bits_out is not None:
        _pack_bits(out, bits_out)
    return out


def bn_bwd_reduce(g, y, mean, invstd, scale, shift, mask_src, mask_mode: int,
                  dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                  pool_idx=None) -> torch.Tensor:
    """``[2, C]``: local ``sum(dy)`` and ``sum(dy * xhat)`` with ``xhat = (y - mean) * invstd``.

    When given, ``dbeta += sum(dy)`` and ``dgamma += sum(dy * xhat)`` (the LOCAL sums are the
    parameter gradients; the bucketed all-reduce averages them over ranks later).
    ``pool_idx``: ``g`` is the gradient of the 3x3/s2 max pooling that follows this layer's ReLU
    (with the pooling's index tensor), not of the activation itself.
    """
    if pool_idx is not None:
        g = maxpool_bwd(g, pool_idx, tuple(y.shape))
    y32 = y.float()
    dy = _masked(g, y32, scale, shift, mask_src, mask_mode)
    xhat = (y32 - mean) * invstd
    C = y.shape[-1]
    sums = torch.stack([dy.reshape(-1, C).sum(0), (dy * xhat).reshape(-1, C).sum(0)])
    if dbeta is not None:
        dbeta.add_(sums[0])
    if dgamma is not None:
        dgamma.add_(sums[1])
    return sums


def bn_bwd_elemt(g, y, mean, invstd, scale, shift, sums, inv_count: float, mask_src, mask_mode: int,
                 pool_idx=None) -> torch.Tensor:
    """``dx = (dy - s0 - xhat * s1) * scale`` with ``[s0, s1] = sums * inv_count``.

    ``sums`` are the (all-reduced) global sums from :func:`bn_bwd_reduce` and ``inv_count`` is
    one over the GLOBAL number of elements per channel
    (``torch/nn/modules/_functions.py:170-179``).  ``pool_idx``: as in :func:`bn_bwd_reduce`.
    """
    if pool_idx is not None:
        g = maxpool_bwd(g, pool_idx, tuple(y.shape))
    y32 = y.float()
    dy = _masked(g, y32, scale, shift, mask_src, mask_mode)
    xhat = (y32 - mean) * invstd
    dx = (dy - sums[0] * inv_count - xhat * (sums[1] * inv_count)) * scale
    return dx.to(y.dtype)



# ---------------------------------------------------------------------------- folded BatchNorm backward
bn_fold = True
BN_FOLD_MAX_C = 512


class BnFold:
    __slots__ = ("wfold", "bias", "coef", "C", "N")

    def __init__(self, wfold, bias, coef, C: int, N: int) -> None:
        self.wfold, self.bias, self.coef, self.C, self.N = wfold, bias, coef, C, N

    def tensors(self):
        return self.wfold, self.bias, self.coef


def bn_fold_prepare(sums, scale, invstd, mean, inv_count: float, w: torch.Tensor, wt=None) -> BnFold:
    """BatchNorm backward folded into the 1x1 / stride 1 convolution ``y = a . w^T`` in front of it.

    With the reduced sums of :func:`bn_bwd_reduce`, the gradient w.r.t. the conv output is, per channel n,
    ``d = A g + B y + K`` with ``A = scale``, ``B = -scale * invstd * sums[1] * inv_count``,
    ``K = -scale * sums[0] * inv_count - B * mean`` (:func:`bn_bwd_elemt` written out).  Because ``y`` is a
    product with ``w`` itself, both gradient products of the convolution can be formed from ``g`` and ``a``
    alone -- neither ``d`` nor ``y`` is touched:

    * data gradient ``d . w = g . (A o w) + a . S + bias`` with ``S = w^T diag(B) w`` (``[C, C]``) and
      ``bias = K . w``: :func:`conv_dgrad_folded`;
    * weight gradient ``d^T a = A o (g^T a) + B o (w . (a^T a)) + K (x) colsum(a)``: :func:`conv_wgrad_folded`.

    Returns the folded data-gradient weights ``[C, N + C]`` (``[(A o w)^T | S]``, compute dtype), ``bias [C]``
    and ``coef = [A, B, K]`` (``[3, N]``).  ``y`` here is the exact product; a backend's stored ``y`` is that
    product rounded to the compute dtype, so results agree with :func:`bn_bwd_elemt` + :func:`conv_dgrad` /
    :func:`conv_wgrad` up to that rounding (and are free of the rounding of ``d``).
    """
    N, C = w.shape[0], w.shape[-1]
    w2 = w.reshape(N, C).float()
    k0, k1 = sums[0] * inv_count, sums[1] * inv_count
    A = scale.float()
    Bc = -A * k1 * invstd
    K = -A * k0 - Bc * mean
    S = w2.t() @ (Bc[:, None] * w2)
    wfold = torch.cat([(w2 * A[:, None]).t(), S], dim=1).to(w.dtype).contiguous()
    return BnFold(wfold, K @ w2, torch.stack([A, Bc, K]).contiguous(), C, N)


def bn_fold_sums(prod: torch.Tensor, w: torch.Tensor, s0: torch.Tensor, mean, invstd, dgamma=None, dbeta=None):
    """``[s0, s1]`` with ``s1 = sum g * xhat = invstd * (sum_c w[n, c] prod[n, c] - mean * s0)`` from the unscaled
    weight-gradient product ``prod = g^T a`` -- because ``y = a . w^T``, ``sum_m g y`` needs neither ``g`` nor
    ``y`` again.  ``dbeta += s0``, ``dgamma += s1`` (local sums, like :func:`bn_bwd_reduce`)."""
    N, C = w.shape[0], w.shape[-1]
    s1 = invstd * ((w.reshape(N, C).float() * prod.reshape(N, C)).sum(1) - mean * s0)
    if dbeta is not None:
        dbeta.add_(s0)
    if dgamma is not None:
        dgamma.add_(s1)
    return torch.stack([s0, s1])


def conv_dgrad_folded(g: torch.Tensor, a_in: torch.Tensor, fold: BnFold, bwd_stats=None,
                      out: Optional[torch.Tensor] = None, **_) -> torch.Tensor:
    """``g . (A o w) + a_in . S + bias`` (see :func:`bn_fold_prepare`); ``bwd_stats`` is a fused backend's
    business (ignored here like in :func:`conv_dgrad`)."""
    N, C = fold.N, fold.C
    wf = fold.wfold.float()
    dx = g.float().reshape(-1, N) @ wf[:, :N].t() + a_in.float().reshape(-1, C) @ wf[:, N:].t() + fold.bias
    dx = dx.reshape(a_in.shape).to(g.dtype)
    if out is not None:
        out.copy_(dx)
        return out
    return dx


def gram_and_means(a: torch.Tensor, colsum_slot: Optional[str] = None):
    """``(a^T a, [column means, column means of squares])`` of an NHWC tensor seen as ``[rows, C]`` (fp32)."""
    a2 = a.float().reshape(-1, a.shape[-1])
    return a2.t() @ a2, torch.stack([a2.mean(0), (a2 * a2).mean(0)])


def bn_stats_from_gram(w: torch.Tensor, gram: torch.Tensor, mom: torch.Tensor, rows: int,
                       stat_shift: Optional[torch.Tensor], **_) -> torch.Tensor:
    """Shifted moments (see :func:`channel_moments`) of ``y = a . w^T`` from the Gram matrix and column means of
    ``a`` alone: ``E[y_n] = w_n . mean(a)``, ``E[y_n^2] = w_n (a^T a / rows) w_n^T`` -- the batch statistics of a
    1x1 convolution's output are known before the convolution has run."""
    N, C = w.shape[0], w.shape[-1]
    w2 = w.reshape(N, C).float()
    mean = w2 @ mom[0]
    # centred first: the variance as a quadratic form of its own, not the difference of two numbers ~ mean^2
    var = ((w2 @ (gram / rows - torch.outer(mom[0], mom[0]))) * w2).sum(1)
    off = mean - stat_shift if stat_shift is not None else mean
    return torch.stack([off, var + off * off])


def bn_fuse_fprop_weights(w, scale, shift, w2=None, scale2=None, shift2=None):
    """``([scale o w | scale2 o w2], shift + shift2)`` (rows = output channels; ``w2`` optional, both 1x1)."""
    N = w.shape[0]
    parts = [w.reshape(N, -1).float() * scale[:, None]]
    sh = shift.clone()
    if w2 is not None:
        parts.append(w2.reshape(N, -1).float() * scale2[:, None])
        sh = sh + shift2
    return torch.cat(parts, dim=1).to(w.dtype).contiguous(), sh


def conv_fprop_fused_tail(a: torch.Tensor, wf: torch.Tensor, shift: torch.Tensor, addend: Optional[torch.Tensor] = None,
                          in2: Optional[torch.Tensor] = None, bits_out: Optional[torch.Tensor] = None,
                          out: Optional[torch.Tensor] = None, **_) -> torch.Tensor:
    """``relu(a . wf[:, :C]^T + in2 . wf[:, C:]^T + shift + addend)`` (+ packed sign mask): a residual block's tail
    as one product, see :func:`bn_fuse_fprop_weights`."""
    C = a.shape[-1]
    N = wf.shape[0]
    wf32 = wf.float()
    y = a.float().reshape(-1, C) @ wf32[:, :C].t() + shift
    if in2 is not None:
        y = y + in2.float().reshape(-1, in2.shape[-1]) @ wf32[:, C:].t()
    y = y.reshape(*a.shape[:-1], N)
    if addend is not None:
        y = y + addend.float()
    y = torch.relu(y).to(a.dtype)
    if bits_out is not None:
        _pack_bits(y, bits_out)
    if out is not None:
        out.copy_(y)
        return out
    return y


def channel_sums(g: torch.Tensor) -> torch.Tensor:
    return g.float().reshape(-1, g.shape[-1]).sum(0)


def conv_wgrad_folded(a_in: torch.Tensor, g: torch.Tensor, fold: BnFold, w: torch.Tensor, dw: torch.Tensor,
                      prod: Optional[torch.Tensor] = None, amean_slot: Optional[str] = None, gram=None,
                      mom=None) -> None:
    """``dw += A o (g^T a) + B o (w . (a^T a)) + K (x) colsum(a)`` (see :func:`bn_fold_prepare`); ``prod``: the
    product ``g^T a`` when it has been formed already (:func:`bn_fold_sums`)."""
    N, C = fold.N, fold.C
    a2, g2 = a_in.float().reshape(-1, C), g.float().reshape(-1, N)
    A, Bc, K = fold.coef[0], fold.coef[1], fold.coef[2]
    gram = a2.t() @ a2
    p2 = prod.reshape(N, C) if prod is not None else g2.t() @ a2
    upd = A[:, None] * p2 + Bc[:, None] * (w.reshape(N, C).float() @ gram) + K[:, None] * a2.sum(0)[None, :]
    dw.add_(upd.reshape(dw.shape))

# ---------------------------------------------------------------------------- pooling
def maxpool_fwd(x: torch.Tensor):
    """3x3 / stride 2 / pad 1 max pooling; returns ``(out, idx)`` with an opaque index tensor."""
    out, idx = F.max_pool2d(_nchw(x.float()), 3, 2, 1, return_indices=True)
    return _nhwc(out).to(x.dtype), idx


pool_winners = True


def bn_relu_maxpool_fwd(y, scale, shift, winners: bool = False):
    """``maxpool3x3s2(relu(y*scale + shift))``: the stem's BatchNorm-apply and pooling in one op.  ``winners``: also
    ``y`` at every window's winner, and the packed sign mask of the pooled result attached to it."""
    out, idx = maxpool_fwd(bn_apply(y, scale, shift, None, True))
    if not winners:
        return out, idx
    B, H, W, C = y.shape
    ywin = _nhwc(_nchw(y.float()).reshape(B, C, -1).gather(2, idx.reshape(B, C, -1)).view(idx.shape)).to(y.dtype)
    bits = new_relu_bits(out)
    _pack_bits(out, bits)
    out._pdt_relu_bits = bits
    return out, idx, ywin


def maxpool_bwd(dout: torch.Tensor, idx: torch.Tensor, in_shape) -> torch.Tensor:
    B, H, W, C = in_shape
    # Overlapping 3x3/s2 windows can select the same input element twice and the gradient
    # must then be the SUM of both, so scatter-add (max_unpool2d would overwrite).
    dx = torch.zeros(B, C, H * W, dtype=torch.float32, device=dout.device)
    dx.scatter_add_(2, idx.reshape(B, C, -1), _nchw(dout.float()).reshape(B, C, -1))
    return _nhwc(dx.view(B, C, H, W)).to(dout.dtype)


def avgpool_fwd(x: torch.Tensor) -> torch.Tensor:
    return x.float().mean(dim=(1, 2)).to(x.dtype)


def avgpool_bwd(dp: torch.Tensor, H: int, W: int, relu_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, C = dp.shape
    dx = (dp.float() / (H * W)).view(B, 1, 1, C).expand(B, H, W, C).contiguous().to(dp.dtype)
    if relu_mask is not None:
        dx = dx * (relu_mask > 0)
    return dx


# ---------------------------------------------------------------------------- classifier + loss
def linear_fwd(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    return (a.float() @ w.float().t() + bias).to(a.dtype)


def linear_bwd(a, dlogits, w, wt, dw: torch.Tensor, dbias: torch.Tensor) -> torch.Tensor:
    d32 = dlogits.float()
    dw.add_(d32.t() @ a.float())
    dbias.add_(d32.sum(0))
    return (d32 @ w.float()).to(a.dtype)


def softmax_ce(logits: torch.Tensor, labels: torch.Tensor, grad_scale: float = 1.0):
    """Mean cross-entropy and its gradient ``(softmax - onehot) * grad_scale / B`` in one pass."""
    l32 = logits.float()
    logp = torch.log_softmax(l32, dim=1)
    B = l32.shape[0]
    loss = -logp.gather(1, labels.view(-1, 1).long()).mean()
    d = torch.exp(logp)
    d.scatter_add_(1, labels.view(-1, 1).long(), torch.full((B, 1), -1.0, device=d.device))
    return loss, (d * (grad_scale / B)).to(logits.dtype)


def ce_metrics(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """``[loss_sum, correct@1, correct@5, n]`` (fp32) for a validation batch."""
    l32 = logits.float()
    lab = labels.view(-1, 1).long()
    logp = torch.log_softmax(l32, dim=1)
    loss_sum = -logp.gather(1, lab).sum()
    rank = (l32 > l32.gather(1, lab)).sum(dim=1)
    n = torch.tensor(float(l32.shape[0]), device=l32.device)
    return torch.stack([loss_sum, (rank < 1).sum().float(), (rank < 5).sum().float(), n])


# ---------------------------------------------------------------------------- optimizer
def sgd_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, pc: Optional[torch.Tensor], lr: float,
             momentum: float, weight_decay: float, grad_scale: float, nesterov: bool = False,
             dampening: float = 0.0, first_step: bool = False) -> None:
    """Flat SGD: ``g = g*grad_scale + wd*p; m = mu*m + (1-damp)*g; p -= lr*(m | g+mu*m)``.

    Zeroes ``g`` afterwards (weight-gradient kernels accumulate into it) and refreshes the
    compute-dtype mirror ``pc`` when it is a separate buffer.  On the first step the
    momentum buffer is initialised to the gradient, like ``torch.optim.SGD``.
    """
    grad = g * grad_scale
    if weight_decay != 0.0:
        grad = grad + weight_decay * p
    if momentum != 0.0:
        if first_step:
            m.copy_(grad)
        else:
            m.mul_(momentum).add_(grad, alpha=1.0 - dampening)
        step = grad + momentum * m if nesterov else m
    else:
        step = grad
    p.add_(step, alpha=-lr)
    g.zero_()
    if pc is not None and pc.data_ptr() != p.data_ptr():
        pc.copy_(p)
