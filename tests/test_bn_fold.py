"""BatchNorm backward folded into the 1x1 convolution in front of it (ops ``bn_fold_prepare`` /
``conv_dgrad_folded`` / ``conv_wgrad_folded``): the folded products against the plain sequence
BatchNorm-backward elementwise pass -> data gradient / weight gradient, and the engine with the
folded path on and off.  CPU tier: the reference ops in fp64/fp32; GPU tier: the HIP kernels against
the fp32 reference of the same op."""
import os

import pytest
import torch

from pytorch_distributed_training_amd.ops import ref
from pytorch_distributed_training_amd.ops.ref import MASK_NONE, MASK_RECOMPUTE


def _layer(B, H, C, N, dtype, device, seed=0):
    """A 1x1 convolution + BatchNorm in training mode with its forward statistics and an upstream gradient."""
    g_ = torch.Generator().manual_seed(seed)
    a = torch.relu(torch.randn(B, H, H, C, generator=g_) + 0.3).to(dtype).to(device)        # post-ReLU input
    w = (torch.randn(N, 1, 1, C, generator=g_) * (2.0 / C) ** 0.5).to(dtype).to(device)
    gamma = (torch.rand(N, generator=g_) + 0.5).to(device)
    y, _ = ref.conv_fprop(a, w, 1, 1, 1, 0)
    y32 = y.float().reshape(-1, N)
    mean = y32.mean(0)
    invstd = torch.rsqrt(y32.var(0, unbiased=False) + 1e-5)
    scale = (gamma * invstd).float()
    shift = (-mean * scale).float()
    g = (torch.randn(B, H, H, N, generator=g_) * 0.1).to(dtype).to(device)
    return a, w, y, mean.float(), invstd.float(), scale, shift, g


def _plain(a, w, y, mean, invstd, scale, shift, g):
    """The unfolded sequence; returns (sums, dx, dw)."""
    N, C = w.shape[0], w.shape[-1]
    sums = ref.bn_bwd_reduce(g, y, mean, invstd, scale, shift, None, MASK_NONE)
    inv = 1.0 / (y.numel() // N)
    d = ref.bn_bwd_elemt(g, y, mean, invstd, scale, shift, sums, inv, None, MASK_NONE)
    dx = ref.conv_dgrad(d, w, None, tuple(a.shape), 1, 1, 1, 0)
    dw = torch.zeros(N, 1, 1, C, dtype=torch.float32, device=a.device)
    ref.conv_wgrad(a, d, 1, 1, 1, 0, dw)
    return sums, inv, dx, dw


def test_folded_reference_ops_match_the_plain_sequence():
    """fp32 tensors (the reference ops compute in fp32): the two routes are the same algebra."""
    a, w, y, mean, invstd, scale, shift, g = _layer(3, 6, 16, 64, torch.float32, torch.device("cpu"))
    sums, inv, dx, dw = _plain(a, w, y, mean, invstd, scale, shift, g)
    fold = ref.bn_fold_prepare(sums, scale, invstd, mean, inv, w)
    assert fold.wfold.shape == (16, 64 + 16) and fold.coef.shape == (3, 64)
    dx_f = ref.conv_dgrad_folded(g, a, fold)
    dw_f = torch.zeros_like(dw)
    ref.conv_wgrad_folded(a, g, fold, w, dw_f)
    assert torch.allclose(dx_f, dx, rtol=1e-4, atol=1e-5 * dx.abs().max().item())
    assert torch.allclose(dw_f, dw, rtol=1e-4, atol=1e-5 * dw.abs().max().item())
    # accumulation semantics
    ref.conv_wgrad_folded(a, g, fold, w, dw_f)
    assert torch.allclose(dw_f, 2 * dw, rtol=1e-4, atol=2e-5 * dw.abs().max().item())


def test_backward_sums_from_the_weight_gradient_product():
    """sum g * xhat from s0 = sum g and the product g^T a equals the reduction over g and y."""
    a, w, y, mean, invstd, scale, shift, g = _layer(3, 6, 16, 64, torch.float32, torch.device("cpu"), seed=4)
    sums, inv, dx, dw = _plain(a, w, y, mean, invstd, scale, shift, g)
    prod = torch.zeros(64, 1, 1, 16)
    ref.conv_wgrad(a, g, 1, 1, 1, 0, prod)
    dgamma, dbeta = torch.zeros(64), torch.zeros(64)
    got = ref.bn_fold_sums(prod, w, g.reshape(-1, 64).sum(0), mean, invstd, dgamma, dbeta)
    assert torch.allclose(got, sums, rtol=1e-4, atol=1e-4 * sums.abs().max().item())
    assert torch.equal(dbeta, got[0]) and torch.equal(dgamma, got[1])
    # ... and the weight gradient assembled from that same product
    fold = ref.bn_fold_prepare(got, scale, invstd, mean, inv, w)
    dw_f = torch.zeros_like(dw)
    ref.conv_wgrad_folded(a, g, fold, w, dw_f, prod=prod)
    assert torch.allclose(dw_f, dw, rtol=1e-4, atol=1e-5 * dw.abs().max().item())
    # conv_dgrad(out_sums=True) leaves the sums of what it returns
    wt = None
    d = ref.conv_dgrad(g, w, wt, tuple(a.shape), 1, 1, 1, 0, out_sums=True)
    gs = ref.grad_sums(16)
    assert torch.allclose(gs[0], d.reshape(-1, 16).sum(0), rtol=1e-5, atol=1e-5)


def test_engine_with_and_without_the_folded_path(monkeypatch):
    """Same model, same batch: every gradient of a bottleneck network agrees between the two paths."""
    from pytorch_distributed_training_amd.engine.native_resnet import NativeResNet
    from pytorch_distributed_training_amd.models import get_model, get_spec

    torch.manual_seed(3)
    state = get_model("ResNet50", 10).state_dict()
    img = torch.randn(4, 3, 32, 32)
    label = torch.randint(0, 10, (4,))
    grads = []
    for on, product in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("PDT_BN_FOLD", on)
        monkeypatch.setenv("PDT_BN_FOLD_PRODUCT", product)
        eng = NativeResNet(get_spec("ResNet50", 10), torch.device("cpu"), ref, torch.float32)
        eng.load_state_dict(state)
        assert eng._bn_fold == (on == "1") and eng._bn_fold_product == (product == "1")
        eng.forward_backward(img, label)
        grads.append(eng.params.g.clone())
    assert grads[0].abs().max() > 0
    for gfold in grads[:2]:
        rel = (gfold - grads[2]).abs().max() / grads[2].abs().max()
        assert rel < 2e-4, rel


# ----------------------------------------------------------------------------------------------- GPU tier
@pytest.fixture(scope="module")
def hip():
    from pytorch_distributed_training_amd.ops import hip as hip_ops
    return hip_ops


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


# (B, H, C, N): one output tile, many tiles per workgroup (persistent loop, statistics carried over tiles), both
# tile shapes of the data-gradient kernel (C = 64: 128 x 64, C >= 128: 128 x 128), a ragged last tile
FOLD_SHAPES = [(2, 7, 64, 256), (1, 7, 512, 2048), (3, 14, 256, 1024), (2, 28, 128, 512), (64, 56, 64, 256),
               (16, 28, 128, 512), (5, 9, 64, 256)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", FOLD_SHAPES)
def test_hip_folded_products_match_the_reference(hip, device, shape):
    B, H, C, N = shape
    a, w, y, mean, invstd, scale, shift, g = _layer(B, H, C, N, torch.bfloat16, device, seed=B * 131 + C)
    sums, inv, dx_plain, dw_plain = _plain(a, w, y, mean, invstd, scale, shift, g)
    wt = torch.empty(C, 1, 1, N, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)

    fold_ref = ref.bn_fold_prepare(sums, scale, invstd, mean, inv, w)
    fold = hip.bn_fold_prepare(sums, scale, invstd, mean, inv, w, wt)
    assert torch.allclose(fold.coef, fold_ref.coef, rtol=1e-5, atol=1e-7)
    assert rel_err(fold.wfold[:, :N], fold_ref.wfold[:, :N]) < 1e-2
    assert rel_err(fold.wfold[:, N:], fold_ref.wfold[:, N:]) < 1.5e-2
    assert rel_err(fold.bias, fold_ref.bias) < 2e-3

    # data gradient: against the fp32 reference fed the SAME folded weights, and against the plain sequence
    dx = hip.conv_dgrad_folded(g, a, fold)
    assert rel_err(dx, ref.conv_dgrad_folded(g, a, fold)) < 1.5e-2
    assert rel_err(dx, dx_plain) < 3e-2
    # ... with the backward sums of the consuming BatchNorm accumulated by the epilogue (recomputed ReLU mask)
    yp = (torch.randn(B, H, H, C, device=device) * 0.7).to(torch.bfloat16)
    pm, psc, psh = torch.randn(C, device=device) * 0.1, torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.2
    pinv = torch.rand(C, device=device) + 0.5
    dx2 = hip.conv_dgrad_folded(g, a, fold, bwd_stats=(yp, pm, psc, psh, MASK_RECOMPUTE))
    assert torch.equal(dx2, dx)
    got = hip.bn_bwd_reduce(dx2, yp, pm, pinv, psc, psh, None, MASK_RECOMPUTE, prereduced=True)
    want = ref.bn_bwd_reduce(dx2, yp, pm, pinv, psc, psh, None, MASK_RECOMPUTE)
    assert torch.allclose(got, want, rtol=2e-3, atol=2e-3 * want.abs().max().item())

    # weight gradient (accumulating): against the reference on the same coefficients and against the plain sequence
    dw = torch.zeros(N, 1, 1, C, dtype=torch.float32, device=device)
    hip.conv_wgrad_folded(a, g, fold, w, dw)
    dw_ref = torch.zeros_like(dw)
    ref.conv_wgrad_folded(a, g, fold, w, dw_ref)
    assert rel_err(dw, dw_ref) < 5e-3
    assert rel_err(dw, dw_plain) < 3e-2
    hip.conv_wgrad_folded(a, g, fold, w, dw)
    assert rel_err(dw, 2 * dw_ref) < 5e-3

    # the backward sums from s0 and the unscaled product, and the weight gradient assembled from that product
    prod = hip.zeros_f32((N, 1, 1, C), device)
    hip.conv_wgrad(a, g, 1, 1, 1, 0, prod)
    s0 = g.float().reshape(-1, N).sum(0)
    dgamma, dbeta = torch.zeros(N, device=device), torch.zeros(N, device=device)
    got = hip.bn_fold_sums(prod, w, s0, mean, invstd, dgamma, dbeta)
    want = ref.bn_fold_sums(prod, w, s0, mean, invstd)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4 * want.abs().max().item())
    assert torch.allclose(got[1], sums[1], rtol=2e-2, atol=2e-2 * sums[1].abs().max().item())   # (y rounded to bf16)
    assert torch.equal(dbeta, got[0]) and torch.equal(dgamma, got[1])
    dw3 = torch.zeros_like(dw)
    hip.conv_wgrad_folded(a, g, fold, w, dw3, prod=prod)
    assert rel_err(dw3, dw_ref) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 7, 64), (64, 28, 128), (5, 9, 512)])
def test_hip_bn_apply_leaves_the_column_sums_of_its_result(hip, device, shape):
    B, H, C = shape
    torch.manual_seed(B + C)
    y = torch.randn(B, H, H, C, device=device).to(torch.bfloat16)
    scale, shift = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.3
    out = hip.bn_apply(y, scale, shift, None, True, colsum_slot="test")
    assert rel_err(out, ref.bn_apply(y, scale, shift, None, True)) < 1e-2
    got = hip.colsum_take("test", C, device)
    want = out.float().reshape(-1, C).sum(0)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-3)
    assert hip.colsum_take("test", C, device).abs().max().item() == 0.0


@pytest.mark.gpu
def test_hip_subsample2(hip, device):
    x = torch.randn(3, 10, 6, 64, device=device).to(torch.bfloat16)
    assert torch.equal(hip.subsample2(x), ref.subsample2(x))


@pytest.mark.gpu
def test_hip_dgrad_leaves_the_sums_of_its_result(hip, device):
    B, H, Ci, Co = 6, 14, 256, 64
    torch.manual_seed(2)
    dy = (torch.randn(B, H, H, Co, device=device) * 0.3).to(torch.bfloat16)
    w = (torch.randn(Co, 1, 1, Ci, device=device) * 0.1).to(torch.bfloat16)
    wt = torch.empty(Ci, 1, 1, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    addend = (torch.randn(B, H, H, Ci, device=device) * 0.3).to(torch.bfloat16)
    bits = hip.new_relu_bits(addend)
    bits.copy_(torch.randint(0, 256, tuple(bits.shape), device=device, dtype=torch.uint8))
    dx = hip.conv_dgrad(dy, w, wt, (B, H, H, Ci), 1, 1, 1, 0, addend=addend, relu_bits=bits, out_sums=True)
    gs = hip.grad_sums(Ci, device)
    d2 = dx.float().reshape(-1, Ci)
    assert torch.allclose(gs[0], d2.sum(0), rtol=1e-3, atol=1e-3 * d2.sum(0).abs().max().item())
    assert torch.allclose(gs[1], (d2 * d2).sum(0), rtol=1e-3, atol=1e-3)
    assert rel_err(dx, ref.conv_dgrad(dy, w, None, (B, H, H, Ci), 1, 1, 1, 0, addend=addend, relu_bits=bits)) < 1.5e-2
    assert hip.grad_sums(Ci, device).abs().max().item() == 0.0           # collected: the accumulator is clean again


@pytest.mark.gpu
def test_hip_engine_with_and_without_the_folded_path(hip, device, monkeypatch):
    from pytorch_distributed_training_amd.engine.native_resnet import NativeResNet
    from pytorch_distributed_training_amd.models import get_model, get_spec

    torch.manual_seed(5)
    state = get_model("ResNet50", 10).state_dict()
    img = torch.randn(8, 3, 64, 64, device=device)
    label = torch.randint(0, 10, (8,), device=device)
    grads, losses = [], []
    monkeypatch.setenv("PDT_BN_FOLD_MIN_MB", "0")        # fold every bottleneck block, however small its tensors
    for on, product in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("PDT_BN_FOLD", on)
        monkeypatch.setenv("PDT_BN_FOLD_PRODUCT", product)
        eng = NativeResNet(get_spec("ResNet50", 10), device, hip, torch.bfloat16, hip_graphs=True)
        eng.load_state_dict(state)
        assert eng._bn_fold == (on == "1") and eng._bn_fold_product == (product == "1")
        for _ in range(3):                               # eager, recorded, replayed
            loss = float(eng.forward_backward(img, label))
            torch.cuda.synchronize()
            last = eng.params.g.clone()
            eng.params.g.zero_()
        losses.append(loss)
        grads.append(last)
    assert abs(losses[0] - losses[2]) < 1e-3 and abs(losses[1] - losses[2]) < 1e-3    # the forward pass is untouched
    # per parameter tensor: the paths differ by bf16 rounding of different intermediates
    for gfold in grads[:2]:
        for key, e in eng.params.entries.items():
            a = gfold[e.offset:e.offset + e.numel]
            b = grads[2][e.offset:e.offset + e.numel]
            cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
            assert cos > 0.99, (key, cos)
