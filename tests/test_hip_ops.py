"""GPU tier: every HIP kernel against the plain-PyTorch fp32 reference of the same op."""
import pytest
import torch

from pytorch_distributed_training_amd.ops import ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytorch_distributed_training_amd.ops import hip as hip_ops
    return hip_ops


def rnd(*shape, device, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(*shape, device=device) * scale).to(dtype)


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


# 23 distinct ResNet-50 convolution shapes: (Cin, Cout, k, stride, pad, H)
R50_SHAPES = [
    (64, 64, 1, 1, 0, 56), (64, 64, 3, 1, 1, 56), (64, 256, 1, 1, 0, 56), (256, 64, 1, 1, 0, 56),
    (256, 128, 1, 1, 0, 56), (128, 128, 3, 2, 1, 56), (128, 512, 1, 1, 0, 28), (256, 512, 1, 2, 0, 56),
    (512, 128, 1, 1, 0, 28), (128, 128, 3, 1, 1, 28), (512, 256, 1, 1, 0, 28), (256, 256, 3, 2, 1, 28),
    (256, 1024, 1, 1, 0, 14), (512, 1024, 1, 2, 0, 28), (1024, 256, 1, 1, 0, 14), (256, 256, 3, 1, 1, 14),
    (1024, 512, 1, 1, 0, 14), (512, 512, 3, 2, 1, 14), (512, 2048, 1, 1, 0, 7), (1024, 2048, 1, 2, 0, 14),
    (2048, 512, 1, 1, 0, 7), (512, 512, 3, 1, 1, 7),
]


@pytest.mark.parametrize("shape", R50_SHAPES)
@pytest.mark.parametrize("B", [1, 5])
def test_conv_fprop_dgrad_wgrad(hip, device, shape, B):
    Ci, Co, k, stride, pad, H = shape
    torch.manual_seed(Ci * 7 + Co + k + stride + B)
    x = rnd(B, H, H, Ci, device=device)
    w = rnd(Co, k, k, Ci, device=device, scale=(2.0 / (k * k * Ci)) ** 0.5)
    shift = torch.randn(Co, device=device) * 0.1
    y, mom = hip.conv_fprop(x, w, k, k, stride, pad, shift)
    y_ref, mom_ref = ref.conv_fprop(x, w, k, k, stride, pad, shift)
    assert rel_err(y, y_ref) < 1.5e-2
    # the fused statistics are those of the bf16 values actually stored
    assert torch.allclose(mom, ref.bn_stats(y, shift), rtol=1e-3, atol=1e-3)
    assert torch.allclose(mom, mom_ref, rtol=1e-2, atol=1e-2)

    dy = rnd(*y.shape, device=device)
    wt = torch.empty(Ci, k, k, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    assert torch.equal(wt, w.permute(3, 1, 2, 0).contiguous())
    addend = rnd(*x.shape, device=device)
    mask = rnd(*x.shape, device=device)
    for a, m in ((None, None), (addend, None), (addend, mask), (None, mask)):
        dx = hip.conv_dgrad(dy, w, wt, tuple(x.shape), k, k, stride, pad, a, m)
        dx_ref = ref.conv_dgrad(dy, w, None, tuple(x.shape), k, k, stride, pad, a, m)
        assert rel_err(dx, dx_ref) < 1.5e-2

    dw = torch.zeros(Co, k, k, Ci, dtype=torch.float32, device=device)
    dw_ref = torch.zeros_like(dw)
    hip.conv_wgrad(x, dy, k, k, stride, pad, dw)
    ref.conv_wgrad(x, dy, k, k, stride, pad, dw_ref)
    assert rel_err(dw, dw_ref) < 5e-3
    # accumulation semantics
    hip.conv_wgrad(x, dy, k, k, stride, pad, dw)
    assert rel_err(dw, 2 * dw_ref) < 5e-3


def test_conv_identity_asymmetric(hip, device):
    """A = identity weights, asymmetric data: catches a row/column swap in the C write."""
    B, H, C = 2, 9, 64
    x = torch.arange(B * H * H * C, device=device, dtype=torch.float32).reshape(B, H, H, C)
    x = ((x % 251) - 125).to(torch.bfloat16)
    w = torch.eye(C, device=device).reshape(C, 1, 1, C).to(torch.bfloat16)
    y, _ = hip.conv_fprop(x, w, 1, 1, 1, 0, None)
    assert torch.equal(y, x)


@pytest.mark.parametrize("tile", [1, 2, 3, 5])
def test_conv_tiles(hip, device, tile):
    torch.manual_seed(tile)
    x = rnd(3, 14, 14, 128, device=device)
    w = rnd(192, 3, 3, 128, device=device, scale=0.03)
    y = hip.conv_fprop(x, w, 3, 3, 1, 1, None, tile=tile)[0]
    y_ref = ref.conv_fprop(x, w, 3, 3, 1, 1, None)[0]
    assert rel_err(y, y_ref) < 1.5e-2


@pytest.mark.parametrize("k,stride,pad", [(1, 1, 0), (3, 1, 1), (3, 2, 1)])
def test_conv_launches_split_over_the_batch(hip, device, monkeypatch, k, stride, pad):
    """Register-staged kernel: operands beyond the 2 GB buffer-descriptor window go out as several launches over
    batch ranges (threshold lowered here): same outputs, statistics and weight gradients."""
    torch.manual_seed(k + stride)
    B, H, Ci, Co = 7, 12, 64, 128
    x = rnd(B, H, H, Ci, device=device)
    w = rnd(Co, k, k, Ci, device=device, scale=(2.0 / (k * k * Ci)) ** 0.5)
    wt = torch.empty(Ci, k, k, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    shift = torch.zeros(Co, device=device)
    bits = hip.new_relu_bits(x)
    ref._pack_bits(x, bits)

    def run():
        y, mom = hip.conv_fprop(x, w, k, k, stride, pad, shift)
        dy = (y.float() * 0.5 + 0.1).to(torch.bfloat16)
        mean = torch.randn(Ci, device=device) * 0.1
        dx = hip.conv_dgrad(dy, w, wt, tuple(x.shape), k, k, stride, pad, addend=x, relu_bits=bits,
                            bwd_stats=(x, mean, None, None, ref.MASK_NONE))
        sums = hip.bn_bwd_reduce(dx, x, mean, torch.ones(Ci, device=device), None, None, None, ref.MASK_NONE,
                                 prereduced=True)
        dw = torch.zeros(Co, k, k, Ci, dtype=torch.float32, device=device)
        hip.conv_wgrad(x, dy, k, k, stride, pad, dw)
        return y, mom, dx, sums, dw

    # (only the register-staged kernel splits: the streaming kernels rebase their descriptors and take any size)
    monkeypatch.setattr(hip, "_USE_CONV_DMA", False)
    torch.manual_seed(1)
    whole = run()
    monkeypatch.setattr(hip, "MAX_OPERAND_BYTES", x[:2].numel() * 2)       # -> 4 launches of <= 2 images
    assert len(hip._batch_parts(B, x.numel() * 2)) == 4
    torch.manual_seed(1)
    split = run()
    assert torch.equal(whole[0], split[0]) and torch.equal(whole[2], split[2])
    assert torch.allclose(whole[1], split[1], rtol=1e-4, atol=1e-5)
    assert torch.allclose(whole[3], split[3], rtol=1e-3, atol=1e-2)
    assert rel_err(split[4], whole[4]) < 1e-4


@pytest.mark.parametrize("k", [1, 3])
def test_tensors_beyond_two_gigabytes_in_one_launch(hip, device, k):
    """2.25 GB activations (1400 images of 112 x 112 x 64) through ONE launch each of the streaming kernels -- forward,
    data gradient with its fused epilogue, weight gradient, and the two-operand block tail -- against the same ops on
    two halves that each fit a descriptor.  Outputs are equal bit for bit (a row's value does not depend on the launch
    it is computed in), sums and weight gradients to accumulation order."""
    B, H, C = 1400, 112, 64
    pad = k // 2
    g = torch.Generator(device=device).manual_seed(k)
    x = (torch.randn(B, H, H, C, device=device, generator=g) * 0.5).to(torch.bfloat16)
    assert x.numel() * 2 > 2 ** 31
    w = (torch.randn(C, k, k, C, device=device, generator=g) * (2.0 / (k * k * C)) ** 0.5).to(torch.bfloat16)
    wt = torch.empty(C, k, k, C, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    shift = torch.zeros(C, device=device)
    halves = [(0, B // 2), (B // 2, B)]

    y, mom = hip.conv_fprop(x, w, k, k, 1, pad, shift)
    parts = [hip.conv_fprop(x[a:b], w, k, k, 1, pad, shift) for a, b in halves]
    for (a, b), (yp, _) in zip(halves, parts):
        assert torch.equal(y[a:b], yp)
    assert torch.allclose(mom, 0.5 * (parts[0][1] + parts[1][1]), rtol=1e-3, atol=1e-4)

    bits = hip.new_relu_bits(x)
    ref._pack_bits(x, bits)
    dx = hip.conv_dgrad(y, w, wt, tuple(x.shape), k, k, 1, pad, addend=x, relu_bits=bits)
    for a, b in halves:
        part = hip.conv_dgrad(y[a:b], w, wt, (b - a, H, H, C), k, k, 1, pad, addend=x[a:b], relu_bits=bits[a:b])
        assert torch.equal(dx[a:b], part)
    del dx

    dw = torch.zeros(C, k, k, C, dtype=torch.float32, device=device)
    hip.conv_wgrad(x, y, k, k, 1, pad, dw)
    dw2 = torch.zeros_like(dw)
    for a, b in halves:
        hip.conv_wgrad(x[a:b], y[a:b], k, k, 1, pad, dw2)
    assert rel_err(dw, dw2) < 1e-4

    if k == 1:                                           # second operand: both tensors beyond 2 GB
        wf = (torch.randn(C, 2 * C, device=device, generator=g) * 0.1).to(torch.bfloat16)
        sh = torch.randn(C, device=device, generator=g)
        out = hip.conv_fprop_fused_tail(x, wf, sh, in2=y)
        for a, b in halves:
            assert torch.equal(out[a:b], hip.conv_fprop_fused_tail(x[a:b], wf, sh, in2=y[a:b]))


@pytest.mark.parametrize("tile", [1, 2, 4, 8, 9, 10, 11])
def test_wgrad_tiles(hip, device, tile):
    torch.manual_seed(tile)
    x = rnd(3, 14, 14, 128, device=device)
    dy = rnd(3, 14, 14, 192, device=device)
    dw = torch.zeros(192, 3, 3, 128, dtype=torch.float32, device=device)
    dw_ref = torch.zeros_like(dw)
    hip.conv_wgrad(x, dy, 3, 3, 1, 1, dw, tile=tile)
    ref.conv_wgrad(x, dy, 3, 3, 1, 1, dw_ref)
    assert rel_err(dw, dw_ref) < 5e-3


@pytest.mark.parametrize("tile,Co", [(6, 64), (7, 128), (7, 192)])
@pytest.mark.parametrize("stride", [1, 2])
def test_wgrad_three_taps_per_workgroup(hip, device, tile, Co, stride):
    """3x3 weight gradient with one kernel row (three taps) per workgroup, Cin <= 64."""
    torch.manual_seed(tile + Co)
    H = 14
    Ho = (H + 2 - 3) // stride + 1
    x = rnd(5, H, H, 64, device=device)
    dy = rnd(5, Ho, Ho, Co, device=device)
    dw = torch.zeros(Co, 3, 3, 64, dtype=torch.float32, device=device)
    dw_ref = torch.zeros_like(dw)
    hip.conv_wgrad(x, dy, 3, 3, stride, 1, dw, tile=tile)
    ref.conv_wgrad(x, dy, 3, 3, stride, 1, dw_ref)
    assert rel_err(dw, dw_ref) < 5e-3


@pytest.mark.parametrize("tile", [0, 10, 11])
@pytest.mark.parametrize("B,H,W,Ci,Co", [(3, 7, 7, 64, 64), (2, 14, 14, 128, 128), (5, 28, 28, 64, 192), (1, 56, 56, 64, 64),
                                         (3, 9, 5, 192, 72), (2, 3, 63, 64, 128), (37, 7, 7, 512, 512)])
def test_wgrad_kernel_row_per_workgroup(hip, device, tile, B, H, W, Ci, Co):
    """3x3 / stride 1 / padding 1 with the gathered operand staged once per kernel row: image rows padded to a power of two
    positions (7 -> 8, 14 -> 16, 28 -> 32, 56 -> 64, and widths that leave several padding positions), position counts
    that are no multiple of the K step, partial channel tiles, many slices."""
    torch.manual_seed(B * 100 + W)
    x = rnd(B, H, W, Ci, device=device)
    dy = rnd(B, H, W, Co, device=device)
    dw = torch.zeros(Co, 3, 3, Ci, dtype=torch.float32, device=device)
    dw_ref = torch.zeros_like(dw)
    hip.conv_wgrad(x, dy, 3, 3, 1, 1, dw, tile=tile)
    ref.conv_wgrad(x, dy, 3, 3, 1, 1, dw_ref)
    assert rel_err(dw, dw_ref) < 5e-3
    hip.conv_wgrad(x, dy, 3, 3, 1, 1, dw, tile=tile, target_blocks=64)      # accumulates; another split
    assert rel_err(dw, 2 * dw_ref) < 5e-3


@pytest.mark.parametrize("fmt", ["float_nchw", "uint8_nhwc"])
@pytest.mark.parametrize("B,H", [(2, 64), (3, 224)])
def test_stem(hip, device, fmt, B, H):
    torch.manual_seed(B + H)
    mean = torch.tensor((0.485, 0.456, 0.406), device=device)
    std = torch.tensor((0.229, 0.224, 0.225), device=device)
    if fmt == "uint8_nhwc":
        img = torch.randint(0, 256, (B, H, H, 3), device=device, dtype=torch.uint8)
    else:
        img = torch.randn(B, 3, H, H, device=device)
    x0 = hip.input_transform(img, mean, std, torch.bfloat16)
    x0_ref = ref.input_transform(img, mean, std, torch.bfloat16)
    assert torch.equal(x0[:, 3:3 + H, 3:3 + H, :3], x0_ref)
    assert x0[:, :3].abs().sum() == 0 and x0[..., 3].abs().sum() == 0
    w = rnd(64, 7, 7, 3, device=device, scale=0.1)
    wp = hip.prepare_stem_weight(w)
    shift = torch.zeros(64, device=device)
    y, mom = hip.stem_fprop(x0, wp, shift)
    y_ref, mom_ref = ref.stem_fprop(x0_ref, w, shift)
    assert rel_err(y, y_ref) < 1.5e-2
    assert torch.allclose(mom, ref.bn_stats(y, shift), rtol=1e-3, atol=1e-3)
    assert torch.allclose(mom, mom_ref, rtol=1e-2, atol=1e-2)
    dy = rnd(*y.shape, device=device)
    dw = torch.zeros(64, 7, 7, 3, dtype=torch.float32, device=device)
    dw_ref = torch.zeros_like(dw)
    hip.stem_wgrad(x0, dy, dw)
    ref.stem_wgrad(x0_ref, dy, dw_ref)
    assert rel_err(dw, dw_ref) < 5e-3


@pytest.mark.parametrize("C,rows", [(64, 5000), (256, 777), (2048, 98), (128, 64 * 3136)])
def test_bn_family(hip, device, C, rows):
    torch.manual_seed(C + rows)
    y = (torch.randn(rows, C, device=device) * 1.5 + torch.randn(C, device=device)).to(torch.bfloat16).view(1, rows, 1, C)
    shift = torch.randn(C, device=device) * 0.2
    mom = hip.bn_stats(y, shift)
    mom_ref = ref.bn_stats(y, shift)
    assert torch.allclose(mom, mom_ref, rtol=1e-3, atol=1e-3)

    gamma = torch.rand(C, device=device) + 0.5
    beta = torch.randn(C, device=device) * 0.1
    rm, rv = shift.clone(), torch.ones(C, device=device)
    rm_r, rv_r = shift.clone(), torch.ones(C, device=device)
    sc, sh, mu, istd = hip.bn_finalize(mom_ref.clone(), 1.0, gamma, beta, rm, rv, 0.1, 1e-5, rows)
    sc_r, sh_r, mu_r, istd_r = ref.bn_finalize(mom_ref.clone(), 1.0, gamma, beta, rm_r, rv_r, 0.1, 1e-5, rows)
    for a, b in ((sc, sc_r), (sh, sh_r), (mu, mu_r), (istd, istd_r), (rm, rm_r), (rv, rv_r)):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
    esc, esh = hip.bn_eval_affine(gamma, beta, rm, rv, 1e-5)
    esc_r, esh_r = ref.bn_eval_affine(gamma, beta, rm, rv, 1e-5)
    assert torch.allclose(esc, esc_r, rtol=1e-5) and torch.allclose(esh, esh_r, rtol=1e-4, atol=1e-6)

    res = rnd(1, rows, 1, C, device=device)
    for r, relu in ((None, True), (res, True), (None, False), (res, False)):
        out = hip.bn_apply(y, sc_r, sh_r, r, relu)
        out_ref = ref.bn_apply(y, sc_r, sh_r, r, relu)
        assert rel_err(out, out_ref) < 1e-2

    if C <= 2048 and 256 % (C // 8) == 0:
        y2 = rnd(1, rows, 1, C, device=device)
        sc2, sh2 = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.1
        for relu in (True, False):
            out = hip.bn_apply2(y, sc_r, sh_r, y2, sc2, sh2, relu)
            assert rel_err(out, ref.bn_apply2(y, sc_r, sh_r, y2, sc2, sh2, relu)) < 1e-2

    g = rnd(1, rows, 1, C, device=device)
    msrc = rnd(1, rows, 1, C, device=device)
    for mode, m in ((ref.MASK_NONE, None), (ref.MASK_TENSOR, msrc), (ref.MASK_RECOMPUTE, None)):
        dg, db = torch.zeros(C, device=device), torch.zeros(C, device=device)
        dg_r, db_r = torch.zeros(C, device=device), torch.zeros(C, device=device)
        sums = hip.bn_bwd_reduce(g, y, mu_r, istd_r, sc_r, sh_r, m, mode, dg, db)
        sums_ref = ref.bn_bwd_reduce(g, y, mu_r, istd_r, sc_r, sh_r, m, mode, dg_r, db_r)
        tol = 2e-3 * max(rows ** 0.5, 1.0)
        assert torch.allclose(sums, sums_ref, rtol=2e-3, atol=tol)
        assert torch.allclose(dg, dg_r, rtol=2e-3, atol=tol) and torch.allclose(db, db_r, rtol=2e-3, atol=tol)
        dx = hip.bn_bwd_elemt(g, y, mu_r, istd_r, sc_r, sh_r, sums_ref, 1.0 / rows, m, mode)
        dx_ref = ref.bn_bwd_elemt(g, y, mu_r, istd_r, sc_r, sh_r, sums_ref, 1.0 / rows, m, mode)
        assert rel_err(dx, dx_ref) < 1.5e-2


@pytest.mark.parametrize("Ci,Co,H", [(64, 256, 14), (128, 64, 9), (256, 2048, 7)])
def test_fused_finalize_from_conv_accumulator(hip, device, Ci, Co, H):
    """conv (statistics left in the accumulator) + ONE fused kernel == conv + moments + finalize."""
    torch.manual_seed(Ci + Co)
    x = rnd(4, H, H, Ci, device=device)
    w = rnd(Co, 1, 1, Ci, device=device, scale=0.1)
    shift = torch.randn(Co, device=device) * 0.05
    gamma = torch.rand(Co, device=device) + 0.5
    beta = torch.randn(Co, device=device) * 0.1
    rows = 4 * H * H
    y, pend = hip.conv_fprop(x, w, 1, 1, 1, 0, stat_shift=shift, defer_stats=True)
    assert isinstance(pend, hip.PendingStats) and pend.count == rows
    rm, rv = shift.clone(), torch.ones(Co, device=device)
    out = hip.bn_finalize_fused(pend, None, 1.0, gamma, beta, rm, rv, 0.1, 1e-5, rows)
    assert float(pend.acc.abs().max()) == 0.0                       # the replicas were re-zeroed
    mom_ref = ref.bn_stats(y, shift)
    rm_r, rv_r = shift.clone(), torch.ones(Co, device=device)
    out_ref = ref.bn_finalize(mom_ref, 1.0, gamma, beta, rm_r, rv_r, 0.1, 1e-5, rows)
    for a, b in zip(list(out) + [rm, rv], list(out_ref) + [rm_r, rv_r]):
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-4)
    # the deferred and the immediate form of the same conv agree
    y2, mom = hip.conv_fprop(x, w, 1, 1, 1, 0, stat_shift=shift)
    assert torch.equal(y, y2) and torch.allclose(hip.stats_to_moments(mom), mom_ref, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("Ci,Co,k,stride,pad,H", [(64, 256, 1, 1, 0, 14), (128, 128, 3, 2, 1, 16), (256, 64, 3, 1, 1, 9),
                                                  (512, 1024, 1, 2, 0, 14)])
@pytest.mark.parametrize("mode", ["none", "recompute"])
def test_dgrad_epilogue_accumulates_bn_backward_sums(hip, device, Ci, Co, k, stride, pad, H, mode):
    """conv_dgrad(bwd_stats=...) + bn_bwd_reduce(prereduced=True) == conv_dgrad + bn_bwd_reduce."""
    torch.manual_seed(Ci + Co + k)
    B = 3
    Ho = (H + 2 * pad - k) // stride + 1
    dy = rnd(B, Ho, Ho, Co, device=device)
    w = rnd(Co, k, k, Ci, device=device, scale=0.05)
    wt = torch.empty(Ci, k, k, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    y = rnd(B, H, H, Ci, device=device)                       # raw conv output of the consuming BN layer
    mean, istd = torch.randn(Ci, device=device) * 0.1, torch.rand(Ci, device=device) + 0.5
    sc, sh = torch.rand(Ci, device=device) + 0.5, torch.randn(Ci, device=device) * 0.2
    m = ref.MASK_NONE if mode == "none" else ref.MASK_RECOMPUTE
    g_plain = hip.conv_dgrad(dy, w, wt, (B, H, H, Ci), k, k, stride, pad)
    dg0, db0 = torch.zeros(Ci, device=device), torch.zeros(Ci, device=device)
    sums_ref = hip.bn_bwd_reduce(g_plain, y, mean, istd, sc, sh, None, m, dg0, db0)
    g_fused = hip.conv_dgrad(dy, w, wt, (B, H, H, Ci), k, k, stride, pad, bwd_stats=(y, mean, sc, sh, m))
    dg1, db1 = torch.zeros(Ci, device=device), torch.zeros(Ci, device=device)
    sums = hip.bn_bwd_reduce(g_fused, y, mean, istd, sc, sh, None, m, dg1, db1, prereduced=True)
    assert torch.equal(g_plain, g_fused)
    tol = 1e-3 * max(float(sums_ref.abs().max()), 1.0)
    assert torch.allclose(sums, sums_ref, rtol=1e-3, atol=tol)
    assert torch.allclose(dg1, dg0, rtol=1e-3, atol=tol) and torch.allclose(db1, db0, rtol=1e-3, atol=tol)


def test_batched_weight_transpose(hip, device):
    shapes = [(64, 9, 64), (256, 1, 64), (128, 9, 128), (2048, 1, 512), (24, 1, 40)]
    offs, total = [], 0
    for co, t, ci in shapes:
        offs.append(total)
        total += (co * t * ci + 127) // 128 * 128
    flat = rnd(total, device=device)
    wt = torch.zeros_like(flat)
    table = hip.make_transpose_table([(o, co, t, ci) for o, (co, t, ci) in zip(offs, shapes)], device)
    hip.transpose_weights_batched(flat, wt, *table)
    for o, (co, t, ci) in zip(offs, shapes):
        n = co * t * ci
        ref_t = flat[o:o + n].view(co, t, ci).permute(2, 1, 0).contiguous().view(-1)
        assert torch.equal(wt[o:o + n], ref_t)


@pytest.mark.parametrize("B,H,C", [(2, 112, 64), (3, 30, 64), (1, 8, 128), (3, 32, 128), (2, 16, 64)])
def test_pooling(hip, device, B, H, C):
    torch.manual_seed(B + H)
    x = torch.relu(rnd(B, H, H, C, device=device))       # many exact ties at 0, like post-ReLU activations
    out, idx = hip.maxpool_fwd(x)
    out_ref, idx_ref = ref.maxpool_fwd(x)
    assert torch.equal(out, out_ref)
    dout = rnd(*out.shape, device=device)
    dx = hip.maxpool_bwd(dout, idx, tuple(x.shape))
    dx_ref = ref.maxpool_bwd(dout, idx_ref, tuple(x.shape))
    assert rel_err(dx, dx_ref) < 1e-2
    raw = rnd(B, H, H, C, device=device)
    sc, sh = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.3
    fo, fi = hip.bn_relu_maxpool_fwd(raw, sc, sh)
    ro, ri = hip.maxpool_fwd(hip.bn_apply(raw, sc, sh, None, True))
    assert torch.equal(fo, ro) and torch.equal(fi, ri)
    p = hip.avgpool_fwd(x)
    assert rel_err(p, ref.avgpool_fwd(x)) < 1e-2
    dp = rnd(B, C, device=device)
    assert rel_err(hip.avgpool_bwd(dp, H, H), ref.avgpool_bwd(dp, H, H)) < 1e-2
    mk = rnd(B, H, H, C, device=device)
    assert rel_err(hip.avgpool_bwd(dp, H, H, mk), ref.avgpool_bwd(dp, H, H, mk)) < 1e-2


@pytest.mark.parametrize("C,H,stride,k", [(256, 14, 1, 1), (128, 16, 2, 3), (64, 9, 1, 3)])
def test_packed_relu_mask_roundtrip(hip, device, C, H, stride, k):
    """bn_apply / bn_apply2 write the ReLU mask of their output as bits; conv_dgrad applies it."""
    torch.manual_seed(C + H)
    B = 3
    y, res = rnd(B, H, H, C, device=device), rnd(B, H, H, C, device=device)
    scale, shift = torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.3
    bits = hip.new_relu_bits(y)
    out = hip.bn_apply(y, scale, shift, res, True, bits_out=bits)
    assert torch.equal(out, hip.bn_apply(y, scale, shift, res, True))
    want = torch.empty_like(bits)
    ref._pack_bits(out, want)
    assert torch.equal(bits, want) and torch.equal(ref._unpack_bits(bits), out > 0)
    bits2 = hip.new_relu_bits(y)
    out2 = hip.bn_apply2(y, scale, shift, res, scale * 0.5, shift, True, bits_out=bits2)
    ref._pack_bits(out2, want)
    assert torch.equal(bits2, want)
    # data gradient masked by the bits == masked by the tensor
    Co = 64
    w = rnd(Co, k, k, C, device=device, scale=0.05)
    wt = torch.empty(C, k, k, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    dy = rnd(B, Ho, Ho, Co, device=device)
    a = hip.conv_dgrad(dy, w, wt, tuple(out.shape), k, k, stride, pad, addend=res, relu_mask=out)
    b = hip.conv_dgrad(dy, w, wt, tuple(out.shape), k, k, stride, pad, addend=res, relu_bits=bits)
    # (the tensor form runs on the register-staged kernel, which rounds the convolution result to bf16 before it adds;
    # the packed form rounds the sum once: zeros where the mask is off, values within one rounding)
    off = ~(out > 0)
    assert bool((a[off] == 0).all()) and bool((b[off] == 0).all()) and rel_err(a, b) < 8e-3
    c = ref.conv_dgrad(dy, w, None, tuple(out.shape), k, k, stride, pad, addend=res, relu_bits=bits)
    assert rel_err(b, c) < 1.5e-2
    # with fused backward statistics the mask is accepted in packed form only
    mean = torch.zeros(C, device=device)
    d = hip.conv_dgrad(dy, w, wt, tuple(out.shape), k, k, stride, pad, addend=res, relu_bits=bits,
                       bwd_stats=(y, mean, None, None, ref.MASK_NONE))
    hip.bn_bwd_reduce(d, y, mean, torch.ones(C, device=device), None, None, None, ref.MASK_NONE, prereduced=True)
    assert torch.equal(d, b)
    with pytest.raises(ValueError):
        hip.conv_dgrad(dy, w, wt, tuple(out.shape), k, k, stride, pad, addend=res, relu_mask=out,
                       bwd_stats=(y, mean, None, None, ref.MASK_NONE))


@pytest.mark.parametrize("k,stride,pad", [(1, 1, 0), (3, 2, 1), (3, 1, 1)])
@pytest.mark.parametrize("fused", [False, True])
def test_half_resolution_addend(hip, device, k, stride, pad, fused):
    """conv_dgrad(addend_half=True): an addend stored at the even output pixels only equals the full
    tensor with zeros at the other pixels."""
    torch.manual_seed(k + stride)
    B, H, Ci, Co = 3, 12, 128, 64
    Ho = (H + 2 * pad - k) // stride + 1
    dy = rnd(B, Ho, Ho, Co, device=device)
    w = rnd(Co, k, k, Ci, device=device, scale=0.05)
    wt = torch.empty(Ci, k, k, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    half = rnd(B, H // 2, H // 2, Ci, device=device)
    full = torch.zeros(B, H, H, Ci, dtype=torch.bfloat16, device=device)
    full[:, ::2, ::2] = half
    y = rnd(B, H, H, Ci, device=device)
    bits = hip.new_relu_bits(y)
    ref._pack_bits(y, bits)
    mean = torch.randn(Ci, device=device) * 0.1
    kw = {"bwd_stats": (y, mean, None, None, ref.MASK_NONE)} if fused else {}

    def sums():
        return hip.bn_bwd_reduce(y, y, mean, torch.ones(Ci, device=device), None, None, None, ref.MASK_NONE,
                                 prereduced=True) if fused else None

    a = hip.conv_dgrad(dy, w, wt, (B, H, H, Ci), k, k, stride, pad, addend=full, relu_bits=bits, **kw)
    sa = sums()
    b = hip.conv_dgrad(dy, w, wt, (B, H, H, Ci), k, k, stride, pad, addend=half, addend_half=True, relu_bits=bits, **kw)
    sb = sums()
    assert torch.equal(a, b)
    if fused:
        assert torch.allclose(sa, sb, rtol=1e-4, atol=1e-3)
    c = ref.conv_dgrad(dy, w, None, (B, H, H, Ci), k, k, stride, pad, addend=half, addend_half=True, relu_bits=bits)
    assert rel_err(b, c) < 1.5e-2


@pytest.mark.parametrize("B,H,C", [(2, 112, 64), (3, 30, 64), (1, 9, 128), (3, 32, 128), (2, 16, 256)])
def test_bn_backward_through_fused_relu_maxpool(hip, device, B, H, C):
    """BatchNorm-backward passes fed with the POOLED gradient + winner indices against the
    materialised pooling gradient followed by the ordinary passes."""
    torch.manual_seed(B * H + C)
    y = rnd(B, H, H, C, device=device)
    gamma = torch.rand(C, device=device) + 0.5
    beta = torch.randn(C, device=device) * 0.2
    mean = y.float().mean(dim=(0, 1, 2))
    invstd = (y.float().var(dim=(0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    pooled, idx = hip.bn_relu_maxpool_fwd(y, scale, shift)
    dout = rnd(*pooled.shape, device=device)
    # the dense path: pooling gradient materialised (bf16), then the ordinary reduce / elementwise passes
    dense = hip.maxpool_bwd(dout, idx, tuple(y.shape))
    dg0, db0 = torch.zeros(C, device=device), torch.zeros(C, device=device)
    sums0 = hip.bn_bwd_reduce(dense, y, mean, invstd, scale, shift, None, ref.MASK_RECOMPUTE, dg0, db0)
    dx0 = hip.bn_bwd_elemt(dense, y, mean, invstd, scale, shift, sums0, 1.0 / (B * H * H), None, ref.MASK_RECOMPUTE)
    dg1, db1 = torch.zeros(C, device=device), torch.zeros(C, device=device)
    sums1 = hip.bn_bwd_reduce(dout, y, mean, invstd, scale, shift, None, ref.MASK_RECOMPUTE, dg1, db1, pool_idx=idx)
    dx1 = hip.bn_bwd_elemt(dout, y, mean, invstd, scale, shift, sums1, 1.0 / (B * H * H), None, ref.MASK_RECOMPUTE,
                           pool_idx=idx)
    tol = dict(rtol=2e-2, atol=2e-2 * sums0.abs().max().item())      # dense rounds the summed gradient to bf16
    assert torch.allclose(sums1, sums0, **tol) and torch.allclose(dg1, dg0, **tol) and torch.allclose(db1, db0, **tol)
    assert rel_err(dx1, dx0) < 2e-2
    # and against the fp32 reference ops on the same pooled gradient
    pooled_r, idx_r = ref.bn_relu_maxpool_fwd(y, scale, shift)
    sums_r = ref.bn_bwd_reduce(dout, y, mean, invstd, scale, shift, None, ref.MASK_RECOMPUTE, pool_idx=idx_r)
    dx_r = ref.bn_bwd_elemt(dout, y, mean, invstd, scale, shift, sums_r, 1.0 / (B * H * H), None, ref.MASK_RECOMPUTE,
                            pool_idx=idx_r)
    assert torch.allclose(sums1, sums_r, **tol)
    assert rel_err(dx1, dx_r) < 2e-2


@pytest.mark.parametrize("B,H,C", [(4, 112, 64), (3, 30, 64), (2, 18, 128)])
def test_stem_backward_sums_from_the_first_block_data_gradient(hip, device, B, H, C):
    """The pooling keeps y at its winners and the pooled tensor's sign mask; a data gradient produced with that mask
    and ``bwd_stats`` on the winners leaves the same BatchNorm-backward sums as the pass over the dense tensor."""
    torch.manual_seed(B + H + C)
    y = rnd(B, H, H, C, device=device)
    gamma = torch.rand(C, device=device) + 0.5
    beta = torch.randn(C, device=device) * 0.2
    mean = y.float().mean(dim=(0, 1, 2))
    invstd = (y.float().var(dim=(0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    pooled, idx, ywin = hip.bn_relu_maxpool_fwd(y, scale, shift, winners=True)
    p0, i0 = hip.bn_relu_maxpool_fwd(y, scale, shift)
    assert torch.equal(pooled, p0) and torch.equal(idx, i0)
    pr, ir, ywin_r = ref.bn_relu_maxpool_fwd(y, scale, shift, winners=True)
    assert torch.equal(pooled, pr) and torch.equal(ywin, ywin_r)
    assert torch.equal(pooled._pdt_relu_bits, pr._pdt_relu_bits)
    # a 1x1 convolution's data gradient lands on the pooled tensor (as the first bottleneck block's does)
    Co = 64
    w = rnd(Co, 1, 1, C, device=device, scale=0.1)
    wt = torch.empty(C, 1, 1, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    dy = rnd(B, pooled.shape[1], pooled.shape[2], Co, device=device)
    addend = rnd(*pooled.shape, device=device)
    bits = pooled._pdt_relu_bits
    g_plain = hip.conv_dgrad(dy, w, wt, tuple(pooled.shape), 1, 1, 1, 0, addend=addend, relu_bits=bits)
    g = hip.conv_dgrad(dy, w, wt, tuple(pooled.shape), 1, 1, 1, 0, addend=addend, relu_bits=bits,
                       bwd_stats=(ywin, mean, scale, shift, ref.MASK_NONE))
    assert torch.equal(g, g_plain)
    dg1, db1 = torch.zeros(C, device=device), torch.zeros(C, device=device)
    sums1 = hip.bn_bwd_reduce(g, y, mean, invstd, scale, shift, None, ref.MASK_RECOMPUTE, dg1, db1, prereduced=True,
                              pool_idx=idx)
    dg0, db0 = torch.zeros(C, device=device), torch.zeros(C, device=device)
    sums0 = hip.bn_bwd_reduce(g, y, mean, invstd, scale, shift, None, ref.MASK_RECOMPUTE, dg0, db0, pool_idx=idx)
    tol = dict(rtol=2e-3, atol=2e-3 * sums0.abs().max().item())
    assert torch.allclose(sums1, sums0, **tol) and torch.allclose(dg1, dg0, **tol) and torch.allclose(db1, db0, **tol)


@pytest.mark.parametrize("B,K,N", [(64, 2048, 1000), (7, 512, 10)])
def test_classifier_and_loss(hip, device, B, K, N):
    torch.manual_seed(B + N)
    a = rnd(B, K, device=device)
    w = rnd(N, K, device=device, scale=0.02)
    bias = torch.randn(N, device=device) * 0.1
    logits = hip.linear_fwd(a, w, bias)
    logits_ref = ref.linear_fwd(a, w, bias)
    assert rel_err(logits, logits_ref) < 1.5e-2
    labels = torch.randint(0, N, (B,), device=device)
    logits[0, labels[0]] += 30.0          # one extreme row
    loss, dlog = hip.softmax_ce(logits, labels, 1.0)
    loss_ref, dlog_ref = ref.softmax_ce(logits, labels, 1.0)
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * max(1.0, abs(loss_ref.item()))
    assert (dlog.float() - dlog_ref.float()).abs().max().item() < 2e-2 / B + 1e-4
    met = hip.ce_metrics(logits, labels)
    met_ref = ref.ce_metrics(logits, labels)
    assert torch.allclose(met, met_ref, rtol=2e-3, atol=1e-2)
    wt = hip.prepare_fc(w)
    dw, db = torch.zeros(N, K, device=device), torch.zeros(N, device=device)
    dw_r, db_r = torch.zeros(N, K, device=device), torch.zeros(N, device=device)
    da = hip.linear_bwd(a, dlog, w, wt, dw, db)
    da_ref = ref.linear_bwd(a, dlog, w, None, dw_r, db_r)
    assert rel_err(da, da_ref) < 2e-2
    assert rel_err(dw, dw_r) < 1e-2 and rel_err(db, db_r) < 1e-2


def test_fused_sgd_matches_torch(hip, device):
    torch.manual_seed(0)
    n = 64 * 1000
    p0 = torch.randn(n, device=device)
    p = p0.clone(); m = torch.zeros(n, device=device); pc = torch.zeros(n, device=device, dtype=torch.bfloat16)
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([tp], lr=0.1, momentum=0.9, weight_decay=1e-4)
    for step in range(100):
        grad = torch.randn(n, device=device)
        g = grad.clone() * 4.0
        hip.sgd_step(p, g, m, pc, 0.1, 0.9, 1e-4, 0.25, False, 0.0, step == 0)
        assert g.abs().sum() == 0
        tp.grad = grad.clone()
        opt.step()
    assert torch.allclose(p, tp.data, rtol=1e-4, atol=1e-5)
    assert torch.equal(pc, p.to(torch.bfloat16))


@pytest.mark.parametrize("S,O,B", [(256, 224, 16), (48, 32, 5), (64, 64, 3)])
def test_crop_resize_flip_kernel_matches_reference(hip, device, S, O, B):
    """GPU-side RandomResizedCrop + flip against the F.interpolate definition (uint8 results may differ by one
    count where the fp32 interpolation lands within rounding of .5)."""
    import numpy as np
    from pytorch_distributed_training_amd.data.packed import random_resized_crop_boxes
    g = torch.Generator().manual_seed(3)
    src = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    boxes = torch.from_numpy(random_resized_crop_boxes(B, S, np.random.default_rng(4)))
    boxes[0] = torch.tensor([0, 0, S, S, 0], dtype=torch.int32)          # whole image
    boxes[1] = torch.tensor([S - 9, S - 7, 9, 7, 1], dtype=torch.int32)  # tiny box in the far corner, flipped
    got = hip.crop_resize_flip(src.to(device), boxes, O).cpu()
    want = ref.crop_resize_flip(src, boxes, O)
    diff = (got.int() - want.int()).abs()
    assert got.shape == want.shape and diff.max().item() <= 1, diff.max().item()
    assert (diff > 0).float().mean().item() < 0.01
    if S == O:
        assert torch.equal(got[0], src[0])                                # identity box at the stored size
    bad = boxes.clone()
    bad[2] = torch.tensor([S - 4, 0, 8, 8, 0], dtype=torch.int32)         # leaves the image
    with pytest.raises(ValueError):
        hip.crop_resize_flip(src.to(device), bad, O)
