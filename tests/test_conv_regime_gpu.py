"""GPU tier: the convolution kernels in the regime training runs them in.

The per-shape oracle tests use batches of 1 and 5, where no workgroup of the persistent 128x128
kernel processes more than one tile.  Here the launches are large enough (and the tile is
forced) that workgroups loop over several tiles: the cross-tile operand prefetch, BatchNorm
statistics carried in registers from tile to tile, a change of the channel tile in the middle of
a workgroup's sequence, the backward-statistics epilogue (two rows per trip) with a packed ReLU
mask and a half-resolution addend, and the weight gradient at real split-K depths.  Oracle:
``ops/ref.py`` (fp32 ``F.conv2d`` and its gradients) on the same bf16 operands.
"""
import pytest
import torch

from pytorch_distributed_training_amd.ops import ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytorch_distributed_training_amd.ops import hip as hip_ops
    return hip_ops


def rnd(*shape, device, scale=1.0):
    return (torch.randn(*shape, device=device) * scale).to(torch.bfloat16)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def max_rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


# (B, H, Cin, Cout, k, stride, pad, tile): >= 2 tiles per resident workgroup at 2-3 workgroups per CU
MULTI_TILE = [
    (16, 56, 64, 256, 1, 1, 0, 1),      # 392 x 2 tiles of 128x128
    (16, 56, 64, 384, 1, 1, 0, 1),      # 3 channel tiles: the channel tile changes inside a workgroup's sequence
    (64, 28, 128, 512, 1, 1, 0, 1),     # 392 x 4
    (64, 28, 128, 128, 3, 1, 1, 1),     # 3x3, 18 K-steps per tile
    (64, 56, 64, 64, 3, 1, 1, 2),       # 1568 tiles of 128x64
    (16, 56, 256, 128, 3, 2, 1, 1),     # stride 2 (dgrad: four parity classes)
    (64, 14, 256, 1024, 1, 1, 0, 0),    # automatic choice
    (64, 14, 1024, 256, 1, 1, 0, 5),    # 64x128 tile, 16 K-steps
    (64, 28, 128, 512, 1, 1, 0, 3),     # 64x64 tile: 6272 tiles
    # the streaming DMA kernel (negative tile = its own tile numbering): 128x128 and 128x64 tiles, one to
    # sixteen K steps per tile, several tiles per workgroup, a channel-tile change inside a workgroup's sequence
    (16, 56, 64, 256, 1, 1, 0, -1), (16, 56, 64, 384, 1, 1, 0, -1), (64, 28, 128, 512, 1, 1, 0, -1),
    (16, 56, 256, 64, 1, 1, 0, -2), (64, 14, 1024, 256, 1, 1, 0, -1), (64, 14, 1024, 256, 1, 1, 0, -2),
    (16, 56, 256, 512, 1, 2, 0, -1), (33, 14, 512, 192, 1, 1, 0, -2), (5, 7, 2048, 512, 1, 1, 0, -1),
    # the same kernel with K slices of 64: 128x128 and 128x64 tiles (two-stage rings)
    (16, 56, 64, 256, 1, 1, 0, -3), (64, 28, 128, 512, 1, 1, 0, -3), (16, 56, 256, 64, 1, 1, 0, -4),
    (64, 14, 1024, 256, 1, 1, 0, -3), (33, 14, 512, 192, 1, 1, 0, -4), (3, 7, 64, 64, 1, 1, 0, -2),
    # ... and with tap tables: 3x3 with zero padding (taps outside the image must read zeros through the DMA), stride 2
    # (forward gather; data gradient in four output-parity launches), 14x14 and 7x7 images whose rows wrap inside a tile
    (64, 28, 128, 128, 3, 1, 1, -3), (64, 56, 64, 64, 3, 1, 1, -4), (16, 56, 256, 128, 3, 2, 1, -3),
    (64, 14, 256, 256, 3, 1, 1, -1), (33, 7, 512, 512, 3, 1, 1, -3), (16, 28, 256, 256, 3, 2, 1, -2),
    (5, 9, 64, 192, 3, 1, 1, -4), (2, 15, 128, 64, 3, 2, 1, -3),
]


@pytest.mark.parametrize("cfg", MULTI_TILE)
def test_fprop_multi_tile_with_fused_statistics(hip, device, cfg):
    B, H, Ci, Co, k, stride, pad, tile = cfg
    torch.manual_seed(sum(cfg))
    x = rnd(B, H, H, Ci, device=device)
    w = rnd(Co, k, k, Ci, device=device, scale=(2.0 / (k * k * Ci)) ** 0.5)
    shift = torch.randn(Co, device=device) * 0.1
    y, mom = hip.conv_fprop(x, w, k, k, stride, pad, shift, tile=tile)
    y_ref, _ = ref.conv_fprop(x, w, k, k, stride, pad, shift)
    assert max_rel(y, y_ref) < 1.5e-2 and rel_l2(y, y_ref) < 4e-3
    # statistics carried over the tiles == statistics of the stored tensor
    assert torch.allclose(mom, ref.bn_stats(y, shift), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("cfg", MULTI_TILE)
def test_dgrad_multi_tile_plain_and_backward_statistics(hip, device, cfg):
    B, H, Ci, Co, k, stride, pad, tile = cfg
    torch.manual_seed(sum(cfg) + 1)
    Ho = (H + 2 * pad - k) // stride + 1
    dy = rnd(B, Ho, Ho, Co, device=device)
    w = rnd(Co, k, k, Ci, device=device, scale=(2.0 / (k * k * Ci)) ** 0.5)
    wt = torch.empty(Ci, k, k, Co, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    shape = (B, H, H, Ci)
    addend = rnd(*shape, device=device)
    x_act = torch.relu(rnd(*shape, device=device))           # the tensor whose ReLU backward is applied
    bits = hip.new_relu_bits(x_act)
    ref._pack_bits(x_act, bits)
    dx = hip.conv_dgrad(dy, w, wt, shape, k, k, stride, pad, addend=addend, relu_bits=bits, tile=tile)
    dx_ref = ref.conv_dgrad(dy, w, None, shape, k, k, stride, pad, addend=addend, relu_bits=bits)
    assert max_rel(dx, dx_ref) < 1.5e-2 and rel_l2(dx, dx_ref) < 4e-3

    # the backward-statistics epilogue: same gradient, sums == a reduction pass over the result
    yb = rnd(*shape, device=device)
    mean, istd = torch.randn(Ci, device=device) * 0.1, torch.rand(Ci, device=device) + 0.5
    sc, sh = torch.rand(Ci, device=device) + 0.5, torch.randn(Ci, device=device) * 0.2
    for mode in (ref.MASK_NONE, ref.MASK_RECOMPUTE):
        fused = hip.conv_dgrad(dy, w, wt, shape, k, k, stride, pad, addend=addend, relu_bits=bits, tile=tile,
                               bwd_stats=(yb, mean, sc, sh, mode))
        dg1, db1 = torch.zeros(Ci, device=device), torch.zeros(Ci, device=device)
        sums = hip.bn_bwd_reduce(fused, yb, mean, istd, sc, sh, None, mode, dg1, db1, prereduced=True)
        assert torch.equal(fused, dx)
        dg0, db0 = torch.zeros(Ci, device=device), torch.zeros(Ci, device=device)
        sums_ref = ref.bn_bwd_reduce(dx, yb, mean, istd, sc, sh, None, mode, dg0, db0)
        tol = 2e-3 * float(sums_ref.abs().max())
        assert torch.allclose(sums, sums_ref, rtol=2e-3, atol=tol)
        assert torch.allclose(dg1, dg0, rtol=2e-3, atol=tol) and torch.allclose(db1, db0, rtol=2e-3, atol=tol)


@pytest.mark.parametrize("B,H,Ci,Co,tile", [(16, 56, 128, 256, 1), (64, 28, 256, 512, 1), (64, 28, 256, 512, 0),
                                            (16, 56, 128, 256, -1), (64, 28, 256, 512, -2), (3, 14, 512, 1024, -1)])
def test_first_conv_dgrad_of_a_downsampling_block(hip, device, B, H, Ci, Co, tile):
    """The data gradient that closes a stride-2 block: 1x1 product + half-resolution addend (the
    projection shortcut's gradient) + packed ReLU mask of the block input + backward sums of the
    previous block's last BatchNorm, over several tiles per workgroup."""
    torch.manual_seed(B + H + Ci)
    dy = rnd(B, H, H, Ci, device=device)
    w = rnd(Ci, 1, 1, Co, device=device, scale=(2.0 / Co) ** 0.5)       # conv Co -> Ci; the gradient has Co channels
    wt = torch.empty(Co, 1, 1, Ci, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    shape = (B, H, H, Co)
    half = rnd(B, H // 2, H // 2, Co, device=device)
    x_act = torch.relu(rnd(*shape, device=device))
    bits = hip.new_relu_bits(x_act)
    ref._pack_bits(x_act, bits)
    yb = rnd(*shape, device=device)
    mean, istd = torch.randn(Co, device=device) * 0.1, torch.rand(Co, device=device) + 0.5
    g = hip.conv_dgrad(dy, w, wt, shape, 1, 1, 1, 0, addend=half, addend_half=True, relu_bits=bits, tile=tile,
                       bwd_stats=(yb, mean, None, None, ref.MASK_NONE))
    dg1, db1 = torch.zeros(Co, device=device), torch.zeros(Co, device=device)
    sums = hip.bn_bwd_reduce(g, yb, mean, istd, None, None, None, ref.MASK_NONE, dg1, db1, prereduced=True)
    g_ref = ref.conv_dgrad(dy, w, None, shape, 1, 1, 1, 0, addend=half, addend_half=True, relu_bits=bits)
    assert max_rel(g, g_ref) < 1.5e-2 and rel_l2(g, g_ref) < 4e-3
    dg0, db0 = torch.zeros(Co, device=device), torch.zeros(Co, device=device)
    sums_ref = ref.bn_bwd_reduce(g, yb, mean, istd, None, None, None, ref.MASK_NONE, dg0, db0)
    tol = 2e-3 * float(sums_ref.abs().max())
    assert torch.allclose(sums, sums_ref, rtol=2e-3, atol=tol)
    assert torch.allclose(dg1, dg0, rtol=2e-3, atol=tol) and torch.allclose(db1, db0, rtol=2e-3, atol=tol)


@pytest.mark.parametrize("out_sums", [False, True])
@pytest.mark.parametrize("B,H,Ci,Co,tile", [(16, 56, 128, 256, 0), (3, 14, 512, 1024, 0), (5, 6, 64, 128, 0), (64, 28, 256, 512, 0),
                                            (7, 10, 64, 64, 0), (16, 56, 128, 256, -2)])
def test_half_resolution_addend_in_the_prefetching_epilogue(hip, device, B, H, Ci, Co, tile, out_sums):
    """The same data gradient without backward sums of a BatchNorm (the previous block's tail is folded): that launch
    takes the prefetching epilogue, which fetches the half-resolution addend through one descriptor per wave tile and
    per-lane offsets (odd pixels and rows past the end point past the descriptor and read zeros).  Tiles that span
    several small images, a partial last tile, both tile shapes, with and without the plain sums of the result."""
    torch.manual_seed(B * 7 + H + Ci)
    dy = rnd(B, H, H, Ci, device=device)
    w = rnd(Ci, 1, 1, Co, device=device, scale=(2.0 / Co) ** 0.5)
    wt = torch.empty(Co, 1, 1, Ci, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    shape = (B, H, H, Co)
    half = rnd(B, H // 2, H // 2, Co, device=device)
    x_act = torch.relu(rnd(*shape, device=device))
    bits = hip.new_relu_bits(x_act)
    ref._pack_bits(x_act, bits)
    g = hip.conv_dgrad(dy, w, wt, shape, 1, 1, 1, 0, addend=half, addend_half=True, relu_bits=bits, tile=tile,
                       out_sums=out_sums)
    g_ref = ref.conv_dgrad(dy, w, None, shape, 1, 1, 1, 0, addend=half, addend_half=True, relu_bits=bits)
    assert max_rel(g, g_ref) < 1.5e-2 and rel_l2(g, g_ref) < 4e-3
    # the addend is there at the even pixels and nowhere else
    plain = ref.conv_dgrad(dy, w, None, shape, 1, 1, 1, 0, relu_bits=bits)
    odd = (g.float() - plain.float())[:, 1::2, :, :]
    assert odd.abs().max() < 2e-2 * plain.float().abs().max()
    if out_sums:
        sums = hip.grad_sums(Co, device)
        want = g.float().reshape(-1, Co).sum(0)
        assert torch.allclose(sums, want, rtol=2e-3, atol=2e-3 * float(want.abs().max()))


WGRAD = [
    (16, 56, 64, 256, 1, 1, 0, 0), (64, 56, 64, 64, 3, 1, 1, 0), (64, 28, 128, 128, 3, 1, 1, 1),
    (64, 28, 512, 128, 1, 1, 0, 1), (64, 14, 256, 256, 3, 1, 1, 1), (64, 14, 1024, 256, 1, 1, 0, 2),
    (16, 56, 128, 128, 3, 2, 1, 0), (64, 7, 512, 2048, 1, 1, 0, 0), (64, 28, 256, 512, 1, 2, 0, 0),
    # tile 8: the LDS-DMA 128 x 128 kernel forced (linear 1x1, 3x3 with padding taps, stride 2, channel remainders)
    (64, 14, 1024, 256, 1, 1, 0, 8), (64, 14, 256, 256, 3, 1, 1, 8), (16, 56, 128, 128, 3, 2, 1, 8),
    (64, 28, 256, 512, 1, 2, 0, 8), (16, 28, 192, 320, 1, 1, 0, 8), (8, 14, 136, 200, 3, 1, 1, 8),
    # 64 channels on one side: the automatic choice is 128 wide on the other (64 x 128 / 128 x 64); tile 9 forced on
    # a 3x3 layer and with channel remainders
    (16, 56, 256, 64, 1, 1, 0, 0), (16, 56, 64, 256, 1, 1, 0, 0), (16, 28, 128, 64, 3, 1, 1, 9),
    (16, 28, 200, 72, 1, 1, 0, 9),
]


@pytest.mark.parametrize("cfg", WGRAD)
def test_wgrad_split_k_depths(hip, device, cfg):
    """Reduction lengths of 3 000 - 200 000 pixels: split-K with many partial sums per element."""
    B, H, Ci, Co, k, stride, pad, tile = cfg
    torch.manual_seed(sum(cfg) + 2)
    Ho = (H + 2 * pad - k) // stride + 1
    x = rnd(B, H, H, Ci, device=device)
    dy = rnd(B, Ho, Ho, Co, device=device)
    dw = torch.zeros(Co, k, k, Ci, dtype=torch.float32, device=device)
    dw_ref = torch.zeros_like(dw)
    hip.conv_wgrad(x, dy, k, k, stride, pad, dw, tile=tile)
    ref.conv_wgrad(x, dy, k, k, stride, pad, dw_ref)
    assert rel_l2(dw, dw_ref) < 2e-3 and max_rel(dw, dw_ref) < 5e-3
    # a forced number of K slices and the launch split over batch ranges accumulate the same result
    dw2 = torch.zeros_like(dw)
    hip.conv_wgrad(x[: B // 2], dy[: B // 2], k, k, stride, pad, dw2, tile=tile, target_blocks=1024)
    hip.conv_wgrad(x[B // 2:], dy[B // 2:], k, k, stride, pad, dw2, tile=tile, target_blocks=64)
    assert rel_l2(dw2, dw_ref) < 2e-3


def _guarded(shape, dtype, device, fill):
    """A tensor of ``shape`` in the middle of a larger allocation filled with ``fill``: (tensor, check) where check()
    says whether the bands in front of and behind it are untouched."""
    n = 1
    for s in shape:
        n *= s
    pad = 64 * 1024
    whole = torch.full((n + 2 * pad,), fill, dtype=dtype, device=device)
    t = whole[pad:pad + n].view(*shape)

    def untouched() -> bool:
        return bool((whole[:pad] == fill).all()) and bool((whole[pad + n:] == fill).all())

    return t, untouched


@pytest.mark.parametrize("B,H,C,N", [(3, 7, 64, 256), (5, 9, 128, 64), (1, 5, 64, 128), (13, 7, 256, 1024), (2, 10, 64, 64)])
def test_epilogue_stays_inside_its_tensors(hip, device, B, H, C, N):
    """Row counts that are no multiple of the 128-row tile (the last tile is partial, some wave tiles are empty): results
    and sign masks are written through buffer descriptors whose range check must end every tensor exactly -- the bytes
    around the outputs stay untouched, and the partial tile's rows are right."""
    torch.manual_seed(B * 100 + H)
    a = rnd(B, H, H, C, device=device)
    x = rnd(B, H, H, N, device=device)
    w = rnd(N, 1, 1, C, device=device, scale=0.1)
    shift = torch.randn(N, device=device) * 0.1
    # forward with fused statistics
    y, ok_y = _guarded((B, H, H, N), torch.bfloat16, device, 7.0)
    _, mom = hip.conv_fprop(a, w, 1, 1, 1, 0, stat_shift=shift, out=y)
    y_ref, mom_ref = ref.conv_fprop(a, w, 1, 1, 1, 0, stat_shift=shift)
    assert ok_y() and rel_l2(y, y_ref) < 1e-2
    assert torch.allclose(hip.stats_to_moments(mom), mom_ref, rtol=2e-2, atol=2e-2)
    # fused block tail: addend, ReLU, packed sign mask
    wf = w.view(N, C).contiguous()
    out, ok_o = _guarded((B, H, H, N), torch.bfloat16, device, 7.0)
    bits, ok_b = _guarded((B, H, H, N // 8), torch.uint8, device, 0x5a)
    hip.conv_fprop_fused_tail(a, wf, shift, addend=x, bits_out=bits, out=out)
    want_bits = torch.zeros(B, H, H, N // 8, dtype=torch.uint8, device=device)
    out_ref = ref.conv_fprop_fused_tail(a, wf, shift, addend=x, bits_out=want_bits)
    assert ok_o() and ok_b() and rel_l2(out, out_ref) < 1e-2
    assert torch.equal(ref._unpack_bits(bits), out > 0)
    # block-entry data gradient: addend, packed mask, per-channel sums of the result
    dy = rnd(B, H, H, N, device=device)
    wt = torch.empty(C, 1, 1, N, dtype=torch.bfloat16, device=device)
    hip.transpose_weights(w, wt)
    res = rnd(B, H, H, C, device=device)
    mbits = hip.new_relu_bits(res)
    mbits.copy_(torch.randint(0, 256, tuple(mbits.shape), device=device, dtype=torch.uint8))
    dx, ok_d = _guarded((B, H, H, C), torch.bfloat16, device, 7.0)
    hip.conv_dgrad(dy, w, wt, (B, H, H, C), 1, 1, 1, 0, addend=res, relu_bits=mbits, out=dx, out_sums=True)
    sums = hip.grad_sums(C, device)
    dx_ref = ref.conv_dgrad(dy, w, None, (B, H, H, C), 1, 1, 1, 0, addend=res, relu_bits=mbits)
    assert ok_d() and rel_l2(dx, dx_ref) < 1.5e-2
    want = dx.float().reshape(-1, C).sum(0)
    assert torch.allclose(sums, want, rtol=1e-3, atol=1e-3 * want.abs().max().item() + 1e-3)
