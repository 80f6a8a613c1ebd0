"""Which part of the folded data-gradient launch costs time: plain kernel / FOLD instantiation with accumulator start
values only / weight rows of the folded length / the second operand (layer3 and layer4 shapes, batch 512)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.ops import hip  # noqa: E402
from pytorch_distributed_training_amd.ops.ref import MASK_RECOMPUTE  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / iters


dev = torch.device("cuda", 0)
B = 512
for H, C, N in ((14, 256, 1024), (7, 512, 2048), (28, 128, 512)):
    g = (torch.randn(B, H, H, N, device=dev) * 0.1).to(torch.bfloat16)
    a = torch.relu(torch.randn(B, H, H, C, device=dev)).to(torch.bfloat16)
    wt = (torch.randn(C, N, device=dev) * 0.05).to(torch.bfloat16)
    for pad in (0, 64, 128):
        wf = (torch.randn(C, N + C + pad, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(C, device=dev)
        yp = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
        pm, psc, psh = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        acc = hip._accumulator(dev, C, "bwd")
        dx = torch.empty_like(a)
        taps = hip._one_tap()
        M = B * H * H
        p = lambda t: None if t is None else t.data_ptr()

        def run(wgt, wrow, in2, b, bst=True, tile=0):
            rc = hip._real.pdt_conv_dma_ex(p(g), p(wgt), p(dx), None, 0, None, p(acc) if bst else None, None,
                                           p(yp) if bst else None, p(pm) if bst else None, p(psc) if bst else None,
                                           p(psh) if bst else None, M, H, H, N, N, H, H, 1, H, H, 1, 0, 0, C, wrow, 1,
                                           taps.offy, taps.offx, taps.widx, tile, p(in2), C if in2 is not None else 0,
                                           C if in2 is not None else 0, p(b), hip._stream())
            assert rc == 0, rc
        wr = N + C + pad
        res = {
            "plain": timed(lambda: run(wt, N, None, None)),
            "plain,wrow": timed(lambda: run(wf, wr, None, None)),
            "bias": timed(lambda: run(wt, N, None, bias)),
            "bias,wrow": timed(lambda: run(wf, wr, None, bias)),
            "in2": timed(lambda: run(wf, wr, a, None)),
            "in2+bias": timed(lambda: run(wf, wr, a, bias)),
            "in2+bias nostat": timed(lambda: run(wf, wr, a, bias, bst=False)),
            "plain nostat": timed(lambda: run(wt, N, None, None, bst=False)),
        }
        print(f"H{H} C{C} N{N} wrow {wr}: " + "  ".join(f"{k} {v:.1f}" for k, v in res.items()))
