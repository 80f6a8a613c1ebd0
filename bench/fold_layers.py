"""Stand-alone times of the pieces of a bottleneck block's last-unit backward, plain against folded
(ops.bn_fold_prepare / conv_dgrad_folded / conv_wgrad_folded), on the four ResNet-50 stage shapes:

    python bench/fold_layers.py [--batch 512] [--iters 20]

plain:   bn_bwd_elemt -> conv_dgrad (+ backward sums of the consumer) , conv_wgrad
folded:  bn_fold_prepare -> conv_dgrad_folded (+ sums) , column means + Gram + scaled weight gradient + correction
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_training_amd.ops import hip  # noqa: E402
from pytorch_distributed_training_amd.ops.ref import MASK_NONE, MASK_RECOMPUTE  # noqa: E402

STAGES = [("layer1", 56, 64, 256), ("layer2", 28, 128, 512), ("layer3", 14, 256, 1024), ("layer4", 7, 512, 2048)]


def timed(fn, iters):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.batch
    print(f"batch {B}; microseconds per call")
    print(f"{'stage':8}{'elemt':>9}{'dgrad':>9}{'wgrad':>9}{'| prep':>9}{'dgrad_f':>9}{'means':>9}{'gram':>9}"
          f"{'wgrad_s':>9}{'corr':>9}{'| plain':>10}{'folded':>9}")
    for name, H, C, N in STAGES:
        torch.manual_seed(0)
        a = torch.relu(torch.randn(B, H, H, C, device=dev)).to(torch.bfloat16)
        w = (torch.randn(N, 1, 1, C, device=dev) * (2.0 / C) ** 0.5).to(torch.bfloat16)
        wt = torch.empty(C, 1, 1, N, dtype=torch.bfloat16, device=dev)
        hip.transpose_weights(w, wt)
        y, _ = hip.conv_fprop(a, w, 1, 1, 1, 0)
        g = (torch.randn(B, H, H, N, device=dev) * 0.1).to(torch.bfloat16)
        mean, invstd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
        scale, shift = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1
        sums = torch.randn(2, N, device=dev)
        inv = 1.0 / (B * H * H)
        yp = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
        pm, psc, psh = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        bst = (yp, pm, psc, psh, MASK_RECOMPUTE)
        d = torch.empty_like(y)
        dx = torch.empty_like(a)
        dw = torch.zeros(N, 1, 1, C, dtype=torch.float32, device=dev)
        rows = B * H * H
        t = {}
        t["elemt"] = timed(lambda: hip.bn_bwd_elemt(g, y, mean, invstd, scale, shift, sums, inv, None, MASK_NONE, out=d),
                           args.iters)
        t["dgrad"] = timed(lambda: hip.conv_dgrad(d, w, wt, tuple(a.shape), 1, 1, 1, 0, bwd_stats=bst, out=dx), args.iters)
        t["wgrad"] = timed(lambda: hip.conv_wgrad(a, d, 1, 1, 1, 0, dw), args.iters)
        fold = hip.bn_fold_prepare(sums, scale, invstd, mean, inv, w, wt)
        t["prep"] = timed(lambda: hip.bn_fold_prepare(sums, scale, invstd, mean, inv, w, wt), args.iters)
        t["dgrad_f"] = timed(lambda: hip.conv_dgrad_folded(g, a, fold, bwd_stats=bst, out=dx), args.iters)
        acc = hip._accumulator(dev, C, "fold")
        mom = torch.empty(2, C, device=dev)

        def means():
            hip._lib.pdt_bn_stats_accum(a.data_ptr(), None, acc.data_ptr(), rows, C, hip._stream())
            hip._lib.pdt_bn_moments(acc.data_ptr(), mom.data_ptr(), C, 1.0 / rows, hip._stream())
        t["means"] = timed(means, args.iters)
        gram = torch.zeros(C, C, device=dev)
        taps = hip._one_tap()
        t["gram"] = timed(lambda: hip._wgrad(a, a, gram, rows, H, H, C, C, H, H, 1, C, C, C, taps), args.iters)
        t["wgrad_s"] = timed(lambda: hip._wgrad(g, a, dw, rows, H, H, C, C, H, H, 1, N, N, C, taps,
                                                row_scale=fold.coef[0]), args.iters)
        t["corr"] = timed(lambda: hip._lib.pdt_bnfold_wgrad(dw.data_ptr(), w.data_ptr(), gram.data_ptr(), mom.data_ptr(),
                                                            float(rows), fold.coef.data_ptr(), C, N, hip._stream()),
                          args.iters)
        plain = t["elemt"] + t["dgrad"] + t["wgrad"]
        folded = t["prep"] + t["dgrad_f"] + t["means"] + t["gram"] + t["wgrad_s"] + t["corr"]
        print(f"{name:8}{t['elemt']:9.1f}{t['dgrad']:9.1f}{t['wgrad']:9.1f}{t['prep']:9.1f}{t['dgrad_f']:9.1f}"
              f"{t['means']:9.1f}{t['gram']:9.1f}{t['wgrad_s']:9.1f}{t['corr']:9.1f}{plain:10.1f}{folded:9.1f}")


if __name__ == "__main__":
    main()
