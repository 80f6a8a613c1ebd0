"""Achieved HBM rate of the BatchNorm elementwise / reduction passes on the ResNet-50 activation shapes.

    python bench/bn_bandwidth.py        (on a GPU box)

Each line: op, activation shape, time, bytes moved / time.  These passes have no reuse, so the rate
against the ~8 TB/s HBM3E peak is the whole story.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_training_amd.ops import hip  # noqa: E402
from pytorch_distributed_training_amd.ops.ref import MASK_NONE, MASK_RECOMPUTE  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


def main():
    dev = torch.device("cuda", 0)
    B = 256
    for (C, H) in [(64, 112), (256, 56), (64, 56), (512, 28), (128, 28), (1024, 14), (2048, 7)]:
        y = torch.randn(B, H, H, C, device=dev).bfloat16()
        res = torch.randn(B, H, H, C, device=dev).bfloat16()
        g = torch.randn(B, H, H, C, device=dev).bfloat16()
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        mu, istd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        sums = torch.randn(2, C, device=dev)
        nb = y.numel() * 2
        rows = [
            ("bn_apply relu", lambda: hip.bn_apply(y, sc, sh, None, True), 2 * nb),
            ("bn_apply res+relu", lambda: hip.bn_apply(y, sc, sh, res, True), 3 * nb),
            ("bn_bwd_reduce none", lambda: hip.bn_bwd_reduce(g, y, mu, istd, sc, sh, None, MASK_NONE), 2 * nb),
            ("bn_bwd_reduce recompute", lambda: hip.bn_bwd_reduce(g, y, mu, istd, sc, sh, None, MASK_RECOMPUTE), 2 * nb),
            ("bn_bwd_elemt none", lambda: hip.bn_bwd_elemt(g, y, mu, istd, sc, sh, sums, 1e-5, None, MASK_NONE), 3 * nb),
            ("bn_bwd_elemt recompute", lambda: hip.bn_bwd_elemt(g, y, mu, istd, sc, sh, sums, 1e-5, None, MASK_RECOMPUTE), 3 * nb),
        ]
        for name, fn, nbytes in rows:
            us = timeit(fn)
            print(f"{name:24s} [{B},{H},{H},{C:4d}]  {us:8.1f} us  {nbytes / us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
