"""Cost of the fused BatchNorm-statistics epilogue of the implicit-GEMM kernel.

Times write-heavy 1x1 convolutions (few input channels, many output channels: the GEMMs whose
time is set by the output stream) with and without the in-kernel statistics, and prints the
achieved HBM rate next to each.  Run on a GPU box:  python bench/epilogue_cost.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_training_amd.ops import hip  # noqa: E402


def timeit(fn, iters=30, warmup=5):
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
    batch = 256
    for (ci, co, h) in [(64, 256, 56), (128, 512, 28), (256, 1024, 14), (256, 64, 56), (512, 2048, 7)]:
        x = torch.randn(batch, h, h, ci, device=dev).bfloat16()
        w = (torch.randn(co, 1, 1, ci, device=dev) * 0.1).bfloat16()
        shift = torch.zeros(co, device=dev)
        nbytes = (x.numel() + batch * h * h * co) * 2
        for stats in (False, True):
            fn = (lambda: hip.conv_fprop(x, w, 1, 1, 1, 0, stat_shift=shift)) if stats else \
                 (lambda: hip.conv_fprop(x, w, 1, 1, 1, 0))
            us = timeit(fn)
            print(f"1x1 {ci:4d}->{co:4d} @{h:2d}^2  stats={int(stats)}  {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
