"""Block-entry data gradients with their full epilogue (residual-gradient addend, packed ReLU mask,
backward sums of the previous block's last BatchNorm) per tile shape, one MI355X.

    python bench/dgrad_epilogue.py [--batch 512]

These launches have a K of one or two steps and move 3-4x the bytes of their main loop in the
epilogue, so the best tile is the one that keeps the most bytes in flight, not the one with the
best MFMA efficiency.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (name, Cin (gradient produced), Cout (gradient consumed), H, count in ResNet-50)
SHAPES = [("layer1.conv1", 256, 64, 56, 2), ("layer2.conv1", 512, 128, 28, 3), ("layer3.conv1", 1024, 256, 14, 5),
          ("layer4.conv1", 2048, 512, 7, 2)]


# positive: tiles of the general implicit-GEMM kernel; 0: automatic; negative: tiles of the 1x1 streaming kernel
TILES = (1, 2, 5, 0, -1, -2, -3, -4)


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()
    from pytorch_distributed_training_amd.ops import hip, ref
    dev = torch.device("cuda", 0)
    B = args.batch
    print(f"{'layer':14s} " + " ".join(f"tile{t:>2d} us" for t in TILES) + "   GB moved   best TB/s")
    for name, Ci, Co, H, count in SHAPES:
        dy = torch.randn(B, H, H, Co, device=dev).to(torch.bfloat16)
        w = (torch.randn(Co, 1, 1, Ci, device=dev) * 0.05).to(torch.bfloat16)
        wt = torch.empty(Ci, 1, 1, Co, dtype=torch.bfloat16, device=dev)
        hip.transpose_weights(w, wt)
        addend = torch.randn(B, H, H, Ci, device=dev).to(torch.bfloat16)
        y = torch.randn(B, H, H, Ci, device=dev).to(torch.bfloat16)
        bits = hip.new_relu_bits(y)
        bits.random_(0, 256)
        mean = torch.zeros(Ci, device=dev)
        ts = []
        for tile in TILES:
            def fn():
                g = hip.conv_dgrad(dy, w, wt, (B, H, H, Ci), 1, 1, 1, 0, addend=addend, relu_bits=bits, tile=tile,
                                   bwd_stats=(y, mean, None, None, ref.MASK_NONE))
                hip.bn_bwd_reduce(g, y, mean, torch.ones_like(mean), None, None, None, ref.MASK_NONE, prereduced=True)
            ts.append(timeit(fn))
        gb = (dy.numel() + 3 * y.numel()) * 2 / 1e9 + bits.numel() / 1e9
        print(f"{name:14s} " + " ".join(f"{t:9.1f}" for t in ts) + f"   {gb:8.3f}   {gb / min(ts) * 1e3:8.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
