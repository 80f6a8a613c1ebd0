"""Per-layer convolution benchmark on one MI355X: every distinct ResNet-50 convolution shape,
forward / data-gradient / weight-gradient, HIP kernels vs the stock PyTorch (MIOpen) path.

    python bench/conv_layers.py [--batch 256] [--ops fprop,dgrad,wgrad] [--miopen 1] [--csv out.csv]

For each (layer, op) it prints the time, the achieved TFLOP/s, the "ideal" HBM traffic
(each operand read once, the result written once) and the bandwidth that traffic would
correspond to -- i.e. how far a layer is from its own roofline.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (id, Cin, Cout, k, stride, pad, H_in, count in ResNet-50)
LAYERS = [
    ("a1", 64, 64, 1, 1, 0, 56, 1), ("a2", 64, 64, 3, 1, 1, 56, 3), ("a3", 64, 256, 1, 1, 0, 56, 4),
    ("a4", 256, 64, 1, 1, 0, 56, 2), ("b1", 256, 128, 1, 1, 0, 56, 1), ("b2", 128, 128, 3, 2, 1, 56, 1),
    ("b3", 128, 512, 1, 1, 0, 28, 4), ("b4", 256, 512, 1, 2, 0, 56, 1), ("b5", 512, 128, 1, 1, 0, 28, 3),
    ("b6", 128, 128, 3, 1, 1, 28, 3), ("c1", 512, 256, 1, 1, 0, 28, 1), ("c2", 256, 256, 3, 2, 1, 28, 1),
    ("c3", 256, 1024, 1, 1, 0, 14, 6), ("c4", 512, 1024, 1, 2, 0, 28, 1), ("c5", 1024, 256, 1, 1, 0, 14, 5),
    ("c6", 256, 256, 3, 1, 1, 14, 5), ("d1", 1024, 512, 1, 1, 0, 14, 1), ("d2", 512, 512, 3, 2, 1, 14, 1),
    ("d3", 512, 2048, 1, 1, 0, 7, 3), ("d4", 1024, 2048, 1, 2, 0, 14, 1), ("d5", 2048, 512, 1, 1, 0, 7, 2),
    ("d6", 512, 512, 3, 1, 1, 7, 2),
]


def timeit(fn, iters: int = 10, warmup: int = 3) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters * 1e3  # us


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ops", type=str, default="fprop,dgrad,wgrad")
    ap.add_argument("--miopen", type=int, default=1)
    ap.add_argument("--layers", type=str, default="")
    ap.add_argument("--csv", type=str, default="")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--wtile", type=int, default=0)
    ap.add_argument("--target-blocks", type=int, default=0)
    ap.add_argument("--atomic-mb", type=int, default=0)
    ap.add_argument("--stats", type=int, default=1)
    ap.add_argument("--blas", type=int, default=0, help="also time the plain library GEMM (1x1 stride-1 layers)")
    args = ap.parse_args()
    from pytorch_distributed_training_amd.ops import hip

    dev = torch.device("cuda", 0)
    B = args.batch
    ops = args.ops.split(",")
    want = set(args.layers.split(",")) if args.layers else None
    rows = []
    totals = {op: [0.0, 0.0] for op in ops}
    print(f"{'layer':5s} {'op':6s} {'hip us':>9s} {'TF/s':>7s} {'GB ideal':>9s} {'TB/s':>6s} {'miopen us':>10s} {'speedup':>8s}")
    for lid, Ci, Co, k, stride, pad, H, count in LAYERS:
        if want and lid not in want:
            continue
        Ho = (H + 2 * pad - k) // stride + 1
        x = torch.randn(B, H, H, Ci, device=dev).to(torch.bfloat16)
        w = (torch.randn(Co, k, k, Ci, device=dev) * (2.0 / (k * k * Ci)) ** 0.5).to(torch.bfloat16)
        dy = torch.randn(B, Ho, Ho, Co, device=dev).to(torch.bfloat16)
        wt = torch.empty(Ci, k, k, Co, dtype=torch.bfloat16, device=dev)
        hip.transpose_weights(w, wt)
        dw = torch.zeros(Co, k, k, Ci, dtype=torch.float32, device=dev)
        shift = torch.zeros(Co, device=dev)
        flops = 2.0 * B * Ho * Ho * Co * k * k * Ci
        xb, yb, wb = x.numel() * 2, dy.numel() * 2, w.numel() * 2
        # torch (MIOpen) operands, channels_last
        xt = x.permute(0, 3, 1, 2)
        wtch = w.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        dyt = dy.permute(0, 3, 1, 2)
        for op in ops:
            if op == "fprop":
                fn = lambda: hip.conv_fprop(x, w, k, k, stride, pad, shift if args.stats else None, tile=args.tile)
                ideal = xb + yb + wb
                ref = lambda: torch.nn.functional.conv2d(xt, wtch, None, stride, pad)
            elif op == "dgrad":
                fn = lambda: hip.conv_dgrad(dy, w, wt, tuple(x.shape), k, k, stride, pad, tile=args.tile)
                ideal = xb + yb + wb
                ref = lambda: torch.nn.grad.conv2d_input(xt.shape, wtch, dyt, stride, pad)
            else:
                fn = lambda: hip.conv_wgrad(x, dy, k, k, stride, pad, dw, tile=args.wtile,
                                            target_blocks=args.target_blocks, atomic_budget_mb=args.atomic_mb)
                ideal = xb + yb + dw.numel() * 4
                ref = lambda: torch.nn.grad.conv2d_weight(xt, wtch.shape, dyt, stride, pad)
            t = timeit(fn)
            tm = timeit(ref, iters=5, warmup=2) if args.miopen else float("nan")
            blas = ""
            if args.blas and k == 1 and stride == 1:
                x2, y2, w2 = x.view(-1, Ci), dy.view(-1, Co), w.view(Co, Ci)
                gemm = {"fprop": lambda: torch.mm(x2, w2.t()), "dgrad": lambda: torch.mm(y2, w2),
                        "wgrad": lambda: torch.mm(y2.t(), x2)}[op]
                tb = timeit(gemm)
                blas = f"  blas {tb:8.1f} us {flops / tb / 1e6:7.1f} TF/s  x{tb / t:5.2f}"
            print(f"{lid:5s} {op:6s} {t:9.1f} {flops / t / 1e6:7.1f} {ideal / 1e9:9.3f} {ideal / t / 1e6:6.2f} "
                  f"{tm:10.1f} {tm / t:8.2f}{blas}", flush=True)
            rows.append((lid, op, count, t, flops / t / 1e6, ideal / 1e9, ideal / t / 1e6, tm))
            totals[op][0] += t * count
            totals[op][1] += tm * count
    print("\nweighted by layer count (one ResNet-50 pass):")
    for op, (th, tm) in totals.items():
        print(f"  {op:6s} hip {th / 1e3:8.2f} ms   miopen {tm / 1e3:8.2f} ms")
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("layer,op,count,hip_us,tflops,ideal_gb,ideal_tbps,miopen_us\n")
            for r in rows:
                f.write(",".join(str(v) for v in r) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
