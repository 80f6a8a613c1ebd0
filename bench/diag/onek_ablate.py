"""Where does a one-K-step convolution launch spend its time?  Times the memory-bound launches of a bottleneck block
(stage-1 shapes at batch 1024 by default) on a library built with -DPDT_DMA_DIAG, with parts of the kernel switched
off through PDT_DMA_DIAG (results are then garbage; only the time counts):

    hipcc ... -DPDT_DMA_DIAG  (bench/diag/build_diag.sh)  ->  bench/ab_libs/libpdt_hip_diag.so
    PDT_HIP_LIB=bench/ab_libs/libpdt_hip_diag.so python bench/diag/onek_ablate.py [--batch 1024]

Each variant runs in a child process (the switch is read when the library loads its launch parameters).
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

VARIANTS = [(0, "full kernel"), (1, "no result stores"), (2, "weights loaded once per workgroup"), (4, "no MFMA"),
            (16, "no epilogue loads"), (1 | 16, "no stores, no epilogue loads"), (8 | 2, "operands loaded once"),
            (1 | 2 | 8 | 16, "MFMA + epilogue arithmetic only"), (32, "no epilogue"), (32 | 2 | 8, "main loop only, operands loaded once"),
            (32 | 2 | 8 | 4, "main loop without MFMA, operands loaded once")]


def child(batch: int, tile: int) -> None:
    import torch
    from pytorch_distributed_training_amd.ops import hip

    dev = torch.device("cuda", 0)

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

    out = []
    for (c, n, h) in [(64, 256, 56), (128, 512, 28), (256, 1024, 14)]:
        a = torch.randn(batch, h, h, c, device=dev).bfloat16()
        x = torch.randn(batch, h, h, n, device=dev).bfloat16()
        wf = (torch.randn(n, c, device=dev) * 0.1).bfloat16()
        w4 = wf.view(n, 1, 1, c).contiguous()
        shift = torch.zeros(n, device=dev)
        bits = torch.empty(batch, h, h, n // 8, dtype=torch.uint8, device=dev)
        y = torch.empty(batch, h, h, n, dtype=torch.bfloat16, device=dev)
        gb_plain = (a.numel() + y.numel()) * 2 / 1e9
        gb_tail = (a.numel() + 2 * y.numel() + y.numel() // 16) * 2 / 1e9
        t_plain = timeit(lambda: hip.conv_fprop(a, w4, 1, 1, 1, 0, out=y, tile=-tile))
        t_tail = timeit(lambda: hip.conv_fprop_fused_tail(a, wf, shift, addend=x, bits_out=bits, out=y, tile=tile))
        out.append(f"{c}->{n}@{h}: plain {t_plain:7.1f} us {gb_plain / t_plain * 1e3:5.2f} TB/s | "
                   f"tail+addend {t_tail:7.1f} us {gb_tail / t_tail * 1e3:5.2f} TB/s")
    print(" || ".join(out), flush=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--tile", type=int, default=0, help="tile of the LDS-DMA kernel (3 = 128 x 128, 4 = 128 x 64)")
    ap.add_argument("--variants", type=str, default="", help="comma-separated switch values (default: all)")
    args = ap.parse_args()
    if args.child:
        child(args.batch, args.tile)
        return 0
    only = {int(v) for v in args.variants.split(",") if v}
    for bits, name in VARIANTS:
        if only and bits not in only:
            continue
        env = dict(os.environ, PDT_DMA_DIAG=str(bits))
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--batch", str(args.batch), "--tile", str(args.tile)], env=env,
                             capture_output=True, text=True, timeout=170)
        line = [ln for ln in res.stdout.splitlines() if "->" in ln]
        print(f"{bits:3d} {name:36s} {line[-1] if line else 'FAILED: ' + res.stderr[-300:]}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
