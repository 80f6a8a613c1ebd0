"""Per-dispatch roofline of the training step: minimum bytes and FLOPs of every library call of one
recorded step, joined with a rocprofv3 kernel trace of the same step.

    python bench/roofline.py dump OPLIST.json [--batch 1024] [--model ResNet50]     (on the GPU)
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...     (same batch / model)
    python bench/roofline.py join OPLIST.json TRACE.csv [--steps 4] [--top 40]

``dump`` builds the engine exactly as ``bench.py`` does, lets it record its command list and writes one
entry per library call, in launch order: kernel family, stream, the bytes the call must move at the
very least (every operand read once, every result written once, 2-byte activations) and its FLOPs.
``join`` matches those entries with the kernels of the trace by order (per queue) and prints, per
dispatch and per kernel class: time inside the running step (co-runners included), achieved TB/s
against the minimum bytes, achieved TFLOP/s.  A kernel far below ~4.5 TB/s and far below the
matrix rate is where time is being lost.
"""
import argparse
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _igemm_entry(name, a):
    if name == "pdt_conv_igemm":
        (inp, wgt, out, addend, addend_half, mask, mask_bits, bias, stat_acc, stat_shift, M, Hi, Wi, ps, Ci, Hg, Wg, IS,
         Ho, Wo, OS, oy0, ox0, N, ldo, wrow, T) = a[:27]
        bstat_y = None
    else:
        (inp, wgt, out, addend, addend_half, mask, mask_bits, stat_acc, bstat_y, mean, scale, shift, M, Hi, Wi, ps, Ci, Hg,
         Wg, IS, Ho, Wo, OS, oy0, ox0, N, ldo, wrow, T) = a[:29]
    images = M // max(Hg * Wg, 1)
    in_b = images * Hi * Wi * ps * 2
    if T == 1 and IS > 1:                      # a strided 1x1 product touches one input pixel per output pixel
        in_b = M * ps * 2
    b = in_b + N * wrow * 2 + M * N * 2
    if addend:
        b += M * N * 2 // (4 if addend_half else 1)
    if mask:
        b += M * N // 8 if mask_bits else M * N * 2
    if bstat_y:
        b += M * N * 2
    kind = "igemm+bstat" if bstat_y else ("igemm+addend" if addend else "igemm")
    return dict(kind=kind, bytes=b, flops=2 * M * N * T * Ci, desc=f"M{M} N{N} K{T}x{Ci}")


def _conv_dma_entry(a):
    (inp, w, out, addend, addend_half, mask_bits, stat_acc, stat_shift, bstat_y, mean, scale, shift, M, Hi, Wi, ps, Ci, Hg,
     Wg, IS, Ho, Wo, OS, oy0, ox0, N, wrow, T) = a[:28]
    images = M // max(Hg * Wg, 1)
    in_b = M * ps * 2 if (T == 1 and IS > 1) else images * Hi * Wi * ps * 2
    b = in_b + N * wrow * 2 + M * N * 2
    if addend:
        b += M * N * 2 // (4 if addend_half else 1)
    if mask_bits:
        b += M * N // 8
    if bstat_y:
        b += M * N * 2
    kind = "igemm+bstat" if bstat_y else ("igemm+addend" if addend else "igemm")
    return dict(kind=kind, bytes=b, flops=2 * M * N * T * Ci, desc=f"M{M} N{N} K{T}x{Ci} (dma)")


def _entry(name, a):
    if name in ("pdt_conv_igemm", "pdt_conv_igemm_bstats"):
        return _igemm_entry(name, a)
    if name == "pdt_conv_dma":
        return _conv_dma_entry(a)
    if name == "pdt_conv_dma_ex":            # + second operand (row-aligned, Ci2 channels) and accumulator start values
        e = _conv_dma_entry(a)
        M, N = a[12], a[25]
        in2, Ci2, ps2 = a[32], a[33], a[34]
        if in2:
            e["bytes"] += M * ps2 * 2
            e["flops"] += 2 * M * N * Ci2
            e["kind"] = e["kind"].replace("igemm", "igemm2")
            e["desc"] = e["desc"].replace(" (dma)", f"+{Ci2} (dma)")
        return e
    if name in ("pdt_conv_wgrad", "pdt_conv_wgrad_ex"):
        dy, inp, dw, M, Hi, Wi, ps, Ci, Hg, Wg, IS, N, ldy, wrow, T = a[:15]
        images = M // max(Hg * Wg, 1)
        return dict(kind="wgrad", bytes=M * ldy * 2 + images * Hi * Wi * ps * 2 + N * wrow * 4,
                    flops=2 * M * N * T * Ci, desc=f"M{M} N{N} K{T}x{Ci}")
    if name in ("pdt_bn_apply", "pdt_bn_apply_ex"):
        y, sc, sh, res, out, bits, rows, C, relu = a[:9]
        return dict(kind="bn_apply", bytes=rows * C * 2 * (3 if res else 2) + (rows * C // 8 if bits else 0), flops=0,
                    desc=f"rows{rows} C{C}")
    if name == "pdt_bn_apply2":
        rows, C = a[8], a[9]
        return dict(kind="bn_apply2", bytes=rows * C * 2 * 3 + (rows * C // 8 if a[7] else 0), flops=0, desc=f"rows{rows} C{C}")
    if name == "pdt_bn_bwd_elemt":
        rows, C = a[10], a[11]
        return dict(kind="bn_bwd_elemt", bytes=rows * C * 2 * (4 if a[2] else 3), flops=0, desc=f"rows{rows} C{C}")
    if name == "pdt_bn_bwd_reduce_accum":
        rows, C = a[7], a[8]
        return dict(kind="bn_bwd_reduce", bytes=rows * C * 2 * (3 if a[2] else 2), flops=0, desc=f"rows{rows} C{C}")
    if name == "pdt_bn_stats_accum":
        rows, C = a[3], a[4]
        return dict(kind="bn_stats", bytes=rows * C * 2, flops=0, desc=f"rows{rows} C{C}")
    if name == "pdt_bn_relu_maxpool_fwd":
        B, H, W, C, Ho, Wo = a[5:11]
        return dict(kind="maxpool_fwd", bytes=B * H * W * C * 2 + B * Ho * Wo * C * 3, flops=0, desc=f"B{B} {H}x{W}x{C}")
    if name == "pdt_bn_relu_maxpool_fwd_winners":
        B, H, W, C, Ho, Wo = a[7:13]
        return dict(kind="maxpool_fwd", bytes=B * H * W * C * 2 + B * Ho * Wo * (C * 5 + C // 8), flops=0,
                    desc=f"B{B} {H}x{W}x{C} +winners")
    if name in ("pdt_pool_bn_bwd_reduce_accum", "pdt_pool_bn_bwd_elemt"):
        off = 7 if name.endswith("accum") else 10
        B, H, W, C, Ho, Wo = a[off:off + 6]
        wr = 0 if name.endswith("accum") else B * H * W * C * 2
        return dict(kind="pool_bn_bwd", bytes=B * H * W * C * 2 + B * Ho * Wo * C * 3 + wr, flops=0, desc=f"B{B} {H}x{W}x{C}")
    if name == "pdt_input_pad":
        B, H, W, Hp, Wp = a[2:7]
        return dict(kind="input_pad", bytes=B * H * W * 3 + B * Hp * Wp * 8, flops=0, desc=f"B{B} {H}x{W}")
    if name == "pdt_avgpool_fwd":
        B, HW, C = a[2:5]
        return dict(kind="avgpool", bytes=B * HW * C * 2, flops=0, desc="")
    if name == "pdt_avgpool_bwd":
        B, HW, C = a[3:6]
        return dict(kind="avgpool", bytes=B * HW * C * 2 * (2 if a[1] else 1), flops=0, desc="")
    if name == "pdt_sgd_step":
        return dict(kind="sgd", bytes=a[4] * 26, flops=0, desc=f"n{a[4]}")
    return dict(kind=name[4:], bytes=0, flops=0, desc="")


# library call -> substring of the kernel name it launches (None: no kernel, e.g. a stream wait)
KERNEL_OF = {
    "pdt_conv_igemm": "igemm_kernel", "pdt_conv_igemm_bstats": "igemm_kernel", "pdt_conv_wgrad": "wgrad_",
    "pdt_conv_dma": "conv_dma_kernel", "pdt_conv_dma_ex": "conv_dma_kernel", "pdt_conv_wgrad_ex": "wgrad_",
    "pdt_bnfold_prepare": "bnfold_prepare_kernel", "pdt_bnfold_wgrad": "bnfold_wgrad_kernel",
    "pdt_bnfold_sums": "bnfold_sums_kernel", "pdt_bnfold_fwd_stats": "bnfold_fwd_stats_kernel",
    "pdt_bnfold_scale_weights": "bnfold_scale_weights_kernel", "pdt_bn_apply_ex": "bn_apply_kernel", "pdt_subsample2": "subsample2_kernel",
    "pdt_bn_apply": "bn_apply_kernel", "pdt_bn_apply2": "bn_apply2_kernel", "pdt_bn_bwd_elemt": "bn_bwd_elemt_kernel",
    "pdt_bn_bwd_reduce_accum": "bn_bwd_reduce_kernel", "pdt_bn_stats_accum": "bn_stats_kernel",
    "pdt_bn_moments": "bn_moments_kernel", "pdt_bn_finalize": "bn_finalize_kernel",
    "pdt_bn_finalize_fused": "bn_finalize_fused_kernel", "pdt_bn_bwd_sums": "bn_bwd_sums_kernel",
    "pdt_bn_bwd_sums_fused": "bn_bwd_sums_fused_kernel", "pdt_bn_relu_maxpool_fwd": "maxpool_fwd_kernel",
    "pdt_bn_relu_maxpool_fwd_winners": "maxpool_fwd_kernel",
    "pdt_pool_bn_bwd_reduce_accum": "pool_bn_bwd_reduce_kernel", "pdt_pool_bn_bwd_elemt": "pool_bn_bwd_elemt_kernel",
    "pdt_input_pad": "input_pad_kernel", "pdt_avgpool_fwd": "avgpool_fwd_kernel", "pdt_avgpool_bwd": "avgpool_bwd_kernel",
    "pdt_softmax_ce": "softmax_ce_kernel", "pdt_colsum_add": "colsum_kernel", "pdt_zero_f32": "zero_f32_kernel",
    "pdt_stem_unpack_add": "stem_unpack_add_kernel", "pdt_car_allreduce": "car_allreduce_kernel",
    "pdt_stream_wait_stream": None,
}


def dump(args) -> None:
    import torch
    from pytorch_distributed_training_amd.config import validate_cfg
    from pytorch_distributed_training_amd.engine.engines import build_engine
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = validate_cfg({
        "dataset": {"name": "synthetic", "n_classes": 1000, "image_size": 224, "input_format": "uint8_nhwc"},
        "training": {"optimizer": {"name": "SGD", "lr": 0.1, "weight_decay": 1.0e-4, "momentum": 0.9},
                     "lr_schedule": {"name": "constant"}, "train_iters": 4, "batch_size": args.batch,
                     "per_gpu_batch_size": args.batch, "precision": "bf16"},
        "model": {"name": args.model},
    })
    torch.manual_seed(1)
    eng = build_engine(cfg, dev)
    img = torch.randint(0, 256, (args.batch, 224, 224, 3), dtype=torch.uint8, device=dev)
    lab = torch.randint(0, 1000, (args.batch,), device=dev)
    for _ in range(3):
        eng.train_step(img, lab)
    torch.cuda.synchronize()
    plans = [p for p in eng.model._plans.values() if p is not None]
    assert plans, "the step was not recorded (step replay off?)"
    program = plans[0][3]
    main = program.stream
    ops = []
    for fn, a in program.items:
        if fn is None:
            ops.append(dict(kind="host", name="host", stream="host", bytes=0, flops=0, desc=""))
            continue
        e = _entry(fn.__name__, a)
        e["name"] = fn.__name__
        e["stream"] = "main" if (a[-1] in (main, None) or fn.__name__ == "pdt_stream_wait_stream") else "side"
        ops.append(e)
    with open(args.oplist, "w") as f:
        json.dump(dict(batch=args.batch, model=args.model, ops=ops), f)
    print(f"{len(ops)} entries -> {args.oplist}")


def join(args) -> None:
    with open(args.oplist) as f:
        rec = json.load(f)
    ops = [o for o in rec["ops"] if KERNEL_OF.get(o["name"], "?") is not None and o["name"] != "host"]
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "input_pad" in r[2]]
    spans = list(zip(starts[-args.steps - 1:-1], starts[-args.steps:]))
    main_q = rows[starts[-1]][3]
    wall = (rows[spans[-1][1]][0] - rows[spans[0][0]][0]) / 1e6 / len(spans)
    wanted = [KERNEL_OF.get(o["name"], o["name"][4:]) for o in ops]
    sums = [[0.0, 0.0] for _ in ops]            # duration, start offset
    used = 0
    for lo, hi in spans:
        step = rows[lo:hi]
        t0 = step[0][0]
        # the recorded step begins AFTER the input transform of its own batch was launched by the trainer:
        # walk both sequences, matching each library call with the next unmatched kernel of its family
        pos = {"main": 0, "side": 0}
        per_q = {"main": [r for r in step if r[3] == main_q], "side": [r for r in step if r[3] != main_q]}
        ok = True
        got = []
        for o, want in zip(ops, wanted):
            q = per_q[o["stream"]]
            i = pos[o["stream"]]
            while i < len(q) and want not in q[i][2]:
                i += 1
            if i >= len(q):
                ok = False
                break
            got.append((q[i][1] - q[i][0], q[i][0] - t0))
            pos[o["stream"]] = i + 1
        if not ok:
            continue
        used += 1
        for s, g in zip(sums, got):
            s[0] += g[0]
            s[1] += g[1]
    if not used:
        sys.exit("could not match the op list with the trace (different batch / model / build?)")
    print(f"{rec['model']} batch {rec['batch']}: {len(ops)} dispatches per step, wall {wall:.3f} ms/step "
          f"(mean of {used} steps)\n")
    table = []
    for o, s in zip(ops, sums):
        us = s[0] / used / 1e3
        table.append(dict(o, us=us, start=s[1] / used / 1e3, tbs=o["bytes"] / (us * 1e-6) / 1e12 if us else 0,
                          tfs=o["flops"] / (us * 1e-6) / 1e12 if us else 0))
    klass = collections.OrderedDict()
    for t in table:
        k = klass.setdefault((t["stream"], t["kind"]), dict(us=0.0, bytes=0, flops=0, n=0))
        k["us"] += t["us"]; k["bytes"] += t["bytes"]; k["flops"] += t["flops"]; k["n"] += 1
    print(f"{'stream':6} {'class':16} {'calls':>5} {'ms/step':>8} {'share':>6} {'min GB':>7} {'TB/s':>6} {'TFLOP':>7} {'TF/s':>6}")
    tot = sum(k["us"] for (s, _), k in klass.items() if s == "main")
    for (s, kind), k in sorted(klass.items(), key=lambda kv: -kv[1]["us"]):
        sec = k["us"] * 1e-6
        print(f"{s:6} {kind:16} {k['n']:5d} {k['us'] / 1e3:8.3f} {100 * k['us'] / tot if s == 'main' else 0:5.1f}% "
              f"{k['bytes'] / 1e9:7.2f} {k['bytes'] / sec / 1e12 if sec else 0:6.2f} {k['flops'] / 1e12:7.3f} "
              f"{k['flops'] / sec / 1e12 if sec else 0:6.0f}")
    gb = sum(t["bytes"] for t in table) / 1e9
    tf = sum(t["flops"] for t in table) / 1e12
    print(f"\nwhole step: {gb:.1f} GB minimum traffic, {tf:.2f} TFLOP -> {gb / wall:.2f} TB/s and {tf / wall * 1e3:.0f} TF/s "
          f"averaged over the wall time; at 5 TB/s the traffic alone is {gb / 5:.1f} ms")
    print(f"\nheaviest dispatches (time inside the step):\n{'#':>4} {'q':4} {'start ms':>8} {'us':>7} {'TB/s':>5} {'TF/s':>5}  kind / shape")
    order = sorted(range(len(table)), key=lambda i: -table[i]["us"])[:args.top]
    for i in order:
        t = table[i]
        print(f"{i:4d} {t['stream']:4} {t['start'] / 1e3:8.2f} {t['us']:7.1f} {t['tbs']:5.2f} {t['tfs']:5.0f}  {t['kind']} {t['desc']}")
    if args.all:
        print("\nall dispatches in launch order:")
        for i, t in enumerate(table):
            print(f"{i:4d} {t['stream']:4} {t['start'] / 1e3:8.2f} {t['us']:7.1f} {t['tbs']:5.2f} {t['tfs']:5.0f}  {t['kind']} {t['desc']}")


def main() -> None:
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    d = sub.add_parser("dump")
    d.add_argument("oplist")
    d.add_argument("--batch", type=int, default=1024)
    d.add_argument("--model", default="ResNet50")
    j = sub.add_parser("join")
    j.add_argument("oplist")
    j.add_argument("trace")
    j.add_argument("--steps", type=int, default=4)
    j.add_argument("--top", type=int, default=40)
    j.add_argument("--all", action="store_true")
    args = ap.parse_args()
    dump(args) if args.cmd == "dump" else join(args)


if __name__ == "__main__":
    main()
