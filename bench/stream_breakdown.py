"""Critical-path composition of a step from a rocprofv3 kernel trace with two streams.

    python bench/stream_breakdown.py <kernel_trace.csv> [--steps N]

Kernels are grouped by the queue they ran on; a step starts with the input-transform kernel and the
queue that kernel runs on is the main (critical-path) stream.  For the last N steps it prints, per kernel class, the summed duration on
the main queue, and the summed duration of everything that ran on other queues (the overlapped
weight-gradient stream), next to the wall time per step.
"""
import argparse
import collections
import csv
import re


def klass(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    for key in ("igemm_kernel", "wgrad_kernel", "bn_apply2", "bn_apply", "bn_bwd_elemt", "bn_bwd_reduce", "bn_finalize_fused",
                "bn_bwd_sums", "maxpool", "avgpool", "sgd_kernel", "transpose_w", "softmax_ce", "input_pad", "stem_",
                "colsum", "zero_f"):
        if key in name:
            return key
    return name.split("(")[0][:30]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dispatches", action="store_true",
                    help="also list every dispatch of a step in launch order (mean over the steps): which layer's "
                         "kernel costs what inside the running step, next to what overlaps it")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"],
                         int(r.get("Grid_Size_X", 0) or 0) // max(int(r.get("Workgroup_Size_X", 1) or 1), 1)))
    rows.sort()
    sgd = [i - 1 for i, r in enumerate(rows) if "input_pad" in r[2]]      # index of the last dispatch of the step before
    lo, hi = sgd[-args.steps - 1], sgd[-1]
    main_q = rows[hi + 1][3]
    win = rows[lo + 1:hi + 1]
    wall = (rows[hi + 1][0] - rows[lo + 1][0]) / 1e6 / args.steps
    on_main, off_main = collections.Counter(), collections.Counter()
    for s, e, name, q, _ in win:
        (on_main if q == main_q else off_main)[klass(name)] += (e - s) / 1e6 / args.steps
    print(f"wall {wall:.3f} ms/step; main queue {sum(on_main.values()):.3f} ms/step, other queues "
          f"{sum(off_main.values()):.3f} ms/step")
    for k, v in on_main.most_common():
        print(f"  main  {k:24s} {v:7.3f}")
    for k, v in off_main.most_common():
        print(f"  other {k:24s} {v:7.3f}")
    if args.dispatches:
        per_step = [rows[a + 1:b + 1] for a, b in zip(sgd[-args.steps - 1:-1], sgd[-args.steps:])]
        n = len(per_step[0])
        if any(len(p) != n for p in per_step):
            print("(steps differ in their number of dispatches: listing the last step only)")
            per_step = per_step[-1:]
        t0s = [p[0][0] for p in per_step]
        print(f"\n{'#':>4} {'q':>2} {'start us':>9} {'dur us':>8} {'blocks':>7}  kernel")
        for i in range(len(per_step[0])):
            dur = sum(p[i][1] - p[i][0] for p in per_step) / len(per_step) / 1e3
            start = sum(p[i][0] - t for p, t in zip(per_step, t0s)) / len(per_step) / 1e3
            _, _, name, q, blocks = per_step[-1][i]
            short = re.sub(r"^void ", "", name).split("(")[0][:60]
            print(f"{i:4d} {'M' if q == main_q else 'w':>2} {start:9.1f} {dur:8.1f} {blocks:7d}  {short}")


if __name__ == "__main__":
    main()
