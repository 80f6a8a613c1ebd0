"""Critical-path composition of a step from a rocprofv3 kernel trace with two streams.

    python bench/stream_breakdown.py <kernel_trace.csv> [--steps N]

Kernels are grouped by the queue they ran on; the queue that holds the SGD kernel is the main
(critical-path) stream.  For the last N steps it prints, per kernel class, the summed duration on
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
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r[2]]
    lo, hi = sgd[-args.steps - 1], sgd[-1]
    main_q = rows[hi][3]
    win = rows[lo + 1:hi + 1]
    wall = (rows[hi][1] - rows[lo][1]) / 1e6 / args.steps
    on_main, off_main = collections.Counter(), collections.Counter()
    for s, e, name, q in win:
        (on_main if q == main_q else off_main)[klass(name)] += (e - s) / 1e6 / args.steps
    print(f"wall {wall:.3f} ms/step; main queue {sum(on_main.values()):.3f} ms/step, other queues "
          f"{sum(off_main.values()):.3f} ms/step")
    for k, v in on_main.most_common():
        print(f"  main  {k:24s} {v:7.3f}")
    for k, v in off_main.most_common():
        print(f"  other {k:24s} {v:7.3f}")


if __name__ == "__main__":
    main()
