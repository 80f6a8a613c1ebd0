"""Idle-gap analysis of a rocprofv3 kernel trace (CSV).

    python bench/trace_gaps.py <kernel_trace.csv> [--steps N]

Sorts kernels by start time, takes the last N occurrences of the SGD kernel as step boundaries,
and reports for that window: wall time, summed kernel time, idle time between consecutive
kernels, and the kernels that most often precede a long gap (where the host or a cross-stream
dependency failed to keep the GPU fed).
"""
import argparse
import collections
import csv
import re


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:60]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--gap-us", type=float, default=3.0, help="report gaps longer than this")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r[2]]
    if len(sgd) < args.steps + 1:
        raise SystemExit("not enough steps in the trace")
    lo, hi = sgd[-args.steps - 1], sgd[-1]
    win = rows[lo:hi + 1]
    wall = (win[-1][1] - win[0][1]) / 1e3
    busy = 0.0
    gaps = collections.Counter()
    gap_time = collections.Counter()
    idle = 0.0
    cur_end = win[0][1]
    for s, e, name in win[1:]:
        if s > cur_end:
            g = (s - cur_end) / 1e3
            idle += g
            if g > args.gap_us:
                gaps[short(prev)] += 1
                gap_time[short(prev)] += g
        busy += (e - max(s, cur_end)) / 1e3 if e > cur_end else 0.0
        if e > cur_end:
            cur_end, prev = e, name
    n = args.steps
    print(f"steps {n}: wall {wall / n / 1e3:.3f} ms/step, busy {busy / n / 1e3:.3f} ms/step, "
          f"idle {idle / n / 1e3:.3f} ms/step, kernels {len(win) // n}/step")
    print(f"gaps > {args.gap_us} us, by preceding kernel (count/step, us/step):")
    for k, t in gap_time.most_common(15):
        print(f"  {gaps[k] / n:7.1f} {t / n:9.1f}  {k}")


if __name__ == "__main__":
    main()
