"""Join the CSV files of bench/tile_table.sh into one per-shape table.

    python bench/tile_table.py DIR 256 512 1024

Per (layer, op, batch): time of the automatic choice, of every forced variant, which variant wins and how far the
automatic choice is from the winner.  The bottom lines give the count-weighted totals of one ResNet-50 pass.
"""
from __future__ import annotations

import csv
import os
import sys


def load(path):
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {(r["layer"], r["op"]): (float(r["hip_us"]), int(r["count"]), float(r["tflops"]), float(r["ideal_tbps"]))
                for r in csv.DictReader(f)}


def main() -> int:
    d = sys.argv[1]
    batches = [int(b) for b in sys.argv[2:]] or [256, 512, 1024]
    for b in batches:
        conv = {"auto": load(f"{d}/b{b}_dma_t0.csv"), "dma128": load(f"{d}/b{b}_dma_t3.csv"),
                "dma64": load(f"{d}/b{b}_dma_t4.csv"), "igemm": load(f"{d}/b{b}_igemm.csv")}
        wg = {"auto": load(f"{d}/b{b}_wgrad_t0.csv"), "w128x128": load(f"{d}/b{b}_wgrad_t1.csv"),
              "w64x64": load(f"{d}/b{b}_wgrad_t2.csv"), "w128x64": load(f"{d}/b{b}_wgrad_t4.csv")}
        for title, table in (("forward / data gradient", conv), ("weight gradient", wg)):
            names = list(table)
            print(f"\n== batch {b}: {title} (us per launch) ==")
            print(f"{'layer':5s} {'op':6s} " + " ".join(f"{n:>9s}" for n in names) + f" {'best':>9s} {'auto/best':>9s} {'TF/s':>7s} {'TB/s':>6s}")
            tot = {n: 0.0 for n in names}
            tot_best = 0.0
            for key in table["auto"]:
                ts = {n: table[n][key][0] for n in names if key in table[n]}
                best = min(ts, key=ts.get)
                auto, count, tf, tbs = table["auto"][key]
                print(f"{key[0]:5s} {key[1]:6s} " + " ".join(f"{ts.get(n, float('nan')):9.1f}" for n in names) +
                      f" {best:>9s} {auto / ts[best]:9.3f} {tf:7.1f} {tbs:6.2f}")
                for n in names:
                    tot[n] += ts.get(n, auto) * count
                tot_best += ts[best] * count
            print("one ResNet-50 pass, ms: " + "  ".join(f"{n} {tot[n] / 1e3:.2f}" for n in names) +
                  f"  per-shape best {tot_best / 1e3:.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
