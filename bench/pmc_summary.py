"""Per-kernel summary of rocprofv3 counter passes (the directory bench/rocprof_pmc.sh writes).

    python bench/pmc_summary.py <OUT_DIR>

Sums every counter over the dispatches of each kernel name and prints the ratios that matter for
the conv kernels: share of wave cycles spent waiting / issuing, MFMA-busy share of the busy cycles,
LDS bank-conflict share of LDS cycles, HBM bytes fetched (FETCH_SIZE is reported in KB and counts
a 128-byte request as 64 bytes for wide streaming reads on gfx950, so it is doubled here) and
written, and the L2 hit rate.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:44]


def main() -> None:
    root = sys.argv[1]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (path, r["Dispatch_Id"])
                if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE"):
                    seen.add(key)
                    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                        calls[k] += 1
    print("%-44s %6s %7s %7s %7s %8s %8s %9s %9s %6s" % (
        "kernel", "calls", "wait%", "issue%", "mfma%", "ldsconf%", "", "fetch MB", "write MB", "L2hit%"))
    for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        busy = c.get("SQ_BUSY_CYCLES", 0.0)
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        hit, miss = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        print("%-44s %6d %7.1f %7.1f %7.1f %8.2f %8s %9.1f %9.1f %6.1f" % (
            k, calls[k], 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(busy, 1.0),
            100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(lds, 1.0), "",
            2 * c.get("FETCH_SIZE", 0) / 1024.0, c.get("WRITE_SIZE", 0) / 1024.0,
            100 * hit / max(hit + miss, 1.0)))


if __name__ == "__main__":
    main()
