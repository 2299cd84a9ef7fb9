#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a tools/pmc_bench_sq.sh output directory.
    python tools/pmc_table.py gpurun_out/pmc_sq [kernel-substring]"""
import collections
import csv
import glob
import sys


def main():
    root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{root}/*/p_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        n = max(len(v) for v in cs.values())
        print(f"== {k[:110]}  ({n} launches)")
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} avg {sum(v) / len(v):16.1f}   max {max(v):16.1f}")


if __name__ == "__main__":
    main()
