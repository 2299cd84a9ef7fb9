#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_bench.sh).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is taken as is (uncalibrated).
    python tools/pmc_traffic.py gpurun_out/pmc_bench profiles/r01_traffic.json"""
import collections
import csv
import json
import sys


def avg_by_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    root, out = sys.argv[1], sys.argv[2]
    f = avg_by_kernel(f"{root}/FETCH_SIZE/p_counter_collection.csv", "FETCH_SIZE")
    w = avg_by_kernel(f"{root}/WRITE_SIZE/p_counter_collection.csv", "WRITE_SIZE")
    res = {}
    for k in f:
        fetch = f[k][0] * 1024 * 2.0   # KiB -> bytes, x2 gfx950 correction for wide streaming reads
        write = w.get(k, (0.0, 0))[0] * 1024
        res[k] = dict(launches=f[k][1], fetch_bytes=fetch, write_bytes=write, hbm_bytes=fetch + write)
    json.dump({"note": "avg per launch; FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE uncalibrated", "kernels": res},
              open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:10]:
        print(f"{k[:90]:90s} n={v['launches']:4d} fetch={v['fetch_bytes'] / 1e6:9.1f} MB write={v['write_bytes'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
