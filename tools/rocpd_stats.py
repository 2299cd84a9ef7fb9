#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table
(the --stats view): calls, total/avg/min/max duration, % of GPU time.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size), max(d.workgroup_size_x) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPR,AGPR,LDS,WG"]
    for n, c, t, mn, mx, v, a, l, wg in rows:
        n = re.sub(r"\s+", " ", n).replace(",", ";")
        lines.append(f"\"{n}\",{c},{t},{t / c:.0f},{100.0 * t / tot:.2f},{mn},{mx},{v},{a},{l},{wg}")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    for ln in lines[:40]:
        print(ln[:230])
    print("total GPU kernel time (ms):", tot / 1e6)


if __name__ == "__main__":
    main()
