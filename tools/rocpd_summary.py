#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into the per-kernel table that
`rocprofv3 --stats` prints: name, calls, total/avg/min/max duration (us), share of GPU time.
usage: python tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/<name>.md"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    print("| kernel | calls | total_us | avg_us | min_us | max_us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {a[0]} | {a[1] / 1e3:.1f} | {a[1] / 1e3 / a[0]:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
