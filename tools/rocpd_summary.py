#!/usr/bin/env python
"""Summarise rocprofv3 rocpd sqlite databases.
  kernel trace :  python tools/rocpd_summary.py trace  <trace_results.db>
  PMC passes   :  python tools/rocpd_summary.py pmc    <pmc_fetch_results.db> <pmc_write_results.db>
Prints markdown tables (what `rocprofv3 --stats` reports per kernel: calls, total/avg/min/max, share)."""
import sqlite3
import sys


def short(name, n=96):
    name = name.replace("wfst::(anonymous namespace)::", "")
    return name if len(name) < n else name[:n - 3] + "..."


def trace(path, alg_bytes=212e6):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
        a[4] += 1 if d >= 3000 else 0
    tot = sum(a[1] for a in agg.values()) or 1
    print("| kernel | calls | total_us | avg_us | min_us | max_us | % of GPU time |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(name)}` | {a[0]} | {a[1] / 1e3:.1f} | {a[1] / 1e3 / a[0]:.2f} | {a[2] / 1e3:.2f} | "
              f"{a[3] / 1e3:.2f} | {100 * a[1] / tot:.1f} |")
    # the relaxation launches of a solve: time per kernel, and ONE roofline fraction over their SUM (20 E + 12 N bytes belong to
    # the whole solve: dividing them by one kernel's share of the time means nothing)
    n_solves = sum(1 for name, _, _ in rows if "setup_kernel" in name) or 1
    chain = 0.0
    for kname in ("sssp_relax_kernel", "sssp_mbox_kernel", "sssp_mbox_resident_kernel", "sssp_bin_expand_kernel", "sssp_bin_apply_kernel"):
        rel = [(e - s) for name, s, e in rows if kname + "(" in name or kname + "<" in name]
        if not rel:
            continue
        work = [d for d in rel if d >= 5500]  # launches after convergence inside a batch only find an empty frontier
        per_solve = sum(rel) / n_solves / 1e3
        chain += per_solve
        print()
        print(f"{kname}: {len(rel)} launches in {n_solves} solves ({len(rel) / n_solves:.1f} per solve); "
              f"avg over all launches {sum(rel) / len(rel) / 1e3:.2f} us; "
              f"{len(work)} launches >= 5.5 us avg {sum(work) / max(1, len(work)) / 1e3:.2f} us; "
              f"kernel time per solve {per_solve:.1f} us")
    if chain > 0:
        print()
        print(f"relaxation kernels per solve (sum of the above): {chain:.1f} us -> {alg_bytes / 1e6:.0f} MB (20 E + 12 N) / that = "
              f"{alg_bytes / chain / 1e3:.1f} GB/s = {alg_bytes / chain / 1e3 / 8000:.4f} of the 8 TB/s peak")


def pmc(fetch_db, write_db):
    def load(path):
        c = sqlite3.connect(path)
        out = {}
        for name, cname, n, tot in c.execute(
                "select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
            out[name] = (cname, n, tot)
        nsolves = c.execute("select count(*) from pmc_events where name like '%setup_kernel%'").fetchone()[0]
        return out, max(1, nsolves)
    f, nf = load(fetch_db)
    w, nw = load(write_db)
    print("| kernel | launches | FETCH_SIZE raw (MB) | 2x FETCH_SIZE (MB) | WRITE_SIZE (MB) | per solve: 2xFETCH+WRITE (MB) |")
    print("|---|---:|---:|---:|---:|---:|")
    for name in sorted(f, key=lambda k: -f[k][2]):
        fr = f[name][2] / 1024.0
        wr = w.get(name, ("", 0, 0.0))[2] / 1024.0
        if fr + wr < 1.0:
            continue
        print(f"| `{short(name)}` | {f[name][1]} | {fr:.1f} | {2 * fr:.1f} | {wr:.1f} | {(2 * fr / nf + wr / nw):.1f} |")
    print()
    print(f"solves in the fetch pass: {nf}, in the write pass: {nw}. FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE is doubled "
          "as MI355X_MICROARCH.md §HBM prescribes for gfx950 (calibrated there on wide coalesced reads; "
          "the 8-B gathers of this kernel are uncalibrated, so read the column as an upper bound and the raw column as a lower bound).")


def timeline(path, which=-1):
    """Kernels of one solve (default: the last) in launch order: duration and the idle gap before each."""
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    setups = [i for i, r in enumerate(rows) if "setup_kernel" in r[0]]
    if not setups:
        return
    i0 = setups[which]
    # the solve ends with its tail: sssp_tail_kernel (repeated queries) or the header kernel of the three-launch tail
    i1 = next((i for i in range(i0, len(rows)) if "sssp_header_kernel" in rows[i][0] or "sssp_tail_kernel" in rows[i][0]), len(rows) - 1)
    print("| # | kernel | us | gap before (us) |")
    print("|---:|---|---:|---:|")
    tot = gaps = 0
    for k in range(i0, i1 + 1):
        name, s, e = rows[k]
        gap = (s - rows[k - 1][2]) / 1e3 if k > i0 else 0.0
        tot += e - s
        gaps += max(0.0, gap)
        print(f"| {k - i0} | `{short(name, 40)}` | {(e - s) / 1e3:.2f} | {gap:.2f} |")
    print(f"\nsolve: first start -> last end {(rows[i1][2] - rows[i0][1]) / 1e3:.1f} us; kernel time {tot / 1e3:.1f} us; gaps {gaps:.1f} us")


def window(path, which=-2):
    """every kernel (all streams) between two consecutive setup kernels: start offset, duration"""
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    setups = [i for i, r in enumerate(rows) if "setup_kernel" in r[0]]
    i0 = setups[which]
    t0 = min(rows[i0][1], min(r[1] for r in rows[max(0, i0 - 3):i0 + 1]))
    i1 = setups[which + 1] if which + 1 < 0 and which + 1 + len(setups) < len(setups) else len(rows)
    print("| start_us | dur_us | kernel |")
    print("|---:|---:|---|")
    for k in range(max(0, i0 - 3), min(i1, len(rows))):
        name, s, e = rows[k]
        print(f"| {(s - t0) / 1e3:8.1f} | {(e - s) / 1e3:7.2f} | `{short(name, 36)}` |")


if __name__ == "__main__":
    if sys.argv[1] == "window":
        window(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -2)
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -1)
    elif sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
