"""Per-level kernel timeline of the LAST solve in a rocprofv3 kernel trace (rocpd sqlite): one line per launch slot of the
atomic sweeps / binned levels (sssp_relax_kernel | sssp_bin_expand_kernel | sssp_bin_apply_kernel), or per launch otherwise.
usage: level_timeline.py <results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, (n, s, e) in enumerate(rows) if "setup_kernel" in n]
seq = rows[idx[-1]:] if idx else rows
t0 = seq[0][1]
lvl, line, tot = -1, [], {}
def flush():
    if line:
        print(f"L{lvl:2d} " + " | ".join(line))
for n, s, e in seq:
    short = ("relax" if "sssp_relax" in n else "expand" if "bin_expand" in n else "apply" if "bin_apply" in n
             else "resident" if "resident" in n else "mbox" if "sssp_mbox_kernel" in n else None)
    if short is None:
        continue
    if short in ("relax", "resident", "mbox"):
        flush(); line = []; lvl += 1
    d = (e - s) / 1e3
    tot[short] = tot.get(short, 0.0) + d
    line.append(f"{short:6s} @{(s - t0) / 1e3:8.1f} {d:7.1f} us")
flush()
print("per solve:", ", ".join(f"{k} {v:.1f} us" for k, v in tot.items()), f"| sum {sum(tot.values()):.1f} us")
