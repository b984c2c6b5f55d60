"""Reads a WFST_SSSP_RES_TRACE file (u32 levels, u32 nb, u64 [levels][nb][4] wall-clock stamps at 100 MHz: inbox applied,
scanned, expanded, published) and prints, per level, when the LAST / median block passed each stamp (us since the first
block's `applied` stamp of level 0)."""
import sys, struct
import numpy as np
raw = open(sys.argv[1], "rb").read()
levels, nb = struct.unpack("II", raw[:8])
a = np.frombuffer(raw[8:], dtype=np.uint64).reshape(levels, nb, 4).astype(np.float64)
t0 = a[a > 0].min()  # (rows 0-1 may belong to a later, idle launch of the same solve)
prev_end = None
print("level | applied max (med) | scanned | expanded | published | level span (prev published max -> this published max)")
for l in range(levels):
    if not (a[l, :, 0] > 0).any():
        break
    row = []
    for p in range(4):
        v = a[l, :, p]
        v = v[v > 0]
        row.append((float("nan"), float("nan")) if v.size == 0 else ((v.max() - t0) / 100.0, (np.median(v) - t0) / 100.0))
    end = row[3][0] if row[3][0] == row[3][0] else row[1][0]
    span = end - (prev_end if prev_end is not None else 0.0)
    prev_end = end
    slow = int(np.argmax(a[l, :, 2])) if (a[l, :, 2] > 0).any() else -1
    print(f"{l:5d} | " + " | ".join(f"{m:8.2f} ({md:8.2f})" for m, md in row) + f" | {span:7.2f} | slowest expander: block {slow}")
