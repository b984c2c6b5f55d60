"""Reads the phase stamps a mailbox solve leaves under WFST_SSSP_MBOX_TRACE (sssp_mailbox.h MB_STAMP) and prints, per
sweep, when the slowest / median block passed each phase (us since the first block of the sweep started)."""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
ns, nb = np.frombuffer(raw[:8], dtype=np.uint32)
a = np.frombuffer(raw[8:], dtype=np.uint64).reshape(ns, nb, 16).astype(np.int64)
TICK = 0.01  # wall_clock64: 100 MHz
names = ["start", "trip1", "applied", "scanned", "staged", "expanded", "end"]
order = [0, 1, 3, 4, 12, 5, 6]
print("sweep  busy  idle |  msgs_in  active  sent | per phase: max over busy blocks (median) in us since sweep start | idle-exit max")
for k in range(ns):
    st = a[k, :, 0]
    if not st.any():
        break
    t0 = st[st > 0].min()
    busy = a[k, :, 6] > 0
    idle = a[k, :, 15] > 0
    line = f"{k:4d} {busy.sum():5d} {idle.sum():5d} | {0:8d} {a[k, busy, 7].sum():7d} {a[k, busy, 8].sum():7d} |"
    for p, slot in enumerate(order):
        v = a[k, busy, slot]
        if v.size:
            line += f" {names[p]} {(v.max() - t0) * TICK:5.2f} ({(np.median(v) - t0) * TICK:5.2f})"
    if idle.any():
        line += f" | {(a[k, idle, 15].max() - t0) * TICK:5.2f}"
    print(line)
