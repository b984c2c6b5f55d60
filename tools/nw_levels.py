"""Per-level stamps of the NARROW launches (a library built with -DWFST_NW_TRACE, WFST_SSSP_MBOX_TRACE=<file>): for each sweep
whose blocks followed work lists, the level starts of the busiest block (us since its first level) and the entries per level."""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
ns, nb = np.frombuffer(raw[:8], dtype=np.uint32)
a = np.frombuffer(raw[8:], dtype=np.uint64).reshape(ns, nb, 16)
for k in range(ns):
    lv = a[k, :, 2:15]
    if not lv.any():
        continue
    t = (lv >> np.uint64(12)).astype(np.int64)
    cnt = (lv & np.uint64(4095)).astype(np.int64)
    depth = (t > 0).sum(axis=1)
    b = int(np.argmax(depth))
    busy = int((depth > 0).sum())
    t0 = t[b, 0]
    print(f"sweep {k}: {busy} blocks with work; deepest block {b}: " + " ".join(f"L{i}@{(t[b, i] - t0) * 0.01:.2f}us(n={cnt[b, i]})" for i in range(depth[b])))
    tot = [(int(cnt[:, i].sum()), int((t[:, i] > 0).sum())) for i in range(13)]
    print("   entries per level over all blocks (blocks): " + " ".join(f"{c}({n})" for c, n in tot if n))
