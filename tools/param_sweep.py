"""Schedule parameters of the relaxation on the C3 graph, one process, one upload: for every (delta multiplier, near_low,
first-band multiplier) the best-of-N host clock of shortest_path(T), the launches it took and whether the keys are those
of the default schedule.   python tools/param_sweep.py [states] [reps]"""
import itertools, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
fanout = int(sys.argv[3]) if len(sys.argv) > 3 else 10
fine = len(sys.argv) > 4 and sys.argv[4] == "fine"
t = synth.make_transducer(states, fanout, 256, 0.0, seed=3)
mean_w = float(np.asarray(t["arcs"]["weight"], dtype=np.float64).mean())
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)


def run(env):
    for k in ("WFST_SSSP_DELTA", "WFST_SSSP_NEAR_LOW", "WFST_SSSP_TAU0_MULT", "WFST_SSSP_NARROW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    best, w = 1e9, None
    for i in range(reps + 6):
        s0 = ctx.stats()["sweeps"]
        t0 = time.perf_counter(); p = d.shortest_path(); dt = time.perf_counter() - t0
        if i >= 6:
            best = min(best, dt)
    sw = ctx.stats()["sweeps"] - s0
    return best, sw, d.shortest_distance().view(np.uint32).copy()


base, sw, w0 = run({})
print(f"states {states} fan-out {fanout}; mean arc weight {mean_w:.4f}; default: {base*1e3:.3f} ms, {sw} launches")
rows = []
dms = [1.3, 1.4, 1.5, 1.6, 1.75] if fine else [1.0, 1.25, 1.5, 2.0, 2.5, 3.0, 4.0]
nls = [4096, 16384, 65536, 262144, 1 << 20] if fine else [2048, 4096, 8192, 16384, 32768, 65536]
for dm, nl in itertools.product(dms, nls):
    b, s, w = run({"WFST_SSSP_DELTA": repr(dm * mean_w), "WFST_SSSP_NEAR_LOW": str(nl)})
    rows.append((b, dm, nl, 1.0, 8192, s, w))
rows.sort()
# around the best (delta, near_low): first band and hand-over threshold
_, dm, nl, _, _, _, _ = rows[0]
for t0m, nt in itertools.product([0.9, 1.0, 1.1] if fine else [0.5, 0.75, 1.0, 1.5, 2.0], [8192, 16384, 32768] if fine else [4096, 8192, 16384]):
    if t0m == 1.0 and nt == 8192:
        continue
    b, s, w = run({"WFST_SSSP_DELTA": repr(dm * mean_w), "WFST_SSSP_NEAR_LOW": str(nl), "WFST_SSSP_TAU0_MULT": str(t0m), "WFST_SSSP_NARROW": str(nt)})
    rows.append((b, dm, nl, t0m, nt, s, w))
rows.sort()
print("   ms    delta_mult near_low tau0_mult narrow launches same_distances")
for b, dm, nl, t0m, nt, s, w in rows:
    print(f"{b*1e3:7.3f}  {dm:8.2f} {nl:8d} {t0m:9.2f} {nt:6d} {s:8d}  {bool(np.array_equal(w, w0))}")
b2, s2, _ = run({})
print(f"default again: {b2*1e3:.3f} ms, {s2} launches")
