"""Schedule parameters of the relaxation under the RESIDENT kernel (round 6): band width x near_low x hand-over threshold,
the relaxation chain by HIP events (profiling mode 2: what roofline.frac is computed from), one process, one upload; every
setting's keys (distance, hops) must be the default's.   python tools/res_sweep.py [states] [reps] [quick]"""
import itertools, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
quick = len(sys.argv) > 3 and sys.argv[3] == "quick"
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
mean_w = float(np.asarray(t["arcs"]["weight"], dtype=np.float64).mean())
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
KNOBS = ("WFST_SSSP_DELTA", "WFST_SSSP_NEAR_LOW", "WFST_SSSP_TAU0_MULT", "WFST_SSSP_NARROW", "WFST_SSSP_LOG13")
ref = None


def run(env):
    global ref
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx.set_profiling(0)
    dist, hops = d.shortest_distance(want_hops=True)
    key = (dist.view(np.uint32).astype(np.uint64) << 32) | hops
    if ref is None:
        ref = key
    same = bool(np.array_equal(key, ref))
    for _ in range(3):
        d.shortest_path()
    ctx.set_profiling(2)
    chain, launches = [], 0
    for _ in range(reps):
        d.shortest_path()
        st = ctx.stats()
        if st["relax_launches"]:
            chain.append(st["relax_ms"] * 1e3)
            launches = st["relax_launches"]
    ctx.set_profiling(0)
    st = ctx.stats()
    return (statistics.median(chain) if chain else float("nan")), (min(chain) if chain else float("nan")), launches, st["relax_kernel"], st["resident_aborts"], same


base = run({})
print(f"states {states}; mean arc weight {mean_w:.4f}; default: chain median {base[0]:.1f} us (min {base[1]:.1f}) in {base[2]} launches, kernel {base[3]}", flush=True)
rows = []
dms = [1.25, 1.5, 1.75] if quick else [1.0, 1.25, 1.4, 1.5, 1.6, 1.75, 2.0, 2.5]
nls = [16384, 65536] if quick else [4096, 8192, 16384, 32768, 65536, 131072]
for dm, nl in itertools.product(dms, nls):
    r = run({"WFST_SSSP_DELTA": repr(dm * mean_w), "WFST_SSSP_NEAR_LOW": str(nl)})
    rows.append((r[0], dm, nl, 1.0, "dflt", r))
    print(f"  delta x{dm:4.2f} near_low {nl:7d}: {r[0]:7.1f} us (min {r[1]:.1f}) {r[2]} launches same {r[5]}", flush=True)
rows.sort(key=lambda x: x[0])
_, dm, nl, _, _, _ = rows[0]
for t0m, nt in itertools.product([0.75, 1.0, 1.5], [4096, 8192, 16384, 32768, 65536]):
    r = run({"WFST_SSSP_DELTA": repr(dm * mean_w), "WFST_SSSP_NEAR_LOW": str(nl), "WFST_SSSP_TAU0_MULT": str(t0m), "WFST_SSSP_NARROW": str(nt)})
    rows.append((r[0], dm, nl, t0m, nt, r))
    print(f"  best + tau0 x{t0m:4.2f} narrow {nt:6d}: {r[0]:7.1f} us (min {r[1]:.1f}) {r[2]} launches same {r[5]}", flush=True)
r = run({"WFST_SSSP_LOG13": "1"})
print(f"  8192-state blocks (half the workgroups), default schedule: {r[0]:.1f} us (min {r[1]:.1f}) {r[2]} launches kernel {r[3]} same {r[5]}", flush=True)
rows.sort(key=lambda x: x[0])
print("chain_us delta_mult near_low tau0_mult narrow launches aborts same_keys")
for c, dm, nl, t0m, nt, r in rows[:12]:
    print(f"{c:8.1f} {dm:10.2f} {nl:8d} {t0m:9.2f} {str(nt):>6s} {r[2]:8d} {r[4]:6d} {r[5]}")
b2 = run({})
print(f"default again: {b2[0]:.1f} us (min {b2[1]:.1f})")
