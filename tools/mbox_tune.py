"""Schedule-parameter sweep for shortest_path(T) on the C3 graph: near-far band width, first band, widening threshold.
Prints best-of-N wall time, sweeps and arcs relaxed for every combination (env knobs of sssp.hip)."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "1"
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
os.environ["WFST_SSSP_MAILBOX"] = mode
deltas = [float(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["5", "7.5", "10"])]
mults = [float(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["1", "1.5"])]
lows = [int(x) for x in (sys.argv[5].split(",") if len(sys.argv) > 5 else ["4096", "32768"])]
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
d.shortest_path(); d.shortest_path()
for delta, mult, low in itertools.product(deltas, mults, lows):
    os.environ["WFST_SSSP_DELTA"] = str(delta)
    os.environ["WFST_SSSP_TAU0_MULT"] = str(mult)
    os.environ["WFST_SSSP_NEAR_LOW"] = str(low)
    best = 1e9
    for _ in range(8):
        t0 = time.perf_counter(); d.shortest_path(); best = min(best, time.perf_counter() - t0)
    sweeps = ctx.stats()["sweeps"]
    ctx.reset_stats(); ctx.set_profiling(True); d.shortest_path(); ctx.set_profiling(False)
    ms, arcs, st = ctx.sweep_trace()
    print(f"delta {delta:5.2f} tau0x {mult:4.2f} near_low {low:6d}: best {best*1e3:.3f} ms  sweeps {sweeps:3d}  arcs {arcs.sum()/1e6:6.2f} M  states {st.sum()/1e6:5.2f} M  profiled kernel sum {ms.sum()*1e3:6.1f} us", flush=True)
