"""Per-launch trace of the relaxation of shortest_path(T) (profiling mode): arcs, frontier states, ms, GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
ctx = rustfst_amd.default_context()
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for _ in range(2):
    d.shortest_path()
ctx.reset_stats(); ctx.set_profiling(True); d.shortest_path(); ctx.set_profiling(False)
ms, arcs, st = ctx.sweep_trace()
print("sweep  states     arcs      us    Garcs/s  algGB/s")
for k in range(len(ms)):
    b = 20.0 * arcs[k] + 12.0 * st[k]
    print(f"{k:4d} {st[k]:8d} {arcs[k]:9d} {ms[k]*1e3:8.2f} {arcs[k]/max(ms[k],1e-9)/1e6:8.2f} {b/max(ms[k],1e-9)/1e6:8.1f}")
print("total", st.sum(), arcs.sum(), ms.sum() * 1e3, "us")
