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
modes = ctx.sweep_modes()
kernel = int(ctx.stats()["relax_kernel"])
names = {0: "atomic", 7: "binned"} if kernel in (0, 3) else {0: "wide", 1: "collect", 2: "narrow"}
print(f"relax_kernel {kernel} (0 atomic sweeps, 1 mailbox, 2 resident mailbox, 3 atomic sweeps + binned levels)")
print("sweep  states     arcs      us    Garcs/s  algGB/s  ran as")
for k in range(len(ms)):
    b = 20.0 * arcs[k] + 12.0 * st[k]
    print(f"{k:4d} {st[k]:8d} {arcs[k]:9d} {ms[k]*1e3:8.2f} {arcs[k]/max(ms[k],1e-9)/1e6:8.2f} {b/max(ms[k],1e-9)/1e6:8.1f}  {names.get(int(modes[k]), modes[k])}")
print("total", st.sum(), arcs.sum(), ms.sum() * 1e3, "us")
for m in sorted(set(modes.tolist())):
    sel = modes == m
    print(f"  {names.get(int(m), m)}: {int(sel.sum())} launches, {arcs[sel].sum()} arcs, {ms[sel].sum()*1e3:.1f} us")
