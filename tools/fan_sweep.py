"""Relaxation chain (HIP events) of shortest_path on T(states, fan-out): python tools/fan_sweep.py <fan-out> [states]   (WFST_SSSP_DELTA=x sweeps the band)"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rustfst_amd
from rustfst_amd import synth
fan = int(sys.argv[1]); states = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
t = synth.make_transducer(states, fan, 256, 0.0, seed=3)
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for _ in range(5): d.shortest_path()
ctx.set_profiling(2); v=[]
for _ in range(9):
    d.shortest_path(); st=ctx.stats(); v.append(1e3*st["relax_ms"])
ctx.set_profiling(0)
B = 20*len(t["arcs"]) + 12*states
m = statistics.median(v)
print(f"fan-out {fan}, {states} states: chain {m:.1f} us, launches {st['relax_launches']}, kernel {st['relax_kernel']}, frac {B / (m * 1e-6) / 8e12:.4f}, env LPS={os.environ.get('WFST_SSSP_LPS')} UMAX={os.environ.get('WFST_SSSP_UMAX')} STG={os.environ.get('WFST_SSSP_STG')}")
