"""shortest_path(T) under different chase settings (WFST_SSSP_CHASE_CAP / _ROUNDS): sweeps, arcs relaxed, ms per solve.
Every setting must return the same FST as chasing switched off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
synth.make_acceptors(t, 1, 8, seed0=1000)  # marks finals like the bench does
ctx = rustfst_amd.default_context()
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
settings = [(0, 0, 0), (32, 8, 4096), (32, 16, 4096), (32, 32, 4096), (32, 64, 4096), (64, 128, 4096), (16, 16, 4096), (16, 32, 4096),
            (32, 16, 1024), (32, 32, 1024), (32, 16, 16384), (32, 32, 16384), (8, 16, 4096), (8, 32, 4096)]
if len(sys.argv) > 2:
    settings = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]]
ref = None
print(" cap rounds    low sweeps   arcs_relaxed  relax_us(profiled)  ms/solve")
for cap, rounds, low in settings:
    os.environ["WFST_SSSP_CHASE_CAP"] = str(cap)
    os.environ["WFST_SSSP_CHASE_ROUNDS"] = str(rounds)
    os.environ["WFST_SSSP_CHASE_LOW"] = str(low)
    for _ in range(4):
        out = d.shortest_path()
    flat = out.to_flat()
    if ref is None:
        ref = flat
    else:
        assert flat["n_states"] == ref["n_states"] and np.array_equal(flat["arcs"], ref["arcs"]) \
            and np.array_equal(flat["finals"].view(np.uint32), ref["finals"].view(np.uint32)), (cap, rounds, low)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 20
    for _ in range(N):
        d.shortest_path()
    ms = (time.perf_counter() - t0) / N * 1e3
    sweeps = ctx.stats()["sweeps"]
    ctx.reset_stats(); ctx.set_profiling(True); d.shortest_path(); ctx.set_profiling(False)
    st = ctx.stats()
    print(f"{cap:4d} {rounds:6d} {low:6d} {sweeps:6d} {st['relax_arcs']:14d} {st['relax_ms']*1e3:12.1f} {ms:14.4f}", flush=True)
    d.shortest_path()
