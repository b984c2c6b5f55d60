"""Fused batch of B acceptors (len 200) against T(1M): host ms per synchronous call, handles and packed (best of 7)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rustfst_amd
from rustfst_amd import synth
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for B in (64, 512, 1024, 2048, 4096):
    accs = synth.make_acceptors(t, B, 200, seed0=5000)
    d = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx))
    best = [1e9, 1e9]
    for rep in range(7):
        a = time.perf_counter(); outs, na = rustfst_amd.compose_shortest_path_batch(d, dt); b = time.perf_counter()
        tab, na2 = rustfst_amd.compose_shortest_path_batch_packed(d, dt, 208); c = time.perf_counter()
        del outs
        if rep: best = [min(best[0], b - a), min(best[1], c - b)]
    print(f"B={B}: handles {best[0]*1e3:.3f} ms, packed {best[1]*1e3:.3f} ms")
