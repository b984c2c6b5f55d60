"""BASELINE configs[1]: T 100k states / 1M arcs, one linear acceptor of 1000 arcs: single-problem latency, GPU vs CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rustfst_amd
from rustfst_amd import synth
from oracle import oracle_py as O
t = synth.make_transducer(100_000, 10, 256, 0.0, seed=3)
acc = synth.make_acceptors(t, 1, 1000, seed0=42)[0]
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
da = rustfst_amd.DeviceFst.upload_many([acc], ctx)[0]
def best(fn, n=10):
    b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); a = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - a)
    return b * 1e3
print("GPU compose(A,T)                  %.3f ms" % best(lambda: da.compose(dt)))
print("GPU compose+shortest_path (fused) %.3f ms" % best(lambda: rustfst_amd.compose_shortest_path_batch([da], dt)))
c = da.compose(dt)
print("   composed: %d states, %d arcs" % (c.num_states, c.num_arcs))
ot = O.OracleFst.from_flat(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"])
oa = O.OracleFst.from_flat(acc["n_states"], acc["start"], acc["offsets"], acc["arcs"], acc["finals"], acc["props"])
def cbest(fn, n=10):
    b = 1e9
    for _ in range(n):
        a = time.perf_counter(); fn(); b = min(b, time.perf_counter() - a)
    return b * 1e3
print("CPU oracle compose(A,T)           %.3f ms" % cbest(lambda: oa.compose(ot)))
print("CPU oracle compose+shortest_path  %.3f ms" % cbest(lambda: oa.compose(ot).shortest_path()))
