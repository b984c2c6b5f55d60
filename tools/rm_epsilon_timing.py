"""rm_epsilon on a large FST with sparse epsilon:epsilon arcs: GPU (host schedule + per-depth launches + connect) vs the
CPU oracle.  python tools/rm_epsilon_timing.py [states] [p_eps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustfst_amd
from rustfst_amd import synth
from oracle import oracle_py as O
from helpers import to_oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
t = synth.make_transducer(N, 10, 256, p, seed=21)
arcs = t["arcs"].copy()
arcs["olabel"] = np.where(arcs["ilabel"] == 0, 0, arcs["olabel"])  # the input epsilons become epsilon:epsilon arcs
t = dict(t); t["arcs"] = arcs; t["props"] = 0
ctx = rustfst_amd.default_context()
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
out = d.rm_epsilon()  # warm-up (pool growth, host copy)
ctx.synchronize()
t0 = time.perf_counter(); out = d.rm_epsilon(); ctx.synchronize(); t1 = time.perf_counter()
print(f"GPU rm_epsilon: {N} states / {len(arcs)} arcs ({int((arcs['ilabel'] == 0).sum())} epsilon:epsilon) -> "
      f"{out.num_states} states / {out.num_arcs} arcs in {(t1 - t0) * 1e3:.1f} ms", flush=True)
if N <= 1_000_000:
    o = to_oracle(O, t)
    t0 = time.perf_counter(); o.rm_epsilon(); t1 = time.perf_counter()
    f1, f2 = out.to_flat(), o.to_flat()
    same = f1["n_states"] == f2["n_states"] and np.array_equal(f1["arcs"], f2["arcs"]) and np.array_equal(f1["finals"].view(np.uint32), f2["finals"].view(np.uint32))
    print(f"oracle: {(t1 - t0) * 1e3:.1f} ms; identical: {same}")
