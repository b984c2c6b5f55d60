import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, rustfst_amd
from rustfst_amd import synth
N=1_000_000
t = synth.make_transducer(N, 10, 256, 0.05, seed=9)
accs = synth.make_acceptors(t, 64, 200, seed0=77)
arcs = t["arcs"].copy(); arcs["ilabel"], arcs["olabel"] = t["arcs"]["olabel"].copy(), t["arcs"]["ilabel"].copy()
t1 = dict(t); t1["arcs"] = arcs; t1["props"] = synth.O_LABEL_SORTED
ctx = rustfst_amd.default_context()
d1 = rustfst_amd.DeviceFst.from_arrays(N, t1["start"], t1["offsets"], t1["arcs"], t1["finals"], t1["props"], ctx)
la = rustfst_amd.LookAhead(d1)
rel = [la.relabel(rustfst_amd.DeviceFst.from_arrays(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"], ctx)) for a in accs]
outs = la.compose_batch(rel)
import hashlib
h = hashlib.sha1()
for o in outs:
    f = o.to_flat(); h.update(f["arcs"].tobytes()); h.update(f["finals"].tobytes())
best = 1e9
for _ in range(8):
    ctx.synchronize(); c0 = time.perf_counter(); outs = la.compose_batch(rel); ctx.synchronize(); best = min(best, time.perf_counter() - c0)
print("lookahead compose batch of 64: %.3f ms; states %d..%d; digest %s" % (best * 1e3, min(o.num_states for o in outs), max(o.num_states for o in outs), h.hexdigest()[:12]))
