"""configs[1] (one 1000-arc string o T(100k / 1M)) and the 64 x 200 batch against T(1M / 10M): best-of-N host clock of the
fused compose -> shortest_path call with the string kernel's scalar-row path on / off (WFST_STRING_SCALAR)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

ctx = rustfst_amd.Context(0)
def dev(f):
    return rustfst_amd.DeviceFst.from_arrays(f["n_states"], f["start"], f["offsets"], f["arcs"], f["finals"], f["props"], ctx)

t2 = synth.make_transducer(100_000, 10, 256, 0.0, seed=2)
a2 = synth.make_acceptors(t2, 1, 1000, seed0=2)
d2 = dev(t2)
da2 = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(a2, ctx))
t3 = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
a3 = synth.make_acceptors(t3, 64, 200, seed0=1000)
d3 = dev(t3)
da3 = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(a3, ctx))
ref = {}
for scalar in ("1", "0", "1"):
    os.environ["WFST_STRING_SCALAR"] = scalar
    for name, da, d, reps in (("configs[1] single 1000-arc string", da2, d2, 30), ("64 x 200 batch vs 1M-state T", da3, d3, 30)):
        for _ in range(3):
            outs, n = rustfst_amd.compose_shortest_path_batch(da, d, ctx=ctx)
        best = 1e9
        for _ in range(reps):
            c0 = time.perf_counter(); outs, n = rustfst_amd.compose_shortest_path_batch(da, d, ctx=ctx); best = min(best, time.perf_counter() - c0)
        sig = [(o.to_flat()["arcs"].tobytes(), o.to_flat()["finals"].tobytes()) for o in outs]
        same = ref.setdefault(name, sig) == sig
        print(f"scalar={scalar} {name:36s} {best*1e3:.4f} ms  composed arcs {n}  string problems {ctx.stats()['string_problems']}  same result {same}", flush=True)
