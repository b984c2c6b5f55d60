"""Device / host memory stays flat over thousands of calls (pool reuse, handle destruction)."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustfst_amd
from rustfst_amd import synth, ShortestPathConfig
t = synth.make_transducer(100_000, 8, 64, 0.02, seed=5)
accs = synth.make_acceptors(t, 32, 50, seed0=10)
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
da = rustfst_amd.DeviceFst.upload_many(accs, ctx)
def snap():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for rnd in range(4):
    for i in range(800):
        outs, _ = rustfst_amd.compose_shortest_path_batch(da, dt)
        sp = dt.shortest_path()
        if i % 50 == 0:
            c = da[i % 32].compose(dt); c.shortest_path(ShortestPathConfig(nshortest=3)); c2 = rustfst_amd.DeviceFst.from_bytes(c.to_bytes("const")); c2.tr_sort(False)
    print("round %d: device used %.0f MiB, host max RSS %.0f MiB" % ((rnd,) + snap()), flush=True)
