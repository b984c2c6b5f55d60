"""Where job.finish() spends its time once the kernel is done: the C call, the Python wrapper, releasing the previous results."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rustfst_amd
from rustfst_amd import synth, _lib
from rustfst_amd.fst import PathList
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
daccs = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx))
for _ in range(3): rustfst_amd.compose_shortest_path_batch(daccs, dt)
L = _lib.lib()
rows = []
keep = None
for it in range(60):
    job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = (C.c_void_p * 64)(); na = C.c_uint64()
    t1 = time.perf_counter()
    h, job._job = job._job, None
    rc = L.wfst_compose_shortest_path_batch_end(h, outs, C.byref(na))
    t2 = time.perf_counter()
    pl = PathList(outs, 64, ctx)
    t3 = time.perf_counter()
    keep = pl   # releases the previous step's 64 results
    t4 = time.perf_counter()
    if it >= 10: rows.append([(t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6])
r = np.median(np.array(rows), axis=0)
print("ctypes arrays %.1f | C _end %.1f | PathList %.1f | releasing the previous results %.1f us" % tuple(r))
