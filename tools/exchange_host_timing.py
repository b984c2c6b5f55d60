"""Host cost of the result exchange's two calls at one rank (RCCL path): gather_paths_begin / gather_paths_end / order_after."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rustfst_amd
from rustfst_amd import synth, dist
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
daccs = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx))
comm = dist.Comm(ctx, dist.Comm.unique_id(), 0, 1)
outs, _ = rustfst_amd.compose_shortest_path_batch(daccs, dt)
rows = []
for it in range(60):
    torch.cuda.synchronize()
    a = time.perf_counter(); comm.order_after(ctx); b = time.perf_counter(); comm.gather_paths_begin(outs, 208); c = time.perf_counter()
    torch.cuda.synchronize()
    d = time.perf_counter(); g = comm.gather_paths_end(); e = time.perf_counter()
    if it >= 10: rows.append([(b - a) * 1e6, (c - b) * 1e6, (e - d) * 1e6])
r = np.median(np.array(rows), axis=0)
print("order_after %.1f us | gather_paths_begin %.1f us | gather_paths_end (device done) %.1f us" % tuple(r))
os.environ["WFST_HOST_TIMING"] = "1"
comm.gather_paths_begin(outs, 208); torch.cuda.synchronize(); comm.gather_paths_end()
