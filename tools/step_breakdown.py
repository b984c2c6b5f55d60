"""Host-side timeline of one overlapped bench step: begin (enqueue batch) | shortest_path(T) | finish (collect batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustfst_amd
from rustfst_amd import synth
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
ctx, ctx2 = rustfst_amd.Context(0, stream=s1.cuda_stream), rustfst_amd.Context(0, stream=s2.cuda_stream)
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
daccs = rustfst_amd.DeviceFst.upload_many(accs, ctx2)
daccs = rustfst_amd.HandleArray(daccs)
acc = np.zeros(5)
N = 200
for it in range(N + 10):  # bench.py's schedule: batch begin, shortest_path begin, batch finish, shortest_path finish
    a = time.perf_counter()
    job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt, ctx=ctx2)
    b = time.perf_counter()
    sp_job = dt.shortest_path_begin()
    c = time.perf_counter()
    outs, na = job.finish()
    d = time.perf_counter()
    sp = sp_job.finish()
    e = time.perf_counter()
    if it >= 10:
        acc += [b - a, c - b, d - c, e - d, e - a]
print("batch begin %.1f us | shortest_path begin %.1f us | batch finish %.1f us | shortest_path finish %.1f us | step %.1f us" % tuple(acc / N * 1e6))
# each alone
for name, fn in (("shortest_path(T) alone", lambda: dt.shortest_path()),
                 ("batch alone", lambda: rustfst_amd.compose_shortest_path_batch(daccs, dt, ctx=ctx2))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(N): fn()
    torch.cuda.synchronize(); print(name, "%.1f us" % ((time.perf_counter() - a) / N * 1e6))
