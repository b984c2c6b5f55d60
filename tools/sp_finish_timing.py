"""Host cost of finishing a shortest_path(T) job whose GPU work is already complete, and of beginning one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rustfst_amd
from rustfst_amd import synth
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for _ in range(5): dt.shortest_path()
b, f = [], []
for _ in range(50):
    t0 = time.perf_counter(); job = dt.shortest_path_begin(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter(); sp = job.finish(); t3 = time.perf_counter()
    b.append((t1 - t0) * 1e6); f.append((t3 - t2) * 1e6)
print("begin med %.1f us, finish after the GPU is done med %.1f us (min %.1f)" % (np.median(b), np.median(f), min(f)))
