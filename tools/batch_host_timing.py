import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rustfst_amd
from rustfst_amd import synth
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)
ctx = rustfst_amd.default_context()
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
daccs = rustfst_amd.DeviceFst.upload_many(accs, ctx)
for _ in range(3): rustfst_amd.compose_shortest_path_batch(daccs, dt)
os.environ["WFST_HOST_TIMING"] = "1"
for _ in range(3):
    job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt)
    torch.cuda.synchronize()   # kernel done: what remains in finish() is host work
    a = time.perf_counter(); outs, na = job.finish(); b = time.perf_counter()
    print("finish after the kernel is done: %.1f us" % ((b - a) * 1e6))
