"""Does giving the fused batch its own CUs (hipExtStreamCreateWithCUMask) remove the interference of the relaxation sweeps?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustfst_amd
from rustfst_amd import synth
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)

def masks(pattern):
    small = np.zeros(8, dtype=np.uint32)
    for cu in pattern:
        small[cu // 32] |= np.uint32(1) << np.uint32(cu % 32)
    return small, ~small

def run(name, ctx, ctx2):
    dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
    daccs = rustfst_amd.DeviceFst.upload_many(accs, ctx2)
    acc = np.zeros(4); N = 30
    for it in range(N + 5):
        torch.cuda.synchronize()
        a = time.perf_counter()
        job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt, ctx=ctx2)
        b = time.perf_counter()
        sp = dt.shortest_path()
        c = time.perf_counter()
        outs, na = job.finish()
        d = time.perf_counter()
        if it >= 5: acc += [b - a, c - b, d - c, d - a]
    print("%-44s begin %.1f | shortest_path(T) %.1f | finish %.1f | step %.1f us" % ((name,) + tuple(acc / N * 1e6)))

s_, big = masks(range(64))
run("S2 on CUs 0..63 (bench default)", rustfst_amd.Context(0, cu_mask=big), rustfst_amd.Context(0, cu_mask=s_))
run("no masks", rustfst_amd.Context(0), rustfst_amd.Context(0))
