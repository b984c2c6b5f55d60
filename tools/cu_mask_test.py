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

run("no masks", rustfst_amd.Context(0), rustfst_amd.Context(0))
pats = [("S2 on CUs 0..31", range(32)), ("S2 on CUs 0..47", range(48)), ("S2 on CUs 0..63", range(64)),
        ("S2 on CUs 0..23", range(24)), ("S2 on CUs 224..255", range(224, 256)), ("S2 on CUs 0..15 + 128..143", list(range(16)) + list(range(128, 144))),
        ("S2 on CUs 0..7 of each 32", [c for c in range(256) if c % 32 < 8])]
for name, pat in pats:
    s_, big = masks(pat)
    run(name, rustfst_amd.Context(0, cu_mask=big), rustfst_amd.Context(0, cu_mask=s_))
s_, big = masks(range(32))
run("S2 on CUs 0..31, S1 unmasked", rustfst_amd.Context(0), rustfst_amd.Context(0, cu_mask=s_))
