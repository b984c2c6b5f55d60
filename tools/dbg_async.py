import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rustfst_amd
from rustfst_amd import synth
t = synth.make_transducer(60000, 8, 64, 0.0, seed=77)
ctx = rustfst_amd.default_context()
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for rep in range(3):
    os.environ["WFST_SSSP_DELTA"] = "0"
    few = []
    for q in range(4):
        job = d.shortest_path_begin(); r = job.finish(); few.append(ctx.stats()["sweeps"])
    os.environ["WFST_SSSP_DELTA"] = "0.5"
    job = d.shortest_path_begin(); r = job.finish()
    st = ctx.stats()
    print(few, st["sweeps"], "kernel", st["relax_kernel"], "aborts", st["resident_aborts"])
