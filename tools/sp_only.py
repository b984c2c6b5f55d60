import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rustfst_amd
from rustfst_amd import synth
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
synth.make_acceptors(t, 64, 200, seed0=1000)
ctx = rustfst_amd.default_context()
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for _ in range(8):
    d.shortest_path()
