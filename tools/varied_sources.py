"""shortest_path(T) from random sources on ONE resident handle (wfst_fst_set_start between queries): host ms per query, launches
per query (bench.py `varied_sources` alone).   python tools/varied_sources.py [states] [queries] [fan-out]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 48
fan = int(sys.argv[3]) if len(sys.argv) > 3 else 10
t = synth.make_transducer(states, fan, 256, 0.0, seed=3)
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for _ in range(6):
    d.shortest_path()
rng = np.random.default_rng(20261001)
ms, launches = [], []
for s in rng.integers(0, states, nq):
    d.set_start(int(s))
    ctx.synchronize()
    c0 = time.perf_counter()
    d.shortest_path()
    ms.append(1e3 * (time.perf_counter() - c0))
    launches.append(int(ctx.stats()["sweeps"]))
ms_s = sorted(ms)
print(f"{states} states, fan-out {fan}, kernel {ctx.stats()['relax_kernel']}: {nq} sources: median {ms_s[nq // 2]:.4f} ms, mean {sum(ms) / nq:.4f}, min {ms_s[0]:.4f}, max {ms_s[-1]:.4f}; launches per query "
      f"{sum(launches) / nq:.2f} (histogram {dict(zip(*np.unique(launches, return_counts=True)))})")
print("slowest:", [round(x, 3) for x in ms_s[-6:]])
