"""shortest_path(T(1M, fan-out 10)) under other weight distributions than the benchmark's uniform k/512 in [0, 10): relaxation chain
(HIP events), launches.   python tools/weight_shapes.py [states]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, rustfst_amd
from rustfst_amd import synth
states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
base = synth.make_transducer(states, 10, 256, 0.0, seed=3)
rng = np.random.default_rng(5)
E_ = len(base["arcs"])
fams = {
    "uniform [0, 10) (the benchmark's)": None,
    "integers 1..4": rng.integers(1, 5, E_).astype(np.float32),
    "all 1": np.ones(E_, np.float32),
    "exponential, mean 5": (np.floor(rng.exponential(5.0, E_) * 512) / 512).astype(np.float32),
    "lognormal (sigma 1.5)": (np.floor(rng.lognormal(0.0, 1.5, E_) * 512) / 512).astype(np.float32),
    "uniform [9, 10)": (9 + np.floor(rng.random(E_) * 512) / 512).astype(np.float32),
    "two classes: 0.01 (10 %) and 10": np.where(rng.random(E_) < 0.1, np.float32(0.0078125), np.float32(10.0)).astype(np.float32),
}
ctx = rustfst_amd.Context(0)
only = os.environ.get("ONLY")
for name, w in fams.items():
    if only and only not in name:
        continue
    arcs = base["arcs"].copy()
    if w is not None:
        arcs["weight"] = w
    d = rustfst_amd.DeviceFst.from_arrays(base["n_states"], base["start"], base["offsets"], arcs, base["finals"], base["props"], ctx)
    for _ in range(5):
        d.shortest_path()
    ctx.set_profiling(2)
    v = []
    for _ in range(7):
        d.shortest_path()
        st = ctx.stats()
        v.append(1e3 * st["relax_ms"])
    ctx.set_profiling(0)
    if w is not None:
        mins = np.minimum.reduceat(w, base["offsets"][:-1].astype(np.int64))
        name += f" [mean {w.mean():.2f}, mean of per-state minima {mins.mean():.3f}]"
    print(f"{name:36s}: chain {statistics.median(v):8.1f} us in {st['relax_launches']} launches (kernel {st['relax_kernel']})", flush=True)
    del d
