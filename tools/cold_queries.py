"""shortest_path(T) on FRESH handles of the benched T in a warm process: host ms of the 1st / 2nd / 3rd / 4th query of each of N
handles (bench.py cold_query_ms.fresh_handle_warm_process), medians.  usage: cold_queries.py [states] [handles]
Knob under test: WFST_SSSP_TRANSPOSE_PLAN=0/1 (the transpose through the mailbox plan / by two atomic passes)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
handles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
ctx = rustfst_amd.Context(0)
warm = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
ref = None
for _ in range(30):
    ref = warm.shortest_path().to_flat()
rows = []
for h in range(handles):
    d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
    ctx.synchronize()
    ms = []
    for q in range(4):
        c0 = time.perf_counter(); got = d.shortest_path(); ms.append((time.perf_counter() - c0) * 1e3)
        g = got.to_flat()
        assert np.array_equal(g["arcs"], ref["arcs"]) and np.array_equal(g["finals"], ref["finals"])
    rows.append(ms)
    del d
med = [statistics.median(r[q] for r in rows[1:]) for q in range(4)]
print(f"plan={os.environ.get('WFST_SSSP_TRANSPOSE_PLAN', 'default')}: "
      f"1st {med[0]:.3f}  2nd {med[1]:.3f}  3rd {med[2]:.3f}  4th {med[3]:.3f} ms (median over {handles - 1} fresh handles; sum of the first two {med[0] + med[1]:.3f})")
