"""bench.py's `config5.wide_lookahead` workload alone (for rocprofv3 and A/B): ONE acceptor of 40 labels composed with the
5M-state look-ahead operand at |Sigma| = 8 on the wide look-ahead driver.  usage: wide_lookahead_run.py [states] [reps] [len]
Prints first-call and per-repetition milliseconds, states / arcs composed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ctx = rustfst_amd.Context(0)
tw = synth.make_transducer(n, 10, 8, 0.05, seed=9)
aw = synth.make_acceptors(tw, 1, L, seed0=77)[0]
tw["arcs"]["ilabel"], tw["arcs"]["olabel"] = tw["arcs"]["olabel"].copy(), tw["arcs"]["ilabel"].copy()
tw["props"] = synth.O_LABEL_SORTED
dw = rustfst_amd.DeviceFst.from_arrays(tw["n_states"], tw["start"], tw["offsets"], tw["arcs"], tw["finals"], tw["props"], ctx)
del tw
c0 = time.perf_counter(); law = rustfst_amd.LookAhead(dw); print(f"look-ahead create {time.perf_counter() - c0:.3f} s", flush=True)
daw = rustfst_amd.DeviceFst.from_arrays(aw["n_states"], aw["start"], aw["offsets"], aw["arcs"], aw["finals"], aw["props"], ctx)
relw = law.relabel(daw)
times = []
ref = None
for r in range(reps + 1):
    ctx.synchronize(); c0 = time.perf_counter(); ow = law.compose(relw); ctx.synchronize(); times.append((time.perf_counter() - c0) * 1e3)
    st = ctx.stats()
    sig = (int(ow.num_states), int(st["compose_states"]), int(st["compose_arcs"]))
    assert ref is None or sig == ref, (sig, ref)
    ref = sig
    del ow
print(f"first call {times[0]:.2f} ms; then {' '.join(f'{t:.2f}' for t in times[1:])} ms; best {min(times[1:]):.2f} ms; "
      f"result {ref[0]} states, composed {ref[1]} states / {ref[2]} arcs", flush=True)
