"""Probe for bench.py's configs[4] extra: (1) the CPU restatement at the FULL operand size (how long does it take?),
(2) a look-ahead composition whose result is large (|Sigma| = 8: every acceptor label matches ~1.25 arcs per state, the
lattice grows by that factor per level) so that the wide look-ahead driver is what runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
do_cpu = len(sys.argv) > 2 and sys.argv[2] == "cpu"
ctx = rustfst_amd.Context(0)

def build(n, n_acc, seed, sigma, L):
    t5 = synth.make_transducer(n, 10, sigma, 0.05, seed=seed)
    accs = synth.make_acceptors(t5, n_acc, L, seed0=77)
    arcs = t5["arcs"].copy()
    arcs["ilabel"], arcs["olabel"] = t5["arcs"]["olabel"].copy(), t5["arcs"]["ilabel"].copy()
    t1 = dict(t5)
    t1["arcs"], t1["props"] = arcs, synth.O_LABEL_SORTED
    return t1, accs

for sigma, Ls in ((8, (40, 48, 56)),):
    c0 = time.perf_counter()
    t1, _ = build(n, 1, 9, sigma, 8)
    print(f"sigma {sigma}: generated {n} states in {time.perf_counter()-c0:.1f} s", flush=True)
    d1 = rustfst_amd.DeviceFst.from_arrays(t1["n_states"], t1["start"], t1["offsets"], t1["arcs"], t1["finals"], t1["props"], ctx)
    c0 = time.perf_counter(); la = rustfst_amd.LookAhead(d1); print(f"  look-ahead create {time.perf_counter()-c0:.2f} s", flush=True)
    for L in Ls:
        t5 = dict(t1); 
        # acceptors must be walks in the ORIGINAL orientation: rebuild from the same seed
        tt = synth.make_transducer(n, 10, sigma, 0.05, seed=9)
        acc = synth.make_acceptors(tt, 1, L, seed0=77)[0]
        da = rustfst_amd.DeviceFst.from_arrays(acc["n_states"], acc["start"], acc["offsets"], acc["arcs"], acc["finals"], acc["props"], ctx)
        rel = la.relabel(da)
        for rep in range(2):
            ctx.synchronize(); c0 = time.perf_counter(); out = la.compose(rel); ctx.synchronize(); dt = time.perf_counter() - c0
        st = ctx.stats()
        print(f"  L={L}: composed {out.num_states} states (pre-trim {st['compose_states']} states / {st['compose_arcs']} arcs) in {dt*1e3:.2f} ms", flush=True)
if do_cpu:
    from oracle import oracle_py
    t1c, accs_c = build(n, 1, 9, 256, 200)
    c0 = time.perf_counter()
    o1 = oracle_py.OracleFst.from_flat(t1c["n_states"], t1c["start"], t1c["offsets"], t1c["arcs"], t1c["finals"], t1c["props"])
    a = accs_c[0]
    oa = oracle_py.OracleFst.from_flat(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"])
    print(f"cpu: oracle FST built in {time.perf_counter()-c0:.1f} s", flush=True)
    c0 = time.perf_counter(); oc = o1.compose_lookahead(oa); print(f"cpu: look-ahead precompute + one composition at {n} states: {time.perf_counter()-c0:.1f} s", flush=True)
