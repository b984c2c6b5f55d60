"""BASELINE configs[4]: HCLG-shaped FST (5M states / 50M arcs, 5 % epsilon arcs) under look-ahead composition + n = 10
shortest paths.  The big FST is the FIRST operand, as in look-ahead decoding graphs (HCL o G): its reachability data is
computed once (wfst_lookahead_create), then every acceptor is relabelled, composed with the look-ahead filter stack and
searched for 10 paths.  Checks that need no oracle at this size: the plain composition of the same pair has the same n-best
weights; every path spells the acceptor.  usage: python tools/config5_lookahead.py [states] [n_acceptors]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth, ShortestPathConfig, ComposeConfig

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
NA = int(sys.argv[2]) if len(sys.argv) > 2 else 16
t0 = time.time()
t = synth.make_transducer(N, 10, 256, 0.05, seed=9)
accs = synth.make_acceptors(t, NA, 200, seed0=77)  # walks over t's ilabels (marks their end states final)
# the look-ahead operand emits what the acceptors read: swap t's label columns (its 5 % epsilons are now on the output side)
arcs = t["arcs"].copy()
arcs["ilabel"], arcs["olabel"] = t["arcs"]["olabel"].copy(), t["arcs"]["ilabel"].copy()
t1 = dict(t); t1["arcs"] = arcs; t1["props"] = synth.O_LABEL_SORTED  # (make_transducer sorts by ilabel = the new olabel)
print("generated in %.1f s: %d states, %d arcs" % (time.time() - t0, N, t["offsets"][-1]), flush=True)
ctx = rustfst_amd.default_context()
d1 = rustfst_amd.DeviceFst.from_arrays(N, t1["start"], t1["offsets"], t1["arcs"], t1["finals"], t1["props"], ctx)
t0 = time.perf_counter(); la = rustfst_amd.LookAhead(d1); dt_la = time.perf_counter() - t0
info = la.data() if N <= 200_000 else None
print("wfst_lookahead_create (host reachability + relabel + upload): %.2f s" % dt_la, flush=True)
cfg10 = ShortestPathConfig(nshortest=10)
tt = dict(relabel=0.0, compose=0.0, nbest=0.0, plain=0.0)
states = []
for i, a in enumerate(accs):
    da = rustfst_amd.DeviceFst.from_arrays(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"], ctx)
    x0 = time.perf_counter(); ar = la.relabel(da)
    x1 = time.perf_counter(); out = la.compose(ar)
    x2 = time.perf_counter(); nb = out.shortest_path(cfg10)
    x3 = time.perf_counter(); plain = d1.compose(da, ComposeConfig(connect=True))
    x4 = time.perf_counter()
    if i:  # (first iteration = warm-up: pool growth)
        tt["relabel"] += x1 - x0; tt["compose"] += x2 - x1; tt["nbest"] += x3 - x2; tt["plain"] += x4 - x3
    states.append(out.num_states)
    # same weighted relation: the 10 best path weights of the plain composition are the same
    def weights(f):
        f = f.to_flat()
        if f["n_states"] == 0:
            return []
        # path tree of n_shortest_path: sum weights along each path from the start
        off, arcs_, fin = f["offsets"], f["arcs"], f["finals"]
        res = []
        stack = [(f["start"], 0.0)]
        while stack:
            s, w = stack.pop()
            if np.isfinite(fin[s]):
                res.append(round((w + float(fin[s])) * 512))
            for k in range(off[s], off[s + 1]):
                stack.append((int(arcs_[k]["nextstate"]), w + float(arcs_[k]["weight"])))
        return sorted(res)
    w_la, w_plain = weights(nb), weights(plain.shortest_path(cfg10))
    assert w_la == w_plain and len(w_la) >= 1, (i, w_la, w_plain)
    best = out.shortest_path().to_flat()
    ol = best["arcs"]["olabel"][::-1]
    assert np.array_equal(ol[ol != 0], a["arcs"]["olabel"]), i  # the best path spells the acceptor
k = max(1, NA - 1)
print("per acceptor (len 200), mean over %d: relabel %.3f ms, look-ahead compose %.3f ms (%d..%d composed states), n=10 %.3f ms; "
      "plain compose+connect of the same pair %.3f ms" % (k, tt["relabel"] / k * 1e3, tt["compose"] / k * 1e3, min(states), max(states),
                                                          tt["nbest"] / k * 1e3, tt["plain"] / k * 1e3))
# the same problems as ONE batch (one wave per acceptor)
das = [rustfst_amd.DeviceFst.from_arrays(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"], ctx) for a in accs]
rel = [la.relabel(d) for d in das]
outs = la.compose_batch(rel)
ctx.synchronize()
x0 = time.perf_counter(); outs = la.compose_batch(rel); ctx.synchronize(); x1 = time.perf_counter()
assert [o.num_states for o in outs] == states
print("batch of %d look-ahead compositions in one launch: %.3f ms (%.3f ms each)" % (NA, (x1 - x0) * 1e3, (x1 - x0) * 1e3 / NA))
print("OK")
