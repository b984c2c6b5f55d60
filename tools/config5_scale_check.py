"""BASELINE configs[4] scale (5M states / 50M arcs, 5 % epsilon arcs): shortest path n=1 and n=10, a fused batch, tr_sort —
size-independent properties only (the oracle would take minutes here)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth, ShortestPathConfig

t0 = time.time()
t = synth.make_transducer(5_000_000, 10, 256, 0.05, seed=9)
print("generated in %.1f s: %d states, %d arcs" % (time.time() - t0, t["n_states"], t["offsets"][-1]))
accs = synth.make_acceptors(t, 64, 200, seed0=77)  # (marks the walks' end states final in t: before the upload)
ctx = rustfst_amd.default_context()
t0 = time.time()
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
print("uploaded in %.2f s" % (time.time() - t0))
for q in range(3):
    t0 = time.perf_counter(); sp = d.shortest_path(); dt_ = time.perf_counter() - t0
    f = sp.to_flat()
    w = float(np.add.reduce(f["arcs"]["weight"][::-1].astype(np.float32), dtype=np.float32) + f["finals"][0]) if f["n_states"] else float("inf")
    print("shortest_path query %d: %.2f ms, %d arcs, weight %.6f, sweeps %d" % (q, dt_ * 1e3, max(f["n_states"] - 1, 0), w, ctx.stats()["sweeps"]))
dist = d.shortest_distance()
fin = t["finals"]
best = np.min(np.where(np.isfinite(fin) & np.isfinite(dist), dist + fin, np.inf))
assert abs(best - w) < 1e-4, (best, w)
# triangle inequality on a sample of arcs: d[t] <= d[s] + w
off = t["offsets"].astype(np.int64); src = np.repeat(np.arange(t["n_states"]), np.diff(off))
idx = np.random.default_rng(0).integers(0, len(src), 2_000_000)
a = t["arcs"][idx]
ok = dist[a["nextstate"]] <= dist[src[idx]] + a["weight"] + 1e-4
assert ok[np.isfinite(dist[src[idx]])].all()
t0 = time.perf_counter(); nb = d.shortest_path(ShortestPathConfig(nshortest=10)); print("n=10: %.1f ms, %d states" % ((time.perf_counter() - t0) * 1e3, nb.num_states))
da = rustfst_amd.DeviceFst.upload_many(accs, ctx)
t0 = time.perf_counter(); outs, na = rustfst_amd.compose_shortest_path_batch(da, d); print("batch of 64: %.2f ms, %d composed arcs" % ((time.perf_counter() - t0) * 1e3, na))
for a, o in zip(accs, outs):  # every random walk is accepted; the path reads exactly the acceptor's labels (T's input epsilons aside)
    f = o.to_flat()
    assert f["n_states"] >= 201
    il = f["arcs"]["ilabel"][::-1]  # the path FST is numbered backwards
    assert np.array_equal(il[il != 0], a["arcs"]["ilabel"])
t0 = time.perf_counter(); d.tr_sort(False); ctx.synchronize(); print("tr_sort(olabel) of 50M arcs: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
print("OK")
