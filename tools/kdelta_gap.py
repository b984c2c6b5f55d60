"""How far is the reference's shortest path (approximate TropicalWeight ==, KDELTA = 1/1024:
rustfst/src/semirings/semiring.rs:159-168, shortest_path.rs:226-228; FIFO / top-order AutoQueue) from the exact
(min,+) fixed point this engine returns, on REAL-VALUED (non-grid) weights?

For every instance: w_gpu = weight of the GPU path (left-folded f32 sum), w_exact = the CPU restatement with exact ==,
w_ref = the CPU restatement in reference mode.  Reports, per shape, the number of instances, how many have
|w_gpu - w_ref| > 1e-5, the largest gap, and checks w_gpu == w_exact bit for bit and w_gpu <= w_ref.

  python tools/kdelta_gap.py [quick]        (quick = the subset the -m gpu test runs)
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustfst_amd
from rustfst_amd import synth
from oracle import oracle_py as O

GOLDEN = os.path.join(ROOT, "tests", "golden")


def real_weights(flat, seed, scale=10.0):
    """uniform f32 weights in [0, scale) that are not on any coarse grid"""
    f = dict(flat)
    arcs = flat["arcs"].copy()
    r = synth.splitmix64(seed, len(arcs), 21)
    arcs["weight"] = ((r >> np.uint64(40)).astype(np.float64) * (scale / (1 << 24))).astype(np.float32)
    fin = flat["finals"].copy()
    rf = synth.splitmix64(seed, len(fin), 22)
    fin = np.where(np.isfinite(fin), ((rf >> np.uint64(40)).astype(np.float64) * (scale / (1 << 24))).astype(np.float32), fin).astype(np.float32)
    f["arcs"], f["finals"] = arcs, fin
    return f


def path_weight(flat):
    if flat["n_states"] == 0:
        return np.float32(np.inf)
    acc = np.float32(0.0)
    for w in flat["arcs"]["weight"][::-1]:
        acc = np.float32(acc + w)
    return np.float32(acc + flat["finals"][0])


def dev(f, ctx):
    return rustfst_amd.DeviceFst.from_arrays(f["n_states"], f["start"], f["offsets"], f["arcs"], f["finals"], f["props"], ctx)


def orc(f):
    return O.OracleFst.from_flat(f["n_states"], f["start"], f["offsets"], f["arcs"], f["finals"], f["props"])


def measure(name, cases, rows):
    n = over = 0
    gmax = 0.0
    for d_fst, o_fst in cases:
        w_gpu = path_weight(d_fst.shortest_path().to_flat())
        w_exact = np.float32(o_fst.shortest_path(eq_mode=O.EQ_EXACT).total_weight)
        w_ref = np.float32(o_fst.shortest_path(eq_mode=O.EQ_REF_KDELTA).total_weight)
        assert w_gpu.view(np.uint32) == w_exact.view(np.uint32), (name, w_gpu, w_exact)
        assert not (w_gpu > w_ref), (name, w_gpu, w_ref)
        gap = float(w_ref) - float(w_gpu) if np.isfinite(w_ref) else 0.0
        n += 1
        over += gap > 1e-5
        gmax = max(gmax, gap)
    rows.append((name, n, over, gmax))
    print(f"{name:58s} instances {n:3d}  |gap| > 1e-5: {over:3d}  max gap {gmax:.3e}", flush=True)


def main(quick=False):
    ctx = rustfst_amd.Context(0)
    rows = []
    # (1) shortest path directly on T, real-valued weights (C2 / C3 shapes)
    sizes = [(100_000, 3)] if quick else [(100_000, 6), (1_000_000, 2)]
    for n_states, k in sizes:
        cases = []
        for seed in range(k):
            t = real_weights(synth.make_transducer(n_states, 10, 256, 0.0, seed=3 + seed), 100 + seed)
            cases.append((dev(t, ctx), orc(t)))
        measure(f"shortest_path(T {n_states} states / {10*n_states} arcs), U[0,10) f32 weights", cases, rows)
    # (2) composed lattices A(L) o T, real-valued weights on T
    for n_states, L, k in ([(100_000, 200, 4), (100_000, 1000, 2)] if quick else [(100_000, 200, 16), (100_000, 1000, 8), (1_000_000, 200, 8), (1_000_000, 1000, 4)]):
        t = real_weights(synth.make_transducer(n_states, 10, 256, 0.0, seed=2), 7)
        accs = synth.make_acceptors(t, k, L, seed0=500)
        dt, ot = dev(t, ctx), orc(t)
        cases = [(dev(a, ctx).compose(dt), orc(a).compose(ot)) for a in accs]
        measure(f"shortest_path(A(L={L}) o T({n_states})), U[0,10) f32 weights", cases, rows)
        # the same through the fused batch kernel (string o T)
        outs, _ = rustfst_amd.compose_shortest_path_batch(rustfst_amd.DeviceFst.upload_many(accs, ctx), dt)
        for o, (_, oc) in zip(outs, cases):
            w = path_weight(o.to_flat())
            assert w.view(np.uint32) == np.float32(oc.shortest_path(eq_mode=O.EQ_EXACT).total_weight).view(np.uint32)
    # (2b) where the approximation bites: weights far below KDELTA = 1/1024 (every improvement is "equal" to the reference)
    for scale in ((0.01,) if quick else (0.1, 0.01, 0.001)):
        t = real_weights(synth.make_transducer(100_000, 10, 256, 0.0, seed=2), 9, scale=scale)
        accs = synth.make_acceptors(t, 4 if quick else 16, 200, seed0=900)
        dt, ot = dev(t, ctx), orc(t)
        measure(f"STRESS shortest_path(A(L=200) o T(100000)), U[0,{scale}) f32 weights", [(dev(a, ctx).compose(dt), orc(a).compose(ot)) for a in accs], rows)
        measure(f"STRESS shortest_path(T 100000 states), U[0,{scale}) f32 weights", [(dt, ot)], rows)
    # (3) the reference's own HCL o G pairs (their weights are real-valued already)
    for hcl, g in (("fst_014_hcl.fst", "fst_014_g.fst"), ("fst_012_hcl.fst", "fst_012_gp.fst")):
        da, db = (open(os.path.join(GOLDEN, x), "rb").read() for x in (hcl, g))
        a, b = rustfst_amd.DeviceFst.from_bytes(da, ctx), rustfst_amd.DeviceFst.from_bytes(db, ctx)
        oa, ob = O.OracleFst.load(da), O.OracleFst.load(db)
        a.tr_sort(False), b.tr_sort(True)
        oa.tr_sort(by_olabel=True), ob.tr_sort(by_olabel=False)
        measure(f"shortest_path({hcl} o {g}), weights of the files", [(a.compose(b), oa.compose(ob))], rows)
    return rows


if __name__ == "__main__":
    rows = main(quick=len(sys.argv) > 1 and sys.argv[1] == "quick")
    print(json.dumps([{"shape": r[0], "instances": r[1], "over_1e-5": int(r[2]), "max_gap": r[3]} for r in rows]))
