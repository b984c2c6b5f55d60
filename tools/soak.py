"""Randomised differential soak: GPU path vs the CPU oracle on many seeded inputs (compose with every filter,
connect on/off, shortest path canonical, n-best, tr_sort, fused batch).  Not part of the test suite: run it on a GPU
box for a few minutes (python tools/soak.py [seconds] [seed0])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustfst_amd
from rustfst_amd import ComposeConfig, ComposeFilter, ShortestPathConfig
from oracle import oracle_py as O
from helpers import assert_flat_identical, random_fst_flat, to_device, to_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
n = dict(compose=0, sp=0, nbest=0, sort=0, batch=0)
seed = seed0
FILTERS = list(ComposeFilter)
while time.time() < t_end:
    rng = np.random.default_rng(10_000 + seed)
    kind = seed % 5
    try:
        if kind in (0, 1):
            n1, n2 = int(rng.integers(1, 60)), int(rng.integers(1, 60))
            f1 = int(rng.integers(1, 9)) if rng.random() < 0.9 else int(rng.integers(60, 100))
            f2 = int(rng.integers(1, 9)) if rng.random() < 0.9 else int(rng.integers(60, 100))
            sig = int(rng.integers(1, 8))
            s1 = "olabel" if rng.random() < 0.85 else "none"
            s2 = "ilabel" if (rng.random() < 0.85 or s1 == "none") else "none"
            a = random_fst_flat(rng, n1, f1, sig, p_eps_i=rng.random() * 0.4, p_eps_o=rng.random() * 0.5, p_final=rng.random() * 0.6, sort=s1)
            b = random_fst_flat(rng, n2, f2, sig, p_eps_i=rng.random() * 0.5, p_eps_o=rng.random() * 0.4, p_final=rng.random() * 0.6, sort=s2)
            flt = FILTERS[int(rng.integers(0, len(FILTERS)))]
            connect = bool(rng.integers(0, 2))
            da, db, oa, ob = to_device(a), to_device(b), to_oracle(O, a), to_oracle(O, b)
            try:
                ref = oa.compose(ob, connect=connect, compose_filter=flt.value)
            except O.OracleError:
                try:
                    da.compose(db, ComposeConfig(flt, connect=connect))
                    raise AssertionError("oracle refused (unsorted) but the GPU path accepted")
                except rustfst_amd.WfstError:
                    pass
            else:
                got = da.compose(db, ComposeConfig(flt, connect=connect))
                assert_flat_identical(got.to_flat(), ref.to_flat(), f"compose {flt.name} connect={connect}")
                if ref.num_states and seed % 2:
                    assert_flat_identical(got.shortest_path().to_flat(), ref.shortest_path_canonical().to_flat(), "sp of composition")
            n["compose"] += 1
        elif kind == 2:
            f = random_fst_flat(rng, int(rng.integers(1, 3000)), int(rng.integers(1, 7)), 5, p_eps_i=0.1, p_final=rng.random() * 0.3,
                                sort="ilabel", acyclic=bool(rng.integers(0, 2)), weight_grid=512 if rng.random() < 0.7 else 7)
            d, o = to_device(f), to_oracle(O, f)
            ref = o.shortest_path_canonical()
            assert_flat_identical(d.shortest_path().to_flat(), ref.to_flat(), "shortest path")
            np.testing.assert_array_equal(d.shortest_distance().view(np.uint32), np.asarray(ref.distance, np.float32).view(np.uint32))
            n["sp"] += 1
        elif kind == 3:
            f = random_fst_flat(rng, int(rng.integers(2, 80)), 4, 5, p_eps_i=0.1, p_final=0.25, min_fanout=1, acyclic=bool(rng.integers(0, 2)))
            k = int(rng.integers(2, 8))
            os.environ["WFST_NBEST_LAZY"] = str(int(rng.integers(0, 2)))
            assert_flat_identical(to_device(f).shortest_path(ShortestPathConfig(nshortest=k)).to_flat(),
                                  to_oracle(O, f).shortest_path_n(k).to_flat(), f"n-best {k}")
            f2 = random_fst_flat(rng, int(rng.integers(1, 200)), int(rng.integers(1, 40)), 9, p_eps_i=0.2, p_eps_o=0.2, sort="none")
            d, o = to_device(f2), to_oracle(O, f2)
            by_o = bool(rng.integers(0, 2))
            d.tr_sort(not by_o); o.tr_sort(by_olabel=by_o)
            assert_flat_identical(d.to_flat(), o.to_flat(), "tr_sort")
            assert_flat_identical(d.reverse().to_flat(), o.reverse().to_flat(), "reverse")
            n["nbest"] += 1; n["sort"] += 1
        else:
            t = random_fst_flat(rng, int(rng.integers(5, 400)), int(rng.integers(1, 6)), 4, p_eps_i=rng.random() * 0.3, p_final=0.3, sort="ilabel", min_fanout=1)
            accs = [random_fst_flat(rng, int(rng.integers(1, 25)), 2, 4, p_eps_o=rng.random() * 0.3, p_final=0.4, sort="olabel", acyclic=bool(rng.integers(0, 2)))
                    for _ in range(int(rng.integers(1, 9)))]
            # ... and linear acceptors (the string o T kernel when t has no input epsilons), small alphabet => ties
            from rustfst_amd import synth as _synth
            accs += [_synth.linear_acceptor_flat(rng.integers(1, 3 if seed % 3 else 5, int(rng.integers(0, 30))).astype(np.uint32),
                                                 final_weight=float(rng.integers(0, 3)))
                     for _ in range(int(rng.integers(16, 40)) if seed % 4 == 0 else int(rng.integers(0, 5)))]  # 16+: packed workgroups
            os.environ["WFST_STRING_KERNEL"] = str(int(rng.integers(0, 2)))
            if rng.integers(0, 2):
                os.environ["WFST_BATCH_COPY"] = "1"  # results by copy commands instead of kernel writes to pinned memory
            else:
                os.environ.pop("WFST_BATCH_COPY", None)
            flt = FILTERS[int(rng.integers(0, len(FILTERS)))]
            outs, _ = rustfst_amd.compose_shortest_path_batch([to_device(x) for x in accs], to_device(t), ComposeConfig(flt))
            ot = to_oracle(O, t)
            for x, out in zip(accs, outs):
                want = to_oracle(O, x).compose(ot, compose_filter=flt.value).shortest_path_canonical().to_flat()
                assert_flat_identical(out.to_flat(), want, f"fused batch {flt.name}")
            n["batch"] += 1
    except Exception:
        print("FAILED at seed", seed, "kind", kind, flush=True)
        raise
    seed += 1
print("soak OK:", n, "seeds", seed0, "..", seed - 1)
