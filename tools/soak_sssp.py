"""Randomised differential soak of the relaxation kernels (atomic sweeps with and without binned levels, mailbox sweeps with their NARROW launches)
against the canonical CPU oracle: python tools/soak_sssp.py [seconds] [seed0].  Every case draws a kernel, a band width
and, for the mailbox kernel, a hand-over threshold and a gating mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustfst_amd
from rustfst_amd import synth
from oracle import oracle_py as O
from helpers import assert_flat_identical, random_fst_flat, to_device, to_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
counts = {}
ctx = rustfst_amd.Context(0)
while time.time() < t_end:
    rng = np.random.default_rng(77_000 + seed)
    mode = ["0", "1", "1"][int(rng.integers(0, 3))]
    os.environ["WFST_SSSP_MAILBOX"] = mode
    os.environ["WFST_SSSP_NARROW"] = str(int(rng.choice([0, 8, 64, 8192, 1_000_000_000])))  # hand-over threshold of the NARROW launches
    os.environ["WFST_SSSP_BIG"] = str(int(rng.integers(0, 2)))  # the many-blocks variant of the kernel
    os.environ["WFST_SSSP_STG"] = str(int(rng.choice([1, 2, 7, 24])))  # staging slots per destination
    # the binned levels behind the atomic sweeps (mode 0 only): per-level choice / every level / off, 8192- or 16384-state bins,
    # a hop limit of the message format low enough that deep states relax through the atomic path of the expand kernel
    os.environ["WFST_SSSP_BINNED"] = str(int(rng.integers(0, 2)))
    os.environ["WFST_SSSP_DENSE_LOW"] = str(int(rng.choice([0, 50, 3000, 4_000_000_000])))
    os.environ["WFST_SSSP_BIN_LOG"] = str(int(rng.choice([13, 14])))
    os.environ["WFST_SSSP_BIN_HOPCAP"] = str(int(rng.choice([2, 5, 1 << 18])))
    for var, choices in (("WFST_SSSP_LPS", ["", "", "2", "3", "5", "8"]), ("WFST_SSSP_UMAX", ["", "2", "4"])):  # lanes per state / states per group of the resident rounds
        v = str(rng.choice(choices))
        if v:
            os.environ[var] = v
        else:
            os.environ.pop(var, None)
    os.environ["WFST_SSSP_RES_RETRY_MS"] = "0"
    os.environ["WFST_SSSP_TRANSPOSE_PLAN"] = str(int(rng.integers(0, 2)))
    hint = rng.choice(["", "0", "1"])
    if hint:
        os.environ["WFST_SSSP_HINT"] = str(hint)
    else:
        os.environ.pop("WFST_SSSP_HINT", None)
    d = rng.choice(["", "0", "0.3", "2.5", "40"])
    if d:
        os.environ["WFST_SSSP_DELTA"] = str(d)
    else:
        os.environ.pop("WFST_SSSP_DELTA", None)
    kind = int(rng.integers(0, 3))
    exact = True  # weights on a dyadic grid: float sums are exact, so the (d, hops) fixed point is unique
    if kind == 0:
        grid = int(rng.choice([512, 7, 1]))
        exact = grid != 7
        f = random_fst_flat(rng, int(rng.integers(1, 6000)), int(rng.integers(1, 9)), 5, p_eps_i=0.1, p_final=rng.random() * 0.3,
                            sort="ilabel", acyclic=bool(rng.integers(0, 2)), weight_grid=grid, max_w=int(rng.choice([3, 12, 5000])))
    elif kind == 1:
        fan = int(rng.integers(1, 12))
        if fan < 3 and d == "0.3":  # a deep chain under a band 30x narrower than its arcs needs more sweeps than the cap allows
            os.environ["WFST_SSSP_DELTA"] = "2.5"
        f = synth.make_transducer(int(rng.integers(2, 40000)), fan, 16, float(rng.random() * 0.2), seed=int(rng.integers(0, 1 << 30)))
    else:  # hub: one state with thousands of arcs, long chains
        n = int(rng.integers(10, 9000))
        if d == "0.3":
            os.environ["WFST_SSSP_DELTA"] = "2.5"
        f = synth.make_transducer(n, 2, 8, 0.0, seed=int(rng.integers(0, 1 << 30)))
    dev, orc = to_device(f, ctx), to_oracle(O, f)
    ref = orc.shortest_path_canonical()
    try:
        for q in range(2):
            got = dev.shortest_path().to_flat()
            dist, hops = dev.shortest_distance(want_hops=True)
            np.testing.assert_array_equal(dist.view(np.uint32), np.asarray(ref.distance, np.float32).view(np.uint32))
            if exact:
                assert_flat_identical(got, ref.to_flat(), "shortest path")
                np.testing.assert_array_equal(hops, ref.hops)
            else:  # inexact sums: hop counts of tied labels depend on the relaxation order; the path must
                   # be a path of the input with exactly the optimal weight
                want = ref.to_flat()
                assert got["n_states"] == 0 or want["n_states"] > 0
                if want["n_states"]:
                    acc = np.float32(0.0)
                    for w in got["arcs"]["weight"][::-1]:
                        acc = np.float32(acc + w)
                    acc = np.float32(acc + got["finals"][0])
                    assert acc.view(np.uint32) == np.float32(ref.total_weight).view(np.uint32), (acc, ref.total_weight)
                    ok, _ = orc.contains_path(O.OracleFst.from_flat(**{k: got[k] for k in ("n_states", "start", "offsets", "arcs", "finals", "props")}))
                    assert ok, "not a path of the input"
    except rustfst_amd.WfstError as e:
        # a FORCED band far narrower than the arcs (delta 0.3, or 2.5 against weights up to 5000 on a few hundred states)
        # needs more sweeps than the 4 n + 64 the driver allows before it suspects a negative cycle; the automatic band is
        # 1.5 x the mean weight and only used from 65 536 states on
        if "did not converge" in str(e) and d in ("0.3", "2.5"):
            counts["skipped"] = counts.get("skipped", 0) + 1
            seed += 1
            continue
        print("FAILED at seed", seed, "mode", mode, "delta", d, flush=True)
        raise
    except Exception:
        print("FAILED at seed", seed, "mode", mode, "delta", d, "narrow", os.environ["WFST_SSSP_NARROW"], "hint", hint, "big", os.environ["WFST_SSSP_BIG"], "stg", os.environ["WFST_SSSP_STG"], "kind", kind, "lps", os.environ.get("WFST_SSSP_LPS"), "umax", os.environ.get("WFST_SSSP_UMAX"), flush=True)
        raise
    counts[mode] = counts.get(mode, 0) + 1
    seed += 1
print("soak_sssp OK:", counts, "seeds", seed0, "..", seed - 1)
