"""Randomised differential soak of look-ahead composition: both GPU drivers vs the CPU oracle on many seeded pairs
(epsilon-rich, cyclic and acyclic, small label sets so that pushed labels / weights and multi-epsilon loops are common;
weights on the 1/512 grid and, in a third of the cases, on a 1/7 grid where pushed weights quantise).
Not part of the test suite: python tools/soak_lookahead.py [seconds] [seed0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustfst_amd
from oracle import oracle_py as O
from helpers import assert_flat_identical, random_fst_flat, to_device, to_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
n = dict(wave=0, wide=0, big=0, skipped=0)
while time.time() < t_end:
    rng = np.random.default_rng(50_000 + seed)
    big = seed % 25 == 0
    n1 = int(rng.integers(100, 400)) if big else int(rng.integers(1, 70))
    n2 = int(rng.integers(5, 25)) if big else int(rng.integers(1, 50))
    sig = int(rng.integers(3, 9)) if big else int(rng.integers(1, 7))
    grid = 512 if seed % 3 else 7
    a = random_fst_flat(rng, n1, int(rng.integers(1, 6)), sig, p_eps_i=rng.random() * 0.4, p_eps_o=rng.random() * 0.6,
                        p_final=rng.random() * 0.5, sort="olabel", acyclic=bool(rng.integers(0, 3) == 0), weight_grid=grid)
    b = random_fst_flat(rng, n2, sig + 2 if big else int(rng.integers(1, 7)), sig, p_eps_i=rng.random() * 0.5, p_eps_o=rng.random() * 0.4,
                        p_final=rng.random() * 0.5, sort="ilabel", acyclic=bool(rng.integers(0, 3) == 0), weight_grid=grid)
    try:
        ref = to_oracle(O, a).compose_lookahead(to_oracle(O, b)).to_flat()
    except O.OracleError:
        n["skipped"] += 1
        seed += 1
        continue
    # the wide driver's knobs, drawn per case: lanes per composed state, levels queued per look, a first arena that has to
    # grow (ahead of the level that would not fit, or after it has overflowed)
    knobs = {}
    if rng.integers(0, 3):
        knobs["WFST_WIDE_GROUP"] = str(int(rng.choice([8, 16, 64])))
    if rng.integers(0, 3):
        knobs["WFST_WIDE_BATCH"] = str(int(rng.integers(1, 9)))
    if rng.integers(0, 2):
        knobs["WFST_WIDE_EST_STATES"] = str(int(rng.choice([64, 256, 1024])))
    if rng.integers(0, 2):
        knobs["WFST_WIDE_NO_FORESIGHT"] = "1"
    for k in ("WFST_WIDE_GROUP", "WFST_WIDE_BATCH", "WFST_WIDE_EST_STATES", "WFST_WIDE_NO_FORESIGHT"):
        os.environ.pop(k, None)
    os.environ.update(knobs)
    for path in ("wave", "wide"):
        if path == "wave" and ref["n_states"] > 20000:
            continue
        os.environ["WFST_LOOKAHEAD_PATH"] = path
        la = rustfst_amd.LookAhead(to_device(a))
        out = la.compose(la.relabel(to_device(b))).to_flat()
        assert_flat_identical(out, ref, f"seed {seed} path {path} knobs {knobs}")
        n[path] += 1
    # the same pair through compose() pinned to the wide driver, a random filter, with and without connect
    flt = int(rng.integers(0, 7))
    connect = bool(rng.integers(0, 2))
    try:
        refc = to_oracle(O, a).compose(to_oracle(O, b), connect=connect, compose_filter=flt).to_flat()
    except O.OracleError:
        refc = None
    if refc is not None:
        os.environ["WFST_COMPOSE_PATH"] = "wide"
        got = to_device(a).compose(to_device(b), rustfst_amd.ComposeConfig(rustfst_amd.ComposeFilter(flt), connect=connect)).to_flat()
        os.environ.pop("WFST_COMPOSE_PATH")
        assert_flat_identical(got, refc, f"seed {seed} plain wide filter {flt} connect {connect} knobs {knobs}")
        n["plain_wide"] = n.get("plain_wide", 0) + 1
    if ref["n_states"] > 2048:
        n["big"] += 1
    seed += 1
print("soak_lookahead ok:", n, "next seed", seed)
