"""CPU tests of the oracle (oracle/oracle.cpp) against everything the reference holds in-repo for this
path: K1/K2/K3 known-answer tests, K4 loader fixtures, the hand-derived App. B vectors, plus
invariants (brute force, path membership, REF == canonical when the optimum is unique)."""
import json
import os

import numpy as np
import pytest

from helpers import random_fst_flat, to_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def build(oracle, spec):
    f = oracle.OracleFst()
    for _ in range(spec["n_states"]):
        f.add_state()
    if spec.get("start") is not None:
        f.set_start(spec["start"])
    for s, il, ol, w, ns in spec["arcs"]:
        f.add_tr(s, il, ol, w, ns)
    for s, w in spec["finals"]:
        f.set_final(s, w)
    return f


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def flat_matches_spec(flat, spec):
    assert flat["n_states"] == spec["n_states"]
    assert flat["start"] == spec.get("start")
    got = []
    for s in range(flat["n_states"]):
        for a in flat["arcs"][flat["offsets"][s]:flat["offsets"][s + 1]]:
            got.append([s, int(a["ilabel"]), int(a["olabel"]), float(a["weight"]), int(a["nextstate"])])
    exp = spec["arcs"]
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g[:3] == e[:3] and g[4] == e[4], (g, e)
        assert abs(g[3] - e[3]) <= 1.0 / 1024.0, (g, e)
    fin = {s: w for s, w in spec["finals"]}
    for s in range(flat["n_states"]):
        if s in fin:
            assert abs(float(flat["finals"][s]) - fin[s]) <= 1.0 / 1024.0
        else:
            assert np.isinf(flat["finals"][s])


# ---------------------------------------------------------------- K1: test_compose.py:13-81
def test_k1_compose_known_answer(oracle):
    g = load_golden("k1_compose.json")
    res = build(oracle, g["fst1"]).compose(build(oracle, g["fst2"]))
    flat_matches_spec(res.to_flat(), g["expected"])


def test_k1_compose_config_trivial_filter_known_answer(oracle):
    """The reference's own vector for a NON-default filter: test_compose.py:84-154 composes the K1 pair with
    ComposeConfig(ComposeFilter.TRIVIALFILTER, connect=True) and expects the same 4-state machine."""
    g = load_golden("k1_compose.json")
    res = build(oracle, g["fst1"]).compose(build(oracle, g["fst2"]), connect=True, compose_filter=2)
    flat_matches_spec(res.to_flat(), g["expected"])


# ---------------------------------------------------------------- K2: test_shortest_path.py:5-51
def test_k2_shortest_path_known_answer(oracle):
    g = load_golden("k2_shortest_path.json")
    f = build(oracle, g["fst"])
    sp = f.shortest_path()
    flat_matches_spec(sp.to_flat(), g["expected"])
    assert sp.queue_kind == "scc"  # self-loop at s1 -> SccQueue with a FIFO (auto_queue.rs:81-95)
    spc = f.shortest_path_canonical()
    flat_matches_spec(spc.to_flat(), g["expected"])
    assert spc.n_tied_choices == 0


# ---------------------------------------------------------------- K13: hand-traced n_shortest_path (shortest_path.rs:409-518)
def test_k13_nshortest_known_answers(oracle):
    """n = 2 and n = 3 on the K2 graph, traced by hand through the reference's source (tests/golden/K13_DERIVATION.md):
    the second-best path TIES (0-2-3 and 0-1-1-3 both weigh 11) and the reference's heap returns the loop path first —
    state numbering, arc order, weights and which tied path comes first are all pinned."""
    g = load_golden("k13_nshortest_k2.json")
    f = build(oracle, g["fst"])
    flat_matches_spec(f.shortest_path_n(2).to_flat(), g["n2"])
    flat_matches_spec(f.shortest_path_n(3).to_flat(), g["n3"])


def test_k13_lookahead_known_answers(oracle):
    """Look-ahead composition (rustfst-cli compose --compose-type lookahead) on six pairs traced by hand through the
    reference's source (tests/golden/K13_DERIVATION.md section 2): label reachability and relabelling, pruning of a dead end,
    weight pushing (and its division at the following arcs), label pushing through a multi-epsilon, an unseen label, an
    epsilon CYCLE on fst1's output side (condensation: state_reachable.rs:36-67) and the linear-scan branch of
    LabelReachable::reach (label_reachable.rs:325-342)."""
    g = load_golden("k13_lookahead.json")
    for case in g["cases"]:
        out, r1, r2 = build(oracle, case["fst1"]).compose_lookahead(build(oracle, case["fst2"]), want_relabeled=True)
        flat_matches_spec(out.to_flat(), case["expected"])
        if "relabeled_fst1" in case:
            flat_matches_spec(r1.to_flat(), case["relabeled_fst1"])
        if "relabeled_fst2" in case:
            flat_matches_spec(r2.to_flat(), case["relabeled_fst2"])


@pytest.mark.parametrize("seed", range(12))
def test_nshortest_against_independent_brute_force(oracle, seed):
    """The oracle's n-best results against an exhaustive enumeration written independently of it (tests/helpers.py): the
    right number of paths, every one a path of the input, their weights the n smallest."""
    from helpers import check_nbest_against_brute_force
    rng = np.random.default_rng(9100 + seed)
    flat = random_fst_flat(rng, int(rng.integers(3, 13)), 3, 4, p_eps_i=0.15, p_final=0.35, min_fanout=1, acyclic=True,
                           weight_grid=4 if seed % 2 else 512, max_w=12 if seed % 2 else 2560)
    for n in (1, 2, 4, 9):
        check_nbest_against_brute_force(to_oracle(oracle, flat).shortest_path_n(n).to_flat(), flat, n, f"seed {seed} n={n}")


# ---------------------------------------------------------------- K12: determinize_static.rs:210-270
def test_k12_determinize_known_answers(oracle):
    """The reference's two known answers for `determinize` of a tropical acceptor (DeterminizeFsa, default common divisor,
    delta = KDELTA): the determinization that the `unique` branch of shortest_path rests on (shortest_path.rs:157-165)."""
    g = load_golden("k12_determinize.json")
    for case in g["cases"]:
        flat_matches_spec(build(oracle, case["fst"]).determinize_fsa().to_flat(), case["expected"])


# ---------------------------------------------------------------- K3: doctest compose_static.rs:282-289
def test_k3_linear_compose(oracle):
    a = oracle.OracleFst()
    for _ in range(3):
        a.add_state()
    a.set_start(0)
    a.add_tr(0, 1, 2, 0.0, 1)
    a.add_tr(1, 2, 3, 0.0, 2)
    a.set_final(2, 0.0)
    b = oracle.OracleFst()
    for _ in range(3):
        b.add_state()
    b.set_start(0)
    b.add_tr(0, 2, 3, 0.0, 1)
    b.add_tr(1, 3, 4, 0.0, 2)
    b.set_final(2, 0.0)
    r = a.compose(b).to_flat()
    assert r["n_states"] == 3 and r["start"] == 0
    assert [(int(x["ilabel"]), int(x["olabel"]), int(x["nextstate"])) for x in r["arcs"]] == [(1, 3, 1), (2, 4, 2)]


# ---------------------------------------------------------------- App. B.3 (hand-derived from the reference sources)
def test_b3_fst_003(oracle):
    g = load_golden("b3_fst_003_004.json")
    c = build(oracle, g["fst_003"]).compose(build(oracle, g["fst_003_compose"]))
    flat_matches_spec(c.to_flat(), g["fst_003_expected_compose"])
    sp = c.shortest_path()
    flat_matches_spec(sp.to_flat(), g["fst_003_expected_shortest_path"])
    assert abs(sp.total_weight - 7.0) < 1e-5


def test_b3_fst_004_one_side_sorted(oracle):
    g = load_golden("b3_fst_003_004.json")
    f2 = build(oracle, g["fst_004_compose"])
    assert f2.properties & 0x2000_0000  # NOT_I_LABEL_SORTED: ilabels [25,26,5] in insertion order
    c = build(oracle, g["fst_004"]).compose(f2)
    flat_matches_spec(c.to_flat(), g["fst_004_expected_compose"])
    sp = c.shortest_path()
    flat_matches_spec(sp.to_flat(), g["fst_004_expected_shortest_path"])
    assert abs(sp.total_weight - 3.9) < 1e-5


def test_b3_literal_003_o_004_is_empty(oracle):
    g = load_golden("b3_fst_003_004.json")
    c = build(oracle, g["fst_003"]).compose(build(oracle, g["fst_004"]))
    assert c.num_states == 0 and c.start is None
    sp = c.shortest_path()
    assert sp.num_states == 0 and sp.start is None


def test_compose_unsorted_is_an_error(oracle):
    a = oracle.OracleFst()
    a.add_state()
    a.add_state()
    a.set_start(0)
    a.add_tr(0, 1, 5, 0.0, 1)
    a.add_tr(0, 1, 3, 0.0, 1)  # olabels [5,3]: NOT_O_LABEL_SORTED
    b = oracle.OracleFst()
    b.add_state()
    b.add_state()
    b.set_start(0)
    b.add_tr(0, 7, 1, 0.0, 1)
    b.add_tr(0, 2, 1, 0.0, 1)  # ilabels [7,2]: NOT_I_LABEL_SORTED
    with pytest.raises(oracle.OracleError, match="sort"):
        a.compose(b)


# ---------------------------------------------------------------- K4: binary loader fixtures
@pytest.mark.parametrize("name,n_states,n_arcs", [("sigma_matcher_2_left.fst", 8, 10), ("sigma_matcher_2_right.fst", 12, 14)])
def test_k4_loader_fixtures(oracle, name, n_states, n_arcs):
    data = open(os.path.join(GOLDEN, name), "rb").read()
    f = oracle.OracleFst.load(data)
    assert (f.num_states, f.num_arcs) == (n_states, n_arcs)
    again = oracle.OracleFst.load(f.store())
    assert again == f and again.properties == f.properties


# const-format files of the reference's test data (rustfst-tests-data/fst_012, fst_014: ConstFst::Read;
# parse_const_fst, const_fst/serializable_fst.rs:176-237).  Independent check of the aligned v1 layout: the
# per-state niepsilons/noepsilons counters STORED in the file must equal the counters recomputed from the arcs.
@pytest.mark.parametrize("name,n_states,n_arcs", [("fst_012_hcl.fst", 215, 942), ("fst_014_hcl.fst", 85, 310)])
def test_const_loader_fixtures(oracle, name, n_states, n_arcs):
    data = open(os.path.join(GOLDEN, name), "rb").read()
    assert data[8:13] == b"const"
    f = oracle.OracleFst.load(data)
    assert (f.num_states, f.num_arcs, f.start) == (n_states, n_arcs, 0)
    flat = f.to_flat()
    ni, no = f.eps_counts()
    for s in range(n_states):
        a = flat["arcs"][flat["offsets"][s]:flat["offsets"][s + 1]]
        assert (ni[s], no[s]) == (int((a["ilabel"] == 0).sum()), int((a["olabel"] == 0).sum()))
    assert int(flat["arcs"]["nextstate"].max()) < n_states
    again = oracle.OracleFst.load(f.store())  # written back as a vector file
    assert again == f
    # ... and as a const file (ConstFst::store writes version 2, unaligned): same states / arcs / counters, and the
    # unaligned body is exactly the aligned one without its padding
    c2 = f.store("const")
    assert c2[8:13] == b"const" and int.from_bytes(c2[25:29], "little") == 2
    back = oracle.OracleFst.load(c2)
    assert back == f and back.eps_counts()[0].tolist() == ni.tolist() and back.eps_counts()[1].tolist() == no.tolist()
    body = 20 * n_states + 16 * n_arcs
    assert c2[-16 * n_arcs:] == data[-16 * n_arcs:] and len(c2) + 32 >= len(data) >= len(c2)
    assert c2[-body:-16 * n_arcs] in data  # the 20-byte state records, in one piece


@pytest.mark.parametrize("hcl,g", [("fst_014_hcl.fst", "fst_014_g.fst"), ("fst_012_hcl.fst", "fst_012_gp.fst")])
def test_hcl_compose_g_oracle(oracle, hcl, g):
    """The pairing the reference's own data generator sets up (fst_014.h / fst_012.h): HCL o G after arc sorting."""
    a = oracle.OracleFst.load(open(os.path.join(GOLDEN, hcl), "rb").read())
    b = oracle.OracleFst.load(open(os.path.join(GOLDEN, g), "rb").read())
    a.tr_sort(by_olabel=True)
    b.tr_sort(by_olabel=False)
    c = a.compose(b)
    assert c.num_states > 0 and c.num_arcs > 0
    sp = c.shortest_path()
    can = c.shortest_path_canonical()
    assert sp.num_states > 0 and can.num_states > 0
    ok, w = c.contains_path(sp)
    assert ok and w == pytest.approx(can.total_weight, abs=1e-5)


# ---------------------------------------------------------------- invariants on random small FSTs
@pytest.mark.parametrize("seed", range(12))
def test_shortest_path_weight_is_bruteforce_min(oracle, seed):
    rng = np.random.default_rng(seed)
    flat = random_fst_flat(rng, n_states=int(rng.integers(2, 8)), max_fanout=3, sigma=4, p_final=0.4, acyclic=True)
    f = to_oracle(oracle, flat)
    sp = f.shortest_path()
    brute = f.bruteforce_min_weight(10)
    if np.isinf(brute):
        assert sp.num_states == 0
    else:
        assert sp.total_weight == pytest.approx(brute, abs=1e-6)
        ok, w = f.contains_path(sp)
        assert ok and w == pytest.approx(brute, abs=1e-6)


@pytest.mark.parametrize("seed", range(20))
def test_ref_equals_canonical_when_optimum_unique(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    flat = random_fst_flat(rng, n_states=int(rng.integers(5, 60)), max_fanout=4, sigma=6, p_final=0.2, min_fanout=1)
    f = to_oracle(oracle, flat)
    ref = f.shortest_path()
    can = f.shortest_path_canonical()
    exact = f.shortest_path(eq_mode=oracle.EQ_EXACT)
    assert ref.total_weight == can.total_weight == exact.total_weight or \
        (np.isinf(ref.total_weight) and np.isinf(can.total_weight))
    if can.n_tied_choices == 0:
        assert ref == can and exact == can
        assert ref.properties == can.properties


@pytest.mark.parametrize("seed", range(10))
def test_compose_paths_are_paths_of_both(oracle, seed):
    """every successful path of A o T reads a string accepted by A with weight w_A (x) w_T: checked
    through shortest path weights: sp(A o T) == brute-force min over the composition."""
    rng = np.random.default_rng(200 + seed)
    a = random_fst_flat(rng, n_states=5, max_fanout=2, sigma=3, p_eps_o=0.2, p_final=0.5, sort="olabel", acyclic=True)
    t = random_fst_flat(rng, n_states=6, max_fanout=3, sigma=3, p_eps_i=0.2, p_final=0.5, sort="ilabel", acyclic=True)
    fa, ft = to_oracle(oracle, a), to_oracle(oracle, t)
    c_trim = fa.compose(ft, connect=True)
    c_raw = fa.compose(ft, connect=False)
    assert c_trim.num_states <= c_raw.num_states
    # trim must not change the best path weight
    w1 = c_trim.shortest_path().total_weight
    w2 = c_raw.shortest_path().total_weight
    assert (np.isinf(w1) and np.isinf(w2)) or w1 == pytest.approx(w2, abs=1e-6)
    assert w2 == pytest.approx(c_raw.bruteforce_min_weight(14), abs=1e-6) or np.isinf(w2)
    # connect() is idempotent and keeps ids stable
    again = oracle.OracleFst.from_flat(**{k: c_trim.to_flat()[k] for k in ("n_states", "start", "offsets", "arcs", "finals", "props")})
    again.connect()
    assert again == c_trim


def test_queue_disciplines(oracle):
    # linear acceptor: TOP_SORTED bit -> StateOrderQueue (auto_queue.rs:32-33)
    a = oracle.OracleFst()
    for _ in range(3):
        a.add_state()
    a.set_start(0)
    a.add_tr(0, 1, 1, 1.5, 1)
    a.add_tr(1, 2, 2, 2.5, 2)
    a.set_final(2, 0.0)
    assert a.shortest_path().queue_kind == "state_order"
    # composed lattice: props carry neither TOP_SORTED nor ACYCLIC knowledge once T is cyclic -> SCC DFS
    rng = np.random.default_rng(5)
    from rustfst_amd import synth
    t = synth.make_transducer(200, 4, 8, 0.0, seed=11)
    accs = synth.make_acceptors(t, 1, 12, seed0=77)
    c = to_oracle(oracle, accs[0]).compose(to_oracle(oracle, t))
    sp = c.shortest_path()
    assert sp.queue_kind in ("top_order_scc", "scc")
    assert sp.num_states == 13


# ---------------------------------------------------------------- n > 1 (shortest_distance + reverse + n_shortest_path)
def all_path_weights(flat, max_len=12):
    """weights of all successful paths of a small acyclic FST (left fold f32)."""
    out = []
    off, arcs, fin = flat["offsets"], flat["arcs"], flat["finals"]

    def rec(s, acc, depth):
        if np.isfinite(fin[s]):
            out.append(float(np.float32(acc + fin[s])))
        if depth == max_len:
            return
        for a in arcs[off[s]:off[s + 1]]:
            rec(int(a["nextstate"]), np.float32(acc + a["weight"]), depth + 1)

    if flat["start"] is not None:
        rec(flat["start"], np.float32(0.0), 0)
    return sorted(out)


def tree_path_weights(flat):
    """path weights of the tree-shaped n-shortest output."""
    return all_path_weights(flat, max_len=64)


@pytest.mark.parametrize("seed", range(12))
def test_nshortest_weights_are_the_n_smallest(oracle, seed):
    rng = np.random.default_rng(900 + seed)
    flat = random_fst_flat(rng, int(rng.integers(3, 9)), 3, 4, p_final=0.4, acyclic=True, min_fanout=1)
    f = to_oracle(oracle, flat)
    brute = all_path_weights(flat)
    for n in (1, 2, 4, 50):
        out = f.shortest_path_n(n).to_flat()
        got = tree_path_weights(out)
        exp = brute[:n]
        assert len(got) == len(exp), (n, got, exp)
        np.testing.assert_allclose(got, exp, atol=1e-5)


def acceptor_flat(rng, n_states, max_fanout, sigma, **kw):
    """random_fst_flat with olabel = ilabel and the ACCEPTOR property bit (what `unique` asks for)."""
    flat = random_fst_flat(rng, n_states, max_fanout, sigma, sort="none", **kw)
    flat["arcs"]["olabel"] = flat["arcs"]["ilabel"]
    flat["props"] = 0x0000_0000_0001_0000  # ACCEPTOR; sortedness left unknown
    return flat


def all_strings(flat, max_len=64):
    """{label string: least weight} over all successful paths of a small acyclic acceptor (epsilons dropped)."""
    best = {}
    off, arcs, fin = flat["offsets"], flat["arcs"], flat["finals"]

    def rec(s, labs, acc, depth):
        if np.isfinite(fin[s]):
            w = float(np.float32(acc + fin[s]))
            if labs not in best or w < best[labs]:
                best[labs] = w
        if depth == max_len:
            return
        for a in arcs[off[s]:off[s + 1]]:
            rec(int(a["nextstate"]), labs + ((int(a["ilabel"]),) if a["ilabel"] else ()), np.float32(acc + a["weight"]), depth + 1)

    if flat["start"] is not None:
        rec(flat["start"], (), np.float32(0.0), 0)
    return best


@pytest.mark.parametrize("seed", range(16))
def test_nshortest_unique_gives_the_n_best_distinct_strings(oracle, seed):
    """unique = true (shortest_path.rs:157-165: determinize_with_distance of the reversed FST, then the same search).  Unpinned
    by reference output (its goldens need OpenFST), so checked through what the branch is for: every string of the result
    appears once, with the least weight any of its paths has in the input, and the n strings are the n lightest."""
    rng = np.random.default_rng(4200 + seed)
    flat = acceptor_flat(rng, int(rng.integers(3, 10)), 4, 2 + seed % 2, p_final=0.4, acyclic=True, min_fanout=1,
                         weight_grid=4, max_w=12)  # few labels, coarse weights: many paths share a string
    f = to_oracle(oracle, flat)
    truth = all_strings(flat)
    ranked = sorted(truth.values())
    for n in (2, 3, 6, 50):
        out = f.shortest_path_n(n, unique=True).to_flat()
        got = all_strings(out)
        n_paths = len(all_path_weights(out, max_len=64))
        assert n_paths == len(got) == min(n, len(truth)), (n, n_paths, len(got), len(truth))  # one path per string
        for labs, w in got.items():
            assert labs in truth and abs(truth[labs] - w) < 1e-4, (labs, w, truth.get(labs))
        np.testing.assert_allclose(sorted(got.values()), ranked[:len(got)], atol=1e-4)
    # without `unique` the same input returns strings more than once (that is what the flag is for)
    if len(all_path_weights(flat)) > len(truth):
        many = f.shortest_path_n(50).to_flat()
        assert len(all_path_weights(many, max_len=64)) > len(all_strings(many))


@pytest.mark.parametrize("seed", range(20))
def test_determinize_is_deterministic_and_equivalent(oracle, seed):
    """determinize_fsa (pinned on K12) on random acyclic acceptors: no state of the result has two arcs with one label, and
    every string keeps exactly the least weight it had in the input (weights on a coarse grid: the quantization by delta
    is the identity there)."""
    rng = np.random.default_rng(4600 + seed)
    flat = acceptor_flat(rng, int(rng.integers(2, 12)), 4, 2 + seed % 3, p_eps_i=0.15 * (seed % 2), p_final=0.4, acyclic=True,
                         min_fanout=1, weight_grid=4, max_w=16)
    det = to_oracle(oracle, flat).determinize_fsa().to_flat()
    for s_ in range(det["n_states"]):
        labels = [int(a["ilabel"]) for a in det["arcs"][det["offsets"][s_]:det["offsets"][s_ + 1]]]
        assert len(labels) == len(set(labels)) and labels == sorted(labels), (s_, labels)  # one arc per label, label order

    def strings_with_eps(fl):  # (epsilon is a label like any other for determinize: keep it in the string)
        best = {}
        off, arcs, fin = fl["offsets"], fl["arcs"], fl["finals"]

        def rec(st, labs, acc):
            if np.isfinite(fin[st]):
                w = float(np.float32(acc + fin[st]))
                best[labs] = min(best.get(labs, np.inf), w)
            for a in arcs[off[st]:off[st + 1]]:
                rec(int(a["nextstate"]), labs + (int(a["ilabel"]),), np.float32(acc + a["weight"]))

        if fl["start"] is not None:
            rec(fl["start"], (), np.float32(0.0))
        return best

    want, got = strings_with_eps(flat), strings_with_eps(det)
    assert set(want) == set(got)
    for k in want:
        assert abs(want[k] - got[k]) < 1e-3, (k, want[k], got[k])


def test_nshortest_unique_needs_an_acceptor(oracle):
    rng = np.random.default_rng(5)
    flat = random_fst_flat(rng, 6, 3, 3, p_final=0.5, acyclic=True, min_fanout=1)  # a transducer: no ACCEPTOR bit
    with pytest.raises(Exception, match="expected acceptor"):  # determinize_fsa_op.rs:138-140
        to_oracle(oracle, flat).shortest_path_n(3, unique=True)
    assert to_oracle(oracle, flat).shortest_path_n(1, unique=True).num_states >= 0  # nshortest == 1: `unique` is not looked at


def test_shortest_distance_and_reverse(oracle):
    rng = np.random.default_rng(31)
    flat = random_fst_flat(rng, 40, 4, 5, p_final=0.2, min_fanout=1)
    f = to_oracle(oracle, flat)
    d = f.shortest_distance()
    can = f.shortest_path_canonical()
    np.testing.assert_array_equal(d, can.distance)  # grid weights: approx_equal(1e-6) == exact
    r = f.reverse().to_flat()
    assert r["n_states"] == flat["n_states"] + 1 and r["start"] == 0
    n_final = int(np.isfinite(flat["finals"]).sum())
    assert r["offsets"][1] == n_final and len(r["arcs"]) == len(flat["arcs"]) + n_final
    rr = f.reverse().reverse().to_flat()  # reversing twice gives the language back (two extra states)
    assert rr["n_states"] == flat["n_states"] + 2


# ---------------------------------------------------------------- §8(f) N4: the six compose filters of ComposeFilterEnum
# Unpinned by reference output (the reference's filter goldens are produced by OpenFST at CI time); checked through
# what every epsilon filter must preserve: the weighted relation.  All filters except Null (which drops epsilon-only
# moves by design) accept exactly the same (input, output) strings; the sequencing filters merely remove redundant
# epsilon paths, so the best weight agrees with Trivial's (which keeps every path).
@pytest.mark.parametrize("seed", range(10))
def test_compose_filters_preserve_the_best_path(oracle, seed):
    rng = np.random.default_rng(6000 + seed)
    a = random_fst_flat(rng, int(rng.integers(3, 9)), 3, 2, p_eps_o=0.4, p_final=0.4, sort="olabel", min_fanout=1)
    b = random_fst_flat(rng, int(rng.integers(3, 9)), 3, 2, p_eps_i=0.4, p_final=0.4, sort="ilabel", min_fanout=1)
    oa, ob = to_oracle(oracle, a), to_oracle(oracle, b)
    best = {}
    for flt in (0, 2, 3, 4, 5, 6):
        c = oa.compose(ob, compose_filter=flt)
        best[flt] = float(c.shortest_path_canonical().total_weight)
    assert oa.compose(ob, compose_filter=0) == oa.compose(ob, compose_filter=3)  # Auto is the sequence filter
    ref = best[2]
    for flt, w in best.items():
        assert (np.isinf(w) and np.isinf(ref)) or w == pytest.approx(ref, abs=1e-5), (flt, best)
    # Null: a sub-relation (no lone epsilon moves) — never better than the others
    wn = float(oa.compose(ob, compose_filter=1).shortest_path_canonical().total_weight)
    assert wn >= ref - 1e-5


# ------------------------------------------------------------------ look-ahead composition (row A12 / N1)
def test_interval_set_reference_unit_tests(oracle):
    """compose/interval_set.rs:208-275 (the reference's own unit tests): normalize collapses overlapping and adjacent
    intervals and counts points; member() on the normalized set; Ord = begin ascending then end DESCENDING."""
    assert not oracle.interval_set_member([], 3)
    norm, count = oracle.interval_set_normalize([(0, 5), (3, 10)])
    assert norm == [(0, 10)] and count == 10
    assert oracle.interval_set_member(norm, 3)
    norm, count = oracle.interval_set_normalize(norm + [(12, 13)])  # union + normalize
    assert norm == [(0, 10), (12, 13)] and count == 11
    assert oracle.interval_set_normalize([(1, 4), (1, 3)])[0] == [(1, 4)]   # (1,4) < (1,3): the longer one comes first
    assert oracle.interval_set_normalize([(1, 4), (4, 6)])[0] == [(1, 6)]   # adjacent intervals merge
    assert oracle.interval_set_normalize([(3, 4), (2, 3), (7, 9)]) == ([(2, 4), (7, 9)], 4)
    assert not oracle.interval_set_member([(2, 4), (7, 9)], 4) and oracle.interval_set_member([(2, 4), (7, 9)], 8)


def _reachable_olabels(flat, s):
    """labels readable on the output side from s after any number of output-epsilon arcs; 'final' reachable likewise"""
    off, arcs, fin = flat["offsets"], flat["arcs"], flat["finals"]
    seen, stack, labels, final = {s}, [s], set(), False
    while stack:
        q = stack.pop()
        final = final or bool(np.isfinite(fin[q]))
        for k in range(off[q], off[q + 1]):
            if arcs[k]["olabel"] == 0:
                t = int(arcs[k]["nextstate"])
                if t not in seen:
                    seen.add(t)
                    stack.append(t)
            else:
                labels.add(int(arcs[k]["olabel"]))
    return labels, final


@pytest.mark.parametrize("seed", range(25))
def test_label_reachable_against_bruteforce(oracle, seed):
    """LabelReachable::compute_data on output labels (label_reachable.rs:135-273): for every state the interval set
    holds exactly the (relabelled) labels reachable through output-epsilon paths, and the final label iff a final
    state is; cyclic epsilon structure goes through the condensation (state_reachable.rs:37-67)."""
    rng = np.random.default_rng(600 + seed)
    f = random_fst_flat(rng, int(rng.integers(1, 25)), 4, 6, p_eps_i=0.2, p_eps_o=0.5, p_final=0.25, sort="olabel",
                        acyclic=(seed % 3 == 0))
    of = to_oracle(oracle, f)
    try:
        data = of.label_reachable(reach_input=False)
    except oracle.OracleError as e:
        assert "Final state contained in a cycle" in str(e)  # the reference bails on such inputs too
        return
    l2i = data["label2index"]
    assert sorted(v for v in l2i.values()) == list(range(1, len(l2i) + 1))  # DFS indices 1..n
    for s in range(f["n_states"]):
        labels, final = _reachable_olabels(f, s)
        ivs = data["intervals"][s]
        assert ivs == sorted(ivs) and all(b < e for b, e in ivs)
        members = {i for b, e in ivs for i in range(b, e)}
        expect = {l2i[l] for l in labels} | ({data["final_label"]} if final else set())
        assert members == expect, (s, members, expect)


def _successful_paths(flat, max_paths=200000):
    from collections import Counter
    off, arcs, fin = flat["offsets"], flat["arcs"], flat["finals"]
    out = Counter()
    if flat["start"] is None or flat["start"] < 0:
        return out
    stack = [(flat["start"], (), (), 0.0)]
    while stack:
        s, il, ol, w = stack.pop()
        if np.isfinite(fin[s]):
            out[(il, ol, round(float(w + fin[s]) * 1024))] += 1
            assert sum(out.values()) < max_paths
        for k in range(off[s], off[s + 1]):
            a = arcs[k]
            stack.append((int(a["nextstate"]), il + ((int(a["ilabel"]),) if a["ilabel"] else ()),
                          ol + ((int(a["olabel"]),) if a["olabel"] else ()), w + float(a["weight"])))
    return out


@pytest.mark.parametrize("seed", range(60))
def test_lookahead_compose_is_the_same_weighted_relation(oracle, seed):
    """The look-ahead configuration of cmds/compose.rs:77-181 prunes dead ends and pushes labels / weights forward, but
    the result must accept exactly the successful paths of the plain composition: same multiset of (input string, output
    string, weight) on acyclic inputs (weights on the 1/512 grid: sums and the pushed differences are exact)."""
    rng = np.random.default_rng(7000 + seed)
    a = random_fst_flat(rng, int(rng.integers(2, 9)), 3, 3, p_eps_i=0.2, p_eps_o=0.4, p_final=0.4, sort="olabel", acyclic=True)
    b = random_fst_flat(rng, int(rng.integers(2, 9)), 3, 3, p_eps_i=0.3, p_eps_o=0.2, p_final=0.4, sort="ilabel", acyclic=True)
    oa, ob = to_oracle(oracle, a), to_oracle(oracle, b)
    plain = oa.compose(ob, connect=False).to_flat()
    la = oa.compose_lookahead(ob).to_flat()
    assert _successful_paths(plain) == _successful_paths(la)


@pytest.mark.parametrize("seed", range(60))
def test_lookahead_compose_keeps_the_best_path_on_cyclic_inputs(oracle, seed):
    rng = np.random.default_rng(17000 + seed)
    a = random_fst_flat(rng, int(rng.integers(2, 12)), 3, 3, p_eps_i=0.2, p_eps_o=0.45, p_final=0.3, sort="olabel")
    b = random_fst_flat(rng, int(rng.integers(2, 12)), 3, 3, p_eps_i=0.3, p_eps_o=0.2, p_final=0.3, sort="ilabel")
    oa, ob = to_oracle(oracle, a), to_oracle(oracle, b)
    try:
        la, r1, r2 = oa.compose_lookahead(ob, want_relabeled=True)
    except oracle.OracleError as e:
        assert "Final state contained in a cycle" in str(e)
        return
    # the relabelled inputs: same shape, fst1 sorted by olabel, fst2 by ilabel, epsilons stay epsilons
    f1, f2 = r1.to_flat(), r2.to_flat()
    assert f1["n_states"] == a["n_states"] and len(f1["arcs"]) == len(a["arcs"])
    for k in range(f1["n_states"]):
        ol = f1["arcs"]["olabel"][f1["offsets"][k]:f1["offsets"][k + 1]]
        assert np.all(np.diff(ol.astype(np.int64)) >= 0)
    assert np.array_equal(np.sort(f1["arcs"]["olabel"] == 0), np.sort(a["arcs"]["olabel"] == 0))
    assert np.array_equal(np.sort(f2["arcs"]["ilabel"] == 0), np.sort(b["arcs"]["ilabel"] == 0))

    def total(p):
        p = p.to_flat()
        return None if p["n_states"] == 0 else round((float(p["arcs"]["weight"].sum()) + float(p["finals"][0])) * 1024)
    assert total(oa.compose(ob).shortest_path()) == total(la.shortest_path())


def test_k7_project_known_answer(oracle):
    """rustfst-python/tests/algorithms/test_project.py:5-97: input and output projection of the same 3-state FST."""
    g = load_golden("k7_project.json")
    flat_matches_spec(build(oracle, g["fst"]).project(False).to_flat(), g["expected_input"])
    flat_matches_spec(build(oracle, g["fst"]).project(True).to_flat(), g["expected_output"])
    f = build(oracle, g["fst"]).project(False)
    from rustfst_amd import synth
    assert f.to_flat()["props"] & synth.ACCEPTOR  # projection.rs:105-119 (the reference's proptest)


def test_k8_connect_known_answer(oracle):
    """rustfst-python/tests/algorithms/test_connect.py:4-55."""
    g = load_golden("k8_connect.json")
    f = build(oracle, g["fst"])
    f.connect()
    flat_matches_spec(f.to_flat(), g["expected"])


def test_k9_reverse_known_answer(oracle):
    """rustfst-python/tests/algorithms/test_reverse.py:4-57."""
    g = load_golden("k9_reverse.json")
    flat_matches_spec(build(oracle, g["fst"]).reverse().to_flat(), g["expected"])


def test_k10_tr_sort_known_answer(oracle):
    """rustfst-python/tests/algorithms/test_tr_sort.py:4-97 (ilabel and olabel comparators, stable)."""
    g = load_golden("k10_tr_sort.json")
    a = build(oracle, g["fst"])
    a.tr_sort(by_olabel=False)
    flat_matches_spec(a.to_flat(), g["expected_ilabel"])
    b = build(oracle, g["fst"])
    b.tr_sort(by_olabel=True)
    flat_matches_spec(b.to_flat(), g["expected_olabel"])


def test_k11_rm_epsilon_known_answer(oracle):
    """rustfst-python/tests/algorithms/test_rm_epsilon.py:4-54 (default config: connect, no thresholds)."""
    g = load_golden("k11_rm_epsilon.json")
    flat_matches_spec(build(oracle, g["fst"]).rm_epsilon().to_flat(), g["expected"])


@pytest.mark.parametrize("seed", range(40))
def test_rm_epsilon_keeps_the_weighted_relation(oracle, seed):
    """rm_epsilon removes every epsilon:epsilon arc and keeps the weighted relation: no such arc is left, and on acyclic
    inputs the (min,+)-combined weight of every (input string, output string) pair is unchanged; on cyclic inputs the
    best path weight is."""
    rng = np.random.default_rng(12_000 + seed)
    acyclic = seed % 2 == 0
    f = random_fst_flat(rng, int(rng.integers(1, 14)), 3, 3, p_eps_i=0.5, p_eps_o=0.5, p_final=0.35, acyclic=acyclic, sort="none")
    o = to_oracle(oracle, f)
    before = o.to_flat()
    best_before = o.shortest_path().to_flat()
    o.rm_epsilon()
    after = o.to_flat()
    assert not np.any((after["arcs"]["ilabel"] == 0) & (after["arcs"]["olabel"] == 0))

    def total(p):
        return None if p["n_states"] == 0 else round((float(p["arcs"]["weight"].sum()) + float(p["finals"][0])) * 1024)
    assert total(best_before) == total(to_oracle(oracle, after).shortest_path().to_flat())
    if acyclic:
        def best_per_string(flat):
            best = {}
            for (il, ol, w), _ in _successful_paths(flat).items():
                best[(il, ol)] = min(best.get((il, ol), 1 << 60), w)
            return best
        assert best_per_string(before) == best_per_string(after)


def test_lookahead_tuples_have_no_kdelta_neighbours():
    """The only composed-state tuples the reference's approximate PartialEq (semiring.rs:159-168) could merge while this
    engine keeps them apart are twins whose quantized pushed weights are one KDELTA step apart.  On grid and real-valued
    weights of ordinary scale there are none; weights of the size of the quantum do produce them (so the counter works)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import lookahead_tuple_gap as G
    shapes = [(300, 20, 3, 8, 8), (1000, 30, 3, 10, 10)]
    rows = {r["family"]: r for r in G.measure(shapes, [("grid", 512), ("real", 10.0), ("real", 0.01)], [1, 2])}
    assert rows["grid 512"]["tuples"] > 1000
    assert rows["grid 512"]["adjacent"] == 0
    assert rows["real 10.0"]["adjacent"] == 0
    assert rows["real 0.01"]["adjacent"] > 0
