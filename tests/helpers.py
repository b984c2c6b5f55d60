"""Shared test helpers: seeded random FSTs, flat <-> oracle conversion, comparison."""
import numpy as np

from rustfst_amd._lib import TR_DTYPE
from rustfst_amd import synth

P = synth  # property bit names
NOT_I_LABEL_SORTED = 0x0000_0000_2000_0000
NOT_O_LABEL_SORTED = 0x0000_0000_8000_0000


def random_fst_flat(rng, n_states, max_fanout, sigma, p_eps_i=0.0, p_eps_o=0.0, p_final=0.3, sort="ilabel",
                    weight_grid=512, max_w=2560, acyclic=False, min_fanout=0):
    """Small random FST as flat CSR with weights on the 1/weight_grid grid; arcs sorted by `sort`
    (ilabel|olabel|none); property word states sortedness truthfully."""
    offsets = [0]
    rows = []
    for s in range(n_states):
        k = int(rng.integers(min_fanout, max_fanout + 1))
        arcs = []
        for _ in range(k):
            il = 0 if rng.random() < p_eps_i else int(rng.integers(1, sigma + 1))
            ol = 0 if rng.random() < p_eps_o else int(rng.integers(1, sigma + 1))
            w = float(rng.integers(0, max_w)) / weight_grid
            if acyclic:
                if s + 1 >= n_states:
                    continue
                ns = int(rng.integers(s + 1, n_states))
            else:
                ns = int(rng.integers(0, n_states))
            arcs.append((il, ol, w, ns))
        if sort == "ilabel":
            arcs.sort(key=lambda a: a[0])
        elif sort == "olabel":
            arcs.sort(key=lambda a: a[1])
        rows.extend(arcs)
        offsets.append(len(rows))
    arcs = np.array(rows, dtype=TR_DTYPE) if rows else np.zeros(0, dtype=TR_DTYPE)
    finals = np.where(rng.random(n_states) < p_final, rng.integers(0, max_w, n_states) / weight_grid, np.inf).astype(
        np.float32)
    props = 0
    il_sorted = all(np.all(np.diff(arcs["ilabel"][offsets[s]:offsets[s + 1]].astype(np.int64)) >= 0)
                    for s in range(n_states))
    ol_sorted = all(np.all(np.diff(arcs["olabel"][offsets[s]:offsets[s + 1]].astype(np.int64)) >= 0)
                    for s in range(n_states))
    props |= P.I_LABEL_SORTED if il_sorted else NOT_I_LABEL_SORTED
    props |= P.O_LABEL_SORTED if ol_sorted else NOT_O_LABEL_SORTED
    return dict(n_states=n_states, start=0 if n_states else None, offsets=np.array(offsets, dtype=np.uint32), arcs=arcs,
                finals=finals, props=props)


def to_oracle(oracle, flat):
    return oracle.OracleFst.from_flat(flat["n_states"], flat["start"], flat["offsets"], flat["arcs"], flat["finals"],
                                      flat["props"])


def to_device(flat, ctx=None):
    import rustfst_amd
    return rustfst_amd.DeviceFst.from_arrays(flat["n_states"], flat["start"], flat["offsets"], flat["arcs"],
                                             flat["finals"], flat["props"], ctx)


def assert_flat_identical(a, b, what="", check_props=True):
    """Bit-exact comparison of two flat FSTs (indices AND float bit patterns)."""
    assert a["n_states"] == b["n_states"], f"{what}: num_states {a['n_states']} != {b['n_states']}"
    assert a["start"] == b["start"], f"{what}: start {a['start']} != {b['start']}"
    np.testing.assert_array_equal(a["offsets"], b["offsets"], err_msg=f"{what}: offsets")
    for k in ("ilabel", "olabel", "nextstate"):
        np.testing.assert_array_equal(a["arcs"][k], b["arcs"][k], err_msg=f"{what}: arcs.{k}")
    np.testing.assert_array_equal(a["arcs"]["weight"].view(np.uint32), b["arcs"]["weight"].view(np.uint32),
                                  err_msg=f"{what}: arc weights (bit pattern)")
    np.testing.assert_array_equal(a["finals"].view(np.uint32), b["finals"].view(np.uint32),
                                  err_msg=f"{what}: final weights (bit pattern)")
    if check_props:
        assert a["props"] == b["props"], f"{what}: props {a['props']:#x} != {b['props']:#x}"


def enumerate_paths(flat, max_paths=200_000):
    """Every complete path (start -> a final state) of an ACYCLIC flat FST by exhaustive depth-first enumeration:
    [(total weight as float64 of left-folded f32 sums, (ilabels...), (olabels...))].  Independent of the oracle and of
    the library: the brute force the n-best results are checked against."""
    if flat["start"] is None or flat["n_states"] == 0:
        return []
    off, arcs, fin = flat["offsets"], flat["arcs"], flat["finals"]
    out = []

    def rec(s, w, il, ol):
        if len(out) > max_paths:
            raise RuntimeError("too many paths for the brute force")
        if np.isfinite(fin[s]):
            out.append((float(np.float32(w) + np.float32(fin[s])), tuple(il), tuple(ol)))
        for a in arcs[off[s]:off[s + 1]]:
            rec(int(a["nextstate"]), np.float32(np.float32(w) + np.float32(a["weight"])), il + [int(a["ilabel"])], ol + [int(a["olabel"])])

    rec(int(flat["start"]), np.float32(0.0), [], [])
    return out


def check_nbest_against_brute_force(result_flat, input_flat, n, what=""):
    """`result_flat` = shortest_path(nshortest = n) of the ACYCLIC `input_flat`: it must hold exactly min(n, #paths)
    complete paths, each a real path of the input (same label strings once epsilons are removed, same weight), and their
    weights must be the n smallest path weights of the input (as a multiset: ties may pick either path)."""
    def strip(t):
        return tuple(x for x in t if x != 0)
    brute = enumerate_paths(input_flat)
    got = enumerate_paths(result_flat)
    assert len(got) == min(n, len(brute)), f"{what}: {len(got)} paths in the result, {len(brute)} in the input, n = {n}"
    want_w = sorted(w for w, _, _ in brute)[:len(got)]
    np.testing.assert_allclose(sorted(w for w, _, _ in got), want_w, rtol=0, atol=1e-4, err_msg=f"{what}: the n smallest weights")
    have = {}
    for w, il, ol in brute:
        have.setdefault((strip(il), strip(ol)), []).append(w)
    for w, il, ol in got:
        ws = have.get((strip(il), strip(ol)))
        assert ws is not None, f"{what}: the result holds a string the input does not accept: {il} / {ol}"
        assert min(abs(w - x) for x in ws) <= 1e-4, f"{what}: path {il} has weight {w}, the input gives {ws}"
