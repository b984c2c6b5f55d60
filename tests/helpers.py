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
