"""Deterministic synthetic workloads: transducer T(N,F,Sigma,p_eps,seed) with a ring
backbone and random-walk acceptors.  PRNG = SplitMix64 in counter mode (vectorised with numpy).
Weights lie on the 1/512 grid so that the reference's approximate TropicalWeight == (KDELTA = 1/1024,
rustfst/src/semirings/semiring.rs:159-168) coincides with exact equality.
"""
from __future__ import annotations

import numpy as np

from ._lib import TR_DTYPE

# FstProperties bits used here (rustfst/src/fst_properties/properties.rs:22-103)
ACCEPTOR = 0x0000_0000_0001_0000
NOT_ACCEPTOR = 0x0000_0000_0002_0000
I_DETERMINISTIC = 0x0000_0000_0004_0000
O_DETERMINISTIC = 0x0000_0000_0010_0000
NO_EPSILONS = 0x0000_0000_0080_0000
I_EPSILONS = 0x0000_0000_0100_0000
NO_I_EPSILONS = 0x0000_0000_0200_0000
NO_O_EPSILONS = 0x0000_0000_0800_0000
I_LABEL_SORTED = 0x0000_0000_1000_0000
O_LABEL_SORTED = 0x0000_0000_4000_0000
WEIGHTED = 0x0000_0001_0000_0000
UNWEIGHTED = 0x0000_0002_0000_0000
CYCLIC = 0x0000_0004_0000_0000
ACYCLIC = 0x0000_0008_0000_0000
INITIAL_CYCLIC = 0x0000_0010_0000_0000
INITIAL_ACYCLIC = 0x0000_0020_0000_0000
TOP_SORTED = 0x0000_0040_0000_0000
NOT_TOP_SORTED = 0x0000_0080_0000_0000
ACCESSIBLE = 0x0000_0100_0000_0000
COACCESSIBLE = 0x0000_0400_0000_0000
UNWEIGHTED_CYCLES = 0x0000_8000_0000_0000

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n outputs of SplitMix64 started at `seed` (stream k jumps the seed by a fixed odd constant)."""
    with np.errstate(over="ignore"):
        s0 = np.uint64((seed + stream * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        z = s0 + (np.arange(1, n + 1, dtype=np.uint64) * _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def make_transducer(n_states: int, fanout: int = 10, sigma: int = 256, p_eps: float = 0.0, seed: int = 3,
                    p_final: float = 0.01) -> dict:
    """T(N,F,Sigma,p_eps,seed): state s has F arcs, arc 0 -> (s+1) mod N, the others -> uniform states;
    labels U[1,Sigma] (ilabel 0 with probability p_eps); weight k/512, k ~ U{0..5119}; arcs stably sorted
    by ilabel; start 0; finals Bernoulli(p_final) with weights k/512.  Returns the flat CSR dict."""
    N, F = int(n_states), int(fanout)
    E = N * F
    r_next = splitmix64(seed, E, 1)
    r_il = splitmix64(seed, E, 2)
    r_ol = splitmix64(seed, E, 3)
    r_w = splitmix64(seed, E, 4)
    r_eps = splitmix64(seed, E, 5)
    src = np.repeat(np.arange(N, dtype=np.uint64), F)
    slot = np.tile(np.arange(F, dtype=np.uint64), N)
    nxt = (r_next % np.uint64(N)).astype(np.uint32)
    ring = ((src + np.uint64(1)) % np.uint64(N)).astype(np.uint32)
    nxt = np.where(slot == 0, ring, nxt).astype(np.uint32)
    il = (r_il % np.uint64(sigma)).astype(np.uint32) + np.uint32(1)
    ol = (r_ol % np.uint64(sigma)).astype(np.uint32) + np.uint32(1)
    if p_eps > 0:
        eps = (r_eps >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) < p_eps
        eps &= slot != 0  # keep the ring backbone consuming
        il = np.where(eps, np.uint32(0), il).astype(np.uint32)
    w = ((r_w % np.uint64(5120)).astype(np.float32) / np.float32(512.0)).astype(np.float32)
    # stable sort by ilabel inside each state
    order = np.lexsort((slot, il, src))
    arcs = np.empty(E, dtype=TR_DTYPE)
    arcs["ilabel"] = il[order]
    arcs["olabel"] = ol[order]
    arcs["weight"] = w[order]
    arcs["nextstate"] = nxt[order]
    offsets = (np.arange(N + 1, dtype=np.uint64) * np.uint64(F)).astype(np.uint32)
    r_f = splitmix64(seed, N, 6)
    r_fw = splitmix64(seed, N, 7)
    is_final = (r_f >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) < p_final
    finals = np.where(is_final, (r_fw % np.uint64(5120)).astype(np.float32) / np.float32(512.0),
                      np.float32(np.inf)).astype(np.float32)
    props = (I_LABEL_SORTED | NOT_ACCEPTOR | WEIGHTED | CYCLIC | INITIAL_CYCLIC | NOT_TOP_SORTED | ACCESSIBLE |
             NO_O_EPSILONS | NO_EPSILONS | (I_EPSILONS if p_eps > 0 else NO_I_EPSILONS))
    return dict(n_states=N, start=0, offsets=offsets, arcs=arcs, finals=finals, props=props)


def random_walk_labels(t: dict, length: int, seed: int):
    """ilabels along a uniform random walk of `length` non-epsilon steps in T from its start state.
    Returns (labels uint32[length], end_state)."""
    offsets, arcs = t["offsets"], t["arcs"]
    rnd = splitmix64(seed, 4 * length + 16, 9)
    labels = np.empty(length, dtype=np.uint32)
    s = int(t["start"])
    k = 0
    i = 0
    while k < length:
        b, e = int(offsets[s]), int(offsets[s + 1])
        a = arcs[b + int(rnd[i % rnd.shape[0]] % np.uint64(e - b))]
        i += 1
        if int(a["ilabel"]) == 0:
            if i > 64 * (length + 16):
                raise RuntimeError("random walk stuck on epsilon arcs")
            s = int(a["nextstate"])  # follow the epsilon without consuming a label
            continue
        labels[k] = a["ilabel"]
        s = int(a["nextstate"])
        k += 1
    return labels, s


def linear_acceptor_flat(labels, final_weight: float = 0.0) -> dict:
    """Flat CSR of utils::acceptor (rustfst/src/utils/labels_to_fst.rs:111-132)."""
    labels = np.asarray(labels, dtype=np.uint32)
    L = int(labels.shape[0])
    arcs = np.empty(L, dtype=TR_DTYPE)
    arcs["ilabel"] = labels
    arcs["olabel"] = labels
    arcs["weight"] = 0.0
    arcs["nextstate"] = np.arange(1, L + 1, dtype=np.uint32)
    offsets = np.minimum(np.arange(L + 2, dtype=np.uint32), np.uint32(L))
    finals = np.full(L + 1, np.inf, dtype=np.float32)
    finals[L] = final_weight
    return dict(n_states=L + 1, start=0, offsets=offsets, arcs=arcs, finals=finals, props=acceptor_props(L, final_weight))


def acceptor_props(length: int, final_weight: float = 0.0) -> int:
    """Property word rustfst's VectorFst carries after utils::acceptor built it."""
    from .fst import Tr, VectorFst  # replay on the host mirror once per shape
    key = (min(length, 2), float(final_weight))
    if key not in _props_cache:
        f = VectorFst()
        cur = f.add_state()
        f.set_start(cur)
        for _ in range(min(length, 2)):
            nxt = f.add_state()
            f.add_tr(cur, Tr(1, 1, 0.0, nxt))
            cur = nxt
        f.set_final(cur, final_weight)
        _props_cache[key] = f.properties()
    return _props_cache[key]


_props_cache: dict = {}


def make_acceptors(t: dict, n: int, length: int, seed0: int = 1000, mark_finals: bool = True):
    """n random-walk acceptors A_i(L, seed0+i).  With mark_finals the walk end states become final in
    T (weight 0.5) so that every composition has at least one successful path.  Returns list of flats."""
    accs = []
    for i in range(n):
        labels, end = random_walk_labels(t, length, seed0 + i)
        if mark_finals and not np.isfinite(t["finals"][end]):
            t["finals"][end] = np.float32(0.5)
        accs.append(linear_acceptor_flat(labels))
    return accs
