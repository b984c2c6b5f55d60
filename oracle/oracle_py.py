"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under rustfst_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

EQ_REF_KDELTA = 0
EQ_EXACT = 1

TR_DTYPE = np.dtype([("ilabel", "<u4"), ("olabel", "<u4"), ("weight", "<f4"), ("nextstate", "<u4")])


def build(force=False):
    """Compile liboracle.so with the committed Makefile (g++ only)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("oracle.cpp", "oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, u32, u64, i64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_float
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_last_queue_kind.restype = C.c_char_p
        L.oracle_fst_new.restype = vp
        L.oracle_fst_free.argtypes = [vp]
        L.oracle_fst_add_state.argtypes = [vp]
        L.oracle_fst_add_state.restype = u32
        L.oracle_fst_set_start.argtypes = [vp, u32]
        L.oracle_fst_set_final.argtypes = [vp, u32, f32]
        L.oracle_fst_add_tr.argtypes = [vp, u32, u32, u32, f32, u32]
        L.oracle_fst_tr_sort.argtypes = [vp, C.c_int]
        L.oracle_fst_from_flat.argtypes = [u32, i64, vp, vp, vp, u64]
        L.oracle_fst_from_flat.restype = vp
        L.oracle_fst_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u64), C.POINTER(i64), C.POINTER(u64)]
        L.oracle_fst_to_flat.argtypes = [vp, vp, vp, vp]
        L.oracle_fst_eps_counts.argtypes = [vp, vp, vp]
        L.oracle_fst_load.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_fst_load.restype = vp
        L.oracle_fst_store.argtypes = [vp, vp, C.c_size_t]
        L.oracle_fst_store.restype = C.c_size_t
        L.oracle_fst_store_const.argtypes = [vp, vp, C.c_size_t]
        L.oracle_fst_store_const.restype = C.c_size_t
        L.oracle_compose.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.oracle_compose_filter.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.oracle_connect.argtypes = [vp]
        L.oracle_rm_epsilon.argtypes = [vp]
        L.oracle_fst_project.argtypes = [vp, C.c_int]
        L.oracle_compose_lookahead.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.oracle_last_lookahead_tuples.argtypes = [C.POINTER(u64), C.POINTER(u64)]
        L.oracle_last_lookahead_tuples.restype = None
        L.oracle_interval_set_normalize.argtypes = [vp, C.c_size_t, C.POINTER(u64)]
        L.oracle_interval_set_normalize.restype = i64
        L.oracle_interval_set_member.argtypes = [vp, C.c_size_t, u64]
        L.oracle_label_reachable_new.argtypes = [vp, C.c_int]
        L.oracle_label_reachable_new.restype = vp
        L.oracle_label_reachable_free.argtypes = [vp]
        L.oracle_label_reachable_final_label.argtypes = [vp]
        L.oracle_label_reachable_final_label.restype = u32
        L.oracle_label_reachable_num_labels.argtypes = [vp]
        L.oracle_label_reachable_num_labels.restype = C.c_size_t
        L.oracle_label_reachable_labels.argtypes = [vp, vp, vp]
        L.oracle_label_reachable_num_states.argtypes = [vp]
        L.oracle_label_reachable_num_states.restype = C.c_size_t
        L.oracle_label_reachable_num_intervals.argtypes = [vp, u32]
        L.oracle_label_reachable_num_intervals.restype = C.c_size_t
        L.oracle_label_reachable_intervals.argtypes = [vp, u32, vp]
        L.oracle_shortest_path.argtypes = [vp, C.c_int, C.POINTER(vp), vp, C.POINTER(f32)]
        L.oracle_shortest_path_canonical.argtypes = [vp, C.POINTER(vp), vp, vp, C.POINTER(f32), C.POINTER(u32)]
        L.oracle_shortest_path_n.argtypes = [vp, u64, f32, C.c_int, C.POINTER(vp)]
        L.oracle_shortest_path_n_unique.argtypes = [vp, u64, f32, C.c_int, C.POINTER(vp)]
        L.oracle_determinize_fsa.argtypes = [vp, f32, C.c_int, C.POINTER(vp)]
        L.oracle_shortest_distance.argtypes = [vp, f32, vp, u64]
        L.oracle_shortest_distance.restype = u64
        L.oracle_reverse.argtypes = [vp, C.POINTER(vp)]
        L.oracle_bruteforce_min_weight.argtypes = [vp, u32]
        L.oracle_bruteforce_min_weight.restype = f32
        L.oracle_path_in_fst.argtypes = [vp, vp, C.POINTER(f32)]
        L.oracle_compose_shortest_path_batch.argtypes = [
            C.POINTER(vp), C.c_size_t, vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(u64), C.POINTER(C.c_double)]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _err():
    return OracleError(lib().oracle_last_error().decode())


class OracleFst:
    """Owning wrapper of an oracle VectorFst<TropicalWeight>."""

    def __init__(self, handle=None):
        self._h = handle if handle is not None else lib().oracle_fst_new()

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib().oracle_fst_free(self._h)
            self._h = None

    # -- construction mirroring rustfst's VectorFst
    def add_state(self):
        return lib().oracle_fst_add_state(self._h)

    def set_start(self, s):
        if lib().oracle_fst_set_start(self._h, s):
            raise _err()

    def set_final(self, s, w=0.0):
        if lib().oracle_fst_set_final(self._h, s, w):
            raise _err()

    def add_tr(self, s, il, ol, w, ns):
        if lib().oracle_fst_add_tr(self._h, s, il, ol, w, ns):
            raise _err()

    def tr_sort(self, by_olabel=False):
        lib().oracle_fst_tr_sort(self._h, 1 if by_olabel else 0)

    @classmethod
    def from_flat(cls, n_states, start, offsets, arcs, finals, props):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        arcs = np.ascontiguousarray(arcs, dtype=TR_DTYPE)
        finals = np.ascontiguousarray(finals, dtype=np.float32)
        h = lib().oracle_fst_from_flat(n_states, -1 if start is None else int(start), offsets.ctypes.data,
                                       arcs.ctypes.data, finals.ctypes.data, int(props))
        return cls(h)

    @classmethod
    def load(cls, data: bytes):
        h = lib().oracle_fst_load(data, len(data))
        if not h:
            raise _err()
        return cls(h)

    def store(self, fst_type: str = "vector") -> bytes:
        fn = lib().oracle_fst_store if fst_type == "vector" else lib().oracle_fst_store_const
        n = fn(self._h, None, 0)
        buf = (C.c_uint8 * n)()
        fn(self._h, C.addressof(buf), n)
        return bytes(buf)

    # -- inspection
    def info(self):
        n, a, s, p = C.c_uint32(), C.c_uint64(), C.c_int64(), C.c_uint64()
        lib().oracle_fst_info(self._h, C.byref(n), C.byref(a), C.byref(s), C.byref(p))
        return n.value, a.value, (None if s.value < 0 else s.value), p.value

    @property
    def num_states(self):
        return self.info()[0]

    @property
    def num_arcs(self):
        return self.info()[1]

    @property
    def start(self):
        return self.info()[2]

    @property
    def properties(self):
        return self.info()[3]

    def to_flat(self):
        n, a, start, props = self.info()
        offsets = np.zeros(n + 1, dtype=np.uint32)
        arcs = np.zeros(a, dtype=TR_DTYPE)
        finals = np.zeros(n, dtype=np.float32)
        lib().oracle_fst_to_flat(self._h, offsets.ctypes.data, arcs.ctypes.data, finals.ctypes.data)
        return dict(n_states=n, start=start, offsets=offsets, arcs=arcs, finals=finals, props=props)

    def eps_counts(self):
        n = self.num_states
        ni = np.zeros(n, dtype=np.uint32)
        no = np.zeros(n, dtype=np.uint32)
        lib().oracle_fst_eps_counts(self._h, ni.ctypes.data, no.ctypes.data)
        return ni, no

    # -- algorithms
    def compose(self, other, connect=True, eq_mode=EQ_REF_KDELTA, compose_filter=0):
        """compose_filter: ComposeFilterEnum value (0 Auto, 1 Null, 2 Trivial, 3 Sequence, 4 AltSequence, 5 Match, 6 NoMatch)."""
        out = C.c_void_p()
        if lib().oracle_compose_filter(self._h, other._h, 1 if connect else 0, eq_mode, int(compose_filter), C.byref(out)):
            raise _err()
        return OracleFst(out.value)

    def compose_lookahead(self, other, want_relabeled=False):
        """Look-ahead composition as rustfst-cli/src/cmds/compose.rs:77-181 wires it (no connect).  With
        want_relabeled also returns the relabelled, re-sorted copies of both inputs."""
        out, r1, r2 = C.c_void_p(), C.c_void_p(), C.c_void_p()
        if lib().oracle_compose_lookahead(self._h, other._h, C.byref(out), C.byref(r1) if want_relabeled else None,
                                          C.byref(r2) if want_relabeled else None):
            raise _err()
        if want_relabeled:
            return OracleFst(out.value), OracleFst(r1.value), OracleFst(r2.value)
        return OracleFst(out.value)

    @staticmethod
    def last_lookahead_tuples():
        """(tuples created, tuples with a twin one KDELTA weight step away) of this thread's last compose_lookahead."""
        a, b = C.c_uint64(), C.c_uint64()
        lib().oracle_last_lookahead_tuples(C.byref(a), C.byref(b))
        return a.value, b.value

    def label_reachable(self, reach_input=False):
        """LabelReachable::compute_data (label_reachable.rs:135-273): dict(final_label, label2index {label: index},
        intervals [per state list of (begin, end)])."""
        h = lib().oracle_label_reachable_new(self._h, 1 if reach_input else 0)
        if not h:
            raise _err()
        try:
            L = lib()
            n = L.oracle_label_reachable_num_labels(h)
            labels = np.zeros(n, dtype=np.uint32)
            idx = np.zeros(n, dtype=np.uint32)
            L.oracle_label_reachable_labels(h, labels.ctypes.data, idx.ctypes.data)
            ivs = []
            for s in range(L.oracle_label_reachable_num_states(h)):
                k = L.oracle_label_reachable_num_intervals(h, s)
                a = np.zeros(2 * k, dtype=np.uint64)
                if k:
                    L.oracle_label_reachable_intervals(h, s, a.ctypes.data)
                ivs.append([(int(a[2 * i]), int(a[2 * i + 1])) for i in range(k)])
            return {"final_label": int(L.oracle_label_reachable_final_label(h)),
                    "label2index": {int(l): int(i) for l, i in zip(labels, idx)}, "intervals": ivs}
        finally:
            lib().oracle_label_reachable_free(h)

    def connect(self):
        lib().oracle_connect(self._h)

    def rm_epsilon(self):
        if lib().oracle_rm_epsilon(self._h):
            raise _err()
        return self

    def project(self, project_output=False):
        lib().oracle_fst_project(self._h, 1 if project_output else 0)
        return self

    def shortest_path(self, eq_mode=EQ_REF_KDELTA, want_distance=False):
        out = C.c_void_p()
        tot = C.c_float()
        dist = np.zeros(self.num_states, dtype=np.float32) if want_distance else None
        if lib().oracle_shortest_path(self._h, eq_mode, C.byref(out), dist.ctypes.data if want_distance else None,
                                      C.byref(tot)):
            raise _err()
        res = OracleFst(out.value)
        res.total_weight = tot.value
        res.queue_kind = lib().oracle_last_queue_kind().decode()
        if want_distance:
            res.distance = dist
        return res

    def shortest_path_n(self, nshortest, delta=1e-6, eq_mode=EQ_REF_KDELTA, unique=False):
        out = C.c_void_p()
        fn = lib().oracle_shortest_path_n_unique if unique else lib().oracle_shortest_path_n
        if fn(self._h, nshortest, delta, eq_mode, C.byref(out)):
            raise _err()
        return OracleFst(out.value)

    def determinize_fsa(self, delta=1.0 / 1024.0, eq_mode=EQ_REF_KDELTA):
        """determinize() of an acceptor (DeterminizeConfig::default(): delta = KDELTA)."""
        out = C.c_void_p()
        if lib().oracle_determinize_fsa(self._h, delta, eq_mode, C.byref(out)):
            raise _err()
        return OracleFst(out.value)

    def shortest_distance(self, delta=1e-6):
        n = self.num_states
        dist = np.zeros(n, dtype=np.float32)
        lib().oracle_shortest_distance(self._h, delta, dist.ctypes.data, n)
        return dist

    def reverse(self):
        out = C.c_void_p()
        lib().oracle_reverse(self._h, C.byref(out))
        return OracleFst(out.value)

    def shortest_path_canonical(self):
        out = C.c_void_p()
        tot = C.c_float()
        ties = C.c_uint32()
        n = self.num_states
        dist = np.zeros(n, dtype=np.float32)
        hops = np.zeros(n, dtype=np.uint32)
        if lib().oracle_shortest_path_canonical(self._h, C.byref(out), dist.ctypes.data, hops.ctypes.data,
                                                C.byref(tot), C.byref(ties)):
            raise _err()
        res = OracleFst(out.value)
        res.total_weight = tot.value
        res.distance = dist
        res.hops = hops
        res.n_tied_choices = ties.value
        return res

    def bruteforce_min_weight(self, max_len):
        return lib().oracle_bruteforce_min_weight(self._h, max_len)

    def contains_path(self, path):
        w = C.c_float()
        ok = lib().oracle_path_in_fst(path._h, self._h, C.byref(w))
        return bool(ok), w.value

    def __eq__(self, other):
        """VectorFst PartialEq (data_structure.rs:36-41): states + start, weights within KDELTA."""
        a, b = self.to_flat(), other.to_flat()
        return flat_equal(a, b)


def flat_equal(a, b, delta=1.0 / 1024.0):
    if a["n_states"] != b["n_states"] or a["start"] != b["start"]:
        return False
    if not np.array_equal(a["offsets"], b["offsets"]):
        return False
    for k in ("ilabel", "olabel", "nextstate"):
        if not np.array_equal(a["arcs"][k], b["arcs"][k]):
            return False

    def approx(x, y):
        x = x.astype(np.float64)
        y = y.astype(np.float64)
        both_inf = np.isinf(x) & np.isinf(y)
        with np.errstate(invalid="ignore"):
            close = (x <= y + delta) & (y <= x + delta)
        return bool(np.all(both_inf | close))

    return approx(a["arcs"]["weight"], b["arcs"]["weight"]) and approx(a["finals"], b["finals"])


def interval_set_normalize(pairs):
    """IntervalSet::normalize (interval_set.rs:156-190): returns (normalized [(begin, end)], count)."""
    a = np.array([x for p in pairs for x in p], dtype=np.uint64)
    cnt = C.c_uint64()
    n = lib().oracle_interval_set_normalize(a.ctypes.data if len(a) else None, len(pairs), C.byref(cnt))
    if n < 0:
        raise OracleError("empty interval")
    return [(int(a[2 * i]), int(a[2 * i + 1])) for i in range(n)], cnt.value


def interval_set_member(pairs, value):
    a = np.array([x for p in pairs for x in p], dtype=np.uint64)
    return bool(lib().oracle_interval_set_member(a.ctypes.data if len(a) else None, len(pairs), int(value)))


def compose_shortest_path_batch(accs, t, n_threads=1, eq_mode=EQ_REF_KDELTA, keep_outputs=True):
    n = len(accs)
    arr = (C.c_void_p * n)(*[a._h for a in accs])
    outs = (C.c_void_p * n)()
    na = C.c_uint64()
    sec = C.c_double()
    rc = lib().oracle_compose_shortest_path_batch(arr, n, t._h, n_threads, eq_mode,
                                                  outs if keep_outputs else None, C.byref(na), C.byref(sec))
    if rc:
        raise _err()
    res = [OracleFst(outs[i]) for i in range(n)] if keep_outputs else None
    return res, na.value, sec.value
