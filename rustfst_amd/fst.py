"""Host-side mirror of the reference's Python surface for the compose -> shortest_path path.

Names, argument meaning and error behaviour follow rustfst-python
(rustfst-python/rustfst/fst/vector_fst.py, rustfst/tr.py, rustfst/algorithms/{compose,shortest_path}.py)
so that the parity tests read like the reference's own tests.  Everything that computes goes
through the C-ABI of libwfst_amd.so (include/wfst.h) onto the GPU; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
from enum import Enum
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import TR_DTYPE, WfstError, check

KSHORTESTDELTA = 1e-6  # rustfst/src/lib.rs:271


class Tr:
    """An arc (rustfst-python/rustfst/tr.py:17-136): ilabel, olabel, weight, next_state."""

    __slots__ = ("ilabel", "olabel", "weight", "next_state")

    def __init__(self, ilabel: int = 0, olabel: int = 0, weight: Optional[float] = None, nextstate: int = 0):
        self.ilabel = int(ilabel)
        self.olabel = int(olabel)
        self.weight = 0.0 if weight is None else float(weight)  # weight_one
        self.next_state = int(nextstate)

    def __eq__(self, other):
        return (self.ilabel, self.olabel, self.next_state) == (other.ilabel, other.olabel, other.next_state) and \
            abs(np.float32(self.weight) - np.float32(other.weight)) <= 1.0 / 1024.0

    def __repr__(self):
        return f"<Tr ilabel={self.ilabel}, olabel={self.olabel}, weight={self.weight}, next_state={self.next_state}>"


# ------------------------------------------------------------------ context
class Context:
    """One engine context per (host thread, GPU): a HIP stream + device memory pools."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, cu_mask: Optional[Sequence[int]] = None):
        """stream: an existing HIP stream handle to run on; cu_mask: list of 32-bit words, bit i of word i//32 =
        compute unit i — the context then owns a stream restricted to those CUs."""
        L = _lib.lib()
        h = C.c_void_p()
        if cu_mask is not None:
            words = np.asarray(list(cu_mask), dtype=np.uint32)
            check(L.wfst_ctx_create_with_cu_mask(device, words.ctypes.data, len(words), C.byref(h)),
                  "wfst_ctx_create_with_cu_mask")
        elif stream is None:
            check(L.wfst_ctx_create(device, C.byref(h)), "wfst_ctx_create")
        else:
            check(L.wfst_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)), "wfst_ctx_create_on_stream")
        self._h = h
        self.device = device

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().wfst_ctx_destroy(h)
            except Exception:
                pass
            self._h = None

    def synchronize(self):
        check(_lib.lib().wfst_ctx_synchronize(self._h), "wfst_ctx_synchronize")

    def set_resident_share(self, share: int):
        """0 (default): a resident relaxation launch may fill the device (fastest alone); 1: at most half of it, so that it runs
        beside a large fused batch instead of behind it (wfst_ctx_set_resident_share)."""
        check(_lib.lib().wfst_ctx_set_resident_share(self._h, int(share)), "wfst_ctx_set_resident_share")

    def set_tie_order(self, reference_order: bool):
        """shortest_path(nshortest = 1): False = the canonical tie rule (default); True = the reference's own choice among
        tied optima on ACYCLIC inputs (wfst_ctx_set_tie_order)."""
        check(_lib.lib().wfst_ctx_set_tie_order(self._h, 1 if reference_order else 0), "wfst_ctx_set_tie_order")

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        check(_lib.lib().wfst_ctx_stream(self._h, C.byref(s)))
        return s.value or 0

    def set_profiling(self, on):
        """False / 0: off; True / 1: events around every relaxation launch (synchronises after each); 2: the sweeps of a
        repeated shortest_path query timed as one chain between two events (stats: relax_ms, relax_launches)."""
        check(_lib.lib().wfst_ctx_set_profiling(self._h, int(on)))

    def reset_stats(self):
        check(_lib.lib().wfst_ctx_reset_stats(self._h))

    def sweep_trace(self):
        """(ms, arcs, states) arrays, one entry per relaxation launch of the last profiled solve."""
        n = C.c_size_t()
        check(_lib.lib().wfst_ctx_get_sweep_trace(self._h, None, None, None, 0, C.byref(n)))
        ms = np.zeros(n.value, dtype=np.float64)
        arcs = np.zeros(n.value, dtype=np.uint64)
        states = np.zeros(n.value, dtype=np.uint64)
        check(_lib.lib().wfst_ctx_get_sweep_trace(self._h, ms.ctypes.data, arcs.ctypes.data, states.ctypes.data,
                                                  n.value, C.byref(n)))
        return ms, arcs, states

    def sweep_modes(self):
        """What ran each launch of the last profiled solve: 0 atomic sweep, 7 binned level, or the mailbox mode."""
        n = C.c_size_t()
        check(_lib.lib().wfst_ctx_get_sweep_modes(self._h, None, 0, C.byref(n)))
        modes = np.zeros(n.value, dtype=np.uint32)
        check(_lib.lib().wfst_ctx_get_sweep_modes(self._h, modes.ctypes.data, n.value, C.byref(n)))
        return modes

    def stats(self) -> dict:
        st = _lib.Stats()
        check(_lib.lib().wfst_ctx_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}


_default_ctx = threading.local()


def default_context() -> Context:
    ctx = getattr(_default_ctx, "ctx", None)
    if ctx is None:
        ctx = Context(0)
        _default_ctx.ctx = ctx
    return ctx


def set_default_context(ctx: Context):
    _default_ctx.ctx = ctx


# ------------------------------------------------------------------ device FST handle
class DeviceFst:
    """Owning wrapper of a wfst_fst handle: an FST resident in HBM as CSR."""

    def __init__(self, handle, ctx: Context, owner=None):
        self._h = handle
        self.ctx = ctx
        self._owner = owner  # a PathList that owns the handle (this object is then a view)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and getattr(self, "_owner", None) is None:
            try:
                _lib.lib().wfst_fst_destroy(h)
            except Exception:
                pass
        self._h = None

    @classmethod
    def from_arrays(cls, n_states: int, start: Optional[int], offsets, arcs, finals, props: int,
                    ctx: Optional[Context] = None) -> "DeviceFst":
        ctx = ctx or default_context()
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        arcs = np.ascontiguousarray(arcs, dtype=TR_DTYPE)
        finals = np.ascontiguousarray(finals, dtype=np.float32)
        if offsets.shape[0] != n_states + 1 or finals.shape[0] != n_states:
            raise ValueError("offsets must have n_states+1 entries and finals n_states")
        if n_states and arcs.shape[0] != int(offsets[-1]):
            raise ValueError("arcs must have offsets[-1] entries")
        h = C.c_void_p()
        check(_lib.lib().wfst_fst_upload(ctx._h, n_states, -1 if start is None else int(start), offsets.ctypes.data,
                                         arcs.ctypes.data, finals.ctypes.data, int(props), C.byref(h)),
              "wfst_fst_upload")
        return cls(h, ctx)

    @classmethod
    def from_device_arrays(cls, n_states: int, start: Optional[int], d_offsets: int, d_arcs: int, d_finals: int,
                           props: int, ctx: Optional[Context] = None) -> "DeviceFst":
        """Arrays already in HBM (raw device pointers, e.g. torch tensor .data_ptr())."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(_lib.lib().wfst_fst_upload_device(ctx._h, n_states, -1 if start is None else int(start),
                                                C.c_void_p(d_offsets), C.c_void_p(d_arcs), C.c_void_p(d_finals),
                                                int(props), C.byref(h)), "wfst_fst_upload_device")
        return cls(h, ctx)

    @classmethod
    def upload_many(cls, flats: Sequence[dict], ctx: Optional[Context] = None) -> List["DeviceFst"]:
        """flats: dicts with n_states,start,offsets,arcs,finals,props. One arena, one copy."""
        ctx = ctx or default_context()
        n = len(flats)
        if n == 0:
            return []
        n_states = np.array([f["n_states"] for f in flats], dtype=np.uint32)
        starts = np.array([-1 if f["start"] is None else f["start"] for f in flats], dtype=np.int64)
        props = np.array([f["props"] for f in flats], dtype=np.uint64)
        offsets_cat = np.concatenate([np.asarray(f["offsets"], dtype=np.uint32) for f in flats])
        arcs_cat = np.concatenate([np.asarray(f["arcs"], dtype=TR_DTYPE) for f in flats])
        finals_cat = np.concatenate([np.asarray(f["finals"], dtype=np.float32) for f in flats])
        outs = (C.c_void_p * n)()
        check(_lib.lib().wfst_fst_upload_many(ctx._h, n, n_states.ctypes.data, starts.ctypes.data,
                                              offsets_cat.ctypes.data, arcs_cat.ctypes.data, finals_cat.ctypes.data,
                                              props.ctypes.data, outs), "wfst_fst_upload_many")
        return [cls(C.c_void_p(outs[i]), ctx) for i in range(n)]

    @classmethod
    def from_bytes(cls, data: bytes, ctx: Optional[Context] = None) -> "DeviceFst":
        ctx = ctx or default_context()
        h = C.c_void_p()
        check(_lib.lib().wfst_fst_from_openfst_bytes(ctx._h, data, len(data), C.byref(h)),
              "wfst_fst_from_openfst_bytes")
        return cls(h, ctx)

    def to_bytes(self, fst_type: str = "vector") -> bytes:
        """OpenFST binary: "vector" (VectorFst::store) or "const" (ConstFst::store, version 2)."""
        if fst_type not in ("vector", "const"):
            raise ValueError("fst_type must be 'vector' or 'const'")
        p = C.c_void_p()
        n = C.c_size_t()
        fn = _lib.lib().wfst_fst_to_openfst_bytes if fst_type == "vector" else _lib.lib().wfst_fst_to_openfst_const_bytes
        check(fn(self._h, C.byref(p), C.byref(n)), "wfst_fst_to_openfst_bytes")
        try:
            return C.string_at(p.value, n.value)
        finally:
            _lib.lib().wfst_bytes_destroy(p)

    def info(self):
        n, a, s, p = C.c_uint32(), C.c_uint64(), C.c_int64(), C.c_uint64()
        check(_lib.lib().wfst_fst_info(self._h, C.byref(n), C.byref(a), C.byref(s), C.byref(p)))
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

    def to_flat(self) -> dict:
        n, a, start, props = self.info()
        offsets = np.zeros(n + 1, dtype=np.uint32)
        arcs = np.zeros(a, dtype=TR_DTYPE)
        finals = np.zeros(n, dtype=np.float32)
        check(_lib.lib().wfst_fst_download(self._h, offsets.ctypes.data, arcs.ctypes.data, finals.ctypes.data),
              "wfst_fst_download")
        return dict(n_states=n, start=start, offsets=offsets, arcs=arcs, finals=finals, props=props)

    def to_vector_fst(self) -> "VectorFst":
        v = C.c_void_p()
        check(_lib.lib().wfst_vec_fst_from_device(self._h, C.byref(v)), "wfst_vec_fst_from_device")
        return VectorFst(ptr=v, ctx=self.ctx)

    # algorithms on device handles (no host round trip of the operands)
    def compose(self, other: "DeviceFst", config: Optional["ComposeConfig"] = None) -> "DeviceFst":
        out = C.c_void_p()
        cfg = config._c() if config is not None else None
        check(_lib.lib().wfst_compose(self.ctx._h, self._h, other._h, cfg, C.byref(out)), "Error during composition")
        return DeviceFst(out, self.ctx)

    def shortest_path(self, config: Optional["ShortestPathConfig"] = None) -> "DeviceFst":
        out = C.c_void_p()
        cfg = config._c() if config is not None else None
        check(_lib.lib().wfst_shortest_path(self.ctx._h, self._h, cfg, C.byref(out)), "Error computing shortest path")
        return DeviceFst(out, self.ctx)

    def shortest_path_begin(self, config: Optional["ShortestPathConfig"] = None, ctx: Optional["Context"] = None) -> "ShortestPathJob":
        """Queue shortest_path (nshortest = 1) on this FST's context (or on `ctx`: a handle may be queried from several
        contexts of its device) and return at once (wfst_shortest_path_begin); job.finish() returns what shortest_path() returns."""
        job = C.c_void_p()
        cfg = config._c() if config is not None else None
        check(_lib.lib().wfst_shortest_path_begin((ctx or self.ctx)._h, self._h, cfg, C.byref(job)), "Error computing shortest path")
        return ShortestPathJob(job, self)

    def reverse(self) -> "DeviceFst":
        """algorithms::reverse (reverse.rs:33-87): new FST with a super-initial state 0."""
        out = C.c_void_p()
        check(_lib.lib().wfst_reverse(self.ctx._h, self._h, C.byref(out)), "Error during reverse")
        return DeviceFst(out, self.ctx)

    def rm_epsilon(self) -> "DeviceFst":
        """algorithms::rm_epsilon, default config (rm_epsilon_static.rs:50-163): a NEW FST without epsilon:epsilon arcs."""
        out = C.c_void_p()
        check(_lib.lib().wfst_rm_epsilon(self.ctx._h, self._h, C.byref(out)), "Error during rm_epsilon")
        return DeviceFst(out, self.ctx)

    def connect(self) -> "DeviceFst":
        """algorithms::connect (connect.rs:51-66): a NEW FST with the accessible and coaccessible states only."""
        out = C.c_void_p()
        check(_lib.lib().wfst_connect(self.ctx._h, self._h, C.byref(out)), "Error during connect")
        return DeviceFst(out, self.ctx)

    def project(self, proj_type: Optional["ProjectType"] = None) -> "DeviceFst":
        """In-place projection on the device (algorithms/projection.rs:65-95): PROJECT_INPUT copies the input labels
        over the output labels, PROJECT_OUTPUT the other way round."""
        out = proj_type is not None and ProjectType(proj_type) == ProjectType.PROJECT_OUTPUT
        check(_lib.lib().wfst_fst_project(self.ctx._h, self._h, 1 if out else 0), "Error during projection")
        return self

    def tr_sort(self, ilabel_cmp: bool = True) -> "DeviceFst":
        """In-place stable per-state arc sort on the device by ilabel (ILabelCompare) or olabel
        (OLabelCompare) + the reference's property update: algorithms/tr_sort.rs:13-62,
        rustfst-python fst.tr_sort (rustfst/algorithms/tr_sort.py)."""
        check(_lib.lib().wfst_fst_tr_sort(self.ctx._h, self._h, 1 if ilabel_cmp else 0), "Error during tr_sort")
        return self

    def set_start(self, state: int) -> "DeviceFst":
        """MutableFst::set_start (fst_impls/vector_fst/mutable_fst.rs:35-44) on the device-resident handle: the arcs stay in
        HBM, the derived data that does not depend on the start state stays cached.  KO for a state beyond the FST."""
        check(_lib.lib().wfst_fst_set_start(self.ctx._h, self._h, int(state)), "Error setting start state")
        return self

    def shortest_distance(self, want_hops: bool = False):
        n = self.num_states
        dist = np.zeros(n, dtype=np.float32)
        hops = np.zeros(n, dtype=np.uint32) if want_hops else None
        check(_lib.lib().wfst_shortest_distance(self.ctx._h, self._h, dist.ctypes.data,
                                                hops.ctypes.data if want_hops else None), "wfst_shortest_distance")
        return (dist, hops) if want_hops else dist


class HandleArray:
    """A batch of DeviceFst handles marshalled once for the C-ABI (`const wfst_fst* const*`): callers that submit the
    same acceptors repeatedly build it once instead of paying the ctypes marshalling on every call."""

    def __init__(self, fsts: Sequence["DeviceFst"]):
        self._keep = list(fsts)
        n = len(self._keep)
        self._arr = (C.c_void_p * n)(*[a._h.value if isinstance(a._h, C.c_void_p) else a._h for a in self._keep])

    def __len__(self):
        return len(self._keep)


class PathList(Sequence):
    """The outs[] of a fused batch: owns the n result handles (released with ONE wfst_fst_destroy_many call) and
    hands out DeviceFst views on demand — a step that only forwards the results pays no per-path Python work."""

    def __init__(self, handles, n: int, ctx: Context):
        self._arr, self._n, self.ctx = handles, n, ctx

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return DeviceFst(C.c_void_p(self._arr[i]), self.ctx, owner=self)

    def __del__(self):
        arr = getattr(self, "_arr", None)
        if arr is not None:
            try:
                _lib.lib().wfst_fst_destroy_many(arr, self._n)
            except Exception:
                pass
            self._arr = None


class LookAhead:
    """Look-ahead composition set up on a first operand (wfst_lookahead_*): what the reference builds with
    MatcherFst::new_with_relabeling + LabelLookAheadMatcher + PushLabels(PushWeights(LookAhead(AltSequence)))
    (rustfst-cli/src/cmds/compose.rs:77-181).

        la = LookAhead(fst1_on_device)           # reachability data + relabelled fst1 (MatcherFst::new)
        fst2r = la.relabel(fst2_on_device)        # LabelLookAheadRelabeler::relabel + tr_sort(ILabelCompare)
        out = la.compose(fst2r)                   # ComposeFst(..).compute(), not connected
    """

    def __init__(self, fst1: Optional["DeviceFst"] = None, _handle=None, _ctx=None):
        if _handle is not None:
            self._h, self.ctx = _handle, _ctx
            return
        h = C.c_void_p()
        check(_lib.lib().wfst_lookahead_create(fst1.ctx._h, fst1._h, C.byref(h)), "wfst_lookahead_create")
        self._h, self.ctx = h, fst1.ctx

    @classmethod
    def reachable_from_arrays(cls, n_states: int, offsets, arcs, finals, reach_input: bool = False) -> "LookAhead":
        """Host-only handle: LabelReachable::compute_data on flat CSR arrays (no GPU); serves data() only."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        arcs = np.ascontiguousarray(arcs, dtype=TR_DTYPE)
        finals = np.ascontiguousarray(finals, dtype=np.float32)
        h = C.c_void_p()
        check(_lib.lib().wfst_label_reachable_compute(n_states, offsets.ctypes.data, arcs.ctypes.data if len(arcs) else None,
                                                      finals.ctypes.data if n_states else None, 1 if reach_input else 0,
                                                      C.byref(h)), "wfst_label_reachable_compute")
        return cls(_handle=h, _ctx=None)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().wfst_lookahead_destroy(h)
            except Exception:
                pass
        self._h = None

    @property
    def fst1(self) -> "DeviceFst":
        """The relabelled, olabel-sorted first operand (a view: the handle stays owned by this object)."""
        h = C.c_void_p()
        check(_lib.lib().wfst_lookahead_fst1(self._h, C.byref(h)), "wfst_lookahead_fst1")
        return DeviceFst(h, self.ctx, owner=self)

    def relabel(self, fst2: "DeviceFst") -> "DeviceFst":
        out = C.c_void_p()
        check(_lib.lib().wfst_lookahead_relabel(self._h, fst2._h, C.byref(out)), "wfst_lookahead_relabel")
        return DeviceFst(out, fst2.ctx)

    def compose(self, relabeled_fst2: "DeviceFst", ctx: Optional[Context] = None) -> "DeviceFst":
        ctx = ctx or self.ctx
        out = C.c_void_p()
        check(_lib.lib().wfst_compose_lookahead(ctx._h, self._h, relabeled_fst2._h, C.byref(out)), "Error during look-ahead composition")
        return DeviceFst(out, ctx)

    def compose_batch(self, relabeled_fst2s: Sequence["DeviceFst"], ctx: Optional[Context] = None) -> List["DeviceFst"]:
        """n look-ahead compositions against this first operand in one launch (wfst_compose_lookahead_batch)."""
        ctx = ctx or self.ctx
        n = len(relabeled_fst2s)
        arr = (C.c_void_p * n)(*[f._h for f in relabeled_fst2s])
        outs = (C.c_void_p * n)()
        check(_lib.lib().wfst_compose_lookahead_batch(ctx._h, self._h, arr, n, outs), "Error during look-ahead composition")
        return [DeviceFst(C.c_void_p(outs[i]), ctx) for i in range(n)]

    def data(self) -> dict:
        """LabelReachableData: dict(final_label, label2index {label: index}, intervals [per state list of (begin, end)])."""
        n, ni, nl, fl = C.c_uint32(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        check(_lib.lib().wfst_lookahead_info(self._h, C.byref(n), C.byref(ni), C.byref(nl), C.byref(fl)), "wfst_lookahead_info")
        off = np.zeros(n.value + 1, dtype=np.uint32)
        iv = np.zeros(2 * ni.value, dtype=np.uint32)
        labels = np.zeros(nl.value, dtype=np.uint32)
        idx = np.zeros(nl.value, dtype=np.uint32)
        check(_lib.lib().wfst_lookahead_download(self._h, off.ctypes.data, iv.ctypes.data if len(iv) else None,
                                                 labels.ctypes.data if nl.value else None,
                                                 idx.ctypes.data if nl.value else None), "wfst_lookahead_download")
        ivs = [[(int(iv[2 * k]), int(iv[2 * k + 1])) for k in range(off[s], off[s + 1])] for s in range(n.value)]
        return {"final_label": int(fl.value), "label2index": {int(l): int(i) for l, i in zip(labels, idx)}, "intervals": ivs}


class ShortestPathJob:
    """A single-shortest-path solve in flight (wfst_shortest_path_begin); finish() = wfst_shortest_path_end."""

    def __init__(self, job, fst):
        self._job, self._fst = job, fst  # the input FST must outlive the job

    def finish(self) -> "DeviceFst":
        if self._job is None:
            raise WfstError("shortest_path job already finished")
        out = C.c_void_p()
        job, self._job = self._job, None
        check(_lib.lib().wfst_shortest_path_end(job, C.byref(out)), "Error computing shortest path")
        fst, self._fst = self._fst, None
        return DeviceFst(out, fst.ctx)

    def __del__(self):
        if getattr(self, "_job", None) is not None:  # abandoned: wait for the kernels and free the job
            _lib.lib().wfst_shortest_path_end(self._job, None)
            self._job = None


class BatchJob:
    """A fused batch in flight (wfst_compose_shortest_path_batch_begin); finish() = ..._end."""

    def __init__(self, job, n, ctx, keep_alive):
        self._job, self._n, self._ctx, self._keep = job, n, ctx, keep_alive

    def finish(self):
        if self._job is None:
            raise WfstError("batch job already finished")
        outs = (C.c_void_p * self._n)()
        na = C.c_uint64()
        job, self._job = self._job, None
        check(_lib.lib().wfst_compose_shortest_path_batch_end(job, outs, C.byref(na)),
              "wfst_compose_shortest_path_batch_end")
        self._keep = None
        return PathList(outs, self._n, self._ctx), na.value

    def __del__(self):
        if getattr(self, "_job", None) is not None:  # abandoned: wait for the kernel and free the job
            _lib.lib().wfst_compose_shortest_path_batch_end(self._job, None, None)
            self._job = None


def compose_shortest_path_batch_begin(acceptors: Sequence[DeviceFst], t: DeviceFst,
                                      compose_config: Optional["ComposeConfig"] = None,
                                      shortest_path_config: Optional["ShortestPathConfig"] = None,
                                      ctx: Optional[Context] = None) -> BatchJob:
    """Enqueue the fused batch on `ctx`'s stream (default: t's context) and return at once; work issued
    afterwards on ANOTHER context (e.g. shortest_path of a large FST) overlaps with it on the GPU."""
    n = len(acceptors)
    ctx = ctx or t.ctx
    arr = acceptors._arr if isinstance(acceptors, HandleArray) else HandleArray(acceptors)._arr
    job = C.c_void_p()
    check(_lib.lib().wfst_compose_shortest_path_batch_begin(
        ctx._h, arr, n, t._h, compose_config._c() if compose_config else None,
        shortest_path_config._c() if shortest_path_config else None, C.byref(job)),
        "wfst_compose_shortest_path_batch_begin")
    return BatchJob(job, n, ctx, (acceptors, t, arr))


def compose_shortest_path_batch(acceptors: Sequence[DeviceFst], t: DeviceFst,
                                compose_config: Optional["ComposeConfig"] = None,
                                shortest_path_config: Optional["ShortestPathConfig"] = None,
                                ctx: Optional[Context] = None):
    """for a in acceptors: shortest_path(compose(a, t)) as one device-resident pipeline.
    Returns (list of DeviceFst paths, total composed arcs before trimming)."""
    n = len(acceptors)
    ctx = ctx or t.ctx
    arr = acceptors._arr if isinstance(acceptors, HandleArray) else HandleArray(acceptors)._arr
    outs = (C.c_void_p * n)()
    na = C.c_uint64()
    check(_lib.lib().wfst_compose_shortest_path_batch(
        ctx._h, arr, n, t._h, compose_config._c() if compose_config else None,
        shortest_path_config._c() if shortest_path_config else None, outs, C.byref(na)),
        "wfst_compose_shortest_path_batch")
    return PathList(outs, n, ctx), na.value


def compose_shortest_path_batch_packed(acceptors: Sequence[DeviceFst], t: DeviceFst, max_arcs: int,
                                       compose_config: Optional["ComposeConfig"] = None,
                                       shortest_path_config: Optional["ShortestPathConfig"] = None,
                                       ctx: Optional[Context] = None, out: Optional[np.ndarray] = None):
    """compose_shortest_path_batch with the results as one table instead of handles
    (wfst_compose_shortest_path_batch_packed): returns (uint32 array [n, 4 + 4 * max_arcs] in the record layout of
    dist.pack_paths / dist.unpack_paths, total composed arcs).  No per-path host object is built: the form for large batches.
    `out`: a C-contiguous uint32 array of that shape to fill (a caller that decodes batch after batch reuses one)."""
    n = len(acceptors)
    ctx = ctx or t.ctx
    arr = acceptors._arr if isinstance(acceptors, HandleArray) else HandleArray(acceptors)._arr
    shape = (n, 4 + 4 * int(max_arcs))
    if out is None:
        out = np.empty(shape, dtype=np.uint32)
    elif out.shape != shape or out.dtype != np.uint32 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError(f"out must be a C-contiguous uint32 array of shape {shape}")
    na = C.c_uint64()
    check(_lib.lib().wfst_compose_shortest_path_batch_packed(
        ctx._h, arr, n, t._h, compose_config._c() if compose_config else None,
        shortest_path_config._c() if shortest_path_config else None, int(max_arcs), out.ctypes.data, C.byref(na)),
        "wfst_compose_shortest_path_batch_packed")
    return out, na.value


def shortest_path_batch(fsts: Sequence[DeviceFst], config: Optional["ShortestPathConfig"] = None,
                        ctx: Optional[Context] = None) -> List[DeviceFst]:
    """[f.shortest_path(config) for f in fsts] as ONE call (wfst_shortest_path_batch): with nshortest > 1 the small
    inputs — the composed lattices of a decoding batch — are searched by one launch, one wavefront each."""
    n = len(fsts)
    if n == 0:
        return []
    ctx = ctx or fsts[0].ctx
    arr = fsts._arr if isinstance(fsts, HandleArray) else (C.c_void_p * n)(*[f._h.value if isinstance(f._h, C.c_void_p) else f._h for f in fsts])
    outs = (C.c_void_p * n)()
    check(_lib.lib().wfst_shortest_path_batch(ctx._h, arr, n, config._c() if config is not None else None, outs),
          "wfst_shortest_path_batch")
    return [DeviceFst(C.c_void_p(outs[i]), ctx) for i in range(n)]


def last_nbest_path(ctx: Optional[Context] = None) -> str:
    """Which search the last shortest_path_batch(nshortest > 1) used (wfst_stats.nbest_device_problems)."""
    ctx = ctx or default_context()
    return "wave kernel: %d inputs" % int(ctx.stats()["nbest_device_problems"])


# ------------------------------------------------------------------ configs
class ProjectType(Enum):  # rustfst-python/rustfst/algorithms/project.py:9-24
    PROJECT_INPUT = 0
    PROJECT_OUTPUT = 1


class ComposeFilter(Enum):  # rustfst-python/rustfst/algorithms/compose.py:55-62
    AUTOFILTER = 0
    NULLFILTER = 1
    TRIVIALFILTER = 2
    SEQUENCEFILTER = 3
    ALTSEQUENCEFILTER = 4
    MATCHFILTER = 5
    NOMATCHFILTER = 6


class ComposeConfig:
    """rustfst-python/rustfst/algorithms/compose.py:65-113 (sigma matcher configs are not supported)."""

    def __init__(self, compose_filter: ComposeFilter = ComposeFilter.AUTOFILTER, connect: bool = True,
                 matcher1_config=None, matcher2_config=None):
        if matcher1_config is not None or matcher2_config is not None:
            raise WfstError("unsupported: custom MatcherConfig (sigma matcher) is not implemented on the GPU path")
        self.compose_filter = compose_filter
        self.connect = bool(connect)

    def _c(self):
        return C.pointer(_lib.ComposeConfig(self.compose_filter.value, 1 if self.connect else 0))


class ShortestPathConfig:
    """rustfst-python/rustfst/algorithms/shortest_path.py:14-38."""

    def __init__(self, nshortest: int = 1, unique: bool = False, delta: Union[float, None] = None):
        self.nshortest = int(nshortest)
        self.unique = bool(unique)
        self.delta = KSHORTESTDELTA if delta is None else float(delta)

    def _c(self):
        return C.pointer(_lib.ShortestPathConfig(self.delta, self.nshortest, 1 if self.unique else 0))


# ------------------------------------------------------------------ VectorFst mirror
class VectorFst:
    """Mutable FST stored in vectors (rustfst-python/rustfst/fst/vector_fst.py:29-790, subset on the path)."""

    def __init__(self, ptr=None, ctx: Optional[Context] = None):
        self._ctx = ctx
        self._dev: Optional[DeviceFst] = None
        if ptr is not None:
            self._p = ptr
        else:
            p = C.c_void_p()
            check(_lib.lib().wfst_vec_fst_new(C.byref(p)), "Something went wrong when creating the Fst struct")
            self._p = p

    def __del__(self):
        p = getattr(self, "_p", None)
        if p:
            try:
                _lib.lib().wfst_vec_fst_destroy(p)
            except Exception:
                pass
            self._p = None

    # -- mutation
    def add_state(self) -> int:
        s = C.c_uint32()
        check(_lib.lib().wfst_vec_fst_add_state(self._p, C.byref(s)), "Error during `add_state`")
        self._dev = None
        return s.value

    def add_tr(self, state: int, tr: Tr) -> "VectorFst":
        arr = np.array([(tr.ilabel, tr.olabel, tr.weight, tr.next_state)], dtype=TR_DTYPE)
        check(_lib.lib().wfst_vec_fst_add_tr(self._p, state, arr.ctypes.data), "Error during `add_tr`")
        self._dev = None
        return self

    def set_start(self, state: int):
        check(_lib.lib().wfst_vec_fst_set_start(self._p, state), "Error setting start state")
        self._dev = None

    def set_final(self, state: int, weight: Union[float, None] = None):
        check(_lib.lib().wfst_vec_fst_set_final(self._p, state, 0.0 if weight is None else weight),
              "Error setting final state")
        self._dev = None

    def unset_final(self, state: int):
        check(_lib.lib().wfst_vec_fst_del_final_weight(self._p, state), "Error unsetting final state")
        self._dev = None

    def tr_sort(self, ilabel_cmp: bool = True) -> "VectorFst":
        check(_lib.lib().wfst_vec_fst_tr_sort(self._p, 1 if ilabel_cmp else 0), "Error during tr_sort")
        self._dev = None
        return self

    # -- inspection
    def num_states(self) -> int:
        n = C.c_uint32()
        check(_lib.lib().wfst_vec_fst_num_states(self._p, C.byref(n)), "Error getting number of states")
        return n.value

    def start(self) -> Optional[int]:
        s = C.c_int64()
        check(_lib.lib().wfst_vec_fst_start(self._p, C.byref(s)), "Error getting start state")
        return None if s.value < 0 else s.value

    def final_weight(self, state: int) -> Optional[float]:
        w = C.c_float()
        some = C.c_int()
        check(_lib.lib().wfst_vec_fst_final_weight(self._p, state, C.byref(w), C.byref(some)),
              "Error getting final weight")
        return w.value if some.value else None

    def is_final(self, state: int) -> bool:
        return self.final_weight(state) is not None

    def num_trs(self, state: int) -> int:
        n = C.c_uint64()
        check(_lib.lib().wfst_vec_fst_num_trs(self._p, state, C.byref(n)), "Error getting number of trs")
        return n.value

    def trs(self, state: int) -> List[Tr]:
        n = self.num_trs(state)
        arr = np.zeros(n, dtype=TR_DTYPE)
        got = C.c_uint64()
        check(_lib.lib().wfst_vec_fst_get_trs(self._p, state, arr.ctypes.data, n, C.byref(got)), "Error getting trs")
        return [Tr(int(a["ilabel"]), int(a["olabel"]), float(a["weight"]), int(a["nextstate"])) for a in arr]

    def properties(self) -> int:
        p = C.c_uint64()
        check(_lib.lib().wfst_vec_fst_properties(self._p, C.byref(p)))
        return p.value

    def equals(self, other: "VectorFst") -> bool:
        eq = C.c_int()
        check(_lib.lib().wfst_vec_fst_equals(self._p, other._p, C.byref(eq)), "Error checking equality")
        return bool(eq.value)

    def __eq__(self, other):
        return self.equals(other)

    def copy(self) -> "VectorFst":
        p = C.c_void_p()
        check(_lib.lib().wfst_vec_fst_copy(self._p, C.byref(p)), "Error copying fst")
        return VectorFst(ptr=p, ctx=self._ctx)

    def __str__(self):
        lines = []
        for s in range(self.num_states()):
            for t in self.trs(s):
                lines.append(f"{s}\t{t.next_state}\t{t.ilabel}\t{t.olabel}\t{t.weight}")
            fw = self.final_weight(s)
            if fw is not None:
                lines.append(f"{s}\t{fw}")
        return "\n".join(lines)

    # -- device residency
    def to_device(self, ctx: Optional[Context] = None) -> DeviceFst:
        ctx = ctx or self._ctx or default_context()
        if self._dev is None or self._dev.ctx is not ctx:
            h = C.c_void_p()
            check(_lib.lib().wfst_vec_fst_to_device(ctx._h, self._p, C.byref(h)), "wfst_vec_fst_to_device")
            self._dev = DeviceFst(h, ctx)
        return self._dev

    # -- I/O (rustfst-python vector_fst.py:311-388)
    @classmethod
    def from_bytes(cls, data: bytes, ctx: Optional[Context] = None) -> "VectorFst":
        return DeviceFst.from_bytes(data, ctx).to_vector_fst()

    def to_bytes(self) -> bytes:
        return self.to_device().to_bytes()

    @classmethod
    def read(cls, filename) -> "VectorFst":
        with open(filename, "rb") as f:
            return cls.from_bytes(f.read())

    def write(self, filename):
        with open(filename, "wb") as f:
            f.write(self.to_bytes())

    # -- algorithms (vector_fst.py:419-436, 621-638)
    def compose(self, other: "VectorFst", config: Union[ComposeConfig, None] = None) -> "VectorFst":
        return self.to_device().compose(other.to_device(self.to_device().ctx), config).to_vector_fst()

    def shortest_path(self, config: Union[ShortestPathConfig, None] = None) -> "VectorFst":
        return self.to_device().shortest_path(config).to_vector_fst()

    def rm_epsilon(self) -> "VectorFst":
        """rustfst-python vector_fst.py `rm_epsilon` (algorithms/rm_epsilon.py): in place, returns this FST."""
        res = self.to_device().rm_epsilon().to_vector_fst()
        self._p, res._p = res._p, self._p
        self._dev = None
        return self

    def connect(self) -> "VectorFst":
        """rustfst-python vector_fst.py `connect` (algorithms/connect.py): trims this FST in place and returns it."""
        trimmed = self.to_device().connect().to_vector_fst()
        self._p, trimmed._p = trimmed._p, self._p  # this object now owns the trimmed FST; the old one dies with `trimmed`
        self._dev = None
        return self

    def project(self, proj_type: Union["ProjectType", None] = None) -> "VectorFst":
        """rustfst-python vector_fst.py:525-538 `project` (algorithms/project.py:27-50): projects THIS FST in place and
        returns it (the reference returns self).  The device copy is projected and becomes this object's host data; the
        cached device handle is dropped so that trs(), equality and later compose / shortest_path calls all see the
        projected labels."""
        projected = self.to_device().project(proj_type).to_vector_fst()
        self._p, projected._p = projected._p, self._p
        self._dev = None
        return self


# ------------------------------------------------------------------ free functions
def compose(fst: VectorFst, fst2: VectorFst) -> VectorFst:
    """rustfst-python/rustfst/algorithms/compose.py `compose`."""
    return fst.compose(fst2)


def compose_with_config(fst: VectorFst, fst2: VectorFst, config: ComposeConfig) -> VectorFst:
    return fst.compose(fst2, config)


def project(fst: VectorFst, proj_type: ProjectType = ProjectType.PROJECT_INPUT) -> VectorFst:
    """rustfst-python/rustfst/algorithms/project.py:27-50."""
    return fst.project(proj_type)


def shortestpath(fst: VectorFst) -> VectorFst:
    """rustfst-python/rustfst/algorithms/shortest_path.py:41-56."""
    return fst.shortest_path()


def shortestpath_with_config(fst: VectorFst, config: ShortestPathConfig) -> VectorFst:
    return fst.shortest_path(config)


def acceptor(labels: Sequence[int], weight: float = 0.0) -> VectorFst:
    """utils::acceptor (rustfst/src/utils/labels_to_fst.rs:111-132): L+1 states, unit arc weights,
    last state final with `weight`."""
    fst = VectorFst()
    cur = fst.add_state()
    fst.set_start(cur)
    for l in labels:
        nxt = fst.add_state()
        fst.add_tr(cur, Tr(l, l, 0.0, nxt))
        cur = nxt
    fst.set_final(cur, weight)
    return fst
