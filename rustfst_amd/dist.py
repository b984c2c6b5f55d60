"""Multi-GPU plumbing for the batch path: the unit of work is one
(acceptor, T) problem; units are independent, so acceptor i goes to rank i mod G, T is replicated
in every GPU's HBM, there is NO collective during compute, and the only exchange is a gather of
the finished paths (a few KB) — an all-gather over RCCL/xGMI on GPUs, gloo in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from ._lib import TR_DTYPE


def shard_indices(n_total: int, rank: int, world: int) -> List[int]:
    """Global problem indices owned by `rank` (round robin: i mod world == rank)."""
    return list(range(rank, n_total, world))


def pack_paths(paths: Sequence[dict], max_arcs: int) -> np.ndarray:
    """Fixed-size records so that one all-gather moves every rank's results:
    per path: [n_arcs u32, final_weight f32 bits, pad, pad] + max_arcs x 16-byte arcs, as uint32 words.
    `paths` are flat FST dicts of shortest-path outputs (n_states = n_arcs + 1, or 0 when empty)."""
    rec = 4 + 4 * max_arcs
    out = np.zeros((len(paths), rec), dtype=np.uint32)
    for i, p in enumerate(paths):
        n_arcs = max(int(p["n_states"]) - 1, 0)
        if n_arcs > max_arcs:
            raise ValueError(f"path with {n_arcs} arcs exceeds the gather record ({max_arcs})")
        out[i, 0] = n_arcs
        out[i, 1] = np.float32(p["finals"][0]).view(np.uint32) if p["n_states"] else np.float32(np.inf).view(np.uint32)
        out[i, 2] = 1 if p["n_states"] else 0
        if n_arcs:
            out[i, 4:4 + 4 * n_arcs] = np.ascontiguousarray(p["arcs"]).view(np.uint32)
    return out


def pack_device_paths(paths, max_arcs: int) -> np.ndarray:
    """Same record layout as pack_paths, filled by the library from DeviceFst path handles (one C call)."""
    import ctypes as C

    from . import _lib
    n = len(paths)
    out = np.zeros((n, 4 + 4 * max_arcs), dtype=np.uint32)
    if hasattr(paths, "_arr"):  # a PathList: its handle array goes to the library as it is
        arr = paths._arr
    else:
        arr = (C.c_void_p * n)(*[p._h.value if isinstance(p._h, C.c_void_p) else p._h for p in paths])
    _lib.check(_lib.lib().wfst_fst_pack_paths(arr, n, max_arcs, out.ctypes.data), "wfst_fst_pack_paths")
    return out


def unpack_paths(packed: np.ndarray) -> List[dict]:
    """Inverse of pack_paths (property words are not transported; they are a function of the path)."""
    res = []
    for row in packed:
        n_arcs = int(row[0])
        if not row[2]:
            res.append(dict(n_states=0, start=None, offsets=np.zeros(1, np.uint32), arcs=np.zeros(0, TR_DTYPE),
                            finals=np.zeros(0, np.float32)))
            continue
        arcs = row[4:4 + 4 * n_arcs].copy().view(TR_DTYPE)
        finals = np.full(n_arcs + 1, np.inf, dtype=np.float32)
        finals[0] = row[1:2].view(np.float32)[0]
        offsets = np.concatenate([[0], np.arange(0, n_arcs + 1)]).astype(np.uint32)
        res.append(dict(n_states=n_arcs + 1, start=n_arcs, offsets=offsets, arcs=arcs, finals=finals))
    return res


_gather_bufs = {}


def gather_paths(local_packed: np.ndarray, world: int, device=None):
    """All-gather equally shaped per-rank result blocks. Returns [world, n_local, rec] uint32 (numpy).
    Uses torch.distributed's default group: backend "nccl" (= RCCL over xGMI) when `device` is a GPU,
    gloo on CPU.  On a GPU the staging buffers (pinned host in/out, device in/out) are allocated once per shape,
    so a call is: memcpy into pinned, async H2D, all-gather, async D2H, one stream synchronisation."""
    import torch
    import torch.distributed as dist

    shape = tuple(local_packed.shape)
    if device is None:
        t = torch.from_numpy(local_packed.view(np.int32))
        # concatenation layout along dim 0 (accepted by both the gloo and the nccl/RCCL backends)
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(out, t)
        return out.numpy().view(np.uint32).reshape((world,) + shape)
    key = (shape, world, str(device))
    bufs = _gather_bufs.get(key)
    if bufs is None:
        h_in = torch.empty(shape, dtype=torch.int32).pin_memory()
        h_out = torch.empty((world * shape[0],) + shape[1:], dtype=torch.int32).pin_memory()
        d_in = torch.empty(shape, dtype=torch.int32, device=device)
        d_out = torch.empty((world * shape[0],) + shape[1:], dtype=torch.int32, device=device)
        bufs = _gather_bufs[key] = (h_in, h_out, d_in, d_out)
    h_in, h_out, d_in, d_out = bufs
    h_in.numpy()[...] = local_packed.view(np.int32)
    d_in.copy_(h_in, non_blocking=True)
    dist.all_gather_into_tensor(d_out, d_in)
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.current_stream(device).synchronize()
    return h_out.numpy().view(np.uint32).reshape((world,) + shape).copy()


class PendingGather:
    """An all-gather of packed result records in flight on the current torch stream (gather_paths_async)."""

    def __init__(self, h_out, event, world, shape):
        self._h_out, self._event, self._world, self._shape = h_out, event, world, shape

    def result(self) -> np.ndarray:
        """Waits for the exchange; returns [world, n_local, rec] uint32."""
        self._event.synchronize()
        return self._h_out.numpy().view(np.uint32).reshape((self._world,) + self._shape).copy()


_async_bufs = {}


def gather_paths_async(local_packed: np.ndarray, world: int, device) -> PendingGather:
    """gather_paths without the final wait: H2D, all-gather and D2H are queued on the current torch stream and the
    call returns; the exchange then overlaps with whatever the caller does next (the next decoding step) and
    `.result()` collects it.  Two buffer sets alternate, so one exchange may be in flight while the next is prepared."""
    import torch
    import torch.distributed as dist

    shape = tuple(local_packed.shape)
    key = (shape, world, str(device))
    st = _async_bufs.get(key)
    if st is None:
        sets = []
        for _ in range(2):
            sets.append((torch.empty(shape, dtype=torch.int32).pin_memory(),
                         torch.empty((world * shape[0],) + shape[1:], dtype=torch.int32).pin_memory(),
                         torch.empty(shape, dtype=torch.int32, device=device),
                         torch.empty((world * shape[0],) + shape[1:], dtype=torch.int32, device=device)))
        st = _async_bufs[key] = {"sets": sets, "next": 0}
    h_in, h_out, d_in, d_out = st["sets"][st["next"]]
    st["next"] ^= 1
    h_in.numpy()[...] = local_packed.view(np.int32)
    d_in.copy_(h_in, non_blocking=True)
    dist.all_gather_into_tensor(d_out, d_in)
    h_out.copy_(d_out, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return PendingGather(h_out, ev, world, shape)


class Comm:
    """RCCL communicator of the library itself (include/wfst.h: wfst_comm_*): the exchange runs inside libwfst_amd —
    packing into pinned memory, H2D, ncclAllGather and D2H on a stream of the communicator's own — with no torch call on
    the way.  `unique_id()` on rank 0, hand the 128 bytes to every rank (any channel: here `from_torch_group` broadcasts
    them over the process group torchrun set up), then `Comm(ctx, id, rank, world)` on every rank."""

    def __init__(self, ctx, unique_id: bytes, rank: int, world: int):
        import ctypes as C

        from . import _lib
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of Comm.unique_id()")
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().wfst_comm_create(ctx._h, buf, rank, world, C.byref(h)), "wfst_comm_create")
        self._h, self.ctx, self.rank, self.world = h, ctx, rank, world
        self._shape = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                from . import _lib
                _lib.lib().wfst_comm_destroy(h)
            except Exception:
                pass
            self._h = None

    @classmethod
    def from_host_transport(cls, rank: int, world: int, allgather) -> "Comm":
        """The library's communicator over a HOST all-gather (wfst_comm_create_host): `allgather(send, recv)` gets two
        uint8 numpy views (bytes of this rank / world * bytes, rank-major) and fills `recv`.  No GPU, no RCCL: MPI or gloo
        deployments, and the CPU tests of the exchange code (staging sets, record layout, ragged gathers)."""
        import ctypes as C

        from . import _lib

        def _cb(_user, send, recv, nbytes):
            try:
                s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(max(int(nbytes), 1),))[:int(nbytes)]
                r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(max(int(nbytes) * world, 1),))[:int(nbytes) * world]
                allgather(s, r)
                return 0
            except Exception:  # (an exception must not cross the C frame)
                return 1

        self = cls.__new__(cls)
        self._cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)(_cb)  # kept alive with the object
        h = C.c_void_p()
        _lib.check(_lib.lib().wfst_comm_create_host(rank, world, C.cast(self._cb, C.c_void_p), None, C.byref(h)), "wfst_comm_create_host")
        self._h, self.ctx, self.rank, self.world = h, None, rank, world
        self._shape = None
        return self

    @classmethod
    def from_torch_group_host(cls) -> "Comm":
        """Host transport over torch.distributed's default group (gloo): the CPU counterpart of from_torch_group."""
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()

        def allgather(send, recv):
            t = torch.from_numpy(np.ascontiguousarray(send))
            out = torch.empty(world * max(t.numel(), 1), dtype=torch.uint8)[:world * t.numel()]
            if t.numel():
                dist.all_gather_into_tensor(out, t)
                recv[...] = out.numpy()

        return cls.from_host_transport(rank, world, allgather)

    def gather_records_begin(self, records: np.ndarray, max_arcs: int):
        """gather_paths_begin for records that exist already ([n, 4 + 4 * max_arcs] uint32: pack_paths, or the table of
        compose_shortest_path_batch_packed): no handles, no packing."""
        from . import _lib
        r = np.ascontiguousarray(records, dtype=np.uint32)
        n = r.shape[0]
        if r.ndim != 2 or r.shape[1] != 4 + 4 * max_arcs:
            raise ValueError("records must be [n, 4 + 4 * max_arcs] uint32")
        _lib.check(_lib.lib().wfst_gather_records_begin(self._h, r.ctypes.data, n, max_arcs), "wfst_gather_records_begin")
        self._shape = (n, 4 + 4 * max_arcs)

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _lib
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().wfst_comm_unique_id(buf), "wfst_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_group(cls, ctx, device=None) -> "Comm":
        """Rendezvous through torch.distributed's default group (gloo or nccl): rank 0's id is broadcast, every rank joins."""
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        kw = {} if device is None else {"device": device}
        t = torch.zeros(128, dtype=torch.uint8, **kw)
        if rank == 0:
            t = torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8).to(t.device)
        dist.broadcast(t, 0)
        return cls(ctx, bytes(t.cpu().numpy().tobytes()), rank, world)

    def order_after(self, ctx):
        """The next exchange starts after everything queued on ctx's stream so far (wfst_comm_order_after)."""
        from . import _lib
        _lib.check(_lib.lib().wfst_comm_order_after(self._h, ctx._h), "wfst_comm_order_after")

    def gather_paths_begin(self, paths, max_arcs: int):
        """Queues the all-gather of this rank's path FSTs (a PathList or a list of DeviceFst) and returns at once."""
        import ctypes as C

        from . import _lib
        n = len(paths)
        if hasattr(paths, "_arr"):
            arr = paths._arr
        else:
            arr = (C.c_void_p * n)(*[p._h.value if isinstance(p._h, C.c_void_p) else p._h for p in paths])
        _lib.check(_lib.lib().wfst_gather_paths_begin(self._h, arr, n, max_arcs), "wfst_gather_paths_begin")
        self._shape = (n, 4 + 4 * max_arcs)

    def gather_paths_end(self) -> np.ndarray:
        """Waits for the exchange queued by gather_paths_begin; returns [world, n_local, rec] uint32."""
        from . import _lib
        n, rec = self._shape
        out = np.empty((self.world, n, rec), dtype=np.uint32)
        _lib.check(_lib.lib().wfst_gather_paths_end(self._h, out.ctypes.data), "wfst_gather_paths_end")
        self._shape = None
        return out

    def gather_paths(self, paths, max_arcs: int) -> np.ndarray:
        self.gather_paths_begin(paths, max_arcs)
        return self.gather_paths_end()

    def allgather(self, block: np.ndarray) -> np.ndarray:
        """Fixed-size all-gather of a contiguous array per rank; returns [world, ...block.shape]."""
        from . import _lib
        b = np.ascontiguousarray(block)
        _lib.check(_lib.lib().wfst_comm_allgather_begin(self._h, b.ctypes.data, b.nbytes), "wfst_comm_allgather_begin")
        out = np.empty((self.world,) + b.shape, dtype=b.dtype)
        _lib.check(_lib.lib().wfst_comm_allgather_end(self._h, out.ctypes.data), "wfst_comm_allgather_end")
        return out

    def gather_fsts(self, local: Sequence[bytes]) -> List[List[bytes]]:
        """Ragged all-gather of serialised FSTs (OpenFST binary): every rank gets `[rank][i] -> bytes` (same contract as
        the module-level gather_fsts, one wfst_comm_allgatherv call: the sizes, then the payloads)."""
        import ctypes as C

        from . import _lib
        sizes = np.array([len(local)] + [len(b) for b in local], dtype=np.int64)
        blob = sizes.tobytes() + b"".join(local)
        rsizes = (C.c_uint64 * self.world)()
        recv, total = C.c_void_p(), C.c_size_t()
        _lib.check(_lib.lib().wfst_comm_allgatherv(self._h, blob, len(blob), rsizes, C.byref(recv), C.byref(total)),
                   "wfst_comm_allgatherv")
        try:
            raw = C.string_at(recv, total.value)
        finally:
            _lib.lib().wfst_bytes_destroy(recv)
        res, o = [], 0
        for r in range(self.world):
            chunk = raw[o:o + int(rsizes[r])]
            o += int(rsizes[r])
            n_r = int(np.frombuffer(chunk[:8], dtype=np.int64)[0])
            sz = np.frombuffer(chunk[8:8 + 8 * n_r], dtype=np.int64)
            items, p = [], 8 + 8 * n_r
            for s_ in sz:
                items.append(chunk[p:p + int(s_)])
                p += int(s_)
            res.append(items)
        return res


def interleave(gathered: np.ndarray, n_total: int) -> np.ndarray:
    """[world, n_local, rec] (rank r holds problems r, r+world, ...) -> [n_total, rec] in problem order."""
    world, n_local, rec = gathered.shape
    out = np.zeros((n_total, rec), dtype=gathered.dtype)
    for r in range(world):
        idx = shard_indices(n_total, r, world)
        out[idx] = gathered[r, :len(idx)]
    return out


def gather_fsts(local: Sequence[bytes], world: int, device=None) -> List[List[bytes]]:
    """All-gather results that are general FSTs (n-best trees, look-ahead compositions: BASELINE configs[4]) rather than
    linear paths: each rank passes its results serialised in the OpenFST binary format (`DeviceFst.to_bytes()` /
    `VectorFst.to_bytes()`), every rank gets `[rank][i] -> bytes`.  Two collectives: the byte counts, then the payloads
    padded to the largest rank (RCCL when `device` is a GPU, gloo on CPU).  Ranks may hold different numbers of results."""
    import torch
    import torch.distributed as dist

    sizes = np.array([len(b) for b in local], dtype=np.int64)
    kw = {} if device is None else {"device": device}
    n_local = torch.tensor([len(local), int(sizes.sum())], dtype=torch.int64, **kw)
    counts = torch.empty(world * 2, dtype=torch.int64, **kw)  # (concatenation layout: accepted by gloo and nccl/RCCL)
    dist.all_gather_into_tensor(counts, n_local)
    counts = counts.cpu().numpy().reshape(world, 2)
    max_n, max_bytes = int(counts[:, 0].max()), int(counts[:, 1].max())
    # one record per rank: [sizes (max_n x i64) | payload (max_bytes, padded)]
    rec = np.zeros(8 * max_n + max_bytes, dtype=np.uint8)
    rec[:8 * len(local)] = sizes.view(np.uint8)
    if len(local):
        rec[8 * max_n:8 * max_n + int(sizes.sum())] = np.frombuffer(b"".join(local), dtype=np.uint8)
    t = torch.from_numpy(rec)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * rec.shape[0], dtype=torch.uint8, **kw)
    dist.all_gather_into_tensor(out, t)
    out = out.cpu().numpy().reshape(world, rec.shape[0])
    res = []
    for r in range(world):
        n_r = int(counts[r, 0])
        sz = out[r, :8 * n_r].view(np.int64)
        off = 8 * max_n
        items = []
        for s in sz:
            items.append(out[r, off:off + int(s)].tobytes())
            off += int(s)
        res.append(items)
    return res


# ---- workload distribution: rank 0 builds the shared transducer and the acceptors ONCE, the other ranks receive them
_FLAT_KEYS = ("offsets", "arcs", "finals")


def broadcast_flat_fsts(flats, src: int = 0, device=None):
    """Broadcasts a list of flat FST dicts (n_states, start, offsets, arcs, finals, props) from rank `src` to every rank
    of the default group: one small header broadcast (shapes) and ONE payload broadcast of the concatenated arrays
    (RCCL when `device` is a GPU, gloo on CPU).  On rank `src` pass the list; elsewhere pass None.  Returns the list."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    kw = {} if device is None else {"device": device}
    if rank == src:
        hdr = []
        for f in flats:
            hdr += [int(f["n_states"]), -1 if f["start"] is None else int(f["start"]), int(f["props"]), len(f["arcs"])]
        hdr_t = torch.tensor([len(flats)] + hdr, dtype=torch.int64, **kw)
        n_hdr = torch.tensor([hdr_t.numel()], dtype=torch.int64, **kw)
    else:
        n_hdr = torch.zeros(1, dtype=torch.int64, **kw)
    dist.broadcast(n_hdr, src)
    if rank != src:
        hdr_t = torch.zeros(int(n_hdr.item()), dtype=torch.int64, **kw)
    dist.broadcast(hdr_t, src)
    hdr = hdr_t.cpu().numpy()
    n = int(hdr[0])
    meta = hdr[1:].reshape(n, 4)
    # payload: per FST offsets (n_states + 1 u32) | arcs (16 B each) | finals (n_states f32), as bytes
    sizes = [4 * (int(m[0]) + 1) + 16 * int(m[3]) + 4 * int(m[0]) for m in meta]
    total = int(sum(sizes))
    if rank == src:
        buf = np.empty(total, dtype=np.uint8)
        o = 0
        for f in flats:
            for k, dt in (("offsets", np.uint32), ("arcs", TR_DTYPE), ("finals", np.float32)):
                b = np.ascontiguousarray(f[k], dtype=dt).view(np.uint8).reshape(-1)
                buf[o:o + b.size] = b
                o += b.size
        payload = torch.from_numpy(buf)
        if device is not None:
            payload = payload.to(device)
    else:
        payload = torch.empty(total, dtype=torch.uint8, **kw)
    dist.broadcast(payload, src)
    if rank == src:
        return list(flats)
    raw = payload.cpu().numpy()
    out, o = [], 0
    for m in meta:
        ns, start, props, na = int(m[0]), int(m[1]), int(m[2]), int(m[3])
        offsets = raw[o:o + 4 * (ns + 1)].view(np.uint32).copy()
        o += 4 * (ns + 1)
        arcs = raw[o:o + 16 * na].view(TR_DTYPE).copy()
        o += 16 * na
        finals = raw[o:o + 4 * ns].view(np.float32).copy()
        o += 4 * ns
        out.append(dict(n_states=ns, start=None if start < 0 else start, offsets=offsets, arcs=arcs, finals=finals, props=props))
    return out
