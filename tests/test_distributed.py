"""world_size-2 gloo test (CPU) of the N>1 path: acceptor sharding + result gather (rustfst_amd/dist.py).
The compute on each rank is done by the oracle here (no GPU in this container); on the GPU box the same
plumbing carries the HIP results (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import oracle_py
    from rustfst_amd import dist as wdist
    from rustfst_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = synth.make_transducer(500, 6, 16, 0.05, seed=13)
        accs = synth.make_acceptors(t, n_total, 12, seed0=1000)  # identical on every rank (T gets the same finals)
        ot = oracle_py.OracleFst.from_flat(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"])
        mine = wdist.shard_indices(n_total, rank, world)
        local = []
        for i in mine:
            a = accs[i]
            oa = oracle_py.OracleFst.from_flat(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"])
            local.append(oa.compose(ot).shortest_path_canonical().to_flat())
        n_local = (n_total + world - 1) // world
        packed = wdist.pack_paths(local, 12 + 8)
        if packed.shape[0] < n_local:  # ragged tail: pad so that every rank contributes the same shape
            packed = np.concatenate([packed, np.zeros((n_local - packed.shape[0], packed.shape[1]), np.uint32)])
        gathered = wdist.gather_paths(packed, world, None)
        allp = wdist.interleave(gathered, n_total)
        if rank == 0:
            # reference: all problems computed in one process
            ok = True
            for i, a in enumerate(accs):
                oa = oracle_py.OracleFst.from_flat(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"])
                exp = oa.compose(ot).shortest_path_canonical().to_flat()
                got = wdist.unpack_paths(allp[i:i + 1])[0]
                ok &= got["n_states"] == exp["n_states"] and got["start"] == exp["start"]
                ok &= np.array_equal(got["arcs"], exp["arcs"]) and np.array_equal(got["finals"], exp["finals"])
                ok &= np.array_equal(got["offsets"], exp["offsets"])
            q.put(("ok" if ok else "mismatch", int(sum(int(r[0]) for r in allp))))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 7])
def test_sharded_batch_gather_gloo_world2(n_total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    status, total_arcs = q.get(timeout=10)
    assert status == "ok"
    assert total_arcs > 0


def test_shard_indices_cover_everything():
    from rustfst_amd import dist as wdist
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 512):
            seen = sorted(i for r in range(world) for i in wdist.shard_indices(n, r, world))
            assert seen == list(range(n))


def test_pack_unpack_roundtrip():
    from rustfst_amd import dist as wdist
    from rustfst_amd._lib import TR_DTYPE
    arcs = np.array([(1, 2, 0.5, 0), (3, 4, 1.25, 1)], dtype=TR_DTYPE)
    p = dict(n_states=3, start=2, offsets=np.array([0, 0, 1, 2], np.uint32), arcs=arcs,
             finals=np.array([2.0, np.inf, np.inf], np.float32))
    empty = dict(n_states=0, start=None, offsets=np.zeros(1, np.uint32), arcs=np.zeros(0, TR_DTYPE),
                 finals=np.zeros(0, np.float32))
    packed = wdist.pack_paths([p, empty], 5)
    back = wdist.unpack_paths(packed)
    assert back[0]["n_states"] == 3 and back[0]["start"] == 2
    assert np.array_equal(back[0]["arcs"], arcs) and np.array_equal(back[0]["finals"], p["finals"])
    assert np.array_equal(back[0]["offsets"], p["offsets"])
    assert back[1]["n_states"] == 0


def _worker_fsts(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import oracle_py
    from rustfst_amd import dist as wdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def tree(k):  # a small non-linear FST (what an n-best search returns), different per (rank, k); serialised by the
            f = oracle_py.OracleFst()  # oracle here (no GPU in this container), by DeviceFst.to_bytes() on a GPU box
            for _ in range(3 + k):
                f.add_state()
            f.set_start(0)
            for s in range(1, 3 + k):
                f.add_tr(0, s, s + rank, 0.5 * s, s)
                f.set_final(s, float(rank))
            return f
        mine = [tree(k) for k in range(2 + rank)]  # ragged: rank 0 holds 2 results, rank 1 holds 3
        blobs = [f.store() for f in mine]
        got = wdist.gather_fsts(blobs, world, None)
        ok = len(got) == world and [len(g) for g in got] == [2 + r for r in range(world)]
        ok &= got[rank] == blobs
        other = 1 - rank
        back = oracle_py.OracleFst.load(got[other][0]).to_flat()
        ok &= back["n_states"] == 3 and int(back["arcs"]["olabel"][0]) == 1 + other
        ok &= wdist.gather_fsts([], world, None) == [[] for _ in range(world)]
        if rank == 0:
            q.put("ok" if ok else "mismatch")
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_gather_general_fsts_gloo_world2():
    """dist.gather_fsts: results that are general FSTs (n-best trees, look-ahead compositions) travel serialised in the
    OpenFST binary format; ragged per-rank counts, empty contribution."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fsts, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert q.get(timeout=10) == "ok"


def _worker_bcast(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from rustfst_amd import dist as wdist
    from rustfst_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = synth.make_transducer(300, 5, 16, 0.05, seed=21)
        accs = synth.make_acceptors(t, 5, 9, seed0=40)
        empty = dict(n_states=0, start=None, offsets=np.zeros(1, np.uint32), arcs=accs[0]["arcs"][:0], finals=np.zeros(0, np.float32), props=0)
        want = [t] + accs + [empty]
        got = wdist.broadcast_flat_fsts(want if rank == 0 else None, 0, None)
        ok = len(got) == len(want)
        for a, b in zip(got, want):
            ok &= a["n_states"] == b["n_states"] and a["start"] == b["start"] and int(a["props"]) == int(b["props"])
            ok &= np.array_equal(a["offsets"], b["offsets"]) and np.array_equal(a["arcs"], b["arcs"])
            ok &= np.array_equal(a["finals"].view(np.uint32), b["finals"].view(np.uint32))
        q.put((rank, "ok" if ok else "mismatch"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_broadcast_workload_gloo_world2():
    """dist.broadcast_flat_fsts: rank 0 builds the transducer and the acceptors once, the other rank receives identical
    arrays (one header + one payload broadcast), including an empty FST."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bcast, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=10) for _ in range(2)) == [(0, "ok"), (1, "ok")]


def _worker_cabi(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import oracle_py
    from rustfst_amd import dist as wdist
    from rustfst_amd import synth
    from rustfst_amd._lib import WfstError
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = wdist.Comm.from_torch_group_host()  # wfst_comm_create_host: the library's exchange code over gloo
        ok = (comm.rank, comm.world) == (rank, world)
        t = synth.make_transducer(400, 6, 16, 0.05, seed=17)
        accs = synth.make_acceptors(t, n_total, 10, seed0=300)
        ot = oracle_py.OracleFst.from_flat(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"])

        def solve(i):
            a = accs[i]
            oa = oracle_py.OracleFst.from_flat(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"])
            return oa.compose(ot).shortest_path_canonical().to_flat()

        mine = wdist.shard_indices(n_total, rank, world)
        n_local = (n_total + world - 1) // world
        max_arcs = 10 + 8
        packed = wdist.pack_paths([solve(i) for i in mine], max_arcs)
        if packed.shape[0] < n_local:
            packed = np.concatenate([packed, np.zeros((n_local - packed.shape[0], packed.shape[1]), np.uint32)])
        # 1. records -> wfst_gather_records_begin / wfst_gather_paths_end: rank-major layout, interleave restores global order
        comm.gather_records_begin(packed, max_arcs)
        try:  # one exchange in flight per communicator
            comm.gather_records_begin(packed, max_arcs)
            ok = False
        except WfstError:
            pass
        g = comm.gather_paths_end()
        ok &= g.shape == (world,) + packed.shape and np.array_equal(g[rank], packed)
        allp = wdist.unpack_paths(wdist.interleave(g, n_total))
        for i in range(n_total):
            exp = solve(i)
            ok &= allp[i]["n_states"] == exp["n_states"] and np.array_equal(allp[i]["arcs"], exp["arcs"])
            ok &= np.array_equal(allp[i]["finals"], exp["finals"])
        # 2. the two staging sets alternate and grow: exchanges of different sizes back to back, each rank's block in its slot
        for rep, words in enumerate((3, 5000, 7, 20000, 1)):
            blk = (np.arange(words, dtype=np.uint32) * (rank + 1) + rep).reshape(1, words)
            got = comm.allgather(blk)
            ok &= got.shape == (world, 1, words)
            for r in range(world):
                ok &= np.array_equal(got[r, 0], np.arange(words, dtype=np.uint32) * (r + 1) + rep)
        # 3. ragged: wfst_comm_allgatherv (sizes, then payloads padded to the largest rank), empty contributions
        blobs = [bytes([rank + 1]) * (100 * (k + 1) + 37 * rank) for k in range(2 + 3 * rank)]
        got = comm.gather_fsts(blobs)
        ok &= len(got) == world and got[rank] == blobs
        for r in range(world):
            ok &= got[r] == [bytes([r + 1]) * (100 * (k + 1) + 37 * r) for k in range(2 + 3 * r)]
        ok &= comm.gather_fsts([] if rank == 0 else [b"x" * 9]) == [[]] + [[b"x" * 9] for _ in range(world - 1)]
        ok &= comm.gather_fsts([]) == [[] for _ in range(world)]
        comm.order_after(None) if False else None  # (a host transport has no stream: nothing to order)
        q.put((rank, "ok" if ok else "mismatch"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 7])
def test_c_abi_exchange_over_gloo_world2(n_total):
    """The exchange code behind the C-ABI (csrc/gather.cpp: staging sets, rank-major records, `one exchange in flight`,
    the two-step ragged gather) at world = 2 without RCCL: wfst_comm_create_host runs it over a host all-gather (gloo
    here).  Records of sharded oracle results come back in global order and equal the one-process answers; exchanges of
    growing and shrinking sizes alternate between the two sets; ragged and empty contributions keep their rank slots."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cabi, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=10) for _ in range(2)) == [(0, "ok"), (1, "ok")]


@pytest.mark.gpu
def test_rccl_single_rank_gather_of_hip_results():
    """The N > 1 plumbing on the hardware that exists here: a 1-rank nccl (= RCCL) process group carries the results of the
    HIP batch path through the same calls bench.py --gpus N uses (broadcast of the workload, pack_device_paths, the
    asynchronous all-gather, interleave), and the gathered paths equal the local ones; then the same exchange through the
    C-ABI (wfst_comm_create / wfst_gather_paths_begin / _end / wfst_comm_allgatherv: librccl called by the library on its
    own stream), which is what bench.py --gpus N uses."""
    import subprocess
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import rustfst_amd
from rustfst_amd import dist as wdist, synth
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
ctx = rustfst_amd.Context(0)
t = synth.make_transducer(20000, 8, 64, 0.0, seed=5)
flats = wdist.broadcast_flat_fsts([t] + synth.make_acceptors(t, 6, 30, seed0=7), 0, dev)
t, accs = flats[0], flats[1:]
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
outs, _ = rustfst_amd.compose_shortest_path_batch(rustfst_amd.DeviceFst.upload_many(accs, ctx), dt)
packed = wdist.pack_device_paths(outs, 30 + 8)
g1 = wdist.gather_paths(packed, 1, dev)
g2 = wdist.gather_paths_async(packed, 1, dev).result()
assert g1.shape == (1,) + packed.shape and np.array_equal(g1, g2) and np.array_equal(g1[0], packed)
back = wdist.unpack_paths(wdist.interleave(g1, 6))
for o, b in zip(outs, back):
    f = o.to_flat()
    assert f["n_states"] == b["n_states"] == 31 and np.array_equal(f["arcs"], b["arcs"]) and f["finals"][0] == b["finals"][0]
blobs = [o.to_bytes() for o in outs[:2]]
assert wdist.gather_fsts(blobs, 1, dev) == [blobs]
# the same exchange through the C-ABI (wfst_comm_* / wfst_gather_paths_*): the library's own RCCL communicator, no torch
# call between the batch result and the gathered records
comm = wdist.Comm.from_torch_group(ctx, dev)
assert (comm.rank, comm.world) == (0, 1)
g3 = comm.gather_paths(outs, 30 + 8)
assert g3.shape == g1.shape and np.array_equal(g3, g1)
comm.order_after(ctx)  # (the exchange queued next starts after what the context has queued so far)
comm.gather_paths_begin(outs, 30 + 8)  # asynchronous form: something else runs meanwhile
sp = dt.shortest_path()
assert np.array_equal(comm.gather_paths_end(), g1) and sp.num_states > 0
blk = np.arange(12, dtype=np.uint32).reshape(3, 4)
assert np.array_equal(comm.allgather(blk), blk[None])
assert comm.gather_fsts(blobs) == [blobs] and comm.gather_fsts([]) == [[]]
import time
best = [1e9, 1e9]
for _ in range(20):
    c0 = time.perf_counter(); comm.gather_paths_begin(outs, 38); c1 = time.perf_counter(); comm.gather_paths_end(); c2 = time.perf_counter()
    best = [min(best[0], c1 - c0), min(best[1], c2 - c0)]
print("C-ABI gather: begin %%.1f us, begin+end %%.1f us" %% (best[0] * 1e6, best[1] * 1e6))
del comm
dist.barrier(); dist.destroy_process_group()
print("RCCL-1 OK")
''' % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL-1 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_rccl_multi_rank_exchange_matches_host_transport():
    """The RCCL branch of the C-ABI exchange with MORE THAN ONE rank (csrc/gather.cpp: ncclCommInitRank(world > 1),
    ncclAllGather behind wfst_gather_paths_* / wfst_comm_allgather_* / wfst_comm_allgatherv, wfst_comm_order_after): one process
    per visible GPU (at most 8), every rank composing its shard of one global batch on its own device.  The gathered records
    equal those of the host transport (the same exchange code over a gloo all-gather of the same processes: what the CPU
    suite runs at world = 2) and, interleaved, a one-GPU run of the whole batch; ragged and resized exchanges included.
    Skips on a box with one GPU (the builder's): it runs the day the suite meets a multi-GPU node."""
    import subprocess
    torch = pytest.importorskip("torch")
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs at least 2 GPUs (visible: {n})")
    world = min(n, 8)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "rccl_exchange_check.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and f"RCCL-N OK world={world}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_rccl_exchange_check_script_runs_at_world_1():
    """The same script at one rank (the hardware that exists here): keeps tools/rccl_exchange_check.py itself green, so that
    the multi-rank test above cannot fail on the day it first runs for a reason that has nothing to do with RCCL."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "rccl_exchange_check.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "RCCL-N OK world=1" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
