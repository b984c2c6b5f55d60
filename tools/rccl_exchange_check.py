"""One rank of the multi-GPU exchange check (tests/test_distributed.py: test_rccl_multi_rank_exchange_matches_host_transport;
launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <this>).

Every rank composes ITS shard of one global batch on its own GPU (acceptor i -> rank i mod N, T replicated: SURVEY 8(e)) and
the results are exchanged three ways through the C-ABI's communicator:
  * over RCCL (wfst_comm_create: ncclCommInitRank with N ranks -> wfst_gather_paths_begin/_end, wfst_comm_allgather_*,
    wfst_comm_allgatherv, wfst_comm_order_after) — the path bench.py --gpus N takes;
  * over a host transport carried by a gloo group of the same processes (wfst_comm_create_host): the same exchange code of
    csrc/gather.cpp above another all-gather — what the CPU suite tests at world = 2.
Both must give the same records, in global order equal to a one-rank run of the whole batch (every rank computes that too).
Rank 0 prints "RCCL-N OK world=N" when every rank agrees."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import rustfst_amd
from rustfst_amd import dist as wdist, synth

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
gloo = dist.new_group(backend="gloo")
ok = dist.get_world_size() == world
ctx = rustfst_amd.Context(local)
n_total, L = 8 * world + 3, 30  # (ragged: the last ranks hold one problem less)
t = synth.make_transducer(20000, 8, 64, 0.0, seed=5)
flats = wdist.broadcast_flat_fsts([t] + synth.make_acceptors(t, n_total, L, seed0=7), 0, dev)
t, accs = flats[0], flats[1:]
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
max_arcs = L + 8
# the whole batch on this one GPU: what the sharded run must reproduce
whole, _ = rustfst_amd.compose_shortest_path_batch(rustfst_amd.DeviceFst.upload_many(accs, ctx), dt)
whole_packed = wdist.pack_device_paths(whole, max_arcs)
mine = wdist.shard_indices(n_total, rank, world)
n_local = (n_total + world - 1) // world
outs, _ = rustfst_amd.compose_shortest_path_batch(rustfst_amd.DeviceFst.upload_many([accs[i] for i in mine], ctx), dt)
packed = wdist.pack_device_paths(outs, max_arcs)
if packed.shape[0] < n_local:
    packed = np.concatenate([packed, np.zeros((n_local - packed.shape[0], packed.shape[1]), np.uint32)])

comm = wdist.Comm.from_torch_group(ctx, dev)  # RCCL, N ranks
ok &= (comm.rank, comm.world) == (rank, world)


def host_allgather(send, recv):
    tsend = torch.from_numpy(np.ascontiguousarray(send))
    out = torch.empty(world * max(tsend.numel(), 1), dtype=torch.uint8)[:world * tsend.numel()]
    if tsend.numel():
        dist.all_gather_into_tensor(out, tsend, group=gloo)
        recv[...] = out.numpy()


hcomm = wdist.Comm.from_host_transport(rank, world, host_allgather)
# 1. records of the sharded results: RCCL == host transport == the one-GPU run, in global order
comm.order_after(ctx)
comm.gather_records_begin(packed, max_arcs)
sp = dt.shortest_path()  # (something else runs on the context meanwhile)
g_rccl = comm.gather_paths_end()
hcomm.gather_records_begin(packed, max_arcs)
g_host = hcomm.gather_paths_end()
ok &= g_rccl.shape == (world,) + packed.shape and np.array_equal(g_rccl, g_host) and np.array_equal(g_rccl[rank], packed)
ok &= np.array_equal(wdist.interleave(g_rccl, n_total), whole_packed)
# ... and from the handles (wfst_gather_paths_begin packs them itself); ranks with a short shard pad with an empty path
if len(outs) == n_local:
    g2 = comm.gather_paths(outs, max_arcs)
else:
    comm.gather_records_begin(packed, max_arcs)
    g2 = comm.gather_paths_end()
ok &= np.array_equal(g2, g_rccl)
# 2. exchanges of growing and shrinking sizes back to back (the two staging sets alternate and grow)
for rep, words in enumerate((3, 5000, 7, 200000, 1)):
    blk = (np.arange(words, dtype=np.uint32) * (rank + 1) + rep).reshape(1, words)
    a, b = comm.allgather(blk), hcomm.allgather(blk)
    ok &= a.shape == (world, 1, words) and np.array_equal(a, b)
    for r in range(world):
        ok &= np.array_equal(a[r, 0], np.arange(words, dtype=np.uint32) * (r + 1) + rep)
# 3. ragged gather of whole FSTs (n-best trees / look-ahead results travel like this), empty contributions
blobs = [o.to_bytes() for o in outs[:1 + rank % 3]]
a, b = comm.gather_fsts(blobs), hcomm.gather_fsts(blobs)
ok &= a == b and a[rank] == blobs and len(a) == world
ok &= comm.gather_fsts([] if rank == 0 else [b"x" * 9]) == [[]] + [[b"x" * 9] for _ in range(world - 1)]
# 4. the bench's overlapped step under world = N: shortest_path(T) from this rank's own start state on one context, the fused
#    batch of this rank's shard on a second one, the previous step's results all-gathered over RCCL meanwhile (queued behind
#    what the first context holds: wfst_comm_order_after) — three steps in a row; what every exchange delivers must be the
#    host transport's and the one-GPU run's records, and the solve beside it must not change (its path is compared with a
#    solve that ran alone)
ctx2 = rustfst_amd.Context(local)
daccs2 = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many([accs[i] for i in mine], ctx2))
s1 = int((int(t["start"]) + rank * 104729) % int(t["n_states"]))
dts = rustfst_amd.DeviceFst.from_arrays(t["n_states"], s1, t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
alone = dts.shortest_path().to_flat()
pending, prev_outs = False, None
for it in range(4):
    sp_job = dts.shortest_path_begin()
    job = rustfst_amd.compose_shortest_path_batch_begin(daccs2, dt, ctx=ctx2)
    if prev_outs is not None:
        if pending:
            g = comm.gather_paths_end()
            pending = False
            ok &= np.array_equal(g, g_rccl)
        comm.order_after(ctx)
        if len(prev_outs) == n_local:
            comm.gather_paths_begin(prev_outs, max_arcs)
        else:
            comm.gather_records_begin(packed, max_arcs)
        pending = True
    prev_outs, _ = job.finish()
    spf = sp_job.finish().to_flat()
    ok &= np.array_equal(spf["arcs"], alone["arcs"]) and np.array_equal(spf["finals"], alone["finals"])
    ok &= np.array_equal(wdist.pack_device_paths(prev_outs, max_arcs), packed[:len(prev_outs)])
if pending:
    ok &= np.array_equal(comm.gather_paths_end(), g_rccl)
del daccs2, dts
flag = torch.tensor([1 if ok else 0], device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
del comm, hcomm
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print(f"RCCL-N {'OK' if int(flag.item()) == 1 else 'MISMATCH'} world={world}", flush=True)
sys.exit(0 if int(flag.item()) == 1 else 1)
