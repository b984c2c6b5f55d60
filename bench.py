#!/usr/bin/env python
"""bench.py — headline benchmark of the compose -> shortest_path hot path on MI355X.

Workload (BASELINE.json configs[2] + configs[3]): one shared synthetic transducer
T(1M states, 10M arcs, fan-out 10, |Sigma| = 256, weights on the 1/512 grid, seed 3), HBM-resident.
One "step" on every rank =
  (S1) shortest_path(T)                      — the 10M-arc frontier relaxation (the roofline kernel)
  (S2) for each of this rank's B linear acceptors (random walks of length 200 in T):
       shortest_path(compose(A_i, T))        — the fused wave-per-problem pipeline
  (N > 1 only) all-gather of the B result paths over RCCL, queued asynchronously and collected one step later
  (the last one before the closing barrier), so it overlaps with the next step's compute.
S1 and S2 are independent requests: S2 is enqueued first, asynchronously, on a second context / HIP stream
(its single long kernel uses one wave per acceptor), S1 then runs on the first stream and overlaps with it,
and S2's results are collected last (--serial runs them back to back on one stream instead).
Acceptor i of the global batch (B x N acceptors) lives on rank i mod N (weak scaling: per-GPU work is
fixed); T is replicated; no collective during compute.  At N > 1 the S1 query of rank r is the single-source solve from
rank r's own start state of T (distinct queries on the replicated T, never N copies of one solve); `batch_only` in the
line carries the sharded batch's own rate (acceptors/s, us per compose->shortest_path), the figure that scales.
value = arcs/s over the whole job, arcs per rank-step = E(T) [each arc of T relaxed at least once]
        + E_composed (arcs emitted by compose before trim) + E_relaxed on the composed FSTs (== E_composed).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 3
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def kernel_sources_sha256():
    """hash of the relaxation sources: profiles/pmc_relax_traffic.json records it, so that a PMC figure taken at another
    state of the kernel is flagged stale in the bench line"""
    import hashlib
    h = hashlib.sha256()
    for name in ("sssp.hip", "sssp_mailbox.h", "sssp_resident.h", "sssp_binned.h", "api.cpp"):  # (api.cpp: the device pool — block reuse decides what a solve finds in the Infinity Cache)
        with open(os.path.join(ROOT, "rustfst_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


SWEEP_MAX = 4096  # largest batch of the batch_sweep extra
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (default: ~0.5 s of steps)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--states", type=int, default=1_000_000)
    ap.add_argument("--fanout", type=int, default=10)
    ap.add_argument("--sigma", type=int, default=256)
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--acc-len", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config5-states", type=int, default=5_000_000,
                    help="states of the HCLG-shaped operand of the configs[4] extra (0 = skip it)")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--roofline-sizes", default="1000000,2000000,5000000,16000000",
                    help="states of the extra `roofline_vs_size` solves (same generator; empty = off)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra legs (first-query times, the single-string case of configs[1], the "
                         "all-cores CPU batch leg)")
    ap.add_argument("--serial", action="store_true",
                    help="run S1 then S2 of a step on ONE stream.  Default: S2's batch is enqueued asynchronously on a "
                         "second context / HIP stream (wfst_compose_shortest_path_batch_begin), S1 runs on the first, "
                         "then the batch is collected — the two independent requests overlap on the GPU.")
    ap.add_argument("--batch-cus", type=int, default=0,
                    help="compute units reserved for the batch context when the two requests overlap (the other "
                         "context gets the rest; hipExtStreamCreateWithCUMask).  0 = no partitioning (default: with the "
                         "string o T kernel the batch is 0.19 ms and partitioning only takes CUs from shortest_path; it "
                         "paid, 1.04 -> 0.88 ms, while the batch ran on the general 0.7 ms kernel).")
    ap.add_argument("--order", choices=["s2-first", "s1-first"], default="s1-first",
                    help="which request of a step is enqueued first when they overlap.  s1-first (default): the "
                         "relaxation is the longer chain of the two now that a batch's results cost the host ~6 us "
                         "(round 4), and its resident launch leaves the batch kernel the compute units it needs.  "
                         "s2-first was the order of rounds 2-3, when every sweep was a GPU-wide launch.")
    args = ap.parse_args()
    args.overlap = not args.serial
    return args


def host_cpu_info():
    """which CPU the single-threaded baseline ran on (core count AND model): /proc/cpuinfo model name, the
    scaling governor, the CPU this thread is on right now and that CPU's clock, the number of NUMA nodes"""
    info = {"cpu_model": None, "governor": None, "cpu_of_thread": None, "mhz": None, "numa_nodes": None}
    try:
        cpu = None
        if hasattr(os, "sched_getcpu"):
            cpu = os.sched_getcpu()
        else:
            with open("/proc/self/stat") as fh:
                cpu = int(fh.read().rsplit(")", 1)[1].split()[36])
        info["cpu_of_thread"] = cpu
        cur, model, mhz = None, None, None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "processor":
                    cur = int(v)
                elif k == "model name" and (model is None or cur == cpu):
                    model = v
                elif k == "cpu MHz" and cur == cpu:
                    mhz = float(v)
        info["cpu_model"], info["mhz"] = model, mhz
    except Exception:  # noqa: BLE001  (a container without /proc/cpuinfo: the fields stay null)
        pass
    try:
        with open(f"/sys/devices/system/cpu/cpu{info['cpu_of_thread'] or 0}/cpufreq/scaling_governor") as fh:
            info["governor"] = fh.read().strip()
    except Exception:  # noqa: BLE001
        info["governor"] = "unavailable"
    try:
        info["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except Exception:  # noqa: BLE001
        pass
    return info


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def config5_extra(n_states, ctx, device):
    """BASELINE configs[4] as an untimed extra of the bench line: an HCLG-shaped FST (fan-out 10, 5 % epsilon arcs) as the
    look-ahead operand, 64 linear acceptors (len 200) composed with the look-ahead filter stack in one batch, n = 10
    shortest paths of every result; the CPU restatement beside it on a smaller operand (the largest it finishes in a few
    seconds).  No oracle at full size: the 10 best weights of the first results are checked against the plain composition."""
    import numpy as np
    import torch
    import rustfst_amd
    from rustfst_amd import ShortestPathConfig, ComposeConfig, synth
    res = {"workload": f"configs[4]: HCLG-shaped FST ({n_states} states, fan-out 10, 5 % output epsilons) as look-ahead operand, "
                       "64 linear acceptors (len 200), look-ahead composition (batch of 64) + n = 10 shortest paths each"}

    def build(n, n_acc, seed):
        t5 = synth.make_transducer(n, 10, 256, 0.05, seed=seed)
        accs = synth.make_acceptors(t5, n_acc, 200, seed0=77)
        arcs = t5["arcs"].copy()  # the look-ahead operand emits what the acceptors read: epsilons to the output side
        arcs["ilabel"], arcs["olabel"] = t5["arcs"]["olabel"].copy(), t5["arcs"]["ilabel"].copy()
        t1 = dict(t5)
        t1["arcs"], t1["props"] = arcs, synth.O_LABEL_SORTED
        return t1, accs

    c0 = time.perf_counter()
    t1, accs = build(n_states, 64, 9)
    res["arcs"] = int(t1["offsets"][-1])
    res["generate_s"] = round(time.perf_counter() - c0, 2)
    d1 = rustfst_amd.DeviceFst.from_arrays(t1["n_states"], t1["start"], t1["offsets"], t1["arcs"], t1["finals"], t1["props"], ctx)
    torch.cuda.synchronize(device)
    c0 = time.perf_counter()
    la = rustfst_amd.LookAhead(d1)
    res["lookahead_create_s"] = round(time.perf_counter() - c0, 3)
    das = rustfst_amd.DeviceFst.upload_many(accs, ctx)
    c0 = time.perf_counter()
    rel = [la.relabel(d) for d in das]
    res["relabel_64_ms"] = round(1e3 * (time.perf_counter() - c0), 3)
    outs = la.compose_batch(rel)  # warm-up (pool growth)
    best = float("inf")
    for _ in range(3):
        ctx.synchronize()
        c0 = time.perf_counter()
        outs = la.compose_batch(rel)
        ctx.synchronize()
        best = min(best, time.perf_counter() - c0)
    res["lookahead_compose_batch64_ms"] = round(1e3 * best, 3)
    res["composed_states_min_max"] = [int(min(o.num_states for o in outs)), int(max(o.num_states for o in outs))]
    cfg10 = ShortestPathConfig(nshortest=10)
    nb = rustfst_amd.shortest_path_batch(outs, cfg10, ctx=ctx)  # warm-up
    best = float("inf")
    for _ in range(3):
        ctx.synchronize()
        c0 = time.perf_counter()
        nb = rustfst_amd.shortest_path_batch(outs, cfg10, ctx=ctx)
        best = min(best, time.perf_counter() - c0)
    res["nbest10_x64_ms"] = round(1e3 * best, 3)
    res["nbest_path"] = rustfst_amd.last_nbest_path(ctx)
    cfg1 = ShortestPathConfig(nshortest=1)
    rustfst_amd.shortest_path_batch(outs, cfg1, ctx=ctx)  # warm-up
    best = float("inf")
    for _ in range(3):
        ctx.synchronize()
        c0 = time.perf_counter()
        rustfst_amd.shortest_path_batch(outs, cfg1, ctx=ctx)
        best = min(best, time.perf_counter() - c0)
    res["onebest_x64_ms"] = round(1e3 * best, 3)  # (nshortest = 1 of the same 64 lattices: one launch, one wavefront each)

    def weights(f):  # total weights of the paths of an n-best tree, sorted
        f = f.to_flat()
        if f["n_states"] == 0:
            return []
        off, arcs_, fin, out, stack = f["offsets"], f["arcs"], f["finals"], [], [(f["start"], 0.0)]
        while stack:
            s_, w_ = stack.pop()
            if np.isfinite(fin[s_]):
                out.append(round((w_ + float(fin[s_])) * 512))
            for k in range(off[s_], off[s_ + 1]):
                stack.append((int(arcs_[k]["nextstate"]), w_ + float(arcs_[k]["weight"])))
        return sorted(out)
    ok = True
    for i in range(2):  # same weighted relation as the plain composition of the same pair
        plain = d1.compose(das[i], ComposeConfig(connect=True))
        ok = ok and weights(nb[i]) == weights(plain.shortest_path(cfg10)) and len(weights(nb[i])) >= 1
    res["nbest_weights_match_plain_composition"] = bool(ok)
    del la, d1, outs, nb
    # the CPU restatement (1 core) on the SAME operand (full size): precompute, one composition, its n = 10
    from oracle import oracle_py
    o1 = oracle_py.OracleFst.from_flat(t1["n_states"], t1["start"], t1["offsets"], t1["arcs"], t1["finals"], t1["props"])
    a0 = accs[0]
    oa0 = oracle_py.OracleFst.from_flat(a0["n_states"], a0["start"], a0["offsets"], a0["arcs"], a0["finals"], a0["props"])
    c0 = time.perf_counter()
    oc = o1.compose_lookahead(oa0)
    t_first = time.perf_counter() - c0
    c0 = time.perf_counter()
    onb = oc.shortest_path_n(10)
    t_nb = time.perf_counter() - c0
    res["cpu"] = {"states": int(t1["n_states"]), "cores": 1, "kind": "port",
                  "lookahead_precompute_plus_one_composition_s": round(t_first, 3), "nbest10_ms": round(1e3 * t_nb, 3),
                  "note": "oracle restatement on the same operand: MatcherFst::new (label reachability + relabelling) is redone per "
                          "composition, as in rustfst-cli; one acceptor"}
    del onb, oc, o1, t1

    # ---- the wide look-ahead driver at scale (with |Sigma| = 256 a composed lattice has ~250 states: the figures above are
    # latencies of tiny problems).  |Sigma| = 8: every acceptor label matches ~1.25 arcs per state, the lattice fills the
    # operand within a few levels — tens of millions of composed states from ONE acceptor of 40 labels.
    import math
    W_SIGMA, W_LEN = 8, 40
    c0 = time.perf_counter()
    tw = synth.make_transducer(n_states, 10, W_SIGMA, 0.05, seed=9)
    aw = synth.make_acceptors(tw, 1, W_LEN, seed0=77)[0]
    tw["arcs"]["ilabel"], tw["arcs"]["olabel"] = tw["arcs"]["olabel"].copy(), tw["arcs"]["ilabel"].copy()
    tw["props"] = synth.O_LABEL_SORTED
    gen_w = time.perf_counter() - c0
    dw = rustfst_amd.DeviceFst.from_arrays(tw["n_states"], tw["start"], tw["offsets"], tw["arcs"], tw["finals"], tw["props"], ctx)
    del tw
    c0 = time.perf_counter()
    law = rustfst_amd.LookAhead(dw)
    create_w = time.perf_counter() - c0
    daw = rustfst_amd.DeviceFst.from_arrays(aw["n_states"], aw["start"], aw["offsets"], aw["arcs"], aw["finals"], aw["props"], ctx)
    relw = law.relabel(daw)
    # first call: the arena grows into the result (eight growths, each a multi-GB allocation on a fresh pool); the second
    # starts from the first's size (a fresh allocation of that size once); from the third on the pool holds it
    wide_ms = []
    for _ in range(5):
        ctx.synchronize()
        c0 = time.perf_counter()
        ow = law.compose(relw)
        ctx.synchronize()
        wide_ms.append(1e3 * (time.perf_counter() - c0))
        if len(wide_ms) < 5:
            del ow
    best = min(wide_ms[2:]) * 1e-3
    stw = ctx.stats()
    cs, ca = int(stw["compose_states"]), int(stw["compose_arcs"])
    m = ca / max(1, cs)
    per_state = 24 + 16 * 1.0 + 4 * 1.0 * math.ceil(math.log2(10 + 1)) + 48 * m  # B_comp per expanded state
    res["wide_lookahead"] = {
        "workload": f"look-ahead composition of ONE acceptor of {W_LEN} labels with the same generator at |Sigma| = {W_SIGMA} "
                    f"({n_states} states): the wide look-ahead driver", "generate_s": round(gen_w, 2), "lookahead_create_s": round(create_w, 3),
        "composed_states": cs, "composed_arcs": ca, "result_states": int(ow.num_states), "ms": round(1e3 * best, 2),
        "first_call_ms": round(wide_ms[0], 2), "second_call_ms": round(wide_ms[1], 2), "repeated_calls_ms": [round(v, 2) for v in wide_ms[2:]],
        "spread": round((max(wide_ms[2:]) - min(wide_ms[2:])) / min(wide_ms[2:]), 4),
        "states_per_s": round(cs / best, 1), "arcs_per_s": round(ca / best, 1),
        "roofline_compose": {"bound": "hbm", "accounting": "SURVEY 8(d): B_comp = sum over expanded states of 24 + 16 f1 + 4 f1 ceil(log2(f2 + 1)) + 48 m "
                             "(f1 = 1, f2 = 10, m = arcs / states) / host clock around the synchronous call, best of the 3rd..5th (an upper bound of the kernel time; kernel table: profiles/r05c_wide_lookahead.md)",
                             "algorithmic_bytes": int(per_state * cs), "achieved": round(per_state * cs / best / 1e9, 2), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(per_state * cs / best / 1e9 / HBM_PEAK_GBS, 5)}}
    del ow, law, dw
    return res


def roofline_vs_size(ctx, sizes, fanout, sigma):
    """The relaxation's fraction of the HBM roofline (SURVEY 8(d) accounting: (20 E + 12 N) bytes per solve / summed kernel
    time / 8 TB/s) on the SAME generator at several sizes: where the per-level floor of a 1M-state solve is amortised,
    and which kernel the library picks there."""
    import numpy as np
    import rustfst_amd
    from rustfst_amd import synth
    names = ("sssp_relax_kernel", "sssp_mbox_kernel", "sssp_mbox_resident_kernel", "sssp_relax_kernel + sssp_bin_expand/apply_kernel (per level)")
    rows = []
    for n in sizes:
        c0 = time.perf_counter()
        t = synth.make_transducer(n, fanout, sigma, 0.0, seed=3)
        gen_s = time.perf_counter() - c0
        e = int(t["offsets"][-1])
        d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
        del t
        for _ in range(5):  # plan, transpose, prediction settle
            d.shortest_path()
        ctx.set_profiling(2)
        chain, host = [], []
        for _ in range(9):
            c0 = time.perf_counter()
            d.shortest_path()
            host.append(1e3 * (time.perf_counter() - c0))
            cs = ctx.stats()
            if cs["relax_launches"]:
                chain.append((cs["relax_ms"], int(cs["relax_launches"])))
        ctx.set_profiling(0)
        st = ctx.stats()
        row = {"states": n, "arcs": e, "kernel": names[int(st["relax_kernel"])], "ms_shortest_path": round(min(host), 4),
               "generate_s": round(gen_s, 1)}
        if chain:
            chain.sort()
            ms, launches = chain[len(chain) // 2]
            by = 20.0 * e + 12.0 * n
            row.update({"launches": launches, "relax_kernel_ms": round(ms, 4), "algorithmic_bytes": int(by),
                        "achieved_GBps": round(by / (ms * 1e-3) / 1e9, 1), "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
        rows.append(row)
        del d
    return {"accounting": "SURVEY 8(d): (20 E + 12 N) bytes per solve / summed relaxation-kernel time of a repeated query (HIP events "
                          "around the pre-queued launch chain, median of 9) / 8 TB/s", "generator": f"T(N, fan-out {fanout}, |Sigma| = {sigma}, seed 3)",
            "points": rows}


def main():
    args = parse_args()
    import numpy as np
    import torch

    import rustfst_amd
    from rustfst_amd import dist as wdist
    from rustfst_amd import synth

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # WFST_BENCH_FORCE_DIST=1 exercises the RCCL plumbing with a single rank (1-GPU boxes)
    force_dist = os.environ.get("WFST_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        if int(os.environ.get("WORLD_SIZE", "1")) != world:
            raise SystemExit("--gpus N must match WORLD_SIZE (launch with torch.distributed.run)")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    stream = torch.cuda.Stream(device=device, priority=int(os.environ.get("WFST_BENCH_PRIO1", "0")))
    if args.overlap and args.batch_cus > 0:
        # Two contexts on DISJOINT compute units: the batch kernel is 64 lone waves chasing dependent loads, and a
        # load issued from a CU that also hosts streaming relaxation waves waits 2-4x longer in that CU's memory
        # queue (tools/cu_mask_probe.py: 1.08 -> 0.90 ms per step).  CU i = bit i%32 of word i//32.
        n_cus = torch.cuda.get_device_properties(device).multi_processor_count
        k = min(args.batch_cus, n_cus - 1)
        small = np.zeros((n_cus + 31) // 32, dtype=np.uint32)
        for cu in range(k):
            small[cu // 32] |= np.uint32(1 << (cu % 32))
        big = np.zeros_like(small)
        for cu in range(k, n_cus):
            big[cu // 32] |= np.uint32(1 << (cu % 32))
        ctx = rustfst_amd.Context(local_rank, cu_mask=big)
        ctx2 = rustfst_amd.Context(local_rank, cu_mask=small)
    else:
        ctx = rustfst_amd.Context(local_rank, stream=stream.cuda_stream)
        # second context (own HIP stream + pools) for the batch pipeline: S1 and S2 are independent requests
        stream2 = torch.cuda.Stream(device=device, priority=int(os.environ.get("WFST_BENCH_PRIO2", "-1")))
        ctx2 = ctx if (not args.overlap) else rustfst_amd.Context(local_rank, stream=stream2.cuda_stream)
    rustfst_amd.set_default_context(ctx)

    # ------------------------------------------------------------------ synthetic workload (identical on all ranks)
    t0 = time.time()
    n_total = args.batch_per_gpu * world
    mine = wdist.shard_indices(n_total, rank, world)
    if world > 1 or force_dist:
        # rank 0 builds T and the global batch of acceptors once; the other ranks receive the bytes over RCCL (one payload
        # broadcast) and keep T plus their own shard
        flats = None
        if rank == 0:
            t = synth.make_transducer(args.states, args.fanout, args.sigma, 0.0, seed=3)
            flats = [t] + synth.make_acceptors(t, n_total, args.acc_len, seed0=1000)
            if not args.no_extras:  # (before T leaves rank 0: their end states are final states of T everywhere)
                sweep_accs = synth.make_acceptors(t, SWEEP_MAX, args.acc_len, seed0=50_000)
            if not args.no_extras:  # (every rank takes its share of them in the multi-GPU extras: batch_strong / batch_weak_4096)
                flats = flats + sweep_accs
        flats = wdist.broadcast_flat_fsts(flats, 0, device)
        t, accs_all = flats[0], flats[1:1 + n_total]
        if not args.no_extras:
            sweep_accs = flats[1 + n_total:]
    else:
        t = synth.make_transducer(args.states, args.fanout, args.sigma, 0.0, seed=3)
        accs_all = synth.make_acceptors(t, n_total, args.acc_len, seed0=1000)
        if not args.no_extras:  # the acceptors of the batch_sweep extra: generated BEFORE T is uploaded (make_acceptors
            # marks the walks' end states final in T: the device copy and the CPU baseline must see the same T)
            sweep_accs = synth.make_acceptors(t, SWEEP_MAX, args.acc_len, seed0=50_000)
    # Several GPUs: every rank holds T.  The unit that scales is the (acceptor, T) problem; the direct shortest_path(T) of a
    # step is a DIFFERENT query on every rank — the single-source solve from rank r's own start state (T is strongly
    # connected: every start reaches everything) — so the arcs summed over the ranks are arcs of distinct work, never N
    # copies of one solve.  Rank 0's query is the 1-GPU one.
    s1_start = int(t["start"]) if world == 1 else int((int(t["start"]) + rank * 104729) % int(t["n_states"]))
    dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], s1_start, t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
    daccs = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many([accs_all[i] for i in mine], ctx2))
    # T is read-only for both pipelines: one HBM copy serves both contexts — except where this rank's S1 query starts
    # elsewhere (the acceptors are walks from T's own start state: the compositions need T as it is)
    dt2 = dt if s1_start == int(t["start"]) else rustfst_amd.DeviceFst.from_arrays(
        t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
    e_t = int(t["offsets"][-1])
    gen_s = time.time() - t0

    # ------------------------------------------------------------------ cold queries (untimed setup, reported)
    # a FRESH handle of T, HBM-resident: the first shortest_path builds the mailbox region plan and takes the parent
    # pass; the second builds the transpose for the backtrace; from the third on the solve is one predicted batch
    cold = None
    if rank == 0 and not args.no_extras:
        dcold = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
        torch.cuda.synchronize(device)
        cold = []
        for _ in range(4):
            c0 = time.perf_counter()
            dcold.shortest_path()
            cold.append(round(1e3 * (time.perf_counter() - c0), 4))
        del dcold
        # the first figure above is the first shortest_path of the PROCESS (code objects loaded, pools grown); what a fresh
        # handle costs in a warm process is measured on a second one
        dcold = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
        torch.cuda.synchronize(device)
        cold2 = []
        for _ in range(3):
            c0 = time.perf_counter()
            dcold.shortest_path()
            cold2.append(round(1e3 * (time.perf_counter() - c0), 4))
        del dcold

    # ------------------------------------------------------------------ the reference harness's split (untimed setup, reported)
    # rustfst-cli --bench (rustfst-cli/src/binary_fst_algorithm.rs:88-214) times, per iteration, parsing the input file,
    # the algorithm, and serialising the result, and reports mean +- sigma of each; the same three legs for
    # shortest_path on T through this engine: OpenFST vector file -> host parse + upload to HBM, the (cold: fresh handle)
    # solve, result path -> OpenFST bytes -> file
    harness = None
    if rank == 0 and not args.no_extras:
        import tempfile
        f_in = os.path.join(tempfile.gettempdir(), f"wfst_bench_T_{os.getpid()}.fst")
        f_out = f_in + ".out"
        with open(f_in, "wb") as fh:
            fh.write(dt.to_bytes())
        legs = {"parsing": [], "algorithm": [], "serialization": [], "cli": []}
        H_WARM, H_ITERS = 3, 10  # (the reference harness's own protocol: warm-ups 3, iterations 10)
        for i in range(H_WARM + H_ITERS):
            h0 = time.perf_counter()
            with open(f_in, "rb") as fh:
                dparsed = rustfst_amd.DeviceFst.from_bytes(fh.read(), ctx)
            torch.cuda.synchronize(device)
            h1 = time.perf_counter()
            hres = dparsed.shortest_path()
            h2 = time.perf_counter()
            with open(f_out, "wb") as fh:
                fh.write(hres.to_bytes())
            h3 = time.perf_counter()
            if i >= H_WARM:
                for name, v in (("parsing", h1 - h0), ("algorithm", h2 - h1), ("serialization", h3 - h2), ("cli", h3 - h0)):
                    legs[name].append(1e3 * v)
            del dparsed, hres
        harness = {"algorithm": "shortest_path", "input_file_bytes": os.path.getsize(f_in), "warmups": H_WARM, "iterations": H_ITERS,
                   "note": "rustfst-cli --bench legs (binary_fst_algorithm.rs:88-214): parse = read + host parse + upload to HBM; "
                           "algorithm = cold solve on the fresh handle; serialization = path FST -> OpenFST bytes -> file"}
        for name, v in legs.items():
            harness[name + "_ms"] = {"mean": round(float(np.mean(v)), 3), "std": round(float(np.std(v)), 3)}
        os.remove(f_in)
        os.remove(f_out)

    last = {}
    chain_probe = {}

    def chain_us(label, n=15):
        """the relaxation chain of a repeated shortest_path(T) on the bench's handle by HIP events (profiling mode 2), median of n;
        recorded under `label` (roofline.chain_us_by_point: the same bracket at several points of the run)"""
        if rank != 0:
            return None
        ctx.set_profiling(2)
        v = []
        for _ in range(n):
            dt.shortest_path()
            cs = ctx.stats()
            if cs["relax_launches"]:
                v.append(1e3 * cs["relax_ms"])
        ctx.set_profiling(0)
        v.sort()
        chain_probe[label] = round(v[len(v) // 2], 1) if v else None
        return chain_probe[label]
    batch_arcs = [0]  # composed + relaxed arcs of the batch legs only (what scales with the number of GPUs)
    phases = [0.0, 0.0, 0.0, 0.0, 0]  # host seconds in: first begin, second begin, batch finish, shortest_path finish; steps

    # the result exchange runs inside libwfst_amd (wfst_comm_* / wfst_gather_paths_*: its own RCCL communicator, stream and
    # pinned staging); torch.distributed only carries the rendezvous id, the workload broadcast and the closing barrier
    comm = wdist.Comm.from_torch_group(ctx, device) if (world > 1 or force_dist) else None

    def exchange(outs=None):
        # RCCL all-gather of a step's batch results on the communicator's own stream.  The host side of it (collecting the
        # previous exchange, packing 64 paths, the events, the RCCL launch: ~90 us) sits where the host only waits: a
        # step's results are handed over right after the NEXT step's two requests are enqueued, and collected one step
        # after that; the last ones are drained before the closing barrier, inside the timed region.
        if last.get("pending"):
            last["gathered"] = comm.gather_paths_end()
            last["pending"] = False
        if outs is not None:
            # behind the relaxation that is queued on ctx's stream right now: its resident launch needs (nearly) every
            # compute unit, the all-gather kernel runs beside the one-workgroup head of the solve after it instead
            comm.order_after(ctx)
            comm.gather_paths_begin(outs, args.acc_len + 8)
            last["pending"] = True

    exchange_on = [True]  # (the multi-GPU extras time the same steps without the result exchange: what it costs per step)

    def step():
        if not args.overlap:
            sp = dt.shortest_path()
            outs, n_arcs = rustfst_amd.compose_shortest_path_batch(daccs, dt2)
        else:
            # Order of the two requests (--order).  Rounds 2-3 queued S2 first: its one long, narrow kernel (one wave per
            # acceptor) had to be in flight BEFORE a chain of GPU-wide relaxation sweeps, or the hardware ran the chain
            # to its end first (tools/ubench_concurrency.hip: 623 us overlapped vs 904 us serialised).  The resident
            # relaxation launch (one workgroup on 245 of 256 compute units for the whole WIDE phase) leaves the batch its
            # compute units whenever it arrives, and S1 is the longer of the two: it goes first (measured: 0.330 -> 0.323 ms).
            p0 = time.perf_counter()
            if args.order == "s2-first":
                job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt2, ctx=ctx2)
                p1 = time.perf_counter()
                # S1 asynchronously too: the host builds the batch's 64 result FSTs while the sweeps are still running
                sp_job = dt.shortest_path_begin()
            else:
                sp_job = dt.shortest_path_begin()
                p1 = time.perf_counter()
                job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt2, ctx=ctx2)
            p2 = time.perf_counter()
            if (world > 1 or force_dist) and exchange_on[0] and last.get("to_send") is not None:
                exchange(last["to_send"])  # the previous step's results (both requests of this step are running)
                last["to_send"] = None
            outs, n_arcs = job.finish()
            p3 = time.perf_counter()
            if (world > 1 or force_dist) and exchange_on[0]:
                last["to_send"] = outs
            sp = sp_job.finish()
            p4 = time.perf_counter()
            phases[:] = [phases[0] + p1 - p0, phases[1] + p2 - p1, phases[2] + p3 - p2, phases[3] + p4 - p3, phases[4] + 1]
        last["sp"], last["outs"], last["n_arcs"] = sp, outs, n_arcs
        if (world > 1 or force_dist) and not args.overlap and exchange_on[0]:
            exchange(outs)
        batch_arcs[0] += 2 * n_arcs
        return e_t + 2 * n_arcs

    def drain():
        if world > 1 or force_dist:
            if last.get("to_send") is not None:
                exchange(last["to_send"])  # the last step's results
                last["to_send"] = None
            exchange()  # ... collected

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1 or force_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(device)

    with torch.cuda.stream(stream):
        # Setup, part 2 (untimed, like the upload of T): three priming steps.  The engine keeps derived data per
        # resident FST that it only builds once an FST is queried AGAIN (the transpose used by the backtrace on the
        # 2nd query, the sweep graph sized to the previous solve on the 3rd); a serving process reaches that state
        # within its first requests, and the --warmup steps that follow are then warm-up only.
        PRIMING_STEPS = 3
        t1 = time.time()
        for _ in range(PRIMING_STEPS):
            step()
        drain()
        gather_check = None
        if world > 1 or force_dist:
            # (untimed) what the exchange delivered for the last priming step must be a ONE-rank run of the whole global batch:
            # every rank composes all n_total acceptors on its own GPU once and compares, record for record, in global order
            whole, _ = rustfst_amd.compose_shortest_path_batch(rustfst_amd.DeviceFst.upload_many(accs_all, ctx2), dt2, ctx=ctx2)
            whole_packed = wdist.pack_device_paths(whole, args.acc_len + 8)
            got = wdist.interleave(last["gathered"], n_total)
            assert got.shape == whole_packed.shape and np.array_equal(got, whole_packed), \
                "the gathered records differ from a single-rank run of the same global batch"
            gather_check = {"records": int(n_total), "equal_to_single_rank_run": True}
            del whole, whole_packed, got
        barrier()
        gen_s += time.time() - t1
        for _ in range(args.warmup):
            step()
        drain()
        barrier()
        t_start = time.perf_counter()
        arcs = 0
        batch_arcs[0] = 0
        step_s = np.empty(args.steps, dtype=np.float64)  # host clock per step (a step ends with both results on the host)
        prev = t_start
        for k in range(args.steps):
            arcs += step()
            now = time.perf_counter()
            step_s[k] = now - prev
            prev = now
        drain()
        barrier()
        elapsed = time.perf_counter() - t_start
        # what the TIMED steps returned: the untimed extras below run step() again on other batches (step_512_acceptors) and
        # must not change what the CPU leg's parity guards compare with
        timed_last = {k: last[k] for k in ("sp", "outs", "n_arcs")}
        step_phases_us = None
        chain_us("after_timed_steps")
        if args.overlap and phases[4]:
            # (the counters include the warm-up steps: same code path)
            step_phases_us = {k: round(1e6 * phases[i] / phases[4], 1) for i, k in
                              enumerate(("first_begin", "second_begin", "batch_finish", "shortest_path_finish"))}

        if world > 1 or force_dist:
            import torch.distributed as dist
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            ta = torch.tensor([arcs, batch_arcs[0]], dtype=torch.int64, device=device)
            dist.all_reduce(ta, op=dist.ReduceOp.SUM)
            arcs, batch_arcs[0] = int(ta[0].item()), int(ta[1].item())

        # ------------------------------------------------------------------ multi-GPU extras (untimed; every rank takes part)
        # What the headline cannot show at N > 1 (`value` holds N replicated shortest_path(T) queries and a weak-scaled batch):
        #  * exchange_exposed: the same overlapped steps with and without the result all-gather — what the exchange costs a step;
        #  * batch_strong: configs[3] AS WRITTEN — its 512 acceptors split N ways (i mod N), fused batch only, results
        #    all-gathered — against the 512 on ONE GPU (rank 0's batch_sweep[512], same run): the strong-scaling figure;
        #  * batch_weak_4096: 4096 acceptors PER GPU (every rank the same 4096: throughput only), where the batch kernel has
        #    enough waves in flight for the per-acceptor time to be flat (0.23 us): the weak-scaling figure of the sharded leg.
        dist_extras = None
        if (world > 1 or force_dist) and not args.no_extras and args.overlap:
            import torch.distributed as dist

            def timed_all(fn, reps):
                """max over ranks of the host clock around `reps` calls of fn, between two barriers"""
                barrier()
                c0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                drain_local()
                barrier()
                tt_ = torch.tensor([time.perf_counter() - c0], dtype=torch.float64, device=device)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                return float(tt_.item()) / reps

            def drain_local():
                if exchange_on[0]:
                    drain()
            n_x = 200
            for _ in range(10):
                step()
            s_with = timed_all(step, n_x)
            exchange_on[0] = False
            for _ in range(10):
                step()
            s_without = timed_all(step, n_x)
            exchange_on[0] = True
            # strong: the 512 split N ways, records gathered (packed form: what a decoder reads), no S1 beside it
            idx512 = wdist.shard_indices(512, rank, world)
            d512 = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many([sweep_accs[i] for i in idx512], ctx2))
            tab = [None]

            n_loc512 = (512 + world - 1) // world  # (every rank contributes the same number of records: short shards pad with empty ones)
            pad512 = [None]

            def strong():
                tab[0], _ = rustfst_amd.compose_shortest_path_batch_packed(d512, dt2, args.acc_len + 8, ctx=ctx2, out=tab[0])
                send = tab[0]
                if send.shape[0] < n_loc512:
                    if pad512[0] is None:
                        pad512[0] = np.zeros((n_loc512, send.shape[1]), dtype=np.uint32)
                    pad512[0][:send.shape[0]] = send
                    send = pad512[0]
                comm.gather_records_begin(send, args.acc_len + 8)
                comm.gather_paths_end()
            for _ in range(5):
                strong()
            s_strong = timed_all(strong, 50)
            d4096 = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(sweep_accs[:SWEEP_MAX], ctx2))
            tab4 = [None]

            def weak():
                tab4[0], _ = rustfst_amd.compose_shortest_path_batch_packed(d4096, dt2, args.acc_len + 8, ctx=ctx2, out=tab4[0])
            for _ in range(3):
                weak()
            s_weak = timed_all(weak, 20)
            dist_extras = {
                "value_note": f"`value` = {world} replicated shortest_path(T) queries (one per rank, distinct sources) + the weak-scaled "
                              f"batch ({args.batch_per_gpu} acceptors per GPU): it scales with N by construction; the figures below are "
                              "the ones that can fail to",
                "exchange_exposed": {"ms_per_step_with_exchange": round(1e3 * s_with, 4), "ms_per_step_without": round(1e3 * s_without, 4),
                                     "exposed_us_per_step": round(1e6 * (s_with - s_without), 2), "steps_each": n_x},
                "batch_strong": {"workload": "configs[3] as written: 512 acceptors split over the ranks (i mod N), fused compose->shortest_path, "
                                             "records all-gathered every call; no shortest_path(T) beside it",
                                 "acceptors_per_rank": len(idx512), "ms": round(1e3 * s_strong, 4),
                                 "acceptors_per_s": round(512 / s_strong, 1)},
                "batch_weak_4096": {"workload": f"{SWEEP_MAX} acceptors per GPU (every rank the same {SWEEP_MAX}: throughput only), records, no exchange",
                                    "ms": round(1e3 * s_weak, 4), "us_per_acceptor": round(1e6 * s_weak / SWEEP_MAX, 4),
                                    "acceptors_per_s_all_ranks": round(world * SWEEP_MAX / s_weak, 1)},
            }
            del d512, d4096

        # ------------------------------------------------------------------ per-part times (untimed extra pass)
        # each request ALONE on its own context, host clock around the synchronous call (3 repetitions, best)
        def alone(fn):
            best = float("inf")
            for _ in range(3):
                torch.cuda.synchronize(device)
                c0 = time.perf_counter()
                fn()
                torch.cuda.synchronize(device)
                best = min(best, time.perf_counter() - c0)
            return 1e3 * best
        ms_sp_t = alone(lambda: dt.shortest_path())
        ms_batch = alone(lambda: rustfst_amd.compose_shortest_path_batch(daccs, dt2, ctx=ctx2))
        sweeps = ctx.stats()["sweeps"]
        relax_kernel = ("sssp_relax_kernel", "sssp_mbox_kernel", "sssp_mbox_resident_kernel", "sssp_relax_kernel + sssp_bin_expand/apply_kernel")[int(ctx.stats()["relax_kernel"])]

        # ------------------------------------------------------------------ configs[1]: ONE 1000-arc string against a
        # 100k-state T (the case a lone dependent chain makes the GPU lose to one CPU core; reported, not timed above)
        chain_us("after_alone_passes")
        # ------------------------------------------------------------------ varied sources: a serving mix asks from a
        # different state every time (wfst_fst_set_start on the resident handle: region plan, transpose and packed arcs stay;
        # the launch prediction was learned on the PREVIOUS source)
        varied = None
        if rank == 0 and not args.no_extras:
            rng_v = np.random.default_rng(20261001)
            srcs = [int(x) for x in rng_v.integers(0, int(t["n_states"]), 24)]
            v_ms = []
            v_launches = 0
            for s_ in srcs:
                dt.set_start(s_)
                torch.cuda.synchronize(device)
                c0 = time.perf_counter()
                p_ = dt.shortest_path()
                v_ms.append(1e3 * (time.perf_counter() - c0))
                v_launches += int(ctx.stats()["sweeps"])
            dt.set_start(s1_start)
            for _ in range(3):
                dt.shortest_path()  # (the prediction of T's own source again, for the extras below)
            v_sorted = sorted(v_ms)
            varied = {"workload": "shortest_path(T) from 24 random source states on the SAME resident handle (wfst_fst_set_start between "
                                  "queries; host clock around each synchronous call)",
                      "ms_median": round(v_sorted[len(v_sorted) // 2], 4), "ms_mean": round(sum(v_ms) / len(v_ms), 4),
                      "ms_min": round(v_sorted[0], 4), "ms_max": round(v_sorted[-1], 4),
                      "launches_per_query": round(v_launches / len(srcs), 2), "_src": srcs[0]}
            dt.set_start(srcs[0])
            varied["_flat0"] = dt.shortest_path().to_flat()
            dt.set_start(s1_start)
            for _ in range(3):
                dt.shortest_path()
        # ------------------------------------------------------------------ two queries at once: the resident lease of a device has
        # two units, and a context set to half the device takes one (tools/two_queries.py)
        two_q = None
        if rank == 0 and not args.no_extras:
            ch1, ch2 = rustfst_amd.Context(local_rank), rustfst_amd.Context(local_rank)
            ch1.set_resident_share(1)
            ch2.set_resident_share(1)
            for _ in range(5):
                ja, jb = dt.shortest_path_begin(ctx=ch1), dt.shortest_path_begin(ctx=ch2)
                ra, rb = ja.finish(), jb.finish()
            n_pairs = 100
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            for _ in range(n_pairs):
                ja, jb = dt.shortest_path_begin(ctx=ch1), dt.shortest_path_begin(ctx=ch2)
                ja.finish()
                jb.finish()
            s_pair = (time.perf_counter() - c0) / n_pairs
            c0 = time.perf_counter()
            for _ in range(n_pairs):
                dt.shortest_path()
                dt.shortest_path()
            s_seq = (time.perf_counter() - c0) / n_pairs
            same_path = ra.to_flat()["arcs"].tobytes() == rb.to_flat()["arcs"].tobytes() == dt.shortest_path().to_flat()["arcs"].tobytes()
            if not same_path:
                raise SystemExit("bench: two concurrent half-device queries returned another path than the whole-device query")
            two_q = {"workload": "two shortest_path(T) queries in flight at once on two contexts with wfst_ctx_set_resident_share(ctx, 1) "
                                 "(half the device each; begin, begin, end, end from one host thread; the same handle and source: what is "
                                 "measured is the overlap) against the same two queries one after the other on the whole device",
                     "ms_per_pair": round(1e3 * s_pair, 4), "queries_per_s": round(2 / s_pair, 1),
                     "one_after_the_other_ms_per_pair": round(1e3 * s_seq, 4), "one_after_the_other_queries_per_s": round(2 / s_seq, 1),
                     "resident_aborts": int(ch1.stats()["resident_aborts"] + ch2.stats()["resident_aborts"]),
                     "relax_kernels": [int(ch1.stats()["relax_kernel"]), int(ch2.stats()["relax_kernel"])], "paths_identical": True}
            del ja, jb, ra, rb, ch1, ch2
            for _ in range(3):
                dt.shortest_path()
        config2 = None
        if rank == 0 and not args.no_extras:
            t2 = synth.make_transducer(100_000, args.fanout, args.sigma, 0.0, seed=2)
            a2 = synth.make_acceptors(t2, 1, 1000, seed0=2)
            d2 = rustfst_amd.DeviceFst.from_arrays(t2["n_states"], t2["start"], t2["offsets"], t2["arcs"], t2["finals"], t2["props"], ctx)
            da2 = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(a2, ctx))
            for _ in range(3):
                rustfst_amd.compose_shortest_path_batch(da2, d2, ctx=ctx)
            best = float("inf")
            for _ in range(10):
                torch.cuda.synchronize(device)
                c0 = time.perf_counter()
                o2, n2 = rustfst_amd.compose_shortest_path_batch(da2, d2, ctx=ctx)
                best = min(best, time.perf_counter() - c0)
            config2 = {"workload": "configs[1]: one 1000-arc linear acceptor o T(100k states / 1M arcs) -> shortest path (fused call)",
                       "gpu_ms": round(1e3 * best, 4), "composed_arcs": int(n2), "_t2": t2, "_a2": a2}

        # ------------------------------------------------------------------ batch_sweep: where the fused batch saturates
        chain_us("after_config2")
        batch_sweep = None
        if rank == 0 and not args.no_extras:
            batch_sweep = []
            for bsz in (64, 512, SWEEP_MAX):
                db = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(sweep_accs[:bsz], ctx2))
                rustfst_amd.compose_shortest_path_batch(db, dt2, ctx=ctx2)
                best, n_b = float("inf"), 0
                for _ in range(5):
                    torch.cuda.synchronize(device)
                    c0 = time.perf_counter()
                    _, n_b = rustfst_amd.compose_shortest_path_batch(db, dt2, ctx=ctx2)
                    best = min(best, time.perf_counter() - c0)
                best_p, tab = float("inf"), None  # the same batch with its results as one table of records (no path handles)
                for _ in range(6):
                    torch.cuda.synchronize(device)
                    c0 = time.perf_counter()
                    tab, _ = rustfst_amd.compose_shortest_path_batch_packed(db, dt2, args.acc_len + 8, ctx=ctx2, out=tab)
                    best_p = min(best_p, time.perf_counter() - c0)
                batch_sweep.append({"batch": bsz, "ms": round(1e3 * best, 4), "us_per_acceptor": round(1e6 * best / bsz, 3),
                                    "packed_ms": round(1e3 * best_p, 4), "packed_us_per_acceptor": round(1e6 * best_p / bsz, 3),
                                    "composed_arcs": int(n_b), "arcs_per_s": round(2 * n_b / best, 1),
                                    "packed_arcs_per_s": round(2 * n_b / best_p, 1)})
                del db
            batch_sweep = {"workload": f"fused compose->shortest_path of B linear acceptors (len {args.acc_len}) against T, one call, "
                                       "host clock around the synchronous call (best of 5); `ms` = results as path FST handles (what the timed step uses), "
                                       "`packed_ms` = wfst_compose_shortest_path_batch_packed (one table of records: beyond a few hundred acceptors "
                                       "building the handles on the host is most of `ms`)", "points": batch_sweep}

        # ------------------------------------------------------------------ the whole 512-acceptor batch of configs[3] on ONE GPU
        # (the timed step above holds this GPU's share of it, 512 / 8: weak scaling; this is the same overlapped step with all
        # 512 acceptors here — the one-GPU form of configs[3])
        chain_us("after_batch_sweep")
        step_512 = None
        if rank == 0 and world == 1 and not args.no_extras and args.overlap:
            daccs_keep, order_keep = daccs, args.order
            daccs = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(sweep_accs[:512], ctx2))
            step_512 = {"workload": f"the timed step with ALL 512 acceptors (len {args.acc_len}) of configs[3] on this one GPU: "
                                    "shortest_path(T) + fused compose->shortest_path of 512, overlapped on two contexts; both enqueue orders "
                                    "(a 512-string batch holds 64 compute units for ~0.3 ms: the relaxation's resident launch, which wants one "
                                    "workgroup on each of 245, waits for them)", "steps": 200}
            for share, tag in ((0, ""), (1, "_half_share")):
                # (half share: wfst_ctx_set_resident_share(ctx, 1) on the query's context — 123 resident workgroups of 8192 states
                # instead of 245 of 4096: the solve alone is slower, and runs BESIDE the batch's 64 compute units instead of behind them)
                ctx.set_resident_share(share)
                for order in ("s1-first", "s2-first"):
                    args.order = order
                    for _ in range(20):
                        step()
                    torch.cuda.synchronize(device)
                    c0 = time.perf_counter()
                    a512, n512 = 0, 200
                    for _ in range(n512):
                        a512 += step()
                    torch.cuda.synchronize(device)
                    el512 = time.perf_counter() - c0
                    step_512[order + tag] = {"ms_per_step": round(1e3 * el512 / n512, 4), "arcs_per_s": round(a512 / el512, 1),
                                             "acceptors_per_s": round(512 * n512 / el512, 1)}
            ctx.set_resident_share(0)
            step_512["resident_aborts"] = int(ctx.stats()["resident_aborts"])
            args.order = order_keep
            for _ in range(3):  # (the handle's launch hints are those of the 4096-state plan again)
                dt.shortest_path()
            daccs = daccs_keep

        # ------------------------------------------------------------------ configs[4]: HCLG-shaped operand under look-ahead
        # composition + n = 10 shortest paths (rustfst-cli/src/cmds/compose.rs:77-181 wires the look-ahead recipe)
        chain_us("after_step_512")
        config5 = None
        if rank == 0 and not args.no_extras and args.config5_states > 0:
            config5 = config5_extra(args.config5_states, ctx, device)
        chain_us("after_config5")

        # ------------------------------------------------------------------ the same roofline figure at other sizes
        rvs = None
        if rank == 0 and not args.no_extras and args.roofline_sizes:
            rvs = roofline_vs_size(ctx, [int(x) for x in args.roofline_sizes.split(",") if x], args.fanout, args.sigma)
        chain_us("after_roofline_vs_size")

        # ------------------------------------------------------------------ roofline of the relaxation kernel
        # HIP events bracket every sssp_relax_kernel launch on ctx's stream (wfst_ctx_set_profiling);
        # achieved = algorithmic bytes (20 B per arc relaxed + 12 B per frontier state) / kernel time.
        roofline = None
        if rank == 0:
            ctx.reset_stats()
            ctx.set_profiling(True)
            dt.shortest_path()
            ctx.set_profiling(False)
            st = ctx.stats()
            profiled_ms, profiled_launches = st["relax_ms"], st["relax_launches"]
            # The per-launch events above synchronise after every sweep, and a sweep launched onto an idle GPU runs ~30 %
            # longer than inside its chain.  The time that counts is the chain's: two events on the solve's stream around
            # the pre-queued sweeps of an ordinary (un-profiled) repeated query, no synchronisation in between
            # (wfst_ctx_set_profiling(ctx, 2)); median of 31 solves.  It is what rocprofv3's kernel trace sums to.
            ctx.set_profiling(2)
            chain = []
            for _ in range(31):
                dt.shortest_path()
                cs = ctx.stats()
                if cs["relax_launches"]:
                    chain.append((cs["relax_ms"], int(cs["relax_launches"])))
            ctx.set_profiling(0)
            if chain:
                chain.sort()
                st = dict(st)
                st["relax_ms"], st["relax_launches"] = chain[len(chain) // 2]
            if st["relax_ms"] > 0:
                # SURVEY §8(d) accounting (the contract figure): every arc of T counted ONCE, B_relax = 20 E + 12 N bytes
                # per solve, over the summed relaxation-kernel time of the solve — re-relaxed arcs earn nothing
                solve_bytes = 20.0 * e_t + 12.0 * args.states
                achieved = solve_bytes / (st["relax_ms"] * 1e-3) / 1e9
                # the same time against the arcs / states the launches actually relaxed (bands re-relax some arcs)
                relaxed_bytes = 20.0 * st["relax_arcs"] + 12.0 * st["relax_states"]
                algo_bytes = solve_bytes
                roofline = {
                    "kernel": relax_kernel, "bound": "hbm", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "accounting": "SURVEY 8(d): B_relax = 20 E + 12 N bytes per solve (each arc once) / summed kernel time of the solve's launches",
                    "launches": int(st["relax_launches"]),
                    "avg_launch_us": round(1e3 * st["relax_ms"] / max(1, st["relax_launches"]), 2),
                    "algorithmic_bytes_per_launch": round(solve_bytes / max(1, st["relax_launches"]), 1),
                    "solve_algorithmic_bytes": 20 * e_t + 12 * args.states,
                    "solve_relax_kernel_ms": round(st["relax_ms"], 4),
                    "chain_us_by_point": dict(chain_probe),
                    "solve_frac": round(achieved / HBM_PEAK_GBS, 5),
                    "arcs_relaxed": int(st["relax_arcs"]), "frontier_states": int(st["relax_states"]),
                    "relax_arcs_per_s": round(st["relax_arcs"] / (st["relax_ms"] * 1e-3), 1),
                    "re_relaxation_factor": round(st["relax_arcs"] / max(1, e_t), 3),
                    # (round 2 quoted this one as `frac`: it credits the re-relaxed arcs)
                    "frac_of_relaxed_arcs": round(relaxed_bytes / (st["relax_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "timing": ("HIP events around the pre-queued launch chain of un-profiled repeated queries (median of "
                               f"{len(chain)}); arcs / states counted by one separate profiled solve") if chain else
                              "HIP events around every launch of one profiled solve (synchronised after each)",
                    "profiled_solve": {"launches": int(profiled_launches), "relax_kernel_ms": round(profiled_ms, 4),
                                       "avg_launch_us": round(1e3 * profiled_ms / max(1, profiled_launches), 2),
                                       "note": "per-launch events + a synchronisation after every launch: each starts on an idle GPU"},
                }
                # HBM-side traffic cannot be counted live: it comes from the committed rocprofv3 PMC passes of this
                # same command (profiles/pmc_relax_traffic.json, regenerated by tools/profile_round.sh), per launch
                tp = os.path.join(ROOT, "profiles", "pmc_relax_traffic.json")
                default_cfg = (args.states, args.fanout, args.sigma) == (1_000_000, 10, 256)
                if os.path.exists(tp) and default_cfg:
                    with open(tp) as fh:
                        pmc = json.load(fh)
                    roofline["traffic"] = round(pmc["traffic_bytes_per_solve"] / max(1, st["relax_launches"]))
                    roofline["traffic_unit"] = "bytes per launch (per solve / launches, like `achieved`'s algorithmic_bytes_per_launch)"
                    roofline["traffic_bytes_per_solve"] = round(pmc["traffic_bytes_per_solve"])
                    roofline["traffic_over_algorithmic"] = round(pmc["traffic_bytes_per_solve"] / algo_bytes, 2)
                    roofline["traffic_over_solve_algorithmic"] = round(pmc["traffic_bytes_per_solve"] / (20.0 * e_t + 12.0 * args.states), 2)
                    roofline["traffic_source"] = pmc["source"] + "; " + pmc["correction"]
                    # the PMC file is a committed measurement, not a live one: say what it was taken at, and whether
                    # the relaxation sources have changed since
                    roofline["traffic_commit"] = pmc.get("commit")
                    roofline["traffic_kernel"] = pmc.get("kernel")
                    roofline["traffic_stale"] = bool(pmc.get("kernel_sources_sha256") != kernel_sources_sha256()
                                                     or pmc.get("kernel") != relax_kernel)
                    for k in ("l2_hit_rate", "tcc_ea_atomic_per_solve", "counters_file"):
                        if k in pmc:
                            roofline[k] = pmc[k]

        # the other kernel of the step, for completeness: the fused batch kernel is LATENCY bound (one wave per
        # problem walking ~200 dependent BFS levels), so its fraction of the HBM peak is tiny by construction;
        # algorithmic bytes per expanded state = 24 + 16 f1 + 4 f1 ceil(log2(f2 + 1)) + 48 m  
        batch_kernel = None
        if rank == 0:
            import math
            ctx2.reset_stats()
            ctx2.set_profiling(True)
            rustfst_amd.compose_shortest_path_batch(daccs, dt2, ctx=ctx2)
            ctx2.set_profiling(False)
            st2 = ctx2.stats()
            if st2["compose_ms"] > 0 and st2["compose_states"] > 0:
                f1, f2 = 1.0, float(args.fanout)
                m = st2["compose_arcs"] / st2["compose_states"]
                per_state = 24 + 16 * f1 + 4 * f1 * math.ceil(math.log2(f2 + 1)) + 48 * m
                ab = per_state * st2["compose_states"]
                batch_kernel = {
                    "kernel": "string_compose_sp_kernel" if st2["string_problems"] == len(mine) else "compose_wave_kernel<FLAG_SP>",
                    "string_problems": int(st2["string_problems"]), "bound": "latency (dependent round trips per BFS level)",
                    "kernel_ms": round(st2["compose_ms"], 4), "problems": len(mine),
                    "composed_states": int(st2["compose_states"]), "composed_arcs": int(st2["compose_arcs"]),
                    "algorithmic_bytes": round(ab), "achieved_GBps": round(ab / (st2["compose_ms"] * 1e-3) / 1e9, 3),
                    "frac_of_hbm_peak": round(ab / (st2["compose_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                    "us_per_bfs_level": round(1e3 * st2["compose_ms"] / (args.acc_len + 1), 3),
                }

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1): the oracle
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py
        ot = oracle_py.OracleFst.from_flat(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"])
        oaccs = [oracle_py.OracleFst.from_flat(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"])
                 for a in (accs_all[i] for i in mine)]
        # bounded sample: whole steps of the same workload until ~10 s of CPU work have been timed (at most 8 steps)
        cpu_s = t_sp = t_batch = 0.0
        cpu_steps = 0
        while cpu_steps < 8 and (cpu_steps == 0 or cpu_s < 10.0):
            c0 = time.perf_counter()
            osp = ot.shortest_path()
            c1 = time.perf_counter()
            o_outs, o_arcs, o_sec = oracle_py.compose_shortest_path_batch(oaccs, ot, n_threads=args.cpu_threads)
            t_sp += c1 - c0
            t_batch += o_sec
            cpu_s += (c1 - c0) + o_sec
            cpu_steps += 1
        cpu_arcs = cpu_steps * (e_t + 2 * o_arcs)
        # the batch leg on every host core (the reference's algorithms are single-threaded; independent acceptors are the
        # only parallelism a CPU deployment has): threads of the C++ restatement, one acceptor at a time each
        all_cores = None
        n_cores = min(os.cpu_count() or 1, len(oaccs))  # one acceptor per thread at most
        if not args.no_extras and n_cores > 1:
            _, o_arcs_mt, sec_mt = oracle_py.compose_shortest_path_batch(oaccs, ot, n_threads=n_cores)
            reps = max(1, min(50, int(2.0 / max(sec_mt, 1e-4))))
            sec_mt = min(oracle_py.compose_shortest_path_batch(oaccs, ot, n_threads=n_cores)[2] for _ in range(reps))
            all_cores = {"cores": n_cores, "host_cores": os.cpu_count(), "ms_batch": round(1e3 * sec_mt, 3), "batch_arcs_per_s": round(2 * o_arcs_mt / sec_mt, 1),
                         "note": "batch leg only (compose -> shortest_path of the acceptors, one thread per acceptor); "
                                 "shortest_path(T) is one sequential search on any number of cores"}
        if varied is not None:  # one of the varied sources against the oracle: the whole path FST, bit for bit
            c0 = time.perf_counter()
            ov = oracle_py.OracleFst.from_flat(t["n_states"], varied["_src"], t["offsets"], t["arcs"], t["finals"], t["props"]).shortest_path_canonical().to_flat()
            varied["cpu_ms_one_source"] = round(1e3 * (time.perf_counter() - c0), 1)
            gv = varied["_flat0"]
            ok_v = (int(gv["n_states"]) == int(ov["n_states"]) and np.array_equal(gv["offsets"], ov["offsets"]) and
                    gv["arcs"].tobytes() == ov["arcs"].tobytes() and gv["finals"].tobytes() == ov["finals"].tobytes())
            if not ok_v:
                raise SystemExit(f"bench: shortest_path from source {varied['_src']} differs from the oracle's")
            varied["path_of_first_source_bit_exact_vs_oracle"] = True
        if config2 is not None:
            ot2 = oracle_py.OracleFst.from_flat(*(config2["_t2"][k] for k in ("n_states", "start", "offsets", "arcs", "finals", "props")))
            oa2 = [oracle_py.OracleFst.from_flat(*(a[k] for k in ("n_states", "start", "offsets", "arcs", "finals", "props"))) for a in config2["_a2"]]
            config2["cpu_ms"] = round(1e3 * min(oracle_py.compose_shortest_path_batch(oa2, ot2, n_threads=1)[2] for _ in range(10)), 4)
            config2["cpu_cores"] = 1
        host = host_cpu_info()
        # parity guards of what was just timed.  (1) the composed arc count of the timed batch == the oracle's; (2) the paths
        # of two of its acceptors, bit for bit, against the oracle's own compose (connect) -> canonical shortest path;
        # (3) the total weight of shortest_path(T).  A guard that fails ends the run: a line must not carry a false one.
        assert o_arcs == timed_last["n_arcs"], \
            f"composed arcs of the timed batch ({timed_last['n_arcs']}) differ from the oracle's ({o_arcs})"
        paths_checked = []
        for i in sorted({0, len(oaccs) - 1}):
            can = oaccs[i].compose(ot, connect=True).shortest_path_canonical().to_flat()
            got = timed_last["outs"][i].to_flat()
            same = (got["n_states"] == can["n_states"] and got["start"] == can["start"] and np.array_equal(got["offsets"], can["offsets"])
                    and np.array_equal(got["arcs"], can["arcs"]) and np.array_equal(got["finals"].view(np.uint32), can["finals"].view(np.uint32)))
            assert same, f"path of acceptor {i} of the timed batch differs from the oracle's"
            paths_checked.append(int(mine[i]))
        gw = timed_last["sp"].to_flat()
        gpu_total = float(np.float32(np.add.reduce(gw["arcs"]["weight"][::-1].astype(np.float32), dtype=np.float32) + gw["finals"][0])) if gw["n_states"] else float("inf")
        cpu_baseline = {
            "value": round(cpu_arcs / cpu_s, 1), "unit": "arcs/s", "cores": args.cpu_threads, "kind": "port",
            "sample": f"{cpu_steps} full steps, each: shortest_path(T {args.states} states/{e_t} arcs) once + "
                      f"compose->shortest_path of {len(oaccs)} acceptors (len {args.acc_len}); C++ restatement of "
                      f"rustfst 1.3.1 (oracle/), not rustfst binaries",
            "seconds": round(cpu_s, 3), "steps": cpu_steps, "ms_shortest_path_T": round(1e3 * t_sp / cpu_steps, 2),
            "ms_batch": round(1e3 * t_batch / cpu_steps, 2), "queue_kind": osp.queue_kind,
            "shortest_path_T_weight_cpu": osp.total_weight, "shortest_path_T_weight_gpu": gpu_total,
            "composed_arcs_match": bool(o_arcs == timed_last["n_arcs"]),
            "paths_checked_bit_exact": paths_checked,
            "cpu_model": host["cpu_model"], "cpu_governor": host["governor"], "cpu_of_thread": host["cpu_of_thread"],
            "cpu_mhz_of_thread": host["mhz"], "numa_nodes": host["numa_nodes"],
            "all_cores": all_cores,
        }

    if rank == 0:
        value = arcs / elapsed
        if config2 is not None:
            config2 = {k: v for k, v in config2.items() if not k.startswith("_")}
        if varied is not None:
            varied = {k: v for k, v in varied.items() if not k.startswith("_")}
        rccl_world = 1
        if world > 1 or force_dist:
            import torch.distributed as dist
            rccl_world = dist.get_world_size()
            assert rccl_world == world and comm.world == world, "the RCCL communicator does not span --gpus ranks"
        # the figure that scales with the number of GPUs: the sharded batch alone (acceptor i on rank i mod N)
        batch_only = {
            "acceptors_per_s": round(n_total * args.steps / elapsed, 1),
            "us_per_compose_shortest_path": round(1e6 * elapsed / max(1, n_total * args.steps), 4),
            "batch_arcs_per_s": round(batch_arcs[0] / elapsed, 1),
            "note": "global batch (acceptors on all ranks) per timed second of the overlapped step; the direct shortest_path(T) "
                    "of a step is one single-source query per rank (rank r: start state (start + 104729 r) mod N): distinct work, "
                    "counted in `value`, not in these",
        }
        if dist_extras is not None and batch_sweep is not None:
            one = next((pt for pt in batch_sweep["points"] if pt["batch"] == 512), None)
            if one is not None:
                dist_extras["batch_strong"]["one_gpu_ms"] = one["packed_ms"]
                dist_extras["batch_strong"]["speedup_vs_one_gpu"] = round(one["packed_ms"] / dist_extras["batch_strong"]["ms"], 3)
                dist_extras["batch_strong"]["note"] = ("one_gpu_ms = rank 0's batch_sweep[512].packed_ms of this run (no exchange); "
                                                       "the batch kernel is latency-bound at one wave per acceptor, so N GPUs buy less than N")
        ms = 1e3 * step_s
        out = {
            "metric": "arcs relaxed/sec (compose -> shortest_path, 1M-state / 10M-arc FST)",
            "value": round(value, 1), "unit": "arcs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "ms_per_step_stats": {"mean": round(float(ms.mean()), 4), "std": round(float(ms.std()), 4), "min": round(float(ms.min()), 4),
                                  "p50": round(float(np.percentile(ms, 50)), 4), "p99": round(float(np.percentile(ms, 99)), 4),
                                  "timed_seconds": round(elapsed, 3), "clock": "host perf_counter per step on rank 0"},
            "step_host_phases_us": step_phases_us,
            "rccl_world_size": rccl_world, "rccl_used": bool(world > 1 or force_dist), "gather_check": gather_check,
            "batch_only": batch_only, "multi_gpu": dist_extras,
            "s1_start_states": "T's own" if world == 1 else f"rank r: ({int(t['start'])} + 104729 r) mod {int(t['n_states'])}",
            "step_schedule": "serial (one stream)" if (not args.overlap) else ("S1 (shortest_path(T)) enqueued async on stream 1, then the S2 batch on stream 2" if args.order == "s1-first" else "S2 batch enqueued async on stream 2, then S1 on stream 1") + ", S2 collected, S1 collected (two contexts, one host thread); "
                             + (f"batch context on {args.batch_cus} reserved CUs" if args.batch_cus > 0 else "no CU partitioning"),
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"configs[2]+[3]: shortest_path(T) + fused compose->shortest_path of {args.batch_per_gpu} "
                            f"linear acceptors (len {args.acc_len}) per GPU against one shared T",
                "transducer": {"states": args.states, "arcs": e_t, "fanout": args.fanout, "sigma": args.sigma, "seed": 3},
                "batch_per_gpu": args.batch_per_gpu, "global_batch": n_total, "acceptor_len": args.acc_len,
                "batch_note": "configs[3]'s batch of 512 acceptors is sharded 64 per GPU over 8 GPUs (weak scaling: the timed step "
                              "holds one GPU's share); the same step with all 512 on one GPU is `step_512_acceptors`",
                "parallelism": f"acceptor-sharded x{world}, T replicated, RCCL all-gather of results only",
            },
            "ms_shortest_path_T": round(ms_sp_t, 4), "ms_compose_shortest_path_batch": round(ms_batch, 4),
            "ms_per_compose_shortest_path": round(ms_batch / max(1, len(mine)), 5),
            "relaxation_sweeps": int(sweeps), "composed_arcs_per_batch": int(timed_last["n_arcs"]),
            "setup_seconds": round(gen_s, 2), "priming_steps": 3,
            "cold_query_ms": None if cold is None else {
                "first_in_process": cold[0], "second": cold[1], "third": cold[2], "fourth": cold[3],
                "fresh_handle_warm_process": {"first": cold2[0], "second": cold2[1], "third": cold2[2]},
                "fresh_handle_frac_of_hbm_roofline": {k: round((20.0 * e_t + 12.0 * args.states) / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                                      for k, v in (("first", cold2[0]), ("second", cold2[1]), ("third", cold2[2]))},
                "note": "shortest_path(T) on a fresh HBM-resident handle: the 1st query builds the mailbox region plan and takes the parent "
                        "pass; the 2nd builds the transpose for the backtrace (through the plan's regions: ~0.3 ms); from the 3rd on the solve "
                        "is one predicted batch with the one-launch tail; `first_in_process` also pays the process's first launches (code "
                        "objects, the pool's first large allocations), `fresh_handle_warm_process` does not"},
            "config5": config5, "batch_sweep": batch_sweep, "step_512_acceptors": step_512,
            "config2_single_string": config2,
            "varied_sources": varied,
            "two_queries_half_share": two_q,
            "reference_harness_split": harness,
            "roofline": roofline, "roofline_vs_size": rvs, "batch_kernel": batch_kernel, "cpu_baseline": cpu_baseline,
        }
        out_line = json.dumps(out)
    # RCCL prints a version banner through C stdio when its communicator comes up; piped, that buffer only drains at exit
    # and the banner would land BEHIND the result line.  Everybody drains it now, rank 0 prints the line last.
    _flush_c_stdio()
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.barrier()
    if rank == 0:
        print(out_line, flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()
        _flush_c_stdio()


if __name__ == "__main__":
    main()
