"""A/B of shortest_path(T) under several environment settings in ONE process (the relaxation reads its knobs per solve).

usage: sp_ab.py STATES REPS cfg [cfg ...]      cfg = "name:VAR=val,VAR=val" or "name:" (defaults)
Prints, per configuration: best / median host ms of shortest_path(T), the relaxation chain's device time (HIP events around
the pre-queued launches, profiling mode 2), launches, the kernel that ran, and whether the distances are bit-identical to
the first configuration's."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

states, reps = int(sys.argv[1]), int(sys.argv[2])
fan = int(os.environ.get("R4_FANOUT", "10"))
cfgs = sys.argv[3:] or ["default:"]
t = synth.make_transducer(states, fan, 256, 0.0, seed=3)
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
ref = None
for cfg in cfgs:
    name, _, kv = cfg.partition(":")
    env = dict(x.split("=", 1) for x in kv.split(",") if x)
    for k, v in env.items():
        os.environ[k] = v
    try:
        ctx.set_profiling(0)
        dist, hops = d.shortest_distance(want_hops=True)
        key = (dist.view(np.uint32).astype(np.uint64) << 32) | hops
        same = True if ref is None else bool(np.array_equal(key, ref))
        if ref is None:
            ref = key
        for _ in range(4):
            d.shortest_path()
        host = []
        for _ in range(reps):
            t0 = time.perf_counter(); d.shortest_path(); host.append((time.perf_counter() - t0) * 1e3)
        ctx.set_profiling(2)
        chain, launches = [], 0
        for _ in range(max(5, reps // 2)):
            d.shortest_path()
            st = ctx.stats()
            if st["relax_launches"]:
                chain.append(st["relax_ms"] * 1e3); launches = st["relax_launches"]
        ctx.set_profiling(0)
        st = ctx.stats()
        print(f"{name:28s} host best {min(host):.3f} med {statistics.median(host):.3f} ms | chain med "
              f"{(statistics.median(chain) if chain else float('nan')):.1f} min {(min(chain) if chain else float('nan')):.1f} us in {launches} launches | "
              f"kernel {st['relax_kernel']} sweeps {st['sweeps']} aborts {st['resident_aborts']} | same keys {same}", flush=True)
    finally:
        for k in env:
            del os.environ[k]
