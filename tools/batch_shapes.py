"""The fused batch (compose -> shortest_path of 64 acceptors against T) on other shapes than the benchmark's: acceptor length,
alphabet size (a small alphabet means wide lattice levels), fan-out of T.  Prints ms per batch, composed arcs, how many of the
problems the string o T kernel served.   python tools/batch_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

ctx = rustfst_amd.Context(0)
for states, fan, sigma, length, B in [(1_000_000, 10, 256, 200, 64), (1_000_000, 10, 256, 50, 64), (1_000_000, 10, 256, 1000, 64),
                                      (1_000_000, 10, 64, 200, 64), (1_000_000, 10, 16, 200, 64),
                                      (1_000_000, 24, 256, 200, 64), (1_000_000, 4, 256, 200, 64), (100_000, 10, 256, 200, 64),
                                      (1_000_000, 10, 256, 200, 8)]:  # (an alphabet smaller than the fan-out makes the lattice grow by fan-out / sigma per level: sigma 8, length 100 does not fit the device)
    t = synth.make_transducer(states, fan, sigma, 0.0, seed=3)
    accs = synth.make_acceptors(t, B, length, seed0=7000)
    dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
    da = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx))
    for _ in range(3):
        outs, n_arcs = rustfst_amd.compose_shortest_path_batch(da, dt, ctx=ctx)
    ctx.reset_stats() if hasattr(ctx, "reset_stats") else None
    best = float("inf")
    for _ in range(5):
        ctx.synchronize()
        c0 = time.perf_counter()
        outs, n_arcs = rustfst_amd.compose_shortest_path_batch(da, dt, ctx=ctx)
        best = min(best, time.perf_counter() - c0)
    st = ctx.stats()
    print(f"T({states}, fan-out {fan}, sigma {sigma}) x {B} acceptors of length {length}: {1e3 * best:.3f} ms, {n_arcs} composed arcs "
          f"({n_arcs / B / length:.2f} per level), string-kernel problems so far {st['string_problems']}, {1e6 * best / B / length:.3f} us per level", flush=True)
    del da, dt, outs
