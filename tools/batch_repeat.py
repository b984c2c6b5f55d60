"""N fused compose -> shortest_path batches (64 linear acceptors of 200 labels against the 1M-state T) for rocprofv3
kernel traces / PMC passes of string_compose_sp_kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nacc = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L = int(sys.argv[4]) if len(sys.argv) > 4 else 200
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, nacc, L, seed0=1000)
ctx = rustfst_amd.Context(0)
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
da = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx))
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter(); rustfst_amd.compose_shortest_path_batch(da, dt, ctx=ctx); best = min(best, time.perf_counter() - t0)
ctx.reset_stats(); ctx.set_profiling(True); rustfst_amd.compose_shortest_path_batch(da, dt, ctx=ctx); ctx.set_profiling(False)
st = ctx.stats()
print(f"best of {reps}: {best*1e3:.3f} ms; kernel {st['compose_ms']*1e3:.1f} us = {st['compose_ms']*1e3/(L+1):.3f} us per level; string problems {st['string_problems']}")
