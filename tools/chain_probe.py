"""Why does bench.py's `roofline` block read ~266 us for the relaxation chain where every standalone tool reads ~285 us?
The same HIP-event bracket (profiling mode 2) on the same T under the conditions that differ: the context's stream (its own /
a torch stream), a per-launch-profiled solve right before (bench does one), a second context with a batch in the process."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustfst_amd
from rustfst_amd import synth

t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)


def chain(ctx, d, n=31):
    ctx.set_profiling(2)
    v = []
    for _ in range(n):
        d.shortest_path()
        st = ctx.stats()
        if st["relax_launches"]:
            v.append((st["relax_ms"] * 1e3, int(st["relax_launches"])))
    ctx.set_profiling(0)
    v.sort()
    return "median %.1f us (%d launches), min %.1f, max %.1f over %d" % (v[len(v) // 2][0], v[len(v) // 2][1], v[0][0], v[-1][0], len(v))


def up(ctx):
    return rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)


ctx = rustfst_amd.Context(0)
d = up(ctx)
for _ in range(10):
    d.shortest_path()
print("own stream                                  :", chain(ctx, d))
ctx.reset_stats(); ctx.set_profiling(True); d.shortest_path(); ctx.set_profiling(False)
print("own stream, right after a profiled solve    :", chain(ctx, d))
print("own stream, again                           :", chain(ctx, d))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ctx_t = rustfst_amd.Context(0, stream=s1.cuda_stream)
ctx2 = rustfst_amd.Context(0, stream=s2.cuda_stream)
dt = up(ctx_t)
for _ in range(10):
    dt.shortest_path()
print("torch stream                                :", chain(ctx_t, dt))
daccs = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx2))
for _ in range(300):
    sp_job = dt.shortest_path_begin()
    job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt, ctx=ctx2)
    job.finish(); sp_job.finish()
print("torch stream, after 300 overlapped steps    :", chain(ctx_t, dt))
ctx_t.reset_stats(); ctx_t.set_profiling(True); dt.shortest_path(); ctx_t.set_profiling(False)
print("torch stream, right after a profiled solve  :", chain(ctx_t, dt))
print("own stream, same process, now               :", chain(ctx, d))
