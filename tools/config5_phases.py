"""configs[4] pipeline on a smaller operand, phase by phase: relabel (device handles), look-ahead compose batch, n-best.
python tools/config5_phases.py [states]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth, ShortestPathConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t5 = synth.make_transducer(n, 10, 256, 0.05, seed=9)
accs = synth.make_acceptors(t5, 64, 200, seed0=77)
arcs = t5["arcs"].copy()
arcs["ilabel"], arcs["olabel"] = t5["arcs"]["olabel"].copy(), t5["arcs"]["ilabel"].copy()
t1 = dict(t5); t1["arcs"], t1["props"] = arcs, synth.O_LABEL_SORTED
ctx = rustfst_amd.Context(0)
d1 = rustfst_amd.DeviceFst.from_arrays(t1["n_states"], t1["start"], t1["offsets"], t1["arcs"], t1["finals"], t1["props"], ctx)
t0 = time.perf_counter(); la = rustfst_amd.LookAhead(d1); print(f"lookahead_create {time.perf_counter()-t0:.3f} s")
das = rustfst_amd.DeviceFst.upload_many(accs, ctx)
t0 = time.perf_counter(); rel = [la.relabel(d) for d in das]; print(f"relabel 64: {1e3*(time.perf_counter()-t0):.3f} ms")
outs = la.compose_batch(rel)
for _ in range(3):
    ctx.synchronize(); t0 = time.perf_counter(); outs = la.compose_batch(rel); ctx.synchronize()
    print(f"compose_batch 64: {1e3*(time.perf_counter()-t0):.3f} ms, states {min(o.num_states for o in outs)}..{max(o.num_states for o in outs)}")
cfg = ShortestPathConfig(nshortest=10)
rustfst_amd.shortest_path_batch(outs, cfg, ctx=ctx)
t0 = time.perf_counter(); rustfst_amd.shortest_path_batch(outs, cfg, ctx=ctx); print(f"nbest10 x64: {1e3*(time.perf_counter()-t0):.3f} ms")
one = la.compose(rel[0]); ctx.synchronize()
t0 = time.perf_counter(); one = la.compose(rel[0]); ctx.synchronize(); print(f"one look-ahead composition: {1e3*(time.perf_counter()-t0):.3f} ms ({one.num_states} states)")
cfg1 = ShortestPathConfig(nshortest=1)
rustfst_amd.shortest_path_batch(outs, cfg1, ctx=ctx)
t0 = time.perf_counter(); p1 = rustfst_amd.shortest_path_batch(outs, cfg1, ctx=ctx); print(f"1-best x64 (shortest_path_batch, nshortest = 1): {1e3*(time.perf_counter()-t0):.3f} ms")
t0 = time.perf_counter(); p1 = [o.shortest_path() for o in outs]; print(f"1-best x64 (one call each): {1e3*(time.perf_counter()-t0):.3f} ms")
cfgu = ShortestPathConfig(nshortest=10, unique=True)
accept = []
for o in outs:  # the composed lattices as acceptors of their output labels (what a decoder ranks)
    f = o.to_flat(); f["arcs"]["ilabel"] = f["arcs"]["olabel"]; f["props"] = 0x10000
    accept.append(f)
da = rustfst_amd.DeviceFst.upload_many(accept, ctx)
rustfst_amd.shortest_path_batch(da, cfgu, ctx=ctx)
t0 = time.perf_counter(); u = rustfst_amd.shortest_path_batch(da, cfgu, ctx=ctx); print(f"unique 10-best x64 (batch): {1e3*(time.perf_counter()-t0):.3f} ms")
t0 = time.perf_counter(); u1 = [d.shortest_path(cfgu) for d in da[:8]]; print(f"unique 10-best, one call each: {1e3*(time.perf_counter()-t0)/8:.3f} ms per lattice")
