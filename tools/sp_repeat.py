"""N un-profiled shortest_path(T) solves on the C3 graph (for rocprofv3 --kernel-trace timelines and PMC passes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # solves before anything is timed (the GPU's clocks ramp over the first few hundred)
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
import hashlib
dist, hops = d.shortest_distance(want_hops=True)  # (a digest of the labels: library variants run in separate processes)
print("labels sha1", hashlib.sha1(dist.tobytes() + hops.tobytes()).hexdigest()[:16], "kernel", ctx.stats()["relax_kernel"])
for _ in range(warm):
    d.shortest_path()
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter(); d.shortest_path(); best = min(best, time.perf_counter() - t0)
print(f"best of {reps}: {best*1e3:.3f} ms, sweeps {ctx.stats()['sweeps']} (after {warm} warm-up solves)")
# the same solves with the relaxation chain bracketed by HIP events (profiling mode 2): under `rocprofv3 --kernel-trace` this
# process then holds BOTH clocks for the same launches — the events' chain time and the trace's per-kernel durations
import statistics
ctx.set_profiling(2)
chain = []
for _ in range(max(5, reps // 2)):
    d.shortest_path()
    st = ctx.stats()
    if st["relax_launches"]:
        chain.append(st["relax_ms"] * 1e3)
ctx.set_profiling(0)
if chain:
    print(f"relaxation chain by HIP events (this process): median {statistics.median(chain):.1f} us, min {min(chain):.1f} us "
          f"over {len(chain)} solves, {int(ctx.stats()['relax_launches'])} launches, kernel {int(ctx.stats()['relax_kernel'])}")
