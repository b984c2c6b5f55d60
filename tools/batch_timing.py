"""Where a large fused batch spends its time: B linear acceptors against the C3 transducer, host phases of the C call
(WFST_HOST_TIMING=1 prints them to stderr) and the kernel's own time.   python tools/batch_timing.py [B ...]"""
import os, sys, time
os.environ.setdefault("WFST_HOST_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth

sizes = [int(a) for a in sys.argv[1:]] or [64, 512, 4096]
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, max(sizes), 200, seed0=50_000)
ctx = rustfst_amd.Context(0)
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
for b in sizes:
    db = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs[:b], ctx))
    rustfst_amd.compose_shortest_path_batch(db, dt)
    best = 1e9
    for _ in range(4):
        sys.stderr.write(f"--- B = {b}\n")
        t0 = time.perf_counter(); outs, na = rustfst_amd.compose_shortest_path_batch(db, dt); dt_call = time.perf_counter() - t0
        t1 = time.perf_counter(); del outs; dt_del = time.perf_counter() - t1
        best = min(best, dt_call)
    bestp, tab = 1e9, None
    for _ in range(5):
        sys.stderr.write(f"--- B = {b} packed\n")
        t0 = time.perf_counter(); tab, _ = rustfst_amd.compose_shortest_path_batch_packed(db, dt, 208, out=tab); bestp = min(bestp, time.perf_counter() - t0)
    print(f"B = {b}: handles {best*1e3:.3f} ms (best of 4) + destroying them {dt_del*1e3:.3f} ms; packed records {bestp*1e3:.3f} ms; composed arcs {na}", flush=True)
