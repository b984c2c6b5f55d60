"""GPU-side durations inside one overlapped bench step (events on the two streams): the batch kernel's stream, the
relaxation's stream, and the host phases around them.  usage: step_gpu_times.py [N] [acceptors] [resident share 0/1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustfst_amd
from rustfst_amd import synth
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
ctx, ctx2 = rustfst_amd.Context(0, stream=s1.cuda_stream), rustfst_amd.Context(0, stream=s2.cuda_stream)
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
accs = synth.make_acceptors(t, B, 200, seed0=1000)
ctx.set_resident_share(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
dt2 = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx2)
daccs = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx2))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ev = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for it in range(N + 10):
    e0, e1, f0, f1 = ev(), ev(), ev(), ev()
    a = time.perf_counter()
    e0.record(s2)
    job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt2, ctx=ctx2)
    e1.record(s2)
    b = time.perf_counter()
    f0.record(s1)
    sp_job = dt.shortest_path_begin()
    f1.record(s1)
    c = time.perf_counter()
    outs, na = job.finish()
    d = time.perf_counter()
    sp = sp_job.finish()
    e = time.perf_counter()
    torch.cuda.synchronize()
    if it >= 10:
        rows.append([(b - a) * 1e6, (c - b) * 1e6, (d - c) * 1e6, (e - d) * 1e6, (e - a) * 1e6, e0.elapsed_time(e1) * 1e3, f0.elapsed_time(f1) * 1e3,
                     e0.elapsed_time(f0) * 1e3, e0.elapsed_time(f1) * 1e3])
r = np.median(np.array(rows), axis=0)
# each request alone, same events
al = []
for it in range(30):
    e0, e1, f0, f1 = ev(), ev(), ev(), ev()
    e0.record(s2); job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt2, ctx=ctx2); e1.record(s2); job.finish(); torch.cuda.synchronize()
    f0.record(s1); sp_job = dt.shortest_path_begin(); f1.record(s1); sp_job.finish(); torch.cuda.synchronize()
    al.append([e0.elapsed_time(e1) * 1e3, f0.elapsed_time(f1) * 1e3])
al = np.median(np.array(al[5:]), axis=0)
print("alone: batch stream %.1f us | relaxation stream %.1f us   (%d acceptors, resident share %s)" % (al[0], al[1], B, sys.argv[3] if len(sys.argv) > 3 else "0"))
print("host: batch begin %.1f | sp begin %.1f | batch finish %.1f | sp finish %.1f | step %.1f us" % tuple(r[:5]))
print("gpu : batch stream %.1f us | relaxation stream %.1f us | relaxation starts %.1f us after the batch | relaxation ends %.1f us after the batch started" % tuple(r[5:]))
