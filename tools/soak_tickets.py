"""Soak of the ticket hand-over (results read from pinned memory as soon as the kernels' tickets arrive): N overlapped steps,
every result of every step compared with the first step's.  usage: soak_tickets.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rustfst_amd
from rustfst_amd import synth, dist
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ctx, ctx2 = rustfst_amd.Context(0, stream=s1.cuda_stream), rustfst_amd.Context(0, stream=s2.cuda_stream)
t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
accs = synth.make_acceptors(t, 64, 200, seed0=1000)
dt = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
dt2 = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx2)
daccs = rustfst_amd.HandleArray(rustfst_amd.DeviceFst.upload_many(accs, ctx2))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
want_rec = want_sp = None
bad = 0
t0 = time.time()
for it in range(N):
    sp_job = dt.shortest_path_begin()
    job = rustfst_amd.compose_shortest_path_batch_begin(daccs, dt2, ctx=ctx2)
    outs, na = job.finish()
    sp = sp_job.finish()
    rec = dist.pack_device_paths(outs, 208)
    spf = sp.to_flat()
    key = (spf["n_states"], spf["arcs"].tobytes(), spf["finals"].tobytes(), spf["props"])
    if want_rec is None:
        want_rec, want_sp = rec.copy(), key
        props0 = [outs[k].properties for k in range(64)]
        flats0 = [outs[k].to_flat() for k in range(0, 64, 7)]
    else:
        if not np.array_equal(rec, want_rec) or key != want_sp:
            bad += 1
        if it % 97 == 0:
            if [outs[k].properties for k in range(64)] != props0: bad += 1
            for j, k in enumerate(range(0, 64, 7)):
                f = outs[k].to_flat()
                if f["arcs"].tobytes() != flats0[j]["arcs"].tobytes() or f["props"] != flats0[j]["props"]: bad += 1
print(f"{N} steps in {time.time() - t0:.1f} s, {bad} mismatching steps")
sys.exit(1 if bad else 0)
