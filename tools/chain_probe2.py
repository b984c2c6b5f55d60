"""Follow-up of chain_probe.py: bench.py's `roofline` block runs AFTER its extras (config5: multi-GB arenas; roofline_vs_size: a
16M-state T solved on the same context).  Does the chain of the 1M-state T read lower after such work has gone through the pool?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth
from chain_probe import chain, up, ctx, d  # (runs chain_probe's own measurements first)

for n in (2_000_000, 5_000_000, 16_000_000):
    tb = synth.make_transducer(n, 10, 256, 0.0, seed=3)
    db = rustfst_amd.DeviceFst.from_arrays(tb["n_states"], tb["start"], tb["offsets"], tb["arcs"], tb["finals"], tb["props"], ctx)
    del tb
    for _ in range(6):
        db.shortest_path()
    del db
    print(f"1M-state T after a {n // 1000000}M-state T was solved and dropped :", chain(ctx, d))
d_new = up(ctx)
for _ in range(10):
    d_new.shortest_path()
print("a NEW handle of the 1M-state T now                   :", chain(ctx, d_new))
