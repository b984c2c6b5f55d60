"""Does the relaxation chain's device time depend on WHICH handle of the same T is solved?  (bench.py's line reads 266 us on its
long-lived handle where roofline_vs_size reads 285 us on a fresh T of the same generator, same process, same HIP-event bracket.)
Prints the chain's median for handle A (uploaded first), for handles B and C uploaded later, and for A again."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustfst_amd
from rustfst_amd import synth

t = synth.make_transducer(1_000_000, 10, 256, 0.0, seed=3)
ctx = rustfst_amd.Context(0)
up = lambda: rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)


def chain(d, n=21):
    for _ in range(6):
        d.shortest_path()
    ctx.set_profiling(2)
    v = []
    for _ in range(n):
        d.shortest_path()
        st = ctx.stats()
        if st["relax_launches"]:
            v.append(st["relax_ms"] * 1e3)
    ctx.set_profiling(0)
    return statistics.median(v), min(v)


a = up()
print("A (first upload)      median %.1f min %.1f us" % chain(a))
junk = [up() for _ in range(3)]
b = up()
print("B (after 3 more T's)  median %.1f min %.1f us" % chain(b))
del junk
c = up()
print("C (after freeing them) median %.1f min %.1f us" % chain(c))
print("A again               median %.1f min %.1f us" % chain(a))
print("B again               median %.1f min %.1f us" % chain(b))
