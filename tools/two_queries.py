"""Two shortest_path(T) queries in flight at once on ONE GPU: two contexts with wfst_ctx_set_resident_share(ctx, 1) (half the device
each: the resident lease has two units), two handles of T with different start states, begin / begin / end / end from one host
thread — against the same two queries one after the other on a whole-device context.  Prints ms per PAIR and queries per second.
   python tools/two_queries.py [states] [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

states = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t = synth.make_transducer(states, 10, 256, 0.0, seed=3)
src2 = (int(t["start"]) + 104729) % states


TWO_COPIES = os.environ.get("TWO_COPIES") == "1"  # a second copy of T with another start state (2 x 400 MB: more than the Infinity Cache holds)


def handles(ctx_a, ctx_b):
    a = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx_a)
    if not TWO_COPIES:
        return a, a  # the same handle queried from two contexts (the same source twice: what is measured is the overlap)
    b = rustfst_amd.DeviceFst.from_arrays(t["n_states"], src2, t["offsets"], t["arcs"], t["finals"], t["props"], ctx_b)
    return a, b


def flat_key(f):
    fl = f.to_flat()
    return fl["arcs"].tobytes() + fl["finals"].tobytes()


# ---- one after the other, whole device
c0 = rustfst_amd.Context(0)
a0, b0 = handles(c0, c0)
for _ in range(5):
    ra, rb = a0.shortest_path(), b0.shortest_path()
ref = (flat_key(ra), flat_key(rb))
c0.synchronize()
t0 = time.perf_counter()
for _ in range(pairs):
    a0.shortest_path()
    b0.shortest_path()
seq = (time.perf_counter() - t0) / pairs
print(f"one after the other (whole device each): {1e3 * seq:.4f} ms per pair = {2 / seq:,.0f} queries/s; kernel {c0.stats()['relax_kernel']}")
del a0, b0

# ---- two at once, half the device each
c1, c2 = rustfst_amd.Context(0), rustfst_amd.Context(0)
c1.set_resident_share(1)
c2.set_resident_share(1)
a1, b1 = handles(c1, c2)
for _ in range(5):
    ja, jb = a1.shortest_path_begin(ctx=c1), b1.shortest_path_begin(ctx=c2)
    ra, rb = ja.finish(), jb.finish()
assert (flat_key(ra), flat_key(rb)) == ref, "the paths of the concurrent solves differ from the sequential ones"
k1, k2 = c1.stats()["relax_kernel"], c2.stats()["relax_kernel"]
c1.synchronize(); c2.synchronize()
t0 = time.perf_counter()
for _ in range(pairs):
    ja, jb = a1.shortest_path_begin(ctx=c1), b1.shortest_path_begin(ctx=c2)
    ja.finish()
    jb.finish()
par = (time.perf_counter() - t0) / pairs
ab = c1.stats()["resident_aborts"] + c2.stats()["resident_aborts"]
print(f"two at once (half the device each):      {1e3 * par:.4f} ms per pair = {2 / par:,.0f} queries/s; kernels {k1} / {k2}, resident aborts {ab}; paths identical")
bytes_pair = 2 * (20 * len(t['arcs']) + 12 * states)
print(f"algorithmic bytes of a pair {bytes_pair / 1e6:.0f} MB: {bytes_pair / par / 1e9:.0f} GB/s over the whole pair (launches, tails and host included) "
      f"against {bytes_pair / seq / 1e9:.0f} GB/s one after the other")
