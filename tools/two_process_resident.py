"""One of two PROCESSES that solve shortest_path(T) on the same GPU at the same time (tests/test_gpu_parity.py:
test_two_processes_solving_on_one_gpu).  usage: two_process_resident.py STATES SOLVES
Last line: "OK solves=N resident=R one_level=L aborts=A" — every result equal to the first, counts by the kernel that ran."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

states, solves = int(sys.argv[1]), int(sys.argv[2])
t = synth.make_transducer(states, 8, 64, 0.0, seed=17)
ctx = rustfst_amd.Context(0)
d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
ref = d.shortest_path().to_flat()
kinds = {}
for k in range(solves):
    got = d.shortest_path().to_flat()
    st = ctx.stats()
    kinds[int(st["relax_kernel"])] = kinds.get(int(st["relax_kernel"]), 0) + 1
    if not (np.array_equal(got["arcs"], ref["arcs"]) and np.array_equal(got["finals"], ref["finals"])):
        print(f"MISMATCH at solve {k}")
        sys.exit(1)
print(f"OK solves={solves} resident={kinds.get(2, 0)} one_level={kinds.get(1, 0)} aborts={int(ctx.stats()['resident_aborts'])}")
