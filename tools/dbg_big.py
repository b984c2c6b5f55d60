"""Diagnostic: distances of a > 2^20-state graph under the atomic sweeps vs the mailbox sweeps in several configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustfst_amd
from rustfst_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_100_000
fan = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t = synth.make_transducer(n, fan, 64, 0.0, seed=11)
ref = None
for cfg in ({"WFST_SSSP_MAILBOX": "0"}, {"WFST_SSSP_MAILBOX": "1", "WFST_SSSP_NARROW": "0"},
            {"WFST_SSSP_MAILBOX": "1", "WFST_SSSP_NARROW": "0", "WFST_SSSP_HINT": "0"},
            {"WFST_SSSP_MAILBOX": "1", "WFST_SSSP_NARROW": "0", "WFST_SSSP_STG": "1"},
            {"WFST_SSSP_MAILBOX": "1"}, {"WFST_SSSP_MAILBOX": "1", "WFST_SSSP_NARROW": "1000000000"}):
    for k in ("WFST_SSSP_MAILBOX", "WFST_SSSP_NARROW", "WFST_SSSP_HINT", "WFST_SSSP_STG"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    ctx = rustfst_amd.Context(0)
    d = rustfst_amd.DeviceFst.from_arrays(t["n_states"], t["start"], t["offsets"], t["arcs"], t["finals"], t["props"], ctx)
    for q in range(2):
        dist, hops = d.shortest_distance(want_hops=True)
        if ref is None:
            ref = (dist.copy(), hops.copy())
        bad = np.nonzero((dist.view(np.uint32) != ref[0].view(np.uint32)) | (hops != ref[1]))[0]
        print(cfg, "q", q, "kernel", ctx.stats()["relax_kernel"], "sweeps", ctx.stats()["sweeps"], "mismatches", bad.size, flush=True)
        if bad.size:
            blocks, counts = np.unique(bad >> 12, return_counts=True)
            print("   blocks", blocks[:20], counts[:20], "first states", bad[:10], "got", dist[bad[:5]], "want", ref[0][bad[:5]],
                  "worse" if (dist[bad] > ref[0][bad]).all() else "mixed")
