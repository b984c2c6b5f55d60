"""Look-ahead composition (row A12): GPU kernel vs the CPU oracle on synthetic HCL-like (output epsilons) o G-like pairs.
usage: python tools/lookahead_timing.py [n1,n2,fan1,fan2,sigma ...]
fst2 has about as many arcs per state as there are labels, so most labels match somewhere and the composition is large."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch, rustfst_amd
from rustfst_amd import synth
from oracle import oracle_py
from helpers import to_oracle


def swap_labels(t):
    arcs = t["arcs"].copy()
    arcs["ilabel"], arcs["olabel"] = t["arcs"]["olabel"].copy(), t["arcs"]["ilabel"].copy()
    off = t["offsets"]
    key = np.repeat(np.arange(t["n_states"], dtype=np.int64), np.diff(off).astype(np.int64)) * (1 << 32) + arcs["olabel"].astype(np.int64)
    arcs = arcs[np.argsort(key, kind="stable")]
    out = dict(t)
    out["arcs"] = arcs
    out["props"] = synth.O_LABEL_SORTED
    return out


cases = [(300, 20, 3, 8, 8), (2000, 50, 3, 12, 12), (10000, 100, 3, 16, 16), (50000, 200, 3, 16, 16)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
ctx = rustfst_amd.default_context()
print("    n1     n2  states    arcs  levels?  prep_ms  relabel_ms  gpu_ms   cpu_ms  plain_gpu_ms(states)  plain+connect_ms(states)")
for n1, n2, fan1, fan2, sigma in cases:
    a = swap_labels(synth.make_transducer(n1, fan1, sigma, 0.2, seed=1, p_final=0.05))
    b = synth.make_transducer(n2, fan2, sigma, 0.05, seed=2, p_final=0.05)
    da, db = rustfst_amd.DeviceFst.from_arrays(a["n_states"], a["start"], a["offsets"], a["arcs"], a["finals"], a["props"], ctx), \
        rustfst_amd.DeviceFst.from_arrays(b["n_states"], b["start"], b["offsets"], b["arcs"], b["finals"], b["props"], ctx)
    t0 = time.perf_counter(); la = rustfst_amd.LookAhead(da); t1 = time.perf_counter()
    d2 = la.relabel(db); t2 = time.perf_counter()
    out = la.compose(d2)  # warm-up (arena growth retries included)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    la_ms = plain_ms = float("inf")
    for _ in range(3):  # best of three (a pool that has to go back to hipMalloc shows up as a 2x outlier)
        t3 = time.perf_counter()
        out = la.compose(d2)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        la_ms = min(la_ms, (t4 - t3) * 1e3)
    st = ctx.stats()
    plain = da.compose(db, rustfst_amd.ComposeConfig(connect=False)); torch.cuda.synchronize()  # warm-up (pool growth)
    for _ in range(3):
        t5 = time.perf_counter(); plain = da.compose(db, rustfst_amd.ComposeConfig(connect=False)); torch.cuda.synchronize(); t6 = time.perf_counter()
        plain_ms = min(plain_ms, (t6 - t5) * 1e3)
    conn_ms = float("inf")
    for _ in range(3):  # the default compose(): connect on the device after the wide driver
        t9 = time.perf_counter(); trimmed = da.compose(db); torch.cuda.synchronize(); conn_ms = min(conn_ms, (time.perf_counter() - t9) * 1e3)
    cpu_ms = float("nan")
    if n1 * n2 <= 3_000_000:
        oa, ob = to_oracle(oracle_py, a), to_oracle(oracle_py, b)
        t7 = time.perf_counter(); ref = oa.compose_lookahead(ob); t8 = time.perf_counter()
        cpu_ms = (t8 - t7) * 1e3
        f1, f2 = out.to_flat(), ref.to_flat()
        assert f1["n_states"] == f2["n_states"] and np.array_equal(f1["arcs"], f2["arcs"]) and np.array_equal(f1["finals"].view(np.uint32), f2["finals"].view(np.uint32))
    print(f"{n1:6d} {n2:6d} {out.num_states:7d} {out.num_arcs:7d} {'':7s} {(t1-t0)*1e3:8.1f} {(t2-t1)*1e3:10.1f} {la_ms:7.2f} {cpu_ms:8.1f}  {plain_ms:7.2f} ({plain.num_states})  {conn_ms:7.2f} ({trimmed.num_states})", flush=True)
