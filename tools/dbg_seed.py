import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustfst_amd
from rustfst_amd import synth
from oracle import oracle_py as O
from helpers import random_fst_flat, to_device, to_oracle
seed = int(sys.argv[1])
rng = np.random.default_rng(77_000 + seed)
mode = ["0", "1", "2"][int(rng.integers(0, 3))]
rounds = str(int(rng.choice([1, 2, 5, 16384])))
d = rng.choice(["", "0", "0.3", "2.5", "40"])
kind = int(rng.integers(0, 3))
assert kind == 0
f = random_fst_flat(rng, int(rng.integers(1, 6000)), int(rng.integers(1, 9)), 5, p_eps_i=0.1, p_final=rng.random() * 0.3,
                    sort="ilabel", acyclic=bool(rng.integers(0, 2)), weight_grid=int(rng.choice([512, 7, 1])), max_w=int(rng.choice([3, 12, 5000])))
print("mode", mode, "delta", d, "n", f["n_states"], "arcs", len(f["arcs"]), "weights", np.unique(f["arcs"]["weight"])[:8])
orc = to_oracle(O, f)
ref = orc.shortest_path_canonical()
ctx = rustfst_amd.Context(0)
for m in (sys.argv[2:] or [mode]):
    os.environ["WFST_SSSP_MAILBOX"] = m
    for dd in ([d] if len(sys.argv) < 3 else ["", "0", "2.5"]):
        if dd: os.environ["WFST_SSSP_DELTA"] = str(dd)
        else: os.environ.pop("WFST_SSSP_DELTA", None)
        dev = to_device(f, ctx)
        for q in range(3):
            dist, hops = dev.shortest_distance(want_hops=True)
            bad = np.nonzero((hops != ref.hops) | (dist.view(np.uint32) != np.asarray(ref.distance, np.float32).view(np.uint32)))[0]
            print(f"mailbox={m} delta={dd!r} q={q}: {len(bad)} bad; sweeps {ctx.stats()['sweeps']}", [(int(b), float(dist[b]), int(hops[b]), float(ref.distance[b]), int(ref.hops[b])) for b in bad[:6]])
