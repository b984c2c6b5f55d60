"""Randomised differential soak of the operations around the path: rm_epsilon, connect, project — GPU vs the CPU oracle on
seeded FSTs (epsilon-rich, cyclic and acyclic, missing start / finals).  python tools/soak_ops.py [seconds] [seed0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustfst_amd
from rustfst_amd import ProjectType
from oracle import oracle_py as O
from helpers import assert_flat_identical, random_fst_flat, to_device, to_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
n = dict(rm_epsilon=0, connect=0, project=0)
while time.time() < t_end:
    rng = np.random.default_rng(70_000 + seed)
    f = random_fst_flat(rng, int(rng.integers(1, 60)), int(rng.integers(1, 5)), int(rng.integers(1, 5)),
                        p_eps_i=rng.random() * 0.7, p_eps_o=rng.random() * 0.7, p_final=rng.random() * 0.5,
                        acyclic=bool(rng.integers(0, 3) == 0), sort=("none", "ilabel", "olabel")[seed % 3],
                        weight_grid=512 if seed % 4 else 2, max_w=2560 if seed % 4 else 6)
    if seed % 13 == 7:
        f = dict(f); f["start"] = -1
    try:
        ref = to_oracle(O, f); ref.rm_epsilon()
        assert_flat_identical(to_device(f).rm_epsilon().to_flat(), ref.to_flat(), "rm_epsilon")
        n["rm_epsilon"] += 1
        ref = to_oracle(O, f); ref.connect()
        assert_flat_identical(to_device(f).connect().to_flat(), ref.to_flat(), "connect")
        n["connect"] += 1
        out = bool(seed & 1)
        assert_flat_identical(to_device(f).project(ProjectType.PROJECT_OUTPUT if out else ProjectType.PROJECT_INPUT).to_flat(),
                              to_oracle(O, f).project(out).to_flat(), "project")
        n["project"] += 1
    except Exception:
        print("FAILED at seed", seed, flush=True)
        raise
    seed += 1
print("soak_ops OK:", n, "next seed", seed)
