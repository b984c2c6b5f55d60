"""Builds libwfst_amd.so (hand-written HIP for gfx950 + the C-ABI) in-tree with hipcc.

No torch / pybind dependency: the product is a plain C-ABI shared library (include/wfst.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libwfst_amd.so")
SOURCES = ["api.cpp", "vector_fst.cpp", "fst_store.hip", "openfst_io.cpp", "sssp.hip", "nshortest.hip", "nbest_batch.hip", "tr_sort.hip", "compose.hip", "lookahead.cpp", "compose_lookahead.hip", "compose_wide.hip", "rm_epsilon.hip", "gather.cpp"]
HEADERS = ["common.h", "sssp_mailbox.h", "sssp_resident.h", "sssp_binned.h", "fst_props.h", "lookahead.h", "compose_wide.h", "compose_filters.h", "host_parallel.h", os.path.join("..", "..", "include", "wfst.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(LIB_DIR, s.replace(".", "_") + ".o")
        cmd = [_hipcc(), "-x", "hip", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               "-Wall", "-Wno-unused-function", "-c", os.path.join(CSRC, s), "-o", obj]
        if os.environ.get("WFST_PHASE_TIMING") == "1":
            cmd.insert(-4, "-DWFST_PHASE_TIMING")
        for d in os.environ.get("WFST_CXX_DEFS", "").split():  # experiments: e.g. -DWFST_MB_THREADS=512
            cmd.insert(-4, d)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {s} ---\n{out.decode()}\n")
        elif verbose and out:
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("libwfst_amd build failed")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
