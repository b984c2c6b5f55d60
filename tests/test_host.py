"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol that
include/wfst.h declares, the host VectorFst mirror does the reference's property bookkeeping (checked
against the oracle's restatement), the workload generator is deterministic, and the engine refuses to
run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import rustfst_amd
from rustfst_amd import Tr, VectorFst, _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wfst.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wfst_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(wfst_lib):
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(wfst_lib, n), f"libwfst_amd.so does not export {n}"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(names) == bound, f"ctypes binding out of sync with wfst.h: {set(names) ^ bound}"
    assert wfst_lib.wfst_abi_version() == 7


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "wfst.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "hipStream_t" not in text


def test_last_error_protocol(wfst_lib):
    import ctypes as C
    assert wfst_lib.wfst_vec_fst_set_start(None, 0) == 1  # KO
    msg = C.c_char_p()
    assert wfst_lib.wfst_last_error(C.byref(msg)) == 0
    assert b"null" in msg.value
    wfst_lib.wfst_string_destroy(msg)
    assert wfst_lib.wfst_last_error(C.byref(msg)) == 0
    assert msg.value == b"No error message"  # taken once, like rustfst_ffi_get_last_error (lib.rs:64-67)
    wfst_lib.wfst_string_destroy(msg)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_engine_fails_loudly_without_gpu(wfst_lib):
    with pytest.raises(rustfst_amd.WfstError, match="no HIP device|no CPU fallback|HIP error"):
        rustfst_amd.Context(0)


def test_vector_fst_mirror_matches_oracle_bookkeeping(oracle, wfst_lib):
    rng = np.random.default_rng(7)
    for trial in range(30):
        n = int(rng.integers(1, 8))
        v, o = VectorFst(), oracle.OracleFst()
        for _ in range(n):
            assert v.add_state() == o.add_state()
        for _ in range(int(rng.integers(0, 15))):
            s, ns = int(rng.integers(0, n)), int(rng.integers(0, n))
            il, ol = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            w = float(rng.integers(0, 4)) / 2048.0 if rng.random() < 0.5 else float(rng.integers(0, 20)) / 4
            v.add_tr(s, Tr(il, ol, w, ns))
            o.add_tr(s, il, ol, w, ns)
            assert v.properties() == o.properties
        if rng.random() < 0.8:
            s = int(rng.integers(0, n))
            v.set_start(s)
            o.set_start(s)
        for _ in range(int(rng.integers(0, 3))):
            s, w = int(rng.integers(0, n)), float(rng.integers(0, 8)) / 4
            v.set_final(s, w)
            o.set_final(s, w)
        assert v.properties() == o.properties
        assert v.num_states() == o.num_states and v.start() == o.start
        by_ol = bool(rng.integers(0, 2))
        v.tr_sort(ilabel_cmp=not by_ol)
        o.tr_sort(by_olabel=by_ol)
        assert v.properties() == o.properties
        fo = o.to_flat()
        got = []
        for s in range(n):
            got.extend((t.ilabel, t.olabel, np.float32(t.weight), t.next_state) for t in v.trs(s))
        exp = [(int(a["ilabel"]), int(a["olabel"]), a["weight"], int(a["nextstate"])) for a in fo["arcs"]]
        assert got == exp


def test_vector_fst_errors_mirror_reference(wfst_lib):
    f = VectorFst()
    with pytest.raises(rustfst_amd.WfstError, match="doesn't exist"):
        f.set_start(3)  # mutable_fst.rs:36-40
    with pytest.raises(rustfst_amd.WfstError, match="doesn't exist"):
        f.add_tr(0, Tr(1, 1, 0.0, 0))
    a, b = rustfst_amd.acceptor([1, 2, 3]), rustfst_amd.acceptor([1, 2, 3])
    assert a == b and a.copy() == a
    b.set_final(3, 0.5)
    assert a != b
    b.set_final(3, 0.0005)  # within KDELTA: TropicalWeight == is approximate (semiring.rs:159-168)
    assert a == b


def test_synth_generator_is_deterministic_and_well_formed(wfst_lib):
    t1 = synth.make_transducer(2000, 10, 256, 0.0, seed=3)
    t2 = synth.make_transducer(2000, 10, 256, 0.0, seed=3)
    for k in ("offsets", "finals"):
        np.testing.assert_array_equal(t1[k], t2[k])
    np.testing.assert_array_equal(t1["arcs"], t2["arcs"])
    arcs, off = t1["arcs"], t1["offsets"]
    il = arcs["ilabel"].reshape(2000, 10).astype(np.int64)
    assert np.all(np.diff(il, axis=1) >= 0)  # ilabel-sorted
    assert np.all(arcs["nextstate"] < 2000)
    w512 = arcs["weight"].astype(np.float64) * 512
    assert np.all(w512 == np.round(w512)) and w512.max() < 5120  # the 1/512 grid
    # ring backbone: every state has an arc to s+1
    ns = arcs["nextstate"].reshape(2000, 10)
    assert all(((s + 1) % 2000) in ns[s] for s in range(0, 2000, 97))
    # epsilons appear with p_eps, stay first in each state
    te = synth.make_transducer(2000, 10, 256, 0.05, seed=5)
    frac = np.mean(te["arcs"]["ilabel"] == 0)
    assert 0.02 < frac < 0.08
    accs = synth.make_acceptors(t1, 3, 20, seed0=1000)
    assert accs[0]["n_states"] == 21 and accs[0]["props"] == synth.acceptor_props(20)
    assert not np.array_equal(accs[0]["arcs"]["ilabel"], accs[1]["arcs"]["ilabel"])


def test_acceptor_matches_reference_utils(oracle, wfst_lib):
    """utils::acceptor (labels_to_fst.rs:111-132) built through the mirror, the oracle, and the flat helper agree."""
    labels = [5, 9, 2, 2]
    v = rustfst_amd.acceptor(labels)
    o = oracle.OracleFst()
    cur = o.add_state()
    o.set_start(cur)
    for l in labels:
        nxt = o.add_state()
        o.add_tr(cur, l, l, 0.0, nxt)
        cur = nxt
    o.set_final(cur, 0.0)
    flat = synth.linear_acceptor_flat(labels)
    assert v.properties() == o.properties == flat["props"]
    fo = o.to_flat()
    np.testing.assert_array_equal(fo["offsets"], flat["offsets"])
    np.testing.assert_array_equal(fo["arcs"], flat["arcs"])
    np.testing.assert_array_equal(fo["finals"], flat["finals"])


def test_header_is_plain_c_and_the_c_example_links(tmp_path, wfst_lib):
    """include/wfst.h is a self-contained C99 header (no C++, no HIP, no torch types) and examples/decode_batch.c —
    the drop-in boundary used from plain C — compiles and links against the library."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "wfst.h"\nint main(void) { wfst_tr t = {1, 2, 0.5f, 3}; (void)t; return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)],
                   check=True)
    exe = tmp_path / "decode_batch"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-I", inc, os.path.join(ROOT, "examples", "decode_batch.c"),
                    "-L", libdir, "-lwfst_amd", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined", "-lm", "-o", str(exe)],
                   check=True)
    assert exe.exists()


def test_cpp_mirror_header_compiles_and_links(tmp_path, wfst_lib):
    """include/wfst.hpp (the C++17 mirror of the reference's Rust interface for this path) and the reference-style test
    program built on it compile warning-free and link against the library."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "ref_tests"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "reference_style_tests.cpp"), "-L", libdir, "-lwfst_amd",
                    f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined", "-o", str(exe)], check=True)
    assert exe.exists()


@pytest.mark.parametrize("threads", ["4", "seq"])
@pytest.mark.parametrize("seed", range(12))
def test_label_reachable_threads_and_sizes(wfst_lib, oracle, seed, threads, monkeypatch):
    """The parallel path of the host precompute (acyclic epsilon structure: index map from an abandoned replay of the
    reference's visit, interval sets in rounds over all host threads) and the sequential one (the reference's visit in
    full) against the oracle on inputs with a few thousand states, several threads forced on them."""
    import rustfst_amd
    from helpers import random_fst_flat, to_oracle
    if threads == "seq":
        monkeypatch.setenv("WFST_LOOKAHEAD_SEQUENTIAL", "1")
    else:
        monkeypatch.setenv("WFST_HOST_THREADS", threads)
    rng = np.random.default_rng(4200 + seed)
    f = random_fst_flat(rng, int(rng.integers(500, 4000)), 5, int(rng.choice([3, 40, 3000])), p_eps_i=0.1,
                        p_eps_o=float(rng.choice([0.05, 0.3, 0.7])), p_final=0.1, sort="olabel", acyclic=(seed % 4 != 3))
    for reach_input in (False, True):
        ref = to_oracle(oracle, f).label_reachable(reach_input)
        la = rustfst_amd.LookAhead.reachable_from_arrays(f["n_states"], f["offsets"], f["arcs"], f["finals"], reach_input)
        assert la.data() == ref


@pytest.mark.parametrize("seed", range(40))
def test_label_reachable_host_precompute_matches_oracle(wfst_lib, oracle, seed):
    """Product host code of look-ahead composition (rustfst_amd/csrc/lookahead.cpp: LabelReachable::compute_data on flat
    arrays, no GPU involved) against the oracle's restatement: same label -> index map (depth-first discovery order of
    the per-label sink states), same final label, same interval sets for every state — acyclic and cyclic epsilon
    structure (the latter through the condensation)."""
    import rustfst_amd
    from helpers import random_fst_flat, to_oracle
    rng = np.random.default_rng(900 + seed)
    f = random_fst_flat(rng, int(rng.integers(1, 60)), 4, 6, p_eps_i=0.2, p_eps_o=0.5, p_final=0.25, sort="olabel",
                        acyclic=(seed % 3 == 0))
    for reach_input in (False, True):
        ref = to_oracle(oracle, f).label_reachable(reach_input)
        la = rustfst_amd.LookAhead.reachable_from_arrays(f["n_states"], f["offsets"], f["arcs"], f["finals"], reach_input)
        assert la.data() == ref
        with pytest.raises(rustfst_amd.WfstError, match="host-only"):
            la.fst1
    bad = f["arcs"].copy()
    if len(bad):
        bad["nextstate"][0] = f["n_states"] + 3
        with pytest.raises(rustfst_amd.WfstError, match="does not exist"):
            rustfst_amd.LookAhead.reachable_from_arrays(f["n_states"], f["offsets"], bad, f["finals"])


def test_path_props_from_fact_union(tmp_path):
    """The property word of a batch result is computed from the OR of its arcs' facts (the string kernel gathers them while
    it writes the path; fst_props.h: linear_path_props_from_facts) instead of a walk over the arcs (linear_path_props, the
    reference's incremental add_state / add_tr bookkeeping): the two agree on every sequence of up to six arcs over the
    fact combinations a single arc can have (3.3 M cases, exhaustive)."""
    import shutil, subprocess
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "props_check"
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-I", os.path.join(root, "rustfst_amd", "csrc"), "-o", str(exe),
                           os.path.join(root, "tests", "props_check.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout
