"""ctypes binding of libwfst_amd.so (include/wfst.h).  Fails loudly when the library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwfst_amd.so")

TR_DTYPE = np.dtype([("ilabel", "<u4"), ("olabel", "<u4"), ("weight", "<f4"), ("nextstate", "<u4")])


class WfstError(RuntimeError):
    """KO status from the C-ABI; carries the thread-local error text (rustfst-ffi/src/lib.rs:58-76)."""


class ComposeConfig(C.Structure):
    _fields_ = [("compose_filter", C.c_uint32), ("connect", C.c_uint32)]


class ShortestPathConfig(C.Structure):
    _fields_ = [("delta", C.c_float), ("nshortest", C.c_uint64), ("unique", C.c_uint32)]


TIES_UNKNOWN = (1 << 64) - 1  # WFST_TIES_UNKNOWN


class Stats(C.Structure):
    _fields_ = [("relax_launches", C.c_uint64), ("relax_ms", C.c_double), ("relax_arcs", C.c_uint64),
                ("relax_states", C.c_uint64), ("sweeps", C.c_uint64), ("compose_states", C.c_uint64),
                ("compose_arcs", C.c_uint64), ("compose_retries", C.c_uint64), ("compose_ms", C.c_double),
                ("string_problems", C.c_uint64), ("relax_kernel", C.c_uint64), ("nbest_device_problems", C.c_uint64),
                ("resident_aborts", C.c_uint64), ("tied_choices", C.c_uint64)]


# every symbol include/wfst.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _i64, _f32, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_float, C.c_size_t
_P = C.POINTER
SYMBOLS = [
    ("wfst_last_error", C.c_int, [_P(C.c_char_p)]),
    ("wfst_string_destroy", C.c_int, [C.c_char_p]),
    ("wfst_abi_version", _u32, []),
    ("wfst_ctx_create", C.c_int, [C.c_int, _P(_vp)]),
    ("wfst_ctx_create_on_stream", C.c_int, [C.c_int, _vp, _P(_vp)]),
    ("wfst_ctx_create_with_cu_mask", C.c_int, [C.c_int, _vp, _u32, _P(_vp)]),
    ("wfst_ctx_destroy", C.c_int, [_vp]),
    ("wfst_ctx_synchronize", C.c_int, [_vp]),
    ("wfst_ctx_stream", C.c_int, [_vp, _P(_vp)]),
    ("wfst_fst_upload", C.c_int, [_vp, _u32, _i64, _vp, _vp, _vp, _u64, _P(_vp)]),
    ("wfst_fst_upload_device", C.c_int, [_vp, _u32, _i64, _vp, _vp, _vp, _u64, _P(_vp)]),
    ("wfst_fst_upload_many", C.c_int, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _P(_vp)]),
    ("wfst_fst_from_openfst_bytes", C.c_int, [_vp, C.c_char_p, _sz, _P(_vp)]),
    ("wfst_fst_to_openfst_bytes", C.c_int, [_vp, _P(_vp), _P(_sz)]),
    ("wfst_fst_to_openfst_const_bytes", C.c_int, [_vp, _P(_vp), _P(_sz)]),
    ("wfst_bytes_destroy", C.c_int, [_vp]),
    ("wfst_fst_info", C.c_int, [_vp, _P(_u32), _P(_u64), _P(_i64), _P(_u64)]),
    ("wfst_fst_download", C.c_int, [_vp, _vp, _vp, _vp]),
    ("wfst_fst_destroy", C.c_int, [_vp]),
    ("wfst_fst_destroy_many", C.c_int, [_P(_vp), _sz]),
    ("wfst_compose", C.c_int, [_vp, _vp, _vp, _P(ComposeConfig), _P(_vp)]),
    ("wfst_shortest_path", C.c_int, [_vp, _vp, _P(ShortestPathConfig), _P(_vp)]),
    ("wfst_connect", C.c_int, [_vp, _vp, _P(_vp)]),
    ("wfst_rm_epsilon", C.c_int, [_vp, _vp, _P(_vp)]),
    ("wfst_fst_project", C.c_int, [_vp, _vp, C.c_int]),
    ("wfst_lookahead_create", C.c_int, [_vp, _vp, _P(_vp)]),
    ("wfst_lookahead_relabel", C.c_int, [_vp, _vp, _P(_vp)]),
    ("wfst_lookahead_fst1", C.c_int, [_vp, _P(_vp)]),
    ("wfst_compose_lookahead", C.c_int, [_vp, _vp, _vp, _P(_vp)]),
    ("wfst_compose_lookahead_batch", C.c_int, [_vp, _vp, _P(_vp), _sz, _P(_vp)]),
    ("wfst_lookahead_destroy", C.c_int, [_vp]),
    ("wfst_lookahead_info", C.c_int, [_vp, _P(_u32), _P(_u64), _P(_u32), _P(_u32)]),
    ("wfst_lookahead_download", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("wfst_label_reachable_compute", C.c_int, [_u32, _vp, _vp, _vp, C.c_int, _P(_vp)]),
    ("wfst_ctx_set_tie_order", C.c_int, [_vp, C.c_int]),
    ("wfst_ctx_set_resident_share", C.c_int, [_vp, _u32]),
    ("wfst_shortest_path_batch", C.c_int, [_vp, _P(_vp), _sz, _P(ShortestPathConfig), _P(_vp)]),
    ("wfst_shortest_path_begin", C.c_int, [_vp, _vp, _P(ShortestPathConfig), _P(_vp)]),
    ("wfst_shortest_path_end", C.c_int, [_vp, _P(_vp)]),
    ("wfst_shortest_distance", C.c_int, [_vp, _vp, _vp, _vp]),
    ("wfst_compose_shortest_path_batch", C.c_int,
     [_vp, _P(_vp), _sz, _vp, _P(ComposeConfig), _P(ShortestPathConfig), _P(_vp), _P(_u64)]),
    ("wfst_compose_shortest_path_batch_begin", C.c_int,
     [_vp, _P(_vp), _sz, _vp, _P(ComposeConfig), _P(ShortestPathConfig), _P(_vp)]),
    ("wfst_compose_shortest_path_batch_end", C.c_int, [_vp, _P(_vp), _P(_u64)]),
    ("wfst_fst_pack_paths", C.c_int, [_P(_vp), _sz, _u32, _vp]),
    ("wfst_compose_shortest_path_batch_packed", C.c_int,
     [_vp, _P(_vp), _sz, _vp, _P(ComposeConfig), _P(ShortestPathConfig), _u32, _vp, _P(_u64)]),
    ("wfst_comm_unique_id", C.c_int, [_vp]),
    ("wfst_comm_create", C.c_int, [_vp, _vp, _u32, _u32, _P(_vp)]),
    ("wfst_comm_create_host", C.c_int, [_u32, _u32, _vp, _vp, _P(_vp)]),
    ("wfst_gather_records_begin", C.c_int, [_vp, _vp, _sz, _u32]),
    ("wfst_comm_info", C.c_int, [_vp, _P(_u32), _P(_u32)]),
    ("wfst_comm_destroy", C.c_int, [_vp]),
    ("wfst_comm_order_after", C.c_int, [_vp, _vp]),
    ("wfst_gather_paths_begin", C.c_int, [_vp, _P(_vp), _sz, _u32]),
    ("wfst_gather_paths_end", C.c_int, [_vp, _vp]),
    ("wfst_comm_allgather_begin", C.c_int, [_vp, _vp, _sz]),
    ("wfst_comm_allgather_end", C.c_int, [_vp, _vp]),
    ("wfst_comm_allgatherv", C.c_int, [_vp, _vp, _sz, _vp, _P(_vp), _P(_sz)]),
    ("wfst_fst_tr_sort", C.c_int, [_vp, _vp, C.c_int]),
    ("wfst_fst_set_start", C.c_int, [_vp, _vp, C.c_uint32]),
    ("wfst_reverse", C.c_int, [_vp, _vp, _P(_vp)]),
    ("wfst_vec_fst_new", C.c_int, [_P(_vp)]),
    ("wfst_vec_fst_destroy", C.c_int, [_vp]),
    ("wfst_vec_fst_copy", C.c_int, [_vp, _P(_vp)]),
    ("wfst_vec_fst_add_state", C.c_int, [_vp, _P(_u32)]),
    ("wfst_vec_fst_add_tr", C.c_int, [_vp, _u32, _vp]),
    ("wfst_vec_fst_set_start", C.c_int, [_vp, _u32]),
    ("wfst_vec_fst_set_final", C.c_int, [_vp, _u32, _f32]),
    ("wfst_vec_fst_del_final_weight", C.c_int, [_vp, _u32]),
    ("wfst_vec_fst_num_states", C.c_int, [_vp, _P(_u32)]),
    ("wfst_vec_fst_start", C.c_int, [_vp, _P(_i64)]),
    ("wfst_vec_fst_final_weight", C.c_int, [_vp, _u32, _P(_f32), _P(C.c_int)]),
    ("wfst_vec_fst_num_trs", C.c_int, [_vp, _u32, _P(_u64)]),
    ("wfst_vec_fst_get_trs", C.c_int, [_vp, _u32, _vp, _u64, _P(_u64)]),
    ("wfst_vec_fst_properties", C.c_int, [_vp, _P(_u64)]),
    ("wfst_vec_fst_tr_sort", C.c_int, [_vp, C.c_int]),
    ("wfst_vec_fst_equals", C.c_int, [_vp, _vp, _P(C.c_int)]),
    ("wfst_vec_fst_to_device", C.c_int, [_vp, _vp, _P(_vp)]),
    ("wfst_vec_fst_from_device", C.c_int, [_vp, _P(_vp)]),
    ("wfst_ctx_set_profiling", C.c_int, [_vp, C.c_int]),
    ("wfst_ctx_get_stats", C.c_int, [_vp, _P(Stats)]),
    ("wfst_ctx_reset_stats", C.c_int, [_vp]),
    ("wfst_ctx_get_sweep_trace", C.c_int, [_vp, _vp, _vp, _vp, _sz, _P(_sz)]),
    ("wfst_ctx_get_sweep_modes", C.c_int, [_vp, _vp, _sz, _P(_sz)]),
]

_lib = None


def lib():
    """Loads libwfst_amd.so. Raises if it has not been built (python -m rustfst_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m rustfst_amd.build` "
                "(hipcc --offload-arch=gfx950). rustfst_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME
        # libamdhip64.so.7, requested by file name).  If the system copy were loaded first, torch would load
        # a second runtime and then see no GPU; loaded in this order the dynamic linker resolves our
        # NEEDED libamdhip64.so.7 to the copy torch already mapped.
        if os.environ.get("WFST_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        # (experiments only: WFST_LIB_PATH points tools/ A/B scripts at another build of the same ABI)
        L = C.CDLL(os.environ.get("WFST_LIB_PATH") or LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what=""):
    """check_ffi_error (rustfst-python/rustfst/ffi_utils.py): raise with the library's message."""
    if status != 0:
        msg = C.c_char_p()
        L = lib()
        text = "unknown error"
        if L.wfst_last_error(C.byref(msg)) == 0 and msg.value is not None:
            text = msg.value.decode(errors="replace")
            L.wfst_string_destroy(msg)
        raise WfstError(f"{what}: {text}" if what else text)
