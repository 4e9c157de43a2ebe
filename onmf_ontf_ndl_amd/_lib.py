"""ctypes binding of libonmf_hip.so (the C-ABI declared in include/onmf_hip.h).

The shared library is built in-tree (``onmf_ontf_ndl_amd/csrc/Makefile`` or
``__graft_entry__.build()``).  There is no CPU fallback: if the library is missing, or no GPU is
present when a context is created, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ONMF_HIP_LIB: another build of the same library (diagnostic builds, A/B comparisons) without touching this one
LIB_PATH = os.environ.get("ONMF_HIP_LIB") or os.path.join(_HERE, "libonmf_hip.so")

OK = 0
CODER_DAMPED = 0
CODER_EXACT_CD = 1
OPT_PROFILE = 1
OPT_EXACT_STOP = 2
OPT_SWEEP_KERNEL = 3
OPT_OVERLAP_DICT = 4
OPT_SCREEN_STOP = 5
OPT_DICT_KERNEL = 6
OPT_PERSISTENT_SWEEPS = 7
OPT_SPIN_LIMIT_MS = 8
OPT_DEVICE_RESOLVE = 9
OPT_LARS_SLOTS = 10
OPT_DEFER_VERDICT = 11
OPT_GEMM_TAIL = 12
OPT_VERDICT_LAG3 = 13
CNT_SWEEP_TIMEOUTS = 1
CNT_DICT_TIMEOUTS = 2
CNT_FALLBACK_SOLVES = 3
CNT_DEVICE_RESOLVES = 4
CNT_LARS_RECODED = 5
CNT_LARS_SLOTS_USED = 6
CNT_SWEEPS_TOTAL = 7
CNT_DEFERRED_REDONE = 8
CNT_LAG3_REDONE = 9
NDL_GLAUBER = 0
NDL_PIVOT = 1

_f = C.c_void_p          # device float*
_ctx = C.c_void_p

# name -> (restype, argtypes); this table is also what the symbol-export test walks
SIGNATURES = {
    "onmf_abi_version": (C.c_int, []),
    "onmf_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "onmf_destroy": (C.c_int, [_ctx]),
    "onmf_last_error": (C.c_char_p, [_ctx]),
    "onmf_sparse_code": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, _f, C.c_uint64,
                                   C.c_float, C.c_int, C.c_float, C.c_int, _f,
                                   C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p]),
    "onmf_sparse_code_within_radius": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, _f, C.c_uint64,
                                                 C.c_float, C.c_float, C.c_int, C.c_float, _f,
                                                 C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p]),
    "onmf_aggregate": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, C.c_double,
                                 _f, _f, _f, _f, _f, _f, C.c_void_p]),
    "onmf_update_dict": (C.c_int, [_ctx, C.c_int, C.c_int, _f, _f, _f, _f, C.c_void_p]),
    "onmf_update_dict_static": (C.c_int, [_ctx, C.c_int, C.c_int, _f, _f, _f, _f, C.c_void_p]),
    "onmf_step": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, _f, _f, _f, _f,
                            C.c_uint64, C.c_double, C.c_float, C.c_int, C.c_float,
                            _f, _f, _f, _f, _f, C.POINTER(C.c_int), C.POINTER(C.c_float),
                            C.c_void_p]),
    "onmf_settle": (C.c_int, [_ctx]),
    "onmf_partial_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "onmf_step_partial": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, _f, _f, _f, C.c_uint64,
                                    C.c_float, C.c_int, C.c_float, _f, _f, _f, C.POINTER(C.c_int),
                                    C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    "onmf_step_apply": (C.c_int, [_ctx, C.c_int, C.c_int, _f, C.c_double, _f, _f, _f, _f, C.c_void_p]),
    "onmf_gather_columns": (C.c_int, [_ctx, C.c_int, C.c_int, _f, C.c_int, C.c_void_p, C.c_int, _f,
                                      C.c_void_p]),
    "onmf_code_accumulate": (C.c_int, [_ctx, C.c_int, C.c_int, _f, C.c_void_p, C.c_int, _f,
                                       C.c_void_p]),
    "onmf_unfold3": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_int, _f,
                               C.c_void_p]),
    "onmf_sparse_code_each": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, _f, C.c_uint64,
                                        C.c_float, C.c_int, C.c_float, C.c_int, _f, C.c_void_p, C.c_void_p]),
    "onmf_sparse_code_lars": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, C.c_double, C.c_int, _f,
                                        C.c_void_p, C.c_void_p]),
    "onmf_ntf_step": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, C.c_int, _f, _f, _f, C.c_double, C.c_double, C.c_int,
                                _f, _f, _f, _f, C.c_void_p, C.c_void_p]),
    "onmf_patch_product": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _f, _f, _f, C.c_int, C.c_void_p]),
    "onmf_extract_patches": (C.c_int, [_ctx, _f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, _f, C.c_int, C.c_void_p]),
    "onmf_fold_patches": (C.c_int, [_ctx, _f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, _f, _f, C.c_void_p]),
    "onmf_surrogate_loss": (C.c_int, [_ctx, C.c_int, C.c_int, _f, _f, _f, _f, C.POINTER(C.c_double),
                                      C.c_void_p]),
    "onmf_ndl_init_embeddings": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "onmf_ndl_sample_patches": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, _f, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    "onmf_ndl_accumulate_edges": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, _f, C.c_int, _f, _f,
                                            C.c_void_p]),
    "onmf_numpy_set_jump_table": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_longlong]),
    "onmf_numpy_rand": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_size_t, _f, C.c_void_p, C.POINTER(C.c_int),
                                  C.c_void_p]),
    "onmf_fill_uniform": (C.c_int, [_ctx, _f, C.c_size_t, C.c_uint64, C.c_void_p]),
    "onmf_host_lambda_max": (C.c_double, [C.c_void_p, C.c_int]),
    "onmf_set_option": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "onmf_get_profile": (C.c_int, [_ctx, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "onmf_get_counter": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_longlong)]),
}

_lib = None


class OnmfLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OnmfLibraryError(
            "libonmf_hip.so not found at %s - build it with `make -C onmf_ontf_ndl_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback for this path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, ctx, what):
    if rc != OK:
        msg = load().onmf_last_error(ctx)
        raise OnmfLibraryError("%s failed (code %d): %s" % (what, rc, (msg or b"").decode()))
