"""
ctypes binding of libscfgp_hip.so (C ABI declared in include/scfgp_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of the product path fails loudly (build it with `python __graft_entry__.py`
or `make -C scfgp_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCFGP_LIB_VARIANT selects an alternative build (tuning experiments only, e.g. '_bk32')
LIB_PATH = os.path.join(_HERE, 'lib', 'libscfgp_hip%s.so' % os.environ.get('SCFGP_LIB_VARIANT', ''))

SCFGP_F64, SCFGP_F32, SCFGP_F16X3 = 0, 1, 2        # F16X3: fp32 mode with the four N-sized products as a three-term fp16 split (secondary)
SCFGP_REDO = 1                           # scfgp_finish / scfgp_factor: run the stages again (precision level raised or agreed lower), not an error
SCFGP_EPEER = -5
ERRORS = {-1: 'bad argument', -2: 'HIP error', -3: 'not positive definite', -4: 'non-finite cost', -5: 'another rank failed'}

_c_double_p = C.POINTER(C.c_double)
_c_i64_p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); mirrors include/scfgp_hip.h one to one
SIGNATURES = {
    'scfgp_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'scfgp_destroy': (None, [C.c_void_p]),
    'scfgp_last_error': (C.c_char_p, [C.c_void_p]),
    'scfgp_set_params': (C.c_int, [C.c_void_p, _c_double_p, C.c_int]),
    'scfgp_get_params': (C.c_int, [C.c_void_p, _c_double_p, C.c_int]),
    'scfgp_set_data': (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, C.c_int64, C.c_int64]),
    'scfgp_eval': (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, C.c_int64, C.c_int,
                             _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'scfgp_eval_rows': (C.c_int, [C.c_void_p, _c_i64_p, C.c_int64, C.c_int, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'scfgp_predict': (C.c_int, [C.c_void_p, _c_double_p, C.c_int64, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'scfgp_set_x_scaler': (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'scfgp_predict_raw': (C.c_int, [C.c_void_p, _c_double_p, C.c_int64, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'scfgp_set_y_scaler': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    'scfgp_predict_y': (C.c_int, [C.c_void_p, _c_double_p, C.c_int64, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                  _c_double_p]),
    'scfgp_pass1': (C.c_int, [C.c_void_p]),
    'scfgp_factor': (C.c_int, [C.c_void_p]),
    'scfgp_pass2': (C.c_int, [C.c_void_p, C.c_int]),
    'scfgp_adjoint': (C.c_int, [C.c_void_p]),
    'scfgp_pass3': (C.c_int, [C.c_void_p]),
    'scfgp_finish': (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    'scfgp_fail_stage': (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    'scfgp_fetch_factors': (C.c_int, [C.c_void_p, _c_double_p, _c_double_p]),
    'scfgp_exchange': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), _c_i64_p]),
    'scfgp_stream_fence': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    'scfgp_comm_unique_id': (C.c_int, [C.c_void_p]),
    'scfgp_comm_init': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'scfgp_comm_destroy': (C.c_int, [C.c_void_p]),
    'scfgp_opt_init': (C.c_int, [C.c_void_p, C.c_int, _c_double_p, C.c_int, C.c_double]),
    'scfgp_opt_state': (C.c_int, [C.c_void_p, C.c_int, C.c_int, _c_double_p]),
    'scfgp_opt_step': (C.c_int, [C.c_void_p, _c_double_p, C.c_int]),
    'scfgp_train': (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, _c_double_p]),
    'scfgp_get_condition': (C.c_int, [C.c_void_p, _c_double_p, C.c_int]),
    'scfgp_get_dims': (C.c_int, [C.c_void_p, _c_i64_p, C.c_int]),
    'scfgp_set_profiling': (C.c_int, [C.c_void_p, C.c_int]),
    'scfgp_get_timings': (C.c_int, [C.c_void_p, _c_double_p, C.POINTER(C.c_char_p), C.c_int]),
    'scfgp_debug_read': (C.c_int64, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    'scfgp_box_probe': (C.c_int, [C.c_int, _c_double_p, C.c_int]),
    'scfgp_selftest_row_splits': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]),
    'scfgp_set_option': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
}

_lib = None


def load():
    """Load (once) and return the ctypes library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "scfgp_amd: %s not found. The HIP extension is the product path and has no "
            "fallback; build it with `make -C scfgp_amd/csrc` (or __graft_entry__.build())." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a):
    return None if a is None else a.ctypes.data_as(_c_double_p)
