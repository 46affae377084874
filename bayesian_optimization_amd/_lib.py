"""ctypes binding of libgpbo.so (the C ABI declared in include/gpbo.h).

There is no CPU fallback: if the shared library is missing the import of the
compute layer fails with an explicit message telling how to build it.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GPBO_LIBRARY') or os.path.join(_HERE, 'libgpbo.so')      # GPBO_LIBRARY: another build, for A/B timing

OK, ERR_NOT_PD, ERR_BAD_ARG, ERR_HIP = 0, 1, 2, 3
KERNEL_RBF, KERNEL_MATERN52 = 0, 1
VARIANT_V4, VARIANT_SINGLE, VARIANT_M3 = 0, 1, 3
ACQ_EI, ACQ_UCB, ACQ_PI = 0, 1, 2
INCUMBENT_MAX_MU, INCUMBENT_MAX_Y, INCUMBENT_GIVEN = 0, 1, 2


class Timings(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        'fit_total_ms', 'kbuild_ms', 'chol_ms', 'chol_count', 'trsv_ms', 'inverse_ms',
        'posterior_total_ms', 'ksbuild_ms', 'trmm_ms', 'trmm_launches', 'finalize_ms',
        'acquire_ms', 'trmm_flops', 'append_total_ms', 'append_sweep_ms', 'append_route')]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class StateDesc(C.Structure):
    _fields_ = [('n', C.c_int64), ('n_pad', C.c_int64),
                ('d', C.c_int32), ('kernel', C.c_int32), ('variant', C.c_int32), ('pad_', C.c_int32),
                ('sigmaY', C.c_double),
                ('W', C.c_void_p), ('W_bytes', C.c_int64),
                ('alpha', C.c_void_p), ('alpha_bytes', C.c_int64),
                ('Xts', C.c_void_p), ('Xts_bytes', C.c_int64),
                ('params', C.c_void_p), ('params_bytes', C.c_int64),
                ('y_max', C.c_double)]


_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/gpbo.h declares
SIGNATURES = {
    'gpbo_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'gpbo_destroy': (None, [C.c_void_p]),
    'gpbo_last_error': (C.c_char_p, [C.c_void_p]),
    'gpbo_last_info': (C.c_int64, [C.c_void_p]),
    'gpbo_stream': (C.c_void_p, [C.c_void_p]),
    'gpbo_fit': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, _dp, C.c_double, C.c_int32, C.c_int32, _dp, _dp]),
    'gpbo_set_candidates': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    'gpbo_set_candidates_uniform': (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64]),
    'gpbo_set_candidates_lattice': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int32]),
    'gpbo_posterior': (C.c_int, [C.c_void_p, _dp]),
    'gpbo_acquire': (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_double, _i64p, _dp]),
    'gpbo_first_at_least': (C.c_int, [C.c_void_p, C.c_double, _i64p]),
    'gpbo_acquire_static': (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int32,
                                      C.c_double, C.c_void_p, C.c_void_p, _i64p, _dp, _dp]),
    'gpbo_read': (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int64, C.c_int64]),
    'gpbo_predict_ei': (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_double, C.c_int32, C.c_double, _dp, _dp, _dp, _i64p, _dp, _dp]),
    'gpbo_append': (C.c_int, [C.c_void_p, _dp, C.c_double, _dp]),
    'gpbo_append_acquire': (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_int32, C.c_double, C.c_int32, C.c_double, _dp, _i64p, _dp]),
    'gpbo_kriging_fit': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, C.c_double, C.c_int32, _dp, _i32p, _dp]),
    'gpbo_kriging_cross_scale': (C.c_int, [C.c_void_p, _dp, C.c_double]),
    'gpbo_kriging_lml': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_int32, C.c_double, _dp, _i32p]),
    'gpbo_set_posterior': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64]),
    'gpbo_kernel_matrix': (C.c_int, [C.c_void_p, _dp, C.c_int64, _dp, C.c_int64, C.c_int32, _dp, C.c_int32, _dp]),
    'gpbo_kernel_from_distances': (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int64, C.c_int64, _dp, C.c_int32, _dp]),
    'gpbo_reserve_appends': (C.c_int, [C.c_void_p, C.c_int64]),
    'gpbo_lml_batch': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, _dp, _dp, C.c_int32, C.c_int32, _dp, _dp, _i32p]),
    'gpbo_get_factor': (C.c_int, [C.c_void_p, _dp, _dp]),
    'gpbo_model_scalars': (C.c_int, [C.c_void_p, _dp, _dp, _dp, _i64p]),
    'gpbo_shape': (C.c_int, [C.c_void_p, _i64p, _i32p, _i64p, _i64p]),
    'gpbo_cond_estimate': (C.c_int, [C.c_void_p, _dp]),
    'gpbo_last_timings': (C.c_int, [C.c_void_p, C.POINTER(Timings)]),
    'gpbo_set_timing': (C.c_int, [C.c_void_p, C.c_int32]),
    'gpbo_descent_small': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, _dp, C.c_int32, C.c_double, C.c_int32, C.c_double,
                                    C.c_double, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _i32p, _i32p]),
    'gpbo_state_describe': (C.c_int, [C.c_void_p, C.POINTER(StateDesc)]),
    'gpbo_state_alloc': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    'gpbo_state_commit': (C.c_int, [C.c_void_p, C.c_double]),
    'gpbo_device_count': (C.c_int, []),
    'gpbo_multi_create': (C.c_int, [_i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    'gpbo_multi_destroy': (None, [C.c_void_p]),
    'gpbo_multi_last_error': (C.c_char_p, [C.c_void_p]),
    'gpbo_multi_last_info': (C.c_int64, [C.c_void_p]),
    'gpbo_multi_size': (C.c_int32, [C.c_void_p]),
    'gpbo_multi_context': (C.c_void_p, [C.c_void_p, C.c_int32]),
    'gpbo_multi_block': (C.c_int, [C.c_void_p, C.c_int32, _i64p, _i64p]),
    'gpbo_multi_shape': (C.c_int, [C.c_void_p, _i64p, _i32p, _i64p, _i64p]),
    'gpbo_multi_reserve_appends': (C.c_int, [C.c_void_p, C.c_int64]),
    'gpbo_multi_set_chunk': (C.c_int, [C.c_void_p, C.c_int64]),
    'gpbo_multi_set_timing': (C.c_int, [C.c_void_p, C.c_int32]),
    'gpbo_multi_fit': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, _dp, C.c_double, C.c_int32, C.c_int32, _dp, _dp]),
    'gpbo_multi_kriging_fit': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, C.c_double, C.c_int32, _dp, _i32p, _dp]),
    'gpbo_multi_set_candidates': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    'gpbo_multi_set_candidates_uniform': (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64]),
    'gpbo_multi_set_candidates_lattice': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int32]),
    'gpbo_multi_posterior': (C.c_int, [C.c_void_p, _dp]),
    'gpbo_multi_acquire': (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_double, _i64p, _dp]),
    'gpbo_multi_first_at_least': (C.c_int, [C.c_void_p, C.c_double, _i64p]),
    'gpbo_multi_read': (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int64, C.c_int64]),
    'gpbo_multi_set_posterior': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64]),
    'gpbo_multi_predict_ei': (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_double, C.c_int32, C.c_double, _dp, _dp, _dp, _i64p, _dp, _dp]),
    'gpbo_multi_append': (C.c_int, [C.c_void_p, _dp, C.c_double, _dp]),
    'gpbo_multi_append_acquire': (C.c_int, [C.c_void_p, _dp, C.c_double, C.c_int32, C.c_double, C.c_int32, C.c_double, _dp, _i64p, _dp]),
    'gpbo_multi_lml_batch': (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int32, _dp, _dp, C.c_int32, C.c_int32, _dp, _dp, _i32p]),
    'gpbo_multi_last_distribute': (C.c_int, [C.c_void_p, _dp, _dp]),
    'gpbo_set_chunk': (C.c_int, [C.c_void_p, C.c_int64]),
    'gpbo_selftest_gemm': (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
}

_lib = None


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own copy of the HIP and HSA runtimes and load them by path.  libgpbo.so names its HIP
    runtime by soname, so whichever copy is mapped first serves it: loaded after torch it shares torch's runtime, loaded
    before torch the process ends up with two runtimes side by side (device pointers handed between them, two sets of
    queues).  When torch is installed but not imported yet, map its copies first so that there is one runtime whatever
    the import order.  GPBO_SYSTEM_HIP=1 keeps the system runtime."""
    if 'torch' in sys.modules or os.environ.get('GPBO_SYSTEM_HIP'):
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        return
    if not spec or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load libgpbo.so once and attach the prototypes.  Raises ImportError when the
    library has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libgpbo.so not found at %s: the HIP extension is not built and there is no CPU '
            'fallback. Build it with: python -c "import __graft_entry__ as g; g.build()"' % LIB_PATH)
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a):
    return a.ctypes.data_as(_dp)
