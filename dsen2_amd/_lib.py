"""ctypes binding of libdsen2_hip.so (the C ABI declared in include/dsen2_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Nothing in this package computes on the CPU.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DSEN2_HIP_LIB') or os.path.join(_HERE, 'libdsen2_hip.so')   # override: A/B of experimental builds

OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_WEIGHTS, ERR_WORKSPACE, ERR_NO_DEVICE, ERR_NOMEM, ERR_INTERNAL = -1, -2, -3, -4, -5, -6, -7

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/dsen2_hip.h declares
SIGNATURES = {
    'dsen2_version': (ctypes.c_char_p, []),
    'dsen2_last_error': (ctypes.c_char_p, []),
    'dsen2_device_count': (c_int, []),
    'dsen2_model_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int]),
    'dsen2_model_destroy': (None, [c_void_p]),
    'dsen2_model_num_params': (c_size_t, [c_void_p]),
    'dsen2_model_load_weights': (c_int, [c_void_p, c_float_p, c_size_t]),
    'dsen2_model_workspace_bytes': (c_int, [c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    'dsen2_model_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_void_p, c_size_t, c_void_p]),
    'dsen2_model_body_launches': (c_int, [c_void_p, c_int, c_int, c_int]),
    'dsen2_model_forward_timed': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_void_p, c_size_t, c_void_p, c_int, c_float_p]),
    'dsen2_model_forward_profile': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_void_p, c_size_t, c_void_p, c_int, c_int, c_float_p]),
    'dsen2_conv3x3_nhwc': (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_int, ctypes.c_float, c_void_p]),
    'dsen2_conv3x3_nhwc_ref': (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_int, c_int, ctypes.c_float, c_void_p]),
    'dsen2_split_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dsen2_join_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dsen2_conv3x3_body_bf16': (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int, c_int, ctypes.c_float, c_void_p]),
    'dsen2_conv3x3_first_planes': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float_p, c_float_p, c_int, c_int,
                                           c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'dsen2_split3_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dsen2_conv3x3_body_bf16x3': (c_int, [c_void_p, c_float_p, c_float_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_int, c_int, ctypes.c_float, c_void_p]),
    'dsen2_model_time_body_conv': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_void_p, c_float_p]),
    'dsen2_upsample_mirror_bilinear': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_float,
                                               c_void_p]),
    'dsen2_upsample_mirror_bilinear_ref': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_float,
                                                   c_void_p]),
    'dsen2_tile_gather': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, ctypes.c_float,
                                  c_void_p, c_void_p]),
    'dsen2_recompose': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, ctypes.c_float,
                                c_void_p]),
    'dsen2_recompose_rows': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, ctypes.c_float,
                                     c_int, c_int, c_void_p]),
}

_lib = None


class DSen2Error(RuntimeError):
    def __init__(self, code, message):
        super().__init__('libdsen2_hip error %d: %s' % (code, message))
        self.code = code


def load():
    """Load the shared library (once).  Raises if it has not been built — no CPU fallback exists."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the process's HIP runtime BEFORE this
        # library is mapped, otherwise /opt/rocm's copy is loaded first, torch then brings a second runtime
        # and the two do not share devices, streams or allocations ("no HIP device").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s not found: build it with `python -m dsen2_amd.build` (hipcc, gfx950). '
                              'dsen2_amd has no CPU fallback.' % LIB_PATH)
        if os.environ.get('DSEN2_HIP_LIB'):
            # the override exists for tools/ (A/B of experimental and diagnostic builds, whose outputs can be wrong under
            # ablation masks): never silently
            sys.stderr.write('dsen2_amd: DSEN2_HIP_LIB overrides the product library: loading %s\n' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        # Diagnostic builds only (python -m dsen2_amd.build --diag, selected with DSEN2_HIP_LIB): DSEN2_DIAG_SET=
        # "key=value,..." applies dsen2_diag_set() at load time (tools/: A/B of kernel structures, ablations).  The
        # product library has no such switch; asking it for one is an error, not a silent no-op.
        want = list(filter(None, os.environ.get('DSEN2_DIAG_SET', '').split(',')))
        if want and not hasattr(lib, 'dsen2_diag_set'):
            raise ImportError('DSEN2_DIAG_SET is set but %s is not a diagnostic build' % LIB_PATH)
        for kv in want:
            key, value = kv.split('=')
            diag_set(int(key), int(value))
    return _lib


def diag_set(key, value):
    """dsen2_diag_set of a -DDSEN2_DIAG library (tools/ only); raises on the product library."""
    lib = load()
    if not hasattr(lib, 'dsen2_diag_set'):
        raise DSen2Error(ERR_INVALID, '%s is not a diagnostic build (python -m dsen2_amd.build --diag)' % LIB_PATH)
    fn = lib.dsen2_diag_set
    fn.restype, fn.argtypes = c_int, [c_int, c_int]
    check(fn(int(key), int(value)))


def check(code):
    if code == ERR_NOMEM:          # the host ran out of memory inside the library: Python's own exception for that
        raise MemoryError('libdsen2_hip: ' + load().dsen2_last_error().decode('utf-8', 'replace'))
    if code != OK:
        raise DSen2Error(code, load().dsen2_last_error().decode('utf-8', 'replace'))


def call(name, *args):
    check(getattr(load(), name)(*args))
