"""ctypes binding of libamlike.so (the C ABI declared in include/amlike.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no fallback: if the shared object is missing, loading raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMLIKE_LIB_PATH: another build of the same library (tuning aid for A/B runs)
LIB_PATH = os.environ.get("AMLIKE_LIB_PATH") or os.path.join(_HERE, "csrc", "libamlike.so")

OK, EINVAL, EDIM, ENONFINITE, EHIP, EOOM, ENODEV, EBUSY = range(8)

_lib = None


class AmlikeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("amlike error %d: %s" % (code, msg))
        self.code = code


class Component(ctypes.Structure):
    """struct amlike_component of include/amlike.h."""
    _fields_ = [
        ("kind", c_int32), ("input_dim", c_int32), ("output_dim", c_int32),
        ("left_context", c_int32), ("right_context", c_int32), ("const_dim", c_int32),
        ("p", c_float), ("linear", POINTER(c_float)), ("bias", POINTER(c_float)),
        ("sizes", POINTER(c_int32)),
    ]


def _declare(lib):
    fp, ip = POINTER(c_float), POINTER(c_int32)
    sigs = {
        "amlike_init": (ctypes.c_int, [ctypes.c_int]),
        "amlike_last_error": (c_char_p, []),
        "amlike_device_count": (ctypes.c_int, []),
        "amlike_host_alloc": (ctypes.c_int, [ctypes.c_size_t, POINTER(c_void_p)]),
        "amlike_host_free": (ctypes.c_int, [c_void_p]),
        "amlike_gmm_score_submit": (ctypes.c_int, [c_void_p, fp, c_int32, c_int32, c_int32, c_float, fp, c_int32, ip]),
        "amlike_gmm_score_wait": (ctypes.c_int, [c_void_p, c_int32]),
        "amlike_nnet_compute_submit": (ctypes.c_int, [c_void_p, fp, c_int32, c_int32, c_int32, c_int32, c_float, fp, c_int32, ip]),
        "amlike_nnet_compute_wait": (ctypes.c_int, [c_void_p, c_int32]),
        "amlike_xform_create": (ctypes.c_int, [c_int32, c_int32, c_int32, fp, c_int32, c_int32, POINTER(c_void_p)]),
        "amlike_xform_destroy": (ctypes.c_int, [c_void_p]),
        "amlike_xform_input_dim": (c_int32, [c_void_p]),
        "amlike_xform_output_dim": (c_int32, [c_void_p]),
        "amlike_xform_device": (c_int32, [c_void_p]),
        "amlike_gmm_device": (c_int32, [c_void_p]),
        "amlike_nnet_device": (c_int32, [c_void_p]),
        "amlike_xform_apply": (ctypes.c_int, [c_void_p, fp, c_int32, c_int32, fp, c_int32, c_int32, fp, c_int32]),
        "amlike_xform_apply_dev": (ctypes.c_int, [c_void_p, c_void_p, c_int32, ip, c_int32, c_void_p, c_int32, c_int32, ip, c_void_p, c_int32, c_void_p]),
        "amlike_gmm_score_raw": (ctypes.c_int, [c_void_p, c_void_p, fp, c_int32, c_int32, fp, c_int32, c_int32, c_float, fp, c_int32]),
        "amlike_gmm_create": (ctypes.c_int, [fp, fp, fp, ip, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
        "amlike_gmm_destroy": (ctypes.c_int, [c_void_p]),
        "amlike_gmm_compute_gconsts": (ctypes.c_int, [fp, fp, fp, c_int32, c_int32, fp, ip]),
        "amlike_gmm_num_pdfs": (c_int32, [c_void_p]),
        "amlike_gmm_num_gauss": (c_int32, [c_void_p]),
        "amlike_gmm_dim": (c_int32, [c_void_p]),
        "amlike_gmm_score": (ctypes.c_int, [c_void_p, fp, c_int32, c_int32, c_int32, c_float, fp, c_int32]),
        "amlike_gmm_score_dev_noalloc": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
        "amlike_gmm_noalloc_scratch_bytes": (ctypes.c_size_t, []),
        "amlike_gmm_score_dev": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p]),
        "amlike_gmm_poll": (ctypes.c_int, [c_void_p, c_void_p]),
        "amlike_gmm_gauss_loglikes": (ctypes.c_int, [c_void_p, c_int32, c_int32, fp, c_int32, c_int32, c_int32, fp, c_int32]),
        "amlike_gmm_gauss_loglikes_dev": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
        "amlike_gmm_gselect": (ctypes.c_int, [c_void_p, c_int32, fp, c_int32, c_int32, c_int32, c_int32, ip, POINTER(ctypes.c_double)]),
        "amlike_gmm_kernel_info": (c_char_p, [c_void_p]),
        "amlike_gmm_profile_kernel": (ctypes.c_int, [c_void_p, c_int32]),
        "amlike_gmm_kernel_time": (ctypes.c_int, [c_void_p, POINTER(c_float), ip]),
        "amlike_gmm_time_score_dev": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, POINTER(c_float)]),
        "amlike_nnet_create": (ctypes.c_int, [POINTER(Component), c_int32, fp, c_int32, POINTER(c_void_p)]),
        "amlike_nnet_destroy": (ctypes.c_int, [c_void_p]),
        "amlike_nnet_input_dim": (c_int32, [c_void_p]),
        "amlike_nnet_output_dim": (c_int32, [c_void_p]),
        "amlike_nnet_left_context": (c_int32, [c_void_p]),
        "amlike_nnet_right_context": (c_int32, [c_void_p]),
        "amlike_nnet_compute": (ctypes.c_int, [c_void_p, fp, c_int32, c_int32, c_int32, c_int32, c_float, fp, c_int32]),
        "amlike_nnet_compute_dev": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p]),
        "amlike_nnet_compute_batch_dev": (ctypes.c_int, [c_void_p, c_void_p, c_int32, ip, c_int32, c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p, POINTER(ctypes.c_int64)]),
        "amlike_nnet_time_compute_dev": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p, c_int32, POINTER(c_float)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return sigs


EXPORTS = None


def lib():
    """Load libamlike.so (once).  Raises if it has not been built."""
    global _lib, EXPORTS
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 and
        # must load it before any other copy, or it reports "No HIP GPUs".  Loading
        # torch first makes libamlike bind to that same runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = ctypes.CDLL(LIB_PATH)
        EXPORTS = sorted(_declare(_lib).keys())
    return _lib


class PinnedArray:
    """A float32 [rows x cols] array in page-locked host memory (amlike_host_alloc); `.a` is
    the numpy view, valid until close()."""

    def __init__(self, rows, cols):
        import numpy as np
        self._p = c_void_p()
        n = max(1, int(rows) * int(cols))
        check(lib().amlike_host_alloc(n * 4, ctypes.byref(self._p)))
        buf = (c_float * n).from_address(self._p.value)
        self.a = np.frombuffer(buf, dtype=np.float32, count=int(rows) * int(cols)).reshape(int(rows), int(cols))

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.a = None
            lib().amlike_host_free(self._p)
            self._p = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check(code):
    if code != OK:
        raise AmlikeError(code, lib().amlike_last_error().decode(errors="replace"))
