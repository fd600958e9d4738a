"""Python mirror of the reference's GMM interface over the C ABI.

Names follow the reference: ``AmDiagGmm`` (gmm/am-diag-gmm.h:36-111) with
``LogLikelihoods`` computing the dense [T x P] matrix that
``gmm-compute-likes`` builds one (frame, pdf) at a time
(gmmbin/gmm-compute-likes.cc:71-77), and ``ComponentLogLikelihoods`` for
``DiagGmm::LogLikelihoods(Matrix)`` (gmm/diag-gmm.cc:546-562).
All arithmetic happens in libamlike.so on the GPU; this file only marshals.
"""
import ctypes
from ctypes import POINTER, c_float, c_int32, c_void_p

import numpy as np

from . import _lib


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def _ip(a):
    return a.ctypes.data_as(POINTER(c_int32))


def compute_gconsts(weights, means_invvars, inv_vars):
    """DiagGmm::ComputeGconsts (gmm/diag-gmm.cc:114-152). Returns (gconsts, num_bad)."""
    L = _lib.lib()
    w = np.ascontiguousarray(weights, dtype=np.float32)
    miv = np.ascontiguousarray(means_invvars, dtype=np.float32)
    iv = np.ascontiguousarray(inv_vars, dtype=np.float32)
    G, D = miv.shape
    gc = np.empty(G, dtype=np.float32)
    bad = c_int32(0)
    _lib.check(L.amlike_gmm_compute_gconsts(_fp(w), _fp(miv), _fp(iv), G, D, _fp(gc),
                                            ctypes.byref(bad)))
    return gc, bad.value


class AmDiagGmm:
    """A packed acoustic model resident on the GPU."""

    def __init__(self, weights, means_invvars, inv_vars, pdf_offsets, gconsts=None, device=None):
        L = _lib.lib()
        if device is not None:
            _lib.check(L.amlike_init(int(device)))
        self.means_invvars = np.ascontiguousarray(means_invvars, dtype=np.float32)
        self.inv_vars = np.ascontiguousarray(inv_vars, dtype=np.float32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float32)
        self.pdf_offsets = np.ascontiguousarray(pdf_offsets, dtype=np.int32)
        if gconsts is None:
            gconsts, _ = compute_gconsts(self.weights, self.means_invvars, self.inv_vars)
        self.gconsts = np.ascontiguousarray(gconsts, dtype=np.float32)
        G, D = self.means_invvars.shape
        self._h = c_void_p()
        _lib.check(L.amlike_gmm_create(_fp(self.gconsts), _fp(self.means_invvars),
                                       _fp(self.inv_vars), _ip(self.pdf_offsets),
                                       len(self.pdf_offsets) - 1, G, D, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().amlike_gmm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def NumPdfs(self):
        return _lib.lib().amlike_gmm_num_pdfs(self._h)

    def NumGauss(self):
        return _lib.lib().amlike_gmm_num_gauss(self._h)

    def Dim(self):
        return _lib.lib().amlike_gmm_dim(self._h)

    def kernel_info(self):
        return _lib.lib().amlike_gmm_kernel_info(self._h).decode()

    def LogLikelihoods(self, feats, log_sum_exp_prune=-1.0, out=None):
        """[T x D] host features -> [T x NumPdfs] log-likelihoods (host)."""
        feats = np.asarray(feats, dtype=np.float32)
        if feats.ndim != 2 or feats.strides[1] != 4 or feats.strides[0] % 4:
            feats = np.ascontiguousarray(feats)
        T, D = feats.shape
        ld = feats.strides[0] // 4
        P = self.NumPdfs()
        if out is None:
            out = np.empty((T, P), dtype=np.float32)
        ldo = out.strides[0] // 4
        _lib.check(_lib.lib().amlike_gmm_score(self._h, _fp(feats), T, ld, D,
                                               float(log_sum_exp_prune), _fp(out), ldo))
        return out

    def Submit(self, feats, out, log_sum_exp_prune=-1.0):
        """Enqueues LogLikelihoods(feats) -> out without waiting (amlike_gmm_score_submit).
        Both arrays must be float32 with unit column stride and stay alive until Wait(ticket);
        page-locked ones (_lib.PinnedArray) make the call asynchronous."""
        if feats.dtype != np.float32 or out.dtype != np.float32 or feats.strides[1] != 4 or out.strides[1] != 4:
            raise ValueError("Submit needs float32 arrays with contiguous rows")
        T, D = feats.shape
        ticket = c_int32(-1)
        _lib.check(_lib.lib().amlike_gmm_score_submit(self._h, _fp(feats), T, feats.strides[0] // 4, D,
                                                      float(log_sum_exp_prune), _fp(out),
                                                      out.strides[0] // 4, ctypes.byref(ticket)))
        return ticket.value

    def Wait(self, ticket):
        _lib.check(_lib.lib().amlike_gmm_score_wait(self._h, int(ticket)))

    def LogLikelihoodsDevice(self, d_feats_ptr, T, ld, D, d_out_ptr, ldo, stream=None,
                             log_sum_exp_prune=-1.0):
        """Device-pointer form (asynchronous on `stream`)."""
        _lib.check(_lib.lib().amlike_gmm_score_dev(self._h, c_void_p(d_feats_ptr), T, ld, D,
                                                   float(log_sum_exp_prune), c_void_p(d_out_ptr),
                                                   ldo, c_void_p(stream or 0)))

    def LogLikelihoodsDeviceNoAlloc(self, d_feats_ptr, T, ld, D, d_out_ptr, ldo, stream=None):
        """Device-pointer form that allocates nothing (scratch memory built into the library)."""
        _lib.check(_lib.lib().amlike_gmm_score_dev_noalloc(self._h, c_void_p(d_feats_ptr), T, ld, D,
                                                           c_void_p(d_out_ptr), ldo, c_void_p(stream or 0)))

    def Poll(self, stream=None):
        _lib.check(_lib.lib().amlike_gmm_poll(self._h, c_void_p(stream or 0)))

    def ProfileKernel(self, enable=True):
        """Event pairs around the scoring kernel of every launch from now on."""
        _lib.check(_lib.lib().amlike_gmm_profile_kernel(self._h, 1 if enable else 0))

    def KernelTime(self):
        """(mean milliseconds, launches) of the scoring kernel since the last call."""
        ms, n = c_float(0), c_int32(0)
        _lib.check(_lib.lib().amlike_gmm_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def TimeScoreDevice(self, d_feats_ptr, T, ld, D, d_out_ptr, ldo, iters, stream=None):
        ms = c_float(0)
        _lib.check(_lib.lib().amlike_gmm_time_score_dev(self._h, c_void_p(d_feats_ptr), T, ld, D,
                                                        c_void_p(d_out_ptr), ldo,
                                                        c_void_p(stream or 0), iters,
                                                        ctypes.byref(ms)))
        return ms.value

    def ComponentLogLikelihoods(self, feats, pdf_begin=0, pdf_end=None):
        """Per-Gaussian scores of pdfs [pdf_begin, pdf_end): [T x M]."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T, D = feats.shape
        if pdf_end is None:
            pdf_end = self.NumPdfs()
        M = int(self.pdf_offsets[pdf_end] - self.pdf_offsets[pdf_begin])
        out = np.empty((T, M), dtype=np.float32)
        _lib.check(_lib.lib().amlike_gmm_gauss_loglikes(self._h, pdf_begin, pdf_end, _fp(feats),
                                                        T, D, D, _fp(out), M))
        return out

    def GaussianSelection(self, feats, num_gselect, pdf=0):
        """DiagGmm::GaussianSelection (gmm/diag-gmm.cc:801-871) for mixture `pdf`.
        Returns (indices [T x num_gselect] int32, best first; total log-likelihood)."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T, D = feats.shape
        idx = np.empty((T, int(num_gselect)), dtype=np.int32)
        tot = ctypes.c_double(0.0)
        _lib.check(_lib.lib().amlike_gmm_gselect(self._h, int(pdf), _fp(feats), T, D, D, int(num_gselect),
                                                 _ip(idx), ctypes.byref(tot)))
        return idx, tot.value
