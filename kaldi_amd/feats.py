"""Python mirror of the GMM systems' feature front end over the C ABI:
``splice-feats | transform-feats final.mat [| transform-feats --utt2spk=... trans]``
(egs/wsj/s5/steps/decode.sh:84-105; SpliceFrames feat/feature-functions.cc:391-412;
featbin/transform-feats.cc:92-108).  All arithmetic happens in libamlike.so on the GPU."""
import ctypes
from ctypes import POINTER, c_float, c_int32, c_void_p

import numpy as np

from . import _lib


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float)) if a is not None else None


class SpliceTransform:
    """Splicing by (left, right) frames followed by an optional [rows x cols] transform."""

    def __init__(self, dim, left, right, mat=None):
        self.mat = None if mat is None else np.ascontiguousarray(mat, dtype=np.float32)
        rows, cols = (0, 0) if self.mat is None else self.mat.shape
        self._h = c_void_p()
        _lib.check(_lib.lib().amlike_xform_create(int(dim), int(left), int(right), _fp(self.mat), rows,
                                                  cols, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().amlike_xform_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def InputDim(self):
        return _lib.lib().amlike_xform_input_dim(self._h)

    def OutputDim(self):
        return _lib.lib().amlike_xform_output_dim(self._h)

    def Apply(self, feats, spk_mat=None):
        """[T x dim] host features -> spliced and transformed features (host)."""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T, D = feats.shape
        if D != self.InputDim():
            raise ValueError("feature dimension %d, front end expects %d" % (D, self.InputDim()))
        rows2 = cols2 = 0
        if spk_mat is not None:
            spk_mat = np.ascontiguousarray(spk_mat, dtype=np.float32)
            rows2, cols2 = spk_mat.shape
        out = np.empty((T, rows2 if spk_mat is not None else self.OutputDim()), dtype=np.float32)
        _lib.check(_lib.lib().amlike_xform_apply(self._h, _fp(feats), T, D, _fp(spk_mat), rows2, cols2,
                                                 _fp(out), out.shape[1]))
        return out

    def ApplyBatchDevice(self, d_in_ptr, ld, utt_rows, d_out_ptr, ldo, d_spk_mats_ptr=None, rows2=0,
                         cols2=0, utt_mat=None, stream=None):
        """Utterances back to back in device memory; see amlike_xform_apply_dev."""
        rows = np.ascontiguousarray(utt_rows, np.int32)
        um = None if utt_mat is None else np.ascontiguousarray(utt_mat, np.int32)
        _lib.check(_lib.lib().amlike_xform_apply_dev(
            self._h, c_void_p(d_in_ptr), ld, rows.ctypes.data_as(POINTER(c_int32)), len(rows),
            c_void_p(d_spk_mats_ptr or 0), rows2, cols2,
            um.ctypes.data_as(POINTER(c_int32)) if um is not None else None,
            c_void_p(d_out_ptr), ldo, c_void_p(stream or 0)))


def score_raw(am, xform, raw_feats, spk_mat=None, log_sum_exp_prune=-1.0):
    """amlike_gmm_score_raw: un-spliced features in, [T x NumPdfs] log-likelihoods out."""
    raw = np.ascontiguousarray(raw_feats, dtype=np.float32)
    T, D = raw.shape
    rows2 = cols2 = 0
    if spk_mat is not None:
        spk_mat = np.ascontiguousarray(spk_mat, dtype=np.float32)
        rows2, cols2 = spk_mat.shape
    out = np.empty((T, am.NumPdfs()), dtype=np.float32)
    _lib.check(_lib.lib().amlike_gmm_score_raw(am._h, xform._h, _fp(raw), T, D, _fp(spk_mat), rows2, cols2,
                                               float(log_sum_exp_prune), _fp(out), out.shape[1]))
    return out
