"""Python mirror of the reference's nnet2 forward interface over the C ABI.

``AmNnet`` holds a device-resident forward pipeline (nnet2/am-nnet.h, nnet2/nnet-nnet.h:62-);
``Compute`` is NnetComputation() (nnet2/nnet-compute.h:44-47) plus the output
post-processing of nnet-am-compute (nnet2bin/nnet-am-compute.cc:93-96) or of
DecodableAmNnet (nnet2/decodable-am-nnet.h:59-69).  All arithmetic happens in
libamlike.so on the GPU; this file only marshals.
"""
import ctypes
from ctypes import POINTER, c_float, c_int32, c_int64, c_void_p

import numpy as np

from . import _lib

POSTERIORS, LOG, LOGLIKES = 0, 1, 2

KINDS = {"SpliceComponent": 1, "FixedAffineComponent": 2, "AffineComponent": 2,
         "AffineComponentPreconditioned": 2, "AffineComponentPreconditionedOnline": 2,
         "FixedLinearComponent": 2, "BlockAffineComponent": 2, "BlockAffineComponentPreconditioned": 2,
         "PnormComponent": 3, "NormalizeComponent": 4, "SoftmaxComponent": 5,
         "SumGroupComponent": 6, "SigmoidComponent": 7, "TanhComponent": 8,
         "RectifiedLinearComponent": 9, "SoftHingeComponent": 10, "PowerComponent": 11,
         "ScaleComponent": 12, "FixedScaleComponent": 13, "FixedBiasComponent": 14,
         "MaxoutComponent": 15, "PermuteComponent": 16, "DctComponent": 17, "SpliceMaxComponent": 18}


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def block_diagonal(stacked, num_blocks):
    """The dense matrix a BlockAffineComponent stands for (nnet2/nnet-component.cc:1907-1939):
    block b of the output reads block b of the input through rows [b * out_block, (b + 1) *
    out_block) of the stacked [num_blocks * out_block x in_block] parameter matrix."""
    stacked = np.asarray(stacked, np.float32)
    rows, in_block = stacked.shape
    if num_blocks <= 0 or rows % num_blocks:
        raise ValueError("BlockAffineComponent: %d rows do not split into %d blocks" % (rows, num_blocks))
    out_block = rows // num_blocks
    W = np.zeros((rows, in_block * num_blocks), np.float32)
    for b in range(num_blocks):
        W[b * out_block:(b + 1) * out_block, b * in_block:(b + 1) * in_block] = \
            stacked[b * out_block:(b + 1) * out_block]
    return W


def _components_to_c(components):
    arr = (_lib.Component * len(components))()
    keep = []
    for k, c in enumerate(components):
        t = c["type"]
        a = arr[k]
        a.kind = KINDS[t]
        if t == "SpliceComponent":
            a.input_dim = c["input_dim"]
            a.left_context, a.right_context = c["left"], c["right"]
            a.const_dim = c.get("const_dim", 0)
            a.output_dim = (a.input_dim - a.const_dim) * (a.left_context + 1 + a.right_context) + a.const_dim
        elif a.kind == 2:
            W = c["linear"] if "num_blocks" not in c else block_diagonal(c["linear"], c["num_blocks"])
            W = np.ascontiguousarray(W, np.float32)
            keep.append(W)
            a.output_dim, a.input_dim = W.shape
            a.linear = _fp(W)
            if c.get("bias") is not None:  # FixedLinearComponent has none
                b = np.ascontiguousarray(c["bias"], np.float32)
                if b.shape != (W.shape[0],):
                    raise ValueError("%s: bias of %s for %d outputs" % (t, b.shape, W.shape[0]))
                keep.append(b)
                a.bias = _fp(b)
        elif t in ("PnormComponent", "MaxoutComponent"):
            a.input_dim, a.output_dim, a.p = c["input_dim"], c["output_dim"], c.get("p", 2.0)
        elif t == "SumGroupComponent":
            s = np.ascontiguousarray(c["sizes"], np.int32)
            keep.append(s)
            a.sizes = s.ctypes.data_as(POINTER(c_int32))
            a.input_dim, a.output_dim = int(s.sum()), len(s)
        elif t == "PermuteComponent":
            s = np.ascontiguousarray(c["reorder"], np.int32)
            keep.append(s)
            a.sizes = s.ctypes.data_as(POINTER(c_int32))
            a.input_dim = a.output_dim = len(s)
        elif t in ("FixedScaleComponent", "FixedBiasComponent"):
            v = np.ascontiguousarray(c["scales" if t == "FixedScaleComponent" else "bias"], np.float32)
            keep.append(v)
            a.bias = _fp(v)
            a.input_dim = a.output_dim = len(v)
        elif t == "SpliceMaxComponent":
            a.input_dim = a.output_dim = c["dim"]
            a.left_context, a.right_context = c["left"], c["right"]
        elif t == "DctComponent":
            dct_keep = c.get("dct_keep_dim", 0) or c["dct_dim"]
            a.input_dim, a.const_dim = c["dim"], c["dct_dim"]
            a.left_context = 1 if c.get("reorder", False) else 0
            a.output_dim = dct_keep * (c["dim"] // c["dct_dim"])
        else:
            a.input_dim = a.output_dim = c["dim"]
            if t == "PowerComponent":
                a.p = c["power"]
            elif t == "ScaleComponent":
                a.p = c["scale"]
    return arr, keep


class AmNnet:
    """components: list of dicts (type + parameters), priors: [num_pdfs] or None."""

    def __init__(self, components, priors=None, device=None):
        L = _lib.lib()
        if device is not None:
            _lib.check(L.amlike_init(int(device)))
        arr, keep = _components_to_c(components)
        pri = np.ascontiguousarray(priors if priors is not None else np.zeros(0), np.float32)
        self._h = c_void_p()
        _lib.check(L.amlike_nnet_create(arr, len(components), _fp(pri), len(pri), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().amlike_nnet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def InputDim(self):
        return _lib.lib().amlike_nnet_input_dim(self._h)

    def OutputDim(self):
        return _lib.lib().amlike_nnet_output_dim(self._h)

    def LeftContext(self):
        return _lib.lib().amlike_nnet_left_context(self._h)

    def RightContext(self):
        return _lib.lib().amlike_nnet_right_context(self._h)

    def Compute(self, feats, pad_input=True, mode=POSTERIORS, prob_scale=1.0):
        feats = np.asarray(feats, dtype=np.float32)
        if feats.ndim != 2 or feats.strides[1] != 4 or feats.strides[0] % 4:
            feats = np.ascontiguousarray(feats)
        T, _ = feats.shape
        ld = feats.strides[0] // 4
        t_out = T if pad_input else T - self.LeftContext() - self.RightContext()
        if t_out <= 0:
            raise ValueError("empty output")
        out = np.empty((t_out, self.OutputDim()), dtype=np.float32)
        _lib.check(_lib.lib().amlike_nnet_compute(self._h, _fp(feats), T, ld, 1 if pad_input else 0,
                                                  mode, float(prob_scale), _fp(out), out.shape[1]))
        return out

    def Submit(self, feats, out, pad_input=True, mode=POSTERIORS, prob_scale=1.0):
        """Enqueues Compute(feats) -> out without waiting (amlike_nnet_compute_submit); both
        arrays float32 with contiguous rows, alive until Wait(ticket)."""
        if feats.dtype != np.float32 or out.dtype != np.float32 or feats.strides[1] != 4 or out.strides[1] != 4:
            raise ValueError("Submit needs float32 arrays with contiguous rows")
        ticket = c_int32(-1)
        _lib.check(_lib.lib().amlike_nnet_compute_submit(
            self._h, _fp(feats), feats.shape[0], feats.strides[0] // 4, 1 if pad_input else 0, mode,
            float(prob_scale), _fp(out), out.strides[0] // 4, ctypes.byref(ticket)))
        return ticket.value

    def Wait(self, ticket):
        _lib.check(_lib.lib().amlike_nnet_compute_wait(self._h, int(ticket)))

    def ComputeBatchDevice(self, d_feats_ptr, ld, utt_rows, d_out_ptr, ldo, pad_input=True,
                           mode=POSTERIORS, prob_scale=1.0, stream=None):
        rows = np.ascontiguousarray(utt_rows, np.int32)
        total = c_int64(0)
        _lib.check(_lib.lib().amlike_nnet_compute_batch_dev(
            self._h, c_void_p(d_feats_ptr), ld, rows.ctypes.data_as(POINTER(c_int32)), len(rows),
            1 if pad_input else 0, mode, float(prob_scale), c_void_p(d_out_ptr), ldo,
            c_void_p(stream or 0), ctypes.byref(total)))
        return total.value
