"""oracle/ -- TEST INFRASTRUCTURE ONLY (the checker, never the product).

``load()`` gives ctypes access to the plain-C CPU restatement of the reference's
acoustic-likelihood path (gmm_oracle.c, nnet_oracle.c -> liboracle.so);
``ref_driver()`` runs the real reference library built by build_ref.sh
(oracle/_ref/ref_driver), when it is present.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this package.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_float, c_int, c_int32

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_DRIVER = os.path.join(HERE, "_ref", "ref_driver")
SOURCES = ["gmm_oracle.c", "nnet_oracle.c"]

_lib = None


def build(force=False):
    """gcc -O2 (the reference's optimisation level, makefiles/linux_clapack.mk) -> liboracle.so."""
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    if not force and os.path.exists(LIB) and all(
            os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    subprocess.check_call(["gcc", "-O2", "-msse", "-msse2", "-fPIC", "-shared", "-std=c99",
                           "-D_POSIX_C_SOURCE=200809L", "-o", LIB] + srcs + ["-lm"])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        fp, ip = POINTER(c_float), POINTER(c_int32)
        L.oracle_compute_gconsts.restype = c_int
        L.oracle_compute_gconsts.argtypes = [fp, fp, fp, c_int, c_int, fp]
        L.oracle_log_sum_exp.restype = c_float
        L.oracle_log_sum_exp.argtypes = [fp, c_int, c_float]
        L.oracle_gmm_compute_likes.restype = c_int
        L.oracle_gmm_compute_likes.argtypes = [fp, fp, fp, ip, c_int, c_int, fp, c_int, c_int,
                                               c_float, fp, c_int, fp]
        L.oracle_diag_gmm_loglikes_mat.restype = None
        L.oracle_diag_gmm_loglikes_mat.argtypes = [fp, fp, fp, c_int, c_int, fp, c_int, c_int, fp,
                                                   c_int, fp]
        L.oracle_now.restype = c_double
        if hasattr(L, "oracle_nnet_compute"):
            _declare_nnet(L)
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def _ip(a):
    return a.ctypes.data_as(POINTER(c_int32))


def compute_gconsts(weights, miv, iv):
    L = load()
    w = np.ascontiguousarray(weights, np.float32)
    miv = np.ascontiguousarray(miv, np.float32)
    iv = np.ascontiguousarray(iv, np.float32)
    G, D = miv.shape
    gc = np.empty(G, np.float32)
    bad = L.oracle_compute_gconsts(_fp(w), _fp(miv), _fp(iv), G, D, _fp(gc))
    return gc, bad


def gmm_compute_likes(gconsts, miv, iv, offsets, feats, prune=-1.0):
    """Returns (loglikes[T x P], status)."""
    L = load()
    gconsts = np.ascontiguousarray(gconsts, np.float32)
    miv = np.ascontiguousarray(miv, np.float32)
    iv = np.ascontiguousarray(iv, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    feats = np.ascontiguousarray(feats, np.float32)
    T, D = feats.shape
    P = len(offsets) - 1
    out = np.empty((T, P), np.float32)
    scratch = np.empty(int(np.diff(offsets).max()) + D, np.float32)
    st = L.oracle_gmm_compute_likes(_fp(gconsts), _fp(miv), _fp(iv), _ip(offsets), P, D,
                                    _fp(feats), T, D, float(prune), _fp(out), P, _fp(scratch))
    return out, st


def diag_gmm_loglikes_mat(gconsts, miv, iv, feats):
    L = load()
    gconsts = np.ascontiguousarray(gconsts, np.float32)
    miv = np.ascontiguousarray(miv, np.float32)
    iv = np.ascontiguousarray(iv, np.float32)
    feats = np.ascontiguousarray(feats, np.float32)
    T, D = feats.shape
    M = miv.shape[0]
    out = np.empty((T, M), np.float32)
    scratch = np.empty(D, np.float32)
    L.oracle_diag_gmm_loglikes_mat(_fp(gconsts), _fp(miv), _fp(iv), M, D, _fp(feats), T, D,
                                   _fp(out), M, _fp(scratch))
    return out


def gselect(scores, num_gselect):
    """DiagGmm::GaussianSelection on precomputed per-Gaussian scores [T x M].
    Returns (indices [T x num_gselect] int32, total log-likelihood)."""
    L = load()
    scores = np.ascontiguousarray(scores, np.float32)
    T, M = scores.shape
    idx = np.empty((T, num_gselect), np.int32)
    L.oracle_gselect.restype = ctypes.c_double
    tot = L.oracle_gselect(_fp(scores), T, M, M, int(num_gselect), _ip(idx))
    return idx, float(tot)


# ------------------------------------------------------------------- nnet2
class NnetLayer(ctypes.Structure):
    _fields_ = [("kind", c_int32), ("input_dim", c_int32), ("output_dim", c_int32),
                ("left_context", c_int32), ("right_context", c_int32), ("const_dim", c_int32),
                ("p", c_float), ("linear", POINTER(c_float)), ("bias", POINTER(c_float)),
                ("sizes", POINTER(c_int32))]


def splice_frames(feats, left, right):
    """SpliceFrames (feat/feature-functions.cc:391-412): frame t becomes frames t-left ..
    t+right side by side, indices clamped to [0, T-1]."""
    feats = np.asarray(feats, np.float32)
    T = feats.shape[0]
    if T == 0 or feats.shape[1] == 0:
        raise ValueError("SpliceFrames: empty input")
    idx = np.clip(np.arange(T)[:, None] + np.arange(-left, right + 1)[None, :], 0, T - 1)
    return feats[idx].reshape(T, -1)


def transform_feats(feats, trans):
    """The body of transform-feats (featbin/transform-feats.cc:92-108): feat * trans^T, plus
    the last column of trans as an offset when it has one column more than the features."""
    feats = np.asarray(feats, np.float32)
    trans = np.asarray(trans, np.float32)
    d = feats.shape[1]
    if trans.shape[1] == d:
        return (feats @ trans.T).astype(np.float32)
    if trans.shape[1] == d + 1:
        return ((feats @ trans[:, :d].T).astype(np.float32) + trans[:, d]).astype(np.float32)
    raise ValueError("Transform matrix has bad dimension %dx%d versus feat dim %d"
                     % (trans.shape[0], trans.shape[1], d))


def _declare_nnet(L):
    fp = POINTER(c_float)
    L.oracle_nnet_compute.restype = c_int
    L.oracle_nnet_compute.argtypes = [POINTER(NnetLayer), c_int, fp, c_int, fp, c_int, c_int,
                                      c_int, c_int, c_float, fp, c_int]
    L.oracle_nnet_context.restype = None
    L.oracle_nnet_context.argtypes = [POINTER(NnetLayer), c_int, POINTER(c_int), POINTER(c_int)]


KINDS = {"SpliceComponent": 1, "FixedAffineComponent": 2, "AffineComponent": 2,
         "AffineComponentPreconditioned": 2, "AffineComponentPreconditionedOnline": 2,
         "FixedLinearComponent": 2,
         "PnormComponent": 3, "NormalizeComponent": 4, "SoftmaxComponent": 5,
         "SumGroupComponent": 6, "SigmoidComponent": 7, "TanhComponent": 8,
         "RectifiedLinearComponent": 9, "SoftHingeComponent": 10, "PowerComponent": 11,
         "ScaleComponent": 12, "FixedScaleComponent": 13, "FixedBiasComponent": 14,
         "MaxoutComponent": 15, "PermuteComponent": 16, "DctComponent": 17,
         "BlockAffineComponent": 18, "BlockAffineComponentPreconditioned": 18,
         "SpliceMaxComponent": 19}


def make_layers(components, struct=NnetLayer):
    """dict components (kaldi_io.write_component form) -> (ctypes array, keepalive list)."""
    arr = (struct * len(components))()
    keep = []
    for k, c in enumerate(components):
        t = c["type"]
        a = arr[k]
        a.kind = KINDS[t]
        if t == "SpliceComponent":
            a.input_dim = c["input_dim"]
            a.left_context, a.right_context = c["left"], c["right"]
            a.const_dim = c.get("const_dim", 0)
            a.output_dim = (c["input_dim"] - a.const_dim) * (c["left"] + 1 + c["right"]) + a.const_dim
        elif a.kind in (2, 18):
            W = np.ascontiguousarray(c["linear"], np.float32)
            keep.append(W)
            a.output_dim, a.input_dim = W.shape
            a.linear = _fp(W)
            if c.get("bias") is not None:
                b = np.ascontiguousarray(c["bias"], np.float32)
                keep.append(b)
                a.bias = _fp(b)
            if a.kind == 18:
                a.const_dim = c["num_blocks"]
                a.input_dim = W.shape[1] * c["num_blocks"]
        elif t in ("PnormComponent", "MaxoutComponent"):
            a.input_dim, a.output_dim, a.p = c["input_dim"], c["output_dim"], c.get("p", 2.0)
        elif t == "SumGroupComponent":
            s = np.ascontiguousarray(c["sizes"], np.int32)
            keep.append(s)
            a.sizes = _ip(s)
            a.input_dim, a.output_dim = int(s.sum()), len(s)
        elif t == "PermuteComponent":
            s = np.ascontiguousarray(c["reorder"], np.int32)
            keep.append(s)
            a.sizes = _ip(s)
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


def nnet_compute(components, priors, feats, pad_input=True, mode=0, prob_scale=1.0):
    """mode 0: posteriors, 1: +floor+log, 2: pseudo log-likelihoods. Returns [T_out x P]."""
    L = load()
    arr, keep = make_layers(components)
    feats = np.ascontiguousarray(feats, np.float32)
    T, D = feats.shape
    left, right = c_int(0), c_int(0)
    L.oracle_nnet_context(arr, len(components), ctypes.byref(left), ctypes.byref(right))
    t_out = T if pad_input else T - left.value - right.value
    out_dim = arr[len(components) - 1].output_dim
    out = np.empty((t_out, out_dim), np.float32)
    pri = np.ascontiguousarray(priors if priors is not None else np.zeros(0), np.float32)
    st = L.oracle_nnet_compute(arr, len(components), _fp(pri), len(pri), _fp(feats), T, D,
                               1 if pad_input else 0, mode, float(prob_scale), _fp(out), out_dim)
    if st != 0:
        raise RuntimeError("oracle_nnet_compute status %d" % st)
    return out


# ------------------------------------------------------------- real reference
def have_ref():
    return os.path.exists(REF_DRIVER) and os.access(REF_DRIVER, os.X_OK)


def _ref_env():
    env = dict(os.environ)
    env["MKL_THREADING_LAYER"] = "SEQUENTIAL"
    env["MKL_NUM_THREADS"] = "1"
    env["OMP_NUM_THREADS"] = "1"
    env["LD_LIBRARY_PATH"] = os.path.join(HERE, "_ref", "mkl")
    return env


def ref_driver(*args, timeout=3600):
    """Runs oracle/_ref/ref_driver (the real reference library). Returns stdout."""
    p = subprocess.run([REF_DRIVER] + [str(a) for a in args], env=_ref_env(), capture_output=True,
                       text=True, timeout=timeout)
    if p.returncode != 0:
        raise RuntimeError("ref_driver %s failed (%d): %s" % (args[0], p.returncode, p.stderr[-2000:]))
    return p.stdout


def ref_driver_many(arg_lists, timeout=3600):
    """Runs one single-threaded ref_driver process per argument list, all at the same time (the
    reference's JOB=1:$nj practice, steps/decode.sh:113).  Returns the list of stdouts."""
    procs = [subprocess.Popen([REF_DRIVER] + [str(a) for a in args], env=_ref_env(),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for args in arg_lists]
    outs = []
    for args, p in zip(arg_lists, procs):
        out, err = p.communicate(timeout=timeout)
        if p.returncode != 0:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            raise RuntimeError("ref_driver %s failed (%d): %s" % (args[0], p.returncode, err[-2000:]))
        outs.append(out)
    return outs
