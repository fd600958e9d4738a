"""Timing of the smaller entry points on one GPU: per-Gaussian scores (DiagGmm::LogLikelihoods
(Matrix)), Gaussian selection, the pruned log-sum-exp, the feature front end."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kaldi_amd import _lib, feats as front, gmm, synth
from ctypes import c_void_p

_lib.check(_lib.lib().amlike_init(0))
T = 35554
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# 1. a 2048-component UBM, D = 40: per-Gaussian scores [T x 2048]
M, D = 2048, 40
ubm = synth.make_gmm(D, 1, M, seed=5)
am = gmm.AmDiagGmm(ubm["weights"], ubm["means_invvars"], ubm["inv_vars"], ubm["pdf_offsets"])
x = synth.make_feats(ubm, T, seed=6)
d_x = torch.from_numpy(x).cuda()
d_o = torch.empty((T, M), dtype=torch.float32, device="cuda")
L = _lib.lib()
ms = timed(lambda: _lib.check(L.amlike_gmm_gauss_loglikes_dev(am._h, 0, 1, c_void_p(d_x.data_ptr()), T, D, D,
                                                                 c_void_p(d_o.data_ptr()), M, c_void_p(stream))))
out_bytes = 4.0 * T * M
print("gauss_loglikes  T=%d M=%d: %.3f ms  %.2f TB/s of output  %.1f TFLOP/s" %
      (T, M, ms, out_bytes / ms / 1e9, 2.0 * (2 * D + 1) * M * T / ms / 1e9), flush=True)

# 2. Gaussian selection, top 20 of 2048 (host pointers: features up, indices down)
t0 = time.perf_counter()
idx, tot = am.GaussianSelection(x, 20)
t1 = time.perf_counter()
idx, tot = am.GaussianSelection(x, 20)
t2 = time.perf_counter()
print("gselect top-20 of %d, host to host: %.2f ms (first call %.2f ms) = %.2f Mframes/s" %
      (M, (t2 - t1) * 1e3, (t1 - t0) * 1e3, T / (t2 - t1) / 1e6), flush=True)
am.close()

# 3. pruned log-sum-exp (log_sum_exp_prune = 5) on the wsj tri4b shape
Dw, Pw, Gw = synth.GMM_CONFIGS["wsj_tri4b"]
mw = synth.make_gmm(Dw, Pw, Gw, seed=777)
amw = gmm.AmDiagGmm(mw["weights"], mw["means_invvars"], mw["inv_vars"], mw["pdf_offsets"])
xw = np.zeros((T, 40), np.float32)
xw[:, :Dw] = synth.make_feats(mw, T, seed=778)
d_xw = torch.from_numpy(xw).cuda()
d_ow = torch.empty((T, Pw), dtype=torch.float32, device="cuda")
for prune in (-1.0, 5.0):
    ms = timed(lambda: amw.LogLikelihoodsDevice(d_xw.data_ptr(), T, 40, Dw, d_ow.data_ptr(), Pw, stream,
                                                log_sum_exp_prune=prune), iters=5)
    print("wsj_tri4b score, prune=%g: %.3f ms = %.2f Mframes/s" % (prune, ms, T / ms / 1e3), flush=True)
amw.close()

# 4. front end: 13-dim -> splice +-3 (91) -> 40x92 -> per-utterance 40x41
rng = np.random.default_rng(1)
lda = (rng.standard_normal((40, 92)) / 9.5).astype(np.float32)
spk = np.stack([(np.eye(40, 41) + 0.1 * rng.standard_normal((40, 41))).astype(np.float32) for _ in range(4)])
xf = front.SpliceTransform(13, 3, 3, lda)
lens = synth.utterance_lengths(40, 200, 1500, seed=779)
Tt = int(lens.sum())
d_raw = torch.from_numpy(rng.standard_normal((Tt, 13)).astype(np.float32)).cuda()
d_spk = torch.from_numpy(spk).cuda()
d_y = torch.empty((Tt, 40), dtype=torch.float32, device="cuda")
utt_mat = [i % 4 for i in range(len(lens))]
ms = timed(lambda: xf.ApplyBatchDevice(d_raw.data_ptr(), 13, lens, d_y.data_ptr(), 40, d_spk.data_ptr(), 40, 41,
                                       utt_mat, stream))
flop = 2.0 * Tt * (40 * 92 + 40 * 41)
print("front end (splice+LDA+fMLLR) %d frames in %d utterances: %.3f ms = %.1f Mframes/s, %.2f TFLOP/s, "
      "%.2f TB/s of in+out" % (Tt, len(lens), ms, Tt / ms / 1e3, flop / ms / 1e9, 4.0 * Tt * 53 / ms / 1e9), flush=True)
