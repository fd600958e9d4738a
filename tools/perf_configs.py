"""Throughput of the GMM scoring path on each benchmark configuration (device-resident and
through the host-pointer entry point, i.e. including the PCIe copies)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kaldi_amd import _lib, gmm, synth

_lib.check(_lib.lib().amlike_init(0))
for cfg, T in (("yesno_mono", 15000), ("rm_tri1", 35554), ("wsj_tri4b", 35554), ("swbd_tri4", 35554)):
    D, P, G = synth.GMM_CONFIGS[cfg]
    m = synth.make_gmm(D, P, G, seed=777)
    G = int(m["pdf_offsets"][-1])
    feats = synth.make_feats(m, T, seed=778)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    ldp = (D + 3) // 4 * 4
    fpad = np.zeros((T, ldp), np.float32)
    fpad[:, :D] = feats
    d_feats = torch.from_numpy(fpad).cuda()
    d_out = torch.empty((T, P), dtype=torch.float32, device="cuda")
    for _ in range(2):
        ms = am.TimeScoreDevice(d_feats.data_ptr(), T, ldp, D, d_out.data_ptr(), P, 10)
    am.Poll()
    flop = 2.0 * (2 * D + 1) * G * T
    out = np.empty((T, P), np.float32)
    am.LogLikelihoods(feats, out=out)
    t0 = time.perf_counter()
    for _ in range(3):
        am.LogLikelihoods(feats, out=out)
    host_ms = (time.perf_counter() - t0) / 3 * 1e3
    print("%-10s D=%d P=%d G=%d T=%d: device %.3f ms = %.2f Mframes/s (%.1f TFLOP/s); host pointers %.1f ms = %.2f Mframes/s  [%s]"
          % (cfg, D, P, G, T, ms, T / ms / 1e3, flop / ms / 1e9, host_ms, T / host_ms / 1e3,
             am.kernel_info().split()[0]), flush=True)
    am.close()
