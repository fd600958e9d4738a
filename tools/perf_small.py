"""Per-call time of the GMM scorer on utterance-sized batches (swbd tri4 model), device pointers,
calls issued back to back on one stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kaldi_amd import _lib, gmm, synth

D, P, G = synth.GMM_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "swbd_tri4"]
m = synth.make_gmm(D, P, G, seed=777)
_lib.check(_lib.lib().amlike_init(0))
import time
t0 = time.perf_counter()
am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
print("model packed and uploaded in %.3f s" % (time.perf_counter() - t0), flush=True)
for T in (1, 32, 250, 500, 1000, 2000, 4000, 8000):
    feats = np.zeros((T, 40), np.float32)
    feats[:, :D] = synth.make_feats(m, T, seed=5)
    d_f = torch.from_numpy(feats).cuda()
    d_o = torch.empty((T, P), dtype=torch.float32, device="cuda")
    for _ in range(2):
        ms = am.TimeScoreDevice(d_f.data_ptr(), T, 40, D, d_o.data_ptr(), P, 50)
    print("T=%5d: %8.1f us per call = %6.2f Mframes/s   %s" % (T, ms * 1e3, T / ms / 1e3, am.kernel_info().split()[1]), flush=True)
