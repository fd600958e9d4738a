"""Accuracy of the GMM scoring path against the oracle on the benchmark's model shapes: prints max / rms
|difference| and the share of frames with the same best pdf.  A measuring aid for builds that trade
precision (run by hand on the GPU box: python tests/acc_check.py); not collected by pytest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from kaldi_amd import _lib, gmm, synth
_lib.check(_lib.lib().amlike_init(0))
for cfg, T in (("swbd_tri4", 768), ("wsj_tri4b", 1024), ("rm_tri1", 2048)):
    D, P, G = synth.GMM_CONFIGS[cfg]
    m = synth.make_gmm(D, P, G, seed=777)
    feats = synth.make_feats(m, T, seed=778)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    want, st = oracle.gmm_compute_likes(am.gconsts, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    d = np.abs(got - want)
    print(cfg, "max|diff| %.3g  rms %.3g  argmax agree %.5f  kernel %s" % (d.max(), np.sqrt((d**2).mean()), (got.argmax(1) == want.argmax(1)).mean(), am.kernel_info().split()[0]), flush=True)
    am.close()
