"""Time of amlike_gmm_create (gconsts, packing on host threads, upload) for the benchmark's model shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kaldi_amd import _lib, gmm, synth
_lib.check(_lib.lib().amlike_init(0))
for cfg in ("swbd_tri4", "wsj_tri4b", "rm_tri1"):
    D, P, G = synth.GMM_CONFIGS[cfg]
    m = synth.make_gmm(D, P, G, seed=777)
    for rep in range(2):
        t0 = time.time()
        am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
        dt = time.time() - t0
        am.close()
    print(cfg, "create (gconsts + pack + upload): %.3f s" % dt, flush=True)
