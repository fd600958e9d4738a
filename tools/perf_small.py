"""Time per call of small device-pointer scoring calls that repeat on the same buffers (a decoder's
chunk loop): plain launches (default) or the recorded graph (AMLIKE_GMM_GRAPHS=1, set before start).
    python tools/perf_small.py [config] [T ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from kaldi_amd import _lib, gmm, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "swbd_tri4"
Ts = [int(a) for a in sys.argv[2:]] or [64, 500, 2000]
D, P, G = synth.GMM_CONFIGS[cfg]
m = synth.make_gmm(D, P, G, seed=777)
_lib.check(_lib.lib().amlike_init(0))
am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
stream = torch.cuda.Stream()
print("graphs:", "on" if os.environ.get("AMLIKE_GMM_GRAPHS", "0")[0] == "1" else "off", cfg)
for T in Ts:
    feats = torch.from_numpy(synth.make_feats(m, T, seed=778)).cuda()
    out = torch.empty((T, P), dtype=torch.float32, device="cuda")

    def call():
        am.LogLikelihoodsDevice(feats.data_ptr(), T, feats.stride(0), D, out.data_ptr(), P, stream=stream.cuda_stream)

    for _ in range(5):
        call()
    stream.synchronize()
    n = 300
    # back to back (the GPU's time per call) ...
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    stream.synchronize()
    busy = (time.perf_counter() - t0) / n
    # ... and one at a time, waited for (what a decoder that needs the answer sees)
    t0 = time.perf_counter()
    for _ in range(n):
        call()
        stream.synchronize()
    alone = (time.perf_counter() - t0) / n
    print("T=%5d  %.1f us per call back to back, %.1f us per call waited for" % (T, busy * 1e6, alone * 1e6), flush=True)
