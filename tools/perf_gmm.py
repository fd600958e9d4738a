"""Ad-hoc timing of the GMM scoring kernel variants on one GPU (tuning aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kaldi_amd import _lib, gmm, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "swbd_tri4"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["bf16x3:3x8", "bf16x3:2x8", "f32:3x8"]
D, P, G = synth.GMM_CONFIGS[cfg]
t0 = time.time()
m = synth.make_gmm(D, P, G, seed=777)
feats = synth.make_feats(m, T, seed=778)
# PERF_FEATS=zero|const: features with no / little switching activity (how much of the kernel's
# time follows the clock the chip sustains on real data)
if os.environ.get("PERF_FEATS") == "zero":
    feats[:] = 0.0
elif os.environ.get("PERF_FEATS") == "const":
    feats[:] = feats[0]
print("model+feats built in %.1fs  G=%d sizes: max %d" % (time.time() - t0, m["pdf_offsets"][-1],
      np.diff(m["pdf_offsets"]).max()), flush=True)
_lib.check(_lib.lib().amlike_init(0))
ldp = (D + 3) // 4 * 4
fpad = np.zeros((T, ldp), np.float32)
fpad[:, :D] = feats
d_feats = torch.from_numpy(fpad).cuda()
d_out = torch.empty((T, P), dtype=torch.float32, device="cuda")
flop = 2.0 * (2 * D + 1) * G * T
first_out = None
for v in variants:
    if ":" in v:
        os.environ["AMLIKE_GMM_MODE"], v = v.split(":")
    os.environ["AMLIKE_GMM_VARIANT"] = v
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    for it in range(2):
        ms = am.TimeScoreDevice(d_feats.data_ptr(), T, ldp, D, d_out.data_ptr(), P, 5)
    try:
        am.Poll()
    except Exception as e:
        print('poll:', e)
    print("%s  %s\n   %.3f ms/launch  %.3f Mframes/s  %.1f TFLOP/s (%.1f%% of 157.3)" % (
        v, am.kernel_info(), ms, T / ms / 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100), flush=True)
    got = d_out.cpu().numpy()
    if first_out is None:
        first_out = got
    else:  # the variants against the first one named
        d = np.abs(got - first_out)
        print("   against the first variant's scores: max |difference| %.3g, relative to max(1, |score|) %.3g"
              % (float(d.max()), float((d / np.maximum(1.0, np.abs(first_out))).max())))
    am.close()
