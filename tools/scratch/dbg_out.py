import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from kaldi_amd import nnet, synth
os.environ["AMLIKE_NNET_OUT_CUS"] = "2"
comps, priors = synth.make_pnorm_nnet(seed=780)
x = np.random.default_rng(41).normal(size=(1000, 40)).astype(np.float32)
net = nnet.AmNnet(comps, priors)
got = net.Compute(x, mode=nnet.LOG, prob_scale=0.9)
net.close()
os.environ["AMLIKE_NNET_NO_FUSED_OUT"] = "1"
net2 = nnet.AmNnet(comps, priors)
ref = net2.Compute(x, mode=nnet.LOG, prob_scale=0.9)
net2.close()
d = np.abs(got - ref)
bad = d > 1e-3
print("bad elements", bad.sum(), "of", bad.size)
rows = np.where(bad.any(1))[0]
print("bad rows", len(rows), rows[:40], rows[-10:])
cols = np.where(bad.any(0))[0]
print("bad cols", len(cols), cols[:40], cols[-10:])
r = rows[0]
print("row", r, "bad cols", np.where(bad[r])[0][:50])
print(got[r, cols[:8]], ref[r, cols[:8]])
print("row-constant offset?", (got[r] - ref[r])[:10], np.std((got[r]-ref[r])[~bad[r]]) if (~bad[r]).any() else None)
