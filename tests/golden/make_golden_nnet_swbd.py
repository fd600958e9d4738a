#!/usr/bin/env python3
"""Full-size golden vectors for the swbd nnet2 p-norm shape of the benchmark definition
(splice +-4, LDA 360, 4 x (2000 -> p-norm 200 -> normalize), 8600-way softmax).

The net and the utterances are the seeded synthetic ones the tests rebuild (kaldi_amd.synth:
make_pnorm_nnet(seed=780), make_peaky_nnet, nnet_fullsize_utterances); the expected values come from
the REAL reference library (oracle/_ref/ref_driver nnet-loglikes: NnetComputer's padding and
Component::Propagate of nnet2/nnet-compute.cc:62-106, then the floor / log / - log prior of
nnet2/decodable-am-nnet.h:57-69, prob_scale 1).  Per frame the fixture keeps what
make_golden_swbd.py keeps for the GMM shapes: best pdf, its value, the runner-up and its value, a few
fixed columns, the row sum in float64 -- plus the best pdf by log posterior (the pseudo-likelihood's
best pdf is mostly the pdf with the smallest prior) and the number of values on the 1e-20 floor.

    python tests/golden/make_golden_nnet_swbd.py     (needs oracle/_ref/ref_driver; about a minute)

writes nnet_swbd_10k.npz (kinds plain, x100, huge) and nnet_swbd_peaky_2k.npz.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import oracle  # noqa: E402
from oracle import kaldi_io as kio  # noqa: E402
from kaldi_amd import synth  # noqa: E402
from make_golden_swbd import cols_of, summarize  # noqa: E402

FLOOR_LOG = np.float32(np.log(np.float32(1.0e-20)))


def run_reference(comps, priors, kinds, tmp):
    mdl = os.path.join(tmp, "am.mdl")
    with kio.open_out(mdl, True) as f:
        kio.write_am_nnet(f, True, comps, priors)
    args, keys = [], []
    for kind in kinds:
        for i, u in enumerate(synth.nnet_fullsize_utterances(kind)):
            key = "%s_%02d" % (kind, i)
            kio.write_ark(os.path.join(tmp, key + ".ark"), [(key, u)])
            args.append(["nnet-loglikes", mdl, "ark:" + os.path.join(tmp, key + ".ark"),
                         "ark:" + os.path.join(tmp, key + ".out"), "1", "1.0"])
            keys.append((kind, key, len(u)))
    for i in range(0, len(args), 8):  # eight single-threaded processes at a time
        oracle.ref_driver_many(args[i:i + 8])
    logprior = np.log(priors.astype(np.float32))
    out = {}
    for kind in kinds:
        parts = []
        for k, key, n in keys:
            if k != kind:
                continue
            (_, ll), = kio.read_ark(os.path.join(tmp, key + ".out"))
            assert ll.shape == (n, len(priors))
            s = summarize(ll, cols_of(len(priors)))
            post = ll + logprior[None, :]  # log posteriors: what nnet-am-compute --apply-log writes
            s["post_argmax"] = post.argmax(1).astype(np.int32)
            s["post_best"] = post.max(1).astype(np.float32)
            s["floored"] = (ll + logprior[None, :] <= FLOOR_LOG * np.float32(0.999999)).sum(1).astype(np.int32)
            parts.append(s)
            os.remove(os.path.join(tmp, key + ".out"))
        for name in parts[0]:
            out[kind + "_" + name] = np.concatenate([p[name] for p in parts])
    out["pdf_cols"] = cols_of(len(priors))
    return out


def main():
    if not oracle.have_ref():
        sys.exit("oracle/_ref/ref_driver is not built")
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        comps, priors = synth.make_pnorm_nnet(seed=780)
        out = run_reference(comps, priors, ["plain", "x100", "huge"], tmp)
        np.savez_compressed(os.path.join(HERE, "nnet_swbd_10k.npz"), **out)
        for kind in ("plain", "x100", "huge"):
            print("%s: %d frames, mean best %.3f, near-ties (<1e-4) %d, floored values %d" % (
                kind, len(out[kind + "_best"]), out[kind + "_best"].mean(),
                int((out[kind + "_best"] - out[kind + "_second"] < 1e-4).sum()), int(out[kind + "_floored"].sum())))
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        comps, priors = synth.make_peaky_nnet()
        out = run_reference(comps, priors, ["peaky"], tmp)
        np.savez_compressed(os.path.join(HERE, "nnet_swbd_peaky_2k.npz"), **out)
        print("peaky: %d frames, mean best %.3f, floored values per frame %.0f of %d" % (
            len(out["peaky_best"]), out["peaky_best"].mean(), out["peaky_floored"].mean(), len(priors)))
    for n in ("nnet_swbd_10k.npz", "nnet_swbd_peaky_2k.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)), "bytes")


if __name__ == "__main__":
    main()
