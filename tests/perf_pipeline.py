"""Host-to-host rates of the submit / wait path and of the binaries on the swbd tri4 shape
(not a pytest file; run on the GPU box: python tests/perf_pipeline.py [config]).
Model and feature files are written with the test suite's Kaldi-format writer."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import kaldi_io as kio  # noqa: E402  (file writer only)
from kaldi_amd import _lib, gmm, synth  # noqa: E402

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi_amd", "csrc", "bin")


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "swbd_tri4"
    D, P, G = synth.GMM_CONFIGS[cfg]
    m = synth.make_gmm(D, P, G, seed=777)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    lens = synth.utterance_lengths(40, 200, 1500, seed=779)
    T = int(sum(lens))
    feats = synth.make_feats(m, T, seed=778)

    # 1. one batch at a time, ordinary memory (amlike_gmm_score)
    out = np.empty((T, P), np.float32)
    am.LogLikelihoods(feats, out=out)
    t0 = time.perf_counter()
    for _ in range(3):
        am.LogLikelihoods(feats, out=out)
    dt = (time.perf_counter() - t0) / 3
    print("sync, pageable       : %7.2f ms per %d frames = %.2f Mframes/s" % (dt * 1e3, T, T / dt / 1e6), flush=True)

    # 2. the same call on page-locked buffers
    pin_in = [_lib.PinnedArray(T, D) for _ in range(2)]
    pin_out = [_lib.PinnedArray(T, P) for _ in range(2)]
    for p in pin_in:
        p.a[:] = feats
    am.LogLikelihoods(pin_in[0].a, out=pin_out[0].a)
    t0 = time.perf_counter()
    for _ in range(3):
        am.LogLikelihoods(pin_in[0].a, out=pin_out[0].a)
    dt = (time.perf_counter() - t0) / 3
    print("sync, page-locked    : %7.2f ms per %d frames = %.2f Mframes/s" % (dt * 1e3, T, T / dt / 1e6), flush=True)
    assert np.array_equal(pin_out[0].a, out)

    # 3. two batches in flight
    n = 8
    t0 = time.perf_counter()
    tickets = [am.Submit(pin_in[0].a, pin_out[0].a), am.Submit(pin_in[1].a, pin_out[1].a)]
    for i in range(n):
        am.Wait(tickets[i])
        if i + 2 < n:
            tickets.append(am.Submit(pin_in[i % 2].a, pin_out[i % 2].a))
    dt = (time.perf_counter() - t0) / n
    print("submit/wait, 2 in flight: %7.2f ms per %d frames = %.2f Mframes/s" % (dt * 1e3, T, T / dt / 1e6), flush=True)
    assert np.array_equal(pin_out[1].a, out)
    for p in pin_in + pin_out:
        p.close()
    am.close()

    # 4. the binaries on files (tmpfs), utterance by utterance
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        off = m["pdf_offsets"]
        pdfs = [dict(weights=m["weights"][a:b], means_invvars=m["means_invvars"][a:b],
                     inv_vars=m["inv_vars"][a:b]) for a, b in zip(off[:-1], off[1:])]
        with kio.open_out(os.path.join(d, "final.mdl"), True) as f:
            topo, triples = kio.make_transition_model(P)
            kio.write_transition_model(f, True, topo, triples)
            kio.write_am_diag_gmm(f, True, D, pdfs, with_gconsts=False)
        items, r0 = [], 0
        for i, n_ in enumerate(lens):
            items.append(("utt%03d" % i, feats[r0:r0 + n_]))
            r0 += n_
        kio.write_ark(os.path.join(d, "feats.ark"), items)
        for db in ("false", "true"):
            for passes in ("1", "4"):
                p = subprocess.run([os.path.join(BIN, "amlike-mock-decode"), "--double-buffer=" + db,
                                    "--touch=0.25", "--passes=" + passes, "gmm", os.path.join(d, "final.mdl"),
                                    "ark:" + os.path.join(d, "feats.ark"), "ark:/dev/null"],
                                   capture_output=True, timeout=600)
                print("mock-decode passes=%s:" % passes, p.stderr.decode().strip().splitlines()[-1], flush=True)
        p = subprocess.run([os.path.join(BIN, "gmm-compute-likes"), os.path.join(d, "final.mdl"),
                            "ark:" + os.path.join(d, "feats.ark"), "ark:/dev/null"],
                           capture_output=True, timeout=600)
        for line in p.stderr.decode().strip().splitlines()[-2:]:
            print("gmm-compute-likes:", line, flush=True)
        kio.write_ark(os.path.join(d, "feats4.ark"), [("r%d_%s" % (r, k), f) for r in range(4) for k, f in items])
        p = subprocess.run([os.path.join(BIN, "gmm-compute-likes"), os.path.join(d, "final.mdl"),
                            "ark:" + os.path.join(d, "feats4.ark"), "ark:/dev/null"],
                           capture_output=True, timeout=600)
        for line in p.stderr.decode().strip().splitlines()[-2:]:
            print("gmm-compute-likes x4:", line, flush=True)
        # 5. nnet-am-compute on the swbd-sized p-norm net, same utterances (40-dim input)
        comps, priors = synth.make_pnorm_nnet(seed=780)
        with kio.open_out(os.path.join(d, "nnet.mdl"), True) as f:
            topo, triples = kio.make_transition_model(len(priors))
            kio.write_transition_model(f, True, topo, triples)
            kio.write_am_nnet(f, True, comps, priors)
        rng = np.random.default_rng(3)
        nitems = [("utt%03d" % i, rng.standard_normal((int(n_), 40)).astype(np.float32)) for i, n_ in enumerate(lens)]
        kio.write_ark(os.path.join(d, "nfeats.ark"), nitems * 4)
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(BIN, "nnet-am-compute"), "--apply-log=true", os.path.join(d, "nnet.mdl"),
                            "ark:" + os.path.join(d, "nfeats.ark"), "ark:/dev/null"], capture_output=True, timeout=600)
        dt = time.perf_counter() - t0
        print("nnet-am-compute x4:", " | ".join(p.stderr.decode().strip().splitlines()[-2:]),
              "| whole process %.2f s = %.0f frames/s" % (dt, 4 * T / dt), flush=True)
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
