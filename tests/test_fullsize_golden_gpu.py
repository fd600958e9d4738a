"""Full-size parity against the reference itself: the swbd tri4 (8600 pdfs, 200 000 Gaussians) and
wsj tri4b (3400 pdfs, 40 000 Gaussians) shapes of the benchmark definition on 10 240 / 4 096 frames,
checked against per-frame summaries of the [T x P] matrix the real reference library computed
(tests/golden/make_golden_swbd.py: best pdf, its value and the runner-up's, a few fixed columns, the
row sum).  Bar: 1e-3 absolute per value; the same best pdf on >= 99.99 % of the frames, where a
frame whose two best reference values lie closer together than 1e-4 counts as agreeing with either
(on such a frame the reference's own answer turns on its summation order)."""
import json
import os

import numpy as np
import pytest

from kaldi_amd import gmm, synth

pytestmark = pytest.mark.gpu
TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fixture -> (config, model seed, feature seed): as in tests/golden/make_golden_swbd.py
CASES = {"swbd_tri4_10k": ("swbd_tri4", 777, 779), "wsj_tri4b_4k": ("wsj_tri4b", 777, 780)}


def record(name, counts):
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fullsize_argmax_counts.jsonl"), "a") as f:
            f.write(json.dumps(dict(case=name, **counts)) + "\n")
    except OSError:
        pass
    print(name, counts)


def check_against_summary(got, g, name):
    T = len(g["argmax"])
    assert got.shape[0] == T and np.isfinite(got).all()
    rows = np.arange(T)
    assert np.abs(got[rows, g["argmax"]] - g["best"]).max() <= TOL
    assert np.abs(got[rows, g["second_idx"]] - g["second"]).max() <= TOL
    assert np.abs(got[:, g["pdf_cols"]] - g["cols"]).max() <= TOL
    # the row sum sees every one of the P values of a frame: mean deviation per value
    assert np.abs(got.astype(np.float64).sum(1) - g["rowsum"]).max() / got.shape[1] <= 1e-4
    am = got.argmax(1)
    differs = am != g["argmax"]
    tie = (g["best"] - g["second"] < 1e-4) & (am == g["second_idx"])
    # the measured counts go to gpurun_out/fullsize_argmax_counts.jsonl
    record(name, dict(frames=T, best_pdf_differs=int(differs.sum()),
                      of_them_reference_near_ties=int((differs & tie).sum())))
    strict = 1.0 - differs.mean()
    excused = 1.0 - (differs & ~tie).mean()
    assert excused >= 0.9999, "best pdf differs on %d of %d frames (%d of them near-ties)" % (
        differs.sum(), T, (differs & tie).sum())
    # Without the tie clause: measured 2 of 10 240 frames (swbd tri4) and 0 of 4 096 (wsj tri4b), i.e.
    # 99.98 % / 100 %; both swbd frames are reference near-ties (two pdfs within 1e-4 of each other,
    # closer than the reference's own summation-order noise).  Every differing frame must be one.
    assert (differs & ~tie).sum() == 0
    assert differs.sum() <= 2 * T * 1e-4
    # nowhere is the engine's best pdf worse than the reference's best by more than the bar
    assert (g["best"] - got[rows, am] <= TOL).all()
    return strict


@pytest.mark.parametrize("name", sorted(CASES))
def test_full_size_against_the_reference(amlike_device, name):
    cfg, mseed, fseed = CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    D, P, G = synth.GMM_CONFIGS[cfg]
    m = synth.make_gmm(D, P, G, seed=mseed)
    feats = synth.make_feats(m, len(g["argmax"]), seed=fseed)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert "gmm_score_diff_kernel" in am.kernel_info() and "f16x2" in am.kernel_info()
    check_against_summary(got, g, name + " / " + am.kernel_info().split()[0])
    am.close()


@pytest.mark.parametrize("mode", ["bf16x3", "diff-bf16x3"])
def test_wsj_tri4b_other_kernel_families(amlike_device, mode, monkeypatch):
    """The wide-range fallback kernel and the bf16x3-operand form of the default kernel against the
    same reference summaries."""
    monkeypatch.setenv("AMLIKE_GMM_MODE", mode)
    cfg, mseed, fseed = CASES["wsj_tri4b_4k"]
    g = np.load(os.path.join(GOLDEN, "wsj_tri4b_4k.npz"))
    D, P, G = synth.GMM_CONFIGS[cfg]
    m = synth.make_gmm(D, P, G, seed=mseed)
    feats = synth.make_feats(m, len(g["argmax"]), seed=fseed)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert "bf16x3" in am.kernel_info()
    check_against_summary(got, g, "wsj_tri4b_4k / " + mode)
    am.close()
