"""Pins the CPU oracle (oracle/*.c, test infrastructure) to the reference:
  * against the committed golden vectors (outputs of the real reference library,
    tests/golden/make_golden.py), everywhere;
  * against the real reference run live, where oracle/_ref was built (the container that
    has the reference sources); skipped elsewhere.
Tolerance 1e-5 absolute: the reference's BLAS sums in a different order."""
import numpy as np
import pytest

import oracle
from oracle import kaldi_io as kio
from kaldi_amd import synth
from golden_util import gpath, load_am_gmm, load_am_nnet, load_ark, load_vec

ORACLE_TOL = 1e-5


def gmm_oracle(m, feats, prune=-1.0):
    gc, bad = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    out, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats, prune)
    assert st == 0
    return out, gc, bad


@pytest.mark.parametrize("tag", ["gmm_a", "gmm_b"])
def test_gconsts_bit_exact_vs_golden(tag):
    m = load_am_gmm(tag + ".mdl")
    off = m["pdf_offsets"]
    want = load_vec(tag + "_gconsts.vec")
    got = np.concatenate([oracle.compute_gconsts(m["weights"][a:b], m["means_invvars"][a:b],
                                                 m["inv_vars"][a:b])[0]
                          for a, b in zip(off[:-1], off[1:])])
    assert got.dtype == np.float32 and want.dtype == np.float32
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    if tag == "gmm_b":
        assert np.isneginf(got[2])  # the zero-weight Gaussian


@pytest.mark.parametrize("tag", ["gmm_a", "gmm_b"])
def test_gmm_likes_vs_golden(tag):
    m = load_am_gmm(tag + ".mdl")
    likes = dict(load_ark(tag + "_likes.ark"))
    for key, feats in load_ark(tag + "_feats.ark"):
        got, _, _ = gmm_oracle(m, feats)
        assert got.shape == likes[key].shape
        assert np.abs(got - likes[key]).max() <= ORACLE_TOL * max(1.0, np.abs(likes[key]).max() / 50)


def test_gmm_pruned_likes_vs_golden():
    m = load_am_gmm("gmm_a.mdl")
    likes = dict(load_ark("gmm_a_likes_prune5.ark"))
    full = dict(load_ark("gmm_a_likes.ark"))
    for key, feats in load_ark("gmm_a_feats.ark"):
        got, _, _ = gmm_oracle(m, feats, prune=5.0)
        assert np.abs(got - likes[key]).max() <= 2e-5
    assert np.abs(likes["utt1"] - full["utt1"]).max() > 1e-3  # pruning changes the answers


def test_per_gaussian_loglikes_vs_golden():
    m = load_am_gmm("gmm_a.mdl")
    off = m["pdf_offsets"]
    a, b = off[3], off[4]
    want = dict(load_ark("gmm_a_pdf3_gauss.ark"))
    gc, _ = oracle.compute_gconsts(m["weights"][a:b], m["means_invvars"][a:b], m["inv_vars"][a:b])
    for key, feats in load_ark("gmm_a_feats.ark"):
        got = oracle.diag_gmm_loglikes_mat(gc, m["means_invvars"][a:b], m["inv_vars"][a:b], feats)
        assert np.abs(got - want[key]).max() <= 2e-5


@pytest.mark.parametrize("out,pad,mode,scale", [
    ("nnet_a_post.ark", True, 0, 1.0), ("nnet_a_log.ark", True, 1, 1.0),
    ("nnet_a_loglikes.ark", True, 2, 0.1), ("nnet_a_post_nopad.ark", False, 0, 1.0),
])
def test_nnet_vs_golden(out, pad, mode, scale):
    comps, priors = load_am_nnet("nnet_a.mdl")
    want = dict(load_ark(out))
    n = 0
    for key, feats in load_ark("nnet_a_feats.ark"):
        if not pad and feats.shape[0] <= 5:
            assert key not in want
            continue
        got = oracle.nnet_compute(comps, priors, feats, pad_input=pad, mode=mode, prob_scale=scale)
        assert got.shape == want[key].shape
        assert np.abs(got - want[key]).max() <= ORACLE_TOL
        n += 1
    assert n >= 1


def test_nnet_other_components_vs_golden():
    comps, priors = load_am_nnet("nnet_b.mdl")
    want = dict(load_ark("nnet_b_post.ark"))
    for key, feats in load_ark("nnet_b_feats.ark"):
        got = oracle.nnet_compute(comps, priors, feats, pad_input=True, mode=0)
        assert np.abs(got - want[key]).max() <= ORACLE_TOL
        np.testing.assert_allclose(got.sum(1), 1.0, atol=1e-5)


@pytest.mark.parametrize("tag,nopad", [("nnet_c", True), ("nnet_d", False), ("nnet_e", True)])
def test_nnet_remaining_components_vs_golden(tag, nopad):
    """Dct (plain, reordered, truncated), FixedScale, FixedBias, SoftHinge, Permute, FixedLinear,
    Maxout, Scale, Power (every branch of ApplyPowAbs), BlockAffine and SpliceMax against the reference."""
    comps, priors = load_am_nnet(tag + ".mdl")
    want = dict(load_ark(tag + "_post.ark"))
    for key, feats in load_ark(tag + "_feats.ark"):
        got = oracle.nnet_compute(comps, priors, feats, pad_input=True, mode=0)
        assert got.shape == want[key].shape
        assert np.abs(got - want[key]).max() <= ORACLE_TOL
    if nopad:
        want = dict(load_ark(tag + "_post_nopad.ark"))
        assert min(v.shape[0] for v in want.values()) <= 2  # an utterance barely longer than the context
        for key, feats in load_ark(tag + "_feats.ark"):
            got = oracle.nnet_compute(comps, priors, feats, pad_input=False, mode=0)
            assert got.shape == want[key].shape
            assert np.abs(got - want[key]).max() <= ORACLE_TOL


def test_log_sum_exp_edge_cases():
    L = oracle.load()
    import ctypes
    L.oracle_log_sum_exp.restype = ctypes.c_float
    L.oracle_log_sum_exp.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_float]

    def lse(v, prune=-1.0):
        a = np.asarray(v, np.float32)
        return L.oracle_log_sum_exp(a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(a), prune)

    assert lse([3.0]) == 3.0
    assert abs(lse([0.0, 0.0]) - np.log(2.0)) < 1e-7
    assert lse([0.0, -100.0]) == 0.0                     # below max + log(eps): skipped
    assert lse([-np.inf, 2.0]) == 2.0
    assert abs(lse([1000.0, 1000.0]) - (1000.0 + np.log(2.0))) < 1e-4   # no overflow
    assert lse([0.0, -3.0], prune=2.0) == 0.0           # pruned away
    assert abs(lse([0.0, -3.0], prune=4.0) - np.log1p(np.exp(-3.0))) < 1e-6


def test_gconsts_flags_bad_variances():
    w = np.array([0.5, 0.5], np.float32)
    miv = np.zeros((2, 3), np.float32)
    iv = np.ones((2, 3), np.float32)
    w[1] = 0.0  # log(0) -> -inf: the Gaussian is disabled and counted (gmm/diag-gmm.cc:133-141)
    gc, bad = oracle.compute_gconsts(w, miv, iv)
    assert bad == 1 and np.isfinite(gc[0]) and np.isneginf(gc[1])
    iv[1, 0] = 0.0  # 0 * 0 / 0 -> NaN: an error in the reference (:129-131), -1 here
    gc, bad = oracle.compute_gconsts(w, miv, iv)
    assert bad == -1


# ----------------------------------------------------------------- live reference
needs_ref = pytest.mark.skipif(not oracle.have_ref(), reason="real-reference build absent")


def _pdfs(m):
    off = m["pdf_offsets"]
    return [dict(weights=m["weights"][a:b], means_invvars=m["means_invvars"][a:b],
                 inv_vars=m["inv_vars"][a:b]) for a, b in zip(off[:-1], off[1:])]


@needs_ref
@pytest.mark.parametrize("dim,P,G,T,seed", [(40, 50, 600, 30, 1), (24, 9, 64, 20, 2), (39, 11, 400, 25, 3)])
def test_gmm_oracle_vs_live_reference(tmp_path, dim, P, G, T, seed):
    m = synth.make_gmm(dim, P, G, seed=seed)
    feats = synth.make_feats(m, T, seed=seed + 10)
    with kio.open_out(str(tmp_path / "am.mdl"), True) as f:
        kio.write_am_diag_gmm(f, True, dim, _pdfs(m), with_gconsts=False)
    kio.write_ark(str(tmp_path / "f.ark"), [("u", feats)])
    oracle.ref_driver("gmm-likes", str(tmp_path / "am.mdl"), "ark:" + str(tmp_path / "f.ark"),
                      "ark:" + str(tmp_path / "o.ark"))
    want = kio.read_ark(str(tmp_path / "o.ark"))[0][1]
    got, _, _ = gmm_oracle(m, feats)
    assert np.abs(got - want).max() <= 2e-5


@needs_ref
@pytest.mark.parametrize("binary", [True, False])
def test_python_writers_are_readable_by_the_reference(tmp_path, binary):
    m = synth.make_gmm(13, 5, 30, seed=9)
    with kio.open_out(str(tmp_path / "am.mdl"), binary) as f:
        kio.write_am_diag_gmm(f, binary, 13, _pdfs(m), with_gconsts=False)
    oracle.ref_driver("gmm-copy", str(tmp_path / "am.mdl"), str(tmp_path / "c.mdl"), "1")
    r = kio.Reader(open(str(tmp_path / "c.mdl"), "rb").read())
    r.read_binary_marker()
    dim, pdfs = r.am_diag_gmm(True)
    assert dim == 13 and len(pdfs) == 5
    np.testing.assert_allclose(np.concatenate([p["weights"] for p in pdfs]), m["weights"], rtol=2e-6)
    comps, priors = synth.make_pnorm_nnet(input_dim=10, left=1, right=1, hidden=20, pnorm_out=4,
                                          num_hidden=1, num_pdfs=7, mix_up=9, seed=3)
    with kio.open_out(str(tmp_path / "n.mdl"), binary) as f:
        kio.write_am_nnet(f, binary, comps, priors)
    info = oracle.ref_driver("nnet-info", str(tmp_path / "n.mdl")).split()
    info = dict(zip(info[::2], info[1::2]))
    assert info["input-dim"] == "10" and info["output-dim"] == "7"
    assert info["left-context"] == "1" and info["right-context"] == "1"


@needs_ref
def test_nnet_oracle_vs_live_reference(tmp_path):
    comps, priors = synth.make_pnorm_nnet(input_dim=40, left=4, right=4, hidden=100, pnorm_out=20,
                                          num_hidden=2, num_pdfs=60, mix_up=70, seed=11)
    feats = np.random.default_rng(12).standard_normal((35, 40)).astype(np.float32)
    with kio.open_out(str(tmp_path / "n.mdl"), True) as f:
        kio.write_am_nnet(f, True, comps, priors)
    kio.write_ark(str(tmp_path / "f.ark"), [("u", feats)])
    oracle.ref_driver("nnet-loglikes", str(tmp_path / "n.mdl"), "ark:" + str(tmp_path / "f.ark"),
                      "ark:" + str(tmp_path / "o.ark"), "1", "0.25")
    want = kio.read_ark(str(tmp_path / "o.ark"))[0][1]
    got = oracle.nnet_compute(comps, priors, feats, pad_input=True, mode=2, prob_scale=0.25)
    assert np.abs(got - want).max() <= 2e-5


def test_gaussian_selection_vs_golden():
    from golden_util import load_gselect
    m = load_am_gmm("gsel.mdl")
    sel, tot = load_gselect("gsel_top5.txt")
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    for key, feats in load_ark("gsel_feats.ark"):
        scores = oracle.diag_gmm_loglikes_mat(gc, m["means_invvars"], m["inv_vars"], feats)
        idx, total = oracle.gselect(scores, 5)
        want = np.array(sel[key], dtype=np.int32)
        # a pair of scores within rounding of each other may swap between BLAS and plain loops
        differ = np.nonzero((idx != want).any(1))[0]
        for t in differ:
            assert sorted(idx[t]) == sorted(want[t]) or \
                np.abs(np.sort(scores[t])[-6] - np.sort(scores[t])[-5]) < 1e-4
        assert len(differ) <= 1
        assert abs(total - tot[key]) <= 1e-3 * len(feats)


def test_front_end_restatement_matches_the_reference():
    """splice_frames / transform_feats against SpliceFrames and the body of transform-feats run
    from the real library (tests/golden/xf_*.ark)."""
    from golden_util import load_ark
    items = load_ark("xf_feats.ark")
    mats = dict(load_ark("xf_mats.ark"))
    for (k, x), (_, sp), (_, lda), (_, lin), (_, both) in zip(
            items, load_ark("xf_splice.ark"), load_ark("xf_lda.ark"), load_ark("xf_lin.ark"),
            load_ark("xf_lda_fmllr.ark")):
        s = oracle.splice_frames(x, 3, 3)
        np.testing.assert_array_equal(s, sp)
        a = oracle.transform_feats(s, mats["lda"])
        assert np.abs(a - lda).max() <= 1e-5
        assert np.abs(oracle.transform_feats(s, mats["lin"]) - lin).max() <= 1e-5
        assert np.abs(oracle.transform_feats(a, mats["fmllr"]) - both).max() <= 2e-5
    with pytest.raises(ValueError):
        oracle.transform_feats(np.zeros((3, 5), np.float32), np.zeros((4, 7), np.float32))


@needs_ref
def test_bench_cpu_baseline_legs(monkeypatch):
    """The benchmark's CPU-baseline legs: the reference library on a bounded sample, one process
    and several side by side; both report frames/s of the sample they name."""
    import importlib
    import sys
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    monkeypatch.setenv("OMP_NUM_THREADS", "2")
    bench = importlib.import_module("bench")
    m = synth.make_gmm(39, 40, 400, seed=5)
    r = bench.cpu_baseline_gmm(m, None, 39, seconds_target=0.05)
    assert r["kind"] == "reference" and r["cores"] == 1 and r["value"] > 0
    assert r["all_cores"]["cores"] == 2 and r["all_cores"]["value"] > 0
    n = bench.cpu_baseline_nnet(seconds_target=0.1)
    assert n["kind"] == "reference" and n["value"] > 0 and n["all_cores"]["value"] > 0


@pytest.mark.parametrize("name,cfg,mseed,fseed,rows", [
    ("swbd_tri4_10k", "swbd_tri4", 777, 779, 24), ("wsj_tri4b_4k", "wsj_tri4b", 777, 780, 64)])
def test_oracle_vs_full_size_reference_summaries(name, cfg, mseed, fseed, rows):
    """The restatement on a sample of the frames of the full-size fixtures (per-frame summaries of
    what the real reference computed, tests/golden/make_golden_swbd.py)."""
    g = np.load(gpath(name + ".npz"))
    D, P, G = synth.GMM_CONFIGS[cfg]
    m = synth.make_gmm(D, P, G, seed=mseed)
    T = len(g["argmax"])
    feats = synth.make_feats(m, T, seed=fseed)
    pick = np.linspace(0, T - 1, rows).astype(int)
    got, _, _ = gmm_oracle(m, feats[pick])
    r = np.arange(rows)
    assert np.abs(got[r, g["argmax"][pick]] - g["best"][pick]).max() <= 2e-5
    assert np.abs(got[:, g["pdf_cols"]] - g["cols"][pick]).max() <= 2e-5
    assert np.abs(got.astype(np.float64).sum(1) - g["rowsum"][pick]).max() / P <= 1e-5
    near_tie = g["best"][pick] - g["second"][pick] < 1e-4
    assert ((got.argmax(1) == g["argmax"][pick]) | near_tie).all()


@pytest.mark.parametrize("fixture,kind,utts", [("nnet_swbd_10k", "plain", 6), ("nnet_swbd_10k", "huge", 2),
                                               ("nnet_swbd_peaky_2k", "peaky", 2)])
def test_nnet_oracle_vs_full_size_reference_summaries(fixture, kind, utts):
    """The nnet2 restatement on the first utterances of the full-size fixtures (per-frame summaries
    of what the real reference computed for the swbd p-norm shape,
    tests/golden/make_golden_nnet_swbd.py)."""
    g = np.load(gpath(fixture + ".npz"))
    comps, priors = synth.make_peaky_nnet() if kind == "peaky" else synth.make_pnorm_nnet(seed=780)
    us = synth.nnet_fullsize_utterances(kind)[:utts]
    got = np.concatenate([oracle.nnet_compute(comps, priors, u, True, 2, 1.0) for u in us])
    n = len(got)
    r = np.arange(n)
    assert np.abs(got[r, g[kind + "_argmax"][:n]] - g[kind + "_best"][:n]).max() <= 2e-4
    assert np.abs(got[:, g["pdf_cols"]] - g[kind + "_cols"][:n]).max() <= 2e-4
    assert np.abs(got.astype(np.float64).sum(1) - g[kind + "_rowsum"][:n]).max() / got.shape[1] <= 2e-5
    post = got + np.log(priors)[None, :]
    assert (post.argmax(1) == g[kind + "_post_argmax"][:n]).all()
    assert (got.argmax(1) == g[kind + "_argmax"][:n]).all()
