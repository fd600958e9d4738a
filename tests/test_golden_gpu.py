"""The HIP path against the committed golden vectors: outputs of the real reference
library (tests/golden/make_golden.py), so these run on the GPU box where the reference
itself is absent.  Tolerance 1e-3 absolute on log-likelihoods, 1e-5 on posteriors."""
import numpy as np
import pytest

from kaldi_amd import gmm, nnet
from golden_util import load_am_gmm, load_am_nnet, load_ark, load_vec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["gmm_a", "gmm_b"])
@pytest.mark.parametrize("mode", ["default", "diff-f16x2", "diff-bf16x3", "bf16x3", "f32"])
def test_gmm_likes_vs_golden(amlike_device, tag, mode, monkeypatch):
    m = load_am_gmm(tag + ".mdl")
    if mode == "f32" and m["dim"] not in (39, 40):
        pytest.skip("the exact-fp32 kernel is built for dims 39 and 40")
    if mode == "default":
        monkeypatch.delenv("AMLIKE_GMM_MODE", raising=False)
    else:
        monkeypatch.setenv("AMLIKE_GMM_MODE", mode)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    want = dict(load_ark(tag + "_likes.ark"))
    for key, feats in load_ark(tag + "_feats.ark"):
        got = am.LogLikelihoods(feats)
        assert got.shape == want[key].shape
        assert np.abs(got - want[key]).max() <= 1e-3
        assert (got.argmax(1) == want[key].argmax(1)).all()
    if mode == "default":
        assert "gmm_score_diff_kernel" in am.kernel_info()


def test_gmm_gconsts_vs_golden_bits(amlike_device):
    for tag in ("gmm_a", "gmm_b"):
        m = load_am_gmm(tag + ".mdl")
        am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
        np.testing.assert_array_equal(np.asarray(am.gconsts).view(np.uint32),
                                      load_vec(tag + "_gconsts.vec").view(np.uint32))


def test_gmm_pruned_vs_golden(amlike_device):
    m = load_am_gmm("gmm_a.mdl")
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    want = dict(load_ark("gmm_a_likes_prune5.ark"))
    for key, feats in load_ark("gmm_a_feats.ark"):
        got = am.LogLikelihoods(feats, log_sum_exp_prune=5.0)
        # a Gaussian within rounding of the cutoff may fall on either side
        assert (np.abs(got - want[key]) <= 1e-3).mean() >= 0.999


def test_per_gaussian_loglikes_vs_golden(amlike_device):
    m = load_am_gmm("gmm_a.mdl")
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    want = dict(load_ark("gmm_a_pdf3_gauss.ark"))
    for key, feats in load_ark("gmm_a_feats.ark"):
        got = am.ComponentLogLikelihoods(feats, 3, 4)
        assert got.shape == want[key].shape and np.abs(got - want[key]).max() <= 1e-3


@pytest.mark.parametrize("out,pad,mode,scale,tol", [
    ("nnet_a_post.ark", True, nnet.POSTERIORS, 1.0, 1e-5), ("nnet_a_log.ark", True, nnet.LOG, 1.0, 1e-3),
    ("nnet_a_loglikes.ark", True, nnet.LOGLIKES, 0.1, 1e-3),
    ("nnet_a_post_nopad.ark", False, nnet.POSTERIORS, 1.0, 1e-5),
])
def test_nnet_vs_golden(amlike_device, out, pad, mode, scale, tol):
    comps, priors = load_am_nnet("nnet_a.mdl")
    net = nnet.AmNnet(comps, priors)
    want = dict(load_ark(out))
    for key, feats in load_ark("nnet_a_feats.ark"):
        if key not in want:
            continue
        got = net.Compute(feats, pad_input=pad, mode=mode, prob_scale=scale)
        assert got.shape == want[key].shape
        assert np.abs(got - want[key]).max() <= tol
        assert (got.argmax(1) == want[key].argmax(1)).all()


def test_nnet_other_components_vs_golden(amlike_device):
    comps, priors = load_am_nnet("nnet_b.mdl")
    net = nnet.AmNnet(comps, priors)
    want = dict(load_ark("nnet_b_post.ark"))
    for key, feats in load_ark("nnet_b_feats.ark"):
        got = net.Compute(feats, pad_input=True, mode=nnet.POSTERIORS)
        assert np.abs(got - want[key]).max() <= 1e-5


@pytest.mark.parametrize("tag", ["nnet_c", "nnet_d", "nnet_e"])
def test_nnet_remaining_components_vs_golden(amlike_device, tag):
    """Dct (plain, reordered, truncated), FixedScale, FixedBias, SoftHinge, Permute, FixedLinear,
    Maxout, Scale, Power, BlockAffine and SpliceMax: the reference's posteriors, 1e-5 absolute, for one
    utterance at a time and for all of them in one batch."""
    import torch
    comps, priors = load_am_nnet(tag + ".mdl")
    net = nnet.AmNnet(comps, priors)
    want = dict(load_ark(tag + "_post.ark"))
    items = load_ark(tag + "_feats.ark")
    for key, feats in items:
        got = net.Compute(feats, pad_input=True, mode=nnet.POSTERIORS)
        assert got.shape == want[key].shape
        assert np.abs(got - want[key]).max() <= 1e-5
        assert (got.argmax(1) == want[key].argmax(1)).all()
    if tag != "nnet_d":
        nopad = dict(load_ark(tag + "_post_nopad.ark"))
        for key, feats in items:
            got = net.Compute(feats, pad_input=False, mode=nnet.POSTERIORS)
            assert got.shape == nopad[key].shape
            assert np.abs(got - nopad[key]).max() <= 1e-5
    lens = np.array([f.shape[0] for _, f in items], np.int32)
    allf = torch.from_numpy(np.concatenate([f for _, f in items])).cuda()
    out = torch.empty((int(lens.sum()), net.OutputDim()), device="cuda")
    net.ComputeBatchDevice(allf.data_ptr(), allf.shape[1], lens, out.data_ptr(), out.shape[1],
                           pad_input=True, mode=nnet.POSTERIORS)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.abs(got - np.concatenate([want[k] for k, _ in items])).max() <= 1e-5
    net.close()


def test_nnet_create_refuses_malformed_components(amlike_device):
    def one(c, dim):
        return [dict(type="SpliceComponent", input_dim=dim, left=0, right=0), c]
    bad = [
        one(dict(type="PermuteComponent", reorder=[0, 1, 1, 3]), 4),           # not a permutation
        one(dict(type="MaxoutComponent", input_dim=10, output_dim=3), 10),     # 3 does not divide 10
        one(dict(type="DctComponent", dim=12, dct_dim=5, reorder=False), 12),  # 5 does not divide 12
        one(dict(type="DctComponent", dim=12, dct_dim=4, reorder=False, dct_keep_dim=6), 12),
    ]
    for comps in bad:
        with pytest.raises(Exception):
            nnet.AmNnet(comps, None)


def test_gaussian_selection_vs_golden_and_oracle(amlike_device):
    import oracle
    from golden_util import load_gselect
    m = load_am_gmm("gsel.mdl")
    sel, tot = load_gselect("gsel_top5.txt")
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    for key, feats in load_ark("gsel_feats.ark"):
        idx, total = am.GaussianSelection(feats, 5)
        want = np.array(sel[key], dtype=np.int32)
        scores = oracle.diag_gmm_loglikes_mat(gc, m["means_invvars"], m["inv_vars"], feats)
        for t in np.nonzero((idx != want).any(1))[0]:
            # only near-ties (within fp32 rounding of the two code paths) may be ordered differently
            s = np.sort(scores[t])[::-1]
            assert np.abs(np.diff(s[:6])).min() < 1e-4
        assert ((idx == want).all(1)).mean() >= 0.95
        assert abs(total - tot[key]) <= 1e-3 * len(feats)
    # num_gselect larger than the mixture: every Gaussian, then -1
    idx, _ = am.GaussianSelection(load_ark("gsel_feats.ark")[0][1][:3], 70)
    assert (np.sort(idx[:, :64], axis=1) == np.arange(64)).all() and (idx[:, 64:] == -1).all()
    # larger mixture, more frames: against the oracle
    big = __import__("kaldi_amd.synth", fromlist=["x"]).make_gmm(40, 1, 1024, seed=5)
    feats = __import__("kaldi_amd.synth", fromlist=["x"]).make_feats(big, 300, seed=6)
    am2 = gmm.AmDiagGmm(big["weights"], big["means_invvars"], big["inv_vars"], big["pdf_offsets"])
    gc2, _ = oracle.compute_gconsts(big["weights"], big["means_invvars"], big["inv_vars"])
    scores = oracle.diag_gmm_loglikes_mat(gc2, big["means_invvars"], big["inv_vars"], feats)
    widx, wtot = oracle.gselect(scores, 20)
    idx, total = am2.GaussianSelection(feats, 20)
    assert ((idx == widx).all(1)).mean() >= 0.97
    assert abs(total - wtot) <= 1e-3 * len(feats)
