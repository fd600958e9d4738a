"""The feature front end (splice + LDA/MLLT + fMLLR transforms) on the GPU against the
reference's outputs (tests/golden/xf_*.ark, made by the real SpliceFrames and the body of
transform-feats) and against the numpy restatement.  Tolerance: 2e-5 absolute on features of
magnitude <= 10 (BLAS and the kernel sum the <= 92 products in different orders)."""
import numpy as np
import pytest

import oracle
from golden_util import load_ark
from kaldi_amd import _lib, feats, gmm, synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def golden():
    mats = dict(load_ark("xf_mats.ark"))
    return load_ark("xf_feats.ark"), mats


def test_splice_only_is_exact(amlike_device, golden):
    items, _ = golden
    xf = feats.SpliceTransform(13, 3, 3)
    assert (xf.InputDim(), xf.OutputDim()) == (13, 91)
    for (k, x), (k2, want) in zip(items, load_ark("xf_splice.ark")):
        assert k == k2
        np.testing.assert_array_equal(xf.Apply(x), want)


@pytest.mark.parametrize("name", ["lda", "lin"])
def test_splice_transform_matches_the_reference(amlike_device, golden, name):
    items, mats = golden
    xf = feats.SpliceTransform(13, 3, 3, mats[name])
    assert xf.OutputDim() == 40
    for (k, x), (_, want) in zip(items, load_ark("xf_%s.ark" % name)):
        got = xf.Apply(x)
        assert got.shape == want.shape and np.abs(got - want).max() <= TOL
        ours = oracle.transform_feats(oracle.splice_frames(x, 3, 3), mats[name])
        assert np.abs(got - ours).max() <= TOL


def test_second_per_speaker_transform(amlike_device, golden):
    items, mats = golden
    xf = feats.SpliceTransform(13, 3, 3, mats["lda"])
    for (k, x), (_, want) in zip(items, load_ark("xf_lda_fmllr.ark")):
        got = xf.Apply(x, spk_mat=mats["fmllr"])
        assert got.shape == want.shape and np.abs(got - want).max() <= TOL


def test_batch_on_device_respects_utterance_edges_and_picks_speaker_matrices(amlike_device, golden):
    import torch
    items, mats = golden
    rng = np.random.default_rng(8)
    spk = np.stack([mats["fmllr"], (np.eye(40, 41) + 0.05 * rng.standard_normal((40, 41))).astype(np.float32)])
    utt_mat = [1, -1, 0, 1]
    xf = feats.SpliceTransform(13, 3, 3, mats["lda"])
    lens = [x.shape[0] for _, x in items]
    d_in = torch.from_numpy(np.concatenate([x for _, x in items])).cuda()
    d_spk = torch.from_numpy(spk).cuda()
    d_out = torch.zeros((sum(lens), 48), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    xf.ApplyBatchDevice(d_in.data_ptr(), 13, lens, d_out.data_ptr(), 48, d_spk.data_ptr(), 40, 41, utt_mat)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert not got[:, 40:].any()
    r0 = 0
    for (_, x), mi in zip(items, utt_mat):
        want = oracle.transform_feats(oracle.splice_frames(x, 3, 3), mats["lda"])
        if mi >= 0:
            want = oracle.transform_feats(want, spk[mi])
        assert np.abs(got[r0:r0 + x.shape[0], :40] - want).max() <= TOL
        r0 += x.shape[0]


def test_score_raw_equals_transform_then_score(amlike_device, golden):
    items, mats = golden
    m = synth.make_gmm(40, 50, 400, seed=61)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    xf = feats.SpliceTransform(13, 3, 3, mats["lda"])
    for _, x in items:
        for spk_mat in (None, mats["fmllr"]):
            got = feats.score_raw(am, xf, x, spk_mat)
            np.testing.assert_array_equal(got, am.LogLikelihoods(xf.Apply(x, spk_mat)))
            ref_feats = oracle.transform_feats(oracle.splice_frames(x, 3, 3), mats["lda"])
            if spk_mat is not None:
                ref_feats = oracle.transform_feats(ref_feats, spk_mat)
            gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
            want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], ref_feats)
            assert st == 0
            # features differ by <= 2e-5, scores of magnitude ~1e3 for these off-model features
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-3)


def test_errors(amlike_device, golden):
    _, mats = golden
    with pytest.raises(_lib.AmlikeError) as e:
        feats.SpliceTransform(13, 3, 3, mats["lda"][:, :80])
    assert e.value.code == _lib.EDIM and "bad dimension" in str(e.value)
    xf = feats.SpliceTransform(13, 3, 3, mats["lda"])
    with pytest.raises(_lib.AmlikeError) as e:
        xf.Apply(np.zeros((4, 13), np.float32), spk_mat=np.zeros((40, 39), np.float32))
    assert e.value.code == _lib.EDIM
    m = synth.make_gmm(39, 5, 20, seed=62)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    with pytest.raises(_lib.AmlikeError) as e:
        feats.score_raw(am, xf, np.zeros((4, 13), np.float32))
    assert e.value.code == _lib.EDIM and "Dim mismatch" in str(e.value)
