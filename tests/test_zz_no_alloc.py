"""The entry point that allocates nothing (amlike_gmm_score_dev_noalloc: scratch memory in an
array built into the library) against amlike_gmm_score_dev, through the C ABI: the same bits on
two calls in a row, the second with a smaller input; sizes beyond the built-in memory are refused."""
import numpy as np
import pytest
import torch

from kaldi_amd import _lib, gmm, synth

pytestmark = pytest.mark.gpu


def _score(am, feats, fn):
    T, D = feats.shape
    ldp = (D + 3) // 4 * 4
    fp = np.zeros((T, ldp), np.float32)
    fp[:, :D] = feats
    d_feats = torch.from_numpy(fp).cuda()
    d_out = torch.full((T, am.NumPdfs()), float("nan"), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    fn(d_feats.data_ptr(), T, ldp, D, d_out.data_ptr(), am.NumPdfs(), stream)
    am.Poll(stream)
    return d_out.cpu().numpy()


@pytest.mark.parametrize("dim,P,G,big_pdf", [(40, 300, 6000, 0), (39, 64, 900, 0), (13, 6, 60, 700)])
def test_noalloc_matches_the_allocating_entry_point(amlike_device, dim, P, G, big_pdf):
    m = synth.make_gmm(dim, P, G, seed=31)
    if big_pdf:  # one pdf larger than a wave's tiles: partial sums go through the scratch memory too
        big = synth.make_gmm(dim, 1, big_pdf, seed=32)
        m["pdf_offsets"] = np.concatenate([m["pdf_offsets"], m["pdf_offsets"][-1:] + big_pdf]).astype(np.int32)
        for k in ("weights", "means_invvars", "inv_vars", "means"):
            m[k] = np.concatenate([m[k], big[k]])
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    first = synth.make_feats(m, 1500, seed=33)
    second = synth.make_feats(m, 333, seed=34)
    for feats in (first, second):  # two calls in a row, the second smaller
        want = _score(am, feats, am.LogLikelihoodsDevice)
        got = _score(am, feats, am.LogLikelihoodsDeviceNoAlloc)
        assert np.isfinite(got).all()
        np.testing.assert_array_equal(got, want)
    am.close()


def test_noalloc_refuses_what_does_not_fit(amlike_device):
    limit = _lib.lib().amlike_gmm_noalloc_scratch_bytes()
    assert limit == 8 << 20
    m = synth.make_gmm(40, 50, 600, seed=35)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    T = 40000  # operand images alone: 320 + 480 bytes per frame
    d_feats = torch.zeros((T, 40), dtype=torch.float32, device="cuda")
    d_out = torch.zeros((T, 50), dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.AmlikeError) as e:
        am.LogLikelihoodsDeviceNoAlloc(d_feats.data_ptr(), T, 40, 40, d_out.data_ptr(), 50, None)
    assert e.value.code == _lib.EINVAL and "scratch" in str(e.value)
    am.close()
