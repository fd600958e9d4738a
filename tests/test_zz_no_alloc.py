"""The entry point that allocates nothing (amlike_gmm_score_dev_noalloc: scratch memory in an
array built into the library) against amlike_gmm_score_dev, through the C ABI: the same bits on
two calls in a row, the second with a smaller input; sizes beyond the built-in memory are refused."""
import numpy as np
import pytest
import torch

from kaldi_amd import _lib, gmm, synth

pytestmark = pytest.mark.gpu


def _score(am, feats, fn, padded=True):
    T, D = feats.shape
    ldp = (D + 3) // 4 * 4 if padded else D  # unpadded: rows that are not 16-byte aligned
    fp = np.zeros((T, ldp), np.float32)
    fp[:, :D] = feats
    d_feats = torch.from_numpy(fp).cuda()
    d_out = torch.full((T, am.NumPdfs()), float("nan"), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    fn(d_feats.data_ptr(), T, ldp, D, d_out.data_ptr(), am.NumPdfs(), stream)
    am.Poll(stream)
    return d_out.cpu().numpy()


@pytest.mark.parametrize("dim,P,G,big_pdf,padded", [(40, 300, 6000, 0, True), (39, 64, 900, 0, True),
                                                    (39, 64, 900, 0, False), (13, 6, 60, 700, False)])
def test_noalloc_matches_the_allocating_entry_point(amlike_device, dim, P, G, big_pdf, padded):
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
        want = _score(am, feats, am.LogLikelihoodsDevice, padded)
        got = _score(am, feats, am.LogLikelihoodsDeviceNoAlloc, padded)
        assert np.isfinite(got).all()
        np.testing.assert_array_equal(got, want)
        if not padded:
            np.testing.assert_array_equal(want, _score(am, feats, am.LogLikelihoodsDevice, True))
    am.close()


def test_noalloc_call_can_be_recorded_into_a_graph(amlike_device):
    """What the entry point is for: under stream capture any device allocation fails the capture.
    The call is recorded into a HIP graph and the replays must give the allocating entry point's
    bits, also after the input buffer's contents change; free device memory does not move."""
    m = synth.make_gmm(40, 120, 2000, seed=41)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    T, P = 700, am.NumPdfs()
    a, b = synth.make_feats(m, T, seed=42), synth.make_feats(m, T, seed=43)
    want_a = _score(am, a, am.LogLikelihoodsDevice)
    want_b = _score(am, b, am.LogLikelihoodsDevice)
    d_feats = torch.from_numpy(a).cuda()
    d_out = torch.zeros((T, P), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        am.LogLikelihoodsDeviceNoAlloc(d_feats.data_ptr(), T, 40, 40, d_out.data_ptr(), P,
                                       torch.cuda.current_stream().cuda_stream)
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_out.cpu().numpy(), want_a)
    d_feats.copy_(torch.from_numpy(b))
    d_out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_out.cpu().numpy(), want_b)
    assert torch.cuda.mem_get_info()[0] >= free_before - (1 << 20)  # (the graph object itself is host memory)
    del graph
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
