"""Batches at and beyond the benchmark's size, device pointers, through the C ABI: more than 2^31
output elements in one call (row offsets need 64-bit arithmetic everywhere), checked through
properties that need no oracle at that size -- every value finite, rows of the big batch
bit-identical to the same frames scored in a small call of their own -- and against the CPU
oracle on a sample of those rows (1e-3 absolute, same best pdf)."""
import numpy as np
import pytest

import oracle
from kaldi_amd import gmm, nnet, synth

pytestmark = pytest.mark.gpu
TOL = 1e-3
P_SWBD = 8600
T_BIG = 251000  # 251000 * 8600 = 2.159e9 > 2^31


def test_gmm_batch_with_more_than_2_31_outputs(amlike_device):
    import torch
    assert T_BIG * P_SWBD > 2 ** 31
    D, P, G = synth.GMM_CONFIGS["swbd_tri4"]
    m = synth.make_gmm(D, P, G, seed=777)
    feats = synth.make_feats(m, T_BIG, seed=1000)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    d_feats = torch.from_numpy(feats).cuda()
    d_out = torch.full((T_BIG, P), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    am.LogLikelihoodsDevice(d_feats.data_ptr(), T_BIG, D, D, d_out.data_ptr(), P)
    am.Poll()
    for lo in range(0, T_BIG, 50000):  # in pieces: isfinite makes a byte per element
        assert bool(torch.isfinite(d_out[lo:lo + 50000]).all())
    sample = None
    for lo, hi in ((0, 70), (T_BIG // 2 - 13, T_BIG // 2 + 120), (249700, 249800), (T_BIG - 333, T_BIG)):
        alone = am.LogLikelihoods(feats[lo:hi])
        big = d_out[lo:hi].cpu().numpy()
        np.testing.assert_array_equal(big, alone)
        sample = (lo, big)
    lo, big = sample
    rows = np.arange(big.shape[0] - 12, big.shape[0])   # the last frames of the batch
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"],
                                        feats[lo + rows])
    assert st == 0
    assert np.abs(big[rows] - want).max() <= TOL
    assert (big[rows].argmax(1) == want.argmax(1)).all()
    am.close()


def test_nnet_batch_with_more_than_2_31_outputs(amlike_device):
    import torch
    comps, priors = synth.make_pnorm_nnet(seed=780)
    net = nnet.AmNnet(comps, priors)
    P = net.OutputDim()
    assert P == P_SWBD
    lengths = synth.utterance_lengths(400, 200, 1500, seed=779)
    keep = int(np.searchsorted(np.cumsum(lengths), T_BIG)) + 1
    lengths = lengths[:keep]
    T = int(lengths.sum())
    assert T * P > 2 ** 31
    feats = np.random.default_rng(1780).normal(size=(T, 40)).astype(np.float32)
    d_in = torch.from_numpy(feats).cuda()
    d_out = torch.full((T, P), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n = net.ComputeBatchDevice(d_in.data_ptr(), 40, lengths, d_out.data_ptr(), P, pad_input=True,
                               mode=nnet.LOGLIKES)
    torch.cuda.synchronize()
    assert n == T
    for lo in range(0, T, 50000):
        assert bool(torch.isfinite(d_out[lo:lo + 50000]).all())
    starts = np.concatenate([[0], np.cumsum(lengths)])
    for u in (0, keep // 2, keep - 1):   # first, middle and last utterance, each scored alone
        lo, hi = int(starts[u]), int(starts[u + 1])
        alone = net.Compute(feats[lo:hi], pad_input=True, mode=nnet.LOGLIKES)
        big = d_out[lo:hi].cpu().numpy()
        assert np.abs(big - alone).max() <= 1e-5
    lo, hi = int(starts[keep - 1]), int(starts[keep])
    want = oracle.nnet_compute(comps, priors, feats[lo:hi], True, 2, 1.0)
    big = d_out[lo:hi].cpu().numpy()
    assert np.abs(big - want).max() <= TOL
    assert (big.argmax(1) == want.argmax(1)).mean() >= 0.9999
    net.close()
