"""GPU parity of the GMM scoring path: libamlike (HIP, through the C ABI) against
the CPU oracle on the same seeded inputs.  Tolerance: 1e-3 absolute per
(frame, pdf) log-likelihood and identical per-frame argmax pdf (the bar the
benchmark definition states); observed differences are ~1e-5 (fp32 summation order)."""
import numpy as np
import pytest

import oracle
from kaldi_amd import _lib, gmm, synth

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _where(bad, got, want):
    """Which entries are wrong: how many, over which frames and pdfs, and a few of them."""
    rows, cols = np.nonzero(bad)
    some = ["(%d, %d): %r for %r" % (r, c, got[r, c], want[r, c]) for r, c in list(zip(rows, cols))[:6]]
    return ("%d of %d entries; %d frames in [%d, %d] (first %s), %d pdfs in [%d, %d] (first %s); %s"
            % (len(rows), bad.size, len(set(rows)), rows.min(), rows.max(), sorted(set(rows))[:12], len(set(cols)),
               cols.min(), cols.max(), sorted(set(cols))[:12], "; ".join(some)))


def _check(m, feats, got, tol=TOL):
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    assert st == 0
    assert got.shape == want.shape
    assert np.isfinite(got).all(), _where(~np.isfinite(got), got, want)
    err = np.abs(got - want).max()
    assert err < tol, "max abs diff %g" % err
    # argmax agreement, allowing exact ties within the tolerance
    am_got, am_want = got.argmax(1), want.argmax(1)
    bad = am_got != am_want
    if bad.any():
        rows = np.nonzero(bad)[0]
        gap = want[rows, am_want[rows]] - want[rows, am_got[rows]]
        assert (gap < 1e-4).all(), "argmax differs on %d frames" % bad.sum()
    return err


@pytest.mark.parametrize("dim,P,G,T", [
    (40, 7, 50, 33), (39, 30, 200, 64), (40, 100, 1500, 257), (39, 300, 3000, 500),
    (40, 64, 700, 1), (40, 1, 8, 31),
    (13, 20, 160, 100), (24, 33, 400, 70), (60, 50, 500, 130), (8, 5, 30, 40), (64, 9, 90, 65),
    (33, 17, 300, 96), (52, 10, 200, 50),
])
def test_score_matches_oracle(amlike_device, dim, P, G, T):
    m = synth.make_gmm(dim, P, G, seed=100 + P)
    feats = synth.make_feats(m, T, seed=200 + T)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    assert (am.NumPdfs(), am.NumGauss(), am.Dim()) == (P, int(m["pdf_offsets"][-1]), dim)
    got = am.LogLikelihoods(feats)
    _check(m, feats, got)


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_and_frame_counts(amlike_device, seed):
    """Seeded random models (dimension, number and sizes of pdfs) scored on frame counts around the
    kernel's internal boundaries: 32-frame tiles, groups of staged tiles, frame chunks per block,
    the ragged last tile."""
    rng = np.random.default_rng(9000 + seed)
    dim = int(rng.integers(2, 65))
    P = int(rng.integers(1, 400))
    G = max(P, int(P * rng.uniform(1.0, 40.0)))
    edges = [1, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 160, 255, 256, 257, 1023, 2047, 2048, 2049,
             2464, 4095, 4097]
    T = edges[seed % len(edges)] if seed % 3 else int(rng.integers(1, 6000))
    m = synth.make_gmm(dim, P, G, seed=9100 + seed, equal_sizes=bool(seed % 5 == 0))
    feats = synth.make_feats(m, T, seed=9200 + seed)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    _check(m, feats, am.LogLikelihoods(feats))


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_both_kernel_families(amlike_device, mode, monkeypatch):
    """The exact-fp32 MFMA kernel and the bf16x3 kernel give the same answers."""
    monkeypatch.setenv("AMLIKE_GMM_MODE", mode)
    m = synth.make_gmm(40, 120, 2500, seed=71)
    feats = synth.make_feats(m, 333, seed=72)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert ("gmm_score_bf16x3_kernel" in am.kernel_info()) == (mode == "bf16x3")
    _check(m, feats, got)


def test_widely_spread_scores(amlike_device):
    """Scores of one pdf's Gaussians hundreds of nats apart, and large |x|: the running
    (max, sum) must neither overflow nor lose the dominant term."""
    m = synth.make_gmm(40, 30, 600, seed=81)
    m["inv_vars"][::7] *= 400.0            # some very narrow Gaussians
    m["means_invvars"][::7] *= 400.0
    feats = synth.make_feats(m, 200, seed=82) * 2.5
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    assert st == 0
    # relative bar here: |scores| reach 1e5, where one fp32 ulp is already 8e-3
    assert np.all(np.abs(got - want) <= 1e-3 + 2e-6 * np.abs(want))


def test_gconsts_bit_exact_vs_oracle():
    m = synth.make_gmm(40, 50, 600, seed=3)
    a, _ = gmm.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    b, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    assert np.array_equal(a, b)


def test_large_pdfs_are_split_over_waves(amlike_device):
    """pdfs with more Gaussians than one wave holds (UBM-like) take the merge path."""
    rng = np.random.default_rng(5)
    sizes = np.array([700, 3, 1, 250, 97, 2, 513], dtype=np.int64)
    m = synth.make_gmm(40, len(sizes), int(sizes.sum()), seed=9, equal_sizes=True)
    off = np.zeros(len(sizes) + 1, dtype=np.int32)
    np.cumsum(sizes, out=off[1:])
    m["pdf_offsets"] = off
    # renormalise weights per pdf
    for j in range(len(sizes)):
        w = rng.random(sizes[j]).astype(np.float32) + 0.05
        m["weights"][off[j]:off[j + 1]] = w / w.sum()
    feats = synth.make_feats(m, 150, seed=11)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    _check(m, feats, am.LogLikelihoods(feats))


def test_strided_inputs_and_outputs(amlike_device):
    m = synth.make_gmm(39, 40, 400, seed=21)
    T = 77
    base = np.zeros((T, 48), dtype=np.float32)
    base[:, :39] = synth.make_feats(m, T, seed=22)
    feats = base[:, :39]                      # row stride 48 floats
    out_base = np.full((T, 64), np.nan, dtype=np.float32)
    out = out_base[:, :40]
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    am.LogLikelihoods(feats, out=out)
    _check(m, np.ascontiguousarray(feats), np.ascontiguousarray(out))
    assert np.isnan(out_base[:, 40:]).all()   # padding columns untouched


def test_device_pointer_path_unaligned_stride(amlike_device):
    import torch
    m = synth.make_gmm(39, 25, 260, seed=31)
    T = 130
    feats = synth.make_feats(m, T, seed=32)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    d_feats = torch.from_numpy(feats).cuda()          # ld = 39: not 16-byte aligned rows
    d_out = torch.empty((T, 25), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    am.LogLikelihoodsDevice(d_feats.data_ptr(), T, 39, 39, d_out.data_ptr(), 25)
    am.Poll()
    _check(m, feats, d_out.cpu().numpy())


def test_dim_mismatch_and_nonfinite_errors(amlike_device):
    m = synth.make_gmm(40, 10, 100, seed=41)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    with pytest.raises(_lib.AmlikeError) as e:
        am.LogLikelihoods(np.zeros((5, 39), dtype=np.float32))
    assert e.value.code == _lib.EDIM and "Dim mismatch" in str(e.value)
    feats = synth.make_feats(m, 40, seed=42)
    feats[7, 3] = np.inf
    with pytest.raises(_lib.AmlikeError) as e:
        am.LogLikelihoods(feats)
    assert e.value.code == _lib.ENONFINITE and "Invalid answer" in str(e.value)
    feats[7, 3] = np.nan
    with pytest.raises(_lib.AmlikeError) as e:
        am.LogLikelihoods(feats)
    assert e.value.code == _lib.ENONFINITE


def test_zero_weight_gaussians_give_minus_inf_gconst(amlike_device):
    """A zero weight makes gconst -inf (counted, not fatal: gmm/diag-gmm.cc:141-146);
    the pdf's likelihood then ignores that Gaussian."""
    m = synth.make_gmm(40, 6, 60, seed=51)
    m["weights"][3] = 0.0
    gc, bad = gmm.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    assert bad == 1 and np.isneginf(gc[3])
    feats = synth.make_feats(m, 64, seed=52)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    _check(m, feats, am.LogLikelihoods(feats))


def test_component_loglikes_match_oracle(amlike_device):
    """DiagGmm::LogLikelihoods(Matrix): per-Gaussian scores of one pdf / of all."""
    m = synth.make_gmm(40, 12, 300, seed=61)
    feats = synth.make_feats(m, 90, seed=62)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    off = m["pdf_offsets"]
    got = am.ComponentLogLikelihoods(feats, 4, 5)
    want = oracle.diag_gmm_loglikes_mat(am.gconsts[off[4]:off[5]], m["means_invvars"][off[4]:off[5]],
                                        m["inv_vars"][off[4]:off[5]], feats)
    assert np.abs(got - want).max() < 1e-3
    got = am.ComponentLogLikelihoods(feats)
    want = oracle.diag_gmm_loglikes_mat(am.gconsts, m["means_invvars"], m["inv_vars"], feats)
    assert np.abs(got - want).max() < 1e-3


def test_rm_tri1_shape_full_parity(amlike_device):
    """The rm tri1 configuration (D=39, 1500 pdfs, 9000 Gaussians) on 2000 frames."""
    D, P, G = synth.GMM_CONFIGS["rm_tri1"]
    m = synth.make_gmm(D, P, G, seed=777)
    feats = synth.make_feats(m, 2000, seed=778)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    _check(m, feats, am.LogLikelihoods(feats))


@pytest.mark.parametrize("prune", [2.0, 5.0, 30.0])
def test_log_sum_exp_prune(amlike_device, prune):
    """DecodableAmDiagGmmUnmapped's log_sum_exp_prune (gmm/decodable-am-diag-gmm.cc:64):
    terms below max - prune are dropped.  A Gaussian whose score lies within rounding
    of the cutoff may fall on either side, so entries that differ must have one."""
    m = synth.make_gmm(40, 60, 1500, seed=91)
    feats = synth.make_feats(m, 200, seed=92)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats, log_sum_exp_prune=prune)
    assert "pruned" in am.kernel_info()
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"],
                                        feats, prune=prune)
    assert st == 0
    unpruned, _ = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    if prune < 10:
        assert np.abs(want - unpruned).max() > 10 * TOL  # the pruning is visible at this setting
    bad = np.abs(got - want) >= TOL
    assert bad.mean() < 1e-3
    per_gauss = oracle.diag_gmm_loglikes_mat(gc, m["means_invvars"], m["inv_vars"], feats)
    off = m["pdf_offsets"]
    for t, p in zip(*np.nonzero(bad)):
        f = per_gauss[t, off[p]:off[p + 1]]
        cutoff = f.max() - prune
        assert np.abs(f - cutoff).min() < 1e-3, "entry (%d,%d) differs away from the cutoff" % (t, p)


def test_log_sum_exp_prune_in_frame_chunks(amlike_device, monkeypatch):
    """The pruned path bounds its [frames x Gaussians] scratch by working in frame chunks; the
    chunking must not change a bit (1 MiB of scratch = chunks of 174 frames here), also when the
    last chunk is shorter than 64 frames (which takes the small-batch score kernel)."""
    m = synth.make_gmm(40, 60, 1500, seed=91)
    feats = synth.make_feats(m, 560, seed=93)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    whole = am.LogLikelihoods(feats, log_sum_exp_prune=5.0)
    monkeypatch.setenv("AMLIKE_GMM_PRUNE_SCRATCH_MB", "1")
    chunked = am.LogLikelihoods(feats, log_sum_exp_prune=5.0)
    assert np.abs(chunked - whole).max() <= 2e-5   # the two score kernels differ in summation order
    np.testing.assert_array_equal(chunked[:522], whole[:522])
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"],
                                        feats, prune=5.0)
    assert st == 0 and (np.abs(chunked - want) >= TOL).mean() < 1e-3


@pytest.mark.parametrize("mode", ["diff-f16x2", "diff-bf16x3"])
def test_differenced_kernel_operand_types(amlike_device, mode, monkeypatch):
    monkeypatch.setenv("AMLIKE_GMM_MODE", mode)
    m = synth.make_gmm(40, 120, 2500, seed=71)
    feats = synth.make_feats(m, 333, seed=72)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert "gmm_score_diff_kernel" in am.kernel_info()
    assert ("f16x2" in am.kernel_info()) == (mode == "diff-f16x2")
    _check(m, feats, got)


def test_sum_overflow_is_repaired_exactly(amlike_device):
    """A narrow Gaussian that beats the pdf's broadest one by more than 2^128 overflows the
    reference-relative sum of the default kernel; those (frame, pdf) pairs are recomputed."""
    rng = np.random.default_rng(5)
    D, P = 40, 12
    per = 6
    G = P * per
    means = rng.standard_normal((G, D)).astype(np.float32)
    var = np.full((G, D), 100.0, np.float32)
    var[1::per] = 0.01  # second Gaussian of every pdf: very narrow
    w = np.full(G, 1.0 / per, np.float32)
    iv = (1.0 / var).astype(np.float32)
    m = dict(weights=w, inv_vars=iv, means_invvars=(means * iv).astype(np.float32),
             pdf_offsets=np.arange(0, G + 1, per, dtype=np.int32), dim=D)
    feats = (rng.standard_normal((64, D)) * 3).astype(np.float32)
    feats[::4] = means[1::per][rng.integers(0, P, 16)]  # frames sitting on narrow Gaussians
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert "gmm_score_diff_kernel" in am.kernel_info()
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    per_gauss = oracle.diag_gmm_loglikes_mat(gc, m["means_invvars"], m["inv_vars"], feats)
    spread = per_gauss.reshape(64, P, per).max(2) - per_gauss.reshape(64, P, per)[:, :, 0]
    assert (spread > 100).any()  # the case is really exercised (100 nats > 2^128)
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    assert st == 0
    assert np.abs(got - want).max() <= TOL * max(1.0, np.abs(want).max() / 1000)


def test_features_outside_fp16_range_use_the_wide_range_kernel(amlike_device):
    m = synth.make_gmm(40, 30, 400, seed=93)
    feats = synth.make_feats(m, 96, seed=94)
    feats[17, 3] = 3.0e4   # x^2 = 9e8: far outside the scaled fp16 operand range
    feats[40, :] *= 2000.0
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert "f16x2" in am.kernel_info()
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    assert st == 0 and np.isfinite(got).all()
    # scores of the outlier frames are ~1e9 in magnitude: compare relatively there
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=TOL)
    ok = np.ones(96, bool)
    ok[[17, 40]] = False
    assert np.abs(got[ok] - want[ok]).max() <= TOL


def test_model_outside_fp16_range_uses_bf16x3_operands(amlike_device):
    m = synth.make_gmm(24, 20, 200, seed=95)
    m["inv_vars"][5, 2] = 3.0e7          # a variance of 3e-8: cannot be balanced into fp16
    m["means_invvars"][5, 2] = 1.0e7
    feats = synth.make_feats(m, 50, seed=96)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert "gmm_score_diff_kernel" in am.kernel_info() and "bf16x3" in am.kernel_info()
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, _ = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=TOL)


def test_more_overflows_than_the_repair_list_holds(amlike_device):
    """When the overflowed pairs outnumber the repair list, the wide-range kernel redoes the
    launch (decided on the device); the answers must still match the oracle."""
    rng = np.random.default_rng(15)
    D, P, per = 40, 12, 4
    G = P * per
    means = rng.standard_normal((G, D)).astype(np.float32)
    var = np.full((G, D), 100.0, np.float32)
    var[1::per] = 0.01
    # every pdf has its narrow Gaussian at the same place (near the origin, so that the fp32
    # cancellation in x^2/var - 2 x mean/var + mean^2/var stays well inside the tolerance) ...
    means[1::per] = 0.1 * means[1]
    iv = (1.0 / var).astype(np.float32)
    m = dict(weights=np.full(G, 1.0 / per, np.float32), inv_vars=iv,
             means_invvars=(means * iv).astype(np.float32),
             pdf_offsets=np.arange(0, G + 1, per, dtype=np.int32), dim=D)
    T = 1200
    # ... and every frame sits there: each (frame, pdf) pair overflows the reference-relative sum
    feats = (means[1] + 0.05 * rng.standard_normal((T, D))).astype(np.float32)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    per_gauss = oracle.diag_gmm_loglikes_mat(gc, m["means_invvars"], m["inv_vars"], feats)
    spread = per_gauss.reshape(T, P, per).max(2) - per_gauss.reshape(T, P, per)[:, :, 0]
    assert (spread > 100).sum() > 4096  # more than the list's minimum capacity
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats)
    assert st == 0
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=TOL)


@pytest.mark.parametrize("per", [1, 2, 3])
def test_many_tiny_pdfs(amlike_device, per):
    P = 700
    m = synth.make_gmm(40, P, P * per, seed=300 + per, equal_sizes=True)
    assert int(m["pdf_offsets"][-1]) == P * per
    feats = synth.make_feats(m, 97, seed=301)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    _check(m, feats, am.LogLikelihoods(feats))


def test_submit_wait_matches_the_synchronous_call(amlike_device):
    """Two utterances in flight at a time, page-locked buffers; a third submission is refused
    until one has been waited for; results equal amlike_gmm_score's bit for bit."""
    m = synth.make_gmm(40, 300, 3000, seed=41)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    lens = [700, 64, 1301, 5, 999]
    feats = [synth.make_feats(m, t, seed=50 + i) for i, t in enumerate(lens)]
    want = [am.LogLikelihoods(f) for f in feats]
    pin_in = [_lib.PinnedArray(max(lens), 40) for _ in range(2)]
    pin_out = [_lib.PinnedArray(max(lens), 300) for _ in range(2)]
    tickets = {}

    def submit(i):
        s = i % 2
        pin_in[s].a[:lens[i]] = feats[i]
        tickets[i] = am.Submit(pin_in[s].a[:lens[i]], pin_out[s].a[:lens[i]])

    submit(0)
    submit(1)
    with pytest.raises(_lib.AmlikeError) as e:
        am.Submit(feats[2], np.empty((lens[2], 300), np.float32))
    assert e.value.code == _lib.EBUSY
    for i in range(len(lens)):
        am.Wait(tickets[i])
        got = pin_out[i % 2].a[:lens[i]].copy()
        np.testing.assert_array_equal(got, want[i])
        if i + 2 < len(lens):
            submit(i + 2)
    with pytest.raises(_lib.AmlikeError):
        am.Wait(tickets[0])  # no longer in flight
    # ordinary (pageable) buffers work too, and a strided output
    out = np.zeros((lens[3], 320), np.float32)
    t = am.Submit(feats[3], out[:, :300])
    am.Wait(t)
    np.testing.assert_array_equal(out[:, :300], want[3])
    assert not out[:, 300:].any()
    for p in pin_in + pin_out:
        p.close()


def test_submit_reports_non_finite_answers_at_wait(amlike_device):
    m = synth.make_gmm(13, 5, 20, seed=43)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    good = synth.make_feats(m, 30, seed=44)
    bad = good.copy()
    bad[7, 3] = np.nan
    o1, o2 = np.empty((30, 5), np.float32), np.empty((30, 5), np.float32)
    t1 = am.Submit(bad, o1)
    t2 = am.Submit(good, o2)
    with pytest.raises(_lib.AmlikeError) as e:
        am.Wait(t1)
    assert e.value.code == _lib.ENONFINITE
    am.Wait(t2)  # the flag of the first does not leak into the second
    np.testing.assert_array_equal(o2, am.LogLikelihoods(good))


def test_swbd_tri4_full_size(amlike_device, monkeypatch):
    """The benchmark's model at its full size (D=40, 8600 pdfs, 200 000 Gaussians): the oracle
    on a sample of frames, and properties that need no oracle on all of them -- the exact-fp32
    kernel family as an independent second opinion, frames scored alone or in any batch and
    order giving the same bits, and the bounds max_m f_m <= LSE <= max_m f_m + log M."""
    D, P, G = synth.GMM_CONFIGS["swbd_tri4"]
    m = synth.make_gmm(D, P, G, seed=777)
    T = 4100
    feats = synth.make_feats(m, T, seed=778)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    got = am.LogLikelihoods(feats)
    assert "gmm_score_diff_kernel" in am.kernel_info() and "f16x2" in am.kernel_info()
    # the pdfs fill the wave chunks' slots almost to what the even-slot padding allows (0.979)
    fill = float(am.kernel_info().split("fill=")[1].split()[0])
    assert fill >= 0.97, am.kernel_info()
    assert np.isfinite(got).all()
    # (a) oracle on 48 frames spread over the batch
    rows = np.linspace(0, T - 1, 48).astype(int)
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want, st = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats[rows])
    assert st == 0
    assert np.abs(got[rows] - want).max() <= TOL
    assert (got[rows].argmax(1) == want.argmax(1)).all()
    # (b) batch composition and order do not change a frame's scores
    perm = np.random.default_rng(5).permutation(T)
    np.testing.assert_array_equal(am.LogLikelihoods(feats[perm]), got[perm])
    np.testing.assert_array_equal(am.LogLikelihoods(feats[1000:1037]), got[1000:1037])
    # (c) bounds from the per-Gaussian scores of a few pdfs
    for j in (0, 4321, P - 1):
        comp = am.ComponentLogLikelihoods(feats[:512], j, j + 1)
        mx = comp.max(1)
        assert (got[:512, j] >= mx - 1e-3).all()
        assert (got[:512, j] <= mx + np.log(comp.shape[1]) + 1e-3).all()
    am.close()
    # (d) the exact-fp32 kernel family on all frames
    monkeypatch.setenv("AMLIKE_GMM_MODE", "f32")
    am32 = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    ref = am32.LogLikelihoods(feats)
    assert "gmm_score_kernel" in am32.kernel_info()
    assert np.abs(ref - got).max() <= TOL
    assert (ref.argmax(1) == got.argmax(1)).mean() >= 0.9999


def test_threads_sharing_one_handle_and_threads_with_their_own(amlike_device):
    """The *-parallel binaries of the reference score from several threads over one shared model
    (gmmbin/gmm-latgen-faster-parallel.cc:138-155).  Host-pointer calls on one handle are
    serialised inside the library; separate handles run concurrently.  Either way every thread
    gets the bits a lone call gets."""
    import threading
    m = synth.make_gmm(40, 200, 3000, seed=81)
    shared = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    own = [gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"]) for _ in range(3)]
    feats = [synth.make_feats(m, 200 + 131 * i, seed=90 + i) for i in range(6)]
    want = [shared.LogLikelihoods(f) for f in feats]
    results, errors = {}, []

    def work(tag, am, idx):
        try:
            for rep in range(5):
                for i in idx:
                    results[(tag, i, rep)] = am.LogLikelihoods(feats[i])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=("s%d" % t, shared, [t, t + 3])) for t in range(3)]
    threads += [threading.Thread(target=work, args=("o%d" % t, own[t], [t, 5 - t])) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 6 * 2 * 5
    for (tag, i, rep), got in results.items():
        np.testing.assert_array_equal(got, want[i])


def test_two_threads_on_two_streams_of_one_handle_device_pointers(amlike_device):
    """Device-pointer calls on ONE handle from two threads, each with a stream of its own: the
    library keeps separate working memory per stream, so the launches may overlap on the GPU and
    every result still carries the bits of a lone call.  Covers the split-pdf partial sums, the
    repair list of the differenced kernel and the pruned path (the buffers that used to be shared)."""
    import threading
    import torch
    m = synth.make_gmm(39, 120, 2600, seed=83)
    big = synth.make_gmm(39, 1, 700, seed=84)  # a pdf wide enough to be split over several waves
    mm = {k: np.concatenate([m[k], big[k]]) for k in ("weights", "means_invvars", "inv_vars")}
    mm["pdf_offsets"] = np.concatenate([m["pdf_offsets"], [m["pdf_offsets"][-1] + 700]]).astype(np.int32)
    am = gmm.AmDiagGmm(mm["weights"], mm["means_invvars"], mm["inv_vars"], mm["pdf_offsets"])
    P = am.NumPdfs()
    sizes = [[3000, 517, 2048, 64], [33, 4100, 1000, 2500]]
    feats = [[torch.from_numpy(synth.make_feats(m, T, seed=100 + 10 * t + i)).cuda() for i, T in enumerate(sizes[t])]
             for t in range(2)]
    prune = [-1.0, 8.0]

    def score(f, stream, pr):
        out = torch.empty((f.shape[0], P), dtype=torch.float32, device="cuda")
        am.LogLikelihoodsDevice(f.data_ptr(), f.shape[0], f.stride(0), f.shape[1], out.data_ptr(), P,
                                stream=stream, log_sum_exp_prune=pr)
        return out

    # what a lone call gives
    want = [[[score(f, None, pr) for pr in prune] for f in feats[t]] for t in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    errors = []
    start = threading.Barrier(2)

    def work(t):
        try:
            start.wait()
            for rep in range(6):
                for f in feats[t]:
                    got[t].append([score(f, streams[t].cuda_stream, pr) for pr in prune])
            streams[t].synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for t in range(2):
        assert len(got[t]) == 6 * len(feats[t])
        for k, pair in enumerate(got[t]):
            for j in range(2):
                assert torch.equal(pair[j], want[t][k % len(feats[t])][j]), (t, k, j)


@pytest.mark.parametrize("T", [64, 500, 2000])
def test_repeated_small_calls_replay_a_recorded_graph(amlike_device, T, monkeypatch):
    """With AMLIKE_GMM_GRAPHS=1 at the creation of the handle (off by default: the replay measured no
    faster than the plain launches), small device-pointer calls that come again with the same buffers
    and shape are recorded as a HIP graph on their second appearance and replayed from the third: every call must give the bits
    of the first, also after the features changed in place, after a larger call on the same stream
    has moved the working memory, and for a model with a pdf split over several waves."""
    import torch
    m = synth.make_gmm(39, 60, 1300, seed=87)
    big = synth.make_gmm(39, 1, 700, seed=88)
    mm = {k: np.concatenate([m[k], big[k]]) for k in ("weights", "means_invvars", "inv_vars")}
    mm["pdf_offsets"] = np.concatenate([m["pdf_offsets"], [m["pdf_offsets"][-1] + 700]]).astype(np.int32)
    monkeypatch.setenv("AMLIKE_GMM_GRAPHS", "1")
    am = gmm.AmDiagGmm(mm["weights"], mm["means_invvars"], mm["inv_vars"], mm["pdf_offsets"])
    P = am.NumPdfs()
    stream = torch.cuda.Stream()
    f1 = torch.from_numpy(synth.make_feats(m, T, seed=89)).cuda()
    f2 = torch.from_numpy(synth.make_feats(m, T, seed=90)).cuda()
    feats = f1.clone()
    out = torch.empty((T, P), dtype=torch.float32, device="cuda")

    def call():
        am.LogLikelihoodsDevice(feats.data_ptr(), T, feats.stride(0), feats.shape[1], out.data_ptr(), P,
                                stream=stream.cuda_stream)
        stream.synchronize()
        return out.clone()

    first = call()
    gc, _ = oracle.compute_gconsts(mm["weights"], mm["means_invvars"], mm["inv_vars"])
    want, _ = oracle.gmm_compute_likes(gc, mm["means_invvars"], mm["inv_vars"], mm["pdf_offsets"], f1.cpu().numpy()[:32])
    assert np.abs(first.cpu().numpy()[:32] - want).max() < 1e-3
    for _ in range(4):  # recorded on the second call, replayed afterwards
        assert torch.equal(call(), first)
    feats.copy_(f2)  # same buffers, new contents
    second = call()
    assert not torch.equal(second, first)
    feats.copy_(f1)
    assert torch.equal(call(), first)
    # a larger call on the same stream grows the working memory under the recorded graphs
    fbig = torch.from_numpy(synth.make_feats(m, 6000, seed=91)).cuda()
    obig = torch.empty((6000, P), dtype=torch.float32, device="cuda")
    am.LogLikelihoodsDevice(fbig.data_ptr(), 6000, fbig.stride(0), fbig.shape[1], obig.data_ptr(), P,
                            stream=stream.cuda_stream)
    stream.synchronize()
    for _ in range(3):
        assert torch.equal(call(), first)
    feats.copy_(f2)
    assert torch.equal(call(), second)
    am.Poll(stream.cuda_stream)


def test_pruned_path_beyond_65535_frames_in_one_chunk(amlike_device):
    """A small model with many frames: the pruned log-sum-exp pass covers them in launches the
    grid can hold (one block row per frame)."""
    m = synth.make_gmm(13, 4, 40, seed=85)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    T = 70001
    feats = synth.make_feats(m, T, seed=86)
    got = am.LogLikelihoods(feats, log_sum_exp_prune=50.0)
    want = am.LogLikelihoods(feats)
    assert got.shape == (T, 4)
    np.testing.assert_allclose(got, want, atol=1e-3)
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    tail, _ = oracle.gmm_compute_likes(gc, m["means_invvars"], m["inv_vars"], m["pdf_offsets"], feats[-3:])
    np.testing.assert_allclose(got[-3:], tail, atol=1e-3)


@pytest.mark.parametrize("dim,P,G,T", [(100, 12, 150, 70), (72, 30, 300, 5), (117, 3, 10, 200)])
def test_dimensions_beyond_the_fused_kernels(amlike_device, dim, P, G, T):
    """Feature dimensions above 64 have no fused kernel; the model is scored through the
    per-Gaussian score GEMM and a log-sum-exp pass, to the same tolerance."""
    m = synth.make_gmm(dim, P, G, seed=400 + dim)
    feats = synth.make_feats(m, T, seed=401)
    am = gmm.AmDiagGmm(m["weights"], m["means_invvars"], m["inv_vars"], m["pdf_offsets"])
    assert "no fused kernel" in am.kernel_info()
    _check(m, feats, am.LogLikelihoods(feats))
    comp = am.ComponentLogLikelihoods(feats[:3], 0, 2)
    gc, _ = oracle.compute_gconsts(m["weights"], m["means_invvars"], m["inv_vars"])
    want = oracle.diag_gmm_loglikes_mat(gc, m["means_invvars"], m["inv_vars"], feats[:3])
    assert np.abs(comp - want[:, :comp.shape[1]]).max() < TOL


def test_like_the_reference_unit_test_of_diag_gmm(amlike_device):
    """The reference's own check (gmm/diag-gmm-test.cc:86-141): for small random mixtures
    (1-10 dimensions, 1-10 components) the log-likelihood equals the log-sum of the component
    densities written out from first principles; and (gmm/am-diag-gmm-test.cc:32-72) a pdf
    scored inside an acoustic model equals the same mixture scored on its own."""
    rng = np.random.default_rng(2024)
    for trial in range(25):
        dim, nmix = int(rng.integers(1, 11)), int(rng.integers(1, 11))
        feat = rng.standard_normal(dim)
        means = rng.standard_normal((nmix, dim))
        var = np.exp(rng.standard_normal((nmix, dim))) + 1e-5
        w = rng.uniform(size=nmix)
        w /= w.sum()
        comp = np.log(w) - 0.5 * (np.log(2 * np.pi) + np.log(var) + (feat - means) ** 2 / var).sum(1)
        loglike = np.logaddexp.reduce(comp)
        iv = (1.0 / var).astype(np.float32)
        miv = (means / var).astype(np.float32)
        one = gmm.AmDiagGmm(w.astype(np.float32), miv, iv, np.array([0, nmix], np.int32))
        got = one.LogLikelihoods(feat[None, :].astype(np.float32))[0, 0]
        assert abs(got - loglike) <= 1e-3 * max(1.0, abs(loglike)), (trial, got, loglike)
        per = one.ComponentLogLikelihoods(feat[None, :].astype(np.float32))[0]
        assert np.abs(per - comp).max() <= 1e-3 * max(1.0, np.abs(comp).max())
        # the same mixture as pdf 1 of a three-pdf model
        other = synth.make_gmm(dim, 2, 6, seed=trial)
        am3 = gmm.AmDiagGmm(
            np.concatenate([other["weights"][:3], w.astype(np.float32), other["weights"][3:]]),
            np.concatenate([other["means_invvars"][:3], miv, other["means_invvars"][3:]]),
            np.concatenate([other["inv_vars"][:3], iv, other["inv_vars"][3:]]),
            np.array([0, 3, 3 + nmix, 6 + nmix], np.int32))
        got3 = am3.LogLikelihoods(feat[None, :].astype(np.float32))[0, 1]
        assert abs(got3 - got) <= 2e-4 * max(1.0, abs(loglike))
