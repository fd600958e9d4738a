"""GPU parity of the nnet2 forward path: libamlike (HIP, through the C ABI) against the
CPU oracle.  Tolerance 1e-3 absolute on log-posteriors / pseudo-log-likelihoods (the
benchmark definition's bar) and 1e-5 on posteriors; observed ~1e-6."""
import numpy as np
import pytest

import oracle
from kaldi_amd import _lib, nnet, synth

pytestmark = pytest.mark.gpu


def _nets():
    rng = np.random.default_rng(5)

    def aff(n_in, n_out):
        return dict(type="AffineComponent",
                    linear=(rng.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32),
                    bias=rng.normal(size=n_out).astype(np.float32))
    small = synth.make_pnorm_nnet(input_dim=13, left=2, right=3, hidden=60, pnorm_out=12,
                                  num_hidden=2, num_pdfs=100, seed=3)
    mixup = synth.make_pnorm_nnet(input_dim=13, left=2, right=3, hidden=60, pnorm_out=12,
                                  num_hidden=2, num_pdfs=100, mix_up=150, seed=4)
    mid = synth.make_pnorm_nnet(input_dim=40, left=4, right=4, hidden=400, pnorm_out=40,
                                num_hidden=3, num_pdfs=1000, seed=6)
    misc = ([dict(type="SpliceComponent", input_dim=10, left=1, right=1, const_dim=3),
             aff(24, 16), dict(type="SigmoidComponent", dim=16),
             aff(16, 16), dict(type="TanhComponent", dim=16),
             aff(16, 18), dict(type="RectifiedLinearComponent", dim=18),
             dict(type="PnormComponent", input_dim=18, output_dim=6, p=3.0),
             aff(6, 7), dict(type="SoftmaxComponent", dim=7)], np.full(7, 1 / 7, np.float32))
    nosplice = ([aff(20, 33), dict(type="PnormComponent", input_dim=33, output_dim=11, p=1.0),
                 dict(type="NormalizeComponent", dim=11), aff(11, 9),
                 dict(type="SoftmaxComponent", dim=9)], np.full(9, 1 / 9, np.float32))
    return dict(small=small, mixup=mixup, mid=mid, misc=misc, nosplice=nosplice)


@pytest.mark.parametrize("name", ["small", "mixup", "mid", "misc", "nosplice"])
@pytest.mark.parametrize("mode", [nnet.POSTERIORS, nnet.LOG, nnet.LOGLIKES])
@pytest.mark.parametrize("pad", [True, False])
def test_forward_matches_oracle(amlike_device, name, mode, pad):
    comps, priors = _nets()[name]
    net = nnet.AmNnet(comps, priors)
    x = np.random.default_rng(11).normal(size=(37, net.InputDim())).astype(np.float32)
    got = net.Compute(x, pad_input=pad, mode=mode, prob_scale=0.7)
    want = oracle.nnet_compute(comps, priors, x, pad, mode, 0.7)
    assert got.shape == want.shape
    tol = 1e-5 if mode == nnet.POSTERIORS else 1e-3
    assert np.abs(got - want).max() < tol
    if mode == nnet.POSTERIORS:
        assert np.allclose(got.sum(1), 1.0, atol=1e-4)
    assert (got.argmax(1) == want.argmax(1)).all()


def test_batch_respects_utterance_boundaries(amlike_device):
    import torch
    comps, priors = _nets()["mid"]
    net = nnet.AmNnet(comps, priors)
    rng = np.random.default_rng(21)
    lens = [9, 130, 1, 57, 300]
    utts = [rng.normal(size=(n, 40)).astype(np.float32) for n in lens]
    want = np.concatenate([oracle.nnet_compute(comps, priors, u, True, 2, 1.0) for u in utts])
    d_in = torch.from_numpy(np.concatenate(utts)).cuda()
    d_out = torch.empty((sum(lens), net.OutputDim()), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n = net.ComputeBatchDevice(d_in.data_ptr(), 40, lens, d_out.data_ptr(), net.OutputDim(),
                               pad_input=True, mode=nnet.LOGLIKES)
    torch.cuda.synchronize()
    assert n == sum(lens)
    assert np.abs(d_out.cpu().numpy() - want).max() < 1e-3


def test_swbd_pnorm_shape(amlike_device):
    """The benchmark topology (4 x 2000 -> 200 p-norm, 8600-way softmax) on 300 frames."""
    comps, priors = synth.make_pnorm_nnet(seed=780)
    net = nnet.AmNnet(comps, priors)
    assert (net.InputDim(), net.OutputDim(), net.LeftContext(), net.RightContext()) == (40, 8600, 4, 4)
    x = np.random.default_rng(31).normal(size=(300, 40)).astype(np.float32)
    got = net.Compute(x, mode=nnet.LOGLIKES)
    want = oracle.nnet_compute(comps, priors, x, True, 2, 1.0)
    assert np.abs(got - want).max() < 1e-3
    assert (got.argmax(1) == want.argmax(1)).mean() >= 0.9999


def test_errors(amlike_device):
    comps, priors = _nets()["small"]
    net = nnet.AmNnet(comps, priors)
    with pytest.raises(ValueError):
        net.Compute(np.zeros((5, 13), np.float32), pad_input=False)   # 5 frames < context 6
    bad = [dict(type="PnormComponent", input_dim=10, output_dim=3, p=2.0)]
    with pytest.raises(_lib.AmlikeError):
        nnet.AmNnet(bad, None)
    with pytest.raises(_lib.AmlikeError) as e:
        nnet.AmNnet(comps, priors[:50]).Compute(np.zeros((20, 13), np.float32), mode=nnet.LOGLIKES)
    assert "Priors" in str(e.value)


def test_exact_fp32_gemm_gives_the_same_answers(amlike_device, monkeypatch):
    """AMLIKE_NNET_GEMM=f32 selects the exact-fp32 MFMA GEMM; the default bf16x3 GEMM must
    agree with it (and with the oracle) to fp32 rounding."""
    comps, priors = synth.make_pnorm_nnet(input_dim=40, left=4, right=4, hidden=300, pnorm_out=60,
                                          num_hidden=2, num_pdfs=500, mix_up=0, seed=61)
    feats = np.random.default_rng(62).standard_normal((150, 40)).astype(np.float32) * 3.0
    outs = {}
    for gemm in ("bf16x3", "f32"):
        monkeypatch.setenv("AMLIKE_NNET_GEMM", gemm)
        net = nnet.AmNnet(comps, priors)
        outs[gemm] = net.Compute(feats, pad_input=True, mode=nnet.LOGLIKES, prob_scale=1.0)
        net.close()
    want = oracle.nnet_compute(comps, priors, feats, pad_input=True, mode=2, prob_scale=1.0)
    assert np.abs(outs["bf16x3"] - outs["f32"]).max() <= 2e-4
    for o in outs.values():
        assert np.abs(o - want).max() <= 1e-3


@pytest.mark.parametrize("hidden,pnorm_out", [(400, 40), (320, 40), (250, 50), (210, 30)])
def test_fused_affine_pnorm_matches_unfused_and_oracle(amlike_device, hidden, pnorm_out, monkeypatch):
    """Group sizes 10, 8 and 5 run the fused affine + p-norm kernel, 7 the separate kernels;
    all must agree with the oracle, and fused with unfused to fp32 rounding."""
    comps, priors = synth.make_pnorm_nnet(input_dim=24, left=2, right=2, hidden=hidden,
                                          pnorm_out=pnorm_out, num_hidden=2, num_pdfs=333, seed=71)
    feats = np.random.default_rng(72).standard_normal((203, 24)).astype(np.float32) * 2.0
    want = oracle.nnet_compute(comps, priors, feats, pad_input=True, mode=2, prob_scale=1.0)
    outs = {}
    for nofuse in ("", "1"):
        if nofuse:
            monkeypatch.setenv("AMLIKE_NNET_NOFUSE", "1")
        else:
            monkeypatch.delenv("AMLIKE_NNET_NOFUSE", raising=False)
        net = nnet.AmNnet(comps, priors)
        outs[nofuse] = net.Compute(feats, pad_input=True, mode=nnet.LOGLIKES, prob_scale=1.0)
        net.close()
        assert np.abs(outs[nofuse] - want).max() <= 1e-3
    assert np.abs(outs[""] - outs["1"]).max() <= 2e-4


@pytest.mark.parametrize("pad", [True, False])
def test_submit_wait_matches_the_synchronous_call(amlike_device, pad):
    comps, priors = _nets()["mid"]
    net = nnet.AmNnet(comps, priors)
    rng = np.random.default_rng(77)
    lens = [120, 33, 410, 9, 77]
    ctx = 0 if pad else net.LeftContext() + net.RightContext()
    utts = [rng.normal(size=(n, 40)).astype(np.float32) for n in lens]
    want = [net.Compute(u, pad_input=pad, mode=nnet.LOGLIKES, prob_scale=0.5) for u in utts]
    pin_in = [_lib.PinnedArray(max(lens), 40) for _ in range(2)]
    pin_out = [_lib.PinnedArray(max(lens), net.OutputDim()) for _ in range(2)]
    tickets = {}

    def submit(i):
        s = i % 2
        pin_in[s].a[:lens[i]] = utts[i]
        tickets[i] = net.Submit(pin_in[s].a[:lens[i]], pin_out[s].a[:lens[i] - ctx], pad_input=pad,
                                mode=nnet.LOGLIKES, prob_scale=0.5)

    submit(0)
    submit(1)
    with pytest.raises(_lib.AmlikeError) as e:
        net.Submit(utts[2], np.empty((lens[2] - ctx, net.OutputDim()), np.float32), pad_input=pad)
    assert e.value.code == _lib.EBUSY
    for i in range(len(lens)):
        net.Wait(tickets[i])
        np.testing.assert_array_equal(pin_out[i % 2].a[:lens[i] - ctx], want[i])
        if i + 2 < len(lens):
            submit(i + 2)
    for p in pin_in + pin_out:
        p.close()


def test_fp16_operands_on_trial_fall_back_when_the_input_does_not_fit(amlike_device, monkeypatch):
    """Layers whose input is not behind a NormalizeComponent (the LDA-like first affine layers)
    use fp16x2 operands only if every input value fits; one value beyond 32768 (or a NaN)
    switches that call to the bf16x3 kernels, decided on the device."""
    comps, priors = _nets()["mid"]
    rng = np.random.default_rng(91)
    x = rng.normal(size=(300, 40)).astype(np.float32)
    x_big = x.copy()
    x_big[17, 3] = 5.0e4
    x_big[250, 39] = -3.0e5
    trial = nnet.AmNnet(comps, priors)
    monkeypatch.setenv("AMLIKE_NNET_NOF16TRY", "1")
    plain = nnet.AmNnet(comps, priors)  # the first layers on bf16x3 operands, no trial
    monkeypatch.delenv("AMLIKE_NNET_NOF16TRY")
    for mode, tol in ((nnet.POSTERIORS, 1e-5), (nnet.LOGLIKES, 1e-3)):
        want = oracle.nnet_compute(comps, priors, x, True, mode, 1.0)
        got = trial.Compute(x, mode=mode)
        assert np.abs(got - want).max() < tol
        assert np.abs(plain.Compute(x, mode=mode) - want).max() < tol
        # out-of-range input: the very kernels of the plain handle run, so the bits agree
        np.testing.assert_array_equal(trial.Compute(x_big, mode=mode), plain.Compute(x_big, mode=mode))
        # and the next call with ordinary input is back on the fast path, unaffected
        np.testing.assert_array_equal(trial.Compute(x, mode=mode), got)
    x_nan = x.copy()
    x_nan[5, 0] = np.nan
    a, b = trial.Compute(x_nan), plain.Compute(x_nan)
    assert np.isnan(a[:10]).any()
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))


@pytest.mark.parametrize("name", ["small", "mid", "mixup"])
def test_normalised_rows_written_straight_as_operand_planes(amlike_device, name, monkeypatch):
    """A NormalizeComponent that only feeds an fp16x2 affine stage writes that stage's operand
    planes itself; the answers must agree with the two-pass route (fp32 rows, then the split
    pass) to the last few bits (the row's sum of squares is added up in another order)."""
    comps, priors = _nets()[name]
    x = np.random.default_rng(123).normal(size=(333, comps[0]["input_dim"])).astype(np.float32)
    direct = nnet.AmNnet(comps, priors).Compute(x, mode=nnet.LOGLIKES)
    monkeypatch.setenv("AMLIKE_NNET_NO_DIRECT_PLANES", "1")
    two_pass = nnet.AmNnet(comps, priors).Compute(x, mode=nnet.LOGLIKES)
    np.testing.assert_allclose(direct, two_pass, rtol=2e-5, atol=2e-5)
    want = oracle.nnet_compute(comps, priors, x, True, nnet.LOGLIKES, 1.0)
    assert np.abs(direct - want).max() < 1e-3


@pytest.mark.parametrize("cus,rows", [("0", 300), ("2", 1000), ("3", 1000), ("1", 70)])
@pytest.mark.parametrize("mode", [nnet.POSTERIORS, nnet.LOG, nnet.LOGLIKES])
def test_output_layer_and_softmax_in_one_kernel(amlike_device, cus, rows, mode, monkeypatch):
    """The fused output-layer + softmax kernel (k = 200, 8600 units) against the separate GEMM and
    softmax kernels and against the oracle, over the job shapes it cuts the rows into: 256-row jobs
    with 64 rows per wave, short jobs with 32 rows per wave, ragged last jobs (AMLIKE_NNET_OUT_CUS
    makes the host plan jobs as for a chip with that many compute units)."""
    if cus != "0":
        monkeypatch.setenv("AMLIKE_NNET_OUT_CUS", cus)
    comps, priors = synth.make_pnorm_nnet(seed=780)
    x = np.random.default_rng(41).normal(size=(rows, 40)).astype(np.float32)
    net = nnet.AmNnet(comps, priors)
    got = net.Compute(x, mode=mode, prob_scale=0.9)
    net.close()
    monkeypatch.setenv("AMLIKE_NNET_NO_FUSED_OUT", "1")
    net2 = nnet.AmNnet(comps, priors)
    unfused = net2.Compute(x, mode=mode, prob_scale=0.9)
    net2.close()
    tol = 1e-5 if mode == nnet.POSTERIORS else 1e-3
    assert np.isfinite(got).all()
    assert np.abs(got - unfused).max() < tol
    if rows <= 300:
        want = oracle.nnet_compute(comps, priors, x, True, mode, 0.9)
        assert np.abs(got - want).max() < tol
        assert (got.argmax(1) == want.argmax(1)).mean() >= 0.9999
    if mode == nnet.POSTERIORS:
        assert np.allclose(got.sum(1), 1.0, atol=1e-4)


@pytest.mark.parametrize("cus", [None, "2"])
@pytest.mark.parametrize("pitch_pad", [0, 8, 4])
def test_row_stationary_kernels_on_awkward_row_counts(amlike_device, monkeypatch, cus, pitch_pad):
    """The 8-wave kernels of the p-norm recipe (hidden layers behind a Normalize, output layer +
    softmax) on a row count that is no multiple of 32 -- the last wave then takes the matrix's last
    32 rows -- with the job shapes of the whole chip (all short jobs) and of a two-CU chip (256-row
    jobs, a staggered short one, a job with inactive waves), and with an output pitch that is the
    row length (1000 floats: whole-line stores with the boundary moving from row to row), that
    length + 8, and + 4 (no multiple of 8: the older 4-wave kernel takes over).  Against the oracle
    on every frame: 1e-3 absolute, same best pdf."""
    import torch
    if cus:
        monkeypatch.setenv("AMLIKE_NNET_OUT_CUS", cus)
    comps, priors = synth.make_pnorm_nnet(input_dim=40, left=4, right=4, hidden=2000, pnorm_out=200,
                                          num_hidden=2, num_pdfs=1000, seed=77)
    net = nnet.AmNnet(comps, priors)
    rng = np.random.default_rng(78)
    lens = [77, 333, 1001]
    utts = [rng.normal(size=(n, 40)).astype(np.float32) for n in lens]
    ldo = net.OutputDim() + pitch_pad
    d_in = torch.from_numpy(np.concatenate(utts)).cuda()
    d_out = torch.full((sum(lens), ldo), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n = net.ComputeBatchDevice(d_in.data_ptr(), 40, lens, d_out.data_ptr(), ldo, pad_input=True,
                               mode=nnet.LOGLIKES, prob_scale=0.9)
    torch.cuda.synchronize()
    assert n == sum(lens)
    got = d_out.cpu().numpy()
    if pitch_pad:
        assert (got[:, net.OutputDim():] == 7.0).all()  # nothing written between the rows
    got = got[:, :net.OutputDim()]
    want = np.concatenate([oracle.nnet_compute(comps, priors, u, True, 2, 0.9) for u in utts])
    assert np.abs(got - want).max() < 1e-3
    assert (got.argmax(1) == want.argmax(1)).all()
    net.close()


def test_affine_behind_affine_is_folded_into_one_stage(amlike_device, monkeypatch):
    """FixedAffine + Affine (+ a third affine) become one stage when the engine is built, as
    nnet-am-copy --collapse would make them (nnet2/nnet-nnet.cc:624-663); the answers agree with the
    stages kept apart (AMLIKE_NNET_COLLAPSE=0) to rounding, and with the oracle, which runs the
    components one by one."""
    rng = np.random.default_rng(91)

    def aff(kind, n_in, n_out):
        return dict(type=kind, linear=(rng.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32),
                    bias=rng.normal(size=n_out).astype(np.float32))
    comps = [dict(type="SpliceComponent", input_dim=12, left=2, right=2, const_dim=0),
             aff("FixedAffineComponent", 60, 60), aff("AffineComponent", 60, 48),
             aff("AffineComponentPreconditioned", 48, 80),
             dict(type="PnormComponent", input_dim=80, output_dim=8, p=2.0),
             dict(type="NormalizeComponent", dim=8), aff("AffineComponent", 8, 30),
             dict(type="SoftmaxComponent", dim=30)]
    priors = np.full(30, 1 / 30, np.float32)
    x = rng.normal(size=(150, 12)).astype(np.float32)
    folded = nnet.AmNnet(comps, priors).Compute(x, mode=nnet.LOGLIKES)
    monkeypatch.setenv("AMLIKE_NNET_COLLAPSE", "0")
    apart = nnet.AmNnet(comps, priors).Compute(x, mode=nnet.LOGLIKES)
    np.testing.assert_allclose(folded, apart, rtol=0, atol=5e-5)
    want = oracle.nnet_compute(comps, priors, x, True, nnet.LOGLIKES, 1.0)
    assert np.abs(folded - want).max() < 1e-3
    assert (folded.argmax(1) == want.argmax(1)).all()
