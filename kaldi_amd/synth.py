"""Synthetic acoustic models and features of the benchmark configurations' shapes.

The reference ships no models (its tests draw random ones), so measurement and
parity use random-init models with the reference's own test distributions:

  * per-Gaussian parameters as unittest::InitRandDiagGmm
    (gmm/model-test-common.cc:91-108): w ~ exp(N(0,1)) normalised per pdf,
    mean_d ~ N(0,1)/(1+d), inv_var_d = exp(N(0,1)/(1+d)) + 0.01;
  * Gaussians per pdf by the power rule GetSplitTargets
    (gmm/model-common.cc:116-157, power 0.25 as egs/wsj/s5/steps/train_deltas.sh:18)
    on log-normal state occupancies;
  * features drawn from a random Gaussian of a random pdf per frame (the idea
    of DiagGmm::Generate, gmm/diag-gmm.cc:922-942), so that the log-sum-exp has a
    dominant-term structure like real data;
  * nnet2 p-norm network weights N(0, 1/sqrt(in)) and bias N(0, 0.5) as
    egs/wsj/s5/steps/nnet2/train_pnorm_fast.sh:28,219.

numpy's PCG64 with fixed seeds; never libc rand().
"""
import heapq

import numpy as np

# name -> (dim, num_pdfs, num_gauss): the GMM configurations of the benchmark.
GMM_CONFIGS = {
    "yesno_mono": (39, 11, 400),
    "rm_tri1": (39, 1500, 9000),
    "wsj_tri4b": (40, 3400, 40000),
    "swbd_tri4": (40, 8600, 200000),
}


def split_targets(occs, target_components, power=0.25, min_count=0.0):
    """GetSplitTargets (gmm/model-common.cc:116-157): repeatedly give one more
    Gaussian to the pdf with the largest occ^power / num_components."""
    P = len(occs)
    powed = np.asarray(occs, dtype=np.float64) ** power
    num = np.ones(P, dtype=np.int64)
    heap = [(-powed[i], i) for i in range(P)]
    heapq.heapify(heap)
    total = P
    while total < target_components and heap:
        _, i = heapq.heappop(heap)
        if (num[i] + 1) * min_count >= occs[i] and min_count > 0:
            continue
        num[i] += 1
        total += 1
        heapq.heappush(heap, (-powed[i] / num[i], i))
    return num


def make_gmm(dim, num_pdfs, num_gauss, seed=777, equal_sizes=False):
    """Returns dict(weights[G], means_invvars[G,D], inv_vars[G,D], pdf_offsets[P+1], means[G,D])."""
    rng = np.random.default_rng(seed)
    if equal_sizes:
        sizes = np.full(num_pdfs, num_gauss // num_pdfs, dtype=np.int64)
        sizes[: num_gauss - sizes.sum()] += 1
    else:
        occs = np.exp(rng.normal(6.0, 1.5, size=num_pdfs))
        sizes = split_targets(occs, num_gauss)
    offsets = np.zeros(num_pdfs + 1, dtype=np.int32)
    np.cumsum(sizes, out=offsets[1:])
    G = int(offsets[-1])
    scale = 1.0 / (1.0 + np.arange(dim, dtype=np.float64))
    w = np.exp(rng.normal(size=G))
    pdf_of = np.repeat(np.arange(num_pdfs), sizes)
    sums = np.zeros(num_pdfs)
    np.add.at(sums, pdf_of, w)
    w = (w / sums[pdf_of]).astype(np.float32)
    means = (rng.normal(size=(G, dim)) * scale).astype(np.float32)
    inv_vars = (np.exp(rng.normal(size=(G, dim)) * scale) + 1e-2).astype(np.float32)
    miv = (means * inv_vars).astype(np.float32)  # DiagGmm::SetInvVarsAndMeans
    return dict(weights=w, means_invvars=miv, inv_vars=inv_vars, pdf_offsets=offsets,
                means=means, dim=dim)


def make_feats(model, num_frames, seed=778):
    """[num_frames x dim] float32 features sampled from the model's own Gaussians."""
    rng = np.random.default_rng(seed)
    G = model["means_invvars"].shape[0]
    g = rng.integers(0, G, size=num_frames)
    var = 1.0 / model["inv_vars"][g].astype(np.float64)
    x = model["means"][g] + rng.normal(size=var.shape) * np.sqrt(var)
    return x.astype(np.float32)


def utterance_lengths(num_utts, lo, hi, seed=779):
    rng = np.random.default_rng(seed)
    return rng.integers(lo, hi + 1, size=num_utts).astype(np.int64)


def make_pnorm_nnet(input_dim=40, left=4, right=4, hidden=2000, pnorm_out=200, num_hidden=4,
                    num_pdfs=8600, mix_up=0, seed=780):
    """The p-norm nnet2 topology of egs/wsj/s5/steps/nnet2/train_pnorm_fast.sh:220-236:
    Splice -> FixedAffine(LDA) -> [Affine -> Pnorm -> Normalize] x num_hidden ->
    Affine -> Softmax [-> SumGroup when mix_up > num_pdfs].
    Returns (components, priors) in the dict form oracle/kaldi_io.write_am_nnet takes."""
    rng = np.random.default_rng(seed)
    spliced = input_dim * (left + 1 + right)

    def affine(kind, n_in, n_out, bias_std=0.5):
        return dict(type=kind,
                    linear=(rng.normal(size=(n_out, n_in)) / np.sqrt(n_in)).astype(np.float32),
                    bias=(rng.normal(size=n_out) * bias_std).astype(np.float32))

    comps = [dict(type="SpliceComponent", input_dim=input_dim, left=left, right=right, const_dim=0)]
    comps.append(affine("FixedAffineComponent", spliced, spliced, 0.1))
    n_in = spliced
    for _ in range(num_hidden):
        comps.append(affine("AffineComponentPreconditionedOnline", n_in, hidden))
        comps.append(dict(type="PnormComponent", input_dim=hidden, output_dim=pnorm_out, p=2.0))
        comps.append(dict(type="NormalizeComponent", dim=pnorm_out))
        n_in = pnorm_out
    out_dim = mix_up if mix_up > num_pdfs else num_pdfs
    comps.append(affine("AffineComponentPreconditionedOnline", n_in, out_dim))
    comps.append(dict(type="SoftmaxComponent", dim=out_dim))
    if mix_up > num_pdfs:
        sizes = np.ones(num_pdfs, dtype=np.int32)
        extra = rng.integers(0, num_pdfs, size=mix_up - num_pdfs)
        np.add.at(sizes, extra, 1)
        comps.append(dict(type="SumGroupComponent", sizes=sizes))
    priors = rng.dirichlet(np.full(num_pdfs, 2.0)).astype(np.float32)
    priors = np.maximum(priors, 1e-8).astype(np.float32)
    return comps, priors


def nnet_fullsize_utterances(kind="plain", input_dim=40, seed=781):
    """The utterances of the full-size nnet2 parity fixtures (tests/golden/make_golden_nnet_swbd.py).

    kind "plain": 10 240 frames in utterances of 1 ... 1500 frames (edge padding repeats the first
    and last frame of EACH utterance, nnet2/nnet-compute.cc:74-86), unit-variance input;
    "x100": the same distribution times 100 (first-layer operands far from 1, still inside fp16);
    "huge": values beyond 32768 in a few places (an fp16 operand cannot hold them);
    "peaky": 2 048 frames for make_peaky_nnet."""
    rng = np.random.default_rng(seed + {"plain": 0, "x100": 1, "huge": 2, "peaky": 3}[kind])
    if kind == "plain":
        lens = [1, 2, 5, 9, 33, 130, 300, 777, 1500, 1024, 911, 640, 1203, 1377, 999, 1329]
    elif kind == "peaky":
        lens = [3, 700, 1345]
    else:
        lens = [7, 250, 511]
    utts = [rng.normal(size=(n, input_dim)).astype(np.float32) for n in lens]
    if kind == "x100":
        utts = [u * np.float32(100.0) for u in utts]
    if kind == "huge":
        for u in utts:
            for _ in range(1 + len(u) // 100):
                u[rng.integers(len(u)), rng.integers(input_dim)] = np.float32(rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(4.6, 6.0))
    return utts


def make_peaky_nnet(scale=100.0, seed=780):
    """make_pnorm_nnet with an output layer whose answer turns on the input.  P-norm outputs are all
    positive, so a normalised hidden vector is mostly one fixed vector plus a small input-dependent
    part, and with N(0, 1/sqrt(in)) weights one pdf wins every frame.  Here the rows of the last
    affine layer are made orthogonal to the mean hidden vector (estimated on 512 random inputs),
    scaled up, and the bias removed: the best pdf changes from frame to frame.  Every seventh pdf
    gets a bias of -80, which puts its posterior under the 1e-20 floor of
    nnet2/nnet-component.cc:917 on every frame."""
    comps, priors = make_pnorm_nnet(seed=seed)
    rng = np.random.default_rng(seed + 1000)
    h = rng.normal(size=(512, comps[1]["linear"].shape[1]))
    for c in comps[1:-2]:
        if "linear" in c:
            h = h @ c["linear"].astype(np.float64).T + c["bias"].astype(np.float64)
        elif c["type"] == "PnormComponent":
            h = np.sqrt((h.reshape(len(h), c["output_dim"], -1) ** 2).sum(2))
        elif c["type"] == "NormalizeComponent":
            h = h / np.sqrt((h ** 2).mean(1, keepdims=True))
    mean = h.mean(0)
    mean /= np.linalg.norm(mean)
    last = comps[-2]
    w = last["linear"].astype(np.float64)
    last["linear"] = ((w - np.outer(w @ mean, mean)) * scale).astype(np.float32)
    last["bias"] = np.zeros_like(last["bias"])
    last["bias"][::7] = -80.0
    return comps, priors
