#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory.

Inputs are synthetic (seeded numpy); expected outputs come from the REAL reference
library, compiled from /root/reference by oracle/build_ref.sh into oracle/_ref/ref_driver
and run here on the CPU.  Only the data travels: Kaldi-format model / feature files and
the reference's output matrices.

    python tests/golden/make_golden.py        (needs oracle/_ref/ref_driver)

Files written (all small):
  gmm_a.mdl  gmm_a_feats.ark  gmm_a_likes.ark  gmm_a_likes_prune5.ark  gmm_a_gconsts.vec
  gmm_a_pdf3_gauss.ark                         DiagGmm::LogLikelihoods(Matrix) of pdf 3
  gmm_b.mdl  gmm_b_feats.ark  gmm_b_likes.ark  dim 13, one very large pdf (split over waves on the GPU)
  nnet_a.mdl nnet_a_feats.ark nnet_a_post.ark nnet_a_log.ark nnet_a_loglikes.ark nnet_a_post_nopad.ark
  nnet_b.mdl nnet_b_feats.ark nnet_b_post.ark  sigmoid / tanh / relu / mix-up sum-group network
  nnet_c.mdl nnet_c_feats.ark nnet_c_post.ark nnet_c_post_nopad.ark   Dct, FixedScale, FixedBias, SoftHinge,
                                               Permute, FixedLinear, Maxout, Scale, Power, BlockAffine
  nnet_d.mdl nnet_d_feats.ark nnet_d_post.ark  Dct with <Reorder> (whole and truncated), the Power branches
  nnet_e.mdl nnet_e_feats.ark nnet_e_post.ark nnet_e_post_nopad.ark   Splice and SpliceMax (two context components)
               (python tests/golden/make_golden.py nnet_cd   regenerates these three only)
  cm_feats.ark  cm_feats_expanded.ark          CompressedMatrix archive and its expansion
  gsel.mdl gsel_feats.ark gsel_top5.txt        DiagGmm::GaussianSelection, 5 of 64 per frame
  xf_feats.ark xf_mats.ark                     13-dim features; "lda" 40x92, "lin" 40x91, "fmllr" 40x41
  xf_splice.ark xf_lda.ark xf_lin.ark xf_lda_fmllr.ark   SpliceFrames(3,3) and transform-feats once / twice
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import oracle  # noqa: E402
from oracle import kaldi_io as kio  # noqa: E402
from kaldi_amd import synth  # noqa: E402


def path(name):
    return os.path.join(HERE, name)


def pdf_list(model):
    off = model["pdf_offsets"]
    return [dict(weights=model["weights"][a:b], means_invvars=model["means_invvars"][a:b],
                 inv_vars=model["inv_vars"][a:b]) for a, b in zip(off[:-1], off[1:])]


def write_am(name, model):
    with kio.open_out(path(name), True) as f:
        kio.write_am_diag_gmm(f, True, model["dim"], pdf_list(model), with_gconsts=False)


def gmm_case(tag, model, feats_items, prune=None, batch_pdf=None):
    write_am(tag + ".mdl", model)
    kio.write_ark(path(tag + "_feats.ark"), feats_items)
    oracle.ref_driver("gmm-likes", path(tag + ".mdl"), "ark:" + path(tag + "_feats.ark"),
                      "ark:" + path(tag + "_likes.ark"))
    oracle.ref_driver("gmm-gconsts", path(tag + ".mdl"), path(tag + "_gconsts.vec"))
    if prune is not None:
        oracle.ref_driver("gmm-likes-prune", path(tag + ".mdl"), "ark:" + path(tag + "_feats.ark"),
                          "ark:" + path(tag + "_likes_prune%g.ark" % prune), str(prune))
    if batch_pdf is not None:
        oracle.ref_driver("gmm-batch", path(tag + ".mdl"), str(batch_pdf),
                          "ark:" + path(tag + "_feats.ark"),
                          "ark:" + path(tag + "_pdf%d_gauss.ark" % batch_pdf))


def nnet_case(tag, comps, priors, feats_items, full=True):
    with kio.open_out(path(tag + ".mdl"), True) as f:
        kio.write_am_nnet(f, True, comps, priors)
    kio.write_ark(path(tag + "_feats.ark"), feats_items)
    mdl, fe = path(tag + ".mdl"), "ark:" + path(tag + "_feats.ark")
    oracle.ref_driver("nnet-compute", mdl, fe, "ark:" + path(tag + "_post.ark"), "1", "0")
    if full:
        oracle.ref_driver("nnet-compute", mdl, fe, "ark:" + path(tag + "_log.ark"), "1", "1")
        oracle.ref_driver("nnet-loglikes", mdl, fe, "ark:" + path(tag + "_loglikes.ark"), "1", "0.1")
        oracle.ref_driver("nnet-compute", mdl, fe, "ark:" + path(tag + "_post_nopad.ark"), "0", "0")


def nnet_cd_cases():
    """The components outside the p-norm recipe, two small networks (own seeds, so adding to
    them leaves every other fixture as it is)."""
    rng = np.random.default_rng(4040)

    def aff(kind, i, o, **kw):
        return dict(type=kind, linear=(rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32),
                    bias=(0.5 * rng.standard_normal(o)).astype(np.float32), **kw)

    comps_c = [
        dict(type="SpliceComponent", input_dim=10, left=1, right=1, const_dim=0),
        dict(type="DctComponent", dim=30, dct_dim=10, reorder=False, dct_keep_dim=6),
        dict(type="FixedScaleComponent", scales=(0.5 + rng.random(18)).astype(np.float32)),
        dict(type="FixedBiasComponent", bias=rng.standard_normal(18).astype(np.float32)),
        aff("AffineComponent", 18, 24), dict(type="SoftHingeComponent", dim=24),
        dict(type="PermuteComponent", reorder=[int(x) for x in rng.permutation(24)]),
        dict(type="FixedLinearComponent", linear=(rng.standard_normal((24, 24)) / 5).astype(np.float32)),
        dict(type="MaxoutComponent", input_dim=24, output_dim=12),
        dict(type="ScaleComponent", dim=12, scale=0.7),
        dict(type="PowerComponent", dim=12, power=1.5),
        dict(type="BlockAffineComponent", num_blocks=4,
             linear=(rng.standard_normal((20, 3)) / 2).astype(np.float32),
             bias=(0.3 * rng.standard_normal(20)).astype(np.float32)),
        dict(type="SoftmaxComponent", dim=20),
    ]
    pri_c = rng.dirichlet(np.ones(20)).astype(np.float32)
    feats_c = [("c1", (4 * rng.standard_normal((37, 10))).astype(np.float32)),  # some SoftHinge inputs beyond 10
               ("c2", rng.standard_normal((3, 10)).astype(np.float32))]
    nnet_case("nnet_c", comps_c, pri_c, feats_c, full=False)
    oracle.ref_driver("nnet-compute", path("nnet_c.mdl"), "ark:" + path("nnet_c_feats.ark"),
                      "ark:" + path("nnet_c_post_nopad.ark"), "0", "0")
    comps_d = [
        dict(type="SpliceComponent", input_dim=8, left=2, right=2, const_dim=0),
        dict(type="DctComponent", dim=40, dct_dim=5, reorder=True),
        dict(type="DctComponent", dim=40, dct_dim=8, reorder=True, dct_keep_dim=3),
        aff("AffineComponent", 15, 16), dict(type="RectifiedLinearComponent", dim=16),
        dict(type="PowerComponent", dim=16, power=-0.5),  # exact zeros behind the ReLU stay zero
        aff("FixedAffineComponent", 16, 16),
        dict(type="PowerComponent", dim=16, power=2.0), dict(type="PowerComponent", dim=16, power=0.5),
        dict(type="PowerComponent", dim=16, power=1.0),
        dict(type="BlockAffineComponentPreconditioned", num_blocks=2,
             linear=(rng.standard_normal((8, 8)) / 3).astype(np.float32),
             bias=(0.3 * rng.standard_normal(8)).astype(np.float32)),
        dict(type="SoftmaxComponent", dim=8),
    ]
    pri_d = rng.dirichlet(np.ones(8)).astype(np.float32)
    nnet_case("nnet_d", comps_d, pri_d, [("d1", rng.standard_normal((29, 8)).astype(np.float32))], full=False)
    # nnet_e: SpliceMax between an affine stage and the rest, a Splice in front: two context
    # components, padded and unpadded
    comps_e = [
        dict(type="SpliceComponent", input_dim=9, left=1, right=2, const_dim=0),
        aff("AffineComponent", 36, 14), dict(type="TanhComponent", dim=14),
        dict(type="SpliceMaxComponent", dim=14, left=2, right=1),
        aff("AffineComponent", 14, 11), dict(type="SoftmaxComponent", dim=11),
    ]
    pri_e = rng.dirichlet(np.ones(11)).astype(np.float32)
    feats_e = [("e1", rng.standard_normal((31, 9)).astype(np.float32)),
               ("e2", rng.standard_normal((8, 9)).astype(np.float32))]
    nnet_case("nnet_e", comps_e, pri_e, feats_e, full=False)
    oracle.ref_driver("nnet-compute", path("nnet_e.mdl"), "ark:" + path("nnet_e_feats.ark"),
                      "ark:" + path("nnet_e_post_nopad.ark"), "0", "0")


def main():
    if not oracle.have_ref():
        sys.exit("oracle/_ref/ref_driver is not built (run oracle/build_ref.sh in a container "
                 "that has /root/reference)")
    if sys.argv[1:] == ["nnet_cd"]:
        return nnet_cd_cases()
    # gmm_a: an rm_tri1-like slice (D=39), 30 pdfs / 200 Gaussians, two utterances.
    a = synth.make_gmm(39, 30, 200, seed=4001)
    fa = [("utt1", synth.make_feats(a, 50, seed=4002)), ("utt2", synth.make_feats(a, 7, seed=4003))]
    gmm_case("gmm_a", a, fa, prune=5, batch_pdf=3)
    # gmm_b: D=13, one pdf of 700 Gaussians among small ones, one with a zero weight.
    rng = np.random.default_rng(4004)
    b = synth.make_gmm(13, 6, 60, seed=4005)
    big = synth.make_gmm(13, 1, 700, seed=4006)
    off = np.concatenate([b["pdf_offsets"], b["pdf_offsets"][-1:] + 700]).astype(np.int32)
    for k in ("weights", "means_invvars", "inv_vars"):
        b[k] = np.concatenate([b[k], big[k]])
    b["pdf_offsets"] = off
    b["weights"][2] = 0.0  # -inf gconst, counted but not an error (gmm/diag-gmm.cc:133-141)
    fb = [("u", (rng.standard_normal((40, 13)) * 1.5).astype(np.float32))]
    gmm_case("gmm_b", b, fb)
    # nnet_a: the p-norm topology, small.
    comps, priors = synth.make_pnorm_nnet(input_dim=13, left=2, right=3, hidden=60, pnorm_out=12,
                                          num_hidden=2, num_pdfs=25, mix_up=0, seed=4007)
    fn = [("n1", rng.standard_normal((40, 13)).astype(np.float32)),
          ("n2", rng.standard_normal((6, 13)).astype(np.float32))]
    nnet_case("nnet_a", comps, priors, fn)
    # nnet_b: other nonlinearities + mix-up SumGroup.
    def aff(kind, i, o, seed):
        r = np.random.default_rng(seed)
        return dict(type=kind, linear=(r.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32),
                    bias=(0.5 * r.standard_normal(o)).astype(np.float32))
    sizes = [1 + (i % 3) for i in range(20)]
    comps_b = [
        dict(type="SpliceComponent", input_dim=10, left=1, right=1, const_dim=0),
        aff("AffineComponent", 30, 24, 1), dict(type="SigmoidComponent", dim=24),
        aff("AffineComponentPreconditionedOnline", 24, 24, 2), dict(type="TanhComponent", dim=24),
        aff("AffineComponent", 24, 24, 3), dict(type="RectifiedLinearComponent", dim=24),
        aff("AffineComponent", 24, sum(sizes), 4), dict(type="SoftmaxComponent", dim=sum(sizes)),
        dict(type="SumGroupComponent", sizes=sizes),
    ]
    pri_b = np.random.default_rng(5).dirichlet(np.ones(20)).astype(np.float32)
    nnet_case("nnet_b", comps_b, pri_b, [("m", rng.standard_normal((33, 10)).astype(np.float32))], full=False)
    nnet_cd_cases()
    # Gaussian selection (DiagGmm::GaussianSelection) on a 64-component mixture.
    ubm = synth.make_gmm(20, 1, 64, seed=4020)
    write_am("gsel.mdl", ubm)
    kio.write_ark(path("gsel_feats.ark"), [("g", synth.make_feats(ubm, 60, seed=4021))])
    oracle.ref_driver("gmm-gselect", path("gsel.mdl"), "0", "5", "ark:" + path("gsel_feats.ark"),
                      path("gsel_top5.txt"))
    # The GMM systems' front end: SpliceFrames +-3, then transform-feats with an affine
    # (LDA+MLLT-like) matrix, a linear one, and a second per-speaker (fMLLR-like) affine one.
    xrng = np.random.default_rng(4030)
    xf_items = [("x%d" % i, xrng.standard_normal((n, 13)).astype(np.float32) * 3)
                for i, n in enumerate([1, 5, 60, 131])]
    kio.write_ark(path("xf_feats.ark"), xf_items)
    lda = (xrng.standard_normal((40, 92)) / np.sqrt(91)).astype(np.float32)
    lin = lda[:, :91].copy()
    fmllr = (np.eye(40, 41) + 0.1 * xrng.standard_normal((40, 41))).astype(np.float32)
    kio.write_ark(path("xf_mats.ark"), [("lda", lda), ("lin", lin), ("fmllr", fmllr)])
    for name, m in (("lda", lda), ("lin", lin), ("fmllr", fmllr)):
        with kio.open_out(path("tmp_%s.mat" % name), True) as f:
            kio.write_matrix(f, True, m)
    src = "ark:" + path("xf_feats.ark")
    oracle.ref_driver("splice-transform", src, "3", "3", "ark:" + path("xf_splice.ark"))
    oracle.ref_driver("splice-transform", src, "3", "3", "ark:" + path("xf_lda.ark"), path("tmp_lda.mat"))
    oracle.ref_driver("splice-transform", src, "3", "3", "ark:" + path("xf_lin.ark"), path("tmp_lin.mat"))
    oracle.ref_driver("splice-transform", src, "3", "3", "ark:" + path("xf_lda_fmllr.ark"),
                      path("tmp_lda.mat"), path("tmp_fmllr.mat"))
    for name in ("lda", "lin", "fmllr"):
        os.remove(path("tmp_%s.mat" % name))
    # Compressed feature matrices as the reference writes them: "CM" (per-column percentiles +
    # bytes) for the 50- and 40-row matrices, "CM2" (16-bit) for the 7-row one; and what the
    # reference's Matrix<float>::Read makes of them.
    cm_items = [("c1", synth.make_feats(a, 50, seed=4010)), ("c2", synth.make_feats(a, 7, seed=4011)),
                ("c3", (rng.standard_normal((40, 13)) * 20).astype(np.float32))]
    kio.write_ark(path("cm_src.ark"), cm_items)
    oracle.ref_driver("compress-feats", "ark:" + path("cm_src.ark"), "ark:" + path("cm_feats.ark"))
    oracle.ref_driver("copy-feats", "ark:" + path("cm_feats.ark"), "ark:" + path("cm_feats_expanded.ark"))
    os.remove(path("cm_src.ark"))
    total = sum(os.path.getsize(path(f)) for f in os.listdir(HERE) if not f.endswith(".py"))
    print("golden files: %d bytes" % total)


if __name__ == "__main__":
    main()
