"""The command-line drop-ins (gmm-compute-likes, nnet-am-compute) and the Decodable
shims, run as separate processes on the GPU and checked against the oracle on the
same files.  Tolerance: 1e-3 absolute per (frame, pdf) log-likelihood."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from oracle import kaldi_io as kio
from kaldi_amd import synth

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi_amd", "csrc", "bin")
TOL = 1e-3


def run(name, *args, ok=True):
    p = subprocess.run([os.path.join(BIN, name), *args], capture_output=True, timeout=300)
    if ok:
        assert p.returncode == 0, p.stderr.decode()
    return p


def pdf_list(model):
    off = model["pdf_offsets"]
    return [dict(weights=model["weights"][a:b], means_invvars=model["means_invvars"][a:b],
                 inv_vars=model["inv_vars"][a:b]) for a, b in zip(off[:-1], off[1:])]


def write_gmm_mdl(path, model, binary=True):
    P = len(model["pdf_offsets"]) - 1
    with kio.open_out(path, binary) as f:
        topo, triples = kio.make_transition_model(P)
        tid2pdf = kio.write_transition_model(f, binary, topo, triples)
        kio.write_am_diag_gmm(f, binary, model["dim"], pdf_list(model), with_gconsts=False)
    return tid2pdf


def oracle_gmm(model, feats, prune=-1.0):
    gc = np.concatenate([oracle.compute_gconsts(p["weights"], p["means_invvars"], p["inv_vars"])[0]
                         for p in pdf_list(model)])
    out, st = oracle.gmm_compute_likes(gc, model["means_invvars"], model["inv_vars"],
                                       model["pdf_offsets"], feats, prune)
    assert st == 0
    return out


@pytest.fixture(scope="module")
def gmm_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("gmm")
    model = synth.make_gmm(39, 120, 900, seed=21)
    tid2pdf = write_gmm_mdl(str(d / "final.mdl"), model)
    lens = [1, 37, 412, 64, 1000]
    feats = [synth.make_feats(model, n, seed=30 + i) for i, n in enumerate(lens)]
    items = [("utt%02d" % i, f) for i, f in enumerate(feats)]
    kio.write_ark(str(d / "feats.ark"), items)
    return d, model, items, tid2pdf


@pytest.mark.parametrize("fpb", ["32768", "100", "1"])
def test_gmm_compute_likes_binary(gmm_files, fpb):
    d, model, items, _ = gmm_files
    out = str(d / ("likes_%s.ark" % fpb))
    p = run("gmm-compute-likes", "--frames-per-batch=" + fpb, str(d / "final.mdl"),
            "ark:" + str(d / "feats.ark"), "ark:" + out)
    assert b"computed likelihoods for 5 utterances" in p.stderr
    got = kio.read_ark(out)
    assert [k for k, _ in got] == [k for k, _ in items]
    for (_, f), (_, l) in zip(items, got):
        want = oracle_gmm(model, f)
        assert l.shape == want.shape
        assert np.abs(l - want).max() <= TOL
        assert (l.argmax(1) == want.argmax(1)).mean() >= 0.9999


def test_gmm_compute_likes_yesno_shape(tmp_path):
    """The yesno monophone configuration of the benchmark definition (D=39, 11 pdfs, 400 Gaussians,
    30 utterances of 500 frames) file to file, against the oracle."""
    D, P, G = synth.GMM_CONFIGS["yesno_mono"]
    model = synth.make_gmm(D, P, G, seed=777)
    write_gmm_mdl(str(tmp_path / "final.mdl"), model)
    items = [("yesno%02d" % i, synth.make_feats(model, 500, seed=800 + i)) for i in range(30)]
    kio.write_ark(str(tmp_path / "feats.ark"), items)
    out = str(tmp_path / "likes.ark")
    p = run("gmm-compute-likes", str(tmp_path / "final.mdl"), "ark:" + str(tmp_path / "feats.ark"), "ark:" + out)
    assert b"computed likelihoods for 30 utterances" in p.stderr
    got = kio.read_ark(out)
    assert [k for k, _ in got] == [k for k, _ in items]
    want = oracle_gmm(model, np.concatenate([f for _, f in items]))
    have = np.concatenate([l for _, l in got])
    assert have.shape == want.shape == (15000, P)
    assert np.abs(have - want).max() <= TOL
    assert (have.argmax(1) == want.argmax(1)).mean() >= 0.9999


@pytest.fixture(scope="module")
def many_utts(tmp_path_factory):
    """40 utterances of mixed length, one of them without frames."""
    d = tmp_path_factory.mktemp("many")
    model = synth.make_gmm(39, 120, 900, seed=21)
    write_gmm_mdl(str(d / "final.mdl"), model)
    rng = np.random.default_rng(7)
    lens = [int(n) for n in rng.integers(20, 700, size=40)]
    lens[11] = 0
    items = [("spk%d-utt%03d" % (i % 3, i), synth.make_feats(model, n, seed=500 + i) if n else
              np.zeros((0, 39), np.float32)) for i, n in enumerate(lens)]
    kio.write_ark(str(d / "feats.ark"), items)
    return d, model, items


def test_gmm_compute_likes_several_workers_same_archive(many_utts):
    """One worker thread and handle per device (--num-gpus); several workers sharing the only GPU of
    the test box (--devices=0,0,0) must write the archive a single worker writes, byte for byte:
    the jobs are cut the same way, scored by whoever is free, the longest of a window first, and
    committed in input order (the contract of the reference's TaskSequencer,
    thread/kaldi-task-sequence.h:78-129).  An utterance without frames comes out as a matrix of
    no rows, in its place."""
    d, model, items = many_utts
    outs = []
    for tag, opts in [("one", []), ("three", ["--devices=0,0,0"]), ("two", ["--devices=0,0", "--frames-per-batch=1"])]:
        out = str(d / ("likes_%s.ark" % tag))
        fpb = [] if any(o.startswith("--frames-per-batch") for o in opts) else ["--frames-per-batch=900"]
        p = run("gmm-compute-likes", *opts, *fpb, str(d / "final.mdl"), "ark:" + str(d / "feats.ark"), "ark:" + out)
        assert b"computed likelihoods for 40 utterances" in p.stderr
        outs.append(open(out, "rb").read())
        if tag == "three":
            assert p.stderr.count(b"worker ") == 3
    assert outs[0] == outs[1] == outs[2]
    got = kio.read_ark(str(d / "likes_three.ark"))
    assert [k for k, _ in got] == [k for k, _ in items]
    assert got[11][1].shape == (0, 120)
    for (_, f), (_, l) in zip(items, got):
        if f.shape[0]:
            assert np.abs(l - oracle_gmm(model, f)).max() <= TOL


def test_gmm_compute_likes_a_bad_utterance_costs_only_itself(many_utts, tmp_path):
    """Non-finite features in one utterance of a batch: its batch-mates are scored again on their
    own and written, the bad one is reported and counted, the run goes on (the reference's drivers
    count an utterance that fails and continue, gmmbin/gmm-latgen-faster.cc:118-122), and the exit
    status is non-zero, as the reference's is."""
    d, model, items = many_utts
    bad = [(k, f.copy()) for k, f in items[:12] if f.shape[0]]
    bad[4][1][3, 5] = np.nan
    kio.write_ark(str(tmp_path / "feats.ark"), bad)
    out = str(tmp_path / "likes.ark")
    p = run("gmm-compute-likes", "--devices=0,0", "--frames-per-batch=1500", str(d / "final.mdl"),
            "ark:" + str(tmp_path / "feats.ark"), "ark:" + out, ok=False)
    # the exit status tells the calling script that keys are missing (the reference's run ends
    # with -1 at the first such utterance, gmmbin/gmm-compute-likes.cc:85-88)
    assert p.returncode != 0
    warning = [l for l in p.stderr.split(b"\n") if b"utterance " + bad[4][0].encode() + b" failed:" in l]
    assert len(warning) == 1 and b"Invalid answer" in warning[0], p.stderr
    assert b"computed likelihoods for %d utterances, failed for 1" % (len(bad) - 1) in p.stderr
    got = kio.read_ark(out)
    assert [k for k, _ in got] == [k for i, (k, _) in enumerate(bad) if i != 4]
    for k, l in got:
        f = dict(bad)[k]
        assert np.abs(l - oracle_gmm(model, f)).max() <= TOL


def test_job_totals_go_through_rccl(many_utts, tmp_path):
    """--num-procs / --proc-id / --rendezvous: the end-of-run counts are all-reduced over RCCL from
    the C++ host code (process 0 publishes RCCL's id through the rendezvous file and removes it).
    A one-GPU box allows one process per job (RCCL refuses two ranks on a device): the job of one
    process runs the same calls -- id, communicator, all-reduce on the device, teardown."""
    d, _, items = many_utts
    rdv = str(tmp_path / "rccl_id")
    p = run("gmm-compute-likes", "--num-procs=1", "--proc-id=0", "--rendezvous=" + rdv, "--frames-per-batch=900",
            str(d / "final.mdl"), "ark:" + str(d / "feats.ark"), "ark:" + str(tmp_path / "likes.ark"))
    frames = sum(f.shape[0] for _, f in items)
    assert b"all 1 processes: computed likelihoods for 40 utterances, failed for 0, %d frames" % frames in p.stderr
    assert not os.path.exists(rdv)
    run("gmm-compute-likes", "--frames-per-batch=900", str(d / "final.mdl"), "ark:" + str(d / "feats.ark"),
        "ark:" + str(tmp_path / "plain.ark"))
    assert open(str(tmp_path / "likes.ark"), "rb").read() == open(str(tmp_path / "plain.ark"), "rb").read()
    # bad arguments are refused with a message, before anything is read
    q = run("gmm-compute-likes", "--num-procs=2", "--proc-id=2", "--rendezvous=" + rdv, str(d / "final.mdl"),
            "ark:" + str(d / "feats.ark"), "ark:" + str(tmp_path / "x.ark"), ok=False)
    assert q.returncode != 0 and b"--proc-id must lie in" in q.stderr
    q = run("gmm-compute-likes", "--num-procs=2", "--proc-id=1", str(d / "final.mdl"),
            "ark:" + str(d / "feats.ark"), "ark:" + str(tmp_path / "x.ark"), ok=False)
    assert q.returncode != 0 and b"--rendezvous is needed" in q.stderr
    # a process whose job never forms (process 0 does not show up) warns, scores its slice, and
    # keeps the exit status of its scoring run
    q = run("gmm-compute-likes", "--num-procs=2", "--proc-id=1", "--rendezvous=" + rdv + ".never",
            "--rendezvous-timeout=1", "--frames-per-batch=900", str(d / "final.mdl"),
            "ark:" + str(d / "feats.ark"), "ark:" + str(tmp_path / "alone.ark"))
    assert b"WARNING" in q.stderr and b"did not publish" in q.stderr and b"all 2 processes" not in q.stderr
    assert open(str(tmp_path / "alone.ark"), "rb").read() == open(str(tmp_path / "plain.ark"), "rb").read()


def test_num_gpus_beyond_the_visible_devices_is_refused(many_utts, tmp_path):
    d, _, _ = many_utts
    p = run("gmm-compute-likes", "--num-gpus=64", str(d / "final.mdl"), "ark:" + str(d / "feats.ark"),
            "ark:" + str(tmp_path / "x.ark"), ok=False)
    assert p.returncode != 0 and b"visible" in p.stderr


def test_gmm_compute_likes_text_model_and_text_output(gmm_files, tmp_path):
    d, model, items, _ = gmm_files
    small = synth.make_gmm(13, 9, 40, seed=22)
    write_gmm_mdl(str(tmp_path / "t.mdl"), small, binary=False)
    feats = synth.make_feats(small, 50, seed=5)
    kio.write_ark(str(tmp_path / "f.ark"), [("a", feats)], binary=False)
    out = str(tmp_path / "o.ark")
    run("gmm-compute-likes", str(tmp_path / "t.mdl"), "ark:" + str(tmp_path / "f.ark"), "ark,t:" + out)
    (key, l), = kio.read_ark(out)
    # the text model carries ~7 significant digits, so compare against what was read back
    r = kio.Reader(open(str(tmp_path / "t.mdl"), "rb").read())
    txt = open(str(tmp_path / "t.mdl"), "rb").read()
    r = kio.Reader(txt, txt.index(b"<DIMENSION>"))
    dim, pdfs = r.am_diag_gmm(False)
    back = dict(small)
    back["weights"] = np.concatenate([p["weights"] for p in pdfs])
    back["means_invvars"] = np.concatenate([p["means_invvars"] for p in pdfs])
    back["inv_vars"] = np.concatenate([p["inv_vars"] for p in pdfs])
    feats_back = kio.read_ark(str(tmp_path / "f.ark"))[0][1]
    want = oracle_gmm(back, feats_back)
    assert key == "a" and np.abs(l - want).max() <= TOL


def test_gmm_compute_likes_dim_mismatch_fails(gmm_files, tmp_path):
    d, model, _, _ = gmm_files
    kio.write_ark(str(tmp_path / "f.ark"), [("a", np.zeros((3, 40), np.float32))])
    p = run("gmm-compute-likes", str(d / "final.mdl"), "ark:" + str(tmp_path / "f.ark"),
            "ark:" + str(tmp_path / "o.ark"), ok=False)
    assert p.returncode != 0 and b"mismatch" in p.stderr.lower()


def test_decodable_gmm_scaled_and_unmapped(gmm_files):
    d, model, items, tid2pdf = gmm_files
    out = str(d / "dec.ark")
    run("amlike-decodable", "--acoustic-scale=0.0833", "gmm-scaled", str(d / "final.mdl"),
        "ark:" + str(d / "feats.ark"), "ark:" + out)
    got = kio.read_ark(out)
    for (_, f), (_, l) in zip(items, got):
        want = np.float32(0.0833) * oracle_gmm(model, f)[:, tid2pdf[1:]]
        assert l.shape == want.shape and np.abs(l - want).max() <= TOL
    run("amlike-decodable", "gmm-unmapped", str(d / "final.mdl"),
        "ark:" + str(d / "feats.ark"), "ark:" + out)
    got = kio.read_ark(out)
    for (_, f), (_, l) in zip(items, got):
        want = oracle_gmm(model, f)
        assert l.shape == want.shape and np.abs(l - want).max() <= TOL


@pytest.fixture(scope="module")
def nnet_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("nnet")
    comps, priors = synth.make_pnorm_nnet(input_dim=40, left=4, right=4, hidden=200, pnorm_out=40,
                                          num_hidden=2, num_pdfs=150, mix_up=170, seed=41)
    with kio.open_out(str(d / "final.mdl"), True) as f:
        topo, triples = kio.make_transition_model(150)
        tid2pdf = kio.write_transition_model(f, True, topo, triples)
        kio.write_am_nnet(f, True, comps, priors)
    rng = np.random.default_rng(42)
    items = [("u%d" % i, rng.standard_normal((n, 40)).astype(np.float32))
             for i, n in enumerate([5, 9, 300, 77])]
    kio.write_ark(str(d / "feats.ark"), items)
    return d, comps, priors, items, tid2pdf


@pytest.mark.parametrize("apply_log", [False, True])
@pytest.mark.parametrize("pad", [True, False])
def test_nnet_am_compute_binary(nnet_files, apply_log, pad):
    d, comps, priors, items, _ = nnet_files
    out = str(d / ("o_%d_%d.ark" % (apply_log, pad)))
    run("nnet-am-compute", "--apply-log=%s" % str(apply_log).lower(), "--pad-input=%s" % str(pad).lower(),
        str(d / "final.mdl"), "ark:" + str(d / "feats.ark"), "ark:" + out)
    got = dict(kio.read_ark(out))
    for key, f in items:
        if not pad and f.shape[0] <= 8:
            assert key not in got  # skipped: output would be empty
            continue
        want = oracle.nnet_compute(comps, priors, f, pad_input=pad, mode=1 if apply_log else 0)
        l = got[key]
        assert l.shape == want.shape
        if apply_log:
            assert np.abs(l - want).max() <= TOL
        else:
            assert np.abs(l - want).max() <= 1e-5
            assert (l.argmax(1) == want.argmax(1)).mean() >= 0.9999


@pytest.mark.parametrize("tag", ["nnet_c", "nnet_d", "nnet_e"])
@pytest.mark.parametrize("binary", [True, False])
def test_nnet_am_compute_reads_every_forward_component(tmp_path, tag, binary):
    """The model reader of the binaries on the networks that hold Dct, FixedScale, FixedBias,
    SoftHinge, Permute, FixedLinear, Maxout, Scale, Power and BlockAffine components, in the
    binary and the text form, against the posteriors the reference computed for them."""
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = open(os.path.join(golden, tag + ".mdl"), "rb").read()
    r = kio.Reader(data)
    comps, priors = r.am_nnet(r.read_binary_marker())
    with kio.open_out(str(tmp_path / "final.mdl"), binary) as f:
        topo, triples = kio.make_transition_model(len(priors))
        kio.write_transition_model(f, binary, topo, triples)
        kio.write_am_nnet(f, binary, comps, priors)
    out = str(tmp_path / "post.ark")
    run("nnet-am-compute", "--apply-log=false", str(tmp_path / "final.mdl"),
        "ark:" + os.path.join(golden, tag + "_feats.ark"), "ark:" + out)
    got = dict(kio.read_ark(out))
    want = dict(kio.read_ark(os.path.join(golden, tag + "_post.ark")))
    assert set(got) == set(want)
    for key in want:
        assert got[key].shape == want[key].shape
        assert np.abs(got[key] - want[key]).max() <= 1e-5


def test_nnet_am_compute_several_workers_same_archive(nnet_files, tmp_path):
    """As for gmm-compute-likes: three workers on one GPU write the bytes one worker writes."""
    d, comps, priors, _, _ = nnet_files
    rng = np.random.default_rng(43)
    items = [("utt%02d" % i, rng.standard_normal((int(n), 40)).astype(np.float32))
             for i, n in enumerate(rng.integers(9, 400, size=25))]
    kio.write_ark(str(tmp_path / "feats.ark"), items)
    outs = []
    for tag, opts in [("one", []), ("three", ["--devices=0,0,0"])]:
        out = str(tmp_path / ("o_%s.ark" % tag))
        p = run("nnet-am-compute", "--divide-by-priors=true", "--prob-scale=0.8", *opts, str(d / "final.mdl"),
                "ark:" + str(tmp_path / "feats.ark"), "ark:" + out)
        assert b"Processed 25 feature files" in p.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    got = kio.read_ark(str(tmp_path / "o_three.ark"))
    assert [k for k, _ in got] == [k for k, _ in items]
    for (_, f), (_, l) in zip(items[:5], got[:5]):
        want = oracle.nnet_compute(comps, priors, f, pad_input=True, mode=2, prob_scale=0.8)
        assert l.shape == want.shape and np.abs(l - want).max() <= TOL


def test_decodable_am_nnet(nnet_files):
    d, comps, priors, items, tid2pdf = nnet_files
    out = str(d / "dec.ark")
    run("amlike-decodable", "--acoustic-scale=0.1", "nnet", str(d / "final.mdl"),
        "ark:" + str(d / "feats.ark"), "ark:" + out)
    got = kio.read_ark(out)
    for (_, f), (_, l) in zip(items, got):
        want = oracle.nnet_compute(comps, priors, f, pad_input=True, mode=2, prob_scale=0.1)[:, tid2pdf[1:]]
        assert l.shape == want.shape and np.abs(l - want).max() <= TOL


def _touched(seed, T, N, permille):
    """The mock decoder's hashed trace: a [T x N] boolean array over (frame, tid - 1)."""
    f = np.arange(T, dtype=np.uint64)[:, None]
    t = np.arange(1, N + 1, dtype=np.uint64)[None, :]
    m = np.uint64(0xFFFFFFFF)
    h = (f * np.uint64(2654435761) + t * np.uint64(40503) + np.uint64(seed * 97)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & m
    h ^= h >> np.uint64(13)
    return (h % np.uint64(1000)) < np.uint64(permille)


def _check_replay(got, items, dense_of, touched_of):
    assert [k for k, _ in got] == [k for k, _ in items]
    for (_, f), (_, o) in zip(items, got):
        dense = dense_of(f).astype(np.float64)          # [T x N] LogLikelihood(frame, tid)
        mask = touched_of(dense.shape[0], dense.shape[1])
        want_sum = (dense * mask).sum(1)
        want_cnt = mask.sum(1)
        assert o.shape == (dense.shape[0], 2)
        np.testing.assert_array_equal(o[:, 1], want_cnt.astype(np.float32))
        # per look-up tolerance TOL (scaled), summed over the look-ups of the frame
        assert (np.abs(o[:, 0] - want_sum) <= 0.1 * TOL * np.maximum(want_cnt, 1) + 1e-3).all()


@pytest.mark.parametrize("double_buffer", ["true", "false"])
def test_mock_decode_gmm_hashed_trace(gmm_files, double_buffer):
    d, model, items, tid2pdf = gmm_files
    out = str(d / ("mock_%s.ark" % double_buffer))
    p = run("amlike-mock-decode", "--double-buffer=" + double_buffer, "--touch=0.3", "--seed=7",
            "--acoustic-scale=0.1", "gmm", str(d / "final.mdl"), "ark:" + str(d / "feats.ark"), "ark:" + out)
    assert b"5 utterances" in p.stderr and ("double-buffer=" + double_buffer).encode() in p.stderr
    _check_replay(kio.read_ark(out), items,
                  lambda f: np.float32(0.1) * oracle_gmm(model, f)[:, tid2pdf[1:]],
                  lambda T, N: _touched(7, T, N, 300))


def test_mock_decode_both_orders_agree_exactly_on_a_recorded_trace(gmm_files):
    d, model, items, tid2pdf = gmm_files
    rng = np.random.default_rng(3)
    N = len(tid2pdf) - 1
    traces = []
    for key, f in items:
        n = 20 * f.shape[0]
        tr = np.stack([rng.integers(0, f.shape[0], n), rng.integers(1, N + 1, n)], 1).astype(np.float32)
        traces.append((key, tr))
    kio.write_ark(str(d / "trace.ark"), traces)
    outs = []
    for db in ("true", "false"):
        out = str(d / ("mock_tr_%s.ark" % db))
        run("amlike-mock-decode", "--double-buffer=" + db, "--trace-rspecifier=ark:" + str(d / "trace.ark"),
            "gmm", str(d / "final.mdl"), "ark:" + str(d / "feats.ark"), "ark:" + out)
        outs.append(kio.read_ark(out))
    for (ka, a), (kb, b), (_, f), (_, tr) in zip(outs[0], outs[1], items, traces):
        assert ka == kb
        np.testing.assert_array_equal(a, b)
        dense = (np.float32(0.1) * oracle_gmm(model, f)[:, tid2pdf[1:]]).astype(np.float64)
        want = np.zeros(f.shape[0])
        np.add.at(want, tr[:, 0].astype(int), dense[tr[:, 0].astype(int), tr[:, 1].astype(int) - 1])
        cnt = np.bincount(tr[:, 0].astype(int), minlength=f.shape[0])
        np.testing.assert_array_equal(a[:, 1], cnt.astype(np.float32))
        assert (np.abs(a[:, 0] - want) <= 0.1 * TOL * np.maximum(cnt, 1) + 1e-3).all()


@pytest.mark.parametrize("double_buffer", ["true", "false"])
def test_mock_decode_nnet(nnet_files, double_buffer):
    d, comps, priors, items, tid2pdf = nnet_files
    out = str(d / ("mock_%s.ark" % double_buffer))
    run("amlike-mock-decode", "--double-buffer=" + double_buffer, "--touch=0.2", "--seed=2",
        "--acoustic-scale=0.1", "nnet", str(d / "final.mdl"), "ark:" + str(d / "feats.ark"), "ark:" + out)
    _check_replay(kio.read_ark(out), items,
                  lambda f: oracle.nnet_compute(comps, priors, f, pad_input=True, mode=2,
                                                prob_scale=0.1)[:, tid2pdf[1:]],
                  lambda T, N: _touched(2, T, N, 200))


@pytest.mark.parametrize("pad,chunk,batch", [(True, 50, 256), (True, 7, 16), (False, 33, 40), (True, 1000, 64)])
def test_decodable_nnet2_online_equals_whole_utterance_scoring(nnet_files, pad, chunk, batch):
    """DecodableNnet2Online (batches computed as frames arrive) gives the same log-likelihoods
    as the whole-utterance computation; the reference semantics of NumFramesReady / IsLastFrame
    decide how many frames come out."""
    d, comps, priors, items, tid2pdf = nnet_files
    out = str(d / ("online_%d_%d_%d.ark" % (pad, chunk, batch)))
    p = run("amlike-decodable", "--acoustic-scale=0.1", "--pad-input=%s" % str(pad).lower(),
            "--arrival-chunk=%d" % chunk, "--max-nnet-batch-size=%d" % batch, "nnet-online",
            str(d / "final.mdl"), "ark:" + str(d / "feats.ark"), "ark:" + out)
    got = dict(kio.read_ark(out))
    for key, f in items:
        l = got[key]
        if not pad and f.shape[0] <= 8:
            assert l.shape[0] == 0
            continue
        want = oracle.nnet_compute(comps, priors, f, pad_input=pad, mode=2, prob_scale=0.1)[:, tid2pdf[1:]]
        assert l.shape == want.shape
        assert np.abs(l - want).max() <= TOL
    if batch == 16:  # 300 frames cannot have been one batch
        assert b"u2: 300 frames in" in p.stderr
        n = int(p.stderr.split(b"u2: 300 frames in ")[1].split()[0])
        assert n >= 300 // 16
