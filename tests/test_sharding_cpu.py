"""The N>1 path on CPU: utterance sharding and the end-of-job reduction over a
world_size-2 gloo group (the GPU job uses the same code over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from kaldi_amd import synth  # noqa: E402
from kaldi_amd.sharding import reduce_summary, shard_utterances  # noqa: E402


def test_shards_are_a_partition_and_balanced():
    lengths = synth.utterance_lengths(320, 200, 1500, seed=779)
    for world in (1, 2, 4, 8):
        owner = shard_utterances(lengths, world)
        assert owner.shape == lengths.shape and owner.min() >= 0 and owner.max() == world - 1
        loads = np.array([lengths[owner == r].sum() for r in range(world)])
        assert loads.sum() == lengths.sum()
        assert loads.max() - loads.min() <= lengths.max()  # LPT bound
        assert (loads.max() / loads.mean()) < 1.02


def test_sharding_edge_cases():
    assert shard_utterances(np.array([], dtype=np.int64), 4).shape == (0,)
    owner = shard_utterances(np.array([10, 10]), 4)  # fewer utterances than ranks
    assert sorted(owner.tolist()) == [0, 1]
    owner = shard_utterances(np.array([5, 1, 1, 1, 1, 1]), 2)
    assert owner[0] != owner[1] and np.array([5, 1, 1, 1, 1, 1])[owner == owner[0]].sum() == 5


def test_single_process_reduction_is_identity():
    sums, t = reduce_summary(None, [3.0, 4.0], 0.5)
    assert sums == [3.0, 4.0] and t == 0.5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = synth.utterance_lengths(24, 20, 60, seed=5)
        owner = shard_utterances(lengths, world)
        mine = np.nonzero(owner == rank)[0]
        # a stand-in for the scoring step: per-utterance "scores" that depend only on the
        # utterance, so the all-reduced total must equal the single-process total
        frames = float(lengths[mine].sum())
        like = float(sum(np.sin(np.arange(lengths[i]) + i).sum() for i in mine))
        elapsed = 0.25 * (rank + 1)
        (tot_like, tot_frames, n_utts), t = reduce_summary(dist, [like, frames, float(len(mine))], elapsed)
        q.put((rank, tot_like, tot_frames, n_utts, t))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_job_matches_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = synth.utterance_lengths(24, 20, 60, seed=5)
    want_like = float(sum(np.sin(np.arange(lengths[i]) + i).sum() for i in range(24)))
    for rank, tot_like, tot_frames, n_utts, t in results:
        assert tot_frames == float(lengths.sum()) and n_utts == 24.0
        assert abs(tot_like - want_like) < 1e-9 * max(1.0, abs(want_like)) + 1e-9
        assert t == 0.5  # max over ranks


def test_bench_setup_for_eight_ranks_is_light():
    """bench.py's host-side set-up as every rank of an 8-GPU run performs it, all at once, before the
    first barrier (the 200 000-Gaussian model and the rank's ~217 k frames, NumPy only): measured
    2 s and 0.4 GB per rank; held here to 30 s and 1.5 GB, in a process of its own so that the peak
    memory is this work's."""
    import subprocess
    import sys
    code = (
        "import json, resource, sys, time\n"
        "sys.argv = ['bench.py']\n"
        "import bench\n"
        "out = {}\n"
        "for rank in (0, 7):\n"
        "    t0 = time.time()\n"
        "    model, T, ldp, feats = bench.setup_workload('swbd_tri4', 256, 8, rank)\n"
        "    out[rank] = dict(secs=time.time() - t0, T=T, shape=list(feats.shape), G=int(model['pdf_offsets'][-1]))\n"
        "out['rss_mb'] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0\n"
        "print(json.dumps(out))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    out = json.loads(p.stdout.strip().splitlines()[-1])
    for rank in ("0", "7"):
        assert out[rank]["secs"] < 30.0
        assert out[rank]["G"] == 200000 and out[rank]["shape"] == [out[rank]["T"], 40]
        assert 200000 < out[rank]["T"] < 235000  # 256 utterances of ~U[200, 1500] frames
    assert out["rss_mb"] < 1536.0
