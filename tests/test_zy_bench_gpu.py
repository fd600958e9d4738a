"""bench.py as the driver launches it for N > 1: one process per rank under
torch.distributed.run.  On a box with fewer GPUs than ranks the ranks share the device and the
end-of-job summary goes over gloo (RCCL refuses two ranks on one device); sharding, barriers,
the max-over-ranks timing and rank 0's JSON line are the same code as in the real runs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from kaldi_amd import synth
from kaldi_amd.sharding import shard_utterances

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_two_ranks_and_one_rank_agree(amlike_device):
    common = ["--steps", "2", "--warmup", "1", "--config", "rm_tri1", "--utts-per-gpu", "6",
              "--no-nnet", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = _line(one.stdout)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    j2 = _line(two.stdout)
    assert (j1["n_gpus"], j2["n_gpus"]) == (1, 2)
    assert j2["scaling"] == "weak" and j2["cpu_baseline"] is None
    for j in (j1, j2):
        assert j["metric"] == "acoustic frames/sec" and j["value"] > 0 and j["steps"] == 2
        assert 0 < j["roofline"]["frac"] < 1
        assert np.isfinite(j["config"]["avg_best_pdf_loglike_per_frame"])
    # weak scaling: 6 utterances per rank; the frames rank 0 reports are its shard of the 12
    lengths = synth.utterance_lengths(12, 200, 1500, seed=779)
    owner = shard_utterances(lengths, 2)
    assert "(%d frames on rank 0)" % int(lengths[owner == 0].sum()) in j2["config"]["workload"]
    assert abs(j2["value"] * j2["ms_per_step"] * 1e-3 - float(lengths.sum())) < 1.0


def test_one_rank_under_the_launcher_takes_the_rccl_branch(amlike_device):
    """The branch real multi-GPU runs take (RCCL process group on the rank's own device, barrier,
    device all-reduce of the summary) with the one rank a one-GPU box allows."""
    common = ["--steps", "2", "--warmup", "1", "--config", "rm_tri1", "--utts-per-gpu", "6",
              "--no-nnet", "--no-cpu-baseline", "--no-host-to-host"]
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=280, cwd=ROOT,
                         env=dict(os.environ, TORCH_DISTRIBUTED_DEBUG="INFO"))
    assert run.returncode == 0, run.stderr[-2000:]
    j = _line(run.stdout)
    assert j["n_gpus"] == 1 and "rehearsal" not in j["config"]["parallelism"]
    lengths = synth.utterance_lengths(6, 200, 1500, seed=779)
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - float(lengths.sum())) < 1.0
