"""Utterance sharding over ranks (one process per GPU) and the end-of-job reduction.

The reference parallelises this path by splitting the utterance list into $nj jobs
(egs/wsj/s5/steps/decode.sh:113, `JOB=1:$nj`), each an independent process; there is no
exchange between jobs while scoring.  The same holds here: every rank scores its own
utterances, and the only collective is the reduction of the summary counters the
binaries log at the end (frames, utterances, total best-pdf log-likelihood) plus the
max-over-ranks wall time."""
import numpy as np


def shard_utterances(lengths, world):
    """Longest-processing-time-first assignment of utterances to ranks.
    Deterministic: every rank computes the same owner[] from the same lengths."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(lengths), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += lengths[i]
    return owner


def reduce_summary(dist, sums, elapsed, device=None):
    """All-reduce per-rank counters: `sums` (list of floats) are added over ranks,
    `elapsed` is maximised.  dist is torch.distributed or None (single process).
    Returns (list of totals, max elapsed)."""
    if dist is None:
        return [float(x) for x in sums], float(elapsed)
    import torch
    s = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=device)
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in s.tolist()], float(t.item())
