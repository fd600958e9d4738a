"""The nnet2 p-norm leg of bench.py: nnet-am-compute-style forward (pseudo
log-likelihoods) over one batch of synthetic utterances, on one GPU."""
import time

import numpy as np
import torch

from . import nnet, synth

# The layers run on fp16x2 operands (3 partial products per fp32-grade product) of the fp16
# matrix pipe, 2500 TFLOP/s dense.  The output layer and the softmax are one kernel that forms the
# [T x 8600] product twice and writes the result once (4 bytes per element).
F16X2_PEAK_TFLOPS = 2500.0 / 3.0
HBM_PEAK_TBS = 8.0
# HBM traffic of one forward pass of this workload (256 utterances, 218 541 frames), from rocprofv3
# --pmc passes (FETCH_SIZE, WRITE_SIZE in KB summed over the pass's kernels; FETCH_SIZE x2 as for the
# GMM kernel): constants, not re-measured by a bench run.
PMC_FRAMES, PMC_FETCH_KB, PMC_WRITE_KB = 218541, 1575653.0, 9095452.0
PMC_SOURCE = "profiles/r3_zz_nnet_pmc_summary.txt"


def run(dev, stream, world, utts_per_gpu=40, steps=10, warmup=2, rank=0):
    comps, priors = synth.make_pnorm_nnet(seed=780)
    net = nnet.AmNnet(comps, priors)
    lengths = synth.utterance_lengths(utts_per_gpu, 200, 1500, seed=779 + rank)
    T = int(lengths.sum())
    feats = np.random.default_rng(1780 + rank).normal(size=(T, 40)).astype(np.float32)
    d_in = torch.from_numpy(feats).to(dev)
    P = net.OutputDim()
    d_out = torch.empty((T, P), dtype=torch.float32, device=dev)

    def step():
        net.ComputeBatchDevice(d_in.data_ptr(), 40, lengths, d_out.data_ptr(), P, pad_input=True,
                               mode=nnet.LOGLIKES, stream=stream)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    flop_per_frame = 0.0
    for c in comps:
        if "linear" in c:
            flop_per_frame += 2.0 * c["linear"].shape[0] * c["linear"].shape[1]
    achieved = flop_per_frame * T / dt / 1e12
    res = {
        "workload": "swbd nnet2 p-norm nnet-am-compute: 40 -> splice +-4 -> LDA 360 -> 4 x (2000 -> pnorm 200 "
                    "-> normalize) -> 8600 softmax, log, minus log-prior; %d utterances, %d frames per step" % (len(lengths), T),
        "frames_per_sec_per_gpu": T / dt, "ms_per_step": dt * 1e3,
        "gemm_flop_per_frame": flop_per_frame,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": F16X2_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / F16X2_PEAK_TFLOPS,
                     "note": "whole forward pass (GEMMs + row kernels) against the fp16 matrix peak / 3 "
                             "partial products; the output layer's product is formed twice (counted once) so "
                             "that the [T x 8600] result is written once and never read back",
                     "hbm_floor_ms": 4.0 * T * P / (HBM_PEAK_TBS * 1e12) * 1e3,
                     "traffic": None},
    }
    if T == PMC_FRAMES:
        res["roofline"]["traffic"] = 2.0 * PMC_FETCH_KB * 1024 + PMC_WRITE_KB * 1024
        res["roofline"]["traffic_detail"] = {
            "fetch_bytes": 2.0 * PMC_FETCH_KB * 1024, "write_bytes": PMC_WRITE_KB * 1024,
            "algorithmic_bytes": 4.0 * T * (40 + P),
            "source": "constants from " + PMC_SOURCE + " (rocprofv3 --pmc passes of this workload, counters in KiB: the output kernel's "
                      "WRITE_SIZE x 1024 equals its [T x 8600] x 4 bytes exactly; not re-measured by this run)"}
    net.close()
    return res
