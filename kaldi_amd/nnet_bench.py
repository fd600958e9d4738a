"""The nnet2 p-norm leg of bench.py: nnet-am-compute-style forward (pseudo
log-likelihoods) over one batch of synthetic utterances, on one GPU."""
import time

import numpy as np
import torch

from . import nnet, synth

FP32_MFMA_PEAK_TFLOPS = 157.3


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
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                     "note": "whole forward pass (GEMMs + row kernels) against the fp32 MFMA peak"},
    }
    net.close()
    return res
