#!/usr/bin/env python3
"""Headline benchmark: acoustic frames/sec of the GMM scoring path (and, as a
second figure on the same line, of the nnet2 p-norm forward) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch: the rank's shard of synthetic
utterances (256 per GPU, the benchmark definition's 2048 utterances over 8 GPUs; lengths
~U[200,1500], about 218 k frames, concatenated frame-wise, resident in HBM)
scored against the swbd-tri4-shaped GMM (D=40, 8600 pdfs, 200000 Gaussians),
dense [T x 8600] log-likelihoods left in HBM.  Utterances are sharded over
ranks (one process per GPU, the reference's JOB=1:$nj practice); there is no
data-path collective.  After the timed region the ranks all-reduce (RCCL)
{sum of best-pdf log-likelihood, frames}, the analogue of the summary the
reference binaries log.

With --gpus N > 1 and no torch.distributed environment, the script starts the
torch.distributed.run form itself as a child process and relays its line and exit code.

The line also carries
  host_to_host  the same scoring through host pointers (amlike_gmm_score_submit / _wait, two
                submissions in flight, page-locked buffers): features up and the dense
                [T x P] result down over PCIe -- the rate a caller holding host matrices sees;
  roofline      dominant kernel's algorithmic FLOP / its mean duration, measured with HIP
                events around that kernel in every timed step (on the launch stream), against
                the 16-bit matrix peak divided by the partial products one fp32-grade product
                takes (3 for the default fp16x2 operands); traffic = HBM bytes per launch from
                rocprofv3 PMC passes of the same workload;
  cpu_baseline  the real reference library (oracle/_ref, kind "reference") or the
                C restatement (kind "port") timed on a bounded sample, rank 0, N=1: one thread
                (value, cores = 1) and, under all_cores, one single-threaded process per host
                core of this job's share side by side; the nnet2 leg carries its own.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from kaldi_amd import _lib, gmm, synth
from kaldi_amd.sharding import reduce_summary, shard_utterances

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16)
# bf16x3 operands (the wide-range kernels): every fp32-grade product is 6 bf16 partial products,
# so the matrix pipe bounds them at peak / 6.
BF16X3_PRODUCTS = 6
# The default kernel uses fp16x2 operands (same matrix rate as bf16): 3 partial products.
F16X2_PRODUCTS = 3
# rocprofv3 --pmc passes over the scoring kernel on the default workload (frames on rank 0 at
# N = 1), KB per launch: profiles/r1_i_gmm_diff_pmc_summary.txt
# tools/micro/peaks.hip on the same kind of box (profiles/r1_k_peaks.txt): dense
# v_mfma_f32_32x32x16_f16 on every CU, four waves per SIMD, launches as long as this kernel's;
# with the same small operands in every instruction, and with pseudo-random operands (the clock
# the chip sustains under the matrix pipe depends on the switching activity of the data)
MEASURED_F16_MFMA_TFLOPS = 2419.0
MEASURED_F16_MFMA_TFLOPS_RANDOM = 1700.0
# (profiles/r3_zz_*_pmc_summary.txt: the current kernel; the counters are in KiB -- the nnet2 output
# kernel's WRITE_SIZE x 1024 equals its [T x 8600] x 4 bytes exactly)
PMC_FRAMES, PMC_FETCH_KB, PMC_WRITE_KB = 218541, 2450810.0, 9239440.0
PMC_SOURCE = "profiles/r3_zz_gmm_diff_pmc_summary.txt"


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_workers():
    """Worker processes for the all-cores figure: this process's share of the host, 16 at most."""
    return max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 16))


def _ref_rates(oracle, make_args, items_for, n, workers):
    """Times the real reference library (oracle/_ref) on n frames in one single-threaded process,
    then in `workers` such processes side by side, n frames each (JOB=1:$nj).  Returns
    (frames/s with one process, its seconds, frames/s with all workers, their slowest seconds)."""
    from oracle import kaldi_io as kio
    with tempfile.TemporaryDirectory() as d:
        arg_lists = []
        for w in range(workers):
            fpath = os.path.join(d, "feats%d.ark" % w)
            kio.write_ark(fpath, items_for(w))
            arg_lists.append(make_args(d, "ark:" + fpath))
        # untimed first run on a few frames: the BLAS library's start-up lands in its first call
        kio.write_ark(os.path.join(d, "warm.ark"), [(k, v[:8]) for k, v in items_for(0)[:1]])
        oracle.ref_driver(*make_args(d, "ark:" + os.path.join(d, "warm.ark")))
        secs1 = float(oracle.ref_driver(*arg_lists[0]).split()[1])
        if workers > 1:
            secs_all = max(float(o.split()[1]) for o in oracle.ref_driver_many(arg_lists))
        else:
            secs_all = secs1
    return n / secs1, secs1, workers * n / secs_all, secs_all


def cpu_baseline_gmm(model, gconsts, dim, seconds_target=30.0):
    """Times the reference CPU path on a bounded sample of the same workload."""
    import oracle
    from oracle import kaldi_io as kio
    est_fps = 86.0 * 200000.0 / float(model["pdf_offsets"][-1])  # 1-thread reference on C3: 86 frames/s
    n = int(max(32, min(200000, seconds_target * est_fps)))
    if oracle.have_ref():
        workers = _cpu_workers()
        off = model["pdf_offsets"]
        pdfs = [dict(weights=model["weights"][off[j]:off[j + 1]],
                     means_invvars=model["means_invvars"][off[j]:off[j + 1]],
                     inv_vars=model["inv_vars"][off[j]:off[j + 1]]) for j in range(len(off) - 1)]

        def make_args(d, feats_rspecifier):
            mdl = os.path.join(d, "am.mdl")
            if not os.path.exists(mdl):
                with kio.open_out(mdl, True) as f:
                    kio.write_am_diag_gmm(f, True, dim, pdfs, with_gconsts=False)
            return ("gmm-likes", mdl, feats_rspecifier, "ark:/dev/null")

        fps1, secs1, fps_all, secs_all = _ref_rates(
            oracle, make_args, lambda w: [("sample%d" % w, synth.make_feats(model, n, seed=778 + w))], n, workers)
        return {"value": fps1, "unit": "frames/s", "cores": 1, "kind": "reference", "cpu": _cpu_model(),
                "sample": "%d frames of the same synthetic workload, gmm-compute-likes loop, 1 thread, %.1f s"
                          % (n, secs1),
                "all_cores": {"value": fps_all, "unit": "frames/s", "cores": workers,
                              "sample": "%d single-threaded processes side by side, %d frames each, %.1f s"
                                        % (workers, n, secs_all)}}
    feats = synth.make_feats(model, n, seed=778)
    t0 = time.time()
    oracle.gmm_compute_likes(gconsts, model["means_invvars"], model["inv_vars"],
                             model["pdf_offsets"], feats)
    secs = time.time() - t0
    return {"value": n / secs, "unit": "frames/s", "cores": 1, "kind": "port", "cpu": _cpu_model(),
            "sample": "%d frames of the same synthetic workload, gmm-compute-likes loop, 1 thread, %.1f s" % (n, secs)}


def cpu_baseline_nnet(seconds_target=8.0):
    """The reference's nnet2 forward (NnetComputation + log, minus log-prior) on a bounded sample
    of the p-norm workload; None when the reference library was not built."""
    import oracle
    from oracle import kaldi_io as kio
    if not oracle.have_ref():
        return None
    comps, priors = synth.make_pnorm_nnet(seed=780)
    lengths = synth.utterance_lengths(256, 200, 1500, seed=779)
    want = seconds_target * 4000.0   # 1-thread reference: about 4-5 k frames/s
    lengths = lengths[:max(1, int(np.searchsorted(np.cumsum(lengths), want)))]
    n = int(lengths.sum())
    workers = _cpu_workers()

    def make_args(d, feats_rspecifier):
        mdl = os.path.join(d, "nnet.mdl")
        if not os.path.exists(mdl):
            with kio.open_out(mdl, True) as f:
                kio.write_am_nnet(f, True, comps, priors)
        return ("nnet-loglikes", mdl, feats_rspecifier, "ark:/dev/null", "1", "1.0")

    def items_for(w):
        rng = np.random.default_rng(1780 + w)
        return [("utt%d" % i, rng.normal(size=(int(t), 40)).astype(np.float32)) for i, t in enumerate(lengths)]

    fps1, secs1, fps_all, secs_all = _ref_rates(oracle, make_args, items_for, n, workers)
    return {"value": fps1, "unit": "frames/s", "cores": 1, "kind": "reference", "cpu": _cpu_model(),
            "sample": "%d frames in %d utterances, NnetComputation + log-likelihood conversion, 1 thread, %.1f s"
                      % (n, len(lengths), secs1),
            "all_cores": {"value": fps_all, "unit": "frames/s", "cores": workers,
                          "sample": "%d single-threaded processes side by side, %d frames each, %.1f s"
                                    % (workers, n, secs_all)}}


def host_to_host_rate(am, feats, P, chunk_frames=32768):
    """Frames/s through host pointers: amlike_gmm_score_submit / _wait with two submissions in
    flight over chunks of the rank's frames, features and results in page-locked host memory."""
    T, D = feats.shape[0], am.Dim()
    chunk = int(min(chunk_frames, T, (1 << 30) // P))
    fin = [_lib.PinnedArray(chunk, D) for _ in range(2)]
    fout = [_lib.PinnedArray(chunk, P) for _ in range(2)]
    starts = list(range(0, T, chunk))

    def run():
        tickets = [None, None]
        for i, t0 in enumerate(starts):
            k = i & 1
            if tickets[k] is not None:
                am.Wait(tickets[k])
            n = min(chunk, T - t0)
            fin[k].a[:n] = feats[t0:t0 + n, :D]
            tickets[k] = am.Submit(fin[k].a[:n], fout[k].a[:n])
        for t in tickets:
            if t is not None:
                am.Wait(t)

    run()  # buffers and the handle's device slots get their size
    t0 = time.perf_counter()
    run()
    secs = time.perf_counter() - t0
    for b in fin + fout:
        b.close()
    return {"value": T / secs, "unit": "frames/s", "seconds": secs, "frames": T,
            "bytes_over_pcie": 4.0 * T * (P + D),
            "how": "amlike_gmm_score_submit/_wait, two submissions in flight, chunks of %d frames, page-locked "
                   "host buffers; includes the copy of the features up and of the [T x %d] result down" % (chunk, P)}


def setup_workload(config, utts_per_gpu, world, rank):
    """One rank's share of the synthetic workload, host side only (NumPy, no GPU call): the model
    (the same on every rank), the rank's frames (weak scaling: utts_per_gpu utterances per rank,
    sharded by utterance, longest first) and their padded feature matrix.  Every rank of a node
    runs this at the same time before the first barrier; for the swbd tri4 shape at world = 8 it
    takes about 2 s and 0.4 GB per rank (tests/test_sharding_cpu.py holds it to that)."""
    D, P, G = synth.GMM_CONFIGS[config]
    model = synth.make_gmm(D, P, G, seed=777)
    lengths = synth.utterance_lengths(utts_per_gpu * world, 200, 1500, seed=779)
    owner = shard_utterances(lengths, world)
    mine = np.nonzero(owner == rank)[0]
    T = int(lengths[mine].sum())
    ldp = (D + 3) // 4 * 4
    feats = np.zeros((T, ldp), np.float32)
    feats[:, :D] = synth.make_feats(model, T, seed=1000 + rank)
    return model, T, ldp, feats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="swbd_tri4", choices=sorted(synth.GMM_CONFIGS))
    ap.add_argument("--utts-per-gpu", type=int, default=256)  # 2048 utterances over 8 GPUs
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nnet", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # python bench.py --gpus N: one process per GPU under torch.distributed.run, started as a
        # child before this process touches a GPU; its output and exit code are relayed.
        port = os.environ.get("MASTER_PORT", "29531")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to print a "
                 "line whose n_gpus would not be what was asked for" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "nccl"
    # Under a launcher (WORLD_SIZE set) the process group is always made, one rank included: the
    # RCCL initialisation, barrier and all-reduce of a real multi-GPU run are then the code that a
    # one-GPU box can run too.
    if world > 1 or "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        if ndev >= world:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            if dist.get_world_size() != args.gpus:
                sys.exit("bench.py: the RCCL process group has %d ranks, --gpus says %d"
                         % (dist.get_world_size(), args.gpus))
        else:
            # Rehearsal on a box with fewer GPUs than ranks: ranks share devices, which RCCL does
            # not allow, so the summary is reduced over gloo.  Not a scaling measurement.
            backend = "gloo"
            local_rank = local_rank % max(1, ndev)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0)
    _lib.check(_lib.lib().amlike_init(dev.index))

    D, P, G = synth.GMM_CONFIGS[args.config]
    model, T, ldp, feats = setup_workload(args.config, args.utts_per_gpu, world, rank)
    G = int(model["pdf_offsets"][-1])
    am = gmm.AmDiagGmm(model["weights"], model["means_invvars"], model["inv_vars"],
                       model["pdf_offsets"])
    d_feats = torch.from_numpy(feats).to(dev)
    d_out = torch.empty((T, P), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        am.LogLikelihoodsDevice(d_feats.data_ptr(), T, ldp, D, d_out.data_ptr(), P, stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    am.Poll(stream)
    # HIP events around the scoring kernel of every timed step, on the launch stream
    am.ProfileKernel(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    ms, timed_launches = am.KernelTime()
    am.ProfileKernel(False)
    am.Poll(stream)  # raises if any log-likelihood was NaN/Inf

    # Summary statistics, all-reduced like the reference binaries' totals.
    (tot_like, tot_frames), elapsed = reduce_summary(
        dist, [d_out.max(dim=1).values.double().sum().item(), float(T)], elapsed, device=dev if backend == "nccl" else None)
    # The job's collective work ends here: the process group is taken down before rank 0 goes on
    # to its single-GPU extras (host-to-host rate, nnet2 leg, CPU baseline: tens of seconds), so
    # that no rank sits in a barrier meanwhile and no RCCL watchdog is in play.
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        dist = None

    # Dominant kernel: mean of the event-bracketed durations collected over the timed region.
    if timed_launches == 0:  # more than 256 steps: fall back to a separate timing loop
        ms = am.TimeScoreDevice(d_feats.data_ptr(), T, ldp, D, d_out.data_ptr(), P, 10, stream)
    flop_per_frame = 2.0 * (2 * D + 1) * G
    achieved = flop_per_frame * T / (ms * 1e-3) / 1e12
    if "f16x2" in am.kernel_info():
        dtype = "f16x2-split operands (3 fp16 MFMA products per fp32-grade product), f32 accumulate"
        peak = BF16_MFMA_PEAK_TFLOPS / F16X2_PRODUCTS
        peak_basis = "fp16 MFMA dense peak 2500 TFLOP/s / 3 fp16 partial products per fp32-grade product"
    elif "bf16x3" in am.kernel_info():
        dtype = "bf16x3-split operands (6 bf16 MFMA products per fp32-grade product), f32 accumulate"
        peak = BF16_MFMA_PEAK_TFLOPS / BF16X3_PRODUCTS
        peak_basis = "bf16 MFMA dense peak 2500 TFLOP/s / 6 bf16 partial products per fp32-grade product"
    else:
        dtype = "f32 (fp32 MFMA)"
        peak = FP32_MFMA_PEAK_TFLOPS
        peak_basis = "fp32 MFMA dense peak"

    # HBM traffic of the dominant kernel per launch, from rocprofv3 PMC passes of this same
    # workload (FETCH_SIZE and WRITE_SIZE collected in separate runs; FETCH_SIZE doubled, the
    # gfx950 correction for 16-byte-per-lane loads).  Only valid for that kernel and shape.
    traffic = traffic_detail = None
    if args.config == "swbd_tri4" and "gmm_score_diff_kernel<5,2,8,2,f16x2>" in am.kernel_info() \
            and T == PMC_FRAMES:
        traffic = 2.0 * PMC_FETCH_KB * 1024 + PMC_WRITE_KB * 1024
        traffic_detail = {"fetch_bytes": 2.0 * PMC_FETCH_KB * 1024, "write_bytes": PMC_WRITE_KB * 1024,
                          "algorithmic_bytes": 4.0 * T * P + 4.0 * T * D,
                          "source": "constants from " + PMC_SOURCE + " (rocprofv3 --pmc passes of this workload and kernel; not re-measured by this run)"}

    if rank == 0:
        value = tot_frames * args.steps / elapsed
        line = {
            "metric": "acoustic frames/sec", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {
                "workload": "%s GMM gmm-compute-likes: D=%d, %d pdfs, %d Gaussians; %d utterances/GPU "
                            "(%d frames on rank 0) per step, dense [T x P] log-likelihoods kept in HBM"
                            % (args.config, D, P, G, args.utts_per_gpu, T),
                "parallelism": "utterance-sharded x%d, no data-path collective" % world +
                               ("" if backend == "nccl" else " (rehearsal: ranks share GPUs, summary over gloo)"),
                "kernel": am.kernel_info(),
                "avg_best_pdf_loglike_per_frame": tot_like / tot_frames,
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                "traffic_unit": "bytes per launch", "traffic_detail": traffic_detail,
                "kernel_ms": ms, "kernel_launches_timed": timed_launches,
                "flop_per_frame": flop_per_frame,
                "peak_basis": peak_basis,
                "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                "frac_of_measured_peak": achieved / (peak * MEASURED_F16_MFMA_TFLOPS / BF16_MFMA_PEAK_TFLOPS),
                "frac_of_measured_peak_random_operands":
                    achieved / (peak * MEASURED_F16_MFMA_TFLOPS_RANDOM / BF16_MFMA_PEAK_TFLOPS),
                "measured_peak_basis": "tools/micro/peaks.hip (profiles/r1_k_peaks.txt): dense fp16 MFMA sustains "
                                       "2419 TFLOP/s on constant operands and 1700 TFLOP/s on random ones; "
                                       "both divided like peak",
            },
        }
        if not args.no_host_to_host:
            line["host_to_host"] = host_to_host_rate(am, feats, P)
            line["host_to_host"]["n_gpus"] = 1  # rank 0's frames through rank 0's GPU
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline_gmm(model, am.gconsts, D)
            cb["gpu_over_cpu"] = value / cb["value"]
            if "all_cores" in cb:
                cb["all_cores"]["gpu_over_cpu"] = value / cb["all_cores"]["value"]
            line["cpu_baseline"] = cb
        else:
            line["cpu_baseline"] = None
        if not args.no_nnet:
            try:
                from kaldi_amd import nnet_bench
                line["config"]["nnet2_pnorm"] = nnet_bench.run(dev, stream, world,
                                                               utts_per_gpu=args.utts_per_gpu)
                if world == 1 and not args.no_cpu_baseline:
                    line["config"]["nnet2_pnorm"]["cpu_baseline"] = cpu_baseline_nnet()
            except ImportError:
                pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
