// nnet-am-compute on MI355X: same command line, inputs and outputs as the reference
// binary (/root/reference/src/nnet2bin/nnet-am-compute.cc:27-110); NnetComputation()
// (:91) and the optional floor + log (:93-96) run on the GPU through libamlike.  Extra
// options: --device, --num-gpus (one worker thread and one copy of the net per GPU, two
// utterances in flight on each, outputs still in input order).
#include <chrono>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../host/job_totals.h"
#include "../host/kaldi_io.h"
#include "../host/nnet_model.h"
#include "../host/ordered_scorer.h"
#include "../host/parse_options.h"

int main(int argc, char *argv[]) {
  using namespace kaldi_amd;
  using namespace kaldi_amd::nnet2;
  try {
    const char *usage =
        "Does the neural net computation for each file of input features, and\n"
        "outputs as a matrix the result.  Used mostly for debugging.\n"
        "Note: if you want it to apply a log (e.g. for log-likelihoods), use\n"
        "--apply-log=true\n"
        "\n"
        "Usage:  nnet-am-compute [options] <model-in> <feature-rspecifier> "
        "<feature-or-loglikes-wspecifier>\n";
    bool apply_log = false, pad_input = true, divide_by_priors = false;
    int32 device = 0, num_gpus = 1;
    std::string device_list;
    float prob_scale = 1.0f;
    ParseOptions po(usage);
    po.Register("apply-log", &apply_log, "Apply a log to the result of the computation before outputting.");
    po.Register("pad-input", &pad_input,
                "If true, duplicate the first and last frames of input features as required for "
                "temporal context, to prevent #frames of output being less than those of input.");
    po.Register("divide-by-priors", &divide_by_priors,
                "Output pseudo log-likelihoods log(p) - log(prior), scaled by --prob-scale "
                "(what DecodableAmNnet computes); implies --apply-log.");
    po.Register("prob-scale", &prob_scale, "Scale applied with --divide-by-priors");
    po.Register("device", &device, "GPU to use (the first one with --num-gpus)");
    po.Register("num-gpus", &num_gpus, "Number of GPUs: devices --device .. --device + num-gpus - 1");
    po.Register("devices", &device_list,
                "Comma-separated device of each worker, instead of --device/--num-gpus (a device may "
                "repeat: several workers on one GPU)");
    JobTotalsOptions totals;
    po.Register("num-procs", &totals.num_procs,
                "Processes of a split job (one per GPU, each with its own slice of the features): their "
                "end-of-run counts are summed with one RCCL all-reduce and logged by every process");
    po.Register("proc-id", &totals.proc_id, "This process among --num-procs, from 0");
    po.Register("rendezvous", &totals.rendezvous,
                "File through which process 0 hands RCCL's id to the others; a new path for every run");
    po.Register("rendezvous-timeout", &totals.rendezvous_timeout,
                "Seconds to wait at program start for process 0 to publish RCCL's id");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) {
      po.PrintUsage();
      return 1;
    }
    const std::vector<int32> devices = ParseDeviceList(device_list, num_gpus, device);
    CheckAmlike(amlike_init(devices[0]));
    // The processes of a split job join their communicator now, while all of them are starting up.
    std::unique_ptr<JobTotals> job_totals;
    if (totals.Enabled()) {
      if (totals.num_procs < 1 || totals.proc_id < 0 || totals.proc_id >= totals.num_procs)
        KALDI_AMD_ERR("--proc-id must lie in [0, --num-procs)");
      if (totals.rendezvous.empty()) KALDI_AMD_ERR("--rendezvous is needed with --num-procs");
      job_totals.reset(new JobTotals(totals, devices[0]));
      if (!job_totals->Joined())
        std::cerr << "WARNING (nnet-am-compute) " << job_totals->Error() << "; this process will log its own counts only\n";
    }
    TransitionModel trans_model;
    AmNnet am_nnet;
    {
      bool binary;
      Input ki(po.GetArg(1), &binary);
      trans_model.Read(ki.Stream(), binary);
      am_nnet.Read(ki.Stream(), binary);
    }
    const Nnet &nnet = am_nnet.GetNnet();
    const int32 context = nnet.LeftContext() + nnet.RightContext();
    long long num_done = 0, num_err = 0, num_frames = 0;
    SequentialMatrixReader feature_reader(po.GetArg(2));
    MatrixWriter writer(po.GetArg(3));
    const int mode = divide_by_priors ? AMLIKE_NNET_LOGLIKES : apply_log ? AMLIKE_NNET_LOG : AMLIKE_NNET_POSTERIORS;
    // every worker builds the weight images and uploads them to its device, before the clock starts
    struct NnetDeleter {
      void operator()(amlike_nnet_t *n) const { amlike_nnet_destroy(n); }
    };
    std::vector<std::unique_ptr<amlike_nnet_t, NnetDeleter>> engines(devices.size());
    std::mutex engines_mu;
    size_t engines_made = 0;
    // (declared before the scorer: its workers hold pointers to these jobs until it is gone)
    std::vector<std::unique_ptr<ScoreJob>> window, previous;
    OrderedScorer scorer(devices, [&]() {
      amlike_nnet_t *h = am_nnet.CreateEngine();
      {
        std::lock_guard<std::mutex> lock(engines_mu);
        engines[engines_made++].reset(h);
      }
      return std::unique_ptr<LikelihoodPipeline>(new LikelihoodPipeline(h, context, pad_input, mode, prob_scale));
    });
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<long long> worker_frames(devices.size(), 0);
    // One job per utterance.  Window k is handed to the GPUs, then window k - 1 is written
    // while they work on it.
    auto commit = [&](std::vector<std::unique_ptr<ScoreJob>> *jobs) {
      for (auto &j : *jobs) {
        scorer.WaitFor(j.get());
        if (j->errors[0].empty()) {
          writer.Write(j->keys[0], j->out.Data(), j->out.NumRows(), j->out.NumCols(), j->out.Stride());
          num_done++;
          worker_frames[j->worker] += j->rows[0];
        } else {
          std::cerr << "WARNING (nnet-am-compute) utterance " << j->keys[0] << " failed: " << j->errors[0] << '\n';
          num_err++;
        }
      }
      jobs->clear();
    };
    auto turn = [&]() {
      std::vector<ScoreJob *> jobs;
      for (auto &j : window) jobs.push_back(j.get());
      scorer.Publish(jobs);
      commit(&previous);
      previous.swap(window);
    };
    for (; !feature_reader.Done(); feature_reader.Next()) {
      const std::string utt = feature_reader.Key();
      const Matrix &feats = feature_reader.Value();
      int32 output_frames = feats.NumRows();
      if (!pad_input) output_frames -= context;
      if (output_frames <= 0) {
        std::cerr << "WARNING (nnet-am-compute) Skipping utterance " << utt
                  << " because output would be empty.\n";
        continue;
      }
      if (feats.NumCols() != nnet.InputDim())  // fatal, as in the reference (nnet2/nnet-compute.cc:45-48)
        KALDI_AMD_ERR("Dim mismatch: data dim = " << feats.NumCols() << " vs. model dim = " << nnet.InputDim());
      std::unique_ptr<ScoreJob> job(new ScoreJob);
      job->keys.push_back(utt);
      job->rows.push_back(feats.NumRows());
      job->feats = feats;
      num_frames += feats.NumRows();
      window.push_back(std::move(job));
      if (static_cast<int32>(window.size()) >= scorer.WindowSize()) turn();
    }
    turn();
    commit(&previous);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "LOG (nnet-am-compute) " << num_frames << " frames in " << secs << " s ("
              << (secs > 0 ? num_frames / secs : 0.0) << " frames/s including file I/O)\n";
    if (devices.size() > 1)
      for (size_t w = 0; w < devices.size(); ++w)
        std::cerr << "LOG (nnet-am-compute) worker " << w << " on GPU " << devices[w] << ": " << worker_frames[w]
                  << " frames\n";
    std::cerr << "LOG (nnet-am-compute) Processed " << num_done << " feature files, " << num_frames
              << " frames of input were processed." << (num_err ? " Failed: " + std::to_string(num_err) : std::string())
              << "\n";
    if (job_totals && job_totals->Joined()) {
      long long counts[3] = {num_done, num_err, num_frames};
      if (job_totals->AllReduce(counts, 3))
        std::cerr << "LOG (nnet-am-compute) all " << totals.num_procs << " processes: processed " << counts[0]
                  << " feature files, failed for " << counts[1] << ", " << counts[2] << " frames\n";
      else  // the scoring run itself is complete: its exit status stands
        std::cerr << "WARNING (nnet-am-compute) " << job_totals->Error() << "; the job's totals are not available\n";
    }
    return (num_done == 0 ? 1 : 0);
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
