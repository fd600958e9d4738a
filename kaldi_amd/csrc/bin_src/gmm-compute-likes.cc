// gmm-compute-likes on MI355X: same command line, inputs and outputs as the reference
// binary (/root/reference/src/gmmbin/gmm-compute-likes.cc:30-89); the frame x pdf loop
// (:71-77) runs on the GPU through libamlike.  One deviation: where the reference aborts on the
// first non-finite log-likelihood (status -1, nothing more written), this program reports the
// utterance, scores and writes all the others, and then returns -1.  Extra options: --device, --num-gpus (one worker
// thread and one copy of the model per GPU, outputs still in input order), and
// --frames-per-batch (utterances are scored together in batches of about that many frames,
// two batches in flight per GPU).
#include <chrono>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../host/gmm_model.h"
#include "../host/job_totals.h"
#include "../host/kaldi_io.h"
#include "../host/ordered_scorer.h"
#include "../host/parse_options.h"

int main(int argc, char *argv[]) {
  using namespace kaldi_amd;
  try {
    const char *usage =
        "Compute log-likelihoods from GMM-based model\n"
        "(outputs matrices of log-likelihoods indexed by (frame, pdf)\n"
        "Usage: gmm-compute-likes [options] model-in features-rspecifier likes-wspecifier\n";
    ParseOptions po(usage);
    int32 device = 0, num_gpus = 1, frames_per_batch = 32768;
    std::string device_list;
    float log_sum_exp_prune = -1.0f;
    po.Register("device", &device, "GPU to use (the first one with --num-gpus)");
    po.Register("num-gpus", &num_gpus, "Number of GPUs: devices --device .. --device + num-gpus - 1");
    po.Register("devices", &device_list,
                "Comma-separated device of each worker, instead of --device/--num-gpus (a device may "
                "repeat: several workers on one GPU)");
    po.Register("frames-per-batch", &frames_per_batch, "Frames scored per GPU launch (whole utterances)");
    po.Register("log-sum-exp-prune", &log_sum_exp_prune, "If >0, pruning value for the log-sum-exp");
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
        std::cerr << "WARNING (gmm-compute-likes) " << job_totals->Error() << "; this process will log its own counts only\n";
    }
    AmDiagGmm am_gmm;
    {
      bool binary;
      TransitionModel trans_model;  // not needed, but it is in the file
      Input ki(po.GetArg(1), &binary);
      trans_model.Read(ki.Stream(), binary);
      am_gmm.Read(ki.Stream(), binary);
    }
    MatrixWriter loglikes_writer(po.GetArg(3));
    SequentialMatrixReader feature_reader(po.GetArg(2));
    const auto t_pack = std::chrono::steady_clock::now();
    // every worker packs the model and uploads it to its device
    struct GmmDeleter {
      void operator()(amlike_gmm_t *g) const { amlike_gmm_destroy(g); }
    };
    std::vector<std::unique_ptr<amlike_gmm_t, GmmDeleter>> engines(devices.size());
    std::mutex engines_mu;
    size_t engines_made = 0;
    // The input, in order: batches of whole utterances (jobs), and utterances without frames.
    // (Declared before the scorer: its workers hold pointers to these jobs until it is gone.)
    struct Entry {
      std::unique_ptr<ScoreJob> job;
      std::string empty_key;
    };
    std::vector<Entry> window, previous;
    std::unique_ptr<ScoreJob> cur;
    OrderedScorer scorer(devices, [&]() {
      amlike_gmm_t *g = am_gmm.CreateEngine();
      {
        std::lock_guard<std::mutex> lock(engines_mu);
        engines[engines_made++].reset(g);
      }
      return std::unique_ptr<LikelihoodPipeline>(new LikelihoodPipeline(g, log_sum_exp_prune));
    });
    const auto t0 = std::chrono::steady_clock::now();
    int32 num_done = 0, num_err = 0;
    long long num_frames = 0;
    std::vector<long long> worker_frames(devices.size(), 0);
    auto publish = [&]() {
      std::vector<ScoreJob *> jobs;
      for (Entry &e : window)
        if (e.job) jobs.push_back(e.job.get());
      scorer.Publish(jobs);
    };
    auto commit = [&](std::vector<Entry> *entries) {
      for (Entry &e : *entries) {
        if (!e.job) {  // as the reference: a matrix of no rows
          loglikes_writer.Write(e.empty_key, nullptr, 0, am_gmm.NumPdfs(), am_gmm.NumPdfs());
          num_done++;
          continue;
        }
        ScoreJob &job = *e.job;
        scorer.WaitFor(&job);
        int32 r0 = 0;
        for (size_t u = 0; u < job.keys.size(); ++u) {
          if (job.errors[u].empty()) {
            loglikes_writer.Write(job.keys[u], job.out.RowData(r0), job.out_rows[u], job.out.NumCols(),
                                  job.out.Stride());
            num_done++;
            num_frames += job.out_rows[u];
            worker_frames[job.worker] += job.out_rows[u];
          } else {
            std::cerr << "WARNING (gmm-compute-likes) utterance " << job.keys[u] << " failed: " << job.errors[u]
                      << '\n';
            num_err++;
          }
          r0 += std::max(0, job.out_rows[u]);
        }
      }
      entries->clear();
    };
    // Window k is handed to the GPUs, then window k - 1 is written while they work on it.
    auto window_full = [&]() {
      size_t jobs = 0;
      for (const Entry &e : window) jobs += e.job ? 1 : 0;
      return jobs >= static_cast<size_t>(scorer.WindowSize());
    };
    auto turn = [&]() {
      publish();
      commit(&previous);
      previous.swap(window);
    };
    std::vector<Matrix> pending;
    auto flush = [&]() {
      if (!cur) return;
      int32 total = 0;
      for (const Matrix &m : pending) total += m.NumRows();
      const int32 dim = pending[0].NumCols();
      cur->feats.Resize(total, dim);
      int32 r0 = 0;
      for (const Matrix &m : pending) {
        for (int32 r = 0; r < m.NumRows(); ++r)
          std::copy(m.RowData(r), m.RowData(r) + dim, cur->feats.RowData(r0 + r));
        r0 += m.NumRows();
      }
      window.emplace_back();
      window.back().job = std::move(cur);
      pending.clear();
      if (window_full()) turn();
    };
    int32 in_batch = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      const Matrix &feats = feature_reader.Value();
      if (feats.NumRows() == 0) {
        flush();
        in_batch = 0;
        window.emplace_back();
        window.back().empty_key = feature_reader.Key();
        continue;
      }
      if (feats.NumCols() != am_gmm.Dim())  // fatal, as in the reference (gmm/diag-gmm.cc:517-521)
        KALDI_AMD_ERR("Dim mismatch: data dim = " << feats.NumCols() << " vs. model dim = " << am_gmm.Dim());
      if (!cur) cur.reset(new ScoreJob);
      cur->keys.push_back(feature_reader.Key());
      cur->rows.push_back(feats.NumRows());
      pending.push_back(feats);
      in_batch += feats.NumRows();
      if (in_batch >= frames_per_batch) {
        flush();
        in_batch = 0;
      }
    }
    flush();
    turn();
    commit(&previous);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "LOG (gmm-compute-likes) gmm-compute-likes: computed likelihoods for " << num_done
              << " utterances" << (num_err ? ", failed for " + std::to_string(num_err) : std::string()) << ".\n";
    std::cerr << "LOG (gmm-compute-likes) model packed and uploaded in "
              << std::chrono::duration<double>(t0 - t_pack).count() << " s\n";
    std::cerr << "LOG (gmm-compute-likes) " << num_frames << " frames in " << secs << " s ("
              << (secs > 0 ? num_frames / secs : 0.0) << " frames/s including file I/O)\n";
    if (devices.size() > 1)
      for (size_t w = 0; w < devices.size(); ++w)
        std::cerr << "LOG (gmm-compute-likes) worker " << w << " on GPU " << devices[w] << ": "
                  << worker_frames[w] << " frames\n";
    if (job_totals && job_totals->Joined()) {
      long long counts[3] = {num_done, num_err, num_frames};
      if (job_totals->AllReduce(counts, 3))
        std::cerr << "LOG (gmm-compute-likes) all " << totals.num_procs << " processes: computed likelihoods for "
                  << counts[0] << " utterances, failed for " << counts[1] << ", " << counts[2] << " frames\n";
      else  // the scoring run itself is complete: its exit status stands
        std::cerr << "WARNING (gmm-compute-likes) " << job_totals->Error() << "; the job's totals are not available\n";
    }
    // An utterance with a non-finite log-likelihood ends the reference's run with status -1
    // ("Invalid answer", gmm/diag-gmm.cc:523 -> gmm-compute-likes.cc:85-88).  Here the other
    // utterances are still scored and written, but the status says that keys are missing.
    return num_err > 0 ? -1 : 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
