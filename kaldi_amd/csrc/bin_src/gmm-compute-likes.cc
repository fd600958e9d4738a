// gmm-compute-likes on MI355X: same command line, inputs and outputs as the reference
// binary (/root/reference/src/gmmbin/gmm-compute-likes.cc:30-89); the frame x pdf loop
// (:71-77) runs on the GPU through libamlike.  Extra options: --device, and
// --frames-per-batch (utterances are scored together in batches of about that many
// frames, two batches in flight, and written back in input order).
#include <chrono>
#include <deque>
#include <iostream>
#include <string>
#include <vector>

#include "../host/gmm_model.h"
#include "../host/kaldi_io.h"
#include "../host/parse_options.h"
#include "../host/pipeline.h"

int main(int argc, char *argv[]) {
  using namespace kaldi_amd;
  try {
    const char *usage =
        "Compute log-likelihoods from GMM-based model\n"
        "(outputs matrices of log-likelihoods indexed by (frame, pdf)\n"
        "Usage: gmm-compute-likes [options] model-in features-rspecifier likes-wspecifier\n";
    ParseOptions po(usage);
    int32 device = 0, frames_per_batch = 32768;
    float log_sum_exp_prune = -1.0f;
    po.Register("device", &device, "GPU to use");
    po.Register("frames-per-batch", &frames_per_batch, "Frames scored per GPU launch (whole utterances)");
    po.Register("log-sum-exp-prune", &log_sum_exp_prune, "If >0, pruning value for the log-sum-exp");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) {
      po.PrintUsage();
      return 1;
    }
    CheckAmlike(amlike_init(device));
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
    am_gmm.Engine();  // packs the model and uploads it (once per process)
    const auto t0 = std::chrono::steady_clock::now();
    int32 num_done = 0;
    long long num_frames = 0;
    // Utterances are packed into batches of about frames_per_batch frames; two batches are
    // in flight, so reading and packing batch k+1 and writing batch k-1 overlap the GPU's
    // work on batch k.
    struct Batch {
      std::vector<std::string> keys;
      std::vector<int32> rows;
    };
    LikelihoodPipeline pipe(am_gmm, log_sum_exp_prune);
    std::deque<Batch> in_flight;
    auto drain_one = [&]() {
      const PinnedMatrix &likes = pipe.Front(nullptr);
      const Batch &b = in_flight.front();
      int32 r0 = 0;
      for (size_t u = 0; u < b.keys.size(); ++u) {
        loglikes_writer.Write(b.keys[u], likes.RowData(r0), b.rows[u], likes.NumCols(), likes.Stride());
        r0 += b.rows[u];
        num_done++;
      }
      num_frames += r0;
      pipe.Pop();
      in_flight.pop_front();
    };
    Batch cur;
    std::vector<Matrix> pending;
    Matrix batch;
    auto flush = [&]() {
      if (pending.empty()) return;
      int32 total = 0;
      for (const Matrix &m : pending) total += m.NumRows();
      batch.Resize(total, am_gmm.Dim());
      int32 r0 = 0;
      for (const Matrix &m : pending) {
        if (m.NumCols() != am_gmm.Dim())
          KALDI_AMD_ERR("Dim mismatch: data dim = " << m.NumCols() << " vs. model dim = " << am_gmm.Dim());
        for (int32 r = 0; r < m.NumRows(); ++r)
          std::copy(m.RowData(r), m.RowData(r) + m.NumCols(), batch.RowData(r0 + r));
        r0 += m.NumRows();
      }
      if (pipe.Full()) drain_one();
      pipe.Submit("", batch);
      in_flight.push_back(cur);
      cur = Batch();
      pending.clear();
    };
    int32 in_batch = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      if (feature_reader.Value().NumRows() == 0) continue;
      cur.keys.push_back(feature_reader.Key());
      cur.rows.push_back(feature_reader.Value().NumRows());
      pending.push_back(feature_reader.Value());
      in_batch += feature_reader.Value().NumRows();
      if (in_batch >= frames_per_batch) {
        flush();
        in_batch = 0;
      }
    }
    flush();
    while (!in_flight.empty()) drain_one();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "LOG (gmm-compute-likes) gmm-compute-likes: computed likelihoods for " << num_done
              << " utterances.\n";
    std::cerr << "LOG (gmm-compute-likes) model packed and uploaded in "
              << std::chrono::duration<double>(t0 - t_pack).count() << " s\n";
    std::cerr << "LOG (gmm-compute-likes) " << num_frames << " frames in " << secs << " s ("
              << (secs > 0 ? num_frames / secs : 0.0) << " frames/s including file I/O)\n";
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
