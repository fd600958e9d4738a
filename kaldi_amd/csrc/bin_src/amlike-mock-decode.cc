// A stand-in for the decoders' use of DecodableInterface: replays a trace of
// (frame, transition-id) look-ups per utterance -- what LatticeFasterDecoder::ProcessEmitting
// does under its beam (/root/reference/src/decoder/lattice-faster-decoder.cc:696,724) -- and
// writes, per utterance, a [T x 2] matrix: column 0 the sum of LogLikelihood(frame, tid) over
// the look-ups of that frame, column 1 their number.
//   amlike-mock-decode [options] <gmm|nnet> <mdl> <feats-rspecifier> <wspecifier>
// --double-buffer=true : utterance k+1 is scored on the GPU (LikelihoodPipeline) while the
//                        trace of utterance k is replayed; look-ups go through
//                        DecodableMatrixScaledMapped.
// --double-buffer=false: the reference's order, one utterance at a time through
//                        DecodableAmDiagGmmScaled / DecodableAmNnet.
// The trace comes from --trace-rspecifier (matrices with rows {frame, tid}) or, without it,
// from a seeded hash that touches about --touch of the transition-ids of every frame.
#include <chrono>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../host/gmm_model.h"
#include "../host/kaldi_io.h"
#include "../host/nnet_model.h"
#include "../host/parse_options.h"
#include "../host/pipeline.h"

namespace {

using namespace kaldi_amd;

inline bool Touched(uint32_t seed, uint32_t frame, uint32_t tid, uint32_t permille) {
  uint32_t h = frame * 2654435761u + tid * 40503u + seed * 97u;
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  return h % 1000u < permille;
}

struct Replay {
  const Matrix *trace = nullptr;  // rows {frame, tid}, or null for the hashed trace
  uint32_t seed = 1, permille = 250;
  int32 passes = 1;
  long long lookups = 0;

  void Run(DecodableInterface *dec, Matrix *out) {
    const int32 T = dec->NumFramesReady(), N = dec->NumIndices();
    out->Resize(T, 2);
    std::vector<double> sum(T, 0.0);
    std::vector<long long> cnt(T, 0);
    for (int32 pass = 0; pass < passes; ++pass) {
      std::fill(sum.begin(), sum.end(), 0.0);
      std::fill(cnt.begin(), cnt.end(), 0);
      if (trace) {
        for (int32 r = 0; r < trace->NumRows(); ++r) {
          const int32 f = static_cast<int32>((*trace)(r, 0)), tid = static_cast<int32>((*trace)(r, 1));
          if (f < 0 || f >= T || tid < 1 || tid > N) KALDI_AMD_ERR("trace entry out of range: " << f << " " << tid);
          sum[f] += dec->LogLikelihood(f, tid);
          cnt[f]++;
        }
      } else {
        for (int32 f = 0; f < T; ++f)
          for (int32 tid = 1; tid <= N; ++tid)
            if (Touched(seed, f, tid, permille)) {
              sum[f] += dec->LogLikelihood(f, tid);
              cnt[f]++;
            }
      }
    }
    for (int32 f = 0; f < T; ++f) {
      (*out)(f, 0) = static_cast<float>(sum[f]);
      (*out)(f, 1) = static_cast<float>(cnt[f]);
      lookups += cnt[f];
    }
    if (T > 0 && !dec->IsLastFrame(T - 1)) KALDI_AMD_ERR("IsLastFrame inconsistent");
  }
};

}  // namespace

int main(int argc, char *argv[]) {
  try {
    ParseOptions po("amlike-mock-decode [options] <gmm|nnet> <mdl> <feats-rspecifier> <wspecifier>");
    float acoustic_scale = 0.1f, prune = -1.0f, touch = 0.25f;
    bool pad_input = true, double_buffer = true;
    int32 device = 0, seed = 1, passes = 1;
    std::string trace_rspecifier;
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods");
    po.Register("log-sum-exp-prune", &prune, "Pruning value for the GMM log-sum-exp");
    po.Register("pad-input", &pad_input, "Pad the nnet input with copies of the edge frames");
    po.Register("double-buffer", &double_buffer, "Score the next utterance while this one is replayed");
    po.Register("touch", &touch, "Fraction of transition-ids looked up per frame (hashed trace)");
    po.Register("seed", &seed, "Seed of the hashed trace");
    po.Register("passes", &passes, "Times each utterance's trace is replayed (decoder work per utterance)");
    po.Register("trace-rspecifier", &trace_rspecifier, "Matrices with rows {frame, transition-id}, one per utterance");
    po.Register("device", &device, "GPU to use");
    po.Read(argc, argv);
    if (po.NumArgs() != 4) {
      po.PrintUsage();
      return 1;
    }
    CheckAmlike(amlike_init(device));
    const std::string kind = po.GetArg(1);
    if (kind != "gmm" && kind != "nnet") KALDI_AMD_ERR("unknown model kind " << kind);
    bool binary;
    Input ki(po.GetArg(2), &binary);
    TransitionModel tm;
    tm.Read(ki.Stream(), binary);
    AmDiagGmm am_gmm;
    nnet2::AmNnet am_nnet;
    if (kind == "nnet") am_nnet.Read(ki.Stream(), binary);
    else am_gmm.Read(ki.Stream(), binary);
    SequentialMatrixReader reader(po.GetArg(3));
    MatrixWriter writer(po.GetArg(4));
    std::unique_ptr<SequentialMatrixReader> trace_reader;
    if (!trace_rspecifier.empty()) trace_reader.reset(new SequentialMatrixReader(trace_rspecifier));
    Replay replay;
    replay.seed = static_cast<uint32_t>(seed);
    replay.permille = static_cast<uint32_t>(touch * 1000.0f + 0.5f);
    replay.passes = passes;
    Matrix trace, out;
    auto next_trace = [&](const std::string &key) {
      if (!trace_reader) return;
      if (trace_reader->Done() || trace_reader->Key() != key)
        KALDI_AMD_ERR("no trace for utterance " << key << " (traces must come in feature order)");
      trace = trace_reader->Value();
      replay.trace = &trace;
      trace_reader->Next();
    };
    const auto t0 = std::chrono::steady_clock::now();
    long long frames = 0;
    int32 num_done = 0;
    if (double_buffer) {
      std::unique_ptr<LikelihoodPipeline> pipe;
      if (kind == "gmm") pipe.reset(new LikelihoodPipeline(am_gmm, prune));
      else pipe.reset(new LikelihoodPipeline(am_nnet, pad_input, AMLIKE_NNET_LOGLIKES, 1.0f));
      auto feed = [&]() {
        while (!reader.Done() && !pipe->Full()) {
          if (reader.Value().NumRows() > 0) pipe->Submit(reader.Key(), reader.Value());
          reader.Next();
        }
      };
      feed();
      while (!pipe->Empty()) {
        std::string key;
        const PinnedMatrix &likes = pipe->Front(&key);
        feed();  // the other slot is free again: the GPU works on it during the replay below
        next_trace(key);
        DecodableMatrixScaledMapped dec(tm, likes, acoustic_scale);
        replay.Run(&dec, &out);
        writer.Write(key, out);
        frames += out.NumRows();
        num_done++;
        pipe->Pop();
        feed();
      }
    } else {
      for (; !reader.Done(); reader.Next()) {
        if (reader.Value().NumRows() == 0) continue;
        next_trace(reader.Key());
        std::unique_ptr<DecodableInterface> dec;
        if (kind == "gmm")
          dec.reset(new DecodableAmDiagGmmScaled(am_gmm, tm, reader.Value(), acoustic_scale, prune));
        else
          dec.reset(new nnet2::DecodableAmNnet(tm, am_nnet, reader.Value(), pad_input, acoustic_scale));
        replay.Run(dec.get(), &out);
        writer.Write(reader.Key(), out);
        frames += out.NumRows();
        num_done++;
      }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "LOG (amlike-mock-decode) " << num_done << " utterances, " << frames << " frames, "
              << replay.lookups << " look-ups in " << secs << " s ("
              << (secs > 0 ? frames / secs : 0.0) << " frames/s, double-buffer="
              << (double_buffer ? "true" : "false") << ")\n";
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
