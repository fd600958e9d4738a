// Dumps what a decoder would see through DecodableInterface: for every utterance a
// matrix [NumFramesReady x NumIndices] of LogLikelihood(frame, index) (index 1-based).
//   amlike-decodable gmm-scaled  [--acoustic-scale=] <mdl> <feats-rspecifier> <wspecifier>
//   amlike-decodable gmm-unmapped                     <mdl> <feats-rspecifier> <wspecifier>
//   amlike-decodable nnet        [--acoustic-scale=] [--pad-input=] <mdl> <feats-rspecifier> <wspecifier>
//   amlike-decodable nnet-online [--acoustic-scale=] [--pad-input=] [--max-nnet-batch-size=]
//                                [--arrival-chunk=N] <mdl> <feats-rspecifier> <wspecifier>
//     (DecodableNnet2Online over features that arrive N frames at a time; the loop asks for
//      every frame as soon as NumFramesReady() allows, as an online decoder does)
// The classes exercised follow gmm/decodable-am-diag-gmm.h:45-160 and
// nnet2/decodable-am-nnet.h:37-98 and nnet2/online-nnet2-decodable.{h,cc} of the reference.
#include <algorithm>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../host/gmm_model.h"
#include "../host/kaldi_io.h"
#include "../host/nnet_model.h"
#include "../host/parse_options.h"

int main(int argc, char *argv[]) {
  using namespace kaldi_amd;
  try {
    ParseOptions po("amlike-decodable <gmm-scaled|gmm-unmapped|nnet> <mdl> <feats-rspecifier> <wspecifier>");
    float acoustic_scale = 0.1f, prune = -1.0f;
    bool pad_input = true;
    int32 device = 0, max_nnet_batch_size = 256, arrival_chunk = 100;
    po.Register("max-nnet-batch-size", &max_nnet_batch_size, "Largest batch DecodableNnet2Online computes at once");
    po.Register("arrival-chunk", &arrival_chunk, "nnet-online: frames that arrive between two rounds of look-ups");
    po.Register("acoustic-scale", &acoustic_scale, "Scaling factor for acoustic likelihoods");
    po.Register("log-sum-exp-prune", &prune, "Pruning value for the GMM log-sum-exp");
    po.Register("pad-input", &pad_input, "Pad the nnet input with copies of the edge frames");
    po.Register("device", &device, "GPU to use");
    po.Read(argc, argv);
    if (po.NumArgs() != 4) {
      po.PrintUsage();
      return 1;
    }
    CheckAmlike(amlike_init(device));
    const std::string cmd = po.GetArg(1);
    bool binary;
    Input ki(po.GetArg(2), &binary);
    TransitionModel tm;
    tm.Read(ki.Stream(), binary);
    AmDiagGmm am_gmm;
    nnet2::AmNnet am_nnet;
    if (cmd == "nnet" || cmd == "nnet-online") am_nnet.Read(ki.Stream(), binary);
    else am_gmm.Read(ki.Stream(), binary);
    SequentialMatrixReader reader(po.GetArg(3));
    MatrixWriter writer(po.GetArg(4));
    for (; !reader.Done(); reader.Next()) {
      if (cmd == "nnet-online") {
        const Matrix &feats = reader.Value();
        OnlineMatrixFeature online(feats);
        online.SetFramesReady(0);
        nnet2::DecodableNnet2OnlineOptions opts;
        opts.acoustic_scale = acoustic_scale;
        opts.pad_input = pad_input;
        opts.max_nnet_batch_size = max_nnet_batch_size;
        nnet2::DecodableNnet2Online dec(am_nnet, tm, opts, &online);
        const int32 N = dec.NumIndices();
        std::vector<std::vector<float>> rows;
        int32 done = 0, arrived = 0;
        bool finished = false;
        while (!finished) {
          arrived = std::min(feats.NumRows(), arrived + std::max(1, arrival_chunk));
          online.SetFramesReady(arrived);
          const int32 ready = dec.NumFramesReady();
          for (; done < ready; ++done) {
            rows.emplace_back(N);
            for (int32 i = 1; i <= N; ++i) rows.back()[i - 1] = dec.LogLikelihood(done, i);
            if (dec.IsLastFrame(done)) finished = true;
          }
          if (arrived == feats.NumRows() && done >= dec.NumFramesReady()) finished = true;
        }
        Matrix out(static_cast<int32>(rows.size()), N);
        for (size_t t = 0; t < rows.size(); ++t) std::copy(rows[t].begin(), rows[t].end(), out.RowData(static_cast<int32>(t)));
        std::cerr << "LOG (amlike-decodable) " << reader.Key() << ": " << rows.size() << " frames in "
                  << dec.NumBatchesComputed() << " batches\n";
        writer.Write(reader.Key(), out);
        continue;
      }
      std::unique_ptr<DecodableInterface> dec;
      if (cmd == "gmm-scaled")
        dec.reset(new DecodableAmDiagGmmScaled(am_gmm, tm, reader.Value(), acoustic_scale, prune));
      else if (cmd == "gmm-unmapped")
        dec.reset(new DecodableAmDiagGmmUnmapped(am_gmm, reader.Value(), prune));
      else if (cmd == "nnet")
        dec.reset(new nnet2::DecodableAmNnet(tm, am_nnet, reader.Value(), pad_input, acoustic_scale));
      else
        KALDI_AMD_ERR("unknown mode " << cmd);
      const int32 T = dec->NumFramesReady(), N = dec->NumIndices();
      Matrix out(T, N);
      for (int32 t = 0; t < T; ++t)
        for (int32 i = 1; i <= N; ++i) out(t, i - 1) = dec->LogLikelihood(t, i);
      if (T > 0 && (!dec->IsLastFrame(T - 1) || dec->IsLastFrame(T - 2 >= 0 ? T - 2 : -5)))
        KALDI_AMD_ERR("IsLastFrame inconsistent");
      writer.Write(reader.Key(), out);
    }
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
