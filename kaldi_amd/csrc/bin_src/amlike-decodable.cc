// Dumps what a decoder would see through DecodableInterface: for every utterance a
// matrix [NumFramesReady x NumIndices] of LogLikelihood(frame, index) (index 1-based).
//   amlike-decodable gmm-scaled  [--acoustic-scale=] <mdl> <feats-rspecifier> <wspecifier>
//   amlike-decodable gmm-unmapped                     <mdl> <feats-rspecifier> <wspecifier>
//   amlike-decodable nnet        [--acoustic-scale=] [--pad-input=] <mdl> <feats-rspecifier> <wspecifier>
// The classes exercised follow gmm/decodable-am-diag-gmm.h:45-160 and
// nnet2/decodable-am-nnet.h:37-98 of the reference.
#include <iostream>
#include <memory>
#include <string>

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
    int32 device = 0;
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
    if (cmd == "nnet") am_nnet.Read(ki.Stream(), binary);
    else am_gmm.Read(ki.Stream(), binary);
    SequentialMatrixReader reader(po.GetArg(3));
    MatrixWriter writer(po.GetArg(4));
    for (; !reader.Done(); reader.Next()) {
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
