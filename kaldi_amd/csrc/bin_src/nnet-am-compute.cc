// nnet-am-compute on MI355X: same command line, inputs and outputs as the reference
// binary (/root/reference/src/nnet2bin/nnet-am-compute.cc:27-110); NnetComputation()
// (:91) and the optional floor + log (:93-96) run on the GPU through libamlike.
#include <chrono>
#include <iostream>
#include <string>

#include "../host/kaldi_io.h"
#include "../host/nnet_model.h"
#include "../host/parse_options.h"
#include "../host/pipeline.h"

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
    int32 device = 0;
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
    po.Register("device", &device, "GPU to use");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) {
      po.PrintUsage();
      return 1;
    }
    CheckAmlike(amlike_init(device));
    TransitionModel trans_model;
    AmNnet am_nnet;
    {
      bool binary;
      Input ki(po.GetArg(1), &binary);
      trans_model.Read(ki.Stream(), binary);
      am_nnet.Read(ki.Stream(), binary);
    }
    const Nnet &nnet = am_nnet.GetNnet();
    long long num_done = 0, num_frames = 0;
    SequentialMatrixReader feature_reader(po.GetArg(2));
    MatrixWriter writer(po.GetArg(3));
    const int mode = divide_by_priors ? AMLIKE_NNET_LOGLIKES : apply_log ? AMLIKE_NNET_LOG : AMLIKE_NNET_POSTERIORS;
    // Two utterances in flight: reading utterance k+1 and writing utterance k-1 overlap the
    // GPU's work on utterance k.
    am_nnet.Engine();  // weight images built and uploaded before the clock starts
    const auto t0 = std::chrono::steady_clock::now();
    LikelihoodPipeline pipe(am_nnet, pad_input, mode, prob_scale);
    auto drain_one = [&]() {
      std::string utt;
      const PinnedMatrix &output = pipe.Front(&utt);
      writer.Write(utt, output.Data(), output.NumRows(), output.NumCols(), output.Stride());
      pipe.Pop();
      num_done++;
    };
    for (; !feature_reader.Done(); feature_reader.Next()) {
      const std::string utt = feature_reader.Key();
      const Matrix &feats = feature_reader.Value();
      int32 output_frames = feats.NumRows();
      if (!pad_input) output_frames -= nnet.LeftContext() + nnet.RightContext();
      if (output_frames <= 0) {
        std::cerr << "WARNING (nnet-am-compute) Skipping utterance " << utt
                  << " because output would be empty.\n";
        continue;
      }
      if (pipe.Full()) drain_one();
      pipe.Submit(utt, feats);
      num_frames += feats.NumRows();
    }
    while (!pipe.Empty()) drain_one();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "LOG (nnet-am-compute) " << num_frames << " frames in " << secs << " s ("
              << (secs > 0 ? num_frames / secs : 0.0) << " frames/s including file I/O)\n";
    std::cerr << "LOG (nnet-am-compute) Processed " << num_done << " feature files, " << num_frames
              << " frames of input were processed.\n";
    return (num_done == 0 ? 1 : 0);
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
