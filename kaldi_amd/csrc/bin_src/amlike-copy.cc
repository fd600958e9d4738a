// File-format utility over the host layer (no GPU needed): reads a model or feature
// table with this build's own readers and writes it back.
//   amlike-copy gmm  [--binary=true] <mdl-in> <mdl-out>     TransitionModel + AmDiagGmm, or AmDiagGmm alone
//   amlike-copy feats <rspecifier> <wspecifier>
//   amlike-copy nnet-info <mdl-in>                           TransitionModel + AmNnet, or AmNnet alone
//   amlike-copy tid2pdf <mdl-in>                             prints "tid pdf" lines
//   amlike-copy topo-info <topo-in>                          a <Topology> object alone: per phone, its states
#include <iostream>
#include <string>

#include "../host/gmm_model.h"
#include "../host/kaldi_io.h"
#include "../host/nnet_model.h"
#include "../host/parse_options.h"

int main(int argc, char *argv[]) {
  using namespace kaldi_amd;
  try {
    ParseOptions po("amlike-copy <gmm|feats|nnet-info|tid2pdf|topo-info> ...");
    bool binary_out = true;
    po.Register("binary", &binary_out, "Write output in binary mode");
    po.Read(argc, argv);
    if (po.NumArgs() < 2) {
      po.PrintUsage();
      return 1;
    }
    const std::string cmd = po.GetArg(1);
    auto maybe_trans = [](std::istream &is, bool binary, TransitionModel *tm) {
      // a model file starts with <TransitionModel> when it holds one
      std::streampos pos = is.tellg();
      std::string tok;
      ReadToken(is, binary, &tok);
      is.seekg(pos);
      if (tok != "<TransitionModel>") return false;
      tm->Read(is, binary);
      return true;
    };
    if (cmd == "gmm" && po.NumArgs() == 3) {
      bool binary;
      Input ki(po.GetArg(2), &binary);
      TransitionModel tm;
      maybe_trans(ki.Stream(), binary, &tm);
      AmDiagGmm am;
      am.Read(ki.Stream(), binary);
      Output ko(po.GetArg(3), binary_out);
      am.Write(ko.Stream(), binary_out);
      std::cerr << "pdfs " << am.NumPdfs() << " gauss " << am.NumGauss() << " dim " << am.Dim() << "\n";
      return 0;
    } else if (cmd == "feats" && po.NumArgs() == 3) {
      SequentialMatrixReader reader(po.GetArg(2));
      MatrixWriter writer(po.GetArg(3));
      int n = 0;
      for (; !reader.Done(); reader.Next(), ++n) writer.Write(reader.Key(), reader.Value());
      std::cerr << "copied " << n << " matrices\n";
      return 0;
    } else if (cmd == "nnet-info" && po.NumArgs() == 2) {
      bool binary;
      Input ki(po.GetArg(2), &binary);
      TransitionModel tm;
      maybe_trans(ki.Stream(), binary, &tm);
      nnet2::AmNnet am;
      am.Read(ki.Stream(), binary);
      const nnet2::Nnet &n = am.GetNnet();
      std::cout << "input-dim " << n.InputDim() << " output-dim " << n.OutputDim() << " left-context "
                << n.LeftContext() << " right-context " << n.RightContext() << " num-components "
                << n.NumComponents() << " priors-dim " << am.Priors().Dim() << "\n";
      return 0;
    } else if (cmd == "tid2pdf" && po.NumArgs() == 2) {
      bool binary;
      Input ki(po.GetArg(2), &binary);
      TransitionModel tm;
      tm.Read(ki.Stream(), binary);
      for (int32 tid = 1; tid <= tm.NumTransitionIds(); ++tid)
        std::cout << tid << " " << tm.TransitionIdToPdf(tid) << "\n";
      return 0;
    } else if (cmd == "topo-info" && po.NumArgs() == 2) {
      // one line per (phone, state): phone num-pdf-classes state pdf-class then dst:prob per transition
      bool binary;
      Input ki(po.GetArg(2), &binary);
      TransitionModel tm;
      tm.ReadTopology(ki.Stream(), binary);
      for (int32 phone : tm.GetPhones()) {
        const auto &entry = tm.TopologyForPhone(phone);
        for (size_t st = 0; st < entry.size(); ++st) {
          std::cout << phone << " " << tm.NumPdfClasses(phone) << " " << st << " " << entry[st].pdf_class;
          for (const auto &tr : entry[st].transitions) std::cout << " " << tr.first << ":" << tr.second;
          std::cout << "\n";
        }
      }
      return 0;
    }
    po.PrintUsage();
    return 1;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return -1;
  }
}
