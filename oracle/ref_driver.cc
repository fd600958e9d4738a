// oracle/ref_driver.cc
//
// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or called from the
// product path.  A thin command-line driver over the REAL reference library
// (compiled from /root/reference by oracle/build_ref.sh).  Each sub-command
// reproduces, with the reference's own classes, one loop of the
// acoustic-likelihood path:
//
//   gmm-likes       gmmbin/gmm-compute-likes.cc:68-80  (AmDiagGmm::LogLikelihood per frame,pdf)
//   gmm-likes-prune gmm/decodable-am-diag-gmm.cc:58-64 (same, LogSumExp(prune))
//   gmm-batch       gmm/diag-gmm.cc:546-562            (DiagGmm::LogLikelihoods(Matrix))
//   gmm-gconsts     gmm/diag-gmm.cc:114-152            (gconsts as recomputed by Read)
//   nnet-compute    nnet2/nnet-compute.cc:62-106 + nnet2bin/nnet-am-compute.cc:82-96
//   nnet-loglikes   nnet2/decodable-am-nnet.h:57-69
//   splice-transform featbin/splice-feats.cc (SpliceFrames, feat/feature-functions.cc:391-412)
//                   then featbin/transform-feats.cc:81-108 once or twice (LDA/MLLT, then fMLLR)
//   *-copy          Read then Write with the reference's own IO (pins the file formats)
//
// Model files given to this driver hold ONLY the AmDiagGmm / AmNnet object (no
// TransitionModel in front): hmm/transition-model.h needs an OpenFst header the
// image lacks, so that class is not part of this build.
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "base/kaldi-common.h"
#include "base/timer.h"
#include "util/common-utils.h"
#include "gmm/am-diag-gmm.h"
#include "nnet2/am-nnet.h"
#include "feat/feature-functions.h"

using namespace kaldi;

static void ReadAm(const std::string &f, AmDiagGmm *am) {
  bool binary;
  Input ki(f, &binary);
  am->Read(ki.Stream(), binary);
}

static void ReadAmNnet(const std::string &f, nnet2::AmNnet *am) {
  bool binary;
  Input ki(f, &binary);
  am->Read(ki.Stream(), binary);
}

// nnet2/nnet-compute.cc:62-106 restated over the reference's Component::Propagate.
static void RefNnetComputation(const nnet2::Nnet &nnet,
                               const CuMatrixBase<BaseFloat> &feats, bool pad,
                               CuMatrix<BaseFloat> *out) {
  int32 dim = feats.NumCols();
  if (dim != nnet.InputDim())
    KALDI_ERR << "Feature dimension is " << dim << " but network expects "
              << nnet.InputDim();
  int32 left = pad ? nnet.LeftContext() : 0, right = pad ? nnet.RightContext() : 0;
  int32 num_rows = left + feats.NumRows() + right;
  CuMatrix<BaseFloat> cur(num_rows, dim);
  cur.Range(left, feats.NumRows(), 0, dim).CopyFromMat(feats);
  for (int32 i = 0; i < left; i++) cur.Row(i).CopyFromVec(feats.Row(0));
  for (int32 i = 0; i < right; i++)
    cur.Row(num_rows - i - 1).CopyFromVec(feats.Row(feats.NumRows() - 1));
  for (int32 c = 0; c < nnet.NumComponents(); c++) {
    CuMatrix<BaseFloat> next;
    nnet.GetComponent(c).Propagate(cur, 1, &next);
    cur.Swap(&next);
  }
  out->Swap(&cur);
}

int main(int argc, char *argv[]) {
  try {
    if (argc < 2) { std::cerr << "usage: ref_driver <cmd> ...\n"; return 2; }
    std::string cmd = argv[1];
    if (cmd == "gmm-likes" || cmd == "gmm-likes-prune") {
      AmDiagGmm am; ReadAm(argv[2], &am);
      BaseFloat prune = (cmd == "gmm-likes-prune") ? atof(argv[5]) : -1.0;
      SequentialBaseFloatMatrixReader reader(argv[3]);
      BaseFloatMatrixWriter writer(argv[4]);
      double secs = 0.0; int64 frames = 0;
      for (; !reader.Done(); reader.Next()) {
        const Matrix<BaseFloat> &features(reader.Value());
        Matrix<BaseFloat> loglikes(features.NumRows(), am.NumPdfs());
        Timer t;
        if (cmd == "gmm-likes") {
          for (int32 i = 0; i < features.NumRows(); i++)
            for (int32 j = 0; j < am.NumPdfs(); j++) {
              SubVector<BaseFloat> row(features, i);
              loglikes(i, j) = am.LogLikelihood(j, row);
            }
        } else {
          Vector<BaseFloat> sq(features.NumCols());
          for (int32 i = 0; i < features.NumRows(); i++) {
            sq.CopyFromVec(features.Row(i)); sq.ApplyPow(2.0);
            for (int32 j = 0; j < am.NumPdfs(); j++) {
              const DiagGmm &pdf = am.GetPdf(j);
              Vector<BaseFloat> ll(pdf.gconsts());
              ll.AddMatVec(1.0, pdf.means_invvars(), kNoTrans, features.Row(i), 1.0);
              ll.AddMatVec(-0.5, pdf.inv_vars(), kNoTrans, sq, 1.0);
              loglikes(i, j) = ll.LogSumExp(prune);
            }
          }
        }
        secs += t.Elapsed(); frames += features.NumRows();
        writer.Write(reader.Key(), loglikes);
      }
      std::cout << "TIME " << secs << " FRAMES " << frames << "\n";
      return 0;
    } else if (cmd == "gmm-batch") {
      AmDiagGmm am; ReadAm(argv[2], &am);
      int32 pdf = atoi(argv[3]);
      SequentialBaseFloatMatrixReader reader(argv[4]);
      BaseFloatMatrixWriter writer(argv[5]);
      for (; !reader.Done(); reader.Next()) {
        Matrix<BaseFloat> ll;
        am.GetPdf(pdf).LogLikelihoods(reader.Value(), &ll);
        writer.Write(reader.Key(), ll);
      }
      return 0;
    } else if (cmd == "gmm-gconsts") {
      AmDiagGmm am; ReadAm(argv[2], &am);
      Vector<BaseFloat> all(am.NumGauss());
      int32 k = 0;
      for (int32 j = 0; j < am.NumPdfs(); j++)
        for (int32 m = 0; m < am.GetPdf(j).NumGauss(); m++)
          all(k++) = am.GetPdf(j).gconsts()(m);
      Output ko(argv[3], true);
      all.Write(ko.Stream(), true);
      return 0;
    } else if (cmd == "gmm-copy") {
      AmDiagGmm am; ReadAm(argv[2], &am);
      bool binary = atoi(argv[4]) != 0;
      Output ko(argv[3], binary);
      am.Write(ko.Stream(), binary);
      return 0;
    } else if (cmd == "nnet-compute" || cmd == "nnet-loglikes") {
      nnet2::AmNnet am; ReadAmNnet(argv[2], &am);
      bool pad = atoi(argv[5]) != 0;
      SequentialBaseFloatCuMatrixReader reader(argv[3]);
      BaseFloatCuMatrixWriter writer(argv[4]);
      const nnet2::Nnet &nnet = am.GetNnet();
      double secs = 0.0; int64 frames = 0;
      for (; !reader.Done(); reader.Next()) {
        const CuMatrix<BaseFloat> &feats = reader.Value();
        int32 out_frames = feats.NumRows();
        if (!pad) out_frames -= nnet.LeftContext() + nnet.RightContext();
        if (out_frames <= 0) { KALDI_WARN << "Skipping " << reader.Key(); continue; }
        Timer t;
        CuMatrix<BaseFloat> output;
        RefNnetComputation(nnet, feats, pad, &output);
        if (cmd == "nnet-compute") {
          if (atoi(argv[6]) != 0) { output.ApplyFloor(1.0e-20); output.ApplyLog(); }
        } else {
          BaseFloat prob_scale = atof(argv[6]);
          output.ApplyFloor(1.0e-20);
          output.ApplyLog();
          CuVector<BaseFloat> priors(am.Priors());
          priors.ApplyLog();
          output.AddVecToRows(-1.0, priors);
          output.Scale(prob_scale);
        }
        secs += t.Elapsed(); frames += feats.NumRows();
        writer.Write(reader.Key(), output);
      }
      std::cout << "TIME " << secs << " FRAMES " << frames << "\n";
      return 0;
    } else if (cmd == "nnet-copy") {
      nnet2::AmNnet am; ReadAmNnet(argv[2], &am);
      bool binary = atoi(argv[4]) != 0;
      Output ko(argv[3], binary);
      am.Write(ko.Stream(), binary);
      return 0;
    } else if (cmd == "nnet-info") {
      nnet2::AmNnet am; ReadAmNnet(argv[2], &am);
      const nnet2::Nnet &n = am.GetNnet();
      std::cout << "input-dim " << n.InputDim() << " output-dim " << n.OutputDim()
                << " left-context " << n.LeftContext() << " right-context "
                << n.RightContext() << " num-components " << n.NumComponents()
                << " priors-dim " << am.Priors().Dim() << "\n";
      return 0;
    } else if (cmd == "gmm-gselect") {
      // DiagGmm::GaussianSelection of pdf argv[3]; one text line per frame: "key t idx..." and a
      // final "TOTAL <sum of per-frame log-likelihoods>" line per utterance
      AmDiagGmm am; ReadAm(argv[2], &am);
      int32 pdf = atoi(argv[3]), num_gselect = atoi(argv[4]);
      SequentialBaseFloatMatrixReader reader(argv[5]);
      std::ofstream os(argv[6]);
      os.precision(9);
      for (; !reader.Done(); reader.Next()) {
        std::vector<std::vector<int32> > gselect;
        BaseFloat tot = am.GetPdf(pdf).GaussianSelection(reader.Value(), num_gselect, &gselect);
        for (size_t t = 0; t < gselect.size(); t++) {
          os << reader.Key() << " " << t;
          for (size_t j = 0; j < gselect[t].size(); j++) os << " " << gselect[t][j];
          os << "\n";
        }
        os << reader.Key() << " TOTAL " << tot << "\n";
      }
      return 0;
    } else if (cmd == "compress-feats") {
      // writes every matrix as a CompressedMatrix ("CM" / "CM2" on disk)
      SequentialBaseFloatMatrixReader reader(argv[2]);
      CompressedMatrixWriter writer(argv[3]);
      for (; !reader.Done(); reader.Next()) writer.Write(reader.Key(), CompressedMatrix(reader.Value()));
      return 0;
    } else if (cmd == "splice-transform") {
      // ref_driver splice-transform <feats-rspecifier> <left> <right> <out-wspecifier> [mat1 [mat2]]
      SequentialBaseFloatMatrixReader reader(argv[2]);
      const int32 left = atoi(argv[3]), right = atoi(argv[4]);
      BaseFloatMatrixWriter writer(argv[5]);
      std::vector<Matrix<BaseFloat> > mats;
      for (int a = 6; a < argc; a++) {
        Matrix<BaseFloat> m;
        ReadKaldiObject(argv[a], &m);
        mats.push_back(m);
      }
      for (; !reader.Done(); reader.Next()) {
        Matrix<BaseFloat> feat;
        SpliceFrames(reader.Value(), left, right, &feat);
        for (size_t i = 0; i < mats.size(); i++) {
          // the body of transform-feats.cc:92-108
          const Matrix<BaseFloat> &trans = mats[i];
          int32 transform_rows = trans.NumRows(), transform_cols = trans.NumCols(),
                feat_dim = feat.NumCols();
          Matrix<BaseFloat> feat_out(feat.NumRows(), transform_rows);
          if (transform_cols == feat_dim) {
            feat_out.AddMatMat(1.0, feat, kNoTrans, trans, kTrans, 0.0);
          } else if (transform_cols == feat_dim + 1) {
            SubMatrix<BaseFloat> linear_part(trans, 0, transform_rows, 0, feat_dim);
            feat_out.AddMatMat(1.0, feat, kNoTrans, linear_part, kTrans, 0.0);
            Vector<BaseFloat> offset(transform_rows);
            offset.CopyColFromMat(trans, feat_dim);
            feat_out.AddVecToRows(1.0, offset);
          } else {
            KALDI_ERR << "Transform matrix has bad dimension " << transform_rows << "x"
                      << transform_cols << " versus feat dim " << feat_dim;
          }
          feat.Swap(&feat_out);
        }
        writer.Write(reader.Key(), feat);
      }
      return 0;
    } else if (cmd == "copy-feats") {
      SequentialBaseFloatMatrixReader reader(argv[2]);
      BaseFloatMatrixWriter writer(argv[3]);
      for (; !reader.Done(); reader.Next()) writer.Write(reader.Key(), reader.Value());
      return 0;
    }
    std::cerr << "ref_driver: unknown command " << cmd << "\n";
    return 2;
  } catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return 1;
  }
}
