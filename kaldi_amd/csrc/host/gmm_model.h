// Host-side mirror of the reference's GMM acoustic-model classes, GPU-backed.
// Names, argument meaning and error behaviour follow the reference
// (paths relative to /root/reference/src):
//   hmm/transition-model.h, hmm/hmm-topology.h   TransitionModel (read-only: tid -> pdf)
//   gmm/diag-gmm.h:43-250                        DiagGmm
//   gmm/am-diag-gmm.h:36-111                     AmDiagGmm
//   itf/decodable-itf.h:83-119                   DecodableInterface
//   gmm/decodable-am-diag-gmm.h:45-160           DecodableAmDiagGmmUnmapped / ...Scaled
// Every likelihood is computed on the GPU through the amlike C ABI; there is no
// CPU fallback (a missing GPU raises, as KALDI_ERR would).
#ifndef KALDI_AMD_CSRC_HOST_GMM_MODEL_H_
#define KALDI_AMD_CSRC_HOST_GMM_MODEL_H_

#include <memory>
#include <string>
#include <vector>

#include "../../../include/amlike.h"
#include "kaldi_io.h"

namespace kaldi_amd {

// Throws std::runtime_error carrying amlike_last_error() when rc != AMLIKE_OK.
void CheckAmlike(int rc);

// hmm/hmm-topology.cc:39-139 + hmm/transition-model.cc:72-97,274-294.
class TransitionModel {
 public:
  void Read(std::istream &is, bool binary);
  int32 NumTransitionIds() const { return static_cast<int32>(id2state_.size()) - 1; }
  int32 NumPdfs() const { return num_pdfs_; }
  int32 TransitionIdToPdf(int32 trans_id) const;  // trans_id is 1-based
  const std::vector<int32> &TidToPdfTable() const { return tid2pdf_; }  // index 0 unused

  // The HmmTopology part (hmm/hmm-topology.h:105-160): one entry of states per group of phones.
  struct HmmState {
    int32 pdf_class;  // -1 (kNoPdf) for a non-emitting state
    std::vector<std::pair<int32, float>> transitions;  // (destination state, probability)
  };
  void ReadTopology(std::istream &is, bool binary);  // HmmTopology::Read, a <Topology> object alone
  const std::vector<int32> &GetPhones() const { return phones_; }  // sorted
  const std::vector<HmmState> &TopologyForPhone(int32 phone) const;
  int32 NumPdfClasses(int32 phone) const;  // 1 + the largest pdf class of the phone's entry

 private:
  struct Triple {
    int32 phone, hmm_state, pdf;
  };
  void ComputeDerived();
  std::vector<int32> phones_, phone2idx_;
  std::vector<std::vector<HmmState>> entries_;
  std::vector<Triple> triples_;
  std::vector<int32> state2id_, id2state_, tid2pdf_;
  Vector log_probs_;
  int32 num_pdfs_ = 0;
};

class DiagGmm {
 public:
  int32 Dim() const { return means_invvars_.NumCols(); }
  int32 NumGauss() const { return weights_.Dim(); }
  void Read(std::istream &is, bool binary);  // recomputes the gconsts, as the reference does
  void Write(std::ostream &os, bool binary) const;
  int32 ComputeGconsts();
  bool valid_gconsts() const { return valid_gconsts_; }
  const Vector &gconsts() const { return gconsts_; }
  const Vector &weights() const { return weights_; }
  const Matrix &means_invvars() const { return means_invvars_; }
  const Matrix &inv_vars() const { return inv_vars_; }
  // Test / synthetic-model entry: sets the parameters and computes the gconsts.
  void SetParams(const Vector &weights, const Matrix &means_invvars, const Matrix &inv_vars);

  // gmm/diag-gmm.cc:517-526: log-likelihood of one frame (a T = 1 GPU call).
  BaseFloat LogLikelihood(const Vector &data) const;
  // gmm/diag-gmm.cc:528-543 / :546-562: per-Gaussian log-likelihoods.
  void LogLikelihoods(const Vector &data, Vector *loglikes) const;
  void LogLikelihoods(const Matrix &data, Matrix *loglikes) const;
  // gmm/diag-gmm.cc:801-871: per frame the num_gselect best Gaussians, best first; returns
  // the sum over frames of the log-sum of the selected scores.
  BaseFloat GaussianSelection(const Matrix &data, int32 num_gselect,
                              std::vector<std::vector<int32>> *output) const;

 private:
  amlike_gmm_t *Engine() const;
  Vector gconsts_, weights_;
  Matrix inv_vars_, means_invvars_;
  bool valid_gconsts_ = false;
  struct EngineDeleter {
    void operator()(amlike_gmm_t *g) const { amlike_gmm_destroy(g); }
  };
  mutable std::unique_ptr<amlike_gmm_t, EngineDeleter> engine_;
};

class AmDiagGmm {
 public:
  void Read(std::istream &is, bool binary);
  void Write(std::ostream &os, bool binary) const;
  int32 Dim() const { return densities_.empty() ? 0 : densities_[0].Dim(); }
  int32 NumPdfs() const { return static_cast<int32>(densities_.size()); }
  int32 NumGauss() const;
  const DiagGmm &GetPdf(int32 pdf) const { return densities_[pdf]; }
  void AddPdf(const DiagGmm &gmm) {
    densities_.push_back(DiagGmm());
    densities_.back().SetParams(gmm.weights(), gmm.means_invvars(), gmm.inv_vars());
    engine_.reset();
  }

  // gmm/am-diag-gmm.h:108-111: one (pdf, frame) log-likelihood (a T = 1 GPU call over all pdfs).
  BaseFloat LogLikelihood(int32 pdf_index, const Vector &data) const;
  // The dense form of the gmm-compute-likes loop (gmmbin/gmm-compute-likes.cc:71-77):
  // loglikes(t, j) = LogLikelihood(j, feats.Row(t)); log_sum_exp_prune as in
  // gmm/decodable-am-diag-gmm.cc:64.  Throws "Invalid answer ..." on NaN/Inf.
  void LogLikelihoods(const Matrix &feats, Matrix *loglikes, BaseFloat log_sum_exp_prune = -1.0) const;

  amlike_gmm_t *Engine() const;  // the packed device copy (created on first use)
  // A further device copy on the calling thread's device (amlike_init); the caller destroys it.
  amlike_gmm_t *CreateEngine() const;

 private:
  std::vector<DiagGmm> densities_;
  struct EngineDeleter {
    void operator()(amlike_gmm_t *g) const { amlike_gmm_destroy(g); }
  };
  mutable std::unique_ptr<amlike_gmm_t, EngineDeleter> engine_;
};

// itf/decodable-itf.h:83-119.
class DecodableInterface {
 public:
  virtual BaseFloat LogLikelihood(int32 frame, int32 index) = 0;  // index is 1-based
  virtual bool IsLastFrame(int32 frame) const = 0;
  virtual int32 NumFramesReady() const = 0;
  virtual int32 NumIndices() const = 0;
  virtual ~DecodableInterface() {}
};

// gmm/decodable-am-diag-gmm.h:45-95.  The reference computes a (frame, pdf) pair when
// the decoder first asks for it; here the whole [T x NumPdfs] matrix is computed on the
// GPU in the constructor (the shape DecodableAmNnet already has in the reference,
// nnet2/decodable-am-nnet.h:57-80) and LogLikelihood() is a lookup.
class DecodableAmDiagGmmUnmapped : public DecodableInterface {
 public:
  DecodableAmDiagGmmUnmapped(const AmDiagGmm &am, const Matrix &feats,
                             BaseFloat log_sum_exp_prune = -1.0);
  BaseFloat LogLikelihood(int32 frame, int32 state_index) override {
    return LogLikelihoodZeroBased(frame, state_index - 1);
  }
  int32 NumFramesReady() const override { return loglikes_.NumRows(); }
  int32 NumIndices() const override { return loglikes_.NumCols(); }
  bool IsLastFrame(int32 frame) const override;

 protected:
  BaseFloat LogLikelihoodZeroBased(int32 frame, int32 state) const;
  Matrix loglikes_;
};

// gmm/decodable-am-diag-gmm.h:121-160.
class DecodableAmDiagGmmScaled : public DecodableAmDiagGmmUnmapped {
 public:
  DecodableAmDiagGmmScaled(const AmDiagGmm &am, const TransitionModel &tm, const Matrix &feats,
                           BaseFloat scale, BaseFloat log_sum_exp_prune = -1.0)
      : DecodableAmDiagGmmUnmapped(am, feats, log_sum_exp_prune), trans_model_(tm), scale_(scale) {}
  BaseFloat LogLikelihood(int32 frame, int32 tid) override {
    return scale_ * LogLikelihoodZeroBased(frame, trans_model_.TransitionIdToPdf(tid));
  }
  int32 NumIndices() const override { return trans_model_.NumTransitionIds(); }
  const TransitionModel *TransModel() { return &trans_model_; }

 private:
  const TransitionModel &trans_model_;
  BaseFloat scale_;
};

}  // namespace kaldi_amd

#endif  // KALDI_AMD_CSRC_HOST_GMM_MODEL_H_
