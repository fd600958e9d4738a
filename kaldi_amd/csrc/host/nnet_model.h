// Host-side mirror of the reference's nnet2 forward classes, GPU-backed
// (paths relative to /root/reference/src):
//   nnet2/nnet-nnet.h:62-, nnet2/nnet-nnet.cc:89-103     Nnet (Read, dims, contexts)
//   nnet2/nnet-component.cc:33-43, Read() of each type     Component::ReadNew
//   nnet2/am-nnet.cc:31-42                                 AmNnet = Nnet + priors
//   nnet2/nnet-compute.h:44-47                             NnetComputation()
//   nnet2/decodable-am-nnet.h:37-98                        DecodableAmNnet
// Only what the forward pass needs is kept from each component (the training-side
// fields are parsed and dropped).
#ifndef KALDI_AMD_CSRC_HOST_NNET_MODEL_H_
#define KALDI_AMD_CSRC_HOST_NNET_MODEL_H_

#include <memory>
#include <string>
#include <vector>

#include "../../../include/amlike.h"
#include "gmm_model.h"
#include "kaldi_io.h"

namespace kaldi_amd {
namespace nnet2 {

struct Component {
  std::string type;  // e.g. "AffineComponentPreconditionedOnline"
  int32 kind = 0;    // enum amlike_component_kind
  int32 input_dim = 0, output_dim = 0;
  int32 left_context = 0, right_context = 0;
  int32 const_dim = 0;       // Splice: const_component_dim_; Dct: the DCT dimension
  bool dct_reorder = false;  // Dct: <Reorder>
  float p = 2.0f;            // Pnorm, Power: the exponent; Scale: the factor
  Matrix linear;  // [output_dim x input_dim]
  Vector bias;
  std::vector<int32> sizes;
};

class Nnet {
 public:
  void Read(std::istream &is, bool binary);
  int32 NumComponents() const { return static_cast<int32>(components_.size()); }
  const Component &GetComponent(int32 c) const { return components_[c]; }
  int32 InputDim() const { return components_.empty() ? 0 : components_.front().input_dim; }
  int32 OutputDim() const { return components_.empty() ? 0 : components_.back().output_dim; }
  int32 LeftContext() const;
  int32 RightContext() const;

 private:
  std::vector<Component> components_;
};

class AmNnet {
 public:
  void Read(std::istream &is, bool binary);
  const Nnet &GetNnet() const { return nnet_; }
  const Vector &Priors() const { return priors_; }
  int32 NumPdfs() const { return nnet_.OutputDim(); }
  amlike_nnet_t *Engine() const;  // device-resident pipeline, created on first use
  // A further one on the calling thread's device (amlike_init); the caller destroys it.
  amlike_nnet_t *CreateEngine() const;

 private:
  Nnet nnet_;
  Vector priors_;
  struct EngineDeleter {
    void operator()(amlike_nnet_t *n) const { amlike_nnet_destroy(n); }
  };
  mutable std::unique_ptr<amlike_nnet_t, EngineDeleter> engine_;
};

// nnet2/nnet-compute.h:44-47: output is resized to
// [feats.NumRows() (- left - right context if !pad_input)] x OutputDim().
void NnetComputation(const AmNnet &am_nnet, const Matrix &feats, bool pad_input, Matrix *output);
// The same plus the output post-processing modes of include/amlike.h
// (AMLIKE_NNET_LOG: nnet-am-compute --apply-log; AMLIKE_NNET_LOGLIKES: DecodableAmNnet).
void NnetComputation(const AmNnet &am_nnet, const Matrix &feats, bool pad_input, int mode,
                     BaseFloat prob_scale, Matrix *output);

// nnet2/decodable-am-nnet.h:37-98.
class DecodableAmNnet : public DecodableInterface {
 public:
  DecodableAmNnet(const TransitionModel &trans_model, const AmNnet &am_nnet, const Matrix &feats,
                  bool pad_input = true, BaseFloat prob_scale = 1.0);
  BaseFloat LogLikelihood(int32 frame, int32 transition_id) override {
    return log_probs_(frame, trans_model_.TransitionIdToPdf(transition_id));
  }
  int32 NumFramesReady() const override { return log_probs_.NumRows(); }
  int32 NumIndices() const override { return trans_model_.NumTransitionIds(); }
  bool IsLastFrame(int32 frame) const override { return frame == NumFramesReady() - 1; }

 private:
  const TransitionModel &trans_model_;
  Matrix log_probs_;
};

}  // namespace nnet2

// itf/online-feature-itf.h:48-80.
class OnlineFeatureInterface {
 public:
  virtual int32 Dim() const = 0;
  virtual int32 NumFramesReady() const = 0;
  virtual bool IsLastFrame(int32 frame) const = 0;
  virtual void GetFrame(int32 frame, Vector *feat) = 0;
  virtual ~OnlineFeatureInterface() {}
};

// feat/online-feature.h:114-135, plus a knob for tests and simulations: only the first
// `ready` rows count as arrived until SetFramesReady() raises the number.
class OnlineMatrixFeature : public OnlineFeatureInterface {
 public:
  explicit OnlineMatrixFeature(const Matrix &mat) : mat_(mat), ready_(mat.NumRows()) {}
  void SetFramesReady(int32 ready) { ready_ = ready < mat_.NumRows() ? ready : mat_.NumRows(); }
  int32 Dim() const override { return mat_.NumCols(); }
  int32 NumFramesReady() const override { return ready_; }
  bool IsLastFrame(int32 frame) const override { return frame + 1 == mat_.NumRows(); }
  void GetFrame(int32 frame, Vector *feat) override;

 private:
  const Matrix &mat_;
  int32 ready_;
};

namespace nnet2 {

// nnet2/online-nnet2-decodable.h:31-55.
struct DecodableNnet2OnlineOptions {
  BaseFloat acoustic_scale = 0.1f;
  bool pad_input = true;
  int32 max_nnet_batch_size = 256;
};

// nnet2/online-nnet2-decodable.{h,cc}: log-likelihoods computed in batches as the decoder
// reaches frames that have not been computed yet, from whatever input has arrived.
class DecodableNnet2Online : public DecodableInterface {
 public:
  DecodableNnet2Online(const AmNnet &nnet, const TransitionModel &trans_model,
                       const DecodableNnet2OnlineOptions &opts, OnlineFeatureInterface *input_feats);
  BaseFloat LogLikelihood(int32 frame, int32 index) override;
  bool IsLastFrame(int32 frame) const override;
  int32 NumFramesReady() const override;
  int32 NumIndices() const override { return trans_model_.NumTransitionIds(); }
  int32 NumBatchesComputed() const { return num_batches_; }

 private:
  void ComputeForFrame(int32 frame);
  OnlineFeatureInterface *features_;
  const AmNnet &nnet_;
  const TransitionModel &trans_model_;
  DecodableNnet2OnlineOptions opts_;
  int32 feat_dim_, left_context_, right_context_, num_pdfs_;
  int32 begin_frame_ = -1;
  int32 num_batches_ = 0;
  Matrix scaled_loglikes_;
};

}  // namespace nnet2
}  // namespace kaldi_amd

#endif  // KALDI_AMD_CSRC_HOST_NNET_MODEL_H_
