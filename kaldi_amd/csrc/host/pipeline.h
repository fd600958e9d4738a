// Scoring one utterance ahead of its consumer.
//
// In the reference a GMM decodable computes a likelihood when the decoder asks for it
// (/root/reference/src/gmm/decodable-am-diag-gmm.cc:28-72) and DecodableAmNnet computes the
// whole utterance in its constructor (/root/reference/src/nnet2/decodable-am-nnet.h:57-73);
// either way the decoder waits for the acoustic model.  Here the acoustic model of utterance
// k+1 runs on the GPU while the CPU decodes utterance k out of a finished matrix, which it
// sees through DecodableMatrixScaledMapped, the class the reference's matrix-fed decoders
// use (/root/reference/src/decoder/decodable-matrix.h:33-84,
//  /root/reference/src/bin/latgen-faster-mapped.cc:116).
#ifndef KALDI_AMD_CSRC_HOST_PIPELINE_H_
#define KALDI_AMD_CSRC_HOST_PIPELINE_H_

#include <string>

#include "gmm_model.h"
#include "nnet_model.h"

namespace kaldi_amd {

// A float matrix in page-locked host memory (amlike_host_alloc); grows, never shrinks.
class PinnedMatrix {
 public:
  PinnedMatrix() {}
  ~PinnedMatrix();
  PinnedMatrix(const PinnedMatrix &) = delete;
  PinnedMatrix &operator=(const PinnedMatrix &) = delete;
  void Resize(int32 rows, int32 cols);  // contents undefined afterwards
  int32 NumRows() const { return rows_; }
  int32 NumCols() const { return cols_; }
  int32 Stride() const { return cols_; }
  float *Data() { return data_; }
  const float *Data() const { return data_; }
  float *RowData(int32 r) { return data_ + static_cast<size_t>(r) * cols_; }
  const float *RowData(int32 r) const { return data_ + static_cast<size_t>(r) * cols_; }
  float operator()(int32 r, int32 c) const { return RowData(r)[c]; }
  void CopyFromMat(const Matrix &m);
  void CopyToMat(Matrix *m) const;

 private:
  float *data_ = nullptr;
  size_t capacity_ = 0;  // floats
  int32 rows_ = 0, cols_ = 0;
};

// decoder/decodable-matrix.h:33-84 over a matrix this class does not own:
// LogLikelihood(frame, tid) = scale * likes(frame, TransitionIdToPdf(tid)).
class DecodableMatrixScaledMapped : public DecodableInterface {
 public:
  DecodableMatrixScaledMapped(const TransitionModel &tm, const PinnedMatrix &likes, BaseFloat scale);
  BaseFloat LogLikelihood(int32 frame, int32 tid) override {
    return scale_ * likes_(frame, trans_model_.TransitionIdToPdf(tid));
  }
  int32 NumFramesReady() const override { return likes_.NumRows(); }
  int32 NumIndices() const override { return trans_model_.NumTransitionIds(); }
  bool IsLastFrame(int32 frame) const override { return frame == likes_.NumRows() - 1; }

 private:
  const TransitionModel &trans_model_;
  const PinnedMatrix &likes_;
  BaseFloat scale_;
};

// Up to two utterances in flight on one model: Submit() copies the features to page-locked
// memory and enqueues copy-in, kernels and copy-out; Front() blocks until the oldest one is
// complete.  Results come back in submission order.
class LikelihoodPipeline {
 public:
  // [T x NumPdfs] log-likelihoods of a GMM (unscaled, as gmm-compute-likes writes them).
  explicit LikelihoodPipeline(const AmDiagGmm &am, BaseFloat log_sum_exp_prune = -1.0);
  // nnet2 output in one of the AMLIKE_NNET_* modes (AMLIKE_NNET_LOGLIKES = DecodableAmNnet).
  LikelihoodPipeline(const nnet2::AmNnet &am, bool pad_input, int mode, BaseFloat prob_scale);
  // The same over handles the caller owns (one per device in the multi-GPU drivers); context =
  // the net's left + right context.
  LikelihoodPipeline(amlike_gmm_t *gmm, BaseFloat log_sum_exp_prune);
  LikelihoodPipeline(amlike_nnet_t *nnet, int32 context, bool pad_input, int mode, BaseFloat prob_scale);
  ~LikelihoodPipeline();
  bool Full() const { return count_ == 2; }
  bool Empty() const { return count_ == 0; }
  void Submit(const std::string &key, const Matrix &feats);
  // Rows given by pointer; the result goes to `out` (page-locked memory for full speed, at
  // least OutputRows(rows) x NumPdfs, row stride out_stride) instead of the slot's own matrix.
  void SubmitInto(const std::string &key, const float *feats, int32 rows, int32 stride, float *out,
                  int32 out_stride);
  int32 OutputRows(int32 input_rows) const { return input_rows - ctx_; }
  // Is row t of the result a function of row t of the input alone (so that utterances may be
  // stacked into one submission)?  True for a GMM, false for a net (temporal context, padding).
  bool RowsIndependent() const { return gmm_ != nullptr; }
  int32 NumPdfs() const { return num_pdfs_; }
  int32 Dim() const { return dim_; }
  // Waits for the oldest submission (as Front does) without handing out its matrix.
  void WaitFront(std::string *key);
  // The oldest submission, waited for; valid until Pop().  Throws what the synchronous call
  // would have thrown ("Invalid answer ...").
  const PinnedMatrix &Front(std::string *key);
  void Pop();

 private:
  struct Slot {
    PinnedMatrix in, out;
    bool external_out = false;
    std::string key;
    int32 ticket = -1;
    bool waited = false;
  };
  amlike_gmm_t *gmm_ = nullptr;
  amlike_nnet_t *nnet_ = nullptr;
  int32 num_pdfs_ = 0, dim_ = 0, ctx_ = 0;
  BaseFloat prune_ = -1.0f, prob_scale_ = 1.0f;
  bool pad_input_ = true;
  int mode_ = 0;
  Slot slots_[2];
  int head_ = 0, count_ = 0;
};

}  // namespace kaldi_amd

#endif  // KALDI_AMD_CSRC_HOST_PIPELINE_H_
