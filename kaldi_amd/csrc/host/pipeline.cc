#include "pipeline.h"

#include <cstring>

namespace kaldi_amd {

PinnedMatrix::~PinnedMatrix() {
  if (data_) amlike_host_free(data_);
}

void PinnedMatrix::Resize(int32 rows, int32 cols) {
  const size_t want = static_cast<size_t>(rows) * cols;
  if (want > capacity_) {
    if (data_) CheckAmlike(amlike_host_free(data_));
    data_ = nullptr;
    capacity_ = 0;
    const size_t grow = want + want / 4;
    void *p = nullptr;
    CheckAmlike(amlike_host_alloc(grow * sizeof(float), &p));
    data_ = static_cast<float *>(p);
    capacity_ = grow;
  }
  rows_ = rows;
  cols_ = cols;
}

void PinnedMatrix::CopyFromMat(const Matrix &m) {
  Resize(m.NumRows(), m.NumCols());
  for (int32 r = 0; r < rows_; ++r) memcpy(RowData(r), m.RowData(r), sizeof(float) * cols_);
}

void PinnedMatrix::CopyToMat(Matrix *m) const {
  m->Resize(rows_, cols_);
  for (int32 r = 0; r < rows_; ++r) memcpy(m->RowData(r), RowData(r), sizeof(float) * cols_);
}

DecodableMatrixScaledMapped::DecodableMatrixScaledMapped(const TransitionModel &tm,
                                                         const PinnedMatrix &likes, BaseFloat scale)
    : trans_model_(tm), likes_(likes), scale_(scale) {
  if (likes.NumCols() != tm.NumPdfs())
    KALDI_AMD_ERR("DecodableMatrixScaledMapped: mismatch, matrix has " << likes.NumCols()
                  << " columns but transition-model has " << tm.NumPdfs() << " pdf-ids.");
}

LikelihoodPipeline::LikelihoodPipeline(const AmDiagGmm &am, BaseFloat log_sum_exp_prune)
    : gmm_(am.Engine()), num_pdfs_(am.NumPdfs()), dim_(am.Dim()), prune_(log_sum_exp_prune) {}

LikelihoodPipeline::LikelihoodPipeline(const nnet2::AmNnet &am, bool pad_input, int mode,
                                       BaseFloat prob_scale)
    : nnet_(am.Engine()), num_pdfs_(am.NumPdfs()), dim_(am.GetNnet().InputDim()),
      prob_scale_(prob_scale), pad_input_(pad_input), mode_(mode) {
  ctx_ = pad_input ? 0 : am.GetNnet().LeftContext() + am.GetNnet().RightContext();
}

LikelihoodPipeline::LikelihoodPipeline(amlike_gmm_t *gmm, BaseFloat log_sum_exp_prune)
    : gmm_(gmm), num_pdfs_(amlike_gmm_num_pdfs(gmm)), dim_(amlike_gmm_dim(gmm)), prune_(log_sum_exp_prune) {}

LikelihoodPipeline::LikelihoodPipeline(amlike_nnet_t *nnet, int32 context, bool pad_input, int mode,
                                       BaseFloat prob_scale)
    : nnet_(nnet), num_pdfs_(amlike_nnet_output_dim(nnet)), dim_(amlike_nnet_input_dim(nnet)),
      prob_scale_(prob_scale), pad_input_(pad_input), mode_(mode) {
  ctx_ = pad_input ? 0 : context;
}

LikelihoodPipeline::~LikelihoodPipeline() {
  // the buffers must not be freed under a copy in flight
  while (count_ > 0) {
    Slot &s = slots_[head_];
    if (!s.waited) {
      if (gmm_) (void)amlike_gmm_score_wait(gmm_, s.ticket);
      else (void)amlike_nnet_compute_wait(nnet_, s.ticket);
    }
    head_ ^= 1;
    --count_;
  }
}

void LikelihoodPipeline::Submit(const std::string &key, const Matrix &feats) {
  if (feats.NumCols() != dim_)
    KALDI_AMD_ERR("Dim mismatch: data dim = " << feats.NumCols() << " vs. model dim = " << dim_);
  SubmitInto(key, feats.Data(), feats.NumRows(), feats.Stride(), nullptr, 0);
}

void LikelihoodPipeline::SubmitInto(const std::string &key, const float *feats, int32 rows, int32 stride,
                                    float *out, int32 out_stride) {
  if (Full()) KALDI_AMD_ERR("LikelihoodPipeline::Submit: two utterances already in flight");
  const int32 t_out = rows - ctx_;
  if (t_out <= 0) KALDI_AMD_ERR("LikelihoodPipeline::Submit: utterance " << key << " gives no output");
  Slot &s = slots_[(head_ + count_) & 1];
  s.in.Resize(rows, dim_);
  for (int32 r = 0; r < rows; ++r)
    memcpy(s.in.RowData(r), feats + static_cast<size_t>(r) * stride, sizeof(float) * dim_);
  s.external_out = out != nullptr;
  if (!out) {
    s.out.Resize(t_out, num_pdfs_);
    out = s.out.Data();
    out_stride = s.out.Stride();
  }
  s.key = key;
  s.waited = false;
  if (gmm_)
    CheckAmlike(amlike_gmm_score_submit(gmm_, s.in.Data(), s.in.NumRows(), s.in.Stride(), dim_, prune_,
                                        out, out_stride, &s.ticket));
  else
    CheckAmlike(amlike_nnet_compute_submit(nnet_, s.in.Data(), s.in.NumRows(), s.in.Stride(),
                                           pad_input_ ? 1 : 0, mode_, prob_scale_, out, out_stride,
                                           &s.ticket));
  ++count_;
}

void LikelihoodPipeline::WaitFront(std::string *key) {
  if (Empty()) KALDI_AMD_ERR("LikelihoodPipeline::WaitFront: nothing submitted");
  Slot &s = slots_[head_];
  if (key) *key = s.key;
  if (!s.waited) {
    s.waited = true;
    CheckAmlike(gmm_ ? amlike_gmm_score_wait(gmm_, s.ticket) : amlike_nnet_compute_wait(nnet_, s.ticket));
  }
}

const PinnedMatrix &LikelihoodPipeline::Front(std::string *key) {
  if (Empty()) KALDI_AMD_ERR("LikelihoodPipeline::Front: nothing submitted");
  Slot &s = slots_[head_];
  if (!s.waited) {
    s.waited = true;
    CheckAmlike(gmm_ ? amlike_gmm_score_wait(gmm_, s.ticket) : amlike_nnet_compute_wait(nnet_, s.ticket));
  }
  if (key) *key = s.key;
  if (s.external_out) KALDI_AMD_ERR("LikelihoodPipeline::Front: the result went to the caller's buffer (WaitFront)");
  return s.out;
}

void LikelihoodPipeline::Pop() {
  if (Empty()) return;
  Slot &s = slots_[head_];
  if (!s.waited) {
    s.waited = true;
    CheckAmlike(gmm_ ? amlike_gmm_score_wait(gmm_, s.ticket) : amlike_nnet_compute_wait(nnet_, s.ticket));
  }
  head_ ^= 1;
  --count_;
}

}  // namespace kaldi_amd
