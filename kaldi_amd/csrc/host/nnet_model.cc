// Host-side nnet2 model classes (see nnet_model.h).
#include "nnet_model.h"

#include <algorithm>

namespace kaldi_amd {
namespace nnet2 {

// Reads what follows "<Type>" of a NonlinearComponent (nnet2/nnet-component.cc:365-388).
static void ReadNonlinear(std::istream &is, bool binary, Component *c) {
  const std::string beg = "<" + c->type + ">", end = "</" + c->type + ">";
  ExpectOneOrTwoTokens(is, binary, beg, "<Dim>");
  ReadBasicType(is, binary, &c->input_dim);
  c->output_dim = c->input_dim;
  std::string tok;
  ReadToken(is, binary, &tok);
  if (tok == "<ValueSum>") {
    Vector tmp;
    tmp.Read(is, binary);
    ExpectToken(is, binary, "<DerivSum>");
    tmp.Read(is, binary);
    ExpectToken(is, binary, "<Count>");
    double count;
    ReadBasicType(is, binary, &count);
    ExpectToken(is, binary, end.c_str());
  } else if (tok == "<Counts>") {
    Vector tmp;
    tmp.Read(is, binary);
    ExpectToken(is, binary, end.c_str());
  } else if (tok != end) {
    KALDI_AMD_ERR("Expected " << end << ", got " << tok);
  }
}

static void ReadAffineHead(std::istream &is, bool binary, Component *c) {
  const std::string beg = "<" + c->type + ">";
  ExpectOneOrTwoTokens(is, binary, beg, "<LearningRate>");
  float lr;
  ReadBasicType(is, binary, &lr);
  ExpectToken(is, binary, "<LinearParams>");
  c->linear.Read(is, binary);
  ExpectToken(is, binary, "<BiasParams>");
  c->bias.Read(is, binary);
  c->output_dim = c->linear.NumRows();
  c->input_dim = c->linear.NumCols();
  if (c->bias.Dim() != c->output_dim) KaldiErr("Affine component: bias / linear parameter mismatch");
}

// nnet2/nnet-component.cc:33-43 ReadNew + the Read() of each supported type.
static Component ReadComponent(std::istream &is, bool binary) {
  std::string token;
  ReadToken(is, binary, &token);
  if (token.size() < 3 || token.front() != '<' || token.back() != '>')
    KALDI_AMD_ERR("Unknown component type " << token);
  Component c;
  c.type = token.substr(1, token.size() - 2);
  const std::string end = "</" + c.type + ">";
  std::string tok;
  if (c.type == "SpliceComponent") {  // :2567-2577
    c.kind = AMLIKE_SPLICE;
    ExpectOneOrTwoTokens(is, binary, token, "<InputDim>");
    ReadBasicType(is, binary, &c.input_dim);
    ExpectToken(is, binary, "<LeftContext>");
    ReadBasicType(is, binary, &c.left_context);
    ExpectToken(is, binary, "<RightContext>");
    ReadBasicType(is, binary, &c.right_context);
    ExpectToken(is, binary, "<ConstComponentDim>");
    ReadBasicType(is, binary, &c.const_dim);
    ExpectToken(is, binary, end.c_str());
    c.output_dim = (c.input_dim - c.const_dim) * (1 + c.left_context + c.right_context) + c.const_dim;
  } else if (c.type == "SpliceMaxComponent") {  // :2714-2722
    c.kind = AMLIKE_SPLICE_MAX;
    ExpectOneOrTwoTokens(is, binary, token, "<Dim>");
    ReadBasicType(is, binary, &c.input_dim);
    ExpectToken(is, binary, "<LeftContext>");
    ReadBasicType(is, binary, &c.left_context);
    ExpectToken(is, binary, "<RightContext>");
    ReadBasicType(is, binary, &c.right_context);
    ExpectToken(is, binary, end.c_str());
    c.output_dim = c.input_dim;
  } else if (c.type == "FixedAffineComponent") {  // :3063-3069
    c.kind = AMLIKE_AFFINE;
    ExpectOneOrTwoTokens(is, binary, token, "<LinearParams>");
    c.linear.Read(is, binary);
    ExpectToken(is, binary, "<BiasParams>");
    c.bias.Read(is, binary);
    ExpectToken(is, binary, end.c_str());
    c.output_dim = c.linear.NumRows();
    c.input_dim = c.linear.NumCols();
  } else if (c.type == "AffineComponent") {  // :1162-1192
    c.kind = AMLIKE_AFFINE;
    ReadAffineHead(is, binary, &c);
    ReadToken(is, binary, &tok);
    if (tok == "<AvgInput>") {
      Vector avg;
      avg.Read(is, binary);
      ExpectToken(is, binary, "<AvgInputCount>");
      float cnt;
      ReadBasicType(is, binary, &cnt);
      ReadToken(is, binary, &tok);
    }
    if (tok == "<IsGradient>") {
      bool g;
      ReadBasicType(is, binary, &g);
      ExpectToken(is, binary, end.c_str());
    } else if (tok != end) {
      KALDI_AMD_ERR("Expected " << end << ", got " << tok);
    }
  } else if (c.type == "AffineComponentPreconditioned") {  // :1309-1337
    c.kind = AMLIKE_AFFINE;
    ReadAffineHead(is, binary, &c);
    float f;
    ExpectToken(is, binary, "<Alpha>");
    ReadBasicType(is, binary, &f);
    ReadToken(is, binary, &tok);
    if (tok == "<MaxChange>") {
      ReadBasicType(is, binary, &f);
      ExpectToken(is, binary, end.c_str());
    } else if (tok != end) {
      KALDI_AMD_ERR("Expected " << end << ", got " << tok);
    }
  } else if (c.type == "AffineComponentPreconditionedOnline") {  // :1541-1579
    c.kind = AMLIKE_AFFINE;
    ReadAffineHead(is, binary, &c);
    int32 i;
    float f;
    ReadToken(is, binary, &tok);
    if (tok == "<Rank>") {
      ReadBasicType(is, binary, &i);
    } else if (tok == "<RankIn>") {
      ReadBasicType(is, binary, &i);
      ExpectToken(is, binary, "<RankOut>");
      ReadBasicType(is, binary, &i);
    } else {
      KALDI_AMD_ERR("Expected <Rank> or <RankIn>, got " << tok);
    }
    ReadToken(is, binary, &tok);
    if (tok == "<UpdatePeriod>") {
      ReadBasicType(is, binary, &i);
      ExpectToken(is, binary, "<NumSamplesHistory>");
    } else if (tok != "<NumSamplesHistory>") {
      KALDI_AMD_ERR("Expected <NumSamplesHistory>, got " << tok);
    }
    ReadBasicType(is, binary, &f);
    ExpectToken(is, binary, "<Alpha>");
    ReadBasicType(is, binary, &f);
    ExpectToken(is, binary, "<MaxChangePerSample>");
    ReadBasicType(is, binary, &f);
    ExpectToken(is, binary, end.c_str());
  } else if (c.type == "PnormComponent") {  // :548-556
    c.kind = AMLIKE_PNORM;
    ExpectOneOrTwoTokens(is, binary, token, "<InputDim>");
    ReadBasicType(is, binary, &c.input_dim);
    ExpectToken(is, binary, "<OutputDim>");
    ReadBasicType(is, binary, &c.output_dim);
    ExpectToken(is, binary, "<P>");
    ReadBasicType(is, binary, &c.p);
    ExpectToken(is, binary, end.c_str());
  } else if (c.type == "SumGroupComponent") {  // :2322-2328 (closing token without the slash)
    c.kind = AMLIKE_SUMGROUP;
    ExpectOneOrTwoTokens(is, binary, token, "<Sizes>");
    ReadIntegerVector(is, binary, &c.sizes);
    ExpectToken(is, binary, "<SumGroupComponent>");
    c.output_dim = static_cast<int32>(c.sizes.size());
    c.input_dim = 0;
    for (int32 s : c.sizes) {
      if (s <= 0) KaldiErr("SumGroupComponent: sizes must be positive");
      c.input_dim += s;
    }
  } else if (c.type == "BlockAffineComponent" || c.type == "BlockAffineComponentPreconditioned") {
    // :2045-2055, :2135-2150; Propagate :1907-1939: block b of the output = block b of the
    // input times its [output block x input block] slice of the stacked parameter matrix.
    // Written out as the dense block-diagonal matrix it stands for.
    c.kind = AMLIKE_AFFINE;
    ExpectOneOrTwoTokens(is, binary, token, "<LearningRate>");
    float f;
    ReadBasicType(is, binary, &f);
    ExpectToken(is, binary, "<NumBlocks>");
    int32 num_blocks;
    ReadBasicType(is, binary, &num_blocks);
    Matrix stacked;
    ExpectToken(is, binary, "<LinearParams>");
    stacked.Read(is, binary);
    ExpectToken(is, binary, "<BiasParams>");
    c.bias.Read(is, binary);
    if (c.type == "BlockAffineComponentPreconditioned") {
      ExpectToken(is, binary, "<Alpha>");
      ReadBasicType(is, binary, &f);
      ExpectToken(is, binary, "<IsGradient>");
      bool g;
      ReadBasicType(is, binary, &g);
    }
    ExpectToken(is, binary, end.c_str());
    if (num_blocks <= 0 || stacked.NumRows() % num_blocks != 0 || c.bias.Dim() != stacked.NumRows())
      KaldiErr("BlockAffineComponent: parameters do not match the number of blocks");
    const int32 in_block = stacked.NumCols(), out_block = stacked.NumRows() / num_blocks;
    c.input_dim = in_block * num_blocks;
    c.output_dim = stacked.NumRows();
    c.linear.Resize(c.output_dim, c.input_dim);
    for (int32 r = 0; r < c.output_dim; ++r)
      std::copy(stacked.RowData(r), stacked.RowData(r) + in_block,
                c.linear.RowData(r) + static_cast<size_t>(r / out_block) * in_block);
  } else if (c.type == "FixedLinearComponent") {  // :2979-2983
    c.kind = AMLIKE_AFFINE;
    ExpectOneOrTwoTokens(is, binary, token, "<CuMatrix>");
    c.linear.Read(is, binary);
    ExpectToken(is, binary, end.c_str());
    c.output_dim = c.linear.NumRows();
    c.input_dim = c.linear.NumCols();
  } else if (c.type == "FixedScaleComponent" || c.type == "FixedBiasComponent") {  // :3134-3138, :3201-3205
    const bool scale = c.type == "FixedScaleComponent";
    c.kind = scale ? AMLIKE_FIXED_SCALE : AMLIKE_FIXED_BIAS;
    ExpectOneOrTwoTokens(is, binary, token, scale ? "<Scales>" : "<Bias>");
    c.bias.Read(is, binary);
    ExpectToken(is, binary, end.c_str());
    c.input_dim = c.output_dim = c.bias.Dim();
  } else if (c.type == "ScaleComponent") {  // :891-897
    c.kind = AMLIKE_SCALE;
    ExpectOneOrTwoTokens(is, binary, token, "<Dim>");
    ReadBasicType(is, binary, &c.input_dim);
    ExpectToken(is, binary, "<Scale>");
    ReadBasicType(is, binary, &c.p);
    ExpectToken(is, binary, end.c_str());
    c.output_dim = c.input_dim;
  } else if (c.type == "PowerComponent") {  // :754-762
    c.kind = AMLIKE_POWER;
    ExpectOneOrTwoTokens(is, binary, token, "<InputDim>");
    ReadBasicType(is, binary, &c.input_dim);
    ExpectToken(is, binary, "<OutputDim>");
    ReadBasicType(is, binary, &c.output_dim);
    ExpectToken(is, binary, "<Power>");
    ReadBasicType(is, binary, &c.p);
    ExpectToken(is, binary, end.c_str());
  } else if (c.type == "MaxoutComponent") {  // :481-487
    c.kind = AMLIKE_MAXOUT;
    ExpectOneOrTwoTokens(is, binary, token, "<InputDim>");
    ReadBasicType(is, binary, &c.input_dim);
    ExpectToken(is, binary, "<OutputDim>");
    ReadBasicType(is, binary, &c.output_dim);
    ExpectToken(is, binary, end.c_str());
  } else if (c.type == "PermuteComponent") {  // :2232-2236
    c.kind = AMLIKE_PERMUTE;
    ExpectOneOrTwoTokens(is, binary, token, "<Reorder>");
    ReadIntegerVector(is, binary, &c.sizes);
    ExpectToken(is, binary, end.c_str());
    c.input_dim = c.output_dim = static_cast<int32>(c.sizes.size());
  } else if (c.type == "DctComponent") {  // :2892-2918
    c.kind = AMLIKE_DCT;
    ExpectOneOrTwoTokens(is, binary, token, "<Dim>");
    ReadBasicType(is, binary, &c.input_dim);
    ExpectToken(is, binary, "<DctDim>");
    ReadBasicType(is, binary, &c.const_dim);
    ExpectToken(is, binary, "<Reorder>");
    ReadBasicType(is, binary, &c.dct_reorder);
    int32 keep = c.const_dim;
    ReadToken(is, binary, &tok);
    if (tok == "<DctKeepDim>") {
      ReadBasicType(is, binary, &keep);
      ExpectToken(is, binary, end.c_str());
    } else if (tok != end) {
      KALDI_AMD_ERR("Expected token \"" << end << "\", got instead \"" << tok << "\".");
    }
    if (keep <= 0) keep = c.const_dim;  // DctComponent::Init :2745
    if (c.const_dim <= 0 || c.input_dim <= 0 || c.input_dim % c.const_dim != 0 || keep > c.const_dim)
      KaldiErr("DctComponent: the DCT dimension must divide the dimension and cover the kept part");
    c.output_dim = keep * (c.input_dim / c.const_dim);
  } else {
    if (c.type == "NormalizeComponent") c.kind = AMLIKE_NORMALIZE;
    else if (c.type == "SoftmaxComponent") c.kind = AMLIKE_SOFTMAX;
    else if (c.type == "SigmoidComponent") c.kind = AMLIKE_SIGMOID;
    else if (c.type == "TanhComponent") c.kind = AMLIKE_TANH;
    else if (c.type == "RectifiedLinearComponent") c.kind = AMLIKE_RELU;
    else if (c.type == "SoftHingeComponent") c.kind = AMLIKE_SOFTHINGE;
    else KALDI_AMD_ERR("Unknown component type " << c.type);
    ReadNonlinear(is, binary, &c);
  }
  return c;
}

void Nnet::Read(std::istream &is, bool binary) {  // nnet2/nnet-nnet.cc:89-103
  components_.clear();
  ExpectToken(is, binary, "<Nnet>");
  int32 num_components;
  ExpectToken(is, binary, "<NumComponents>");
  ReadBasicType(is, binary, &num_components);
  ExpectToken(is, binary, "<Components>");
  if (num_components <= 0) KaldiErr("Nnet::Read, no components");
  for (int32 c = 0; c < num_components; ++c) components_.push_back(ReadComponent(is, binary));
  ExpectToken(is, binary, "</Components>");
  ExpectToken(is, binary, "</Nnet>");
  for (int32 c = 0; c + 1 < num_components; ++c)  // Nnet::Check
    if (components_[c].output_dim != components_[c + 1].input_dim)
      KALDI_AMD_ERR("Dimension mismatch between components " << c << " and " << c + 1);
}

int32 Nnet::LeftContext() const {
  int32 ans = 0;
  for (const Component &c : components_) ans += c.left_context;
  return ans;
}
int32 Nnet::RightContext() const {
  int32 ans = 0;
  for (const Component &c : components_) ans += c.right_context;
  return ans;
}

void AmNnet::Read(std::istream &is, bool binary) {  // nnet2/am-nnet.cc:39-42
  nnet_.Read(is, binary);
  priors_.Read(is, binary);
  engine_.reset();
}

amlike_nnet_t *AmNnet::Engine() const {
  if (!engine_) engine_.reset(CreateEngine());
  return engine_.get();
}

amlike_nnet_t *AmNnet::CreateEngine() const {
  {
    const int32 n = nnet_.NumComponents();
    std::vector<amlike_component_t> comps(n);
    std::vector<std::vector<float>> dense(n);
    for (int32 i = 0; i < n; ++i) {
      const Component &c = nnet_.GetComponent(i);
      amlike_component_t &a = comps[i];
      a.kind = c.kind;
      a.input_dim = c.input_dim;
      a.output_dim = c.output_dim;
      a.left_context = c.left_context;
      a.right_context = c.right_context;
      a.const_dim = c.const_dim;
      a.p = c.p;
      a.linear = nullptr;
      a.bias = nullptr;
      a.sizes = nullptr;
      if (c.kind == AMLIKE_AFFINE) {
        dense[i].resize(static_cast<size_t>(c.output_dim) * c.input_dim);
        for (int32 r = 0; r < c.output_dim; ++r)
          std::copy(c.linear.RowData(r), c.linear.RowData(r) + c.input_dim,
                    dense[i].begin() + static_cast<size_t>(r) * c.input_dim);
        a.linear = dense[i].data();
        a.bias = c.bias.Dim() ? c.bias.Data() : nullptr;  // FixedLinearComponent has none
      } else if (c.kind == AMLIKE_SUMGROUP || c.kind == AMLIKE_PERMUTE) {
        a.sizes = c.sizes.data();
      } else if (c.kind == AMLIKE_FIXED_SCALE || c.kind == AMLIKE_FIXED_BIAS) {
        a.bias = c.bias.Data();
      } else if (c.kind == AMLIKE_DCT) {
        a.left_context = c.dct_reorder ? 1 : 0;
      }
    }
    amlike_nnet_t *h = nullptr;
    CheckAmlike(amlike_nnet_create(comps.data(), n, priors_.Dim() ? priors_.Data() : nullptr, priors_.Dim(), &h));
    return h;
  }
}

void NnetComputation(const AmNnet &am, const Matrix &feats, bool pad_input, int mode,
                     BaseFloat prob_scale, Matrix *output) {
  const Nnet &nnet = am.GetNnet();
  if (feats.NumCols() != nnet.InputDim())
    KALDI_AMD_ERR("Feature dimension is " << feats.NumCols() << " but network expects " << nnet.InputDim());
  int32 rows = feats.NumRows();
  if (!pad_input) rows -= nnet.LeftContext() + nnet.RightContext();
  if (rows <= 0) KaldiErr("NnetComputation: output would be empty");
  output->Resize(rows, nnet.OutputDim());
  CheckAmlike(amlike_nnet_compute(am.Engine(), feats.Data(), feats.NumRows(), feats.Stride(), pad_input ? 1 : 0,
                                  mode, prob_scale, output->Data(), output->Stride()));
}

void NnetComputation(const AmNnet &am, const Matrix &feats, bool pad_input, Matrix *output) {
  NnetComputation(am, feats, pad_input, AMLIKE_NNET_POSTERIORS, 1.0f, output);
}

DecodableAmNnet::DecodableAmNnet(const TransitionModel &trans_model, const AmNnet &am_nnet,
                                 const Matrix &feats, bool pad_input, BaseFloat prob_scale)
    : trans_model_(trans_model) {
  const Nnet &nnet = am_nnet.GetNnet();
  const int32 rows = feats.NumRows() - (pad_input ? 0 : nnet.LeftContext() + nnet.RightContext());
  if (rows <= 0) return;  // the reference warns and leaves the object empty
  if (am_nnet.Priors().Dim() != trans_model.NumPdfs()) KaldiErr("Priors in neural network not set up.");
  NnetComputation(am_nnet, feats, pad_input, AMLIKE_NNET_LOGLIKES, prob_scale, &log_probs_);
}

// nnet2/online-nnet2-decodable.cc:25-45.
DecodableNnet2Online::DecodableNnet2Online(const AmNnet &nnet, const TransitionModel &trans_model,
                                           const DecodableNnet2OnlineOptions &opts,
                                           OnlineFeatureInterface *input_feats)
    : features_(input_feats), nnet_(nnet), trans_model_(trans_model), opts_(opts),
      feat_dim_(input_feats->Dim()), left_context_(nnet.GetNnet().LeftContext()),
      right_context_(nnet.GetNnet().RightContext()), num_pdfs_(nnet.GetNnet().OutputDim()) {
  if (opts_.max_nnet_batch_size <= 0) KaldiErr("DecodableNnet2Online: max_nnet_batch_size must be positive");
  if (nnet.Priors().Dim() != trans_model.NumPdfs())
    KaldiErr("Priors in neural network not set up (or mismatch with transition model).");
  if (feat_dim_ != nnet.GetNnet().InputDim())
    KALDI_AMD_ERR("Feature dimension is " << feat_dim_ << " but network expects "
                  << nnet.GetNnet().InputDim());
}

// :49-55
BaseFloat DecodableNnet2Online::LogLikelihood(int32 frame, int32 index) {
  ComputeForFrame(frame);
  const int32 pdf_id = trans_model_.TransitionIdToPdf(index);
  return scaled_loglikes_(frame - begin_frame_, pdf_id);
}

// :58-64
bool DecodableNnet2Online::IsLastFrame(int32 frame) const {
  if (opts_.pad_input) return features_->IsLastFrame(frame);
  return features_->IsLastFrame(frame + left_context_ + right_context_);
}

// :66-80
int32 DecodableNnet2Online::NumFramesReady() const {
  const int32 features_ready = features_->NumFramesReady();
  if (features_ready == 0) return 0;
  const bool input_finished = features_->IsLastFrame(features_ready - 1);
  if (opts_.pad_input) {
    if (input_finished) return features_ready;
    return std::max<int32>(0, features_ready - right_context_);
  }
  return std::max<int32>(0, features_ready - right_context_ - left_context_);
}

// :82-142.  The batch of input frames is gathered on the host exactly as the reference does
// (edge frames repeated when pad_input), then one GPU call computes the network on it with
// no further padding and applies floor, log, minus log-prior and the acoustic scale.
void DecodableNnet2Online::ComputeForFrame(int32 frame) {
  const int32 features_ready = features_->NumFramesReady();
  const bool input_finished = features_->IsLastFrame(features_ready - 1);
  if (frame < 0) KaldiErr("DecodableNnet2Online: negative frame");
  if (frame >= begin_frame_ && frame < begin_frame_ + scaled_loglikes_.NumRows()) return;
  if (frame >= NumFramesReady()) KALDI_AMD_ERR("DecodableNnet2Online: frame " << frame << " is not ready");
  const int32 input_frame_begin = opts_.pad_input ? frame - left_context_ : frame;
  int32 max_possible_input_frame_end = features_ready;
  if (input_finished && opts_.pad_input) max_possible_input_frame_end += right_context_;
  const int32 input_frame_end =
      std::min<int32>(max_possible_input_frame_end,
                      input_frame_begin + left_context_ + right_context_ + opts_.max_nnet_batch_size);
  Matrix features(input_frame_end - input_frame_begin, feat_dim_);
  Vector row(feat_dim_);
  for (int32 t = input_frame_begin; t < input_frame_end; t++) {
    int32 t_modified = t;
    if (t_modified < 0) t_modified = 0;
    if (t_modified >= features_ready) t_modified = features_ready - 1;
    features_->GetFrame(t_modified, &row);
    std::copy(row.Data(), row.Data() + feat_dim_, features.RowData(t - input_frame_begin));
  }
  NnetComputation(nnet_, features, false, AMLIKE_NNET_LOGLIKES, opts_.acoustic_scale, &scaled_loglikes_);
  begin_frame_ = frame;
  ++num_batches_;
}

}  // namespace nnet2

void OnlineMatrixFeature::GetFrame(int32 frame, Vector *feat) {
  if (frame < 0 || frame >= ready_) KALDI_AMD_ERR("OnlineMatrixFeature: frame " << frame << " has not arrived");
  feat->Resize(mat_.NumCols());
  std::copy(mat_.RowData(frame), mat_.RowData(frame) + mat_.NumCols(), feat->Data());
}

}  // namespace kaldi_amd
