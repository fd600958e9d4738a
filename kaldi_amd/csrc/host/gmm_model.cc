// Host-side GMM model classes (see gmm_model.h).
#include "gmm_model.h"

#include <algorithm>
#include <cmath>

namespace kaldi_amd {

void CheckAmlike(int rc) {
  if (rc != AMLIKE_OK) KaldiErr(amlike_last_error());
}

// ---- TransitionModel ------------------------------------------------------------
void TransitionModel::ReadTopology(std::istream &is, bool binary) {
  ExpectToken(is, binary, "<Topology>");
  phones_.clear();
  phone2idx_.clear();
  entries_.clear();
  if (!binary) {  // hmm/hmm-topology.cc:41-114
    std::string token;
    while (!(is >> token).fail()) {
      if (token == "</Topology>") break;
      if (token != "<TopologyEntry>")
        KALDI_AMD_ERR("Reading HmmTopology object, expected </Topology> or <TopologyEntry>, got " << token);
      ExpectToken(is, binary, "<ForPhones>");
      std::vector<int32> phones;
      std::string s;
      while (true) {
        is >> s;
        if (is.fail()) KaldiErr("Reading HmmTopology object, unexpected end of file while expecting phones.");
        if (s == "</ForPhones>") break;
        char *end = nullptr;
        const long p = strtol(s.c_str(), &end, 10);
        if (*end != '\0') KALDI_AMD_ERR("Reading HmmTopology object, expected integer, got instead " << s);
        phones.push_back(static_cast<int32>(p));
      }
      std::vector<HmmState> this_entry;
      ReadToken(is, binary, &token);
      while (token != "</TopologyEntry>") {
        if (token != "<State>") KALDI_AMD_ERR("Expected </TopologyEntry> or <State>, got instead " << token);
        int32 state;
        ReadBasicType(is, binary, &state);
        if (state != static_cast<int32>(this_entry.size()))
          KALDI_AMD_ERR("States are expected to be in order from zero, expected " << this_entry.size()
                                                                                   << ", got " << state);
        ReadToken(is, binary, &token);
        int32 pdf_class = -1;
        if (token == "<PdfClass>") {
          ReadBasicType(is, binary, &pdf_class);
          ReadToken(is, binary, &token);
        }
        this_entry.push_back(HmmState{pdf_class, {}});
        while (token == "<Transition>") {
          int32 dst;
          float prob;
          ReadBasicType(is, binary, &dst);
          ReadBasicType(is, binary, &prob);
          this_entry.back().transitions.push_back(std::make_pair(dst, prob));
          ReadToken(is, binary, &token);
        }
        if (token != "</State>") KALDI_AMD_ERR("Reading HmmTopology,  unexpected token " << token);
        ReadToken(is, binary, &token);
      }
      const int32 my_index = static_cast<int32>(entries_.size());
      entries_.push_back(this_entry);
      for (int32 phone : phones) {
        if (static_cast<int32>(phone2idx_.size()) <= phone) phone2idx_.resize(phone + 1, -1);
        if (phone <= 0 || phone2idx_[phone] != -1)
          KaldiErr("Phone appears in multiple topology entries (or is not positive).");
        phone2idx_[phone] = my_index;
        phones_.push_back(phone);
      }
    }
    std::sort(phones_.begin(), phones_.end());
  } else {  // hmm/hmm-topology.cc:115-137
    ReadIntegerVector(is, binary, &phones_);
    ReadIntegerVector(is, binary, &phone2idx_);
    int32 sz;
    ReadBasicType(is, binary, &sz);
    if (sz < 0) KaldiErr("Reading HmmTopology: negative size");
    entries_.resize(sz);
    for (int32 i = 0; i < sz; ++i) {
      int32 nstates;
      ReadBasicType(is, binary, &nstates);
      entries_[i].resize(nstates);
      for (int32 j = 0; j < nstates; ++j) {
        ReadBasicType(is, binary, &entries_[i][j].pdf_class);
        int32 ntrans;
        ReadBasicType(is, binary, &ntrans);
        entries_[i][j].transitions.resize(ntrans);
        for (int32 k = 0; k < ntrans; ++k) {
          ReadBasicType(is, binary, &entries_[i][j].transitions[k].first);
          ReadBasicType(is, binary, &entries_[i][j].transitions[k].second);
        }
      }
    }
    ExpectToken(is, binary, "</Topology>");
  }
  if (entries_.empty() || phones_.empty() || phone2idx_.empty()) KaldiErr("HmmTopology::Check(), empty object.");
  for (int32 phone : phones_)
    if (static_cast<size_t>(phone) >= phone2idx_.size() ||
        static_cast<size_t>(phone2idx_[phone]) >= entries_.size())
      KaldiErr("HmmTopology::Check(), phone has no valid index.");
}

const std::vector<TransitionModel::HmmState> &TransitionModel::TopologyForPhone(int32 phone) const {
  // hmm/hmm-topology.cc:273-281
  if (phone < 0 || static_cast<size_t>(phone) >= phone2idx_.size() || phone2idx_[phone] == -1)
    KALDI_AMD_ERR("TopologyForPhone(), phone " << phone << " not covered.");
  return entries_[phone2idx_[phone]];
}

int32 TransitionModel::NumPdfClasses(int32 phone) const {  // hmm/hmm-topology.cc:283-292
  int32 max_pdf_class = 0;
  for (const HmmState &st : TopologyForPhone(phone)) max_pdf_class = std::max(max_pdf_class, st.pdf_class);
  return max_pdf_class + 1;
}

void TransitionModel::ComputeDerived() {  // hmm/transition-model.cc:72-97
  state2id_.assign(triples_.size() + 2, 0);
  int32 cur = 1;
  num_pdfs_ = 0;
  for (int32 tstate = 1; tstate <= static_cast<int32>(triples_.size()) + 1; ++tstate) {
    state2id_[tstate] = cur;
    if (static_cast<size_t>(tstate) <= triples_.size()) {
      const Triple &t = triples_[tstate - 1];
      num_pdfs_ = std::max(num_pdfs_, 1 + t.pdf);
      if (t.phone < 0 || static_cast<size_t>(t.phone) >= phone2idx_.size() || phone2idx_[t.phone] < 0)
        KaldiErr("TransitionModel: triple refers to a phone with no topology");
      const std::vector<HmmState> &entry = entries_[phone2idx_[t.phone]];
      if (t.hmm_state < 0 || static_cast<size_t>(t.hmm_state) >= entry.size())
        KaldiErr("TransitionModel: triple refers to a state outside its topology entry");
      cur += static_cast<int32>(entry[t.hmm_state].transitions.size());
    }
  }
  id2state_.assign(cur, 0);
  tid2pdf_.assign(cur, -1);
  for (int32 tstate = 1; tstate <= static_cast<int32>(triples_.size()); ++tstate)
    for (int32 tid = state2id_[tstate]; tid < state2id_[tstate + 1]; ++tid) {
      id2state_[tid] = tstate;
      tid2pdf_[tid] = triples_[tstate - 1].pdf;
    }
}

void TransitionModel::Read(std::istream &is, bool binary) {  // hmm/transition-model.cc:274-294
  ExpectToken(is, binary, "<TransitionModel>");
  ReadTopology(is, binary);
  ExpectToken(is, binary, "<Triples>");
  int32 size;
  ReadBasicType(is, binary, &size);
  if (size < 0) KaldiErr("TransitionModel::Read, negative number of triples");
  triples_.resize(size);
  for (int32 i = 0; i < size; ++i) {
    ReadBasicType(is, binary, &triples_[i].phone);
    ReadBasicType(is, binary, &triples_[i].hmm_state);
    ReadBasicType(is, binary, &triples_[i].pdf);
  }
  ExpectToken(is, binary, "</Triples>");
  ComputeDerived();
  ExpectToken(is, binary, "<LogProbs>");
  log_probs_.Read(is, binary);
  ExpectToken(is, binary, "</LogProbs>");
  ExpectToken(is, binary, "</TransitionModel>");
  if (log_probs_.Dim() != NumTransitionIds() + 1)
    KaldiErr("TransitionModel::Read, log-probs do not match the transition-ids");
}

int32 TransitionModel::TransitionIdToPdf(int32 trans_id) const {
  if (trans_id <= 0 || static_cast<size_t>(trans_id) >= tid2pdf_.size())
    KaldiErr("Likely graph/model mismatch (transition-id out of range)");
  return tid2pdf_[trans_id];
}

// ---- DiagGmm -----------------------------------------------------------------------
void DiagGmm::Read(std::istream &is, bool binary) {  // gmm/diag-gmm.cc:728-756
  std::string token;
  ReadToken(is, binary, &token);
  if (token != "<DiagGMMBegin>" && token != "<DiagGMM>") KALDI_AMD_ERR("Expected <DiagGMM>, got " << token);
  ReadToken(is, binary, &token);
  if (token == "<GCONSTS>") {
    gconsts_.Read(is, binary);
    ExpectToken(is, binary, "<WEIGHTS>");
  } else if (token != "<WEIGHTS>") {
    KALDI_AMD_ERR("DiagGmm::Read, expected <WEIGHTS> or <GCONSTS>, got " << token);
  }
  weights_.Read(is, binary);
  ExpectToken(is, binary, "<MEANS_INVVARS>");
  means_invvars_.Read(is, binary);
  ExpectToken(is, binary, "<INV_VARS>");
  inv_vars_.Read(is, binary);
  ReadToken(is, binary, &token);
  if (token != "<DiagGMMEnd>" && token != "</DiagGMM>") KALDI_AMD_ERR("Expected </DiagGMM>, got " << token);
  if (means_invvars_.NumRows() != weights_.Dim() || inv_vars_.NumRows() != weights_.Dim() ||
      inv_vars_.NumCols() != means_invvars_.NumCols())
    KaldiErr("DiagGmm::Read, inconsistent dimensions");
  ComputeGconsts();  // the values read from the file are not trusted
  engine_.reset();
}

void DiagGmm::Write(std::ostream &os, bool binary) const {  // gmm/diag-gmm.cc:705-720
  if (!valid_gconsts_) KaldiErr("Must call ComputeGconsts() before writing the model.");
  WriteToken(os, binary, "<DiagGMM>");
  if (!binary) os << "\n";
  WriteToken(os, binary, "<GCONSTS>");
  gconsts_.Write(os, binary);
  WriteToken(os, binary, "<WEIGHTS>");
  weights_.Write(os, binary);
  WriteToken(os, binary, "<MEANS_INVVARS>");
  means_invvars_.Write(os, binary);
  WriteToken(os, binary, "<INV_VARS>");
  inv_vars_.Write(os, binary);
  WriteToken(os, binary, "</DiagGMM>");
  if (!binary) os << "\n";
}

void DiagGmm::SetParams(const Vector &weights, const Matrix &means_invvars, const Matrix &inv_vars) {
  weights_ = weights;
  means_invvars_ = means_invvars;
  inv_vars_ = inv_vars;
  ComputeGconsts();
  engine_.reset();
}

// Dense [M x D] copy of a (stride-padded) Matrix.
static std::vector<float> Dense(const Matrix &m) {
  std::vector<float> out(static_cast<size_t>(m.NumRows()) * m.NumCols());
  for (int32 r = 0; r < m.NumRows(); ++r)
    std::copy(m.RowData(r), m.RowData(r) + m.NumCols(), out.begin() + static_cast<size_t>(r) * m.NumCols());
  return out;
}

int32 DiagGmm::ComputeGconsts() {  // gmm/diag-gmm.cc:114-152, arithmetic in amlike_gmm_compute_gconsts
  const int32 M = NumGauss(), D = Dim();
  gconsts_.Resize(M);
  int32 bad = 0;
  if (M > 0) {
    const std::vector<float> miv = Dense(means_invvars_), iv = Dense(inv_vars_);
    for (int32 m = 0; m < M; ++m)
      if (weights_(m) < 0) KaldiErr("Cannot have negative weights.");
    CheckAmlike(amlike_gmm_compute_gconsts(weights_.Data(), miv.data(), iv.data(), M, D, gconsts_.Data(), &bad));
  }
  valid_gconsts_ = true;
  return bad;
}

amlike_gmm_t *DiagGmm::Engine() const {
  if (!engine_) {
    if (!valid_gconsts_) KaldiErr("Must call ComputeGconsts() before computing likelihood");
    const std::vector<float> miv = Dense(means_invvars_), iv = Dense(inv_vars_);
    const int32 offs[2] = {0, NumGauss()};
    amlike_gmm_t *g = nullptr;
    CheckAmlike(amlike_gmm_create(gconsts_.Data(), miv.data(), iv.data(), offs, 1, NumGauss(), Dim(), &g));
    engine_.reset(g);
  }
  return engine_.get();
}

BaseFloat DiagGmm::LogLikelihood(const Vector &data) const {
  if (!valid_gconsts_) KaldiErr("Must call ComputeGconsts() before computing likelihood");
  if (data.Dim() != Dim())
    KALDI_AMD_ERR("DiagGmm::ComponentLogLikelihood, dimension mismatch " << data.Dim() << " vs. " << Dim());
  float out = 0.0f;
  CheckAmlike(amlike_gmm_score(Engine(), data.Data(), 1, data.Dim(), data.Dim(), -1.0f, &out, 1));
  return out;
}

void DiagGmm::LogLikelihoods(const Vector &data, Vector *loglikes) const {
  if (data.Dim() != Dim())
    KALDI_AMD_ERR("DiagGmm::ComponentLogLikelihood, dimension mismatch " << data.Dim() << " vs. " << Dim());
  loglikes->Resize(NumGauss());
  CheckAmlike(amlike_gmm_gauss_loglikes(Engine(), 0, 1, data.Data(), 1, data.Dim(), data.Dim(),
                                        loglikes->Data(), NumGauss()));
}

void DiagGmm::LogLikelihoods(const Matrix &data, Matrix *loglikes) const {
  if (data.NumRows() == 0) KaldiErr("DiagGmm::LogLikelihoods: empty data");
  if (data.NumCols() != Dim())
    KALDI_AMD_ERR("DiagGmm::ComponentLogLikelihood, dimension mismatch " << data.NumCols() << " vs. " << Dim());
  loglikes->Resize(data.NumRows(), NumGauss());
  CheckAmlike(amlike_gmm_gauss_loglikes(Engine(), 0, 1, data.Data(), data.NumRows(), data.Stride(),
                                        data.NumCols(), loglikes->Data(), loglikes->Stride()));
}

BaseFloat DiagGmm::GaussianSelection(const Matrix &data, int32 num_gselect,
                                     std::vector<std::vector<int32>> *output) const {
  if (data.NumRows() == 0) KaldiErr("DiagGmm::GaussianSelection: empty data");
  if (num_gselect <= 0) KaldiErr("DiagGmm::GaussianSelection: num_gselect must be positive");
  std::vector<int32> idx(static_cast<size_t>(data.NumRows()) * num_gselect);
  double tot = 0.0;
  CheckAmlike(amlike_gmm_gselect(Engine(), 0, data.Data(), data.NumRows(), data.Stride(), data.NumCols(),
                                 num_gselect, idx.data(), &tot));
  output->assign(data.NumRows(), std::vector<int32>());
  for (int32 t = 0; t < data.NumRows(); ++t)
    for (int32 j = 0; j < num_gselect; ++j) {
      const int32 g = idx[static_cast<size_t>(t) * num_gselect + j];
      if (g >= 0) (*output)[t].push_back(g);
    }
  return static_cast<BaseFloat>(tot);
}

// ---- AmDiagGmm ---------------------------------------------------------------------
void AmDiagGmm::Read(std::istream &is, bool binary) {  // gmm/am-diag-gmm.cc:147-161
  int32 num_pdfs, dim;
  ExpectToken(is, binary, "<DIMENSION>");
  ReadBasicType(is, binary, &dim);
  ExpectToken(is, binary, "<NUMPDFS>");
  ReadBasicType(is, binary, &num_pdfs);
  if (num_pdfs <= 0) KaldiErr("AmDiagGmm::Read, num_pdfs must be positive");
  densities_.clear();
  densities_.resize(num_pdfs);
  for (int32 i = 0; i < num_pdfs; ++i) {
    densities_[i].Read(is, binary);
    if (densities_[i].Dim() != dim) KaldiErr("AmDiagGmm::Read, dimension mismatch between pdfs");
  }
  engine_.reset();
}

void AmDiagGmm::Write(std::ostream &os, bool binary) const {  // gmm/am-diag-gmm.cc:163-176
  WriteToken(os, binary, "<DIMENSION>");
  WriteBasicType(os, binary, Dim());
  WriteToken(os, binary, "<NUMPDFS>");
  WriteBasicType(os, binary, NumPdfs());
  for (const DiagGmm &g : densities_) g.Write(os, binary);
}

int32 AmDiagGmm::NumGauss() const {
  int32 n = 0;
  for (const DiagGmm &g : densities_) n += g.NumGauss();
  return n;
}

amlike_gmm_t *AmDiagGmm::Engine() const {
  if (!engine_) engine_.reset(CreateEngine());
  return engine_.get();
}

amlike_gmm_t *AmDiagGmm::CreateEngine() const {
  {
    const int32 P = NumPdfs(), D = Dim(), G = NumGauss();
    if (P == 0) KaldiErr("AmDiagGmm: empty model");
    std::vector<float> gc(G), miv(static_cast<size_t>(G) * D), iv(static_cast<size_t>(G) * D);
    std::vector<int32> offs(P + 1, 0);
    int32 g0 = 0;
    for (int32 j = 0; j < P; ++j) {
      const DiagGmm &d = densities_[j];
      if (!d.valid_gconsts()) KALDI_AMD_ERR("State " << j << ": Must call ComputeGconsts() before computing likelihood.");
      for (int32 m = 0; m < d.NumGauss(); ++m) {
        gc[g0 + m] = d.gconsts()(m);
        std::copy(d.means_invvars().RowData(m), d.means_invvars().RowData(m) + D, miv.begin() + static_cast<size_t>(g0 + m) * D);
        std::copy(d.inv_vars().RowData(m), d.inv_vars().RowData(m) + D, iv.begin() + static_cast<size_t>(g0 + m) * D);
      }
      g0 += d.NumGauss();
      offs[j + 1] = g0;
    }
    amlike_gmm_t *g = nullptr;
    CheckAmlike(amlike_gmm_create(gc.data(), miv.data(), iv.data(), offs.data(), P, G, D, &g));
    return g;
  }
}

void AmDiagGmm::LogLikelihoods(const Matrix &feats, Matrix *loglikes, BaseFloat prune) const {
  if (feats.NumCols() != Dim())
    KALDI_AMD_ERR("Dim mismatch: data dim = " << feats.NumCols() << " vs. model dim = " << Dim());
  loglikes->Resize(feats.NumRows(), NumPdfs());
  if (feats.NumRows() == 0) return;
  CheckAmlike(amlike_gmm_score(Engine(), feats.Data(), feats.NumRows(), feats.Stride(), feats.NumCols(),
                               prune, loglikes->Data(), loglikes->Stride()));
}

BaseFloat AmDiagGmm::LogLikelihood(int32 pdf_index, const Vector &data) const {
  if (pdf_index < 0 || pdf_index >= NumPdfs()) KaldiErr("AmDiagGmm::LogLikelihood: pdf index out of range");
  return densities_[pdf_index].LogLikelihood(data);
}

// ---- Decodables ----------------------------------------------------------------------
DecodableAmDiagGmmUnmapped::DecodableAmDiagGmmUnmapped(const AmDiagGmm &am, const Matrix &feats,
                                                       BaseFloat log_sum_exp_prune) {
  am.LogLikelihoods(feats, &loglikes_, log_sum_exp_prune);
}

bool DecodableAmDiagGmmUnmapped::IsLastFrame(int32 frame) const {
  if (frame >= NumFramesReady()) KaldiErr("IsLastFrame: frame out of range");
  return frame == NumFramesReady() - 1;
}

BaseFloat DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased(int32 frame, int32 state) const {
  if (static_cast<size_t>(frame) >= static_cast<size_t>(loglikes_.NumRows()))
    KaldiErr("LogLikelihood: frame out of range");
  if (static_cast<size_t>(state) >= static_cast<size_t>(loglikes_.NumCols()))
    KaldiErr("Likely graph/model mismatch, e.g. using wrong HCLG.fst");
  return loglikes_(frame, state);
}

}  // namespace kaldi_amd
