/* oracle/gmm_oracle.c
 *
 * TEST INFRASTRUCTURE ONLY.  A plain-C, single-threaded CPU restatement of the
 * reference's diagonal-GMM log-likelihood path.  It is the checker that tests,
 * smoke() and bench.py's cpu_baseline leg compare the HIP path against; the
 * product never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_cpu.py checks every function
 * here against the real reference library (oracle/_ref/ref_driver, built from
 * /root/reference by oracle/build_ref.sh) and against the golden fixtures in
 * tests/golden/ that the same driver produced.
 *
 * Every function names the reference lines it follows (paths relative to
 * /root/reference/src).  Loop order and precisions follow the reference:
 * fp32 data, fp32 dot products, expf, log-sum-exp summed in double.
 */
#include <float.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <stddef.h>
#include <stdint.h>
#include <time.h>

#define ORACLE_OK 0
#define ORACLE_ENONFINITE 3 /* "Invalid answer (overflow or invalid variances/features?)" */
#define ORACLE_ENAN_GCONST 4

/* base/kaldi-math.h:137-138  kMinLogDiffFloat = log(FLT_EPSILON) computed in float. */
static float min_log_diff_float(void) { return logf(FLT_EPSILON); }

/* gmm/diag-gmm.cc:114-152  DiagGmm::ComputeGconsts.
 * gc is a float that is widened to double for each += and narrowed again;
 * log() is the double function (checked bit-for-bit against the reference build:
 * log(weight) + offset is summed in double before the first narrowing).
 * Returns the number of +-inf gconsts (mapped to -inf), or -1 on NaN. */
int oracle_compute_gconsts(const float *weights, const float *means_invvars,
                           const float *inv_vars, int num_mix, int dim,
                           float *gconsts) {
  const double M_LOG_2PI_ = 1.8378770664093454835606594728112; /* base/kaldi-math.h:48 */
  float offset = (float)(-0.5 * M_LOG_2PI_ * dim);
  int num_bad = 0;
  for (int mix = 0; mix < num_mix; mix++) {
    float gc = (float)(log((double)weights[mix]) + (double)offset);
    for (int d = 0; d < dim; d++) {
      float iv = inv_vars[(size_t)mix * dim + d], miv = means_invvars[(size_t)mix * dim + d];
      /* both terms in double; the second is ((0.5 * miv) * miv) / iv. */
      double term = 0.5 * log((double)iv) - 0.5 * (double)miv * (double)miv / (double)iv;
      gc = (float)((double)gc + term);
    }
    if (isnan(gc)) return -1;
    if (isinf(gc)) {
      num_bad++;
      if (gc > 0) gc = -gc;
    }
    gconsts[mix] = gc;
  }
  return num_bad;
}

/* matrix/kaldi-vector.cc:766-784  VectorBase<float>::LogSumExp(prune). */
float oracle_log_sum_exp(const float *v, int n, float prune) {
  float max_elem = v[0];
  for (int i = 1; i < n; i++)
    if (v[i] > max_elem) max_elem = v[i];
  float cutoff = max_elem + min_log_diff_float();
  if (prune > 0.0f && max_elem - prune > cutoff) cutoff = max_elem - prune;
  double sum = 0.0;
  for (int i = 0; i < n; i++) {
    float f = v[i];
    if (f >= cutoff) sum += expf(f - max_elem); /* base/kaldi-math.h:330-340 Exp(float)=expf */
  }
  return (float)(max_elem + log(sum));
}

/* gmm/diag-gmm.cc:528-543  DiagGmm::LogLikelihoods(Vector): per-Gaussian scores
 *   ll = gconsts;  ll += means_invvars * x;  ll += -0.5 * inv_vars * x^2
 * as two fp32 matrix-vector products (BLAS sgemv in the reference; sequential
 * fp32 dot here), each added onto ll in turn. */
void oracle_diag_gmm_loglikes_vec(const float *gconsts, const float *means_invvars,
                                  const float *inv_vars, int num_mix, int dim,
                                  const float *x, const float *x_sq, float *ll) {
  for (int m = 0; m < num_mix; m++) {
    const float *miv = means_invvars + (size_t)m * dim, *iv = inv_vars + (size_t)m * dim;
    float dot1 = 0.0f, dot2 = 0.0f;
    for (int d = 0; d < dim; d++) dot1 += miv[d] * x[d];
    for (int d = 0; d < dim; d++) dot2 += iv[d] * x_sq[d];
    float v = gconsts[m];
    v = 1.0f * dot1 + v;
    v = -0.5f * dot2 + v;
    ll[m] = v;
  }
}

/* gmmbin/gmm-compute-likes.cc:68-80 over a packed acoustic model:
 * pdf j owns Gaussians [pdf_offsets[j], pdf_offsets[j+1]).
 * Each (frame,pdf): gmm/diag-gmm.cc:517-526 DiagGmm::LogLikelihood
 * (= LogLikelihoods then LogSumExp(), error on NaN/Inf), or, with prune > 0,
 * gmm/decodable-am-diag-gmm.cc:58-66 (LogSumExp(prune)).
 * x^2 is x*x in fp32 (matrix/kaldi-vector.cc:461-473, non-MKL branch).
 * scratch: at least max_j M_j + dim floats.  Returns ORACLE_OK or
 * ORACLE_ENONFINITE (the output is still filled). */
int oracle_gmm_compute_likes(const float *gconsts, const float *means_invvars,
                             const float *inv_vars, const int32_t *pdf_offsets,
                             int num_pdfs, int dim, const float *feats, int T, int ld,
                             float prune, float *out, int ldo, float *scratch) {
  int status = ORACLE_OK;
  float *x_sq = scratch, *ll = scratch + dim;
  for (int t = 0; t < T; t++) {
    const float *x = feats + (size_t)t * ld;
    for (int d = 0; d < dim; d++) x_sq[d] = x[d] * x[d];
    for (int j = 0; j < num_pdfs; j++) {
      int g0 = pdf_offsets[j], M = pdf_offsets[j + 1] - g0;
      oracle_diag_gmm_loglikes_vec(gconsts + g0, means_invvars + (size_t)g0 * dim,
                                   inv_vars + (size_t)g0 * dim, M, dim, x, x_sq, ll);
      float v = oracle_log_sum_exp(ll, M, prune);
      if (isnan(v) || isinf(v)) status = ORACLE_ENONFINITE;
      out[(size_t)t * ldo + j] = v;
    }
  }
  return status;
}

/* gmm/diag-gmm.cc:546-562  DiagGmm::LogLikelihoods(Matrix): [T x M] per-Gaussian
 * scores, row t = oracle_diag_gmm_loglikes_vec(x_t).  (Two sgemm in the reference.) */
void oracle_diag_gmm_loglikes_mat(const float *gconsts, const float *means_invvars,
                                  const float *inv_vars, int num_mix, int dim,
                                  const float *feats, int T, int ld, float *out, int ldo,
                                  float *scratch /* dim floats */) {
  for (int t = 0; t < T; t++) {
    const float *x = feats + (size_t)t * ld;
    for (int d = 0; d < dim; d++) scratch[d] = x[d] * x[d];
    oracle_diag_gmm_loglikes_vec(gconsts, means_invvars, inv_vars, num_mix, dim, x, scratch,
                                 out + (size_t)t * ldo);
  }
}

/* gmm/diag-gmm.cc:801-871  DiagGmm::GaussianSelection(Matrix): per frame the num_gselect
 * best Gaussians (by per-Gaussian log-likelihood, ties towards the larger index, as
 * std::greater<pair<float,int>> orders them), best first, and the sum over frames of the
 * log-sum of the selected scores (LogAdd in float per frame, double over frames).
 * scores: [T x num_mix] from oracle_diag_gmm_loglikes_mat.  out_idx: [T x num_gselect],
 * -1 where num_gselect > num_mix. */
double oracle_gselect(const float *scores, int T, int num_mix, int lds, int num_gselect,
                      int *out_idx) {
  double ans = 0.0;
  char *used = (char *)malloc((size_t)num_mix);
  for (int t = 0; t < T; t++) {
    const float *f = scores + (size_t)t * lds;
    memset(used, 0, (size_t)num_mix);
    float tot = -INFINITY;
    for (int j = 0; j < num_gselect; j++) {
      int best = -1;
      for (int g = 0; g < num_mix; g++)
        if (!used[g] && (best < 0 || f[g] > f[best] || (f[g] == f[best] && g > best))) best = g;
      out_idx[(size_t)t * num_gselect + j] = best;
      if (best < 0) continue;
      used[best] = 1;
      /* LogAdd(tot, f[best]), base/kaldi-math.h:177-192 */
      float x = tot, y = f[best], diff;
      if (x < y) { diff = x - y; x = y; } else { diff = y - x; }
      if (diff >= -15.9423847198486328125f) tot = x + log1pf(expf(diff));
      else tot = x;
    }
    ans += tot;
  }
  free(used);
  return ans;
}

/* Wall-clock helper for the cpu_baseline leg (seconds, monotonic). */
double oracle_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
