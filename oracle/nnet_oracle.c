/* oracle/nnet_oracle.c
 *
 * TEST INFRASTRUCTURE ONLY.  A plain-C, single-threaded CPU restatement of the
 * reference's nnet2 forward path (the CPU branches that CuMatrix takes when no
 * GPU is selected, which is how nnet-am-compute runs in the reference).  The
 * product never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_cpu.py checks it against the real
 * reference library (oracle/_ref/ref_driver nnet-compute / nnet-loglikes) and
 * against golden fixtures that driver produced.
 *
 * Paths are relative to /root/reference/src.
 */
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846 /* base/kaldi-math.h */
#endif
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { K_SPLICE = 1, K_AFFINE = 2, K_PNORM = 3, K_NORMALIZE = 4, K_SOFTMAX = 5,
       K_SUMGROUP = 6, K_SIGMOID = 7, K_TANH = 8, K_RELU = 9, K_SOFTHINGE = 10, K_POWER = 11,
       K_SCALE = 12, K_FIXED_SCALE = 13, K_FIXED_BIAS = 14, K_MAXOUT = 15, K_PERMUTE = 16,
       K_DCT = 17, K_BLOCK_AFFINE = 18, K_SPLICE_MAX = 19 };

typedef struct {
  int32_t kind, input_dim, output_dim;
  int32_t left_context;  /* SPLICE; DCT: reorder_ */
  int32_t right_context;
  int32_t const_dim;     /* SPLICE; DCT: dct_dim; BLOCK_AFFINE: num_blocks_ */
  float p;               /* PNORM / POWER: exponent; SCALE: scale_ */
  const float *linear; /* AFFINE: [output_dim x input_dim]; BLOCK_AFFINE: the stacked
                          [output_dim x input_dim / num_blocks] linear_params_ */
  const float *bias;   /* [output_dim] (NULL: FixedLinear); FIXED_SCALE / FIXED_BIAS: [dim] */
  const int32_t *sizes; /* SUMGROUP: sizes; PERMUTE: reorder_ */
} oracle_layer;

/* nnet2/nnet-nnet.cc:39-53  Nnet::LeftContext / RightContext: sums over components
 * (only SpliceComponent and SpliceMaxComponent have a non-zero context, nnet2/nnet-component.h). */
void oracle_nnet_context(const oracle_layer *L, int n, int *left, int *right) {
  *left = *right = 0;
  for (int i = 0; i < n; i++)
    if (L[i].kind == K_SPLICE || L[i].kind == K_SPLICE_MAX) { *left += L[i].left_context; *right += L[i].right_context; }
}

/* nnet2/nnet-component.cc:2411-2471  SpliceComponent::Propagate (one chunk):
 * out row r = [in row r+0 | in row r+1 | ... | in row r+L+R] over the first
 * (input_dim - const_dim) columns, then the const_dim tail of in row r. */
static void splice(const oracle_layer *l, const float *in, int rows_in, float *out) {
  int ns = l->left_context + l->right_context + 1, dim = l->input_dim - l->const_dim;
  int rows_out = rows_in - l->left_context - l->right_context, od = l->output_dim;
  for (int r = 0; r < rows_out; r++) {
    for (int c = 0; c < ns; c++)
      memcpy(out + (size_t)r * od + c * dim, in + (size_t)(r + c) * l->input_dim, dim * sizeof(float));
    if (l->const_dim)
      memcpy(out + (size_t)r * od + ns * dim, in + (size_t)r * l->input_dim + dim,
             l->const_dim * sizeof(float));
  }
}

/* nnet2/nnet-component.cc:2623-2654  SpliceMaxComponent::Propagate (one chunk): out row r starts
 * as in row r and takes Max() with in rows r+1 .. r+L+R (matrix/kaldi-matrix.cc Max: the other
 * matrix's element where it is larger). */
static void splice_max(const oracle_layer *l, const float *in, int rows_in, float *out) {
  int ns = l->left_context + l->right_context + 1, dim = l->input_dim;
  int rows_out = rows_in - l->left_context - l->right_context;
  for (int r = 0; r < rows_out; r++)
    for (int c = 0; c < dim; c++) {
      float v = in[(size_t)r * dim + c];
      for (int k = 1; k < ns; k++) {
        float x = in[(size_t)(r + k) * dim + c];
        if (v < x) v = x;
      }
      out[(size_t)r * dim + c] = v;
    }
}

/* nnet2/nnet-component.cc:1123-1131 (Affine: out = bias rows; out += in * W^T) and
 * :3028-3034 (FixedAffine: out = in * W^T; out += bias) -- fp32 sgemm in the
 * reference, a sequential fp32 dot product here. */
static void affine(const oracle_layer *l, const float *in, int rows, float *out) {
  int K = l->input_dim, N = l->output_dim;
  for (int r = 0; r < rows; r++)
    for (int n = 0; n < N; n++) {
      const float *w = l->linear + (size_t)n * K, *x = in + (size_t)r * K;
      float acc = 0.0f;
      for (int k = 0; k < K; k++) acc += x[k] * w[k];
      out[(size_t)r * N + n] = l->bias ? acc + l->bias[n] : acc; /* FixedLinear :2948-2953 */
    }
}

/* nnet2/nnet-component.cc:1907-1939  BlockAffineComponent::Propagate: out = bias rows; block b of
 * out += block b of in * (rows [b*ob, (b+1)*ob) of linear_params_)^T. */
static void block_affine(const oracle_layer *l, const float *in, int rows, float *out) {
  int nb = l->const_dim, ib = l->input_dim / nb, ob = l->output_dim / nb;
  for (int r = 0; r < rows; r++)
    for (int b = 0; b < nb; b++)
      for (int n = 0; n < ob; n++) {
        const float *w = l->linear + (size_t)(b * ob + n) * ib;
        const float *x = in + (size_t)r * l->input_dim + b * ib;
        float acc = 0.0f;
        for (int k = 0; k < ib; k++) acc += x[k] * w[k];
        out[(size_t)r * l->output_dim + b * ob + n] = l->bias[b * ob + n] + acc;
      }
}

/* nnet2/nnet-component.cc:443-454  MaxoutComponent::Propagate: each output column starts at
 * -1e20 and takes the max with every column of its group. */
static void maxout(const oracle_layer *l, const float *in, int rows, float *out) {
  int gs = l->input_dim / l->output_dim;
  for (int r = 0; r < rows; r++)
    for (int j = 0; j < l->output_dim; j++) {
      float m = (float)-1e20;
      for (int i = 0; i < gs; i++) {
        float v = in[(size_t)r * l->input_dim + j * gs + i];
        if (v > m) m = v;
      }
      out[(size_t)r * l->output_dim + j] = m;
    }
}

/* nnet2/nnet-component.cc:2262-2270  PermuteComponent::Propagate: reverse_reorder[reorder[i]] = i,
 * out(r, c) = in(r, reverse_reorder[c])  (CopyCols). */
static void permute(const oracle_layer *l, const float *in, int rows, float *out) {
  int D = l->input_dim;
  int *rev = (int *)malloc(D * sizeof(int));
  for (int i = 0; i < D; i++) rev[l->sizes[i]] = i;
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < D; c++) out[(size_t)r * D + c] = in[(size_t)r * D + rev[c]];
  free(rev);
}

/* nnet2/nnet-component.cc:2776-2800  DctComponent::Reorder on one row of `cols` columns: a zeroed
 * temporary of the row's length, temp(j * block_size_out + b) = row(b * block_size_in + j) for
 * b < block_size_out, j < block_size_in, then the row becomes the temporary. */
static void dct_reorder(float *row, int cols, int block_size_in, int block_size_out) {
  float *temp = (float *)calloc(cols, sizeof(float));
  for (int b = 0; b < block_size_out; b++)
    for (int j = 0; j < block_size_in; j++) temp[j * block_size_out + b] = row[b * block_size_in + j];
  memcpy(row, temp, cols * sizeof(float));
  free(temp);
}

/* nnet2/nnet-component.cc:2802-2830  DctComponent::Propagate with the matrix of
 * matrix/matrix-functions.cc:592-608 (float normalizer times a double cosine, stored as float;
 * only the first dct_keep_dim rows are built, :2744-2756). */
static void dct(const oracle_layer *l, const float *in, int rows, float *out) {
  int dim = l->input_dim, N = l->const_dim, chunks = dim / N, K = l->output_dim / chunks;
  float *M = (float *)malloc((size_t)K * N * sizeof(float));
  float normalizer = (float)sqrt(1.0 / (float)N);
  for (int j = 0; j < N; j++) M[j] = normalizer;
  normalizer = (float)sqrt(2.0 / (float)N);
  for (int k = 1; k < K; k++)
    for (int n = 0; n < N; n++) M[(size_t)k * N + n] = (float)(normalizer * cos((double)M_PI / N * (n + 0.5) * k));
  float *tmp = (float *)malloc(dim * sizeof(float));
  for (int r = 0; r < rows; r++) {
    memcpy(tmp, in + (size_t)r * dim, dim * sizeof(float));
    if (l->left_context) dct_reorder(tmp, dim, dim / N, K);
    float *o = out + (size_t)r * l->output_dim;
    for (int ch = 0; ch < chunks; ch++)
      for (int k = 0; k < K; k++) {
        float acc = 0.0f;
        for (int n = 0; n < N; n++) acc += tmp[ch * N + n] * M[(size_t)k * N + n];
        o[ch * K + k] = acc;
      }
    if (l->left_context) dct_reorder(o, l->output_dim, K, dim / N);
  }
  free(tmp);
  free(M);
}

/* matrix/kaldi-matrix.cc:2451-2459 GroupPnorm -> matrix/kaldi-vector.cc:529-566 Norm(p). */
static float vec_norm(const float *x, int n, float p) {
  float sum = 0.0f;
  if (p == 0.0f) {
    for (int i = 0; i < n; i++) if (x[i] != 0.0f) sum += 1.0f;
    return sum;
  } else if (p == 1.0f) {
    for (int i = 0; i < n; i++) sum += fabsf(x[i]);
    return sum;
  } else if (p == 2.0f) {
    for (int i = 0; i < n; i++) sum += x[i] * x[i];
    return sqrtf(sum);
  } else {
    int ok = 1;
    for (int i = 0; i < n; i++) {
      float tmp = powf(fabsf(x[i]), p);
      if (tmp == HUGE_VALF) ok = 0;
      sum += tmp;
    }
    float tmp = powf(sum, (float)(1.0 / p));
    if (ok) return tmp;
    float max_abs = 0.0f;
    for (int i = 0; i < n; i++) if (fabsf(x[i]) > max_abs) max_abs = fabsf(x[i]);
    float *s = (float *)malloc(n * sizeof(float));
    for (int i = 0; i < n; i++) s[i] = x[i] * (float)(1.0 / max_abs);
    float r = vec_norm(s, n, p) * max_abs;
    free(s);
    return r;
  }
}

static void pnorm(const oracle_layer *l, const float *in, int rows, float *out) {
  int gs = l->input_dim / l->output_dim;
  for (int r = 0; r < rows; r++)
    for (int j = 0; j < l->output_dim; j++)
      out[(size_t)r * l->output_dim + j] = vec_norm(in + (size_t)r * l->input_dim + j * gs, gs, l->p);
}

/* nnet2/nnet-component.cc:578-592  NormalizeComponent::Propagate:
 * f = pow(max(x.x / D, 2^-66), -0.5)  (AddDiagMat2 = alpha * VecVec(row,row), fp32),
 * out = f * in. */
static void normalize(const oracle_layer *l, const float *in, int rows, float *out) {
  int D = l->input_dim;
  const float floor_ = (float)pow(2.0, -66);
  for (int r = 0; r < rows; r++) {
    const float *x = in + (size_t)r * D;
    float dot = 0.0f;
    for (int d = 0; d < D; d++) dot += x[d] * x[d];
    float v = (float)(1.0 / D) * dot;
    if (v < floor_) v = floor_;
    float f = powf(v, -0.5f);
    for (int d = 0; d < D; d++) out[(size_t)r * D + d] = x[d] * f;
  }
}

/* nnet2/nnet-component.cc:905-918: per row matrix/kaldi-vector.cc:861-868 ApplySoftMax
 * (max; e_i = expf(x_i - max); fp32 sum; scale by 1/sum), then floor 1e-20. */
static void softmax(const oracle_layer *l, const float *in, int rows, float *out) {
  int D = l->input_dim;
  for (int r = 0; r < rows; r++) {
    const float *x = in + (size_t)r * D;
    float *y = out + (size_t)r * D;
    float mx = x[0], sum = 0.0f;
    for (int d = 1; d < D; d++) if (x[d] > mx) mx = x[d];
    for (int d = 0; d < D; d++) sum += (y[d] = expf(x[d] - mx));
    float inv = (float)(1.0 / sum);
    for (int d = 0; d < D; d++) {
      y[d] *= inv;
      if (y[d] < 1.0e-20f) y[d] = 1.0e-20f;
    }
  }
}

/* nnet2/nnet-component.cc:2352-2357 SumGroupComponent -> SumColumnRanges. */
static void sumgroup(const oracle_layer *l, const float *in, int rows, float *out) {
  for (int r = 0; r < rows; r++) {
    int c0 = 0;
    for (int j = 0; j < l->output_dim; j++) {
      float s = 0.0f;
      for (int c = c0; c < c0 + l->sizes[j]; c++) s += in[(size_t)r * l->input_dim + c];
      out[(size_t)r * l->output_dim + j] = s;
      c0 += l->sizes[j];
    }
  }
}

/* matrix/kaldi-vector.cc:496-525 ApplyPowAbs(power, include_sign = false) on one element; the
 * power == 1 branch is not chained to the others, so such an element also takes the last one. */
static float pow_abs(float x, float power) {
  if (power == 1.0f) x = fabsf(x);
  if (power == 2.0f) return x * x;
  if (power == 0.5f) return sqrtf(fabsf(x));
  if (power < 0.0f) return x == 0.0f ? 0.0f : (float)pow(fabsf(x), power);
  return (float)pow(fabsf(x), power);
}

/* matrix/kaldi-vector.cc:937-951 Sigmoid, :899-913 Tanh (non-MKL branches: double
 * literals promote the arithmetic to double), nnet-component.cc:782-788 ReLU,
 * matrix/kaldi-matrix.cc:2434-2448 SoftHinge (Exp and Log1p of base/kaldi-math.h: expf, log1pf),
 * nnet-component.cc:731-737 Power, :844-850 Scale, :3103-3109 FixedScale (MulColsVec),
 * :3171-3177 FixedBias (AddVecToRows). */
static void elementwise(const oracle_layer *l, const float *in, size_t n, float *out) {
  int kind = l->kind;
  for (size_t i = 0; i < n; i++) {
    float x = in[i];
    if (kind == K_SOFTHINGE) { out[i] = x > 10.0f ? x : log1pf(expf(x)); continue; }
    if (kind == K_POWER) { out[i] = pow_abs(x, l->p); continue; }
    if (kind == K_SCALE) { out[i] = x * l->p; continue; }
    if (kind == K_FIXED_SCALE) { out[i] = x * l->bias[i % l->input_dim]; continue; }
    if (kind == K_FIXED_BIAS) { out[i] = x + l->bias[i % l->input_dim]; continue; }
    if (kind == K_SIGMOID) {
      if (x > 0.0f) x = (float)(1.0 / (1.0 + expf(-x)));
      else { float ex = expf(x); x = (float)(ex / (ex + 1.0)); }
    } else if (kind == K_TANH) {
      if (x > 0.0f) { float e = expf(-x); x = (float)(-1.0 + 2.0 / (1.0 + e * e)); }
      else { float e = expf(x); x = (float)(1.0 - 2.0 / (1.0 + e * e)); }
    } else {
      if (x < 0.0f) x = 0.0f;
    }
    out[i] = x;
  }
}

/* nnet2/nnet-compute.cc:62-106,158-165 (NnetComputer: pad, propagate layer by
 * layer), then by mode:
 *   0  nothing (posteriors)
 *   1  nnet2bin/nnet-am-compute.cc:93-96   floor 1e-20, log
 *   2  nnet2/decodable-am-nnet.h:59-69     floor, log, minus log(prior), times prob_scale
 * Returns 0, or 1 on a dimension error. */
int oracle_nnet_compute(const oracle_layer *L, int n, const float *priors, int num_priors,
                        const float *feats, int T, int ld, int pad, int mode, float prob_scale,
                        float *out, int ldo) {
  int left, right;
  oracle_nnet_context(L, n, &left, &right);
  if (!pad) left = right = 0;
  int dim = L[0].input_dim;
  int rows = left + T + right;
  float *cur = (float *)malloc((size_t)rows * dim * sizeof(float));
  for (int r = 0; r < rows; r++) {
    int src = r - left;
    if (src < 0) src = 0;
    if (src > T - 1) src = T - 1;
    memcpy(cur + (size_t)r * dim, feats + (size_t)src * ld, dim * sizeof(float));
  }
  for (int i = 0; i < n; i++) {
    const oracle_layer *l = &L[i];
    if (l->input_dim != dim) { free(cur); return 1; }
    int rows_out = rows - (l->kind == K_SPLICE || l->kind == K_SPLICE_MAX ? l->left_context + l->right_context : 0);
    if (rows_out <= 0) { free(cur); return 1; }
    float *next = (float *)malloc((size_t)rows_out * l->output_dim * sizeof(float));
    switch (l->kind) {
      case K_SPLICE: splice(l, cur, rows, next); break;
      case K_SPLICE_MAX: splice_max(l, cur, rows, next); break;
      case K_AFFINE: affine(l, cur, rows, next); break;
      case K_PNORM: pnorm(l, cur, rows, next); break;
      case K_NORMALIZE: normalize(l, cur, rows, next); break;
      case K_SOFTMAX: softmax(l, cur, rows, next); break;
      case K_SUMGROUP: sumgroup(l, cur, rows, next); break;
      case K_BLOCK_AFFINE: block_affine(l, cur, rows, next); break;
      case K_MAXOUT: maxout(l, cur, rows, next); break;
      case K_PERMUTE: permute(l, cur, rows, next); break;
      case K_DCT: dct(l, cur, rows, next); break;
      default: elementwise(l, cur, (size_t)rows * dim, next); break;
    }
    free(cur);
    cur = next;
    rows = rows_out;
    dim = l->output_dim;
  }
  if (mode == 2 && num_priors != dim) { free(cur); return 1; }
  for (int r = 0; r < rows; r++)
    for (int d = 0; d < dim; d++) {
      float v = cur[(size_t)r * dim + d];
      if (mode >= 1) {
        if (v < 1.0e-20f) v = 1.0e-20f;
        v = logf(v);
      }
      if (mode == 2) {
        v = v + -1.0f * logf(priors[d]);
        v *= prob_scale;
      }
      out[(size_t)r * ldo + d] = v;
    }
  free(cur);
  return 0;
}
