// Diagonal-GMM acoustic scoring on gfx950 (MI355X): the packed-model builder,
// the fused contraction + per-pdf log-sum-exp kernel, and the amlike_gmm_* C ABI.
//
// What it replaces (paths relative to /root/reference/src):
//   gmmbin/gmm-compute-likes.cc:71-77   the frame x pdf loop
//   gmm/diag-gmm.cc:517-543             DiagGmm::LogLikelihood / LogLikelihoods(Vector)
//   gmm/diag-gmm.cc:546-562             DiagGmm::LogLikelihoods(Matrix)
//   gmm/decodable-am-diag-gmm.cc:28-72  the per-(frame,pdf) body behind DecodableInterface
//   matrix/kaldi-vector.cc:766-784      LogSumExp
//
// Formulation.  For Gaussian g and frame t
//   f[g,t] = gconst_g + sum_d miv[g,d] x[t,d] + sum_d (-0.5 iv[g,d]) x[t,d]^2
// is a [G x 2D] . [2D x T] contraction.  It runs on the exact-fp32 matrix
// instruction v_mfma_f32_32x32x2_f32: one instruction per feature dimension d,
// whose two k-slots are (miv_d, x_d) and (-0.5 iv_d, x_d^2), with gconst as the
// initial accumulator -- the same order of fp32 fused multiply-adds as
// "ll = gconsts; ll += miv.x; ll += -0.5 iv.x^2", no padding of K.
//
// Layout.  MODEL-STATIONARY: a wave keeps the A operands of NT tiles of 32
// Gaussians in registers for its whole life (NT*D VGPRs) and streams 32-frame
// tiles of features past them; the model is read from HBM once per block, the
// features (160 B/frame) come from L2.  Frames sit on the MFMA's lane axis and
// Gaussians on its register axis, so the per-pdf log-sum-exp is a running
// (max, sum) in two registers per lane with no cross-lane traffic until a pdf
// ends; then the two half-waves are merged with two v_permlane32_swap and lanes
// 0-31 store the pdf's value for their 32 frames.  The [T x G] matrix of
// per-Gaussian scores never exists in memory.
//
// Software pipeline: while the 16 accumulator registers of tile k go through
// the (max,sum) update on the VALU, the D matrix instructions of tile k+1 are
// issued between them (source order: ~D/16 MFMAs, one update, repeat).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "amlike_common.h"

namespace amlike {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kColDiscard = -1;       // segment whose result is dropped (padding chunks)
constexpr float kNegInit = -3.0e38f;  // finite "minus infinity" for the running max
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

// Device view of the packed model (see PackModel below for the layout).
struct GmmPackView {
  const float4 *apack;      // [num_wc][NT][DQ][64 lanes] float4
  const float *gcpack;      // [num_wc][NT][2 halves][16 regs]
  const uint32_t *endmask;  // [num_wc][NT]  bit i: register i closes a segment
  const int32_t *segcol;    // [num_wc][NT*16] output column per segment
  int num_wc;               // multiple of W
};

struct ScoreArgs {
  const float *feats;  // [T x ld], 16-byte aligned rows (ld % 4 == 0)
  int T, ld;
  float *out;  // [T x ldo]
  int ldo;
  int fpb;  // frames per block, multiple of 32
  int nbx;  // blocks along the Gaussian axis
  float *part_m, *part_s;  // [num_partials x part_ld] for pdfs split over waves
  int part_ld;
  int *flag;  // set to 1 when a result is NaN/Inf
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// One step of the running log-sum-exp: (m, s) represent log(sum) = m + ln(s).
__device__ __forceinline__ void lse_update(float &m, float &s, float f) {
  float d = f - m;
  float e = fast_exp2(-fabsf(d) * kLog2e);
  bool gt = d > 0.0f;
  s = gt ? fmaf(s, e, 1.0f) : (s + e);
  m = gt ? f : m;
}

// v_permlane32_swap: returns {x of lanes 0-31, x of lanes 32-63} in every lane.
__device__ __forceinline__ void both_halves(float x, float &lo, float &hi) {
  unsigned u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  lo = __uint_as_float(r[0]);
  hi = __uint_as_float(r[1]);
}

template <int D, int NT, int W>
__global__ void __launch_bounds__(64 * W, (64 * W) / 256)
    gmm_score_kernel(GmmPackView pk, ScoreArgs ar) {
  constexpr int DQ = (D + 3) / 4;
  // XCD-aware block order: the dispatcher deals consecutive block ids round-robin
  // over the 8 XCDs; give each XCD a contiguous run of (frame-chunk, model-chunk)
  // pairs so that all writers of one output row share an L2.  Bijective for any grid.
  int bx, by;
  {
    const int total = gridDim.x, L = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    by = Lp / ar.nbx;
    bx = Lp - by * ar.nbx;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int wc = bx * W + w;

  __shared__ float s_gc[W][NT * 32];
  __shared__ int s_col[W][NT * 16];
  for (int idx = lane; idx < NT * 32; idx += 64)
    s_gc[w][idx] = pk.gcpack[(size_t)wc * (NT * 32) + idx];
  for (int idx = lane; idx < NT * 16; idx += 64)
    s_col[w][idx] = pk.segcol[(size_t)wc * (NT * 16) + idx];
  uint32_t endm[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    endm[nt] = __builtin_amdgcn_readfirstlane(pk.endmask[wc * NT + nt]);

  // Model operands, resident in registers for the life of the wave.
  float a[NT][DQ * 4];
  {
    const float4 *ap = pk.apack + (size_t)wc * (NT * DQ * 64) + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq) {
        float4 v = ap[(nt * DQ + dq) * 64];
        a[nt][4 * dq + 0] = v.x;
        a[nt][4 * dq + 1] = v.y;
        a[nt][4 * dq + 2] = v.z;
        a[nt][4 * dq + 3] = v.w;
      }
  }
  __syncthreads();

  const int t_begin = by * ar.fpb;
  if (t_begin >= ar.T) return;
  const int t_end = min(ar.T, t_begin + ar.fpb);
  const int ntile = (t_end - t_begin + 31) >> 5;

  // B operand of one 32-frame tile: lane (j,h) holds x[t0+j][d] (h=0) or its square (h=1).
  float b[DQ * 4];
  {
    const int row = min(t_begin + j, ar.T - 1);
    const float4 *xp = reinterpret_cast<const float4 *>(ar.feats + (size_t)row * ar.ld);
#pragma unroll
    for (int dq = 0; dq < DQ; ++dq) {
      float4 v = xp[dq];
      b[4 * dq + 0] = h ? v.x * v.x : v.x;
      b[4 * dq + 1] = h ? v.y * v.y : v.y;
      b[4 * dq + 2] = h ? v.z * v.z : v.z;
      b[4 * dq + 3] = h ? v.w * v.w : v.w;
    }
  }

  // Prologue: the chain of (frame tile 0, model tile 0), not overlapped with anything.
  f32x16 acc_cur;
  {
    const float4 *g4 = reinterpret_cast<const float4 *>(&s_gc[w][h * 16]);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      float4 v = g4[q4];
      acc_cur[4 * q4 + 0] = v.x;
      acc_cur[4 * q4 + 1] = v.y;
      acc_cur[4 * q4 + 2] = v.z;
      acc_cur[4 * q4 + 3] = v.w;
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
      acc_cur = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][d], b[d], acc_cur, 0, 0, 0);
  }

  for (int ft = 0; ft < ntile; ++ft) {
    const int t0 = t_begin + (ft << 5);
    // Raw features of the next frame tile, in flight during this tile's work.
    float4 braw[DQ];
    {
      const int rown = min(t0 + 32 + j, ar.T - 1);
      const float4 *xp = reinterpret_cast<const float4 *>(ar.feats + (size_t)rown * ar.ld);
#pragma unroll
      for (int dq = 0; dq < DQ; ++dq) braw[dq] = xp[dq];
    }
    float bn[DQ * 4];
    const int row = t0 + j;
    const bool row_ok = (row < t_end) && (h == 0);
    float *orow = ar.out + (size_t)row * ar.ldo;

    float m = kNegInit, s = 0.0f;
    int seg = 0;
    int col_next = __builtin_amdgcn_readfirstlane(s_col[w][0]);
    // Keep the segment masks opaque per frame tile: otherwise all NT*16 bit tests
    // are hoisted out of this loop as 64-bit lane masks and spill.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+s"(endm[nt]));

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nnt = (nt + 1 == NT) ? 0 : nt + 1;
      const bool last = (nt + 1 == NT);
      if (last) {
#pragma unroll
        for (int dq = 0; dq < DQ; ++dq) {
          float4 v = braw[dq];
          bn[4 * dq + 0] = h ? v.x * v.x : v.x;
          bn[4 * dq + 1] = h ? v.y * v.y : v.y;
          bn[4 * dq + 2] = h ? v.z * v.z : v.z;
          bn[4 * dq + 3] = h ? v.w * v.w : v.w;
        }
      }
      f32x16 acc_next;
      {
        const float4 *g4 = reinterpret_cast<const float4 *>(&s_gc[w][nnt * 32 + h * 16]);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float4 v = g4[q4];
          acc_next[4 * q4 + 0] = v.x;
          acc_next[4 * q4 + 1] = v.y;
          acc_next[4 * q4 + 2] = v.z;
          acc_next[4 * q4 + 3] = v.w;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // This step's share of the next tile's matrix instructions.
#pragma unroll
        for (int d = (i * D) / 16; d < ((i + 1) * D) / 16; ++d)
          acc_next = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nnt][d], last ? bn[d] : b[d],
                                                          acc_next, 0, 0, 0);
        lse_update(m, s, acc_cur[i]);
        if ((endm[nt] >> i) & 1u) {  // wave-uniform: register i closes a pdf segment
          const int col = col_next;
          ++seg;
          col_next = __builtin_amdgcn_readfirstlane(s_col[w][min(seg, NT * 16 - 1)]);
          float m_lo, m_hi, s_lo, s_hi;
          both_halves(m, m_lo, m_hi);
          both_halves(s, s_lo, s_hi);
          const float M = fmaxf(m_lo, m_hi);
          const float S = s_lo * fast_exp2((m_lo - M) * kLog2e) +
                          s_hi * fast_exp2((m_hi - M) * kLog2e);
          if (col >= 0) {
            const float v = M + kLn2 * fast_log2(S);
            if (row_ok) {
              orow[col] = v;
              if (!(fabsf(v) <= 3.0e38f)) atomicOr(ar.flag, 1);
            }
          } else if (col != kColDiscard) {  // one part of a pdf split over several waves
            if (row_ok) {
              const size_t o = (size_t)(-col - 2) * ar.part_ld + row;
              ar.part_m[o] = M;
              ar.part_s[o] = S;
            }
          }
          m = kNegInit;
          s = 0.0f;
        }
      }
      acc_cur = acc_next;
    }
#pragma unroll
    for (int d = 0; d < DQ * 4; ++d) b[d] = bn[d];
  }
}

// Merges the parts of pdfs that were split over several waves:
// out[t][col] = log sum_parts s_p exp(m_p).   One thread per (split pdf, frame).
struct SplitPdf {
  int32_t col, first_part, num_parts;
};

__global__ void gmm_merge_parts_kernel(const SplitPdf *splits, int num_splits,
                                       const float *part_m, const float *part_s, int part_ld,
                                       int T, float *out, int ldo, int *flag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int sp = blockIdx.y;
  if (t >= T || sp >= num_splits) return;
  const SplitPdf d = splits[sp];
  float M = kNegInit;
  for (int p = 0; p < d.num_parts; ++p)
    M = fmaxf(M, part_m[(size_t)(d.first_part + p) * part_ld + t]);
  float S = 0.0f;
  for (int p = 0; p < d.num_parts; ++p) {
    const size_t o = (size_t)(d.first_part + p) * part_ld + t;
    S += part_s[o] * fast_exp2((part_m[o] - M) * kLog2e);
  }
  const float v = M + kLn2 * fast_log2(S);
  out[(size_t)t * ldo + d.col] = v;
  if (!(fabsf(v) <= 3.0e38f)) atomicOr(flag, 1);
}

// Copies [T x dim] features with an arbitrary row stride into a 16-byte aligned
// [T x ldp] image (ldp = dim rounded up to 4, tail zero-filled).
__global__ void gmm_pad_feats_kernel(const float *src, int T, int ld, int dim, float *dst,
                                     int ldp) {
  const size_t n = (size_t)T * ldp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t t = i / ldp;
    const int d = (int)(i - t * ldp);
    dst[i] = d < dim ? src[t * ld + d] : 0.0f;
  }
}

// Per-Gaussian scores (DiagGmm::LogLikelihoods): out[t][g-g0] = gc + miv.x - 0.5 iv.x^2.
// A 64-frame x 64-Gaussian tile per 256-thread block, operands staged in LDS,
// each thread a 4x4 register block.  K is only 2*dim, so this is bound by the
// [T x G] output stream, not by arithmetic.
__global__ void __launch_bounds__(256)
    gmm_gauss_loglikes_kernel(const float *__restrict__ gconsts, const float *__restrict__ miv,
                              const float *__restrict__ iv, int g0, int g1, int dim,
                              const float *__restrict__ feats, int T, int ld,
                              float *__restrict__ out, int ldo) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldk = dim + 1;
  float *xs = smem;              // [64][ldk]
  float *ms = xs + 64 * ldk;     // [64][ldk]
  float *vs = ms + 64 * ldk;     // [64][ldk]
  const int tb = blockIdx.y * 64, gb = g0 + blockIdx.x * 64;
  for (int idx = threadIdx.x; idx < 64 * dim; idx += 256) {
    const int r = idx / dim, d = idx - r * dim;
    const int t = min(tb + r, T - 1), g = min(gb + r, g1 - 1);
    xs[r * ldk + d] = feats[(size_t)t * ld + d];
    ms[r * ldk + d] = miv[(size_t)g * dim + d];
    vs[r * ldk + d] = iv[(size_t)g * dim + d];
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // tx: Gaussians, ty: frames
  float acc1[4][4], acc2[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc1[r][c] = acc2[r][c] = 0.0f;
  for (int d = 0; d < dim; ++d) {
    float x[4], mm[4], vv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = xs[(ty + 16 * r) * ldk + d];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mm[c] = ms[(tx + 16 * c) * ldk + d];
      vv[c] = vs[(tx + 16 * c) * ldk + d];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc1[r][c] = fmaf(mm[c], x[r], acc1[r][c]);
        acc2[r][c] = fmaf(vv[c], x[r] * x[r], acc2[r][c]);
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = tb + ty + 16 * r;
    if (t >= T) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int g = gb + tx + 16 * c;
      if (g < g1) {
        float v = gconsts[g];
        v = acc1[r][c] + v;
        v = fmaf(-0.5f, acc2[r][c], v);
        out[(size_t)t * ldo + (g - g0)] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Host side: the packed model.

struct Variant {
  int D, NT, W;
  void (*kernel)(GmmPackView, ScoreArgs);
  const char *name;
};

#define AMLIKE_VARIANT(D, NT, W) \
  { D, NT, W, gmm_score_kernel<D, NT, W>, "gmm_score_kernel<" #D "," #NT "," #W ">" }

// Register-stationary instantiations.  NT*D model registers per lane:
// W=4 (one wave per SIMD, 512 registers) or W=8 (two waves per SIMD, 256 registers).
static const Variant kVariants[] = {
    AMLIKE_VARIANT(40, 3, 8), AMLIKE_VARIANT(39, 3, 8),
    AMLIKE_VARIANT(40, 6, 4), AMLIKE_VARIANT(39, 6, 4),
};

struct Piece {
  int32_t pdf, g_begin, count, col;
};

}  // namespace amlike

using namespace amlike;

struct amlike_gmm {
  int32_t num_pdfs = 0, num_gauss = 0, dim = 0;
  int device = 0;
  const Variant *variant = nullptr;
  int num_wc = 0;        // wave-chunks, multiple of W
  int num_partials = 0;  // parts of pdfs split over several waves
  int num_splits = 0;
  double fill = 0.0;  // used slots / allocated slots
  // device arrays
  float4 *d_apack = nullptr;
  float *d_gcpack = nullptr;
  uint32_t *d_endmask = nullptr;
  int32_t *d_segcol = nullptr;
  SplitPdf *d_splits = nullptr;
  float *d_gconsts = nullptr, *d_miv = nullptr, *d_iv = nullptr;  // flat copies [G], [G x dim]
  int *d_flag = nullptr;
  std::vector<int32_t> pdf_offsets;
  hipStream_t stream = nullptr;  // used by the host-pointer entry points
  DeviceBuffer ws_feats, ws_out, ws_part, ws_pad;
  std::mutex mu;
  std::string info;
};

namespace amlike {

static const Variant *ChooseVariant(int dim) {
  const char *force = getenv("AMLIKE_GMM_VARIANT");  // e.g. "3x8": tuning aid
  for (const Variant &v : kVariants) {
    if (v.D != dim) continue;
    if (force) {
      char buf[32];
      snprintf(buf, sizeof(buf), "%dx%d", v.NT, v.W);
      if (strcmp(buf, force) != 0) continue;
    }
    return &v;
  }
  return nullptr;
}

// Packs the model for `v`.
//
// A wave-chunk is NT tiles x 32 slots.  Every pdf takes an even number of
// consecutive slots (odd sizes get one slot with gconst = -inf), so that a pdf
// boundary always falls between two accumulator registers: slot n of a tile
// lives in register n>>1 of half-wave n&1, i.e. in matrix row
//   r(n) = (i & 3) + 8 (i >> 2) + 4 (n & 1),  i = n >> 1.
// pdfs are laid out in order; when the next one does not fit in what is left
// of a chunk, a short look-ahead picks a later one that does.  A pdf larger
// than a whole chunk is cut into parts whose (max,sum) are merged afterwards.
static int PackModel(amlike_gmm *g, const float *gconsts, const float *miv, const float *iv,
                     const int32_t *offs) {
  const Variant &v = *g->variant;
  const int D = v.D, NT = v.NT, W = v.W, DQ = (D + 3) / 4;
  const int cap = NT * 32;
  std::vector<Piece> pieces;
  std::vector<SplitPdf> splits;
  int num_partials = 0;
  for (int j = 0; j < g->num_pdfs; ++j) {
    const int M = offs[j + 1] - offs[j];
    if (M <= 0) {
      SetError("amlike_gmm_create: pdf %d has %d Gaussians", j, M);
      return AMLIKE_EINVAL;
    }
    if (((M + 1) & ~1) <= cap) {
      pieces.push_back({j, offs[j], M, j});
    } else {
      SplitPdf sp{j, num_partials, 0};
      for (int b = 0; b < M; b += cap) {
        pieces.push_back({j, offs[j] + b, std::min(cap, M - b), -(num_partials + 2)});
        ++num_partials;
        ++sp.num_parts;
      }
      splits.push_back(sp);
    }
  }
  // Assign pieces to chunks.
  std::vector<std::vector<Piece>> chunks;
  {
    const int kLook = 48;
    std::vector<char> used(pieces.size(), 0);
    size_t head = 0;
    while (head < pieces.size()) {
      std::vector<Piece> cur;
      int left = cap;
      for (;;) {
        while (head < pieces.size() && used[head]) ++head;
        if (head >= pieces.size()) break;
        int pick = -1;
        const int need_head = (pieces[head].count + 1) & ~1;
        if (need_head <= left) {
          pick = (int)head;
        } else {
          int best = 0;
          for (size_t k = head + 1; k < pieces.size() && k < head + kLook; ++k) {
            if (used[k]) continue;
            const int need = (pieces[k].count + 1) & ~1;
            if (need <= left && need > best) {
              best = need;
              pick = (int)k;
            }
          }
        }
        if (pick < 0) break;
        used[pick] = 1;
        cur.push_back(pieces[pick]);
        left -= (pieces[pick].count + 1) & ~1;
        if (left == 0) break;
      }
      chunks.push_back(cur);
    }
  }
  size_t used_slots = 0;
  for (const Piece &p : pieces) used_slots += p.count;
  while (chunks.size() % W) chunks.push_back({});
  const int num_wc = (int)chunks.size();
  g->num_wc = num_wc;
  g->num_partials = num_partials;
  g->num_splits = (int)splits.size();
  g->fill = (double)used_slots / ((double)num_wc * cap);

  std::vector<float> apack((size_t)num_wc * NT * DQ * 64 * 4, 0.0f);
  std::vector<float> gcpack((size_t)num_wc * NT * 32, -std::numeric_limits<float>::infinity());
  std::vector<uint32_t> endmask((size_t)num_wc * NT, 0u);
  std::vector<int32_t> segcol((size_t)num_wc * NT * 16, kColDiscard);
  for (int wc = 0; wc < num_wc; ++wc) {
    int slot = 0, seg = 0;
    for (const Piece &p : chunks[wc]) {
      for (int q = 0; q < p.count; ++q) {
        const int n = slot + q, nt = n >> 5, nn = n & 31, i = nn >> 1, hh = nn & 1;
        const int r = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const size_t gi = (size_t)(p.g_begin + q);
        gcpack[((size_t)wc * NT + nt) * 32 + hh * 16 + i] = gconsts[gi];
        for (int d = 0; d < D; ++d) {
          const size_t base = ((((size_t)wc * NT + nt) * DQ + (d >> 2)) * 64) * 4 + (d & 3);
          apack[base + (size_t)r * 4] = miv[gi * D + d];
          apack[base + (size_t)(r + 32) * 4] = -0.5f * iv[gi * D + d];
        }
      }
      slot += (p.count + 1) & ~1;
      const int last_reg = (slot - 2) >> 1;  // register (chunk-wide index) that closes it
      endmask[(size_t)wc * NT + (last_reg >> 4)] |= 1u << (last_reg & 15);
      segcol[(size_t)wc * NT * 16 + seg++] = p.col;
    }
    if (slot < cap) {
      // Trailing padding: all -inf; it joins the last segment when the chunk is
      // not empty (the closing bit moves to the chunk's last register), else it
      // forms one discarded segment.
      if (seg > 0) {
        const int last_reg = (slot - 2) >> 1;
        endmask[(size_t)wc * NT + (last_reg >> 4)] &= ~(1u << (last_reg & 15));
      }
      endmask[(size_t)wc * NT + NT - 1] |= 1u << 15;
    }
  }

  AMLIKE_HIP(hipMalloc(&g->d_apack, apack.size() * sizeof(float)));
  AMLIKE_HIP(hipMalloc(&g->d_gcpack, gcpack.size() * sizeof(float)));
  AMLIKE_HIP(hipMalloc(&g->d_endmask, endmask.size() * sizeof(uint32_t)));
  AMLIKE_HIP(hipMalloc(&g->d_segcol, segcol.size() * sizeof(int32_t)));
  AMLIKE_HIP(hipMemcpy(g->d_apack, apack.data(), apack.size() * sizeof(float), hipMemcpyHostToDevice));
  AMLIKE_HIP(hipMemcpy(g->d_gcpack, gcpack.data(), gcpack.size() * sizeof(float), hipMemcpyHostToDevice));
  AMLIKE_HIP(hipMemcpy(g->d_endmask, endmask.data(), endmask.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  AMLIKE_HIP(hipMemcpy(g->d_segcol, segcol.data(), segcol.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!splits.empty()) {
    AMLIKE_HIP(hipMalloc(&g->d_splits, splits.size() * sizeof(SplitPdf)));
    AMLIKE_HIP(hipMemcpy(g->d_splits, splits.data(), splits.size() * sizeof(SplitPdf), hipMemcpyHostToDevice));
  }
  return AMLIKE_OK;
}

// Launches the scoring kernels for [T x dim] device features with conforming layout.
static int LaunchScore(amlike_gmm *g, const float *d_feats, int T, int ld, float *d_out, int ldo,
                       hipStream_t stream) {
  const Variant &v = *g->variant;
  const int nbx = g->num_wc / v.W;
  // Frames per block: enough blocks to balance the CUs, few enough that the
  // one-off load of the model operands is amortised over many frame tiles.
  const int cus = std::max(1, NumComputeUnits());
  const int slots = cus * (v.W == 4 ? 1 : 1);
  int nby = std::max(1, (6 * slots + nbx - 1) / nbx);
  int fpb = ((T + nby - 1) / nby + 31) & ~31;
  fpb = std::max(fpb, std::min(256, (T + 31) & ~31));
  fpb = std::min(fpb, 4096);
  nby = (T + fpb - 1) / fpb;
  ScoreArgs ar;
  ar.feats = d_feats;
  ar.T = T;
  ar.ld = ld;
  ar.out = d_out;
  ar.ldo = ldo;
  ar.fpb = fpb;
  ar.nbx = nbx;
  ar.part_m = ar.part_s = nullptr;
  ar.part_ld = 0;
  ar.flag = g->d_flag;
  if (g->num_partials > 0) {
    const int part_ld = (T + 31) & ~31;
    int rc = g->ws_part.Reserve((size_t)2 * g->num_partials * part_ld * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    ar.part_m = static_cast<float *>(g->ws_part.ptr);
    ar.part_s = ar.part_m + (size_t)g->num_partials * part_ld;
    ar.part_ld = part_ld;
  }
  GmmPackView pk{g->d_apack, g->d_gcpack, g->d_endmask, g->d_segcol, g->num_wc};
  hipLaunchKernelGGL(v.kernel, dim3(nbx * nby), dim3(64 * v.W), 0, stream, pk, ar);
  AMLIKE_HIP(hipGetLastError());
  if (g->num_splits > 0) {
    hipLaunchKernelGGL(gmm_merge_parts_kernel, dim3((T + 255) / 256, g->num_splits), dim3(256), 0,
                       stream, g->d_splits, g->num_splits, ar.part_m, ar.part_s, ar.part_ld, T,
                       d_out, ldo, g->d_flag);
    AMLIKE_HIP(hipGetLastError());
  }
  char buf[256];
  snprintf(buf, sizeof(buf), "%s grid=%d block=%d frames_per_block=%d wave_chunks=%d fill=%.4f splits=%d",
           v.name, nbx * nby, 64 * v.W, fpb, g->num_wc, g->fill, g->num_splits);
  g->info = buf;
  return AMLIKE_OK;
}

// Makes the features conform (16-byte aligned rows); returns the pointer/stride to use.
static int ConformFeats(amlike_gmm *g, const float *d_feats, int T, int ld, hipStream_t stream,
                        const float **use, int *use_ld) {
  const bool ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_feats) & 15) == 0) &&
                  ld >= ((g->dim + 3) & ~3);
  if (ok) {
    *use = d_feats;
    *use_ld = ld;
    return AMLIKE_OK;
  }
  const int ldp = (g->dim + 3) & ~3;
  int rc = g->ws_pad.Reserve((size_t)T * ldp * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  const size_t n = (size_t)T * ldp;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(gmm_pad_feats_kernel, dim3(blocks), dim3(256), 0, stream, d_feats, T, ld,
                     g->dim, static_cast<float *>(g->ws_pad.ptr), ldp);
  AMLIKE_HIP(hipGetLastError());
  *use = static_cast<const float *>(g->ws_pad.ptr);
  *use_ld = ldp;
  return AMLIKE_OK;
}

static int CheckScoreArgs(const amlike_gmm *g, int T, int ld, int feat_dim, int ldo,
                          const void *feats, const void *out) {
  if (!g || !feats || !out) {
    SetError("amlike_gmm_score: null argument");
    return AMLIKE_EINVAL;
  }
  if (feat_dim != g->dim) {
    SetError("Dim mismatch: data dim = %d vs. model dim = %d", feat_dim, g->dim);
    return AMLIKE_EDIM;
  }
  if (T <= 0 || ld < feat_dim || ldo < g->num_pdfs) {
    SetError("amlike_gmm_score: bad shape T=%d ld=%d ldo=%d", T, ld, ldo);
    return AMLIKE_EINVAL;
  }
  return AMLIKE_OK;
}

}  // namespace amlike

extern "C" {

int amlike_gmm_create(const float *gconsts, const float *means_invvars, const float *inv_vars,
                      const int32_t *pdf_offsets, int32_t num_pdfs, int32_t num_gauss,
                      int32_t dim, amlike_gmm_t **out) {
  if (!gconsts || !means_invvars || !inv_vars || !pdf_offsets || !out || num_pdfs <= 0 ||
      num_gauss <= 0 || dim <= 0 || pdf_offsets[0] != 0 || pdf_offsets[num_pdfs] != num_gauss) {
    SetError("amlike_gmm_create: invalid argument");
    return AMLIKE_EINVAL;
  }
  if (CurrentDevice() < 0) {
    int rc = amlike_init(0);
    if (rc != AMLIKE_OK) return rc;
  }
  const Variant *v = ChooseVariant(dim);
  if (!v) {
    SetError("amlike_gmm_create: no kernel for feature dimension %d", dim);
    return AMLIKE_EDIM;
  }
  amlike_gmm *g = new amlike_gmm;
  g->num_pdfs = num_pdfs;
  g->num_gauss = num_gauss;
  g->dim = dim;
  g->device = CurrentDevice();
  g->variant = v;
  g->pdf_offsets.assign(pdf_offsets, pdf_offsets + num_pdfs + 1);
  int rc = PackModel(g, gconsts, means_invvars, inv_vars, pdf_offsets);
  if (rc == AMLIKE_OK) {
    auto up = [&](float **dst, const float *src, size_t n) -> int {
      AMLIKE_HIP(hipMalloc(dst, n * sizeof(float)));
      AMLIKE_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
      return AMLIKE_OK;
    };
    rc = up(&g->d_gconsts, gconsts, (size_t)num_gauss);
    if (rc == AMLIKE_OK) rc = up(&g->d_miv, means_invvars, (size_t)num_gauss * dim);
    if (rc == AMLIKE_OK) rc = up(&g->d_iv, inv_vars, (size_t)num_gauss * dim);
  }
  if (rc == AMLIKE_OK) {
    hipError_t e = hipMalloc(&g->d_flag, sizeof(int));
    if (e == hipSuccess) e = hipMemset(g->d_flag, 0, sizeof(int));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      SetError("amlike_gmm_create: %s", hipGetErrorString(e));
      rc = AMLIKE_EHIP;
    }
  }
  if (rc != AMLIKE_OK) {
    amlike_gmm_destroy(g);
    return rc;
  }
  *out = g;
  return AMLIKE_OK;
}

int amlike_gmm_destroy(amlike_gmm_t *g) {
  if (!g) return AMLIKE_OK;
  (void)hipFree(g->d_apack);
  (void)hipFree(g->d_gcpack);
  (void)hipFree(g->d_endmask);
  (void)hipFree(g->d_segcol);
  (void)hipFree(g->d_splits);
  (void)hipFree(g->d_gconsts);
  (void)hipFree(g->d_miv);
  (void)hipFree(g->d_iv);
  (void)hipFree(g->d_flag);
  g->ws_feats.Free();
  g->ws_out.Free();
  g->ws_part.Free();
  g->ws_pad.Free();
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return AMLIKE_OK;
}

int32_t amlike_gmm_num_pdfs(const amlike_gmm_t *g) { return g ? g->num_pdfs : 0; }
int32_t amlike_gmm_num_gauss(const amlike_gmm_t *g) { return g ? g->num_gauss : 0; }
int32_t amlike_gmm_dim(const amlike_gmm_t *g) { return g ? g->dim : 0; }
const char *amlike_gmm_kernel_info(const amlike_gmm_t *g) { return g ? g->info.c_str() : ""; }

int amlike_gmm_score_dev(amlike_gmm_t *g, const float *d_feats, int32_t T, int32_t ld,
                         int32_t feat_dim, float log_sum_exp_prune, float *d_out, int32_t ldo,
                         void *stream) {
  int rc = CheckScoreArgs(g, T, ld, feat_dim, ldo, d_feats, d_out);
  if (rc != AMLIKE_OK) return rc;
  if (log_sum_exp_prune > 0.0f) {
    SetError("amlike_gmm_score: log_sum_exp_prune > 0 is not available in this build");
    return AMLIKE_EINVAL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float *use;
  int use_ld;
  rc = ConformFeats(g, d_feats, T, ld, st, &use, &use_ld);
  if (rc != AMLIKE_OK) return rc;
  return LaunchScore(g, use, T, use_ld, d_out, ldo, st);
}

int amlike_gmm_poll(amlike_gmm_t *g, void *stream) {
  if (!g) return AMLIKE_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int flag = 0;
  AMLIKE_HIP(hipMemcpyAsync(&flag, g->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
  AMLIKE_HIP(hipMemsetAsync(g->d_flag, 0, sizeof(int), st));
  AMLIKE_HIP(hipStreamSynchronize(st));
  if (flag) {
    SetError("Invalid answer (overflow or invalid variances/features?)");
    return AMLIKE_ENONFINITE;
  }
  return AMLIKE_OK;
}

int amlike_gmm_score(amlike_gmm_t *g, const float *feats, int32_t T, int32_t ld,
                     int32_t feat_dim, float log_sum_exp_prune, float *out, int32_t ldo) {
  int rc = CheckScoreArgs(g, T, ld, feat_dim, ldo, feats, out);
  if (rc != AMLIKE_OK) return rc;
  std::lock_guard<std::mutex> lock(g->mu);
  const int ldp = (g->dim + 3) & ~3;
  const int P = g->num_pdfs;
  // Bound the device footprint of one pass: at most ~1 GiB of output.
  const int max_t = std::max(1024, (int)std::min<size_t>(1u << 20, ((size_t)1 << 28) / (size_t)P));
  int status = AMLIKE_OK;
  for (int t0 = 0; t0 < T; t0 += max_t) {
    const int tn = std::min(max_t, T - t0);
    rc = g->ws_feats.Reserve((size_t)tn * ldp * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    rc = g->ws_out.Reserve((size_t)tn * P * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    float *df = static_cast<float *>(g->ws_feats.ptr);
    float *dout = static_cast<float *>(g->ws_out.ptr);
    if (ldp != g->dim) AMLIKE_HIP(hipMemsetAsync(df, 0, (size_t)tn * ldp * sizeof(float), g->stream));
    AMLIKE_HIP(hipMemcpy2DAsync(df, (size_t)ldp * sizeof(float), feats + (size_t)t0 * ld,
                                (size_t)ld * sizeof(float), (size_t)g->dim * sizeof(float), tn,
                                hipMemcpyHostToDevice, g->stream));
    rc = amlike_gmm_score_dev(g, df, tn, ldp, feat_dim, log_sum_exp_prune, dout, P, g->stream);
    if (rc != AMLIKE_OK) return rc;
    AMLIKE_HIP(hipMemcpy2DAsync(out + (size_t)t0 * ldo, (size_t)ldo * sizeof(float), dout,
                                (size_t)P * sizeof(float), (size_t)P * sizeof(float), tn,
                                hipMemcpyDeviceToHost, g->stream));
    rc = amlike_gmm_poll(g, g->stream);
    if (rc == AMLIKE_ENONFINITE) status = rc;
    else if (rc != AMLIKE_OK) return rc;
  }
  return status;
}

int amlike_gmm_gauss_loglikes_dev(amlike_gmm_t *g, int32_t pdf_begin, int32_t pdf_end,
                                  const float *d_feats, int32_t T, int32_t ld, int32_t feat_dim,
                                  float *d_out, int32_t ldo, void *stream) {
  if (!g || !d_feats || !d_out || pdf_begin < 0 || pdf_end > g->num_pdfs || pdf_begin >= pdf_end) {
    SetError("amlike_gmm_gauss_loglikes: invalid argument");
    return AMLIKE_EINVAL;
  }
  if (feat_dim != g->dim) {
    SetError("DiagGmm::ComponentLogLikelihood, dimension mismatch %d vs. %d", feat_dim, g->dim);
    return AMLIKE_EDIM;
  }
  const int g0 = g->pdf_offsets[pdf_begin], g1 = g->pdf_offsets[pdf_end];
  if (T <= 0 || ld < feat_dim || ldo < g1 - g0) {
    SetError("amlike_gmm_gauss_loglikes: bad shape T=%d ld=%d ldo=%d", T, ld, ldo);
    return AMLIKE_EINVAL;
  }
  const size_t smem = (size_t)3 * 64 * (g->dim + 1) * sizeof(float);
  if (smem > 64 * 1024) {
    SetError("amlike_gmm_gauss_loglikes: dim %d too large", g->dim);
    return AMLIKE_EDIM;
  }
  dim3 grid((g1 - g0 + 63) / 64, (T + 63) / 64);
  hipLaunchKernelGGL(gmm_gauss_loglikes_kernel, grid, dim3(256), smem,
                     static_cast<hipStream_t>(stream), g->d_gconsts, g->d_miv, g->d_iv, g0, g1,
                     g->dim, d_feats, T, ld, d_out, ldo);
  AMLIKE_HIP(hipGetLastError());
  return AMLIKE_OK;
}

int amlike_gmm_gauss_loglikes(amlike_gmm_t *g, int32_t pdf_begin, int32_t pdf_end,
                              const float *feats, int32_t T, int32_t ld, int32_t feat_dim,
                              float *out, int32_t ldo) {
  if (!g || !feats || !out || pdf_begin < 0 || pdf_end > g->num_pdfs || pdf_begin >= pdf_end) {
    SetError("amlike_gmm_gauss_loglikes: invalid argument");
    return AMLIKE_EINVAL;
  }
  if (T <= 0 || ld < feat_dim) {
    SetError("amlike_gmm_gauss_loglikes: bad shape T=%d ld=%d", T, ld);
    return AMLIKE_EINVAL;
  }
  std::lock_guard<std::mutex> lock(g->mu);
  const int M = g->pdf_offsets[pdf_end] - g->pdf_offsets[pdf_begin];
  const int max_t = std::max(64, (int)std::min<size_t>(1u << 20, ((size_t)1 << 28) / (size_t)M));
  for (int t0 = 0; t0 < T; t0 += max_t) {
    const int tn = std::min(max_t, T - t0);
    int rc = g->ws_feats.Reserve((size_t)tn * feat_dim * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    rc = g->ws_out.Reserve((size_t)tn * M * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    float *df = static_cast<float *>(g->ws_feats.ptr);
    float *dout = static_cast<float *>(g->ws_out.ptr);
    AMLIKE_HIP(hipMemcpy2DAsync(df, (size_t)feat_dim * sizeof(float), feats + (size_t)t0 * ld,
                                (size_t)ld * sizeof(float), (size_t)feat_dim * sizeof(float), tn,
                                hipMemcpyHostToDevice, g->stream));
    rc = amlike_gmm_gauss_loglikes_dev(g, pdf_begin, pdf_end, df, tn, feat_dim, feat_dim, dout, M,
                                       g->stream);
    if (rc != AMLIKE_OK) return rc;
    AMLIKE_HIP(hipMemcpy2DAsync(out + (size_t)t0 * ldo, (size_t)ldo * sizeof(float), dout,
                                (size_t)M * sizeof(float), (size_t)M * sizeof(float), tn,
                                hipMemcpyDeviceToHost, g->stream));
    AMLIKE_HIP(hipStreamSynchronize(g->stream));
  }
  return AMLIKE_OK;
}

int amlike_gmm_time_score_dev(amlike_gmm_t *g, const float *d_feats, int32_t T, int32_t ld,
                              int32_t feat_dim, float *d_out, int32_t ldo, void *stream,
                              int32_t iters, float *ms_per_launch) {
  if (!ms_per_launch || iters <= 0) return AMLIKE_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  AMLIKE_HIP(hipEventCreate(&e0));
  AMLIKE_HIP(hipEventCreate(&e1));
  AMLIKE_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) {
    int rc = amlike_gmm_score_dev(g, d_feats, T, ld, feat_dim, -1.0f, d_out, ldo, stream);
    if (rc != AMLIKE_OK) return rc;
  }
  AMLIKE_HIP(hipEventRecord(e1, st));
  AMLIKE_HIP(hipEventSynchronize(e1));
  float ms = 0.0f;
  AMLIKE_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = ms / iters;
  return AMLIKE_OK;
}

}  // extern "C"
