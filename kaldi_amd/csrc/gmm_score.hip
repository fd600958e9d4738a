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
// is a [G x 2D] . [2D x T] contraction followed by a log-sum-exp over each pdf's Gaussians.
// Three kernel families compute it, all with the same layout:
//   gmm_score_diff_kernel    (default) 16-bit matrix instructions on split operands (fp16x2:
//                            3 partial products, or bf16x3: 6), the model packed relative to
//                            one reference Gaussian per pdf so that the log-sum-exp is one exp
//                            and one fma per element; see the comment at the kernel;
//   gmm_score_bf16x3_kernel  bf16x3 operands, running (max, sum) log-sum-exp: the wide-range
//                            fallback the default kernel hands over to (decided on the device);
//   gmm_score_kernel         the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32, one per
//                            feature dimension, gconst as the initial accumulator: the same
//                            order of fp32 fused multiply-adds as "ll = gconsts; ll += miv.x;
//                            ll += -0.5 iv.x^2".  Kept as an independent check (AMLIKE_GMM_MODE=f32).
//
// Layout.  MODEL-STATIONARY: a wave keeps the A operands of NT tiles of 32
// Gaussians in registers for its whole life and streams 32-frame tiles of features past
// them; the model is read from HBM once per block, the features come from L2 through LDS.
// Frames sit on the matrix instruction's lane axis and Gaussians on its register axis, so
// the per-pdf log-sum-exp is per-lane arithmetic over a lane's own accumulator registers with
// no cross-lane traffic until a pdf ends; then the two half-waves are merged with
// v_permlane32_swap and the pdf's value for 32 frames is stored.  The [T x G] matrix of
// per-Gaussian scores never exists in memory.
//
// Software pipeline: while the accumulator registers of tile k go through the log-sum-exp
// update on the VALU, the matrix instructions of tile k+1 are issued between them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "amlike_common.h"

namespace amlike {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));  // 8 bf16 = one 32x32x16 operand
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));  // 8 fp16 = one 32x32x16 operand

constexpr int kColDiscard = -1;       // segment whose result is dropped (padding chunks)
constexpr float kNegInit = -3.0e38f;  // finite "minus infinity" for the running max
// bf16x3 kernel: padding slots and zero-weight Gaussians carry this finite gconst
// (their exp2 is exactly 0); a result beyond kMaxAbsLogLike counts as non-finite.
constexpr float kPadGconst = -1.0e30f;
constexpr float kMaxAbsLogLike = 1.0e29f;
constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
// bf16x3 kernel: feature tiles staged per block barrier (LDS: 2 groups x this x 15 KiB at D=40;
// 8-wave blocks own a CU's LDS alone, 4-wave blocks share it two per CU).
constexpr int FramesTilesPerBarrier(int waves, int ks, int parts = 3) {
  const int budget = waves >= 8 ? 120 * 1024 : 60 * 1024, pair = 2 * ks * parts * 64 * 16;
  return budget / pair >= 4 ? 4 : budget / pair >= 2 ? 2 : 1;
}
// fp16x2 operands: mask on the bits of the low parts (tuning aid: fewer mantissa bits switch less
// in the matrix pipe; 0xFFFF keeps them all).
#ifndef AMLIKE_F16_LO_MASK
#define AMLIKE_F16_LO_MASK 0xFFFF
#endif
// fp16x2 operands: scaled feature values beyond this magnitude send the launch to the
// wide-range (bf16x3) kernel.
constexpr float kF16Limit = 60000.0f;

// Device view of the packed model (see PackModel below for the layout).
struct GmmPackView {
  const void *apack;        // f32: [num_wc][NT][DQ][64 lanes] float4
                            // bf16x3: [num_wc][NT][KS][3 parts][64 lanes] 8 x bf16
  const float *gcpack;      // [num_wc][NT][2 halves][16 regs]
  const uint32_t *endmask;  // [num_wc][NT]  bit i: register i closes a segment
  const int32_t *segcol;    // [num_wc][NT*16] output column per segment
  int num_wc;               // multiple of W
};

struct ScoreArgs {
  const float *feats;  // f32: [T x ld], 16-byte aligned rows (ld % 4 == 0)
  const uint4 *bx;     // bf16x3: expanded features [frame tile][KS][3 parts][64 lanes] 8 x bf16
  int T, ld;
  float *out;  // [T x ldo]
  int ldo;
  int fpb;  // frames per block, multiple of 32
  int nbx;  // blocks along the Gaussian axis
  float *part_m, *part_s;  // [num_partials x part_ld] for pdfs split over waves
  int part_ld;
  int *flag;  // set to 1 when a result is NaN/Inf
  // Fast-path bookkeeping.  ctl[0]: number of (frame, pdf) pairs whose sum overflowed
  // (listed in fix_list up to fix_cap); ctl[1]: nonzero when the features do not fit the
  // fp16 operand range.  guard 0: always run; 1: run unless ctl[1]; 2: run only when the
  // differenced kernel could not finish (ctl[1] set, or more overflows than fix_cap).
  int *ctl;
  int2 *fix_list;
  int fix_cap;
  int guard;
};

// An empty volatile asm that takes and returns `x` in vector registers: instructions that
// produce x stay before it and instructions that use x stay after it in program order.
template <typename T>
__device__ __forceinline__ void pin_order(T &x) {
#if __HIP_DEVICE_COMPILE__  // the host pass cannot parse the vector-register constraints
  asm volatile("" : "+v"(x));
#endif
}

// 1.0f when bit `bit` of the wave-uniform `mask` is set, else 0.0f, computed on the scalar
// unit (the compiler would otherwise turn it into a v_cndmask per use).
__device__ __forceinline__ float scalar_bit_as_float(uint32_t mask, int bit) {
  float r = 0.0f;
#if __HIP_DEVICE_COMPILE__  // the host pass cannot parse the scalar-register constraints
  asm volatile("s_bitcmp1_b32 %1, %2\n\ts_cselect_b32 %0, 1.0, 0" : "=s"(r) : "s"(mask), "s"(bit) : "scc");
#endif
  return r;
}

// Waits for every outstanding vector-memory operation of this wave.  The feature tiles are
// staged with global_load ... lds, whose arrival in LDS only the vector-memory counter tracks:
// a wave must have seen its own staging loads complete before it enters the barrier that
// publishes them (the compiler's barrier sequence does not wait on that counter by itself).
__device__ __forceinline__ void wait_vector_memory() {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

__device__ __forceinline__ bool guard_skips(const int *ctl, int fix_cap, int guard) {
  if (guard == 0) return false;
  const int range = ctl_load(ctl + 1), count = ctl_load(ctl);
  const bool fallback = range != 0 || count > fix_cap;
  return guard == 1 ? range != 0 : !fallback;
}

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
    const float4 *ap = static_cast<const float4 *>(pk.apack) + (size_t)wc * (NT * DQ * 64) + lane;
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
              if (!(fabsf(v) <= 3.0e38f)) ctl_or(ar.flag, 1);
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

// ---------------------------------------------------------------------------
// bf16x3 form of the same contraction, on the bf16 matrix pipe.
//
// On gfx950 the fp32-input MFMA executes at the vector-ALU rate and does not
// overlap with VALU instructions (measured: MFMA-busy + VALU-busy cycles add
// up, SQ_VALU_MFMA_COEXEC_CYCLES = 0), so in the kernel above every cycle of
// log-sum-exp arithmetic is a cycle the contraction loses.  v_mfma_f32_32x32x16_bf16
// runs on the matrix cores proper (16x the rate, concurrent with the VALU).  Each
// fp32 operand is written exactly as the sum of three bf16 values
// (x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits) and a product a*b is formed from the
// six partial products a1b1, a1b2, a2b1, a1b3, a2b2, a3b1, each exact in the fp32
// accumulator; the three dropped ones are below 2^-24 of the product, i.e. the
// result is fp32-grade (measured error vs fp64: same rms as an fp32 dot product).
//
// k index of MFMA step s (16 k-values): lane half h = 0 carries x_d, h = 1 carries
// x_d^2, for d = 8s .. 8s+7; the model side carries miv_d / -0.5 iv_d likewise,
// pre-multiplied by log2(e) so that the log-sum-exp runs directly on v_exp_f32 /
// v_log_f32 (base 2); the result is scaled back by ln 2 once per (frame, pdf).
//
// Structure: model operands stay in registers (NT*KS*12 VGPRs per lane); the
// expanded feature tile (15 KiB per 32 frames at D=40) is staged once per block
// in LDS by global_load_lds and read by all 8 waves; the two waves on each SIMD
// alternate between a matrix phase (30 MFMAs per 32x32 tile) and a VALU phase
// (the running (max,sum) update), which the hardware overlaps.

// Splits [T x dim] fp32 features into the bf16x3 operand image of the kernel below.
template <int KS>
__global__ void __launch_bounds__(256)
    gmm_expand_feats_kernel(const float *__restrict__ feats, int T, int ld, int dim,
                            uint4 *__restrict__ bx, int num_ftiles, const float *bscale, int *ctl,
                            int fix_cap, int guard) {
  if (guard_skips(ctl, fix_cap, guard)) return;
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (ftile, s, lane)
  const int lane = idx & 63;
  const int s = (idx >> 6) % KS;
  const int ft = (idx >> 6) / KS;
  if (ft >= num_ftiles) return;
  const int j = lane & 31, h = lane >> 5;
  const int t = ft * 32 + j;
  unsigned short p1[8], p2[8], p3[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int d = 8 * s + jj;
    float x = (t < T && d < dim) ? feats[(size_t)t * ld + d] : 0.0f;
    if (h) x = x * x;
    const __bf16 b1 = static_cast<__bf16>(x);
    const float r1 = x - static_cast<float>(b1);
    const __bf16 b2 = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(b2);
    const __bf16 b3 = static_cast<__bf16>(r2);
    p1[jj] = __builtin_bit_cast(unsigned short, b1);
    p2[jj] = __builtin_bit_cast(unsigned short, b2);
    p3[jj] = __builtin_bit_cast(unsigned short, b3);
  }
  auto pack = [](const unsigned short *p) {
    return make_uint4(p[0] | (unsigned)p[1] << 16, p[2] | (unsigned)p[3] << 16,
                      p[4] | (unsigned)p[5] << 16, p[6] | (unsigned)p[7] << 16);
  };
  uint4 *o = bx + ((size_t)(ft * KS + s) * 3) * 64 + lane;
  o[0] = pack(p1);
  o[64] = pack(p2);
  o[128] = pack(p3);
}

// fp16x2 operand image of the features for the differenced kernel: [frame tile][KS][2 parts][64 lanes]
// 8 x fp16.  Value k of a lane is x_d (h = 0) or x_d^2 (h = 1) times bscale[h][d], an exact
// power of two chosen with the model (PackModel) so that both operands sit well inside the
// fp16 range; x' = p1 + p2 to 22 bits.  A value outside the range sets ctl[1].
template <int KS>
__global__ void __launch_bounds__(256)
    gmm_expand_feats_f16_kernel(const float *__restrict__ feats, int T, int ld, int dim,
                                uint4 *__restrict__ bx, int num_ftiles,
                                const float *__restrict__ bscale, int *ctl, int fix_cap, int guard) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (ftile, s, lane)
  const int lane = idx & 63;
  const int s = (idx >> 6) % KS;
  const int ft = (idx >> 6) / KS;
  if (ft >= num_ftiles) return;
  const int j = lane & 31, h = lane >> 5;
  const int t = ft * 32 + j;
  unsigned short p1[8], p2[8];
  bool out_of_range = false;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int d = 8 * s + jj;
    float x = (t < T && d < dim) ? feats[(size_t)t * ld + d] : 0.0f;
    if (h) x = x * x;
    x *= bscale[h * (KS * 8) + d];
    out_of_range |= !(fabsf(x) <= kF16Limit);
    const _Float16 h1 = static_cast<_Float16>(x);
    const _Float16 h2 = static_cast<_Float16>(x - static_cast<float>(h1));
    p1[jj] = __builtin_bit_cast(unsigned short, h1);
    p2[jj] = __builtin_bit_cast(unsigned short, h2) & (unsigned short)AMLIKE_F16_LO_MASK;
  }
  auto pack = [](const unsigned short *p) {
    return make_uint4(p[0] | (unsigned)p[1] << 16, p[2] | (unsigned)p[3] << 16,
                      p[4] | (unsigned)p[5] << 16, p[6] | (unsigned)p[7] << 16);
  };
  uint4 *o = bx + ((size_t)(ft * KS + s) * 2) * 64 + lane;
  o[0] = pack(p1);
  o[64] = pack(p2);
  if (out_of_range) ctl_or(ctl + 1, 1);
}

// ABL (timing experiments only, never dispatched in production): bit0 no output
// stores, bit1 no (max,sum) updates, bit2 no segment closing, bit3 no feature
// fragment reads, bit4 no barrier / staging.
template <int KS, int NT, int W, int OCC, int ABL = 0>
__global__ void __launch_bounds__(64 * W, OCC)
    gmm_score_bf16x3_kernel(GmmPackView pk, ScoreArgs ar) {
  constexpr int kTileU4 = KS * 3 * 64;  // uint4 per expanded 32-frame tile
  constexpr int FTB = FramesTilesPerBarrier(W, KS);
  static_assert(NT * 16 <= 64, "segment columns are held one per lane");
  if (guard_skips(ar.ctl, ar.fix_cap, ar.guard)) return;
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

  // All LDS in one array: [2 groups of FTB feature tiles][gconst per wave].
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  uint4 *s_b = reinterpret_cast<uint4 *>(smem_h);
  float *s_gc = reinterpret_cast<float *>(smem_h + 2 * FTB * kTileU4 * 16) + w * (NT * 32);
  for (int idx = lane; idx < NT * 32; idx += 64)
    s_gc[idx] = pk.gcpack[(size_t)wc * (NT * 32) + idx];
  // Output column of segment k of this wave's chunk: lane k of colreg (v_readlane, no memory).
  const int colreg = lane < NT * 16 ? pk.segcol[(size_t)wc * (NT * 16) + lane] : kColDiscard;
  uint32_t endm[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    endm[nt] = __builtin_amdgcn_readfirstlane(pk.endmask[wc * NT + nt]);

  bf16x8 a[NT][KS][3];
  {
    const uint4 *ap = static_cast<const uint4 *>(pk.apack) + (size_t)wc * (NT * KS * 3 * 64) + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[nt][s][p] = __builtin_bit_cast(bf16x8, ap[((nt * KS + s) * 3 + p) * 64]);
  }

  const int t_begin = by * ar.fpb;
  const int t_end = min(ar.T, t_begin + ar.fpb);
  const int ntile = t_begin < ar.T ? (t_end - t_begin + 31) >> 5 : 0;
  const int ngroup = (ntile + FTB - 1) / FTB;
  const uint4 *gtile = ar.bx + (size_t)(t_begin >> 5) * kTileU4;

  // Stage group `g` (FTB expanded feature tiles) into LDS buffer g & 1: pieces of
  // 1 KiB (64 lanes x 16 B) dealt round-robin to the waves; the LDS image is lane-linear.
  auto stage = [&](int g) {
    const int npiece = min(FTB, ntile - g * FTB) * (KS * 3);
    for (int piece = w; piece < npiece; piece += W) {
      const uint4 *src = gtile + (size_t)g * FTB * kTileU4 + piece * 64 + lane;
      uint4 *dst = s_b + (g & 1) * FTB * kTileU4 + piece * 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
  };
  if (ngroup > 0) stage(0);
  if (ngroup > 1) stage(1);
  wait_vector_memory();
  __syncthreads();  // gconst visible; groups 0 and 1 landed (the barrier drains LDS-DMA)
  if (ntile == 0) return;

  constexpr int kMfma = KS * 6 * NT;  // matrix instructions per frame tile (NT model tiles)
  constexpr int kSteps = NT * 16;     // (max,sum) update steps per frame tile
  constexpr int kSkip = 2;            // leading steps without matrix work (cover LDS latency)
  // Feature fragments of one k-step (3 parts x 8 bf16 per lane), double-buffered in
  // registers: the fragments of k-step s+1 are requested from LDS when k-step s starts.
  bf16x8 bf[2][3];
  auto load_b = [&](const uint4 *bt, int s) {
#pragma unroll
    for (int p = 0; p < 3; ++p) bf[s & 1][p] = __builtin_bit_cast(bf16x8, bt[(s * 3 + p) * 64]);
  };
  // Matrix instruction n of a frame tile's work: k-step s, partial product `which`
  // (small terms first), model tile nt innermost, so that consecutive instructions
  // go to different accumulators (NT independent chains per wave).
  // The caller has issued load_b(bt, 0) before n = 0.
  auto mfma_n = [&](f32x16 (&acc)[NT], const uint4 *bt, int n) {
    const int s = n / (6 * NT), r = n % (6 * NT), which = r / NT, nt = r % NT;
    if (r == 0 && s + 1 < KS && !(ABL & 8)) load_b(bt, s + 1);
    // (model part, feature part): (3,1) (2,2) (1,3) (2,1) (1,2) (1,1)
    const int pa = which == 0 ? 2 : (which == 1 || which == 3) ? 1 : 0;
    const int pb = (which == 0 || which == 3 || which == 5) ? 0 : (which == 1 || which == 4) ? 1 : 2;
    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nt][s][pa], bf[s & 1][pb], acc[nt], 0, 0, 0);
  };
  auto init_acc = [&](f32x16 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float4 *g4 = reinterpret_cast<const float4 *>(&s_gc[nt * 32 + h * 16]);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        float4 v = g4[q4];
        acc[nt][4 * q4 + 0] = v.x;
        acc[nt][4 * q4 + 1] = v.y;
        acc[nt][4 * q4 + 2] = v.z;
        acc[nt][4 * q4 + 3] = v.w;
      }
    }
  };

  // Prologue: the chains of frame tile 0.
  f32x16 acc_cur[NT];
  init_acc(acc_cur);
  load_b(s_b + lane, 0);
#pragma unroll
  for (int n = 0; n < kMfma; ++n) mfma_n(acc_cur, s_b + lane, n);

  int bad = 0;  // becomes nonzero when a result of this lane is NaN/Inf
  for (int ft = 0; ft < ntile; ++ft) {
    const int t0 = t_begin + (ft << 5);
    const int row = t0 + j;
    const bool row_ok = (row < t_end) && (h == 0);
    float *orow = ar.out + (size_t)row * ar.ldo;

    float m = kNegInit, s_ = 0.0f;
    int seg = 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+s"(endm[nt]));

    // Feature tile ft+1 opens a new group: every wave has completed its LDS reads of
    // the group that ends with tile ft (their chains were issued in earlier stages, and
    // the barrier waits for them), and the new group has landed.  Only now may group
    // g+2 overwrite the buffer of group g.
    const int fn = ft + 1;
    if (fn % FTB == 0 && !(ABL & 16)) {
      wait_vector_memory();
      __syncthreads();
      const int g = fn / FTB;
      if (g + 1 < ngroup) stage(g + 1);
    }
    const uint4 *bt = s_b + (((fn / FTB) & 1) * FTB + fn % FTB) * kTileU4 + lane;
    f32x16 acc_next[NT];
    init_acc(acc_next);
    if (!(ABL & 8)) load_b(bt, 0);

#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const int nt = st >> 4, i = st & 15;
      // This step's share of frame tile ft+1's matrix instructions ...
#pragma unroll
      for (int n = (max(st - kSkip, 0) * kMfma) / (kSteps - kSkip);
           n < (max(st + 1 - kSkip, 0) * kMfma) / (kSteps - kSkip); ++n)
        mfma_n(acc_next, bt, n);
      if (ABL & 2) asm volatile("" ::"v"(acc_cur[nt][i]));  // keep the chain alive when timing
      if (!(ABL & 2)) {  // ... and one log2-sum-exp2 step: log2(sum) = m + log2(s_)
        const float f = acc_cur[nt][i];
        {
          const float d = f - m;
          const float e = fast_exp2(-fabsf(d));
          const bool gt = d > 0.0f;
          s_ = gt ? fmaf(s_, e, 1.0f) : (s_ + e);
          m = gt ? f : m;
        }
      }
      if (((endm[nt] >> i) & 1u) && !(ABL & 4)) {  // wave-uniform: register i closes a pdf segment
        const int col = __builtin_amdgcn_readlane(colreg, seg);
        ++seg;
        float m_lo, m_hi, s_lo, s_hi;
        both_halves(m, m_lo, m_hi);
        both_halves(s_, s_lo, s_hi);
        const float M = fmaxf(m_lo, m_hi);
        const float S = s_lo * fast_exp2(m_lo - M) + s_hi * fast_exp2(m_hi - M);
        if (col >= 0) {
          const float v = (M + fast_log2(S)) * kLn2;
          bad |= !(fabsf(v) <= kMaxAbsLogLike);
          if (row_ok && !(ABL & 1)) orow[col] = v;
        } else if (col != kColDiscard) {
          bad |= !(fabsf(M) <= kMaxAbsLogLike) | !(S <= 3.0e38f);
          if (row_ok) {  // parts are merged in natural-log units
            const size_t o = (size_t)(-col - 2) * ar.part_ld + row;
            ar.part_m[o] = M * kLn2;
            ar.part_s[o] = S;
          }
        }
        m = kNegInit;
        s_ = 0.0f;
      }
    }
    if (ABL & 4) asm volatile("" ::"v"(m), "v"(s_));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc_cur[nt] = acc_next[nt];
  }
  if (bad) ctl_or(ar.flag, 1);
}


// ---------------------------------------------------------------------------
// The default scoring kernel ("differenced").  Same skeleton as gmm_score_bf16x3_kernel
// (model operands stationary in registers, expanded features staged through LDS, matrix
// instructions of frame tile k+1 interleaved with the log-sum-exp of tile k), with two
// changes that halve the matrix work and shrink the vector work per (Gaussian, frame):
//
//  PREC = 1: fp16x2 operands.  With 11 mantissa bits per fp16, value = p1 + p2 carries
//    22 bits and a product needs three partial products (a2 b1, a1 b2, a1 b1) instead of the
//    six of bf16x3.  PackModel scales every k-row by a power of two so that both operands
//    stay inside the fp16 range, and the feature prepass reports values that do not (the
//    launch then runs the bf16x3 kernel, which has the fp32 exponent range).  PREC = 0
//    keeps bf16x3 operands (models that cannot be balanced into the fp16 range).
//
//  The model is packed relative to a reference Gaussian per pdf segment (PackModel,
//    "differenced" image): slot 0 of a segment holds the reference itself (its score
//    f_ref, in natural-log units), every other slot holds (Gaussian - reference) in log2
//    units, so the matrix instructions deliver d_i = f_i - f_ref directly and
//        log sum_i exp f_i = f_ref + log(1 + sum_{i>0} exp d_i).
//    The per-register work (the "pass") is v_exp_f32 + v_fma_f32 with a wave-uniform 0/1
//    factor that restarts the sum at segment boundaries; it has no branch.  Every running sum
//    is kept (sv); the "tail" then picks the sums at the segment ends by register index, joins
//    the two half-waves and writes the results.  A sum that overflows (some Gaussian more
//    than 2^128 above the reference; PackModel makes the broadest Gaussian the reference,
//    which keeps that rare) is listed in fix_list and recomputed exactly by gmm_fix_kernel.
//
//  The two waves that share a SIMD take turns on the matrix pipe: a wave in its tail issues no
//    matrix instruction, so the tail of one should run beside the pass of the other, and it
//    must be short.  The tail therefore works from per-wave tables prepared once (segment end
//    registers packed in scalar registers, per-lane output offsets in LDS), handles two pairs
//    of segments per step in straight-line code, and falls back to the general loop only for
//    a block's ragged last frame tile and for chunks that hold a part of a split pdf.
#ifdef AMLIKE_DIFF_STAMPS
// Diagnostic build only: cycles (s_memtime) that each wave of one interior block spent in its
// register passes, in its tails and in total: [wave][0..3] = pass, tail, total, tiles.
__device__ unsigned long long g_diff_stamps[8 * 6];
__device__ unsigned long long g_diff_blocks[256 * 4];  // per frame chunk: entry time, duration (10 ns), ticks, XCC id
#define AMLIKE_STAMP() __builtin_readcyclecounter()
#endif
// AMLIKE_DIFF_PRIO 1: a wave in its pass gets issue priority over a wave in its tail; 3: the
// other way round; 0: none.
#ifndef AMLIKE_DIFF_PRIO
#define AMLIKE_DIFF_PRIO 1
#endif
// AMLIKE_DIFF_PP 1: the waves of a SIMD alternate pass and tail under a barrier per phase
// (see the kernel body); 0: every wave runs pass and tail back to back.
#ifndef AMLIKE_DIFF_PP
#define AMLIKE_DIFF_PP 0
#endif
// AMLIKE_DIFF_GCLDS 1: the gconsts are read from LDS into the accumulators before every tile
// instead of living in 16 * NT registers.
#ifndef AMLIKE_DIFF_GCLDS
#define AMLIKE_DIFF_GCLDS AMLIKE_DIFF_PP
#endif
#ifndef AMLIKE_DIFF_STAGGER
#define AMLIKE_DIFF_STAGGER 1
#endif
#ifndef AMLIKE_DIFF_EARLYLOAD
#define AMLIKE_DIFF_EARLYLOAD 1
#endif
#ifndef AMLIKE_DIFF_FASTTAIL
#define AMLIKE_DIFF_FASTTAIL 1
#endif
#ifndef AMLIKE_DIFF_PRE
#define AMLIKE_DIFF_PRE 2
#endif
// AMLIKE_DIFF_STATICGROUPS 1: the tile loop is written out per position of a tile in its staged
// group (see the kernel body).
#ifndef AMLIKE_DIFF_STATICGROUPS
#define AMLIKE_DIFF_STATICGROUPS 1
#endif
// Bytes of LDS per wave for the output-offset table of the differenced kernel's tail.
constexpr int DiffTailTableBytes(int nt) { return (nt * 16 / 4) * 64 * 4; }

template <int KS, int NT, int W, int OCC, int PREC, int ABL = 0>
__global__ void __launch_bounds__(64 * W, OCC)
    gmm_score_diff_kernel(GmmPackView pk, ScoreArgs ar) {
  constexpr int NP = PREC ? 2 : 3;
  constexpr int NPROD = PREC ? 3 : 6;
  constexpr int kTileU4 = KS * NP * 64;
  constexpr int FTB = FramesTilesPerBarrier(W, KS, NP);
  constexpr int kSteps = NT * 16;
  constexpr int kMaxPairs = kSteps / 4;  // a segment takes at least two registers
  static_assert(NT == 1 || NT == 2, "segment masks are 32 bits");
  // All accumulator registers of a lane as one vector: register r of the chunk is element r,
  // whichever 32-Gaussian tile it belongs to.
  typedef float accvec __attribute__((ext_vector_type(kSteps)));
  if (guard_skips(ar.ctl, ar.fix_cap, ar.guard)) return;
#ifdef AMLIKE_DIFF_STAMPS
  const unsigned long long real_entry = __builtin_amdgcn_s_memrealtime();
#endif
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

  // LDS: [2 groups of FTB feature tiles][gconsts per wave][output-offset table per wave]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  uint4 *s_b = reinterpret_cast<uint4 *>(smem_h);
  float *s_gc = reinterpret_cast<float *>(smem_h + 2 * FTB * kTileU4 * 16) + w * (NT * 32);
  uint32_t *s_voff = reinterpret_cast<uint32_t *>(smem_h + 2 * FTB * kTileU4 * 16 + W * NT * 32 * 4) +
                     w * (kMaxPairs * 64);
  for (int idx = lane; idx < NT * 32; idx += 64)
    s_gc[idx] = pk.gcpack[(size_t)wc * (NT * 32) + idx];
  const int colreg = lane < kSteps ? pk.segcol[(size_t)wc * kSteps + lane] : kColDiscard;
  // Segment structure of this wave's chunk, all wave-uniform: bit r of `em` = register r
  // closes a segment; keep bit r = the running sum continues into register r (it does not
  // at a segment's first register, nor at the next one, which drops the reference's term).
  uint32_t em = __builtin_amdgcn_readfirstlane(pk.endmask[wc * NT]);
  if (NT == 2) em |= (uint32_t)__builtin_amdgcn_readfirstlane(pk.endmask[wc * NT + 1]) << 16;
  const uint32_t startm = (em << 1) | 1u;
  uint32_t keepm = ~(startm | (startm << 1));
  const int nseg = __builtin_popcount(em);
  const int npair = (nseg + 1) >> 1;
  int endreg = kSteps - 1;  // lane k: last register of segment k
  {
    int seen = 0;
    for (int r = 0; r < kSteps; ++r)
      if ((em >> r) & 1u) {
        if (seen == lane) endreg = r;
        ++seen;
      }
  }
  // Tables of the fast tail.  endw: the end registers of the segments, 5 bits each, six
  // segments (three pairs) per word, in scalar registers.  s_voff[k][lane]: byte offset from
  // the first output row of a frame tile to the element this lane writes for pair k (its own
  // row, the column of segment 2k + h), or ~0 where it writes nothing (discarded padding,
  // no segment).  A chunk that holds a part of a split pdf uses the general tail.
  uint32_t endw[(kMaxPairs + 2) / 3];
  bool fast_chunk;
  {
#pragma unroll
    for (int q = 0; q < (kMaxPairs + 2) / 3; ++q) endw[q] = 0;
#pragma unroll
    for (int k = 0; k < kMaxPairs; ++k) {
      const int ea = __builtin_amdgcn_readlane(endreg, 2 * k);
      const int eb = __builtin_amdgcn_readlane(endreg, 2 * k + 1);
      endw[k / 3] |= (uint32_t)(ea | (eb << 5)) << (10 * (k % 3));
      const int ca = __builtin_amdgcn_readlane(colreg, 2 * k);
      const int cb = __builtin_amdgcn_readlane(colreg, 2 * k + 1);
      const int c = h ? cb : ca;
      s_voff[k * 64 + lane] = c >= 0 ? ((uint32_t)j * (uint32_t)ar.ldo + (uint32_t)c) * 4u : ~0u;
    }
    const bool part_here = lane < nseg && colreg < kColDiscard;
    fast_chunk = AMLIKE_DIFF_FASTTAIL && __builtin_amdgcn_ballot_w64(part_here) == 0 &&
                 (unsigned)ar.ldo < (1u << 24);
  }

  bf16x8 a[NT][KS][NP];
  {
    const uint4 *ap = static_cast<const uint4 *>(pk.apack) + (size_t)wc * (NT * KS * NP * 64) + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          a[nt][s][p] = __builtin_bit_cast(bf16x8, ap[((nt * KS + s) * NP + p) * 64]);
  }

  const int t_begin = by * ar.fpb;
  const int t_end = min(ar.T, t_begin + ar.fpb);
  const int ntile = t_begin < ar.T ? (t_end - t_begin + 31) >> 5 : 0;
  const int ngroup = (ntile + FTB - 1) / FTB;
  const uint4 *gtile = ar.bx + (size_t)(t_begin >> 5) * kTileU4;

  auto stage = [&](int g) {
    const int npiece = min(FTB, ntile - g * FTB) * (KS * NP);
    for (int piece = w; piece < npiece; piece += W) {
      const uint4 *src = gtile + (size_t)g * FTB * kTileU4 + piece * 64 + lane;
      uint4 *dst = s_b + (g & 1) * FTB * kTileU4 + piece * 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
  };
  if (ngroup > 0) stage(0);
  if (ngroup > 1) stage(1);
  wait_vector_memory();
  __syncthreads();
  if (ntile == 0) return;

  constexpr int kMfma = KS * NPROD * NT;
  // Feature fragments are requested from LDS kPre k-steps before the matrix instructions
  // that read them (a wave that waits on LDS issues nothing, its matrix work included).
  constexpr int kPre = AMLIKE_DIFF_PRE < KS ? AMLIKE_DIFF_PRE : (KS > 1 ? KS - 1 : 1);
  constexpr int NB = kPre + 1;
  bf16x8 bf[NB][NP];
  auto load_b = [&](const uint4 *bt, int s) {
#pragma unroll
    for (int p = 0; p < NP; ++p) bf[s % NB][p] = __builtin_bit_cast(bf16x8, bt[(s * NP + p) * 64]);
  };
  auto get_tile = [](const accvec &v, int nt) -> f32x16 {
    if constexpr (NT == 1) {
      return v;
    } else {
      return nt ? v.hi : v.lo;
    }
  };
  auto set_tile = [](accvec &v, int nt, const f32x16 &t) {
    if constexpr (NT == 1) {
      v = t;
    } else {
      if (nt) v.hi = t;
      else v.lo = t;
    }
  };
  // The gconst image of this lane's accumulator registers: the first matrix instruction of every
  // chain takes it as its C operand, either from registers kept for the life of the wave (gcv)
  // or read from LDS into the accumulators before each tile (load_gc; 16 * NT registers fewer).
  auto read_gc = [&](accvec &acc) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float4 *g4 = reinterpret_cast<const float4 *>(&s_gc[nt * 32 + h * 16]);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        float4 v = g4[q4];
        acc[16 * nt + 4 * q4 + 0] = v.x;
        acc[16 * nt + 4 * q4 + 1] = v.y;
        acc[16 * nt + 4 * q4 + 2] = v.z;
        acc[16 * nt + 4 * q4 + 3] = v.w;
      }
    }
  };
#if AMLIKE_DIFF_GCLDS
  auto load_gc = [&](accvec &acc) __attribute__((always_inline)) { read_gc(acc); };
#else
  accvec gcv;
  read_gc(gcv);
  auto load_gc = [&](accvec &) {};
#endif
  auto mfma_n = [&](accvec &acc, const uint4 *bt, int n) {
    const int s = n / (NPROD * NT), r = n % (NPROD * NT), which = r / NT, nt = r % NT;
    if (r == 0 && s + kPre < KS && !(ABL & 8)) load_b(bt, s + kPre);
    int pa, pb;  // (model part, feature part), small terms first
    if (PREC) {
      pa = which == 0 ? 1 : 0;
      pb = which == 1 ? 1 : 0;
    } else {
      pa = which == 0 ? 2 : (which == 1 || which == 3) ? 1 : 0;
      pb = (which == 0 || which == 3 || which == 5) ? 0 : (which == 1 || which == 4) ? 1 : 2;
    }
#if AMLIKE_DIFF_GCLDS
    f32x16 c = get_tile(acc, nt);  // the chain starts from the gconsts read into acc (load_gc)
#else
    f32x16 c = (s == 0 && which == 0) ? get_tile(gcv, nt) : get_tile(acc, nt);
#endif
    if (PREC)
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[nt][s][pa]),
                                                 __builtin_bit_cast(f16x8, bf[s % NB][pb]), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nt][s][pa], bf[s % NB][pb], c, 0, 0, 0);
    set_tile(acc, nt, c);
  };
  auto load_b_head = [&](const uint4 *bt) {
#pragma unroll
    for (int s = 0; s < kPre && s < KS; ++s) load_b(bt, s);
  };

  accvec acc_cur, acc_alt;
  load_gc(acc_cur);
  load_b_head(s_b + lane);
#pragma unroll
  for (int n = 0; n < kMfma; ++n) mfma_n(acc_cur, s_b + lane, n);

  // The 0/1 factors of the running sums, one per accumulator register, fixed for the life
  // of the wave and held in scalar registers.
  float keepf[kSteps];
#pragma unroll
  for (int st = 0; st < kSteps; ++st) keepf[st] = scalar_bit_as_float(keepm, st);

  // Address (this lane's part) of staged feature tile `fn` in LDS.
  auto tile_ptr = [&](int fn) { return s_b + (((fn / FTB) & 1) * FTB + fn % FTB) * kTileU4 + lane; };

#ifdef AMLIKE_DIFF_STAMPS
  unsigned long long stamp_pass = 0, stamp_tail = 0, stamp_tiles = 0, stamp_vm = 0, stamp_bar = 0;
  const unsigned long long stamp_begin = AMLIKE_STAMP();
  const unsigned long long real_begin = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  auto stamps_out = [&]() {
    if (bx == ar.nbx / 2 && by == (int)(gridDim.x / ar.nbx) / 2 && lane == 0) {
      g_diff_stamps[w * 5 + 0] = stamp_pass;
      g_diff_stamps[w * 5 + 1] = stamp_tail;
      g_diff_stamps[w * 5 + 2] = AMLIKE_STAMP() - stamp_begin;
      g_diff_stamps[w * 5 + 3] = stamp_tiles;
      g_diff_stamps[w * 5 + 4] = __builtin_amdgcn_s_memrealtime() - real_begin;
      g_diff_stamps[40 + w] = (stamp_vm << 32) | stamp_bar;
    }
    if (bx == ar.nbx / 2 && by < 256 && lane == 0 && w == 0) {
      g_diff_blocks[by * 4 + 0] = real_entry;
      g_diff_blocks[by * 4 + 1] = __builtin_amdgcn_s_memrealtime() - real_entry;
      g_diff_blocks[by * 4 + 2] = AMLIKE_STAMP() - stamp_begin;
      g_diff_blocks[by * 4 + 3] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
    }
  };
#endif

  // The register pass of frame tile ft: the running sums of its scores (acc_c) on the vector
  // unit, with the matrix instructions of tile ft+1 (into acc_n) issued between them.  The
  // feature fragments of the first kPre k-steps (and the gconsts, where they come from LDS)
  // have been requested by the caller (preload), as early as the staging protocol allows.
  auto report_overflow = [&](int row, int col) {
    const int kk = ctl_add(ar.ctl, 1);
    if (kk < ar.fix_cap) ar.fix_list[kk] = make_int2(row, col);
  };

  // Stores value v of this lane for the frame tile that starts at frame t0 (voff: the lane's entry
  // of s_voff); a value out of range goes to the repair list.
  auto emit_at = [&](const int t0, float v, uint32_t voff) __attribute__((always_inline)) {
    char *tile_base = reinterpret_cast<char *>(ar.out) + (size_t)t0 * (size_t)ar.ldo * sizeof(float);
    if (voff != ~0u) {
      if (!(ABL & 1)) *reinterpret_cast<float *>(tile_base + voff) = v;
      if (!ABL && !(fabsf(v) <= kMaxAbsLogLike))
        report_overflow(t0 + j, (int)(voff / 4u - (uint32_t)j * (uint32_t)ar.ldo));
    }
  };
  accvec sv;
  auto pass_at = [&](const uint4 *bt, accvec &acc_c, accvec &acc_n) __attribute__((always_inline)) {
    float s_ = 0.0f;
#ifdef AMLIKE_DIFF_STAMPS
    const unsigned long long st0 = AMLIKE_STAMP();
#endif
#if AMLIKE_DIFF_PRIO == 1
    __builtin_amdgcn_s_setprio(2);
#elif AMLIKE_DIFF_PRIO == 4
    if (w >= W / 2) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(2);
#endif
    constexpr int skip = 0;
    // exp of step st+1 is issued before the fma of step st consumes the exp of step st: no
    // instruction waits on the result of the transcendental instruction just before it.
    float e_cur = (ABL & 2) ? 0.0f : fast_exp2(acc_c[0]);
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
#pragma unroll
      for (int n = (max(st - skip, 0) * kMfma) / (kSteps - skip);
           n < (max(st + 1 - skip, 0) * kMfma) / (kSteps - skip); ++n)
        mfma_n(acc_n, bt, n);
      // Nothing moves across: the matrix instructions stay spread over the vector work.
      __builtin_amdgcn_sched_barrier(0);
      if (ABL & 2) {
        asm volatile("" ::"v"(acc_c[st]));
      } else {
        float e_next = 0.0f;
        if (st + 1 < kSteps) e_next = fast_exp2(acc_c[st + 1]);
        s_ = fmaf(s_, keepf[st], e_cur);  // keepf[st]: does the sum continue into this register?
        // (pinned: the compiler would otherwise sink this arithmetic to where sv is read)
        pin_order(e_next);
        pin_order(s_);
        e_cur = e_next;
      }
      sv[st] = s_;
      __builtin_amdgcn_sched_barrier(0);
    }
#if AMLIKE_DIFF_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif AMLIKE_DIFF_PRIO == 4
    if (w >= W / 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
#ifdef AMLIKE_DIFF_STAMPS
    stamp_pass += AMLIKE_STAMP() - st0;
    ++stamp_tiles;
#endif
  };
  auto pass = [&](const int ft, accvec &acc_c, accvec &acc_n) __attribute__((always_inline)) {
    pass_at(tile_ptr(ft + 1), acc_c, acc_n);
  };

  // General tail: the segments of this wave's chunk for frame tile ft, two per iteration:
  // lanes 0-31 finish segment 2k and lanes 32-63 segment 2k+1 of the same 32 frames.  One
  // v_permlane32_swap of (value for a, value for b) leaves, in each half-wave, that half's
  // segment as seen by both halves of the original layout.
  auto tail_general = [&](const int ft, const accvec &acc_c) __attribute__((always_inline)) {
    const int t0 = t_begin + (ft << 5);
    const int row = t0 + j;
    float *orow = ar.out + (size_t)row * ar.ldo;
    int start_a = 0;
    for (int k = 0; k < npair; ++k) {
      const int end_a = __builtin_amdgcn_readlane(endreg, 2 * k);
      const int end_b = __builtin_amdgcn_readlane(endreg, 2 * k + 1);
      const int col_a = __builtin_amdgcn_readlane(colreg, 2 * k);
      const int col_b = __builtin_amdgcn_readlane(colreg, 2 * k + 1);
      const int start_b = min(end_a + 1, kSteps - 1);
      const auto ss = __builtin_amdgcn_permlane32_swap(__float_as_uint(sv[end_a]),
                                                       __float_as_uint(sv[end_b]), false, false);
      const auto xx = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc_c[start_a]),
                                                       __float_as_uint(acc_c[start_b]), false, false);
      // ss[0] / ss[1]: the sums of the even / odd slots; xx[0]: score of the reference
      // Gaussian (slot 0, natural-log units), xx[1]: first difference (slot 1).
      const float ref = __uint_as_float(xx[0]);
      const float S = (1.0f + fast_exp2(__uint_as_float(xx[1]))) +
                      (__uint_as_float(ss[0]) + __uint_as_float(ss[1]));
      const int col = h ? col_b : col_a;
      if (row < t_end) {
        if (col >= 0) {
          float v = fmaf(fast_log2(S), kLn2, ref);
          if (!(ABL & 1)) orow[col] = v;
          else pin_order(v);
          if (!ABL && !(fabsf(v) <= kMaxAbsLogLike)) report_overflow(row, col);
        } else if (col != kColDiscard && !(ABL & 1)) {  // a part of a split pdf
          const size_t o = (size_t)(-col - 2) * ar.part_ld + row;
          ar.part_m[o] = ref;
          ar.part_s[o] = S;
        }
      }
      start_a = min(end_b + 1, kSteps - 1);
    }
  };

  // Fast tail (whole frame tiles of chunks without split-pdf parts): two pairs of segments per
  // step, in straight-line code so that their dependent chains overlap; segment positions from
  // endw, output offsets from s_voff, stores relative to the tile's first output row.
  auto tail_fast_at = [&](const int t0, const accvec &acc_c) __attribute__((always_inline)) {
    // Keep the packed words opaque per tile: the field extraction (scalar unit, cheap) is redone
    // every tile instead of every field living in a scalar register of its own.
#if __HIP_DEVICE_COMPILE__
#pragma unroll
    for (int q = 0; q < (kMaxPairs + 2) / 3; ++q) asm volatile("" : "+s"(endw[q]));
#endif
    auto end_of = [&](int seg) -> int { return (endw[seg / 6] >> (5 * (seg % 6))) & 31; };
    auto gather = [&](int end_a, int end_b, int start_a, int start_b, float &s01, float &ref, float &x) {
      const auto ss = __builtin_amdgcn_permlane32_swap(__float_as_uint(sv[end_a]),
                                                       __float_as_uint(sv[end_b]), false, false);
      const auto xx = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc_c[start_a]),
                                                       __float_as_uint(acc_c[start_b]), false, false);
      s01 = __uint_as_float(ss[0]) + __uint_as_float(ss[1]);
      ref = __uint_as_float(xx[0]);
      x = __uint_as_float(xx[1]);
    };
    auto finish = [&](int end_a, int end_b, int start_a, int start_b) -> float {
      float s01, ref, x;
      gather(end_a, end_b, start_a, start_b, s01, ref, x);
      return fmaf(fast_log2((1.0f + fast_exp2(x)) + s01), kLn2, ref);
    };
    // Values and output offsets of step kk: pairs 2 kk and 2 kk + 1.
    auto step_values = [&](const int kk, float &v0, float &v1, uint32_t &voff0, uint32_t &voff1)
                           __attribute__((always_inline)) {
      const int ea0 = end_of(4 * kk), eb0 = end_of(4 * kk + 1);
      const int sa0 = kk == 0 ? 0 : min(end_of(4 * kk - 1) + 1, kSteps - 1);
      const int sb0 = min(ea0 + 1, kSteps - 1);
      const int ea1 = end_of(4 * kk + 2), eb1 = end_of(4 * kk + 3);
      const int sa1 = min(eb0 + 1, kSteps - 1), sb1 = min(ea1 + 1, kSteps - 1);
      voff0 = s_voff[(2 * kk) * 64 + lane];
      voff1 = s_voff[(2 * kk + 1) * 64 + lane];
      v0 = finish(ea0, eb0, sa0, sb0);
      v1 = finish(ea1, eb1, sa1, sb1);
    };
#pragma unroll
    for (int kk = 0; kk < (kMaxPairs + 1) / 2; ++kk) {
      if (2 * kk >= npair) break;
      float v0, v1;
      uint32_t voff0, voff1;
      step_values(kk, v0, v1, voff0, voff1);
      // both values exist before either store's control flow starts
      pin_order(v0);
      pin_order(v1);
      emit_at(t0, v0, voff0);
      emit_at(t0, v1, voff1);
    }
  };
  auto tail_fast = [&](const int ft, const accvec &acc_c) __attribute__((always_inline)) {
    tail_fast_at(t_begin + (ft << 5), acc_c);
  };

  auto tail = [&](const int ft, const accvec &acc_c) __attribute__((always_inline)) {
#ifdef AMLIKE_DIFF_STAMPS
    const unsigned long long st0 = AMLIKE_STAMP();
#endif
#if AMLIKE_DIFF_PRIO == 3
    __builtin_amdgcn_s_setprio(2);
#endif
    if (ABL & 4) {
      accvec keep = sv;
      pin_order(keep);
    } else if (fast_chunk && t_begin + (ft << 5) + 32 <= t_end) {
      tail_fast(ft, acc_c);
    } else {
      tail_general(ft, acc_c);
    }
#if AMLIKE_DIFF_PRIO == 3
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef AMLIKE_DIFF_STAMPS
    stamp_tail += AMLIKE_STAMP() - st0;
#endif
  };

  // What a pass needs first: the feature fragments of the first k-steps of tile fn and, where
  // they come from LDS, the gconsts in the accumulators the pass writes.
  auto preload = [&](const int fn, accvec &acc) __attribute__((always_inline)) {
    load_gc(acc);
    if (!(ABL & 8)) load_b_head(tile_ptr(fn));
  };

  if (AMLIKE_DIFF_PP && W == 8 && !(ABL & 16)) {
    // Ping-pong.  The two waves that share a SIMD (w and w + 4) alternate: while one is in its
    // register pass and owns the matrix pipe, the other finishes the segments of its previous
    // tile; a barrier separates the phases, and waves 4-7 run one phase behind waves 0-3.
    // Phase p: waves 0-3 pass(p/2) for even p, tail(p/2) for odd p; waves 4-7 the same for p-1.
    // Staged feature group G (tiles G*FTB ..) is first touched in phase 2*G*FTB - 3 (the fragment
    // preload of waves 0-3 for pass(G*FTB - 1)) and last in phase 2*G*FTB + 2*FTB - 3.
    // Tiles are taken in pairs (the two accumulator sets exchange roles); a tile past the end of
    // the block's frames computes on stale LDS contents and stores nothing.
    constexpr int PB = 2 * FTB;  // phases per staged group
    auto phase_sync = [&](const int p) __attribute__((always_inline)) {
      if ((p + 3) % PB == 0) wait_vector_memory();  // this wave's share of the next group has landed
#ifdef AMLIKE_DIFF_STAMPS
      const unsigned long long sb0 = AMLIKE_STAMP();
#endif
      __syncthreads();
#ifdef AMLIKE_DIFF_STAMPS
      stamp_bar += AMLIKE_STAMP() - sb0;
#endif
      if ((p + 2) % PB == 0) {
        const int G = (p + 2) / PB;
        if (G + 1 < ngroup) stage(G + 1);
      }
    };
    auto tail_if = [&](const int ft, const accvec &acc) __attribute__((always_inline)) {
      if (ft < ntile) tail(ft, acc);
    };
    const int npairs = (ntile + 1) >> 1;
    if (w < W / 2) {
      preload(1, acc_alt);
      for (int m = 0; m < npairs; ++m) {
        if (m > 0) phase_sync(4 * m);
        pass(2 * m, acc_cur, acc_alt);
        phase_sync(4 * m + 1);
        tail_if(2 * m, acc_cur);
        preload(2 * m + 2, acc_cur);
        phase_sync(4 * m + 2);
        pass(2 * m + 1, acc_alt, acc_cur);
        phase_sync(4 * m + 3);
        tail_if(2 * m + 1, acc_alt);
        preload(2 * m + 3, acc_alt);
      }
    } else {
      for (int m = 0; m < npairs; ++m) {
        if (m > 0) {
          phase_sync(4 * m);
          tail_if(2 * m - 1, acc_alt);
        }
        preload(2 * m + 1, acc_alt);
        phase_sync(4 * m + 1);
        pass(2 * m, acc_cur, acc_alt);
        phase_sync(4 * m + 2);
        tail_if(2 * m, acc_cur);
        preload(2 * m + 2, acc_cur);
        phase_sync(4 * m + 3);
        pass(2 * m + 1, acc_alt, acc_cur);
      }
      tail_if(2 * npairs - 1, acc_alt);
    }
#ifdef AMLIKE_DIFF_STAMPS
    stamps_out();
#endif
    return;
  }
  // Every wave runs pass and tail of a tile back to back; one barrier per staged group.  The two
  // waves of a SIMD (w and w + W/2) settle into taking turns on the matrix pipe -- one in its
  // pass while the other is in its tail -- so the barrier of a group sits half a tile apart in
  // their programs: waves of the first half meet it in front of the pass that first reads the
  // group, waves of the second half (AMLIKE_DIFF_STAGGER) behind the pass before that one, in
  // front of its tail.  By then every wave has finished the pass that last read the group which
  // the new staging loads overwrite.  Two copies of the tile body with the roles of the two
  // accumulator sets exchanged: no register copies between tiles.
  // (Both halves must pass the same number of barriers: the second half skips the one whose
  // tile the first half never starts, and with one tile per group there is no room to shift.)
  const bool late_half = AMLIKE_DIFF_STAGGER && W == 8 && FTB >= 2 && w >= W / 2;
  auto group_barrier = [&](const int g) __attribute__((always_inline)) {
#ifdef AMLIKE_DIFF_STAMPS
    const unsigned long long sb0 = AMLIKE_STAMP();
#endif
    __syncthreads();
#ifdef AMLIKE_DIFF_STAMPS
    stamp_bar += AMLIKE_STAMP() - sb0;
#endif
    if (g + 1 < ngroup) stage(g + 1);
  };
  // The first fragments of a pass are requested before the tail in front of it whenever their
  // tile has been published by then (always for the second half; for the first half unless the
  // tile opens a group), otherwise right before the pass.
  bool requested = false;
  auto tile_body = [&](const int ft, accvec &a_cur, accvec &a_next) __attribute__((always_inline)) {
    const int fn = ft + 1;
    if (fn % FTB == 0 && !(ABL & 16) && !late_half) group_barrier(fn / FTB);
    load_gc(a_next);
    if (!requested && !(ABL & 8)) load_b_head(tile_ptr(fn));
    pass(ft, a_cur, a_next);
    // The barrier publishes a staged group: see this wave's staging loads (issued FTB tiles ago)
    // complete here, before this tile's stores join the queue, rather than at the barrier
    // itself, where the wait would also cover those stores.
    if ((ft + 2) % FTB == 0 && !(ABL & 16)) {
#ifdef AMLIKE_DIFF_STAMPS
      const unsigned long long sv0 = AMLIKE_STAMP();
#endif
      wait_vector_memory();
#ifdef AMLIKE_DIFF_STAMPS
      stamp_vm += AMLIKE_STAMP() - sv0;
#endif
      if (late_half && ft + 1 < ntile) group_barrier((ft + 2) / FTB);
    }
#ifdef AMLIKE_EXTRA_SALU
    {
      int dummy = ft;
#pragma unroll
      for (int i = 0; i < AMLIKE_EXTRA_SALU; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(dummy) : : "scc");
      asm volatile("" ::"s"(dummy));
    }
#endif
    requested = AMLIKE_DIFF_EARLYLOAD && (late_half || (ft + 2) % FTB != 0);
    if (requested && !(ABL & 8)) load_b_head(tile_ptr(ft + 2));
    tail(ft, a_cur);
  };
  int ft_from = 0;
#if AMLIKE_DIFF_STATICGROUPS
  // Every group but a block's last holds FTB whole frame tiles and has a successor: for those the
  // tile's position in its group is spelled out (FTB copies of the tile body), so that where the
  // barrier, the wait for the staging loads and the fragment requests sit, and where the
  // fragments lie in LDS, cost no scalar arithmetic or branches per tile.  Same protocol as
  // tile_body; the block's last group runs through tile_body.
  if constexpr (FTB == 4 && W == 8 && ABL == 0) {
    if (ngroup > 1) {
      const uint4 *buf_cur = s_b + lane, *buf_next = s_b + FTB * kTileU4 + lane;  // this group's / the next group's tiles
      int t0 = t_begin;
      load_b_head(buf_cur + kTileU4);  // tile 1
      auto tail_at = [&](const accvec &acc_c) __attribute__((always_inline)) {
#ifdef AMLIKE_DIFF_STAMPS
        const unsigned long long st0 = AMLIKE_STAMP();
#endif
        if (fast_chunk) {
          tail_fast_at(t0, acc_c);
        } else {
          tail_general((t0 - t_begin) >> 5, acc_c);
        }
#ifdef AMLIKE_DIFF_STAMPS
        stamp_tail += AMLIKE_STAMP() - st0;
#endif
        t0 += 32;
      };
      for (int g = 0; g + 1 < ngroup; ++g) {
        // tile 0 of the group
        load_gc(acc_alt);
        pass_at(buf_cur + kTileU4, acc_cur, acc_alt);
        load_b_head(buf_cur + 2 * kTileU4);
        tail_at(acc_cur);
        // tile 1
        load_gc(acc_cur);
        pass_at(buf_cur + 2 * kTileU4, acc_alt, acc_cur);
        load_b_head(buf_cur + 3 * kTileU4);
        tail_at(acc_alt);
        // tile 2: its pass is the last to read this group's tiles
        load_gc(acc_alt);
        pass_at(buf_cur + 3 * kTileU4, acc_cur, acc_alt);
        wait_vector_memory();
        if (late_half) {
          group_barrier(g + 1);
          load_b_head(buf_next);
        }
        tail_at(acc_cur);
        // tile 3: its pass reads the next group's first tile
        if (!late_half) {
          group_barrier(g + 1);
          load_b_head(buf_next);
        }
        load_gc(acc_cur);
        pass_at(buf_next, acc_alt, acc_cur);
        load_b_head(buf_next + kTileU4);
        tail_at(acc_alt);
        const uint4 *swap = buf_cur;
        buf_cur = buf_next;
        buf_next = swap;
      }
      ft_from = (ngroup - 1) * FTB;
      requested = true;
    }
  }
#endif
  for (int ft = ft_from; ft < ntile; ft += 2) {
    tile_body(ft, acc_cur, acc_alt);
    if (ft + 1 >= ntile) break;
    tile_body(ft + 1, acc_alt, acc_cur);
  }
#ifdef AMLIKE_DIFF_STAMPS
  stamps_out();
#endif
}

__global__ void gmm_clear_ctl_kernel(int *ctl) {
  if (threadIdx.x < 2) ctl_store(ctl + threadIdx.x, 0);
}

// Recomputes the listed (frame, pdf) pairs in plain fp32 with the reference's own
// log-sum-exp (maximum first, then the sum); one wave per pair.
__global__ void __launch_bounds__(256)
    gmm_fix_kernel(const int *ctl, const int2 *fix_list, int fix_cap,
                   const float *__restrict__ gconsts, const float *__restrict__ miv,
                   const float *__restrict__ iv, const int32_t *__restrict__ offsets, int dim,
                   const float *__restrict__ feats, int ld, float *out, int ldo, int *flag) {
  if (ctl_load(ctl + 1) != 0) return;
  const int n = ctl_load(ctl);
  if (n > fix_cap) return;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  for (int e = wave; e < n; e += nwaves) {
    const int2 tp = fix_list[e];
    const float *x = feats + (size_t)tp.x * ld;
    const int g0 = offsets[tp.y], g1 = offsets[tp.y + 1];
    float m = -INFINITY;
    for (int g = g0 + lane; g < g1; g += 64) {
      float a1 = 0.0f, a2 = 0.0f;
      for (int d = 0; d < dim; ++d) {
        const float xd = x[d];
        a1 = fmaf(miv[(size_t)g * dim + d], xd, a1);
        a2 = fmaf(iv[(size_t)g * dim + d], xd * xd, a2);
      }
      m = fmaxf(m, gconsts[g] + a1 - 0.5f * a2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.0f;
    for (int g = g0 + lane; g < g1; g += 64) {
      float a1 = 0.0f, a2 = 0.0f;
      for (int d = 0; d < dim; ++d) {
        const float xd = x[d];
        a1 = fmaf(miv[(size_t)g * dim + d], xd, a1);
        a2 = fmaf(iv[(size_t)g * dim + d], xd * xd, a2);
      }
      const float f = gconsts[g] + a1 - 0.5f * a2;
      if (f - m >= -15.9423847198486328125f) sum += __expf(f - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) {
      const float v = m + __logf(sum);
      out[(size_t)tp.x * ldo + tp.y] = v;
      if (!(fabsf(v) <= kMaxAbsLogLike)) ctl_or(flag, 1);
    }
  }
}

// Merges the parts of pdfs that were split over several waves:
// out[t][col] = log sum_parts s_p exp(m_p).   One thread per (split pdf, frame).
struct SplitPdf {
  int32_t col, first_part, num_parts;
};

__global__ void gmm_merge_parts_kernel(const SplitPdf *splits, int num_splits,
                                       const float *part_m, const float *part_s, int part_ld,
                                       int T, float *out, int ldo, int *flag, int *ctl,
                                       int2 *fix_list, int fix_cap, int guard) {
  if (guard_skips(ctl, fix_cap, guard)) return;
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
  if (!(fabsf(v) <= kMaxAbsLogLike)) {
    if (guard == 1) {  // differenced kernel: recompute this pair exactly
      const int k = ctl_add(ctl, 1);
      if (k < fix_cap) fix_list[k] = make_int2(t, d.col);
    } else {
      ctl_or(flag, 1);
    }
  }
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

// The operands of the per-Gaussian score GEMM as bf16x3 planes (value = p1 + p2 + p3
// exactly), 8 consecutive k of one row per 16-byte element, k-octet-major: element
// (plane q, octet o, row r) at q * pstride + o * rows + r.  Row r of the k axis is
//   MODEL = false (features): [x_0 .. x_{D-1}, x_0^2 .. x_{D-1}^2, 0 ...]
//   MODEL = true  (Gaussians): [miv_0 .. miv_{D-1}, -0.5 iv_0 .. -0.5 iv_{D-1}, 0 ...]
// so that score = gconst + <model row, feature row> (gmm/diag-gmm.cc:546-562).
template <bool MODEL>
__global__ void __launch_bounds__(256)
    gmm_gemm_planes_kernel(const float *__restrict__ a, const float *__restrict__ b, int ld, int dim,
                           int rows, uint4 *__restrict__ out, size_t pstride, int kp8) {
  const size_t n = (size_t)rows * kp8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t oct = i / rows, r = i - oct * rows;
    unsigned short p[3][8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int c = (int)oct * 8 + jj;
      float v = 0.0f;
      if (c < dim) {
        v = a[r * ld + c];
      } else if (c < 2 * dim) {
        const float t = b[r * ld + (c - dim)];
        v = MODEL ? -0.5f * t : t * t;
      }
      const __bf16 b1 = static_cast<__bf16>(v);
      const float r1 = v - static_cast<float>(b1);
      const __bf16 b2 = static_cast<__bf16>(r1);
      const __bf16 b3 = static_cast<__bf16>(r1 - static_cast<float>(b2));
      p[0][jj] = __builtin_bit_cast(unsigned short, b1);
      p[1][jj] = __builtin_bit_cast(unsigned short, b2);
      p[2][jj] = __builtin_bit_cast(unsigned short, b3);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      uint4 v;
      v.x = p[q][0] | ((unsigned)p[q][1] << 16);
      v.y = p[q][2] | ((unsigned)p[q][3] << 16);
      v.z = p[q][4] | ((unsigned)p[q][5] << 16);
      v.w = p[q][6] | ((unsigned)p[q][7] << 16);
      out[q * pstride + i] = v;
    }
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

// Pruned log-sum-exp over each pdf's Gaussians (VectorBase::LogSumExp with prune > 0,
// /root/reference/src/matrix/kaldi-vector.cc:1060-1080): the maximum is found first, then
// only terms f >= max(max + log(FLT_EPSILON), max - prune) are summed.  One thread per
// (frame, pdf); scores is [T x ld] with the pdf's Gaussians in columns offsets[p]..offsets[p+1]-1.
__global__ void __launch_bounds__(256)
    gmm_pruned_lse_kernel(const float *__restrict__ scores, int ld,
                          const int32_t *__restrict__ offsets, int num_pdfs, int T, float prune,
                          float *__restrict__ out, int ldo, int *flag) {
  // (Staging the block's stretch of the row through LDS first measured slower: the walk of
  // each thread over its own pdf then conflicts on the LDS banks, while here it hits L1.)
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (p >= num_pdfs || t >= T) return;
  const float *row = scores + (size_t)t * ld;
  const int g0 = offsets[p], g1 = offsets[p + 1];
  float mx = -INFINITY;
  for (int g = g0; g < g1; ++g) mx = fmaxf(mx, row[g]);
  float cutoff = mx + (-15.9423847198486328125f);  // logf(FLT_EPSILON)
  if (prune > 0.0f && mx - prune > cutoff) cutoff = mx - prune;
  float sum = 0.0f;
  for (int g = g0; g < g1; ++g) {
    const float f = row[g];
    if (f >= cutoff) sum += __expf(f - mx);
  }
  const float v = mx + __logf(sum);
  out[(size_t)t * ldo + p] = v;
  if (!(fabsf(v) <= kMaxAbsLogLike)) ctl_or(flag, 1);
}


// DiagGmm::GaussianSelection (/root/reference/src/gmm/diag-gmm.cc:801-871) on precomputed
// per-Gaussian scores: one wave per frame; the frame's scores sit in registers (kGselKeep per
// lane, i.e. up to 64 * kGselKeep Gaussians), and num_gselect rounds of a wave-wide argmax pick
// the best remaining one.  Ties go to the larger index, as std::greater<pair<float,int>> does.
// out_idx: [T x num_gselect] (-1 where num_gselect > num_mix); out_ll[t] = log-sum of the
// selected scores, accumulated in pick order like the reference's LogAdd chain.
constexpr int kGselKeep = 64;
__global__ void __launch_bounds__(256)
    gmm_gselect_kernel(const float *__restrict__ scores, int lds, int num_mix, int T,
                       int num_gselect, int32_t *__restrict__ out_idx, float *__restrict__ out_ll) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float *f = scores + (size_t)t * lds;
  float v[kGselKeep];
#pragma unroll
  for (int k = 0; k < kGselKeep; ++k) {
    const int g = lane + 64 * k;
    v[k] = g < num_mix ? f[g] : -INFINITY;
  }
  float tot = -INFINITY;
  for (int j = 0; j < num_gselect; ++j) {
    // lane-local best (value, index); index -1 = nothing left
    float bv = -INFINITY;
    int bi = -1;
#pragma unroll
    for (int k = 0; k < kGselKeep; ++k) {
      const int g = lane + 64 * k;
      // v[k] > -inf is false for retired entries (NaN) and for disabled Gaussians (-inf)
      const bool take = g < num_mix && v[k] > -INFINITY && (bi < 0 || v[k] > bv || (v[k] == bv && g > bi));
      // a score of -inf (a disabled Gaussian) can still be selected once nothing else is left
      bv = take ? v[k] : bv;
      bi = take ? g : bi;
    }
    if (bi < 0) {  // only -inf scores (or nothing) left in this lane: offer the largest unused index
#pragma unroll
      for (int k = 0; k < kGselKeep; ++k) {
        const int g = lane + 64 * k;
        if (g < num_mix && v[k] == -INFINITY && g > bi) bi = g;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      const bool take = oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi > bi));
      bv = take ? ov : bv;
      bi = take ? oi : bi;
    }
    if (lane == 0) out_idx[(size_t)t * num_gselect + j] = bi;
    if (bi < 0) continue;
    // retire the winner: NaN marks "used" (a NaN never compares greater or equal)
#pragma unroll
    for (int k = 0; k < kGselKeep; ++k)
      if (lane + 64 * k == bi) v[k] = __builtin_nanf("");
    // LogAdd(tot, bv)
    float x = tot, diff;
    if (x < bv) {
      diff = x - bv;
      x = bv;
    } else {
      diff = bv - x;
    }
    tot = diff >= -15.9423847198486328125f ? x + log1pf(expf(diff)) : x;
  }
  if (lane == 0) out_ll[t] = tot;
}

// ---------------------------------------------------------------------------
// Host side: the packed model.

typedef void (*ExpandFn)(const float *, int, int, int, uint4 *, int, const float *, int *, int, int);
struct Variant {
  int mode;  // 0: exact-fp32 MFMA kernel, 1: bf16x3 kernel, 3: differenced kernel (+ bf16x3 fallback)
  int D;     // mode 0: feature dimension; mode >= 1: KS = ceil(dim / 8) MFMA k-steps
  int NT, W;
  void (*kernel)(GmmPackView, ScoreArgs);
  ExpandFn expand;  // feature prepass of `kernel`
  const char *name;
  int prec = 0;  // mode 3: 1 = fp16x2 operands, 0 = bf16x3 operands
  void (*safe_kernel)(GmmPackView, ScoreArgs) = nullptr;  // mode 3: wide-range / exact fallback
  ExpandFn safe_expand = nullptr;
};

#define AMLIKE_VARIANT(D, NT, W) \
  { 0, D, NT, W, gmm_score_kernel<D, NT, W>, nullptr, "gmm_score_kernel<" #D "," #NT "," #W ">" }
#define AMLIKE_VARIANT_H(KS, NT, W, OCC)                                                  \
  { 1, KS, NT, W, gmm_score_bf16x3_kernel<KS, NT, W, OCC>, gmm_expand_feats_kernel<KS>,   \
    "gmm_score_bf16x3_kernel<" #KS "," #NT "," #W "," #OCC ">" }
#define AMLIKE_VARIANT_D(KS, NT, W, OCC, PREC)                                               \
  { 3, KS, NT, W, gmm_score_diff_kernel<KS, NT, W, OCC, PREC>,                                \
    PREC ? gmm_expand_feats_f16_kernel<KS> : gmm_expand_feats_kernel<KS>,                     \
    PREC ? "gmm_score_diff_kernel<" #KS "," #NT "," #W "," #OCC ",f16x2>"                     \
         : "gmm_score_diff_kernel<" #KS "," #NT "," #W "," #OCC ",bf16x3>",                   \
    PREC, gmm_score_bf16x3_kernel<KS, NT, W, OCC>, gmm_expand_feats_kernel<KS> }
#define AMLIKE_VARIANT_DABL(KS, NT, W, OCC, PREC, ABL)                                       \
  { 300 + ABL, KS, NT, W, gmm_score_diff_kernel<KS, NT, W, OCC, PREC, ABL>,                   \
    PREC ? gmm_expand_feats_f16_kernel<KS> : gmm_expand_feats_kernel<KS>,                     \
    "gmm_score_diff_kernel<" #KS "," #NT "," #W "," #OCC "," #PREC ",ABL=" #ABL ">",          \
    PREC, gmm_score_bf16x3_kernel<KS, NT, W, OCC>, gmm_expand_feats_kernel<KS> }
#define AMLIKE_VARIANT_ABL(KS, NT, W, OCC, ABL)                                           \
  { 100 + ABL, KS, NT, W, gmm_score_bf16x3_kernel<KS, NT, W, OCC, ABL>,                   \
    gmm_expand_feats_kernel<KS>, "gmm_score_bf16x3_kernel<" #KS "," #NT "," #W "," #OCC ",ABL=" #ABL ">" }

// Register-stationary instantiations; the first match in this order is the default.
// bf16x3: KS = ceil(dim / 8) k-steps, NT*KS*12 model registers per lane, two waves per SIMD;
//         covers every feature dimension up to 64.
// f32:    exact-fp32 MFMA, NT*D model registers per lane (AMLIKE_GMM_MODE=f32), D = 39 / 40.
static const Variant kVariants[] = {
    AMLIKE_VARIANT_D(5, 2, 8, 2, 1), AMLIKE_VARIANT_D(5, 2, 8, 2, 0),
    AMLIKE_VARIANT_D(5, 2, 4, 2, 1), AMLIKE_VARIANT_D(5, 2, 4, 2, 0),
    AMLIKE_VARIANT_D(1, 2, 8, 2, 1), AMLIKE_VARIANT_D(1, 2, 8, 2, 0),
    AMLIKE_VARIANT_D(2, 2, 8, 2, 1), AMLIKE_VARIANT_D(2, 2, 8, 2, 0),
    AMLIKE_VARIANT_D(3, 2, 8, 2, 1), AMLIKE_VARIANT_D(3, 2, 8, 2, 0),
    AMLIKE_VARIANT_D(4, 2, 8, 2, 1), AMLIKE_VARIANT_D(4, 2, 8, 2, 0),
    AMLIKE_VARIANT_D(6, 1, 8, 2, 1), AMLIKE_VARIANT_D(6, 1, 8, 2, 0),
    AMLIKE_VARIANT_D(7, 1, 8, 2, 1), AMLIKE_VARIANT_D(7, 1, 8, 2, 0),
    AMLIKE_VARIANT_D(8, 1, 8, 2, 1), AMLIKE_VARIANT_D(8, 1, 8, 2, 0),
    AMLIKE_VARIANT_H(5, 2, 8, 2), AMLIKE_VARIANT_H(5, 2, 4, 2),
    AMLIKE_VARIANT_H(1, 2, 8, 2), AMLIKE_VARIANT_H(2, 2, 8, 2), AMLIKE_VARIANT_H(3, 2, 8, 2),
    AMLIKE_VARIANT_H(4, 2, 8, 2), AMLIKE_VARIANT_H(6, 1, 8, 2), AMLIKE_VARIANT_H(7, 1, 8, 2),
    AMLIKE_VARIANT_H(8, 1, 8, 2),
#ifdef AMLIKE_ABLATION_BUILD
    AMLIKE_VARIANT_ABL(5, 2, 8, 2, 1), AMLIKE_VARIANT_ABL(5, 2, 8, 2, 6), AMLIKE_VARIANT_ABL(5, 2, 8, 2, 14),
    AMLIKE_VARIANT_ABL(5, 2, 8, 2, 30), AMLIKE_VARIANT_ABL(5, 2, 8, 2, 4),
    AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 1), AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 9),
    AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 2), AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 4),
    AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 6), AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 14),
    AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 30), AMLIKE_VARIANT_DABL(5, 2, 8, 2, 1, 8),
#endif
    AMLIKE_VARIANT(40, 3, 8), AMLIKE_VARIANT(39, 3, 8),
    AMLIKE_VARIANT(40, 6, 4), AMLIKE_VARIANT(39, 6, 4),
};

struct Piece {
  int32_t pdf, g_begin, count, col;
};

}  // namespace amlike

using namespace amlike;

// Built-in scratch memory of amlike_gmm_score_dev_noalloc: one array per GPU the library runs
// on, held until the process ends.  In host code the name is a stand-in, not a device address;
// SetupNoAllocScratch asks the runtime for the address on the current device once and
// everything else uses that pointer.
constexpr size_t kNoAllocScratchBytes = (size_t)8 << 20;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
constexpr int kMaxDevices = 64;
static unsigned char *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mu;

namespace amlike {
int SetupNoAllocScratch(int device) {
  if (device < 0 || device >= kMaxDevices) {
    SetError("amlike_init: device index %d beyond the %d this library keeps scratch memory for", device,
             kMaxDevices);
    return AMLIKE_EINVAL;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[device]) return AMLIKE_OK;
  void *ptr = nullptr;
  AMLIKE_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));  // the calling thread's current device
  g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
  return AMLIKE_OK;
}
}  // namespace amlike

namespace amlike {
// Working memory of the device-pointer entry points for the calls made with one stream: the
// differenced kernels' control words, and growable buffers for the operand images, the partial
// sums of split pdfs, the repair list, padded features and per-Gaussian scores.
// The launch sequence of one small scoring call (feature expansion, scoring kernel, repair,
// wide-range twin, merge of split pdfs) as an instantiated HIP graph, for callers that score
// the same buffers again and again (a decoder's chunk loop): one graph launch instead of five
// kernel launches and a memset.
struct ScoreGraph {
  const float *d_feats = nullptr;
  float *d_out = nullptr;
  int T = 0, ld = 0, ldo = 0;
  int seen = 0;        // calls with this key so far
  bool failed = false; // capture or instantiation failed once: not tried again
  uint64_t last_use = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  void Free() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    exec = nullptr;
    graph = nullptr;
  }
};
constexpr int kScoreGraphs = 4;         // per stream
constexpr int kScoreGraphMaxFrames = 4096;

constexpr size_t kCtlBlockBytes = 256;
constexpr int kLaneFlagAt = kCtlBlockBytes / sizeof(int);
struct StreamLane {
  hipStream_t stream = nullptr;
  uint64_t last_use = 0;
  // [0] overflow count, [1] range flag: cleared at the start of every launch (gmm_clear_ctl_kernel);
  // [kLaneFlagAt], in a 256-byte block of its own, a result was NaN/Inf: read and cleared by amlike_gmm_poll
  int *d_ctl = nullptr;
  DeviceBuffer ws_part, ws_pad, ws_expand, ws_expand_safe, ws_fix, ws_gauss, ws_xplanes;
  ScoreGraph graphs[kScoreGraphs];
  void DropGraphs() {
    for (ScoreGraph &sg : graphs) {
      sg.Free();
      sg = ScoreGraph();
    }
  }
  void Free() {
    DropGraphs();
    (void)hipFree(d_ctl);
    d_ctl = nullptr;
    ws_part.Free();
    ws_pad.Free();
    ws_expand.Free();
    ws_expand_safe.Free();
    ws_fix.Free();
    ws_gauss.Free();
    ws_xplanes.Free();
  }
};
constexpr size_t kMaxStreamLanes = 8;
}  // namespace amlike

struct amlike_gmm {
  int32_t num_pdfs = 0, num_gauss = 0, dim = 0;
  int device = 0;
  const Variant *variant = nullptr;
  int num_wc = 0;        // wave-chunks, multiple of W
  int num_partials = 0;  // parts of pdfs split over several waves
  int num_splits = 0;
  double fill = 0.0;  // used slots / allocated slots
  // device arrays
  void *d_apack = nullptr;
  void *d_apack_safe = nullptr;  // differenced kernel: the plain bf16x3 image for the fallback
  float *d_bscale = nullptr;     // fp16x2 operands: per k-row feature scale [2][KS * 8]
  float *d_gcpack = nullptr;
  float *d_gcpack_safe = nullptr;  // differenced kernels: the plain gconsts for the fallback
  uint32_t *d_endmask = nullptr;
  int32_t *d_segcol = nullptr;
  SplitPdf *d_splits = nullptr;
  float *d_gconsts = nullptr, *d_miv = nullptr, *d_iv = nullptr;  // flat copies [G], [G x dim]
  int *d_flag = nullptr;
  int32_t *d_offsets = nullptr;  // [num_pdfs + 1]
  std::vector<int32_t> pdf_offsets;
  hipStream_t stream = nullptr;  // used by the host-pointer entry points
  hipStream_t copy_stream = nullptr;  // submit / wait: the copies out, behind the kernels
  AsyncSlot slots[kAsyncSlots];
  int32_t next_ticket = 0;
  bool use_graphs = false;  // AMLIKE_GMM_GRAPHS=1 at creation: small repeated calls replay a recorded graph
  // Tuning aids, read from the environment once, when the handle is created:
  // AMLIKE_GMM_FPB_TARGET (the frames-per-block figure the launch code aims at), AMLIKE_GMM_FPB
  // (exactly this many frames per block), AMLIKE_GMM_SOLO (one block per compute unit).
  int fpb_target = 4096, fpb_exact = 0;
  bool solo = false;
  bool profile = false;  // amlike_gmm_profile_kernel: event pairs around the scoring kernel
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
  size_t kev_used = 0;
  void *d_wplanes = nullptr;  // per-Gaussian score GEMM: the model as bf16x3 planes (built on first use)
  DeviceBuffer ws_raw;  // un-spliced features + speaker transform of amlike_gmm_score_raw
  DeviceBuffer ws_feats, ws_out, ws_gsel;  // host-pointer entry points (under mu)
  std::mutex mu;  // the host-pointer entry points, one call at a time
  // The device-pointer entry points keep one set of working memory per stream they are called
  // with, so that calls on different streams of one handle may run on the GPU at the same time.
  std::mutex lane_mu;  // lanes, d_wplanes, info; held while a call enqueues its work
  std::vector<std::unique_ptr<amlike::StreamLane>> lanes;
  uint64_t lane_clock = 0;
  std::string info;
};

namespace amlike {

// AMLIKE_GMM_MODE = "diff-f16x2" | "diff-bf16x3" | "bf16x3" | "f32" and
// AMLIKE_GMM_VARIANT = "<NT>x<W>" narrow the choice (tuning / A-B aids); by default the
// first matching entry of kVariants is taken.  f16_ok: the model fits the fp16x2 operands.
static const Variant *ChooseVariant(int dim, bool f16_ok) {
  const char *force = getenv("AMLIKE_GMM_VARIANT");
  const char *mode = getenv("AMLIKE_GMM_MODE");
  for (const Variant &v : kVariants) {
    if (v.mode == 0 && v.D != dim) continue;
    if (v.mode >= 1 && v.D != (dim + 7) / 8) continue;
    if ((v.mode == 3 || v.mode >= 300) && v.prec == 1 && !f16_ok) continue;
    const char *vname = v.mode == 0 ? "f32" : v.mode == 1 ? "bf16x3" : (v.prec ? "diff-f16x2" : "diff-bf16x3");
    if (v.mode >= 100) {  // ablation instantiations: only by explicit request "abl<N>" / "dabl<N>"
      char want[32];
      if (v.mode >= 300) snprintf(want, sizeof(want), "dabl%d", v.mode - 300);
      else snprintf(want, sizeof(want), "abl%d", v.mode - 100);
      if (!mode || strcmp(mode, want) != 0) continue;
    } else if (mode && strcmp(mode, vname) != 0) {
      continue;
    }
    if (force) {
      char buf[32];
      snprintf(buf, sizeof(buf), "%dx%d", v.NT, v.W);
      if (strcmp(buf, force) != 0) continue;
    }
    return &v;
  }
  return nullptr;
}

// x ~= p[0] + p[1] to 22 bits (two fp16); exact powers of two scale it beforehand.
static inline void SplitF16x2(float x, uint16_t p[2]) {
  const _Float16 h1 = static_cast<_Float16>(x);
  const _Float16 h2 = static_cast<_Float16>(x - static_cast<float>(h1));
  memcpy(&p[0], &h1, 2);
  memcpy(&p[1], &h2, 2);
  p[1] &= (uint16_t)AMLIKE_F16_LO_MASK;
}

// Chooses, for every k-row of the contraction (kind 0: miv_d * x_d, kind 1: -0.5 iv_d * x_d^2),
// a power-of-two scale that balances the largest model value `amax` (already in log2 units)
// against the largest feature value the model itself makes plausible (xr_d = |mean| + 8 sigma
// over all Gaussians), so that both operands of the fp16x2 kernels stay far from the ends of
// the fp16 range.  Returns false when some row cannot be balanced into [0, 16384]: the model
// then runs on the bf16x3 operands.
static bool ComputeF16Scales(const std::vector<double> &amax, const std::vector<double> &xr, int D,
                             int KS, std::vector<float> *ascale, std::vector<float> *bscale) {
  ascale->assign((size_t)2 * KS * 8, 1.0f);
  bscale->assign((size_t)2 * KS * 8, 1.0f);
  for (int d = 0; d < D; ++d) {
    for (int kind = 0; kind < 2; ++kind) {
      const double a = amax[(size_t)kind * D + d];
      const double bmax = kind ? xr[d] * xr[d] : xr[d];
      if (!std::isfinite(a) || !std::isfinite(bmax)) return false;
      int e = 0;
      if (a > 0.0 && bmax > 0.0) e = (int)std::lround(0.5 * (std::log2(bmax) - std::log2(a)));
      e = std::max(-60, std::min(60, e));
      if (std::ldexp(a, e) > 16384.0 || std::ldexp(bmax, -e) > 16384.0) return false;
      (*ascale)[(size_t)kind * KS * 8 + d] = std::ldexp(1.0f, e);
      (*bscale)[(size_t)kind * KS * 8 + d] = std::ldexp(1.0f, -e);
    }
  }
  return true;
}

// fn(i) for i in [0, n) on up to 16 host threads (contiguous ranges).
template <typename F>
static void ParallelFor(int n, F fn, int grain = 64) {
  const int nt = std::max(1, std::min({16, (int)std::thread::hardware_concurrency(), n / grain}));
  if (nt <= 1) {
    for (int i = 0; i < n; ++i) fn(i);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([=, &fn]() {
      for (int i = (int)((long long)n * t / nt); i < (int)((long long)n * (t + 1) / nt); ++i) fn(i);
    });
  for (std::thread &th : pool) th.join();
}

static int Upload(void **dst, const void *src, size_t bytes) {
  AMLIKE_HIP(hipMalloc(dst, bytes));
  AMLIKE_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return AMLIKE_OK;
}

// Packs the model for g->variant (and may replace g->variant by its bf16x3 twin when the
// model does not fit the fp16x2 operands).
//
// A wave-chunk is NT tiles x 32 slots.  Every pdf takes an even number of
// consecutive slots (odd sizes get one slot with gconst = -inf), so that a pdf
// boundary always falls between two accumulator registers: slot n of a tile
// lives in register n>>1 of half-wave n&1, i.e. in matrix row
//   r(n) = (i & 3) + 8 (i >> 2) + 4 (n & 1),  i = n >> 1.
// pdfs are laid out in order; when the next one does not fit in what is left
// of a chunk, a short look-ahead picks a later one that does.  A pdf larger
// than a whole chunk is cut into parts whose (max,sum) are merged afterwards.
//
// Within a pdf (or part) the live Gaussians come first, broadest first, then the disabled
// ones (gconst = -inf).  The differenced kernels get a second image in which slot 0 of
// every part keeps the first Gaussian as it is and every other slot holds
// (Gaussian - first Gaussian), gconst and operands alike.
static int PackModel(amlike_gmm *g, const float *gconsts, const float *miv, const float *iv,
                     const int32_t *offs) {
  const Variant *vp = g->variant;
  const int D = g->dim, NT = vp->NT, W = vp->W, DQ = (D + 3) / 4, KS = (D + 7) / 8;
  const int vmode = vp->mode >= 300 ? 3 : vp->mode >= 100 ? 1 : vp->mode;
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
  // Order of the Gaussians inside each piece (index = g_begin + position).
  std::vector<float> logdet(g->num_gauss);  // sum_d log(inv var): smaller = broader
  for (int gi = 0; gi < g->num_gauss; ++gi) {
    double acc = 0.0;
    for (int d = 0; d < D; ++d) acc += std::log(std::max(iv[(size_t)gi * D + d], 1e-30f));
    logdet[gi] = (float)acc;
  }
  std::vector<int> order(g->num_gauss);
  for (const Piece &p : pieces) {
    int lo = 0, hi = p.count;
    int *o = &order[p.g_begin];
    for (int q = 0; q < p.count; ++q)
      if (std::isfinite(gconsts[p.g_begin + q])) o[lo++] = p.g_begin + q;
    for (int q = p.count - 1; q >= 0; --q)
      if (!std::isfinite(gconsts[p.g_begin + q])) o[--hi] = p.g_begin + q;
    std::stable_sort(o, o + lo, [&](int x, int y) { return logdet[x] < logdet[y]; });
  }
  // Operand value of Gaussian gi, kind (0: miv, 1: -0.5 iv), dimension d; `ref` >= 0:
  // differenced against Gaussian ref, in log2 units.  ref < 0: the Gaussian as it is, in log2
  // units, except in the differenced image, where it is a segment's reference and stays in
  // natural-log units (the kernel adds it to ln(sum) unscaled).  Disabled Gaussians have zero
  // operands.
  const double raw_unit = vmode == 3 ? 1.0 : (double)kLog2e;
  auto opval = [&](int gi, int ref, int kind, int d) -> float {
    if (!std::isfinite(gconsts[gi])) return 0.0f;
    auto raw = [&](int x) -> double {
      return kind ? -0.5 * (double)iv[(size_t)x * D + d] : (double)miv[(size_t)x * D + d];
    };
    return (float)(ref >= 0 ? (raw(gi) - raw(ref)) * (double)kLog2e : raw(gi) * raw_unit);
  };
  auto gcval = [&](int gi, int ref) -> float {
    if (!std::isfinite(gconsts[gi])) return kPadGconst;
    const double v = ref >= 0 ? ((double)gconsts[gi] - (double)gconsts[ref]) * (double)kLog2e
                              : (double)gconsts[gi] * raw_unit;
    return std::max((float)v, kPadGconst);
  };
  const bool want_diff = vmode == 3;
  // Slots a piece of `count` Gaussians occupies: an even number; the differenced kernels
  // want at least two registers (four slots) per segment, because the running sum of a
  // segment's first register holds the reference's own term and is never read.
  auto slots_of = [&](int count) { return std::max(want_diff ? 4 : 2, (count + 1) & ~1); };
  bool f16 = vmode == 3 && vp->prec == 1;
  std::vector<float> ascale, bscale;
  if (f16) {
    std::vector<double> amax((size_t)2 * D, 0.0), xr(D, 0.0);
    bool ok = true;
    {
      // per-range maxima on host threads, then reduced
      const int nr = 16;
      std::vector<std::vector<double>> pa(nr, std::vector<double>((size_t)2 * D, 0.0)),
          px(nr, std::vector<double>(D, 0.0));
      std::vector<char> pok(nr, 1);
      ParallelFor(nr, [&](int t) {
        std::vector<double> &am = pa[t], &xm = px[t];
        for (size_t pi = pieces.size() * t / nr; pi < pieces.size() * (t + 1) / nr && pok[t]; ++pi) {
          const Piece &p = pieces[pi];
          const int ref = order[p.g_begin];
          for (int q = 0; q < p.count && pok[t]; ++q) {
            const int gi = order[p.g_begin + q];
            if (!std::isfinite(gconsts[gi])) continue;
            for (int d = 0; d < D; ++d) {
              const double v = iv[(size_t)gi * D + d], m = miv[(size_t)gi * D + d];
              if (!(v > 0.0) || !std::isfinite(v) || !std::isfinite(m)) {
                pok[t] = 0;
                break;
              }
              xm[d] = std::max(xm[d], std::fabs(m / v) + 8.0 / std::sqrt(v));
              for (int kind = 0; kind < 2; ++kind) {
                const float a = opval(gi, (want_diff && q > 0) ? ref : -1, kind, d);
                am[(size_t)kind * D + d] = std::max(am[(size_t)kind * D + d], (double)std::fabs(a));
              }
            }
          }
        }
      }, 1);
      for (int t = 0; t < nr; ++t) {
        ok = ok && pok[t];
        for (size_t i = 0; i < amax.size(); ++i) amax[i] = std::max(amax[i], pa[t][i]);
        for (int d = 0; d < D; ++d) xr[d] = std::max(xr[d], px[t][d]);
      }
    }
    ok = ok && ComputeF16Scales(amax, xr, D, KS, &ascale, &bscale);
    if (!ok) {  // same kernel family on bf16x3 operands
      const Variant *alt = nullptr;
      for (const Variant &c : kVariants)
        if (c.mode == vp->mode && c.D == vp->D && c.NT == NT && c.W == W && c.prec == 0) alt = &c;
      if (!alt) {
        SetError("amlike_gmm_create: no bf16x3 variant to fall back to");
        return AMLIKE_EINVAL;
      }
      g->variant = vp = alt;
      f16 = false;
    }
  }
  const Variant &v = *vp;

  // Assign pieces to chunks.  A chunk takes the first piece not yet placed and, of the kLook
  // pieces behind it, the set that fills the rest of the chunk best (subset sum over their slot
  // counts; of two sets with the same sum the one found first, i.e. made of nearer pieces).
  // Pieces stay in pdf order inside a chunk, and a chunk's pdfs lie within kLook of each other:
  // neighbouring output columns are still written by neighbouring chunks.
  std::vector<std::vector<Piece>> chunks;
  {
    const int kLook = 48;
    std::vector<char> used(pieces.size(), 0);
    size_t head = 0;
    std::vector<int> via_piece(cap + 1), via_sum(cap + 1);
    while (true) {
      while (head < pieces.size() && used[head]) ++head;
      if (head >= pieces.size()) break;
      used[head] = 1;
      const int left = cap - slots_of(pieces[head].count);
      // via_piece[s]: the piece that first reached sum s (-1: not reached), via_sum[s]: from which sum
      std::fill(via_piece.begin(), via_piece.end(), -1);
      std::vector<char> reached(cap + 1, 0);
      reached[0] = 1;
      for (size_t k = head + 1; k < pieces.size() && k < head + kLook; ++k) {
        if (used[k]) continue;
        const int need = slots_of(pieces[k].count);
        for (int sum = left - need; sum >= 0; --sum)
          if (reached[sum] && !reached[sum + need]) {
            reached[sum + need] = 1;
            via_piece[sum + need] = (int)k;
            via_sum[sum + need] = sum;
          }
      }
      int best = left;
      while (!reached[best]) --best;
      std::vector<int> taken;
      for (int sum = best; sum > 0; sum = via_sum[sum]) taken.push_back(via_piece[sum]);
      std::sort(taken.begin(), taken.end());
      std::vector<Piece> cur;
      cur.push_back(pieces[head]);
      for (int k : taken) {
        used[k] = 1;
        cur.push_back(pieces[k]);
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

  // Operand images.  fp32: [wc][nt][dq][lane][4].  16-bit: [wc][nt][s][part][lane][8], lane
  // (r, h): h = 0 -> miv, h = 1 -> -0.5 iv, d = 8 s + jj, values in log2 units.
  //   plain image  (apack_h, bf16x3 / apack fp32, with gcpack):  every Gaussian as it is;
  //   kernel image (apack_k with gcpack_k): what v.kernel reads when that is not the plain
  //   image: fp16x2 and / or differenced.
  const bool plain16 = vmode >= 1;
  const bool kernel_image = f16 || want_diff;
  const int KP = f16 ? 2 : 3;
  std::vector<float> apack;
  std::vector<uint16_t> apack_h, apack_k;
  if (vmode == 0) apack.assign((size_t)num_wc * NT * DQ * 64 * 4, 0.0f);
  if (plain16) apack_h.assign((size_t)num_wc * NT * KS * 3 * 64 * 8, 0);
  if (kernel_image) apack_k.assign((size_t)num_wc * NT * KS * KP * 64 * 8, 0);
  std::vector<float> gcpack((size_t)num_wc * NT * 32,
                            vmode == 0 ? -std::numeric_limits<float>::infinity() : kPadGconst);
  std::vector<float> gcpack_k;
  if (kernel_image) gcpack_k.assign((size_t)num_wc * NT * 32, kPadGconst);
  std::vector<uint32_t> endmask((size_t)num_wc * NT, 0u);
  std::vector<int32_t> segcol((size_t)num_wc * NT * 16, kColDiscard);
  // Every wave chunk fills its own stretch of the images: spread them over host threads.
  ParallelFor(num_wc, [&](int wc) {
    int slot = 0, seg = 0;
    for (const Piece &p : chunks[wc]) {
      const int ref = order[p.g_begin];
      for (int q = 0; q < p.count; ++q) {
        const int n = slot + q, nt = n >> 5, nn = n & 31, i = nn >> 1, hh = nn & 1;
        const int r = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const int gi = order[p.g_begin + q];
        const size_t gcidx = ((size_t)wc * NT + nt) * 32 + hh * 16 + i;
        if (vmode == 0) {
          gcpack[gcidx] = gconsts[gi];
          for (int d = 0; d < D; ++d) {
            const size_t base = ((((size_t)wc * NT + nt) * DQ + (d >> 2)) * 64) * 4 + (d & 3);
            apack[base + (size_t)r * 4] = miv[(size_t)gi * D + d];
            apack[base + (size_t)(r + 32) * 4] = -0.5f * iv[(size_t)gi * D + d];
          }
          continue;
        }
        gcpack[gcidx] = std::max(gconsts[gi] * kLog2e, kPadGconst);
        const int kref = (want_diff && q > 0) ? ref : -1;
        if (kernel_image) gcpack_k[gcidx] = gcval(gi, kref);
        for (int d = 0; d < D; ++d) {
          const int s = d >> 3, jj = d & 7;
          uint16_t pm[3], pv[3];
          SplitBf16x3(miv[(size_t)gi * D + d] * kLog2e, pm);
          SplitBf16x3(-0.5f * iv[(size_t)gi * D + d] * kLog2e, pv);
          for (int part = 0; part < 3; ++part) {
            const size_t base = (((((size_t)wc * NT + nt) * KS + s) * 3 + part) * 64) * 8 + jj;
            apack_h[base + (size_t)r * 8] = pm[part];
            apack_h[base + (size_t)(r + 32) * 8] = pv[part];
          }
          if (!kernel_image) continue;
          float km = opval(gi, kref, 0, d), kv = opval(gi, kref, 1, d);
          uint16_t qm[3], qv[3];
          if (f16) {
            SplitF16x2(km * ascale[d], qm);
            SplitF16x2(kv * ascale[(size_t)KS * 8 + d], qv);
          } else {
            SplitBf16x3(km, qm);
            SplitBf16x3(kv, qv);
          }
          for (int part = 0; part < KP; ++part) {
            const size_t base = (((((size_t)wc * NT + nt) * KS + s) * KP + part) * 64) * 8 + jj;
            apack_k[base + (size_t)r * 8] = qm[part];
            apack_k[base + (size_t)(r + 32) * 8] = qv[part];
          }
        }
      }
      slot += slots_of(p.count);
      const int last_reg = (slot - 2) >> 1;  // register (chunk-wide index) that closes it
      endmask[(size_t)wc * NT + (last_reg >> 4)] |= 1u << (last_reg & 15);
      segcol[(size_t)wc * NT * 16 + seg++] = p.col;
    }
    if (slot < cap) {
      // Trailing padding: all -inf.  The plain kernels let it join the last segment when
      // the chunk is not empty (the closing bit moves to the chunk's last register); the
      // differenced kernels keep it as a discarded segment of its own, since its slots
      // are not differences against that segment's reference.
      if (seg > 0 && !want_diff) {
        const int last_reg = (slot - 2) >> 1;
        endmask[(size_t)wc * NT + (last_reg >> 4)] &= ~(1u << (last_reg & 15));
      }
      endmask[(size_t)wc * NT + NT - 1] |= 1u << 15;
    }
  });

  int rc = AMLIKE_OK;
  if (vmode == 0) {
    rc = Upload(&g->d_apack, apack.data(), apack.size() * sizeof(float));
    if (rc == AMLIKE_OK) rc = Upload((void **)&g->d_gcpack, gcpack.data(), gcpack.size() * sizeof(float));
  } else if (kernel_image) {
    rc = Upload(&g->d_apack, apack_k.data(), apack_k.size() * sizeof(uint16_t));
    if (rc == AMLIKE_OK) rc = Upload((void **)&g->d_gcpack, gcpack_k.data(), gcpack_k.size() * sizeof(float));
    if (rc == AMLIKE_OK) rc = Upload(&g->d_apack_safe, apack_h.data(), apack_h.size() * sizeof(uint16_t));
    if (rc == AMLIKE_OK)
      rc = Upload((void **)&g->d_gcpack_safe, gcpack.data(), gcpack.size() * sizeof(float));
  } else {
    rc = Upload(&g->d_apack, apack_h.data(), apack_h.size() * sizeof(uint16_t));
    if (rc == AMLIKE_OK) rc = Upload((void **)&g->d_gcpack, gcpack.data(), gcpack.size() * sizeof(float));
  }
  if (rc == AMLIKE_OK && f16) rc = Upload((void **)&g->d_bscale, bscale.data(), bscale.size() * sizeof(float));
  if (rc == AMLIKE_OK) rc = Upload((void **)&g->d_endmask, endmask.data(), endmask.size() * sizeof(uint32_t));
  if (rc == AMLIKE_OK) rc = Upload((void **)&g->d_segcol, segcol.data(), segcol.size() * sizeof(int32_t));
  if (rc != AMLIKE_OK) return rc;
  if (!splits.empty())
    rc = Upload((void **)&g->d_splits, splits.data(), splits.size() * sizeof(SplitPdf));
  return rc;
}

// Launches the scoring kernels for [T x dim] device features with conforming layout.
// Scratch memory of one launch comes from the handle's growable buffers, or, when `arena` is
// given, from that fixed region (nothing is allocated; AMLIKE_EINVAL if it is too small).
struct ScratchArena {
  unsigned char *base;
  size_t bytes, used;
};

// The lane of `stream` (g->lane_mu held).  A handle keeps lanes for the kMaxStreamLanes streams
// used last; one more stream takes over the lane unused for longest, after the device has
// finished everything (the stream that lane served may no longer exist).
static int LaneFor(amlike_gmm *g, hipStream_t stream, StreamLane **out) {
  StreamLane *lane = nullptr;
  for (auto &l : g->lanes)
    if (l->stream == stream) lane = l.get();
  if (!lane) {
    if (g->lanes.size() < kMaxStreamLanes) {
      g->lanes.emplace_back(new StreamLane);
      lane = g->lanes.back().get();
    } else {
      for (auto &l : g->lanes)
        if (!lane || l->last_use < lane->last_use) lane = l.get();
      AMLIKE_HIP(hipDeviceSynchronize());
      lane->DropGraphs();
    }
    lane->stream = stream;
    if (!lane->d_ctl) {
      AMLIKE_HIP(hipMalloc(&lane->d_ctl, 2 * kCtlBlockBytes));
      // Zeroed on the lane's own stream, ahead of its first call's kernels.  (A hipMemset runs on the
      // null stream and returns before it has run; the caller's stream need not wait for that
      // stream, and the fill then landed, now and then, in the middle of the first call: a raised
      // flag or an overflow count wiped out.)
      if (int crc = ClearWords(lane->d_ctl, (int)(2 * kCtlBlockBytes / sizeof(int)), stream)) return crc;
    }
  }
  lane->last_use = ++g->lane_clock;
  *out = lane;
  return AMLIKE_OK;
}

// `lane`: the working memory of this stream; null with an arena.
static int LaunchScore(amlike_gmm *g, StreamLane *lane, const float *d_feats, int T, int ld, float *d_out,
                       int ldo, hipStream_t stream, ScratchArena *arena = nullptr) {
  const Variant &v = *g->variant;
  // bytes from the arena, or from the handle's buffer `ws`
  auto take = [&](DeviceBuffer *ws, size_t bytes, void **out) -> int {
    if (!arena) {
      void *before = ws->ptr;
      int rc = ws->Reserve(bytes);
      *out = ws->ptr;
      if (ws->ptr != before && before && lane) lane->DropGraphs();  // recorded launches point into the old block
      return rc;
    }
    const size_t at = (arena->used + 255) & ~(size_t)255;
    if (at + bytes > arena->bytes) {
      SetError("amlike_gmm_score_dev_noalloc: %d frames of this model need more than the %zu bytes of "
               "built-in scratch memory", T, arena->bytes);
      return AMLIKE_EINVAL;
    }
    *out = arena->base + at;
    arena->used = at + bytes;
    return AMLIKE_OK;
  };
  const int nbx = g->num_wc / v.W;
  // Frames per block: few enough blocks that the one-off load of the model
  // operands (NT*D registers per lane) is amortised over many frame tiles, enough
  // blocks to keep every CU busy to the end, and equal shares (no short last chunk).
  const int cus = std::max(1, NumComputeUnits());
  // (measured on 218.5 k frames: 2048 17.57, 3072 17.42, 4096 17.38, 6144 17.34, 8192 17.42 ms)
  const int target = g->fpb_exact ? g->fpb_exact : g->fpb_target;
  int nby = std::max(1, (T + target - 1) / target);
  // Small batches: split the frames further so that every CU has a block.
  while (nbx * nby < cus && (T + nby - 1) / nby > 64) ++nby;
  // One block per CU runs at a time, so the launch takes ceil(blocks / CUs) rounds of
  // (frames per block + start-up); pick the split near the target that wastes least of
  // the last round.  The start-up (loading the operand registers) costs about as much as
  // 96 frames of scoring.
  if (!g->fpb_exact) {
    double best_cost = 0.0;
    int best = nby;
    for (int cand = std::max(1, nby - nby / 4); cand <= std::max(nby + nby / 3 + 1, 6); ++cand) {
      const int f = ((T + cand - 1) / cand + 31) & ~31;
      if (f < 64) break;
      const int nb = nbx * ((T + f - 1) / f);
      const double cost = (double)((nb + cus - 1) / cus) * (f + 96);
      if (best_cost == 0.0 || cost < best_cost) {
        best_cost = cost;
        best = cand;
      }
    }
    nby = best;
  }
  const int fpb = ((T + nby - 1) / nby + 31) & ~31;
  nby = (T + fpb - 1) / fpb;
  ScoreArgs ar;
  ar.feats = d_feats;
  ar.bx = nullptr;
  ar.T = T;
  ar.ld = ld;
  ar.out = d_out;
  ar.ldo = ldo;
  ar.fpb = fpb;
  ar.nbx = nbx;
  ar.part_m = ar.part_s = nullptr;
  ar.part_ld = 0;
  // non-finite results are reported per stream lane; the arena path has no lane and reports per handle
  ar.flag = lane ? lane->d_ctl + kLaneFlagAt : g->d_flag;
  ar.ctl = lane ? lane->d_ctl : nullptr;
  if (arena) {  // the control words come from the arena too
    void *ctl = nullptr;
    int rc = take(nullptr, kCtlBlockBytes, &ctl);
    if (rc != AMLIKE_OK) return rc;
    ar.ctl = static_cast<int *>(ctl);
  }
  ar.fix_list = nullptr;
  ar.fix_cap = 0;
  ar.guard = 0;
  if (g->num_partials > 0) {
    const int part_ld = (T + 31) & ~31;
    void *part = nullptr;
    int rc = take(lane ? &lane->ws_part : nullptr, (size_t)2 * g->num_partials * part_ld * sizeof(float), &part);
    if (rc != AMLIKE_OK) return rc;
    ar.part_m = static_cast<float *>(part);
    ar.part_s = ar.part_m + (size_t)g->num_partials * part_ld;
    ar.part_ld = part_ld;
  }
  const int KS = v.D, num_ftiles = (T + 31) / 32;
  // One scoring pass: feature prepass (mode >= 1), the kernel, the merge of split pdfs.
  auto pass = [&](void (*kernel)(GmmPackView, ScoreArgs), ExpandFn expand, int parts,
                  const void *apack, const float *gcp, DeviceBuffer *ws, int guard) -> int {
    size_t smem = 0;
    ScoreArgs a = ar;
    a.guard = guard;
    if (expand) {
      // fp32 [T x dim] -> operand tiles (160 B per frame, part and k-step).
      const size_t tile_bytes = (size_t)KS * parts * 64 * 16;
      void *image = nullptr;
      int rc = take(ws, (size_t)num_ftiles * tile_bytes, &image);
      if (rc != AMLIKE_OK) return rc;
      a.bx = static_cast<const uint4 *>(image);
      const int threads = num_ftiles * KS * 64;
      hipLaunchKernelGGL(expand, dim3((threads + 255) / 256), dim3(256), 0, stream, d_feats, T, ld,
                         g->dim, static_cast<uint4 *>(image), num_ftiles, g->d_bscale, ar.ctl,
                         a.fix_cap, guard == 2 ? 2 : 0);
      AMLIKE_HIP(hipGetLastError());
      smem = 2 * FramesTilesPerBarrier(v.W, KS, parts) * tile_bytes + (size_t)v.W * v.NT * 32 * 4 +
             (size_t)v.W * DiffTailTableBytes(v.NT);
      if (g->solo) smem = std::max<size_t>(smem, 100 * 1024);
      if (smem > 64 * 1024)
        AMLIKE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    GmmPackView pk{apack, gcp, g->d_endmask, g->d_segcol, g->num_wc};
    // profiling: one event pair around the scoring kernel of the main pass, on its stream
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    if (g->profile && guard != 2 && g->kev_used < g->kev.size()) ev = &g->kev[g->kev_used++];
    if (ev) AMLIKE_HIP(hipEventRecord(ev->first, stream));
    hipLaunchKernelGGL(kernel, dim3(nbx * nby), dim3(64 * v.W), smem, stream, pk, a);
    AMLIKE_HIP(hipGetLastError());
    if (ev) AMLIKE_HIP(hipEventRecord(ev->second, stream));
    if (g->num_splits > 0 && v.mode < 100) {
      hipLaunchKernelGGL(gmm_merge_parts_kernel, dim3((T + 255) / 256, g->num_splits), dim3(256), 0,
                         stream, g->d_splits, g->num_splits, a.part_m, a.part_s, a.part_ld, T, d_out,
                         ldo, ar.flag, ar.ctl, a.fix_list, a.fix_cap, guard);
      AMLIKE_HIP(hipGetLastError());
    }
    return AMLIKE_OK;
  };
  if (v.mode == 3 || v.mode >= 300) {
    // Differenced pass, exact repair of the few pairs whose sum overflowed, and a full pass of
    // the wide-range kernel that runs only when the first pass could not be used (decided on
    // the device: features outside the fp16 range, or more overflows than the list holds).
    // (the same list length with and without an arena: which pairs are repaired exactly and when
    // the wide-range kernel takes over must not depend on where the scratch memory comes from)
    const size_t cap = std::max<size_t>(4096, std::min<size_t>((size_t)1 << 21, (size_t)T * g->num_pdfs / 256));
    void *fix = nullptr;
    int rc = take(lane ? &lane->ws_fix : nullptr, cap * sizeof(int2), &fix);
    if (rc != AMLIKE_OK) return rc;
    ar.fix_list = static_cast<int2 *>(fix);
    ar.fix_cap = (int)cap;
    // Cleared by a kernel, not by hipMemsetAsync: recorded into a HIP graph, the fill node left the
    // words as the previous replay's atomics had set them (every replay after the first then took
    // the wide-range path; seen with 8-byte and 256-byte fills alike), while a kernel's stores are
    // found by the next kernel's atomics and loads in eager launches and replays alike.
    hipLaunchKernelGGL(gmm_clear_ctl_kernel, dim3(1), dim3(64), 0, stream, ar.ctl);
    AMLIKE_HIP(hipGetLastError());
    rc = pass(v.kernel, v.expand, v.prec ? 2 : 3, g->d_apack, g->d_gcpack, lane ? &lane->ws_expand : nullptr, 1);
    if (rc != AMLIKE_OK) return rc;
    if (v.mode >= 300) return AMLIKE_OK;  // timing experiments: the main kernel alone
    hipLaunchKernelGGL(gmm_fix_kernel, dim3(1024), dim3(256), 0, stream, ar.ctl, ar.fix_list, ar.fix_cap,
                       g->d_gconsts, g->d_miv, g->d_iv, g->d_offsets, g->dim, d_feats, ld, d_out, ldo,
                       ar.flag);
    AMLIKE_HIP(hipGetLastError());
    rc = pass(v.safe_kernel, v.safe_expand, 3, g->d_apack_safe ? g->d_apack_safe : g->d_apack,
              g->d_gcpack_safe ? g->d_gcpack_safe : g->d_gcpack,
              !lane ? nullptr : v.prec ? &lane->ws_expand_safe : &lane->ws_expand, 2);
    if (rc != AMLIKE_OK) return rc;
  } else {
    int rc = pass(v.kernel, v.mode >= 1 ? v.expand : nullptr, 3, g->d_apack, g->d_gcpack,
                  lane ? &lane->ws_expand : nullptr, 0);
    if (rc != AMLIKE_OK) return rc;
  }
  if (arena) return AMLIKE_OK;  // (the description below is a heap-allocated string)
  char buf[256];
  snprintf(buf, sizeof(buf), "%s grid=%d block=%d frames_per_block=%d wave_chunks=%d fill=%.4f splits=%d",
           v.name, nbx * nby, 64 * v.W, fpb, g->num_wc, g->fill, g->num_splits);
  g->info = buf;
  return AMLIKE_OK;
}

// Makes the features conform (16-byte aligned rows); returns the pointer/stride to use.
static int ConformFeats(amlike_gmm *g, StreamLane *lane, const float *d_feats, int T, int ld, hipStream_t stream,
                        const float **use, int *use_ld) {
  const bool ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_feats) & 15) == 0) &&
                  ld >= ((g->dim + 3) & ~3);
  if (ok) {
    *use = d_feats;
    *use_ld = ld;
    return AMLIKE_OK;
  }
  const int ldp = (g->dim + 3) & ~3;
  void *pad_before = lane->ws_pad.ptr;
  int rc = lane->ws_pad.Reserve((size_t)T * ldp * sizeof(float));
  if (lane->ws_pad.ptr != pad_before && pad_before) lane->DropGraphs();
  if (rc != AMLIKE_OK) return rc;
  const size_t n = (size_t)T * ldp;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(gmm_pad_feats_kernel, dim3(blocks), dim3(256), 0, stream, d_feats, T, ld,
                     g->dim, static_cast<float *>(lane->ws_pad.ptr), ldp);
  AMLIKE_HIP(hipGetLastError());
  *use = static_cast<const float *>(lane->ws_pad.ptr);
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

// Per-Gaussian scores of pdfs [pdf_begin, pdf_end) for device features (arguments checked by the
// caller, g->lane_mu held).
static int GaussLoglikesDevLocked(amlike_gmm *g, amlike::StreamLane *lane, int pdf_begin, int pdf_end,
                                  const float *d_feats, int T, int ld, float *d_out, int ldo,
                                  hipStream_t st) {
  using namespace amlike;
  const int g0 = g->pdf_offsets[pdf_begin], g1 = g->pdf_offsets[pdf_end];
  const size_t valu_smem = (size_t)3 * 64 * (g->dim + 1) * sizeof(float);
  if ((T >= 64 && g1 - g0 >= 64 && !getenv("AMLIKE_GMM_GAUSS_VALU")) || valu_smem > 64 * 1024) {
    // A [T x 2D] . [2D x M] product with the gconsts as bias: the bf16x3 matrix-instruction
    // GEMM of the nnet2 path, on plane images of the features (per call) and of the model
    // (once per handle).
    const int Kp = (2 * g->dim + 15) & ~15, kp8 = Kp >> 3;
    const size_t G = (size_t)g->num_gauss;
    if (!g->d_wplanes) {
      AMLIKE_HIP(hipMalloc(&g->d_wplanes, 3 * G * kp8 * sizeof(uint4)));
      const int blocks = (int)std::min<size_t>((G * kp8 + 255) / 256, 16384);
      hipLaunchKernelGGL(gmm_gemm_planes_kernel<true>, dim3(blocks), dim3(256), 0, st, g->d_miv, g->d_iv,
                         g->dim, g->dim, g->num_gauss, static_cast<uint4 *>(g->d_wplanes), G * kp8, kp8);
      AMLIKE_HIP(hipGetLastError());
      AMLIKE_HIP(hipStreamSynchronize(st));  // calls on other streams read it from now on
    }
    int rc = lane->ws_xplanes.Reserve((size_t)3 * T * kp8 * sizeof(uint4));
    if (rc != AMLIKE_OK) return rc;
    const int blocks = (int)std::min<size_t>(((size_t)T * kp8 + 255) / 256, 16384);
    hipLaunchKernelGGL(gmm_gemm_planes_kernel<false>, dim3(blocks), dim3(256), 0, st, d_feats, d_feats, ld,
                       g->dim, T, static_cast<uint4 *>(lane->ws_xplanes.ptr), (size_t)T * kp8, kp8);
    AMLIKE_HIP(hipGetLastError());
    return LaunchPlaneGemmBf16x3(lane->ws_xplanes.ptr, (size_t)T * kp8,
                                 static_cast<const uint4 *>(g->d_wplanes) + g0, G * kp8, G, Kp,
                                 g->d_gconsts + g0, d_out, ldo, T, g1 - g0, st);
  }
  const size_t smem = (size_t)3 * 64 * (g->dim + 1) * sizeof(float);
  if (smem > 64 * 1024) {
    SetError("amlike_gmm_gauss_loglikes: dim %d too large", g->dim);
    return AMLIKE_EDIM;
  }
  dim3 grid((g1 - g0 + 63) / 64, (T + 63) / 64);
  hipLaunchKernelGGL(gmm_gauss_loglikes_kernel, grid, dim3(256), smem,
                     st, g->d_gconsts, g->d_miv, g->d_iv, g0, g1,
                     g->dim, d_feats, T, ld, d_out, ldo);
  AMLIKE_HIP(hipGetLastError());
  return AMLIKE_OK;
}

extern "C" {

int amlike_gmm_create(const float *gconsts, const float *means_invvars, const float *inv_vars,
                      const int32_t *pdf_offsets, int32_t num_pdfs, int32_t num_gauss,
                      int32_t dim, amlike_gmm_t **out) {
  if (int brc = BindThread()) return brc;
  if (!gconsts || !means_invvars || !inv_vars || !pdf_offsets || !out || num_pdfs <= 0 ||
      num_gauss <= 0 || dim <= 0 || pdf_offsets[0] != 0 || pdf_offsets[num_pdfs] != num_gauss) {
    SetError("amlike_gmm_create: invalid argument");
    return AMLIKE_EINVAL;
  }
  if (CurrentDevice() < 0) {
    int rc = amlike_init(0);
    if (rc != AMLIKE_OK) return rc;
  }
  // PackModel may switch to the bf16x3 twin.  Dimensions the fused kernels are not built for
  // (above 64) have no variant: such a model is scored through the per-Gaussian score GEMM and
  // a log-sum-exp pass over its result, which take any dimension.
  const Variant *v = ChooseVariant(dim, true);
  if (!v && (getenv("AMLIKE_GMM_MODE") || getenv("AMLIKE_GMM_VARIANT"))) {
    SetError("amlike_gmm_create: no kernel for feature dimension %d with the requested mode", dim);
    return AMLIKE_EDIM;
  }
  amlike_gmm *g = new amlike_gmm;
  g->num_pdfs = num_pdfs;
  g->num_gauss = num_gauss;
  g->dim = dim;
  g->device = CurrentDevice();
  if (const char *e = getenv("AMLIKE_GMM_GRAPHS")) g->use_graphs = e[0] == '1';
  if (const char *e = getenv("AMLIKE_GMM_FPB_TARGET")) g->fpb_target = std::max(64, atoi(e));
  if (const char *e = getenv("AMLIKE_GMM_FPB")) g->fpb_exact = std::max(32, atoi(e));
  if (const char *e = getenv("AMLIKE_GMM_SOLO")) g->solo = e[0] != 0;
  g->variant = v;
  g->pdf_offsets.assign(pdf_offsets, pdf_offsets + num_pdfs + 1);
  int rc = v ? PackModel(g, gconsts, means_invvars, inv_vars, pdf_offsets) : AMLIKE_OK;
  if (!v) g->info = "per-Gaussian score GEMM + log-sum-exp pass (no fused kernel for this dimension)";
  if (rc == AMLIKE_OK) {
    auto up = [&](float **dst, const float *src, size_t n) -> int {
      AMLIKE_HIP(hipMalloc(dst, n * sizeof(float)));
      AMLIKE_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
      return AMLIKE_OK;
    };
    rc = up(&g->d_gconsts, gconsts, (size_t)num_gauss);
    if (rc == AMLIKE_OK) rc = up(&g->d_miv, means_invvars, (size_t)num_gauss * dim);
    if (rc == AMLIKE_OK) rc = up(&g->d_iv, inv_vars, (size_t)num_gauss * dim);
    if (rc == AMLIKE_OK) {
      hipError_t e = hipMalloc(&g->d_offsets, (size_t)(num_pdfs + 1) * sizeof(int32_t));
      if (e == hipSuccess)
        e = hipMemcpy(g->d_offsets, pdf_offsets, (size_t)(num_pdfs + 1) * sizeof(int32_t),
                      hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        SetError("amlike_gmm_create: %s", hipGetErrorString(e));
        rc = AMLIKE_EHIP;
      }
    }
  }
  if (rc == AMLIKE_OK) {
    hipError_t e = hipMalloc(&g->d_flag, sizeof(int));
    const int zero = 0;  // (a blocking copy: finished when it returns, whatever stream comes next)
    if (e == hipSuccess) e = hipMemcpy(g->d_flag, &zero, sizeof(int), hipMemcpyHostToDevice);
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
  if (int brc = BindHandle(g)) return brc;
  if (!g) return AMLIKE_OK;
  (void)hipFree(g->d_apack);
  (void)hipFree(g->d_apack_safe);
  (void)hipFree(g->d_bscale);
  (void)hipFree(g->d_gcpack_safe);
  for (auto &l : g->lanes) l->Free();
  (void)hipFree(g->d_gcpack);
  (void)hipFree(g->d_endmask);
  (void)hipFree(g->d_segcol);
  (void)hipFree(g->d_splits);
  (void)hipFree(g->d_gconsts);
  (void)hipFree(g->d_miv);
  (void)hipFree(g->d_iv);
  (void)hipFree(g->d_flag);
  (void)hipFree(g->d_offsets);
  g->ws_feats.Free();
  g->ws_gsel.Free();
  g->ws_raw.Free();
  (void)hipFree(g->d_wplanes);
  g->ws_out.Free();
  for (AsyncSlot &s : g->slots) s.Release();
  for (auto &e : g->kev) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return AMLIKE_OK;
}

int32_t amlike_gmm_num_pdfs(const amlike_gmm_t *g) { return g ? g->num_pdfs : 0; }
int32_t amlike_gmm_num_gauss(const amlike_gmm_t *g) { return g ? g->num_gauss : 0; }
int32_t amlike_gmm_dim(const amlike_gmm_t *g) { return g ? g->dim : 0; }
int32_t amlike_gmm_device(const amlike_gmm_t *g) { return g ? g->device : -1; }
const char *amlike_gmm_kernel_info(const amlike_gmm_t *g) { return g ? g->info.c_str() : ""; }

int amlike_gmm_score_dev(amlike_gmm_t *g, const float *d_feats, int32_t T, int32_t ld,
                         int32_t feat_dim, float log_sum_exp_prune, float *d_out, int32_t ldo,
                         void *stream) {
  if (int brc = BindHandle(g)) return brc;
  int rc = CheckScoreArgs(g, T, ld, feat_dim, ldo, d_feats, d_out);
  if (rc != AMLIKE_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lane_lock(g->lane_mu);
  StreamLane *lane = nullptr;
  rc = LaneFor(g, st, &lane);
  if (rc != AMLIKE_OK) return rc;
  if (log_sum_exp_prune > 0.0f || !g->variant) {
    // Pruning needs each pdf's maximum before its sum, so this mode goes through the
    // per-Gaussian scores, in frame chunks that bound the scratch matrix.
    const int G = g->num_gauss;
    // (1 GiB: smaller chunks that would fit the memory-side cache measured slower.)
    size_t budget = (size_t)1 << 28;  // floats
    if (const char *e = getenv("AMLIKE_GMM_PRUNE_SCRATCH_MB")) budget = (size_t)std::max(1, atoi(e)) << 18;
    const int chunk = std::max(64, (int)std::min<size_t>((size_t)T, budget / (size_t)G));
    rc = lane->ws_gauss.Reserve((size_t)std::min(chunk, T) * G * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    float *scores = static_cast<float *>(lane->ws_gauss.ptr);
    // the log-sum-exp kernel takes a block per frame in grid.y, which holds 65535
    const int launch_rows = 32768;
    for (int t0 = 0; t0 < T; t0 += chunk) {
      const int tn = std::min(chunk, T - t0);
      rc = GaussLoglikesDevLocked(g, lane, 0, g->num_pdfs, d_feats + (size_t)t0 * ld, tn, ld, scores, G, st);
      if (rc != AMLIKE_OK) return rc;
      for (int r0 = 0; r0 < tn; r0 += launch_rows) {
        const int rn = std::min(launch_rows, tn - r0);
        hipLaunchKernelGGL(gmm_pruned_lse_kernel, dim3((g->num_pdfs + 255) / 256, rn), dim3(256), 0, st,
                           scores + (size_t)r0 * G, G, g->d_offsets, g->num_pdfs, rn, log_sum_exp_prune,
                           d_out + (size_t)(t0 + r0) * ldo, ldo, lane->d_ctl + kLaneFlagAt);
        AMLIKE_HIP(hipGetLastError());
      }
    }
    if (g->variant) g->info = "pruned log-sum-exp over per-Gaussian scores";
    return AMLIKE_OK;
  }
  auto launch = [&]() -> int {
    const float *use = d_feats;
    int use_ld = ld;
    if (g->variant->mode == 0) {  // the fp32 kernel reads rows as float4
      int lrc = ConformFeats(g, lane, d_feats, T, ld, st, &use, &use_ld);
      if (lrc != AMLIKE_OK) return lrc;
    }
    return LaunchScore(g, lane, use, T, use_ld, d_out, ldo, st);
  };
  // Small calls on a stream of the caller's, for handles created with AMLIKE_GMM_GRAPHS=1: the
  // second call with the same buffers and shape records the sequence as a graph, later ones
  // replay it.  (The first call runs as usual and sizes the working memory, so that nothing is
  // allocated while recording.)  Off by default: measured on swbd tri4 with T = 64 / 500 / 2000,
  // a replay takes 45 / 95 / 232 us back to back against 38 / 88 / 224 us for the plain launches,
  // and 53 / 104 / 242 us against 59 / 102 / 240 us when each call is waited for.
  if (!g->use_graphs || T > kScoreGraphMaxFrames || st == nullptr || g->profile) return launch();
  ScoreGraph *sg = nullptr, *oldest = &lane->graphs[0];
  for (ScoreGraph &c : lane->graphs) {
    if (c.seen && c.d_feats == d_feats && c.d_out == d_out && c.T == T && c.ld == ld && c.ldo == ldo) sg = &c;
    if (c.last_use < oldest->last_use) oldest = &c;
  }
  if (!sg) {
    oldest->Free();
    *oldest = ScoreGraph();
    sg = oldest;
    sg->d_feats = d_feats;
    sg->d_out = d_out;
    sg->T = T;
    sg->ld = ld;
    sg->ldo = ldo;
  }
  sg->last_use = ++g->lane_clock;
  sg->seen++;
  if (sg->exec) {
    AMLIKE_HIP(hipGraphLaunch(sg->exec, st));
    return AMLIKE_OK;
  }
  if (sg->seen < 2 || sg->failed) return launch();
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    sg->failed = true;
    return launch();
  }
  rc = launch();
  e = hipStreamEndCapture(st, &sg->graph);
  if (rc == AMLIKE_OK && e == hipSuccess && sg->graph) e = hipGraphInstantiate(&sg->exec, sg->graph, nullptr, nullptr, 0);
  if (rc != AMLIKE_OK || e != hipSuccess || !sg->exec) {
    (void)hipGetLastError();
    sg->Free();
    sg->failed = true;
    return rc != AMLIKE_OK ? rc : launch();  // nothing has run yet: the capture only recorded
  }
  AMLIKE_HIP(hipGraphLaunch(sg->exec, st));
  return AMLIKE_OK;
}

int amlike_gmm_score_dev_noalloc(amlike_gmm_t *g, const float *d_feats, int32_t T, int32_t ld,
                                 int32_t feat_dim, float *d_out, int32_t ldo, void *stream) {
  if (int brc = BindHandle(g)) return brc;
  int rc = CheckScoreArgs(g, T, ld, feat_dim, ldo, d_feats, d_out);
  if (rc != AMLIKE_OK) return rc;
  if (!g->variant || g->variant->mode == 0) {
    SetError("amlike_gmm_score_dev_noalloc: this model is not scored by a fused 16-bit kernel (dimension above "
             "64, or the exact-fp32 mode); use amlike_gmm_score_dev");
    return AMLIKE_EINVAL;
  }
  unsigned char *base = g->device >= 0 && g->device < kMaxDevices ? g_scratch_ptr[g->device] : nullptr;
  if (!base) {
    SetError("amlike_gmm_score_dev_noalloc: amlike_init has not set up device %d", g->device);
    return AMLIKE_EINVAL;
  }
  ScratchArena arena{base, kNoAllocScratchBytes, 0};
  std::lock_guard<std::mutex> lane_lock(g->lane_mu);
  return LaunchScore(g, nullptr, d_feats, T, ld, d_out, ldo, static_cast<hipStream_t>(stream), &arena);
}

size_t amlike_gmm_noalloc_scratch_bytes(void) { return kNoAllocScratchBytes; }

int amlike_gmm_poll(amlike_gmm_t *g, void *stream) {
  if (int brc = BindHandle(g)) return brc;
  if (!g) return AMLIKE_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // The flag of this stream's lane (launches of amlike_gmm_score_dev / _gauss_loglikes_dev on it)
  // and the handle's own (amlike_gmm_score_dev_noalloc, which has no lane).
  int *lane_flag = nullptr;
  {
    std::lock_guard<std::mutex> lane_lock(g->lane_mu);
    for (auto &l : g->lanes)
      if (l->stream == st && l->d_ctl) lane_flag = l->d_ctl + kLaneFlagAt;
  }
  int flags[2] = {0, 0};
  AMLIKE_HIP(hipMemcpyAsync(&flags[0], g->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
  if (int crc = ClearWords(g->d_flag, 1, st)) return crc;
  if (lane_flag) {
    AMLIKE_HIP(hipMemcpyAsync(&flags[1], lane_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    if (int crc = ClearWords(lane_flag, 1, st)) return crc;
  }
  AMLIKE_HIP(hipStreamSynchronize(st));
  const int flag = flags[0] | flags[1];
  if (flag) {
    SetError("Invalid answer (overflow or invalid variances/features?)");
    return AMLIKE_ENONFINITE;
  }
  return AMLIKE_OK;
}

// Scores device-resident features in passes of bounded output size and copies each pass
// out; g->mu is held by the caller.
static int ScoreResidentToHost(amlike_gmm *g, const float *df, int T, int dld, float prune,
                               float *out, int ldo) {
  const int P = g->num_pdfs;
  // Bound the device footprint of one pass: at most ~1 GiB of output.
  const int max_t = std::max(1024, (int)std::min<size_t>(1u << 20, ((size_t)1 << 28) / (size_t)P));
  int status = AMLIKE_OK;
  for (int t0 = 0; t0 < T; t0 += max_t) {
    const int tn = std::min(max_t, T - t0);
    int rc = g->ws_out.Reserve((size_t)tn * P * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    float *dout = static_cast<float *>(g->ws_out.ptr);
    rc = amlike_gmm_score_dev(g, df + (size_t)t0 * dld, tn, dld, g->dim, prune, dout, P, g->stream);
    if (rc != AMLIKE_OK) return rc;
    rc = CopyMatrixToHost(out + (size_t)t0 * ldo, (size_t)ldo, dout, tn, P, g->stream);
    if (rc != AMLIKE_OK) return rc;
    rc = amlike_gmm_poll(g, g->stream);
    if (rc == AMLIKE_ENONFINITE) status = rc;
    else if (rc != AMLIKE_OK) return rc;
  }
  return status;
}

int amlike_gmm_score(amlike_gmm_t *g, const float *feats, int32_t T, int32_t ld,
                     int32_t feat_dim, float log_sum_exp_prune, float *out, int32_t ldo) {
  if (int brc = BindHandle(g)) return brc;
  int rc = CheckScoreArgs(g, T, ld, feat_dim, ldo, feats, out);
  if (rc != AMLIKE_OK) return rc;
  std::lock_guard<std::mutex> lock(g->mu);
  const int ldp = (g->dim + 3) & ~3;
  // Rows travel as one contiguous copy whenever the host layout allows it (a strided
  // 2-D copy from pageable memory degenerates into one transfer per row); the kernels
  // take any row stride, so the features keep the caller's ld on the device.
  const bool feats_1d = (size_t)ld <= (size_t)2 * ldp + 8;
  const int dld = feats_1d ? ld : ldp;
  rc = g->ws_feats.Reserve((size_t)T * dld * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  float *df = static_cast<float *>(g->ws_feats.ptr);
  if (feats_1d) {
    // the last row is only dim wide in the caller's buffer
    AMLIKE_HIP(hipMemcpyAsync(df, feats, ((size_t)(T - 1) * ld + g->dim) * sizeof(float),
                              hipMemcpyHostToDevice, g->stream));
  } else {
    AMLIKE_HIP(hipMemcpy2DAsync(df, (size_t)dld * sizeof(float), feats, (size_t)ld * sizeof(float),
                                (size_t)g->dim * sizeof(float), T, hipMemcpyHostToDevice, g->stream));
  }
  return ScoreResidentToHost(g, df, T, dld, log_sum_exp_prune, out, ldo);
}

int amlike_gmm_score_raw(amlike_gmm_t *g, amlike_xform_t *xf, const float *raw, int32_t T, int32_t ld,
                         const float *spk_mat, int32_t rows2, int32_t cols2,
                         float log_sum_exp_prune, float *out, int32_t ldo) {
  if (int brc = BindHandle(g)) return brc;
  if (!g || !xf || !raw || !out || T <= 0 || ld < amlike_xform_input_dim(xf) || ldo < g->num_pdfs) {
    SetError(T <= 0 && g && xf && raw && out ? "SpliceFrames: empty input" : "amlike_gmm_score_raw: invalid argument");
    return AMLIKE_EINVAL;
  }
  if (amlike_xform_device(xf) != g->device) {
    SetError("amlike_gmm_score_raw: the transform lives on device %d, the model on device %d",
             amlike_xform_device(xf), g->device);
    return AMLIKE_EINVAL;
  }
  const int final_dim = spk_mat ? rows2 : amlike_xform_output_dim(xf);
  if (final_dim != g->dim) {
    SetError("Dim mismatch: data dim = %d vs. model dim = %d", final_dim, g->dim);
    return AMLIKE_EDIM;
  }
  std::lock_guard<std::mutex> lock(g->mu);
  const int ldp = (g->dim + 3) & ~3;
  const int in_dim = amlike_xform_input_dim(xf);
  int rc = g->ws_raw.Reserve((size_t)T * ld * sizeof(float) +
                             (size_t)std::max(1, rows2 * cols2) * sizeof(float) + 256);
  if (rc == AMLIKE_OK) rc = g->ws_feats.Reserve((size_t)T * ldp * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  float *draw = static_cast<float *>(g->ws_raw.ptr);
  float *dspk = draw + (((size_t)T * ld + 63) & ~(size_t)63);
  float *df = static_cast<float *>(g->ws_feats.ptr);
  AMLIKE_HIP(hipMemcpyAsync(draw, raw, ((size_t)(T - 1) * ld + in_dim) * sizeof(float),
                            hipMemcpyHostToDevice, g->stream));
  if (spk_mat)
    AMLIKE_HIP(hipMemcpyAsync(dspk, spk_mat, (size_t)rows2 * cols2 * sizeof(float),
                              hipMemcpyHostToDevice, g->stream));
  if (ldp != g->dim) AMLIKE_HIP(hipMemsetAsync(df, 0, (size_t)T * ldp * sizeof(float), g->stream));
  rc = amlike_xform_apply_dev(xf, draw, ld, &T, 1, spk_mat ? dspk : nullptr, rows2, cols2, nullptr, df,
                              ldp, g->stream);
  if (rc != AMLIKE_OK) return rc;
  return ScoreResidentToHost(g, df, T, ldp, log_sum_exp_prune, out, ldo);
}

int amlike_gmm_score_submit(amlike_gmm_t *g, const float *feats, int32_t T, int32_t ld,
                            int32_t feat_dim, float log_sum_exp_prune, float *out, int32_t ldo,
                            int32_t *ticket) {
  if (int brc = BindHandle(g)) return brc;
  int rc = CheckScoreArgs(g, T, ld, feat_dim, ldo, feats, out);
  if (rc != AMLIKE_OK) return rc;
  const int P = g->num_pdfs;
  if (!ticket || (size_t)T * P > ((size_t)1 << 30)) {
    SetError("amlike_gmm_score_submit: %s", ticket ? "more than 2^30 outputs in one submission" : "null ticket");
    return AMLIKE_EINVAL;
  }
  std::lock_guard<std::mutex> lock(g->mu);
  AsyncSlot *s = nullptr;
  for (AsyncSlot &c : g->slots)
    if (c.ticket < 0 && !s) s = &c;
  if (!s) {
    SetError("amlike_gmm_score_submit: %d submissions in flight, wait for one first", kAsyncSlots);
    return AMLIKE_EBUSY;
  }
  if (!g->copy_stream) AMLIKE_HIP(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
  rc = s->Prepare();
  if (rc != AMLIKE_OK) return rc;
  const int ldp = (g->dim + 3) & ~3;
  const bool feats_1d = (size_t)ld <= (size_t)2 * ldp + 8;
  const int dld = feats_1d ? ld : ldp;
  rc = s->in.Reserve((size_t)T * dld * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  rc = s->out.Reserve((size_t)T * P * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  float *df = static_cast<float *>(s->in.ptr), *dout = static_cast<float *>(s->out.ptr);
  if (feats_1d)
    AMLIKE_HIP(hipMemcpyAsync(df, feats, ((size_t)(T - 1) * ld + g->dim) * sizeof(float),
                              hipMemcpyHostToDevice, g->stream));
  else
    AMLIKE_HIP(hipMemcpy2DAsync(df, (size_t)dld * sizeof(float), feats, (size_t)ld * sizeof(float),
                                (size_t)g->dim * sizeof(float), T, hipMemcpyHostToDevice, g->stream));
  rc = amlike_gmm_score_dev(g, df, T, dld, feat_dim, log_sum_exp_prune, dout, P, g->stream);
  if (rc != AMLIKE_OK) return rc;
  {  // the flag of the lane amlike_gmm_score_dev has just used for the handle's stream
    int *lane_flag = nullptr;
    std::lock_guard<std::mutex> lane_lock(g->lane_mu);
    for (auto &l : g->lanes)
      if (l->stream == g->stream && l->d_ctl) lane_flag = l->d_ctl + kLaneFlagAt;
    if (!lane_flag) {
      SetError("amlike_gmm_score_submit: no working memory for the handle's stream");
      return AMLIKE_EINVAL;
    }
    AMLIKE_HIP(hipMemcpyAsync(s->h_flag, lane_flag, sizeof(int), hipMemcpyDeviceToHost, g->stream));
    if (int crc = ClearWords(lane_flag, 1, g->stream)) return crc;
  }
  AMLIKE_HIP(hipEventRecord(s->computed, g->stream));
  AMLIKE_HIP(hipStreamWaitEvent(g->copy_stream, s->computed, 0));
  if (ldo == P)
    AMLIKE_HIP(hipMemcpyAsync(out, dout, (size_t)T * P * sizeof(float), hipMemcpyDeviceToHost,
                              g->copy_stream));
  else
    AMLIKE_HIP(hipMemcpy2DAsync(out, (size_t)ldo * sizeof(float), dout, (size_t)P * sizeof(float),
                                (size_t)P * sizeof(float), T, hipMemcpyDeviceToHost, g->copy_stream));
  AMLIKE_HIP(hipEventRecord(s->done, g->copy_stream));
  s->ticket = g->next_ticket;
  g->next_ticket = (g->next_ticket + 1) & 0x3fffffff;
  *ticket = s->ticket;
  return AMLIKE_OK;
}

int amlike_gmm_score_wait(amlike_gmm_t *g, int32_t ticket) {
  if (int brc = BindHandle(g)) return brc;
  if (!g || ticket < 0) {
    SetError("amlike_gmm_score_wait: invalid argument");
    return AMLIKE_EINVAL;
  }
  AsyncSlot *s = nullptr;
  {
    std::lock_guard<std::mutex> lock(g->mu);
    for (AsyncSlot &c : g->slots)
      if (c.ticket == ticket) s = &c;
  }
  if (!s) {
    SetError("amlike_gmm_score_wait: ticket %d is not in flight", ticket);
    return AMLIKE_EINVAL;
  }
  AMLIKE_HIP(hipEventSynchronize(s->done));
  const int flag = *s->h_flag;
  {
    std::lock_guard<std::mutex> lock(g->mu);
    s->ticket = -1;
  }
  if (flag) {
    SetError("Invalid answer (overflow or invalid variances/features?)");
    return AMLIKE_ENONFINITE;
  }
  return AMLIKE_OK;
}

int amlike_gmm_gauss_loglikes_dev(amlike_gmm_t *g, int32_t pdf_begin, int32_t pdf_end,
                                  const float *d_feats, int32_t T, int32_t ld, int32_t feat_dim,
                                  float *d_out, int32_t ldo, void *stream) {
  if (int brc = BindHandle(g)) return brc;
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
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lane_lock(g->lane_mu);
  StreamLane *lane = nullptr;
  int rc = LaneFor(g, st, &lane);
  if (rc != AMLIKE_OK) return rc;
  return GaussLoglikesDevLocked(g, lane, pdf_begin, pdf_end, d_feats, T, ld, d_out, ldo, st);
}

int amlike_gmm_gauss_loglikes(amlike_gmm_t *g, int32_t pdf_begin, int32_t pdf_end,
                              const float *feats, int32_t T, int32_t ld, int32_t feat_dim,
                              float *out, int32_t ldo) {
  if (int brc = BindHandle(g)) return brc;
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
    // features: one contiguous copy that keeps the caller's row stride (see amlike_gmm_score)
    const bool feats_1d = (size_t)ld <= (size_t)2 * feat_dim + 8;
    const int dld = feats_1d ? ld : feat_dim;
    int rc = g->ws_feats.Reserve((size_t)tn * dld * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    rc = g->ws_out.Reserve((size_t)tn * M * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    float *df = static_cast<float *>(g->ws_feats.ptr);
    float *dout = static_cast<float *>(g->ws_out.ptr);
    if (feats_1d)
      AMLIKE_HIP(hipMemcpyAsync(df, feats + (size_t)t0 * ld, ((size_t)(tn - 1) * ld + feat_dim) * sizeof(float),
                                hipMemcpyHostToDevice, g->stream));
    else
      AMLIKE_HIP(hipMemcpy2DAsync(df, (size_t)feat_dim * sizeof(float), feats + (size_t)t0 * ld,
                                  (size_t)ld * sizeof(float), (size_t)feat_dim * sizeof(float), tn,
                                  hipMemcpyHostToDevice, g->stream));
    rc = amlike_gmm_gauss_loglikes_dev(g, pdf_begin, pdf_end, df, tn, dld, feat_dim, dout, M,
                                       g->stream);
    if (rc != AMLIKE_OK) return rc;
    AMLIKE_HIP(hipMemcpy2DAsync(out + (size_t)t0 * ldo, (size_t)ldo * sizeof(float), dout,
                                (size_t)M * sizeof(float), (size_t)M * sizeof(float), tn,
                                hipMemcpyDeviceToHost, g->stream));
    AMLIKE_HIP(hipStreamSynchronize(g->stream));
  }
  return AMLIKE_OK;
}

int amlike_gmm_gselect(amlike_gmm_t *g, int32_t pdf, const float *feats, int32_t T, int32_t ld,
                       int32_t feat_dim, int32_t num_gselect, int32_t *out_idx, double *tot_loglike) {
  if (int brc = BindHandle(g)) return brc;
  if (!g || !feats || !out_idx || pdf < 0 || pdf >= g->num_pdfs || num_gselect <= 0 || T <= 0 ||
      ld < feat_dim) {
    SetError("amlike_gmm_gselect: invalid argument");
    return AMLIKE_EINVAL;
  }
  if (feat_dim != g->dim) {
    SetError("DiagGmm::GaussianSelection, dimension mismatch %d vs. %d", feat_dim, g->dim);
    return AMLIKE_EDIM;
  }
  const int M = g->pdf_offsets[pdf + 1] - g->pdf_offsets[pdf];
  if (M > 64 * kGselKeep) {
    SetError("amlike_gmm_gselect: %d Gaussians in one pdf (limit %d)", M, 64 * kGselKeep);
    return AMLIKE_EINVAL;
  }
  std::lock_guard<std::mutex> lock(g->mu);
  // frame chunks that keep the [chunk x M] score matrix near 256 MiB
  const int max_t = std::max(64, (int)std::min<size_t>(1u << 20, ((size_t)1 << 26) / (size_t)M));
  const bool feats_1d = (size_t)ld <= (size_t)2 * feat_dim + 8;
  const int dld = feats_1d ? ld : feat_dim;
  std::vector<float> ll;
  double total = 0.0;
  for (int t0 = 0; t0 < T; t0 += max_t) {
    const int tn = std::min(max_t, T - t0);
    int rc = g->ws_feats.Reserve((size_t)tn * dld * sizeof(float));
    if (rc == AMLIKE_OK) rc = g->ws_gsel.Reserve((size_t)tn * M * sizeof(float));
    if (rc == AMLIKE_OK)
      rc = g->ws_out.Reserve((size_t)tn * num_gselect * sizeof(int32_t) + (size_t)tn * sizeof(float));
    if (rc != AMLIKE_OK) return rc;
    float *df = static_cast<float *>(g->ws_feats.ptr);
    float *dscore = static_cast<float *>(g->ws_gsel.ptr);
    int32_t *didx = static_cast<int32_t *>(g->ws_out.ptr);
    float *dll = reinterpret_cast<float *>(didx + (size_t)tn * num_gselect);
    if (feats_1d)
      AMLIKE_HIP(hipMemcpyAsync(df, feats + (size_t)t0 * ld, ((size_t)(tn - 1) * ld + feat_dim) * sizeof(float),
                                hipMemcpyHostToDevice, g->stream));
    else
      AMLIKE_HIP(hipMemcpy2DAsync(df, (size_t)feat_dim * sizeof(float), feats + (size_t)t0 * ld,
                                  (size_t)ld * sizeof(float), (size_t)feat_dim * sizeof(float), tn,
                                  hipMemcpyHostToDevice, g->stream));
    rc = amlike_gmm_gauss_loglikes_dev(g, pdf, pdf + 1, df, tn, dld, feat_dim, dscore, M, g->stream);
    if (rc != AMLIKE_OK) return rc;
    hipLaunchKernelGGL(gmm_gselect_kernel, dim3((tn + 3) / 4), dim3(256), 0, g->stream, dscore, M, M, tn,
                       num_gselect, didx, dll);
    AMLIKE_HIP(hipGetLastError());
    ll.resize(tn);
    AMLIKE_HIP(hipMemcpyAsync(out_idx + (size_t)t0 * num_gselect, didx,
                              (size_t)tn * num_gselect * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
    AMLIKE_HIP(hipMemcpyAsync(ll.data(), dll, (size_t)tn * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    AMLIKE_HIP(hipStreamSynchronize(g->stream));
    for (int t = 0; t < tn; ++t) total += ll[t];
  }
  if (tot_loglike) *tot_loglike = total;
  return AMLIKE_OK;
}

int amlike_gmm_profile_kernel(amlike_gmm_t *g, int32_t enable) {
  if (int brc = BindHandle(g)) return brc;
  if (!g) return AMLIKE_EINVAL;
  std::lock_guard<std::mutex> lock(g->lane_mu);
  if (enable && g->kev.empty()) {
    g->kev.resize(256);
    for (auto &e : g->kev) {
      e.first = e.second = nullptr;
      AMLIKE_HIP(hipEventCreate(&e.first));
      AMLIKE_HIP(hipEventCreate(&e.second));
    }
  }
  g->profile = enable != 0;
  g->kev_used = 0;
  return AMLIKE_OK;
}

int amlike_gmm_kernel_time(amlike_gmm_t *g, float *mean_ms, int32_t *launches) {
  if (int brc = BindHandle(g)) return brc;
  if (!g || !mean_ms) return AMLIKE_EINVAL;
  std::lock_guard<std::mutex> lock(g->lane_mu);
  double sum = 0.0;
  for (size_t i = 0; i < g->kev_used; ++i) {
    AMLIKE_HIP(hipEventSynchronize(g->kev[i].second));
    float ms = 0.0f;
    AMLIKE_HIP(hipEventElapsedTime(&ms, g->kev[i].first, g->kev[i].second));
    sum += ms;
  }
  *mean_ms = g->kev_used ? (float)(sum / g->kev_used) : 0.0f;
  if (launches) *launches = (int32_t)g->kev_used;
  g->kev_used = 0;
  return AMLIKE_OK;
}

int amlike_gmm_time_score_dev(amlike_gmm_t *g, const float *d_feats, int32_t T, int32_t ld,
                              int32_t feat_dim, float *d_out, int32_t ldo, void *stream,
                              int32_t iters, float *ms_per_launch) {
  if (int brc = BindHandle(g)) return brc;
  if (!ms_per_launch || iters <= 0) return AMLIKE_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  AMLIKE_HIP(hipEventCreate(&e0));
  AMLIKE_HIP(hipEventCreate(&e1));
  AMLIKE_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) {
    int rc = amlike_gmm_score_dev(g, d_feats, T, ld, feat_dim, -1.0f, d_out, ldo, stream);
    if (rc != AMLIKE_OK) {
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      return rc;
    }
  }
  AMLIKE_HIP(hipEventRecord(e1, st));
  AMLIKE_HIP(hipEventSynchronize(e1));
  float ms = 0.0f;
  AMLIKE_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = ms / iters;
#ifdef AMLIKE_DIFF_STAMPS
  {
    {
      unsigned long long hb[256 * 4];
      if (hipMemcpyFromSymbol(hb, HIP_SYMBOL(g_diff_blocks), sizeof(hb)) == hipSuccess) {
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < 256; ++b)
          if (hb[b * 4 + 1]) t0 = std::min(t0, hb[b * 4]);
        for (int b = 0; b < 256; ++b)
          if (hb[b * 4 + 1])
            fprintf(stderr, "chunk %3d: start %8.2f us  duration %7.2f us  loop ticks %llu  xcc %llu\n", b,
                    (hb[b * 4] - t0) * 0.01, hb[b * 4 + 1] * 0.01, hb[b * 4 + 2], hb[b * 4 + 3]);
      }
    }
    unsigned long long hs[8 * 6];
    if (hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_diff_stamps), sizeof(hs)) == hipSuccess)
      for (int wv = 0; wv < 8; ++wv) {
        const double n = (double)std::max(1ull, hs[wv * 5 + 3]);
        fprintf(stderr, "stamps wave %d: tiles %llu  pass %.0f  tail %.0f  total/tile %.0f ticks; %.2f ticks per ns\n",
                wv, hs[wv * 5 + 3], hs[wv * 5] / n, hs[wv * 5 + 1] / n, hs[wv * 5 + 2] / n,
                (double)hs[wv * 5 + 2] / (10.0 * (double)std::max(1ull, hs[wv * 5 + 4])));
        fprintf(stderr, "   per tile: vmcnt wait %.0f, barrier wait %.0f ticks\n", (double)(hs[40 + wv] >> 32) / n,
                (double)(hs[40 + wv] & 0xffffffffull) / n);
      }
  }
#endif
  return AMLIKE_OK;
}

}  // extern "C"
