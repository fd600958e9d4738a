// nnet2 forward pass on gfx950 (MI355X): the HIP matrix backend that replaces the
// reference's src/cudamatrix for the ops nnet2's Propagate() uses, and the
// amlike_nnet_* C ABI.
//
// What it replaces (paths relative to /root/reference/src):
//   nnet2/nnet-compute.cc:62-106,158-165     NnetComputer (pad, layer loop), NnetComputation()
//   nnet2/nnet-component.cc:2411-2471         SpliceComponent::Propagate    (CopyRows per offset)
//   nnet2/nnet-component.cc:1123-1131,3028-   Affine / FixedAffine          (cublasSgemm + bias kernels)
//   nnet2/nnet-component.cc:530-535,578-592   Pnorm (GroupPnorm), Normalize (AddDiagMat2, pow, MulRowsVec)
//   nnet2/nnet-component.cc:905-918           Softmax (ApplySoftMaxPerRow + ApplyFloor)
//   nnet2/nnet-component.cc:2352-2357         SumGroup (SumColumnRanges)
//   nnet2bin/nnet-am-compute.cc:93-96         floor + log
//   nnet2/decodable-am-nnet.h:59-69           floor, log, minus log prior, scale
//
// Kernels (each launch covers ALL frames of a batch of utterances):
//   nnet_pad_splice_kernel   edge-frame padding + frame splicing in one gather
//   nnet_gemm_bias_bf16x3_kernel  Y = X W^T + b on v_mfma_f32_32x32x16_bf16 with bf16x3-split
//                            operands (fp32-grade results), 128x128 block tile, LDS double
//                            buffer, one barrier per k-step
//   nnet_gemm_bias_kernel    the same on v_mfma_f32_32x32x2_f32 (exact fp32 inputs;
//                            AMLIKE_NNET_GEMM=f32, kept as the A/B check)
//   nnet_pnorm_norm_kernel   p-norm over groups of consecutive columns, optionally fused
//                            with the row normalisation that follows it (one HBM pass)
//   nnet_softmax_kernel      row softmax with the row held in registers (one read), fused
//                            with floor / log / minus log-prior / scale when it is last
//   nnet_sumgroup_kernel, nnet_elementwise_kernel, nnet_normalize_kernel, nnet_post_kernel
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "amlike_common.h"

namespace amlike {

typedef float f32x16 __attribute__((ext_vector_type(16)));


static inline int RoundUp(int x, int m) { return (x + m - 1) / m * m; }
// Results larger than this are streamed past the caches (non-temporal stores).
constexpr size_t kStreamingBytes = (size_t)128 << 20;

// Per-utterance bookkeeping of a batch: rows [in_off, in_off + in_rows) of the stage
// input become rows [out_off, out_off + out_rows) of its output.
struct UttMap {
  int32_t in_off, in_rows, out_off, out_rows;
};

// out row r of utterance u = [x[r - left_pad + 0] | ... | x[r - left_pad + ns - 1]] over the
// first `dim` columns (frame indices clamped to the utterance: the edge frames are
// repeated, nnet2/nnet-compute.cc:74-86), then the const tail of frame r - left_pad.
// With ns = 1 and const_dim = 0 this is a plain padded copy.
__global__ void __launch_bounds__(256)
    nnet_pad_splice_kernel(const float *__restrict__ in, int ld_in, const UttMap *__restrict__ utts,
                           int num_utts, int left_pad, int ns, int dim, int const_dim,
                           float *__restrict__ out, int ld_out, int out_dim) {
  const int u = blockIdx.y;
  const UttMap um = utts[u];
  for (int r = blockIdx.x; r < um.out_rows; r += gridDim.x) {
    float *o = out + (size_t)(um.out_off + r) * ld_out;
    for (int c = threadIdx.x; c < ld_out; c += 256) {
      float v = 0.0f;
      if (c < ns * dim) {
        const int k = c / dim, d = c - k * dim;
        int src = r - left_pad + k;
        src = min(max(src, 0), um.in_rows - 1);
        v = in[(size_t)(um.in_off + src) * ld_in + d];
      } else if (c < out_dim) {
        int src = min(max(r - left_pad, 0), um.in_rows - 1);
        v = in[(size_t)(um.in_off + src) * ld_in + dim + (c - ns * dim)];
      }
      o[c] = v;
    }
  }
}

// SpliceMaxComponent (nnet2/nnet-component.cc:2623-2654) with the same edge handling: out row r of
// utterance u = per column the largest of x[r - left_pad + 0 .. r - left_pad + ns - 1], frame
// indices clamped to the utterance.  Columns dim .. ld_out-1 are zeroed when zero_pad is set.
__global__ void __launch_bounds__(256)
    nnet_pad_splicemax_kernel(const float *__restrict__ in, int ld_in, const UttMap *__restrict__ utts,
                              int num_utts, int left_pad, int ns, int dim, float *__restrict__ out,
                              int ld_out, int zero_pad) {
  const int u = blockIdx.y;
  const UttMap um = utts[u];
  for (int r = blockIdx.x; r < um.out_rows; r += gridDim.x) {
    float *o = out + (size_t)(um.out_off + r) * ld_out;
    for (int c = threadIdx.x; c < (zero_pad ? ld_out : dim); c += 256) {
      float v = 0.0f;
      if (c < dim) {
        for (int k = 0; k < ns; ++k) {
          const int src = min(max(r - left_pad + k, 0), um.in_rows - 1);
          const float x = in[(size_t)(um.in_off + src) * ld_in + c];
          v = k == 0 ? x : (v < x ? x : v);  // Max(): the other operand where it is larger
        }
      }
      o[c] = v;
    }
  }
}

// Y[M x N] = X[M x K] * W[N x K]^T + bias.  X rows are K-contiguous with stride ldx
// (multiple of 4, columns K..Kp-1 hold zeros); W is stored [N][Kp], zero padded.
// 256 threads = 4 waves in a 2 x 2 arrangement, each wave a 64 x 64 tile = 2 x 2 MFMA
// tiles of 32 x 32; k-step 8.  Lane (i, h) reads 4 consecutive k (offset 4h) of row i
// with one ds_read_b128; the four k-values feed four v_mfma_f32_32x32x2_f32 whose
// k-slots are {k, k+4} -- the same (permuted) k order for both operands.
__global__ void __launch_bounds__(256)
    nnet_gemm_bias_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W,
                          int Kp, const float *__restrict__ bias, float *__restrict__ Y, int ldy,
                          int M, int N) {
  __shared__ __attribute__((aligned(16))) float4 s_a[2][256];  // [buf][row*2 + kq]
  __shared__ __attribute__((aligned(16))) float4 s_b[2][256];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1, li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int lrow = t >> 1, lkq = t & 1;
  const float4 *xa = reinterpret_cast<const float4 *>(X + (size_t)min(m0 + lrow, M - 1) * ldx) + lkq;
  const float4 *wb = reinterpret_cast<const float4 *>(W + (size_t)min(n0 + lrow, N - 1) * Kp) + lkq;
  const int nk = Kp >> 3;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  float4 ra = xa[0], rb = wb[0];
  s_a[0][t] = ra;
  s_b[0][t] = rb;
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      ra = xa[(ks + 1) * 2];
      rb = wb[(ks + 1) * 2];
    }
    float4 fa[2], fb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) fa[mt] = s_a[cur][(wr * 64 + mt * 32 + li) * 2 + h];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) fb[nt] = s_b[cur][(wc * 64 + nt * 32 + li) * 2 + h];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].x, fb[nt].x, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].y, fb[nt].y, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].z, fb[nt].z, acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].w, fb[nt].w, acc[mt][nt], 0, 0, 0);
      }
    if (ks + 1 < nk) {
      s_a[cur ^ 1][t] = ra;
      s_b[cur ^ 1][t] = rb;
    }
    __syncthreads();
  }
  // Epilogue: bias, store.  Register r of tile (mt, nt): row (r&3) + 8 (r>>2) + 4h, column li.
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + wc * 64 + nt * 32 + li;
    if (n >= N) continue;
    const float bv = bias[n];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) Y[(size_t)m * ldy + n] = acc[mt][nt][r] + bv;
      }
  }
}


// The same product on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the rate of the
// fp32-input instruction): every fp32 operand is written as the exact sum of three bf16
// values and a product is accumulated from the six partial products a1b1, a1b2, a2b1,
// a1b3, a2b2, a3b1 (the three dropped ones are below 2^-24 of the product), which gives
// fp32-grade results on 6/16 of the fp32 pipe's time.  W arrives pre-split
// ([3 planes][N][Kp] bf16, Kp a multiple of 16, zero padded); X is fp32 and is split
// while its tile is staged into LDS.
//
// 256 threads = 4 waves in a 2 x 2 arrangement, each wave a 64 x 64 tile = 2 x 2 MFMA tiles;
// k-step 16.  LDS tiles are [plane][block of 32 rows][k-octet][32 rows] x 16 bytes, i.e.
// exactly the order in which the 64 lanes of a wave read an MFMA operand (lane (r, h): 8
// consecutive k, octet h, of row r): ds_read_b128 at base + lane, conflict-free.
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

// Two floats -> three words of packed bf16 pairs (first, second, third term of each).
__device__ __forceinline__ void split_bf16x3_2(float a, float b, unsigned &o1, unsigned &o2,
                                               unsigned &o3) {
  const __bf16 a1 = static_cast<__bf16>(a), b1 = static_cast<__bf16>(b);
  const float ra = a - static_cast<float>(a1), rb = b - static_cast<float>(b1);
  const __bf16 a2 = static_cast<__bf16>(ra), b2 = static_cast<__bf16>(rb);
  const float sa = ra - static_cast<float>(a2), sb = rb - static_cast<float>(b2);
  const __bf16 a3 = static_cast<__bf16>(sa), b3 = static_cast<__bf16>(sb);
  o1 = __builtin_bit_cast(unsigned short, a1) | (unsigned)__builtin_bit_cast(unsigned short, b1) << 16;
  o2 = __builtin_bit_cast(unsigned short, a2) | (unsigned)__builtin_bit_cast(unsigned short, b2) << 16;
  o3 = __builtin_bit_cast(unsigned short, a3) | (unsigned)__builtin_bit_cast(unsigned short, b3) << 16;
}
__device__ __forceinline__ void split_bf16x3_8(const float4 &lo, const float4 &hi, uint4 &o1,
                                               uint4 &o2, uint4 &o3) {
  split_bf16x3_2(lo.x, lo.y, o1.x, o2.x, o3.x);
  split_bf16x3_2(lo.z, lo.w, o1.y, o2.y, o3.y);
  split_bf16x3_2(hi.x, hi.y, o1.z, o2.z, o3.z);
  split_bf16x3_2(hi.z, hi.w, o1.w, o2.w, o3.w);
}

__global__ void __launch_bounds__(256, 2)
    nnet_gemm_bias_bf16x3_kernel(const float *__restrict__ X, int ldx,
                                 const uint4 *__restrict__ Wp, int Kp,
                                 const float *__restrict__ bias, float *__restrict__ Y, int ldy,
                                 int M, int N) {
  // [buf][plane][row block][octet][32] : 2 * 3 * 4 * 2 * 32 * 16 B = 24 KiB per operand
  __shared__ __attribute__((aligned(16))) uint4 s_x[2][3][4][2][32];
  __shared__ __attribute__((aligned(16))) uint4 s_w[2][3][4][2][32];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1, li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  // Staging assignment: wave w stages row block w of both operands; lane (li, h) -> row li,
  // octet h.
  const int srow = w * 32 + li;
  const float4 *xa = reinterpret_cast<const float4 *>(X + (size_t)min(m0 + srow, M - 1) * ldx) + 2 * h;
  const size_t wrow = (size_t)min(n0 + srow, N - 1) * (Kp >> 3) + h;  // in uint4 units
  const size_t wplane = (size_t)N * (Kp >> 3);
  const int nk = Kp >> 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  float4 rx0 = xa[0], rx1 = xa[1];
  uint4 rw0 = Wp[wrow], rw1 = Wp[wplane + wrow], rw2 = Wp[2 * wplane + wrow];
  {
    uint4 sx0, sx1, sx2;
    split_bf16x3_8(rx0, rx1, sx0, sx1, sx2);
    s_x[0][0][w][h][li] = sx0;
    s_x[0][1][w][h][li] = sx1;
    s_x[0][2][w][h][li] = sx2;
    s_w[0][0][w][h][li] = rw0;
    s_w[0][1][w][h][li] = rw1;
    s_w[0][2][w][h][li] = rw2;
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      rx0 = xa[(ks + 1) * 4];
      rx1 = xa[(ks + 1) * 4 + 1];
      const size_t o = wrow + (size_t)(ks + 1) * 2;
      rw0 = Wp[o];
      rw1 = Wp[wplane + o];
      rw2 = Wp[2 * wplane + o];
    }
    bf16x8_t fa[2][3], fb[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        fa[mt][q] = __builtin_bit_cast(bf16x8_t, s_x[cur][q][wr * 2 + mt][h][li]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        fb[nt][q] = __builtin_bit_cast(bf16x8_t, s_w[cur][q][wc * 2 + nt][h][li]);
    // small terms first; the four accumulators alternate so that consecutive matrix
    // instructions are independent
#pragma unroll
    for (int which = 0; which < 6; ++which) {
      const int pa = which == 0 ? 2 : (which == 1 || which == 3) ? 1 : 0;
      const int pb = (which == 0 || which == 3 || which == 5) ? 0 : (which == 1 || which == 4) ? 1 : 2;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][pa], fb[nt][pb], acc[mt][nt], 0, 0, 0);
    }
    if (ks + 1 < nk) {
      uint4 sx0, sx1, sx2;
      split_bf16x3_8(rx0, rx1, sx0, sx1, sx2);
      s_x[cur ^ 1][0][w][h][li] = sx0;
      s_x[cur ^ 1][1][w][h][li] = sx1;
      s_x[cur ^ 1][2][w][h][li] = sx2;
      s_w[cur ^ 1][0][w][h][li] = rw0;
      s_w[cur ^ 1][1][w][h][li] = rw1;
      s_w[cur ^ 1][2][w][h][li] = rw2;
    }
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + wc * 64 + nt * 32 + li;
    if (n >= N) continue;
    const float bv = bias[n];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) Y[(size_t)m * ldy + n] = acc[mt][nt][r] + bv;
      }
  }
}

// fp16x2: 8 floats -> two words sets of packed fp16 pairs, x = p1 + p2 to 22 bits.  Used where
// the values are known to sit inside the fp16 range (activations behind a NormalizeComponent,
// |x| <= sqrt(dim); weights checked at load): 3 partial products instead of 6.
__device__ __forceinline__ void split_f16x2_2(float a, float b, unsigned &o1, unsigned &o2) {
  const _Float16 a1 = static_cast<_Float16>(a), b1 = static_cast<_Float16>(b);
  const _Float16 a2 = static_cast<_Float16>(a - static_cast<float>(a1));
  const _Float16 b2 = static_cast<_Float16>(b - static_cast<float>(b1));
  o1 = __builtin_bit_cast(unsigned short, a1) | (unsigned)__builtin_bit_cast(unsigned short, b1) << 16;
  o2 = __builtin_bit_cast(unsigned short, a2) | (unsigned)__builtin_bit_cast(unsigned short, b2) << 16;
}
__device__ __forceinline__ void split_f16x2_8(const float4 &lo, const float4 &hi, uint4 &o1, uint4 &o2) {
  split_f16x2_2(lo.x, lo.y, o1.x, o2.x);
  split_f16x2_2(lo.z, lo.w, o1.y, o2.y);
  split_f16x2_2(hi.x, hi.y, o1.z, o2.z);
  split_f16x2_2(hi.z, hi.w, o1.w, o2.w);
}
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// One MFMA of a split product: PREC = 1 fp16 operands, 0 bf16 (both arrive as raw 16-bit lanes).
template <int PREC>
__device__ __forceinline__ f32x16 mfma16(bf16x8_t a, bf16x8_t b, f32x16 c) {
  if (PREC)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// Partial product `which` of a split product, small terms first: parts (pa, pb) of the two
// operands.  bf16x3: 6 products; fp16x2: 3.
template <int PREC>
__device__ __forceinline__ void product_parts(int which, int &pa, int &pb) {
  if (PREC) {
    pa = which == 0 ? 1 : 0;
    pb = which == 1 ? 1 : 0;
  } else {
    pa = which == 0 ? 2 : (which == 1 || which == 3) ? 1 : 0;
    pb = (which == 0 || which == 3 || which == 5) ? 0 : (which == 1 || which == 4) ? 1 : 2;
  }
}

// ---------------------------------------------------------------------------
// Activations as bf16x3 planes.  A matrix that feeds an affine stage is also written as three
// bf16 planes, x = p1 + p2 + p3 exactly, each [rows][Kp] with Kp a multiple of 16 and the
// columns beyond the dimension zero: 8 consecutive k of one row are then one 16-byte MFMA
// operand fragment, and the GEMM stages its X tile without any conversion work (measured:
// splitting inside the GEMM's k-loop costs 40 % of its time, the separate pass 3 %).
struct Planes {
  unsigned short *p;  // plane q starts at p + q * stride
  size_t stride;      // elements per plane = rows * Kp
  int Kp;
};

__device__ __forceinline__ void store_planes(const Planes &o, size_t idx, float x) {
  const __bf16 b1 = static_cast<__bf16>(x);
  const float r1 = x - static_cast<float>(b1);
  const __bf16 b2 = static_cast<__bf16>(r1);
  const float r2 = r1 - static_cast<float>(b2);
  const __bf16 b3 = static_cast<__bf16>(r2);
  o.p[idx] = __builtin_bit_cast(unsigned short, b1);
  o.p[o.stride + idx] = __builtin_bit_cast(unsigned short, b2);
  o.p[2 * o.stride + idx] = __builtin_bit_cast(unsigned short, b3);
}

// fp32 [rows x ld] (first dim columns) -> planes, zero padded, in k-octet-major order:
// plane q, octet o (columns 8o .. 8o+7), row r is the 16-byte element q * stride/8 + o * rows + r.
// With the rows of one octet contiguous, the GEMM's staging loads (32 rows x one octet per
// half-wave) read 512 contiguous bytes instead of 16 bytes from each of 32 cache lines.
// Layers whose input range is not known when the net is built run on fp16x2 operands "on
// trial": the split pass raises *flag when a value does not fit (|x| > 32768, NaN, Inf), the
// fp16x2 kernels (gmode 1) return at once when it is raised, and the bf16x3 twins launched
// behind them (gmode 2) return at once when it is not.  No host round trip.
__device__ __forceinline__ bool nnet_guard_skips(const int *flag, int gmode) {
  if (gmode == 0) return false;
  const bool raised = ctl_load(flag) != 0;
  return gmode == 1 ? raised : !raised;
}

template <int PREC>
__global__ void __launch_bounds__(256)
    nnet_split_planes_kernel(const float *__restrict__ in, int ld, int dim, int rows, Planes out,
                             int one_col, int *flag, int gmode) {
  if (gmode == 2 && nnet_guard_skips(flag, 2)) return;
  const int kp8 = out.Kp >> 3;
  const size_t n = (size_t)rows * kp8;
  uint4 *o = reinterpret_cast<uint4 *>(out.p);
  const size_t pstride = out.stride >> 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t oct = i / rows, r = i - oct * rows;
    const int c0 = (int)oct * 8;
    const float *x = in + r * ld + c0;
    float4 lo, hi;
    if (c0 + 8 <= dim) {
      lo = *reinterpret_cast<const float4 *>(x);
      hi = *reinterpret_cast<const float4 *>(x + 4);
    } else {
      // the tail octets: zero padding, and 1.0 in column one_col (the bias column of the
      // fused affine + p-norm GEMM; one_col < 0: none)
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = c0 + j < dim ? x[j] : (c0 + j == one_col ? 1.0f : 0.0f);
      lo = make_float4(t[0], t[1], t[2], t[3]);
      hi = make_float4(t[4], t[5], t[6], t[7]);
    }
    if (PREC) {
      if (gmode == 1) {
        const float m = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(lo.y)), fmaxf(fabsf(lo.z), fabsf(lo.w))),
                              fmaxf(fmaxf(fabsf(hi.x), fabsf(hi.y)), fmaxf(fabsf(hi.z), fabsf(hi.w))));
        const bool fits = (lo.x == lo.x) & (lo.y == lo.y) & (lo.z == lo.z) & (lo.w == lo.w) &
                          (hi.x == hi.x) & (hi.y == hi.y) & (hi.z == hi.z) & (hi.w == hi.w) & (m <= 32768.0f);
        if (!fits) ctl_or(flag, 1);
      }
      uint4 s0, s1;
      split_f16x2_8(lo, hi, s0, s1);
      o[i] = s0;
      o[pstride + i] = s1;
    } else {
      uint4 s0, s1, s2;
      split_bf16x3_8(lo, hi, s0, s1, s2);
      o[i] = s0;
      o[pstride + i] = s1;
      o[2 * pstride + i] = s2;
    }
  }
}

// SpliceComponent (with the edge padding of the first one) whose result only feeds an affine
// stage: the gathered rows are written straight as that stage's operand planes, in the layout
// of nnet_split_planes_kernel and with its range check / guard; the fp32 spliced matrix and
// the separate split pass disappear.  Element semantics as in nnet_pad_splice_kernel.
constexpr int kSpliceStageRows = 112;  // input rows a block of 64 output rows may stage
template <int PREC>
__global__ void __launch_bounds__(256)
    nnet_splice_planes_kernel(const float *__restrict__ in, int ld_in, const UttMap *__restrict__ utts,
                              int num_utts, int left_pad, int ns, int dim, int out_dim, int total_rows,
                              Planes out, int one_col, int *flag, int gmode) {
  if (gmode == 2 && nnet_guard_skips(flag, 2)) return;
  // One block = 64 consecutive output rows.  Their source rows form one short stretch of the
  // (back to back) input, which is brought into LDS with coalesced loads; thread (row, octet)
  // then assembles its 8 values from LDS.  in_dim = dim + const tail.
  extern __shared__ float s_spl[];
  __shared__ int s_src0[64], s_rows[64];
  const int in_dim = dim + (out_dim - ns * dim);
  const int indp = in_dim | 1;
  const int g0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int g = g0 + lane;
  UttMap um = utts[0];
  int r = 0;
  if (g < total_rows) {
    int lo = 0, hi = num_utts;  // the utterance output row g belongs to
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (utts[mid].out_off <= g) lo = mid; else hi = mid;
    }
    um = utts[lo];
    r = g - um.out_off;
  }
  // first and last input row this block needs (rows of the block are in increasing order)
  const int my_first = um.in_off + min(max(r - left_pad, 0), um.in_rows - 1);
  const int my_last = um.in_off + min(max(r - left_pad + ns - 1, 0), um.in_rows - 1);
  if (q == 0) {
    s_src0[lane] = g < total_rows ? my_first : 0x7fffffff;
    s_rows[lane] = g < total_rows ? my_last : -1;
  }
  __syncthreads();
  int first = 0x7fffffff, last = -1;
  for (int k = 0; k < 64; ++k) {
    first = min(first, s_src0[k]);
    last = max(last, s_rows[k]);
  }
  const int nstage = last - first + 1;
  const bool staged = nstage <= kSpliceStageRows;
  if (staged) {
    for (int i = threadIdx.x; i < nstage * in_dim; i += 256) {
      const int rr = i / in_dim, c = i - rr * in_dim;
      s_spl[rr * indp + c] = in[(size_t)(first + rr) * ld_in + c];
    }
  }
  __syncthreads();
  if (g >= total_rows) return;
  const int kp8 = out.Kp >> 3;
  uint4 *o = reinterpret_cast<uint4 *>(out.p);
  const size_t pstride = out.stride >> 3;
  for (int oct = q; oct < kp8; oct += 4) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = oct * 8 + j;
      float v = 0.0f;
      int src = -1, d = 0;
      if (c < ns * dim) {
        const int k = c / dim;
        d = c - k * dim;
        src = um.in_off + min(max(r - left_pad + k, 0), um.in_rows - 1);
      } else if (c < out_dim) {
        d = dim + (c - ns * dim);
        src = um.in_off + min(max(r - left_pad, 0), um.in_rows - 1);
      } else if (c == one_col) {
        v = 1.0f;
      }
      if (src >= 0) v = staged ? s_spl[(src - first) * indp + d] : in[(size_t)src * ld_in + d];
      t[j] = v;
    }
    const float4 lo4 = make_float4(t[0], t[1], t[2], t[3]), hi4 = make_float4(t[4], t[5], t[6], t[7]);
    const size_t i = (size_t)oct * total_rows + g;
    if (PREC) {
      if (gmode == 1) {
        bool fits = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) fits &= fabsf(t[j]) <= 32768.0f;  // false for NaN too
        if (!fits) ctl_or(flag, 1);
      }
      uint4 s0, s1;
      split_f16x2_8(lo4, hi4, s0, s1);
      o[i] = s0;
      o[pstride + i] = s1;
    } else {
      uint4 s0, s1, s2;
      split_bf16x3_8(lo4, hi4, s0, s1, s2);
      o[i] = s0;
      o[pstride + i] = s1;
      o[2 * pstride + i] = s2;
    }
  }
}

// The bf16x3 GEMM with both operands given as k-octet-major planes (X: [3][Kp/8][M], W:
// [3][Kp/8][N] 16-byte elements): same tiling and LDS image as nnet_gemm_bias_bf16x3_kernel
// (block tile 128 x 128, three blocks per CU), but both operand tiles are plain, fully
// coalesced 16-byte copies global -> registers -> LDS.  (A 128 x 256 block tile with
// 64 x 128 wave tiles was measured 15 % slower: it drops to two blocks per CU.)
// NT_STORE: the result is written with non-temporal stores, for results far larger than the
// caches: a [35554 x 8600] result written normally evicts the operand planes from L2 while
// the kernel still re-reads them (670 -> 525 us on that shape).  With the stores compiled out
// the same launch takes 377 us, with a single k-step 316 us.
template <int PREC, bool NT_STORE>
__global__ void __launch_bounds__(256, 2)
    nnet_gemm_bias_planes_kernel(const uint4 *__restrict__ Xp, size_t xplane,
                                 const uint4 *__restrict__ Wp, size_t wplane, size_t woct, int Kp,
                                 const float *__restrict__ bias, float *__restrict__ Y, int ldy,
                                 int M, int N, const int *flag, int gmode) {
  constexpr int NPL = PREC ? 2 : 3, NPROD = PREC ? 3 : 6;
  if (nnet_guard_skips(flag, gmode)) return;
  __shared__ __attribute__((aligned(16))) uint4 s_x[2][NPL][4][2][32];
  __shared__ __attribute__((aligned(16))) uint4 s_w[2][NPL][4][2][32];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = w >> 1, wc = w & 1, li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int srow = w * 32 + li;
  const int kp8 = Kp >> 3, nk = Kp >> 4;
  // element of (row, octet o): o * rows + row
  const size_t xrow = (size_t)h * M + min(m0 + srow, M - 1);
  // W may be a row range of a larger image: woct = rows of that image (the octet stride),
  // wplane its plane stride
  const size_t wrow = (size_t)h * woct + min(n0 + srow, N - 1);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  uint4 rx0 = Xp[xrow], rx1 = Xp[xplane + xrow], rx2 = rx0;
  uint4 rw0 = Wp[wrow], rw1 = Wp[wplane + wrow], rw2 = rw0;
  if (NPL == 3) {
    rx2 = Xp[2 * xplane + xrow];
    rw2 = Wp[2 * wplane + wrow];
  }
  s_x[0][0][w][h][li] = rx0;
  s_x[0][1][w][h][li] = rx1;
  s_w[0][0][w][h][li] = rw0;
  s_w[0][1][w][h][li] = rw1;
  if (NPL == 3) {
    s_x[0][NPL - 1][w][h][li] = rx2;
    s_w[0][NPL - 1][w][h][li] = rw2;
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) {
      const size_t ox = xrow + (size_t)(ks + 1) * 2 * M, ow = wrow + (size_t)(ks + 1) * 2 * woct;
      rx0 = Xp[ox];
      rx1 = Xp[xplane + ox];
      rw0 = Wp[ow];
      rw1 = Wp[wplane + ow];
      if (NPL == 3) {
        rx2 = Xp[2 * xplane + ox];
        rw2 = Wp[2 * wplane + ow];
      }
    }
    bf16x8_t fa[2][NPL], fb[2][NPL];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        fa[mt][q] = __builtin_bit_cast(bf16x8_t, s_x[cur][q][wr * 2 + mt][h][li]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        fb[nt][q] = __builtin_bit_cast(bf16x8_t, s_w[cur][q][wc * 2 + nt][h][li]);
#pragma unroll
    for (int which = 0; which < NPROD; ++which) {
      int pa, pb;
      product_parts<PREC>(which, pa, pb);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16<PREC>(fa[mt][pa], fb[nt][pb], acc[mt][nt]);
    }
    if (ks + 1 < nk) {
      s_x[cur ^ 1][0][w][h][li] = rx0;
      s_x[cur ^ 1][1][w][h][li] = rx1;
      s_w[cur ^ 1][0][w][h][li] = rw0;
      s_w[cur ^ 1][1][w][h][li] = rw1;
      if (NPL == 3) {
        s_x[cur ^ 1][NPL - 1][w][h][li] = rx2;
        s_w[cur ^ 1][NPL - 1][w][h][li] = rw2;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = n0 + wc * 64 + nt * 32 + li;
    if (n >= N) continue;
    const float bv = bias[n];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) {
          if (NT_STORE) __builtin_nontemporal_store(acc[mt][nt][r] + bv, &Y[(size_t)m * ldy + n]);
          else Y[(size_t)m * ldy + n] = acc[mt][nt][r] + bv;
        }
      }
  }
}

// Affine + p-norm (p = 2) in one kernel: out[t][g] = sqrt(sum_{j < GS} (W x_t + b)[g GS + j]^2).
// The operand roles are swapped against the kernels above: the units (rows of W) are the
// MFMA's A operand, so they land on the accumulator's register axis, and the frames are its
// columns (one per lane).  The host permutes the rows of W (BuildPnormWeights) so that the
// 80 accumulator values a half-lane holds over five 32-unit tiles are 80 / GS complete
// p-norm groups: the group sums are then plain per-lane arithmetic, and the kernel writes
// [T x groups] instead of [T x units] -- a tenth of the bytes at GS = 10, and the separate
// p-norm pass over [T x units] disappears.  The bias rides in W as column K against a
// constant 1.0 column of X (nnet_split_planes_kernel, one_col).
// Block: 160 units x 128 frames, 4 waves, wave w = frames [32 w, 32 w + 32) x all 160 units.
template <int GS, int PREC>
__global__ void __launch_bounds__(256, 2)
    nnet_gemm_pnorm_kernel(const uint4 *__restrict__ Xp, size_t xplane, const uint4 *__restrict__ Wp,
                           int Kp, float *__restrict__ Y, int ldy, int M, int Npad, int G,
                           const int *flag, int gmode, int xcd_order) {
  static_assert(80 % GS == 0, "a half-lane holds 80 units");
  if (nnet_guard_skips(flag, gmode)) return;
  constexpr int NPL = PREC ? 2 : 3, NPROD = PREC ? 3 : 6;
  __shared__ __attribute__((aligned(16))) uint4 s_x[2][NPL][4][64];
  __shared__ __attribute__((aligned(16))) uint4 s_w[2][NPL][5][64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  // Which (unit block, frame block) this workgroup computes.  Workgroups are dealt round-robin
  // over the 8 XCDs in launch order; with xcd_order each XCD gets a contiguous run of the
  // (frame block, unit block) sequence, unit block fastest, so that the gridDim.x workgroups
  // that read one 128-frame panel of X run on one XCD at about the same time: the panel comes
  // from memory once and from that XCD's L2 afterwards (W, 1-3 MB, lives in every L2).
  int bxu = blockIdx.x, bym = blockIdx.y;
  if (xcd_order) {
    const int total = gridDim.x * gridDim.y, L = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    bym = Lp / gridDim.x;
    bxu = Lp - bym * gridDim.x;
  }
  const int u0 = bxu * 160, m0 = bym * 128;
  const int kp8 = Kp >> 3, nk = Kp >> 4;
  const size_t wplane = (size_t)Npad * kp8;
  // staging: wave w copies X row block w and W row blocks w and w + 4 (w + 4 exists for w = 0)
  const size_t xrow = (size_t)h * M + min(m0 + w * 32 + li, M - 1);
  const size_t wrow0 = (size_t)h * Npad + u0 + w * 32 + li;
  const size_t wrow1 = (size_t)h * Npad + u0 + 128 + li;  // row block 4, staged by wave 0
  const bool second = w == 0;

  f32x16 acc[5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;

  uint4 rx0, rx1, rx2, ra0, ra1, ra2, rb0, rb1, rb2;
  rx2 = ra2 = rb0 = rb1 = rb2 = make_uint4(0, 0, 0, 0);
  auto fetch = [&](int ks) {
    const size_t ox = xrow + (size_t)ks * 2 * M, oa = wrow0 + (size_t)ks * 2 * Npad;
    rx0 = Xp[ox];
    rx1 = Xp[xplane + ox];
    ra0 = Wp[oa];
    ra1 = Wp[wplane + oa];
    if (NPL == 3) {
      rx2 = Xp[2 * xplane + ox];
      ra2 = Wp[2 * wplane + oa];
    }
    if (second) {
      const size_t ob = wrow1 + (size_t)ks * 2 * Npad;
      rb0 = Wp[ob];
      rb1 = Wp[wplane + ob];
      if (NPL == 3) rb2 = Wp[2 * wplane + ob];
    }
  };
  auto commit = [&](int buf) {
    s_x[buf][0][w][lane] = rx0;
    s_x[buf][1][w][lane] = rx1;
    s_w[buf][0][w][lane] = ra0;
    s_w[buf][1][w][lane] = ra1;
    if (NPL == 3) {
      s_x[buf][NPL - 1][w][lane] = rx2;
      s_w[buf][NPL - 1][w][lane] = ra2;
    }
    if (second) {
      s_w[buf][0][4][lane] = rb0;
      s_w[buf][1][4][lane] = rb1;
      if (NPL == 3) s_w[buf][NPL - 1][4][lane] = rb2;
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < nk) fetch(ks + 1);
    bf16x8_t fw[5][NPL], fx[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) fx[q] = __builtin_bit_cast(bf16x8_t, s_x[cur][q][w][lane]);
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int q = 0; q < NPL; ++q) fw[a][q] = __builtin_bit_cast(bf16x8_t, s_w[cur][q][a][lane]);
#pragma unroll
    for (int which = 0; which < NPROD; ++which) {
      int pa, pb;
      product_parts<PREC>(which, pa, pb);
#pragma unroll
      for (int a = 0; a < 5; ++a) acc[a] = mfma16<PREC>(fw[a][pa], fx[pb], acc[a]);
    }
    if (ks + 1 < nk) commit(cur ^ 1);
    __syncthreads();
  }
  // Epilogue: this lane is frame m0 + 32 w + li; its 80 values (tile a, register r) are slots
  // a * 16 + r = GS-sized groups number (80 / GS) * (2 * unit block + h) + slot / GS.
  const int m = m0 + w * 32 + li;
  if (m >= M) return;
  float *o = Y + (size_t)m * ldy;
  const int g0 = (80 / GS) * (2 * bxu + h);
#pragma unroll
  for (int gl = 0; gl < 80 / GS; ++gl) {
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      const int slot = gl * GS + j;
      const float v = acc[slot >> 4][slot & 15];
      ss = fmaf(v, v, ss);
    }
    if (g0 + gl < G) o[g0 + gl] = sqrtf(ss);
  }
}

// Zeroes columns [dim, ld) of a [rows x ld] buffer (the k-padding the GEMM reads).
__global__ void nnet_zero_pad_kernel(float *buf, int rows, int ld, int dim) {
  const int padw = ld - dim;
  const size_t n = (size_t)rows * padw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / padw;
    buf[r * ld + dim + (i - r * padw)] = 0.0f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// One wave per row: out[j] = (sum_{k<gs} |x[j gs + k]|^p)^(1/p)  (matrix/kaldi-vector.cc:529-566),
// then, when FUSE_NORM, out *= (max(sum_j out_j^2 / D, 2^-66))^-0.5  (nnet-component.cc:578-592);
// the fused form keeps a lane's outputs in registers and needs out_dim <= 64 * kPnormKeep.
constexpr int kPnormKeep = 8;
__device__ __forceinline__ float group_pnorm(const float *g, int gs, float p) {
  if (p == 2.0f) {
    float s = 0.0f;
    for (int k = 0; k < gs; ++k) s += g[k] * g[k];
    return sqrtf(s);
  }
  if (p == 1.0f) {
    float s = 0.0f;
    for (int k = 0; k < gs; ++k) s += fabsf(g[k]);
    return s;
  }
  if (p == 0.0f) {
    float s = 0.0f;
    for (int k = 0; k < gs; ++k) s += g[k] != 0.0f ? 1.0f : 0.0f;
    return s;
  }
  // general p in the scaled form, which cannot overflow (the reference reaches the same
  // value through its rescue branch when pow() overflows)
  float mx = 0.0f;
  for (int k = 0; k < gs; ++k) mx = fmaxf(mx, fabsf(g[k]));
  if (!(mx > 0.0f)) return 0.0f;
  float s = 0.0f;
  for (int k = 0; k < gs; ++k) s += powf(fabsf(g[k]) / mx, p);
  return mx * powf(s, 1.0f / p);
}

template <bool FUSE_NORM>
__global__ void __launch_bounds__(256)
    nnet_pnorm_norm_kernel(const float *__restrict__ in, int ld_in, int in_dim, int rows,
                           int out_dim, float p, float *__restrict__ out, int ld_out,
                           int zero_pad) {
  extern __shared__ __attribute__((aligned(16))) float s_row[];  // [4 waves][in_dim]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gs = in_dim / out_dim;
  float *row = s_row + (size_t)w * in_dim;
  for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
    const float *x = in + (size_t)r * ld_in;
    for (int c = lane; c < in_dim; c += 64) row[c] = x[c];
    __builtin_amdgcn_wave_barrier();
    float *o = out + (size_t)r * ld_out;
    if (FUSE_NORM) {
      float keep[kPnormKeep];
      float ss = 0.0f;
#pragma unroll
      for (int k = 0; k < kPnormKeep; ++k) {
        const int jg = lane + 64 * k;
        keep[k] = jg < out_dim ? group_pnorm(row + jg * gs, gs, p) : 0.0f;
        ss += keep[k] * keep[k];
      }
      ss = wave_sum(ss);
      const float f = 1.0f / sqrtf(fmaxf(ss / out_dim, 1.3552527156068805e-20f));  // 2^-66
#pragma unroll
      for (int k = 0; k < kPnormKeep; ++k) {
        const int jg = lane + 64 * k;
        if (jg < out_dim) o[jg] = keep[k] * f;
      }
    } else {
      for (int jg = lane; jg < out_dim; jg += 64) o[jg] = group_pnorm(row + jg * gs, gs, p);
    }
    if (zero_pad)
      for (int c = out_dim + lane; c < ld_out; c += 64) o[c] = 0.0f;
    __builtin_amdgcn_wave_barrier();
  }
}

// Stand-alone NormalizeComponent: one wave per row.
__global__ void __launch_bounds__(256)
    nnet_normalize_kernel(const float *__restrict__ in, int ld_in, int dim, int rows,
                          float *__restrict__ out, int ld_out, int zero_pad) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
    const float *x = in + (size_t)r * ld_in;
    float ss = 0.0f;
    for (int c = lane; c < dim; c += 64) ss += x[c] * x[c];
    ss = wave_sum(ss);
    const float f = 1.0f / sqrtf(fmaxf(ss / dim, 1.3552527156068805e-20f));
    float *o = out + (size_t)r * ld_out;
    for (int c = lane; c < dim; c += 64) o[c] = x[c] * f;
    if (zero_pad)
      for (int c = dim + lane; c < ld_out; c += 64) o[c] = 0.0f;
  }
}

// NormalizeComponent whose result only feeds an affine stage on fp16x2 operands: the scaled
// rows are written straight as that stage's operand planes (k-octet-major, zero padded to Kp,
// 1.0 in column one_col when the consumer carries its bias as a k column), and the fp32 copy
// plus the separate split pass disappear.  One block = RB consecutive rows, brought into LDS
// with coalesced loads (row stride odd, so a thread walking one row meets no bank conflict);
// then one scale per row from 256 / RB partial sums, then thread (row, octet) work with the rows
// of an octet contiguous in the output, as in nnet_split_planes_kernel.
template <int RB>
__global__ void __launch_bounds__(256)
    nnet_normalize_planes_kernel(const float *__restrict__ in, int ld_in, int dim, int rows,
                                 Planes out, int one_col, const int *flag, int gmode) {
  constexpr int NQ = 256 / RB;  // threads per row
  if (nnet_guard_skips(flag, gmode)) return;
  extern __shared__ float s_norm[];
  const int dimp = dim | 1;
  float *s_x = s_norm;              // [RB][dimp]
  float *s_p = s_norm + RB * dimp;  // [NQ][RB] partial sums, then [RB] scales
  const int r0 = blockIdx.x * RB;
  const int nrow = min(RB, rows - r0);
  for (int i = threadIdx.x; i < nrow * dim; i += 256) {
    const int rr = i / dim, c = i - rr * dim;
    s_x[rr * dimp + c] = in[(size_t)(r0 + rr) * ld_in + c];
  }
  __syncthreads();
  const int rr = threadIdx.x % RB, q = threadIdx.x / RB;
  {
    float ss = 0.0f;
    if (rr < nrow)
      for (int c = q; c < dim; c += NQ) ss += s_x[rr * dimp + c] * s_x[rr * dimp + c];
    s_p[q * RB + rr] = ss;
  }
  __syncthreads();
  if (threadIdx.x < RB) {
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < NQ; ++k) ss += s_p[k * RB + threadIdx.x];
    s_p[threadIdx.x] = 1.0f / sqrtf(fmaxf(ss / dim, 1.3552527156068805e-20f));
  }
  __syncthreads();
  if (rr >= nrow) return;
  const float f = s_p[rr];
  const int kp8 = out.Kp >> 3;
  uint4 *o = reinterpret_cast<uint4 *>(out.p);
  const size_t pstride = out.stride >> 3;
  const float *x = s_x + rr * dimp;
  for (int oct = q; oct < kp8; oct += NQ) {
    const int c0 = oct * 8;
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = c0 + j < dim ? x[c0 + j] * f : (c0 + j == one_col ? 1.0f : 0.0f);
    uint4 s0, s1;
    split_f16x2_8(make_float4(t[0], t[1], t[2], t[3]), make_float4(t[4], t[5], t[6], t[7]), s0, s1);
    const size_t i = (size_t)oct * rows + (r0 + rr);
    o[i] = s0;
    o[pstride + i] = s1;
  }
}

// mode bits for the fused output post-processing.
constexpr int kPostFloorLog = 1;  // max(v, 1e-20) then log
constexpr int kPostPrior = 2;     // minus log prior, times scale

// ApplyFloor(1e-20) as the reference does it (matrix/kaldi-matrix.cc: `if (x < floor) x = floor`):
// a NaN stays a NaN, where fmaxf would turn it into the floor.
__device__ __forceinline__ float floor_1e20(float x) { return x < 1.0e-20f ? 1.0e-20f : x; }

__device__ __forceinline__ float post_process(float v, int post, const float *log_priors, int c,
                                              float scale) {
  if (post & kPostFloorLog) v = __logf(floor_1e20(v));
  if (post & kPostPrior) v = (v - log_priors[c]) * scale;
  return v;
}

// Row softmax, one 256-thread block per row, the row held in registers (<= kSoftmaxRegs
// per thread, i.e. rows up to 256 * kSoftmaxRegs wide): max, exp, sum, scale by 1/sum, floor
// 1e-20 (nnet-component.cc:905-918), then the optional post-processing.
constexpr int kSoftmaxRegs = 40;
__global__ void __launch_bounds__(256)
    nnet_softmax_kernel(const float *__restrict__ in, int ld_in, int dim, int rows, int post,
                        const float *__restrict__ log_priors, float scale,
                        float *__restrict__ out, int ld_out, int zero_pad) {
  __shared__ float s_red[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *x = in + (size_t)r * ld_in;
    float v[kSoftmaxRegs];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < kSoftmaxRegs; ++k) {
      const int c = t + k * 256;
      v[k] = c < dim ? x[c] : -INFINITY;
      mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    if (lane == 0) s_red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    __syncthreads();
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < kSoftmaxRegs; ++k) {
      v[k] = __expf(v[k] - mx);  // exp(-inf) = 0 for the columns past dim
      sum += v[k];
    }
    sum = wave_sum(sum);
    if (lane == 0) s_red[w] = sum;
    __syncthreads();
    sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    const float inv = 1.0f / sum;
    float *o = out + (size_t)r * ld_out;
#pragma unroll
    for (int k = 0; k < kSoftmaxRegs; ++k) {
      const int c = t + k * 256;
      if (c < dim) o[c] = post_process(floor_1e20(v[k] * inv), post, log_priors, c, scale);
      else if (zero_pad && c < ld_out) o[c] = 0.0f;
    }
  }
}

// The same with 16-byte accesses: thread t holds columns 4 (t + 256 k) .. + 3; needs dim,
// both row strides and both base pointers to be multiples of 4 floats (the swbd output
// layer is: 8600).
template <int R4>
__global__ void __launch_bounds__(256)
    // (non-temporal loads and stores make no difference here: 5.4 TB/s either way)
    nnet_softmax4_kernel(const float *__restrict__ in, int ld_in, int dim, int rows, int post,
                         const float *__restrict__ log_priors, float scale,
                         float *__restrict__ out, int ld_out, int zero_pad) {
  __shared__ float s_red[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int dim4 = dim >> 2;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float4 *x = reinterpret_cast<const float4 *>(in + (size_t)r * ld_in);
    float4 v[R4];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < R4; ++k) {
      const int c = t + k * 256;
      v[k] = c < dim4 ? x[c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      mx = fmaxf(fmaxf(mx, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
    }
    mx = wave_max(mx);
    if (lane == 0) s_red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    __syncthreads();
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < R4; ++k) {
      v[k].x = __expf(v[k].x - mx);
      v[k].y = __expf(v[k].y - mx);
      v[k].z = __expf(v[k].z - mx);
      v[k].w = __expf(v[k].w - mx);
      sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    sum = wave_sum(sum);
    if (lane == 0) s_red[w] = sum;
    __syncthreads();
    sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    const float inv = 1.0f / sum;
    float4 *o = reinterpret_cast<float4 *>(out + (size_t)r * ld_out);
    const float4 *lp = reinterpret_cast<const float4 *>(log_priors);
#pragma unroll
    for (int k = 0; k < R4; ++k) {
      const int c = t + k * 256;
      if (c < dim4) {
        float4 q = make_float4(floor_1e20(v[k].x * inv), floor_1e20(v[k].y * inv),
                               floor_1e20(v[k].z * inv), floor_1e20(v[k].w * inv));
        if (post & kPostFloorLog) q = make_float4(__logf(q.x), __logf(q.y), __logf(q.z), __logf(q.w));
        if (post & kPostPrior) {
          const float4 p = lp[c];
          q = make_float4((q.x - p.x) * scale, (q.y - p.y) * scale, (q.z - p.z) * scale, (q.w - p.w) * scale);
        }
        o[c] = q;
      } else if (zero_pad && 4 * c < ld_out) {
        o[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
  }
}

// Output layer and softmax in one kernel (the last AffineComponent of a net followed by its
// SoftmaxComponent, nnet2/nnet-component.cc:1123-1131 and :905-918, with the floor / log /
// minus log-prior / scale of nnet2bin/nnet-am-compute.cc:93-96 and nnet2/decodable-am-nnet.h:57-73):
// the [rows x N] matrix is written once and never read back.  Two sweeps over the N units:
// the first forms the product tile by tile and keeps only each row's running (max, sum of
// exponentials); the second forms the product again and writes the finished value.  The second
// product costs far less than reading and re-writing the matrix (N = 8600: 34 KB per row).
//
// A wave owns FT tiles of 32 rows (frames) for a whole job and keeps their operand fragments
// of X in registers (frames are the matrix instruction's columns, one per lane); the units are
// its rows, so a lane holds 16 units of a 32-unit tile for its frame.  The W tiles stream
// through LDS (a ring of kOutRing tiles, global_load ... lds), shared by the block's 4 waves;
// one wave per SIMD, 512 registers.  The vector work of tile t - 1 (softmax statistics or
// output stores) is interleaved with the matrix instructions of tile t.
// A job = up to 256 consecutive rows (64 per wave, FT = 2) or up to 128 (32 per wave, FT = 1);
// the host cuts the rows into jobs so that the last round of jobs is short on every CU
// instead of long on a few, and lets half of the CUs start with a short job: the second sweep
// is bound by the writes (measured: 2.3 us per tile without its stores, 3.3 us with them,
// whatever their address pattern), the first writes nothing, and from then on the two halves
// of the chip are in different sweeps.
// Measured on the swbd shape (k = 200, 8600 units, 218.5 k rows): 4.9 ms against 6.8 ms for
// the separate GEMM and softmax kernels; matrix instructions alone 3.2 ms.
constexpr int kOutSideRing = 8;  // (32 biases, 32 log priors) of a tile, kept longer than the tile
// Result tiles on their way to memory: per wave 64 rows x 32 values, rows 36 floats apart
constexpr int kOutStagePitch = 36, kOutStageBytes = 4 * 64 * kOutStagePitch * 4;
#ifndef AMLIKE_OUT_RING_MAX
#define AMLIKE_OUT_RING_MAX 5
#endif
// W tiles of the LDS ring: as many as fit, up to AMLIKE_OUT_RING_MAX
__host__ __device__ constexpr int OutRingTiles(int KS, int PREC) {
  const int tile = (PREC ? 2 : 3) * 2 * KS * 32 * 16, room = 160 * 1024 - kOutSideRing * 256 - kOutStageBytes;
  return room / tile >= AMLIKE_OUT_RING_MAX ? AMLIKE_OUT_RING_MAX : room / tile;
}
// An empty volatile asm that takes and returns `x` in a vector register: instructions that
// produce x stay before it and instructions that use x stay after it in program order.
__device__ __forceinline__ void pin_order(float &x) {
#if __HIP_DEVICE_COMPILE__  // the host pass cannot parse the vector-register constraint
  asm volatile("" : "+v"(x));
#endif
}
// Waits until at most `N` of the wave's vector-memory instructions are outstanding.
template <int N>
__device__ __forceinline__ void wait_vector_memory_but() {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// LDS reads the compiler does not see (it would otherwise order every read of memory that
// global_load ... lds may write behind ALL such loads in flight): 16 bytes at `p`, valid after
// lds_hidden_wait on the same variables.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_hidden_read(f32x4 &v, const float *p) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p));
#endif
}
template <int OFFSET>
__device__ __forceinline__ void lds_hidden_read_at(f32x4 &v, unsigned addr) {  // byte address + constant
#if __HIP_DEVICE_COMPILE__
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET));
#endif
}
// The same for a register set that matrix instructions written as assembly read as an operand: `v`
// counts as in use from its previous contents on ("+v"), so the compiler gives those registers to
// nothing else between the last matrix instruction that reads them and this read.  (A matrix
// instruction fetches its operands when the pipe takes it up, tens of cycles after it was issued
// if others are queued in front; the compiler, which does not know the assembly is a matrix
// instruction, had put address arithmetic of the next step into the registers of the last
// fragment: every row's softmax sum came out wrong by a few 1e-3.)
template <int OFFSET>
__device__ __forceinline__ void lds_hidden_reload_at(f32x4 &v, unsigned addr) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFFSET));
#endif
}
template <int OFFSET>
__device__ __forceinline__ void lds_hidden_write_at(unsigned addr, const f32x4 &v) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFFSET) : "memory");
#endif
}
// Waits until at most N of the wave's LDS reads are outstanding; a and b are ready afterwards.
template <int N>
__device__ __forceinline__ void lds_hidden_wait_but(f32x4 &a, f32x4 &b) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
#endif
}
__device__ __forceinline__ void lds_hidden_wait1(f32x4 &a) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a));
#endif
}
__device__ __forceinline__ void lds_hidden_wait(f32x4 (&a)[4], f32x4 (&b)[4]) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
#endif
}
// The fp16 matrix instruction, acc += a . b, as the assembler sees it: the accumulator and
// (B_AGPR) the b operand are kept in the accumulation half of the register file, which the
// compiler's own allocation only uses as overflow space, with a copy per use.  The compiler
// does not know these are matrix instructions: the caller keeps the required distance between
// them and vector instructions that write their operands or read their result.
template <bool B_AGPR, bool AFTER_INIT>
__device__ __forceinline__ void mfma_f16_asm(f32x16 &acc, const bf16x8_t &a, const bf16x8_t &b) {
#if __HIP_DEVICE_COMPILE__
  if constexpr (B_AGPR) {
    if constexpr (AFTER_INIT) asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
  } else {
    if constexpr (AFTER_INIT) asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  }
#endif
}
// The same with the accumulator in the ordinary registers and the b operand in the accumulation
// half: vector instructions then read the results where they lie.  (A v_accvgpr_read placed
// between matrix instructions was measured at 15-20 cycles: 32 of them per tile cost a fifth of
// the row-stationary affine + p-norm kernel's time.)  FIRST: the chain starts from zero.
template <bool FIRST, class AFrag>
__device__ __forceinline__ void mfma_f16_asm_vacc(f32x16 &acc, const AFrag &a, const bf16x8_t &b) {
#if __HIP_DEVICE_COMPILE__
  if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));  // (the result may not share registers with a or b)
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
#endif
}
// Covers the distance a vector instruction must keep behind a matrix instruction whose result it reads.
__device__ __forceinline__ void mfma_result_gap() {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_nop 15\n\ts_nop 15");
#endif
}
// The workgroup barrier alone: unlike __syncthreads() it does not wait for the wave's
// outstanding global_load ... lds (the caller waits for exactly the ones that matter).
__device__ __forceinline__ void barrier_only() {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_barrier" ::: "memory");
#endif
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f) {  // f(integral_constant<int, I>) for I = 0 .. N-1
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<N, I + 1>(f);
  }
}
#ifdef AMLIKE_OUT_STAMPS  // diagnostic build: clock ticks of the two sweeps of job 0, printed per launch
__device__ unsigned long long g_out_stamps[8];
__device__ unsigned long long g_out_stalls[16];  // per wave: sweep 1 wait, barrier; sweep 2 wait, barrier
#endif
struct OutJobs {
  int nbig;    // jobs of 256 rows
  int small;   // rows per remaining job (multiple of 32, <= 128 for FT = 1); their rows follow the big jobs'
  int lead;    // small jobs that go first (blocks 0 .. lead-1), before the big ones
};

template <int KS, int PREC>
__global__ void __launch_bounds__(256, 1)
    nnet_out_softmax_kernel(const uint4 *__restrict__ Xp, size_t xplane, const uint4 *__restrict__ Wp,
                            size_t wplane, const float *__restrict__ bias_pad,
                            const float *__restrict__ logprior_pad, float *__restrict__ Y, int ldy, int M,
                            int N, int post, float scale, OutJobs jobs) {
  constexpr int NPL = PREC ? 2 : 3, NPROD = PREC ? 3 : 6;
  constexpr int kTileU4 = NPL * 2 * KS * 32;  // uint4 per staged W tile: [plane][octet][32 units]
  constexpr int kOutRing = OutRingTiles(KS, PREC), kLead = kOutRing - 1;  // tile seq + kLead is requested in step seq
  constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_o[];
  uint4 *s_w = reinterpret_cast<uint4 *>(smem_o);
  float *s_bp = reinterpret_cast<float *>(s_w + kOutRing * kTileU4);
  float *s_stage = s_bp + kOutSideRing * 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, h = lane >> 5;
  int row0, nrows;
  const int b = blockIdx.x;
  if (b >= jobs.lead && b < jobs.lead + jobs.nbig) {
    row0 = (b - jobs.lead) * 256;
    nrows = 256;
  } else {
    row0 = jobs.nbig * 256 + (b < jobs.lead ? b : b - jobs.nbig) * jobs.small;
    nrows = min(jobs.small, M - row0);
  }
  if (nrows <= 0) return;
  const int ntile = (N + 31) >> 5;
  const int total = 2 * ntile;  // W tiles to stream: both sweeps
  const float *prior_src = logprior_pad ? logprior_pad : bias_pad;

  // W tile `seq % ntile` travels global -> LDS ring slot seq % kOutRing without passing through
  // registers (global_load ... lds), in pieces of 1 KiB (64 lanes x 16 bytes: two octets of one
  // plane for 32 units) dealt round-robin to the 4 waves; its biases and log priors go to
  // slot seq % kOutSideRing of the side ring.  Every wave issues exactly kMine + 1 of these
  // loads per tile (a wave short of a piece repeats the last one), so "all but the newest
  // kMine + 1 vector-memory instructions have finished" means: every tile requested before the
  // newest has landed.
  constexpr int kPieces = NPL * KS, kMine = (kPieces + 3) / 4;
  // (the blocks take the pieces of a tile in different orders: they all want the same tile at
  // about the same time, and should not all want the same piece of it)
  const int rot = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 7u) % (4u * kMine)));
  auto request = [&](int seq) __attribute__((always_inline)) {
    const int n0 = (seq % ntile) << 5;
    uint4 *dst = s_w + (seq % kOutRing) * kTileU4;
#pragma unroll
    for (int i = 0; i < kMine; ++i) {
      const int piece = min((w + 4 * i + rot) % (4 * kMine), kPieces - 1);
      const int q = piece / KS, op = piece - q * KS;  // plane, octet pair
      const uint4 *src = Wp + (size_t)q * wplane + (size_t)(2 * op + h) * N + min(n0 + li, N - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(dst + piece * 64), 16, 0, 0);
    }
    const float *bsrc = (h ? prior_src : bias_pad) + n0 + li;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)bsrc,
                                     (__attribute__((address_space(3))) void *)(s_bp + (seq % kOutSideRing) * 64),
                                     4, 0, 0);
  };

  auto run = [&](auto ft_tag, auto logm_tag) __attribute__((always_inline)) {
    constexpr int FT = decltype(ft_tag)::value;
    constexpr bool LOGM = decltype(logm_tag)::value;  // log-probabilities (floored) or posteriors
    constexpr int NPF = NPROD * FT;    // matrix instructions per k-step
    constexpr int NMF = KS * NPF;      // per W tile
    constexpr int NE = 16 * FT;        // values of a finished tile per lane
    const int wrow0 = row0 + w * 32 * FT;  // this wave's first row
    // X fragments, resident: [frame tile][k-step][plane]
    bf16x8_t xf[FT][KS][NPL];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const size_t r = (size_t)min(wrow0 + 32 * f + li, M - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int q = 0; q < NPL; ++q)
          xf[f][ks][q] = __builtin_bit_cast(bf16x8_t, Xp[(size_t)q * xplane + (size_t)(2 * ks + h) * M + r]);
    }
    // running softmax statistics per row, in log2 units: maximum and sum of 2^(z log2e - max)
    float mx[FT], sm[FT], logz[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      mx[f] = -INFINITY;
      sm[f] = 0.0f;
      logz[f] = 0.0f;
    }
    // of the lane's 16 units: the biases of the next tile, the log priors of the tile finished last
    f32x4 bnext[4], lpold[4];
    const float pscale = (post & kPostPrior) ? scale : 1.0f, nscale = (post & kPostPrior) ? -scale : 0.0f;
    static_for<kLead>([&](auto t) __attribute__((always_inline)) { request(decltype(t)::value); });

    // Sweep 1, the vector work on one finished tile, cut into units that go between the matrix
    // instructions of the next tile.  Units 0 .. FT-1: a frame tile's maximum and the
    // rescaling of its sum.  Units FT .. FT+NE+1: a three-stage pipeline over the NE values
    // (scaled difference, 2^x, accumulate), one value entering per unit.
    constexpr int kStatUnits = FT + NE + 2;
    float st_t[NE], st_e[NE];
    auto stats_unit = [&](const f32x16 (&acc)[FT], auto u_tag) __attribute__((always_inline)) {
      constexpr int u = decltype(u_tag)::value;
      if constexpr (u < FT) {
        constexpr int f = u;
        float m = fmaxf(acc[f][0], acc[f][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) m = fmaxf(m, fmaxf(acc[f][r], acc[f][r + 1]));
        const float mn = fmaxf(mx[f], m * kLog2e);
        sm[f] *= __builtin_amdgcn_exp2f(mx[f] - mn);  // 2^(-inf) = 0 on the first tile, where sm is 0 too
        mx[f] = mn;
        pin_order(sm[f]);
        pin_order(mx[f]);
      } else {
        constexpr int i = u - FT;  // value i enters; i - 1 is exponentiated; i - 2 is added
        if constexpr (i < NE) {
          st_t[i] = fmaf(acc[i % FT][i / FT], kLog2e, -mx[i % FT]);
          pin_order(st_t[i]);
        }
        if constexpr (i >= 1 && i - 1 < NE) {
          st_e[i - 1] = __builtin_amdgcn_exp2f(st_t[i - 1]);
          pin_order(st_e[i - 1]);
        }
        if constexpr (i >= 2) {
          sm[(i - 2) % FT] += st_e[i - 2];
          pin_order(sm[(i - 2) % FT]);
        }
      }
    };
    // Sweep 2, the vector work on one finished tile.  The matrix instruction leaves a lane with 4
    // consecutive values of ONE row per group (part, f): stored from there, an instruction would
    // touch 32 rows with 32 bytes each.  So the finished values first go to the wave's staging
    // tile in LDS (16 "group" operations of four stages each per frame tile: log softmax; floor or
    // posterior; minus log prior, times scale; 16-byte LDS write) and come back row-wise, 8
    // lanes to a row of 128 bytes, in 4 "chunks" of 8 rows per frame tile (a read operation and,
    // a little later, a store operation).  Order: the groups of frame tile 0; the groups of frame
    // tile 1 interleaved with the chunks of tile 0; the chunks of the last tile.
    constexpr int kStoreOps = 16 * FT + 8 * FT;
    f32x4 sv, rd[2];
    const unsigned stage_w = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)s_stage +
                             (unsigned)w * (64 * kOutStagePitch * 4);
    const unsigned st_waddr = stage_w + (li * kOutStagePitch + 4 * h) * 4;                  // group writes
    const unsigned st_raddr = stage_w + ((lane >> 3) * kOutStagePitch + 4 * (lane & 7)) * 4;  // chunk reads
    // the lane's 16 bytes of a chunk in Y, relative to the job's first row and the tile's first column
    const unsigned st_yoff = ((unsigned)(w * 32 * FT + (lane >> 3)) * (unsigned)ldy + 4u * (lane & 7)) * 4u;
    char *const yjob = reinterpret_cast<char *>(Y + (size_t)row0 * ldy);
    auto group_op = [&](const f32x16 (&acc)[FT], auto f_tag, auto k_tag) __attribute__((always_inline)) {
      constexpr int f = decltype(f_tag)::value, k = decltype(k_tag)::value, part = k / 4, stage = k % 4;
      if constexpr (stage == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = acc[f][4 * part + e] - logz[f];
      } else if constexpr (stage == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (LOGM) sv[e] = fmaxf(sv[e], -46.051701859880914f);  // log(1e-20)
          else sv[e] = floor_1e20(__expf(sv[e]));
        }
      } else if constexpr (stage == 2) {
        sv = sv * pscale + lpold[part];  // lpold: -log prior x scale (or 0)
      } else {
        lds_hidden_write_at<((32 * f) * kOutStagePitch + 8 * part) * 4>(st_waddr, sv);
      }
      if constexpr (stage < 3) {
#if __HIP_DEVICE_COMPILE__
        asm volatile("" : "+v"(sv));
#endif
      }
    };
    auto chunk_op = [&](int n0, auto f_tag, auto j_tag) __attribute__((always_inline)) {
      constexpr int f = decltype(f_tag)::value, j = decltype(j_tag)::value;
      constexpr int is_store[8] = {0, 0, 1, 0, 1, 0, 1, 1}, chunk_of[8] = {0, 1, 0, 2, 1, 3, 2, 3};
      constexpr int c = chunk_of[j], row = 32 * f + 8 * c;
      if constexpr (!is_store[j]) {
        lds_hidden_read_at<row * kOutStagePitch * 4>(rd[c & 1], st_raddr);
      } else {
        lds_hidden_wait1(rd[c & 1]);
#ifndef AMLIKE_OUT_NOSTORE  // (timing experiment: nothing is written)
        if (w * 32 * FT + row + (lane >> 3) < nrows && n0 + 4 * (lane & 7) < N) {
          const float4 v = make_float4(rd[c & 1][0], rd[c & 1][1], rd[c & 1][2], rd[c & 1][3]);
          *reinterpret_cast<float4 *>(yjob + ((size_t)row * ldy + n0) * 4 + st_yoff) = v;
        }
#endif
      }
    };
    auto store_op = [&](const f32x16 (&acc)[FT], int n0, auto i_tag) __attribute__((always_inline)) {
      constexpr int i = decltype(i_tag)::value;
      using F0 = std::integral_constant<int, 0>;
      using FL = std::integral_constant<int, FT - 1>;
      if constexpr (i < 16) {
        group_op(acc, F0(), std::integral_constant<int, i>());
      } else if constexpr (FT == 2 && i < 40) {
        constexpr int i2 = i - 16, blk = i2 / 3, pos = i2 % 3;
        if constexpr (pos < 2) group_op(acc, FL(), std::integral_constant<int, 2 * blk + pos>());
        else chunk_op(n0, F0(), std::integral_constant<int, blk>());
      } else {
        chunk_op(n0, FL(), std::integral_constant<int, i - (FT == 2 ? 40 : 16)>());
      }
    };

    // Fragments of W for the matrix instructions, read from LDS kAhead k-steps before their use;
    // the stream of k-steps runs on across tile boundaries (tile seq + 1 sits in LDS while tile
    // seq is multiplied), so the read latency is exposed once per job, not once per tile.
    constexpr int kAhead = 2;
    static_assert(KS % 3 == 1, "the register slots below assume KS = 3 n + 1");
    // k-step j of any tile uses register slot j % 3, the last k-step slot 3: any kAhead + 1
    // consecutive k-steps of the stream, tile boundaries included, sit in distinct slots
    bf16x8_t wf[4][NPL];
    auto load_frag = [&](int seq, auto ks_tag) __attribute__((always_inline)) {  // k-step ks of tile seq
      constexpr int ks = decltype(ks_tag)::value, slot = ks == KS - 1 ? 3 : ks % 3;
      const uint4 *wt = s_w + (seq % kOutRing) * kTileU4;
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        wf[slot][q] = __builtin_bit_cast(bf16x8_t, wt[(q * 2 * KS + 2 * ks + h) * 32 + li]);
    };
    // requests the biases of tile seq + 1 and the log priors of tile seq (hidden reads)
    auto load_side = [&](int seq) __attribute__((always_inline)) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        lds_hidden_read(bnext[q4], s_bp + ((seq + 1) % kOutSideRing) * 64 + 8 * q4 + 4 * h);
        lds_hidden_read(lpold[q4], s_bp + (seq % kOutSideRing) * 64 + 32 + 8 * q4 + 4 * h);
      }
    };
    // One step of the stream: the matrix instructions of W tile seq into acc_new (initialised
    // with the bias), with the vector work of the previous tile (acc_old; kind 0: none, 1: sweep-1
    // statistics, 2: sweep-2 stores of tile old_n0) placed between them unit by unit.  On entry
    // tiles seq and seq + 1 are in LDS, tile seq + 2 is on its way, and the fragments of k-steps
    // 0 .. kAhead-1 of tile seq and its biases have been requested from LDS.
    // The k-step in which tile seq + kLead is requested: the first, so that the step's stores are
    // all younger than the request and the wait at the end of the step (all but the newest
    // loads and stores) does not wait for any of them.
    constexpr int kRequestKs = 0;
#ifdef AMLIKE_OUT_STAMPS
    unsigned long long stall[4] = {0, 0, 0, 0};  // ticks in the end-of-step wait and barrier, per sweep
#endif
    auto step = [&](int seq, f32x16 (&acc_new)[FT], const f32x16 (&acc_old)[FT], auto kind_tag,
                    int old_n0, bool stores_counted = false) __attribute__((always_inline)) {
      constexpr int kKind = decltype(kind_tag)::value;
      lds_hidden_wait(bnext, lpold);
      if constexpr (kKind == 2) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) lpold[q4] = lpold[q4] * nscale;  // -log prior x scale (0 without priors)
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          acc_new[f][4 * q4 + 0] = bnext[q4].x;
          acc_new[f][4 * q4 + 1] = bnext[q4].y;
          acc_new[f][4 * q4 + 2] = bnext[q4].z;
          acc_new[f][4 * q4 + 3] = bnext[q4].w;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<KS>([&](auto ks_tag) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_tag)::value;
        // the fragments kAhead k-steps on: of this tile, or of the next one (already in LDS)
        if constexpr (ks + kAhead < KS) {
          load_frag(seq, std::integral_constant<int, (ks + kAhead) % KS>());
        } else {
          if (seq + 1 < total) load_frag(seq + 1, std::integral_constant<int, (ks + kAhead) % KS>());
          if (ks == KS - 1) load_side(seq);  // (past the last tile the biases read are not used)
        }
#ifndef AMLIKE_OUT_NOFETCH  // (timing experiment: the W stream stops after its first tiles)
        // (past the end of the stream the request wraps around: harmless, and it keeps the count
        // of outstanding loads the same in every step)
        if constexpr (ks == kRequestKs) request(seq + kLead);
#endif
        constexpr int slot = ks == KS - 1 ? 3 : ks % 3;
        static_for<NPF>([&](auto j_tag) __attribute__((always_inline)) {
          constexpr int j = decltype(j_tag)::value, which = j / FT, f = j % FT, mi = ks * NPF + j;
          int pa, pb;
          product_parts<PREC>(which, pa, pb);
          if constexpr (PREC == 1) {
            // frame tile 0's fragments live in the accumulation registers, tile 1's in the ordinary ones
            if constexpr (f == 0) mfma_f16_asm<true, ks == 0 && j < FT>(acc_new[f], wf[slot][pa], xf[f][ks][pb]);
            else mfma_f16_asm<false, ks == 0 && j < FT>(acc_new[f], wf[slot][pa], xf[f][ks][pb]);
          } else {
            acc_new[f] = mfma16<PREC>(wf[slot][pa], xf[f][ks][pb], acc_new[f]);
          }
#ifndef AMLIKE_OUT_NOVEC  // (timing experiment: matrix instructions only)
          if constexpr (kKind == 1) {
            // the units spread evenly over the tile's matrix instructions
            constexpr int u0 = (mi * kStatUnits + NMF - 1) / NMF, u1 = ((mi + 1) * kStatUnits + NMF - 1) / NMF;
            static_for<u1 - u0>([&](auto d) __attribute__((always_inline)) {
              stats_unit(acc_old, std::integral_constant<int, u0 + decltype(d)::value>());
            });
          } else if constexpr (kKind == 2) {
            // over all k-steps but the last, which requests the next tile's log priors into lpold
            constexpr int span = (KS - 1) * NPF;
            if constexpr (mi < span) {
              constexpr int u0 = (mi * kStoreOps + span - 1) / span, u1 = ((mi + 1) * kStoreOps + span - 1) / span;
              static_for<u1 - u0>([&](auto d) __attribute__((always_inline)) {
                store_op(acc_old, old_n0, std::integral_constant<int, u0 + decltype(d)::value>());
              });
            }
          }
#endif
          // nothing moves across a matrix instruction and its share of the vector work
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      // tile seq + 2 (requested kLead - 1.5 steps ago) has landed; every wave has finished
      // reading tile seq
      if constexpr (PREC == 1) mfma_result_gap();
#ifdef AMLIKE_OUT_STAMPS
      const unsigned long long w0 = __builtin_readcyclecounter();
#endif
      // (stores_counted: this step and the kLead - 2 before it each issued all their 4 FT stores,
      // which are younger than the request waited for and need not have finished)
      if (kKind == 2 && stores_counted) wait_vector_memory_but<(kLead - 2) * (kMine + 1) + (kLead - 1) * 4 * FT>();
      else wait_vector_memory_but<(kLead - 2) * (kMine + 1)>();
#ifdef AMLIKE_OUT_STAMPS
      const unsigned long long w1 = __builtin_readcyclecounter();
#endif
      barrier_only();
#ifdef AMLIKE_OUT_STAMPS
      const unsigned long long w2 = __builtin_readcyclecounter();
      stall[seq >= ntile ? 2 : 0] += w1 - w0;
      stall[seq >= ntile ? 3 : 1] += w2 - w1;
#endif
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    f32x16 acc_a[FT], acc_b[FT];
    wait_vector_memory_but<(kLead - 2) * (kMine + 1)>();
    barrier_only();  // tiles 0 and 1 are in LDS
    load_side(-1);
    static_for<kAhead>([&](auto j) __attribute__((always_inline)) { load_frag(0, j); });
#ifdef AMLIKE_OUT_STAMPS
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // sweep 1
    step(0, acc_a, acc_b, K0(), 0);
    bool in_a = true;  // which set holds the tile finished last
    for (int t = 1; t < ntile; t += 2) {
      step(t, acc_b, acc_a, K1(), 0);
      in_a = false;
      if (t + 1 < ntile) {
        step(t + 1, acc_a, acc_b, K1(), 0);
        in_a = true;
      }
    }
    static_for<kStatUnits>([&](auto u) __attribute__((always_inline)) {
      if (in_a) stats_unit(acc_a, u);
      else stats_unit(acc_b, u);
    });
#pragma unroll
    for (int f = 0; f < FT; ++f) {  // the two half-waves hold different units of the same rows
      const float om = __shfl_xor(mx[f], 32), os = __shfl_xor(sm[f], 32);
      const float mn = fmaxf(mx[f], om);
      const float tot = sm[f] * __builtin_amdgcn_exp2f(mx[f] - mn) + os * __builtin_amdgcn_exp2f(om - mn);
      logz[f] = (mn + __log2f(tot)) * kLn2;
    }
#ifdef AMLIKE_OUT_STAMPS
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
#endif
    // sweep 2
    step(ntile, acc_a, acc_b, K0(), 0);
    in_a = true;
    // does every store instruction of this wave run (all its rows exist)?  The tiles stored
    // inside the loop are whole (only the last tile of a sweep can be narrower).
    const bool all_rows = w * 32 * FT + 32 * FT <= nrows;
    for (int t = 1; t < ntile; t += 2) {
      step(ntile + t, acc_b, acc_a, K2(), (t - 1) << 5, all_rows && t >= kLead);
      in_a = false;
      if (t + 1 < ntile) {
        step(ntile + t + 1, acc_a, acc_b, K2(), t << 5, all_rows && t + 1 >= kLead);
        in_a = true;
      }
    }
    lds_hidden_wait(bnext, lpold);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) lpold[q4] = lpold[q4] * nscale;
    static_for<kStoreOps>([&](auto u) __attribute__((always_inline)) {
      if (in_a) store_op(acc_a, (ntile - 1) << 5, u);
      else store_op(acc_b, (ntile - 1) << 5, u);
    });
    wait_vector_memory_but<0>();  // no load may still be on its way into LDS when the block ends
#ifdef AMLIKE_OUT_STAMPS
    if ((int)blockIdx.x == jobs.lead && threadIdx.x == 0) {  // the first 256-row job
      g_out_stamps[0] = c1 - c0;
      g_out_stamps[1] = __builtin_readcyclecounter() - c1;
      g_out_stamps[2] = r1 - r0;
      g_out_stamps[3] = __builtin_amdgcn_s_memrealtime() - r1;
      g_out_stamps[4] = ntile;
    }
    if ((int)blockIdx.x == jobs.lead && lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g_out_stalls[w * 4 + i] = stall[i];
    }
#endif
  };
  if (post & kPostFloorLog) {
    if (nrows <= 128) run(std::integral_constant<int, 1>(), std::true_type());
    else run(std::integral_constant<int, 2>(), std::true_type());
  } else {
    if (nrows <= 128) run(std::integral_constant<int, 1>(), std::false_type());
    else run(std::integral_constant<int, 2>(), std::false_type());
  }
}

// Affine + p-norm (p = 2, groups of 10) with the ROWS stationary: the hidden layers of the p-norm
// nets behind a NormalizeComponent (fp16x2 operands, k <= 207 plus the bias column).  Same
// skeleton as nnet_out_softmax_kernel -- a wave keeps the operand fragments of its FT x 32 rows
// of X in registers for the whole job, the W tiles of 32 units stream global -> LDS ring ->
// fragments, one wave per SIMD, the vector work on tile t - 1 placed between the matrix
// instructions of tile t -- with one sweep over the units and this vector work: every value is
// squared and added to the sum of its p-norm group.  The host orders the rows of W
// (BuildPnormRowsWeights) so that, of the 16 values a lane holds of any tile, value r belongs to
// group r % 8 of the 8 groups its half-wave owns in the current run of 5 tiles ("super-tile": 160
// units = 16 groups): the group sums are 8 registers per frame tile whatever the tile, and after
// every fifth tile they are complete: square root, 32 bytes per lane to Y, start over.
// Against nnet_gemm_pnorm_kernel (a block per 160 units x 128 rows, X and W both staged per
// k-step behind a barrier): X is read once per job instead of once per unit block, there is one
// barrier per 78 matrix instructions instead of one per 15, and no wave waits for its own X.
// X: planes [2][Kp/8][M] as the other kernels read them, the bias column's 1.0 included.
// W: [2][Kp/8][Npad] in the row order above, Npad a multiple of 160.  Y: [M x ldy], G groups.
#ifndef AMLIKE_ROWS_ABL
#define AMLIKE_ROWS_ABL 0
#endif
#ifndef AMLIKE_ROWS_ONECHAIN  // 1: all three partial products in one accumulator chain (measured 1.6 % faster than two)
#define AMLIKE_ROWS_ONECHAIN 1
#endif
#ifdef AMLIKE_ROWS_STAMPS  // diagnostic build: clock ticks per part of a step, waves 0 and 4 of block 0
__device__ unsigned long long g_rows_stamps[16];
#endif
struct RowJobs {
  int nbig;   // jobs of 256 rows (blocks 0 .. nbig-1)
  int small;  // rows per remaining job (a multiple of 32, at most 256); their rows follow the big jobs'
};

__host__ __device__ constexpr int PnormRowsRing(int KS) {
  const int tile = 2 * 2 * KS * 32 * 16;
  return 160 * 1024 / tile >= 6 ? 5 : 160 * 1024 / tile >= 5 ? 4 : 160 * 1024 / tile;
}

// NW = 8: two waves per SIMD, 256 rows per job (k of 13 steps: the X fragments of a wave take 104
// registers).  NW = 4: one wave per SIMD with the whole register file, 128 rows per job, for wider
// inputs (k of 23 steps, 184 registers of fragments: the first hidden layer behind the splice);
// the wave then runs two accumulator chains, there being no partner to fill the gap a dependent
// matrix instruction leaves, and the W ring holds three tiles.
// flag / gmode: as nnet_guard_skips (layers whose input range is decided on the device).
template <int KS, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 2 : 1)
    nnet_pnorm_rows_kernel(const uint4 *__restrict__ Xp, size_t xplane, const uint4 *__restrict__ Wp,
                           size_t wplane, int Npad, float *__restrict__ Y, int ldy, int M, int G,
                           RowJobs jobs, Planes nout, int one_col, const int *flag, int gmode) {
  constexpr int NPL = 2, NPROD = 3;
  constexpr int kJobRows = NW * 32;
  constexpr bool kOneChain = AMLIKE_ROWS_ONECHAIN && NW == 8;
  constexpr int kTileU4 = NPL * 2 * KS * 32;  // uint4 per staged W tile: [plane][octet][32 units]
  constexpr int kRing = PnormRowsRing(KS), kLead = kRing - 1;  // tile seq + kLead is requested in step seq
  static_assert(kRing >= 3, "the W ring needs three tiles");
  if (nnet_guard_skips(flag, gmode)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
  uint4 *s_w = reinterpret_cast<uint4 *>(smem_r);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, h = lane >> 5;
  int row0, nrows;
  const int b = blockIdx.x;
  if (b < jobs.nbig) {
    row0 = b * kJobRows;
    nrows = kJobRows;
  } else {
    row0 = jobs.nbig * kJobRows + (b - jobs.nbig) * jobs.small;
    nrows = min(jobs.small, M - row0);
  }
  if (nrows <= 0) return;
  const int ntile = Npad >> 5;  // a multiple of 5

  // W tile seq travels global -> LDS ring slot seq % kRing in pieces of 1 KiB (64 lanes x 16 bytes:
  // two octets of one plane for 32 units), dealt round-robin to the 8 waves.  Every wave issues
  // exactly kMine loads per tile (a wave short of a piece repeats the last one), so "all but the
  // newest n * kMine vector-memory instructions have finished" means: every tile but the n
  // requested last has landed.  (The stores of finished groups count too, and they are older
  // than some of those loads: the wait is then for a little more than it names, never less.)
  // A piece's source is Wp + 512 bytes x tile + a per-lane offset that does not change from tile
  // to tile, kept in a register per piece; its place in a ring slot is wave-uniform.
  constexpr int kPieces = NPL * KS, kMine = (kPieces + NW - 1) / NW;
  const int rot = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 7u) % (unsigned)(NW * kMine)));
  unsigned piece_off[kMine];  // bytes from Wp + 512 tile
  int piece_at[kMine];        // uint4 index inside a ring slot
#pragma unroll
  for (int i = 0; i < kMine; ++i) {
    const int piece = min((w + NW * i + rot) % (NW * kMine), kPieces - 1);
    const int q = piece / KS, op = piece - q * KS;  // plane, octet pair
    piece_off[i] = (unsigned)(((size_t)q * wplane + (size_t)(2 * op + h) * Npad + li) * 16);
    piece_at[i] = __builtin_amdgcn_readfirstlane(piece * 64);
  }
  auto request_piece = [&](int seq, auto i_tag) __attribute__((always_inline)) {
    constexpr int i = decltype(i_tag)::value;
    const char *src0 = reinterpret_cast<const char *>(Wp) + (size_t)(seq % ntile) * 512;
    uint4 *dst = s_w + (seq % kRing) * kTileU4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + piece_off[i]),
                                     (__attribute__((address_space(3))) void *)(dst + piece_at[i]), 16, 0, 0);
  };
  auto request = [&](int seq) __attribute__((always_inline)) {
    static_for<kMine>([&](auto i_tag) __attribute__((always_inline)) { request_piece(seq, i_tag); });
  };

  constexpr int NMF = KS * NPROD;  // matrix instructions per W tile
  constexpr int NE = 16;           // values of a finished tile per lane
  constexpr int kSkip = 3;         // matrix instructions of a step issued before the previous tile's values are read
  const int rl = w * 32 + li;      // this lane's row within the job
  bf16x8_t xf[KS][NPL];
  {
    const size_t r = (size_t)min(row0 + rl, M - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        xf[ks][q] = __builtin_bit_cast(bf16x8_t, Xp[(size_t)q * xplane + (size_t)(2 * ks + h) * M + r]);
  }
  float ss[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) ss[g] = 0.0f;
  static_for<kLead>([&](auto t) __attribute__((always_inline)) { request(decltype(t)::value); });

  // A tile's values are the sums of two accumulators: x takes the products of the operands' high
  // parts, y the two cross products (small terms among themselves).  Two chains, because a matrix
  // instruction that continues the chain of the one just before it issues about 9 cycles late
  // (41 instead of 32 cycles per instruction measured for a wave alone on its SIMD).
  struct TileAcc {
    f32x16 x, y;
  };
  // the vector work on value r of a finished tile
  auto square_unit = [&](const TileAcc &acc, auto r_tag) __attribute__((always_inline)) {
    constexpr int r = decltype(r_tag)::value;
    float v = acc.x[r];
    if constexpr (!kOneChain) v += acc.y[r];
    ss[r & 7] = fmaf(v, v, ss[r & 7]);
    pin_order(ss[r & 7]);
  };
  // the groups of super-tile `sup` are complete: this lane's 8 (half h) of its row
  float rowsq = 0.0f;  // the sum of the squares of this lane's p-norm outputs of its row (groups < G)
  auto finish_groups = [&](int sup) __attribute__((always_inline)) {
    const int g0 = 16 * sup + 8 * h;
    if (g0 + 8 <= G) {
      rowsq += ((ss[0] + ss[1]) + (ss[2] + ss[3])) + ((ss[4] + ss[5]) + (ss[6] + ss[7]));
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g)
        if (g0 + g < G) rowsq += ss[g];
    }
    float4 lo = make_float4(sqrtf(ss[0]), sqrtf(ss[1]), sqrtf(ss[2]), sqrtf(ss[3]));
    float4 hi = make_float4(sqrtf(ss[4]), sqrtf(ss[5]), sqrtf(ss[6]), sqrtf(ss[7]));
    if (nout.p) {
      // The rows are normalised by the epilogue: until then Y is this kernel's scratch, laid out so
      // that a half-wave's 32 lanes write (and later read back) 512 consecutive bytes: 16-byte
      // element (2 octet + half) * M + row.  (Groups past G come from all-zero rows of W: zeros.)
      float4 *yt = reinterpret_cast<float4 *>(Y) + (size_t)(2 * (2 * sup + h)) * M + (row0 + rl);
      if (rl < nrows) {
        yt[0] = lo;
        yt[M] = hi;
      }
    } else if (rl < nrows) {
      float *o = Y + (size_t)(row0 + rl) * ldy + g0;
      if (g0 + 8 <= G) {
        *reinterpret_cast<float4 *>(o) = lo;
        *reinterpret_cast<float4 *>(o + 4) = hi;
      } else {
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int g = 0; g < 8; ++g)
          if (g0 + g < G) o[g] = v[g];
      }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) ss[g] = 0.0f;
  };

  constexpr int kAhead = 2;
  // Register slots of the W fragments: k-step ks uses slot ks % 3, except that the last KS % 3
  // k-steps of a tile use slots 3 and 4 -- while they run, the first k-steps of the next tile
  // are already being read into slots 0 and 1.
  constexpr int kTailSteps = KS % 3, kSlots = 3 + kTailSteps;
  auto slot_of = [](int ks) constexpr { return ks >= KS - kTailSteps ? 3 + (ks - (KS - kTailSteps)) : ks % 3; };
  // Fragments of W, read from LDS kAhead k-steps before their use, as reads the compiler does not
  // see, with counted waits: left to the compiler, every third k-step waited for ALL reads in
  // flight, the ones just issued included.  Past the end of the stream the reads go on (to a
  // ring slot that holds some older tile; the values are not used): the count of reads in
  // flight is the same at every k-step.
  f32x4 wf[kSlots][NPL];
#pragma unroll
  for (int sl = 0; sl < kSlots; ++sl)
#pragma unroll
    for (int q = 0; q < NPL; ++q) wf[sl][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const unsigned s_w_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint4 *)s_w +
                            (unsigned)(h * 32 + li) * 16u;
  auto load_frag = [&](int seq, auto ks_tag) __attribute__((always_inline)) {  // k-step ks of tile seq
    constexpr int ks = decltype(ks_tag)::value, slot = slot_of(ks);
    const unsigned at = s_w_addr + (unsigned)(seq % kRing) * (unsigned)(kTileU4 * 16);
    static_for<NPL>([&](auto q_tag) __attribute__((always_inline)) {
      constexpr int q = decltype(q_tag)::value;
      lds_hidden_reload_at<(q * 2 * KS + 2 * ks) * 32 * 16>(wf[slot][q], at);
    });
  };
  // One step: the matrix instructions of W tile seq into acc_new (from zero: the bias is a k
  // column), with the squares of the previous tile's values (acc_old) between them.  On entry
  // tiles seq and seq + 1 are in LDS, and the fragments of k-steps 0 .. kAhead-1 of tile seq
  // have been requested from LDS.
#ifdef AMLIKE_ROWS_STAMPS
  unsigned long long st_mfma = 0, st_wait = 0, st_bar = 0, st_steps = 0, st_last = __builtin_readcyclecounter();
  const unsigned long long st_begin = st_last, st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
  // A wave whose 32 rows lie beyond the job's rows (short jobs of the last round) only does its
  // share of the W stream and keeps the barriers: the SIMD it sits on is then its partner's alone.
  const bool active = w * 32 < nrows;
  auto step = [&](int seq, TileAcc &acc_new, const TileAcc &acc_old, auto work_tag) __attribute__((always_inline)) {
    constexpr bool kWork = decltype(work_tag)::value;
    if (!active) {
#if !(AMLIKE_ROWS_ABL & 2)
      request(seq + kLead);
#endif
      wait_vector_memory_but<(kLead - 2) * kMine>();
      barrier_only();
      return;
    }
    static_for<KS>([&](auto ks_tag) __attribute__((always_inline)) {
      constexpr int ks = decltype(ks_tag)::value;
#if !(AMLIKE_ROWS_ABL & 4)  // (timing experiment: the fragments of W are whatever the registers hold)
      if constexpr (ks + kAhead < KS) load_frag(seq, std::integral_constant<int, (ks + kAhead) % KS>());
      else load_frag(seq + 1, std::integral_constant<int, (ks + kAhead) % KS>());
#endif
      // (past the end of the stream the request wraps around: harmless, and it keeps the count
      // of outstanding loads the same in every step)
#if !(AMLIKE_ROWS_ABL & 2)  // (timing experiment: the W stream stops after its first tiles)
      // one piece every kSpread k-steps rather than all at the head of the step, where the eight
      // waves, fresh from the barrier, would queue 32 pieces at once
      constexpr int kSpread = KS / kMine;
      if constexpr (ks % kSpread == 0 && ks / kSpread < kMine)
        request_piece(seq + kLead, std::integral_constant<int, ks / kSpread>());
#endif
      constexpr int slot = slot_of(ks);
#if !(AMLIKE_ROWS_ABL & 4)
      lds_hidden_wait_but<kAhead * NPL>(wf[slot][0], wf[slot][1]);  // all but the reads of the k-steps ahead
#endif
      static_for<NPROD>([&](auto j_tag) __attribute__((always_inline)) {
        // order within a k-step: cross product, high parts, cross product (y, x, y)
        constexpr int j = decltype(j_tag)::value, mi = ks * NPROD + j;
        // the fragments of X live in the accumulation registers, the accumulators in the ordinary ones
        if constexpr (kOneChain) {  // all three products in one chain
          if constexpr (j == 0) mfma_f16_asm_vacc<ks == 0>(acc_new.x, wf[slot][1], xf[ks][0]);
          else if constexpr (j == 1) mfma_f16_asm_vacc<false>(acc_new.x, wf[slot][0], xf[ks][1]);
          else mfma_f16_asm_vacc<false>(acc_new.x, wf[slot][0], xf[ks][0]);
        } else {  // x: the high parts' products, y: the two cross products
          if constexpr (j == 0) mfma_f16_asm_vacc<ks == 0>(acc_new.y, wf[slot][1], xf[ks][0]);
          else if constexpr (j == 1) mfma_f16_asm_vacc<ks == 0>(acc_new.x, wf[slot][0], xf[ks][0]);
          else mfma_f16_asm_vacc<false>(acc_new.y, wf[slot][0], xf[ks][1]);
        }
        // The previous tile's values are read from the fourth matrix instruction of the step on: by
        // then the last matrix instruction of the previous step has long delivered its result.
        if constexpr (kWork && !(AMLIKE_ROWS_ABL & 1) && mi >= kSkip) {  // (bit 0, timing experiment: no vector work)
          constexpr int u0 = ((mi - kSkip) * NE + NMF - kSkip - 1) / (NMF - kSkip),
                        u1 = ((mi - kSkip + 1) * NE + NMF - kSkip - 1) / (NMF - kSkip);
          static_for<u1 - u0>([&](auto d) __attribute__((always_inline)) {
            square_unit(acc_old, std::integral_constant<int, u0 + decltype(d)::value>());
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
#ifdef AMLIKE_ROWS_STAMPS
    const unsigned long long t1 = __builtin_readcyclecounter();
#endif
    // Every LDS read of this step has delivered before the step ends: the compiler takes a
    // register for written as soon as the read that fills it has been issued, and may copy it
    // when the loop turns round.
    lds_hidden_wait_but<0>(wf[0][0], wf[0][1]);
    lds_hidden_wait_but<0>(wf[1][0], wf[1][1]);
    // tile seq + 2 has landed; every wave has finished reading tile seq
    wait_vector_memory_but<(kLead - 2) * kMine>();
#ifdef AMLIKE_ROWS_STAMPS
    const unsigned long long t2 = __builtin_readcyclecounter();
#endif
    barrier_only();
#ifdef AMLIKE_ROWS_STAMPS
    const unsigned long long t3 = __builtin_readcyclecounter();
    st_mfma += t1 - st_last;
    st_wait += t2 - t1;
    st_bar += t3 - t2;
    st_last = t3;
    ++st_steps;
#endif
  };
  TileAcc acc_a, acc_b;
  wait_vector_memory_but<(kLead - 2) * kMine>();
  barrier_only();  // tiles 0 and 1 are in LDS
#if AMLIKE_ROWS_ABL & 4
  static_for<4>([&](auto sl) __attribute__((always_inline)) {
    wf[decltype(sl)::value][0] = (f32x4){1.0f, 0.5f, 0.25f, 2.0f};
    wf[decltype(sl)::value][1] = (f32x4){1.0f, 0.5f, 0.25f, 2.0f};
  });
#else
  if (active) static_for<kAhead>([&](auto j) __attribute__((always_inline)) { load_frag(0, j); });
#endif
  step(0, acc_a, acc_b, std::false_type());
  // the squares of tile t - 1 ride in step t; tile t - 1 closes a super-tile when t % 5 == 0
  for (int t = 1; t < ntile; t += 2) {
    step(t, acc_b, acc_a, std::true_type());
    if (t % 5 == 0) finish_groups(t / 5 - 1);
    if (t + 1 < ntile) {
      step(t + 1, acc_a, acc_b, std::true_type());
      if ((t + 1) % 5 == 0) finish_groups((t + 1) / 5 - 1);
    }
  }
  // the last tile's values (ntile odd: set a was written last)
  if (!active) {
    wait_vector_memory_but<0>();
    return;
  }
  mfma_result_gap();
  static_for<NE>([&](auto i) __attribute__((always_inline)) {
    if (ntile & 1) square_unit(acc_a, i);
    else square_unit(acc_b, i);
  });
  finish_groups(ntile / 5 - 1);
  wait_vector_memory_but<0>();  // no load may still be on its way into LDS when the block ends
  if (nout.p) {
    // The NormalizeComponent behind the p-norm, and the next affine stage's fp16x2 operand planes
    // (nnet2/nnet-component.cc:578-592; the layout nnet_normalize_planes_kernel writes): the wave
    // reads back what its lanes wrote (it is in L2; every lane its own values, in the scratch
    // layout above), scales each row by 1 / sqrt(max(sum of squares / G, 2^-66)) and writes plane
    // q, octet o, row r as 16 bytes; half-wave h takes the octets o = h (mod 2).  (The stores above
    // have finished: the wait counted them.)
    const float tot = rowsq + __shfl_xor(rowsq, 32);
    const float f = 1.0f / sqrtf(fmaxf(tot / G, 1.3552527156068805e-20f));
    if (rl < nrows) {
      const float4 *yt = reinterpret_cast<const float4 *>(Y) + (row0 + rl);
      uint4 *o = reinterpret_cast<uint4 *>(nout.p);
      const size_t pstride = nout.stride >> 3;
      const int kp8 = nout.Kp >> 3, ny8 = Npad / 80;  // octets of the planes; octets this kernel wrote
      // four octets at a time: their eight loads are in flight together (larger batches measured
      // the same: the epilogue costs what its 2 x 182 MB of traffic cost, in bursts at the job ends)
      constexpr int kBatch = 4;
      for (int oct0 = h; oct0 < kp8; oct0 += 2 * kBatch) {
        float4 a[kBatch], bq[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int oct = oct0 + 2 * u;
          if (oct < ny8) {
            a[u] = yt[(size_t)(2 * oct) * M];
            bq[u] = yt[(size_t)(2 * oct + 1) * M];
          } else {
            a[u] = bq[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int oct = oct0 + 2 * u, c0 = oct * 8;
          if (oct >= kp8) break;
          float t[8] = {a[u].x * f, a[u].y * f, a[u].z * f, a[u].w * f, bq[u].x * f, bq[u].y * f, bq[u].z * f, bq[u].w * f};
          if (one_col >= c0 && one_col < c0 + 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (c0 + j == one_col) t[j] = 1.0f;
          }
          uint4 s0, s1;
          split_f16x2_8(make_float4(t[0], t[1], t[2], t[3]), make_float4(t[4], t[5], t[6], t[7]), s0, s1);
          const size_t i = (size_t)oct * M + (row0 + rl);
          o[i] = s0;
          o[pstride + i] = s1;
        }
      }
    }
  }
#ifdef AMLIKE_ROWS_STAMPS
  if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == 4)) {
    unsigned long long *o = g_rows_stamps + (w ? 8 : 0);
    o[0] = st_mfma;
    o[1] = st_wait;
    o[2] = st_bar;
    o[3] = st_steps;
    o[4] = __builtin_readcyclecounter() - st_begin;
    o[5] = __builtin_amdgcn_s_memrealtime() - st_real0;
  }
#endif
}

// Output layer and softmax with the skeleton of nnet_pnorm_rows_kernel: 8 waves of 32 rows each,
// two to a SIMD, X fragments resident, W tiles of 32 units streamed through the LDS ring, the
// accumulators in the ordinary registers, counted LDS waits, the bias as a k column (the
// units that pad the last tile carry a bias of -60000: no share in any row's sum).  Two sweeps
// over the units as in nnet_out_softmax_kernel; what differs is how the finished values leave:
//   sweep 1  W fragment as the A operand, X fragment as B: a lane holds 16 units of ITS row, the
//            running (max, sum of exponentials) never cross lanes;
//   sweep 2  the same registers the other way round (X as A, W as B): a lane holds ONE unit for 16
//            rows, so a 4-byte store per accumulator register writes, over the half-wave, the 32
//            consecutive units of a row (128 bytes) -- no staging through LDS, no 16 log priors
//            per lane (one), and the LDS the staging tiles took holds a fifth W tile.
// Rows' log Z pass from the first layout to the second through 32 floats of LDS per wave.
__host__ __device__ constexpr int OutRowsRing(int KS) {
  const int tile = 2 * 2 * KS * 32 * 16, room = 160 * 1024 - 8 * 256 - 8 * 128;
  return room / tile >= 5 ? 5 : room / tile;
}
// `acc += a . b` with the a operand in the accumulation half of the register file and b, like
// the accumulator, in the ordinary one (mfma_f16_asm_vacc with the operands' places exchanged).
template <bool FIRST, class BFrag>
__device__ __forceinline__ void mfma_f16_asm_vacc_swapped(f32x16 &acc, const bf16x8_t &a, const BFrag &b) {
#if __HIP_DEVICE_COMPILE__
  if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
#endif
}
// Per lane: `b` where the lane's bit of the wave-uniform mask m is set, else `a`.
template <class T>
__device__ __forceinline__ T select_lanes(T a, T b, unsigned long long m) {
  T d = a;
#if __HIP_DEVICE_COMPILE__
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m));
#endif
  return d;
}
// Stores `val` at base + off: base wave-uniform (scalar registers), off a byte offset per lane.
__device__ __forceinline__ void store_f32_at(const char *base, unsigned off, float val) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(val), "s"(base));
#endif
}
__device__ __forceinline__ void lds_hidden_read32(float &v, unsigned addr) {  // valid after lds_hidden_wait32
#if __HIP_DEVICE_COMPILE__
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
#endif
}
__device__ __forceinline__ void lds_hidden_wait32(float &v) {
#if __HIP_DEVICE_COMPILE__
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
#endif
}
#ifndef AMLIKE_OUTROWS_ABL
#define AMLIKE_OUTROWS_ABL 0
#endif
#ifdef AMLIKE_OUTROWS_STAMPS  // diagnostic build: clock ticks of the two sweeps of the first 256-row job
__device__ unsigned long long g_outrows_stamps[8];
#endif

template <int KS, bool LOGM>
__global__ void __launch_bounds__(512, 2)
    nnet_out_rows_kernel(const uint4 *__restrict__ Xp, size_t xplane, const uint4 *__restrict__ Wp,
                         size_t wplane, int Npad, const float *__restrict__ logprior_pad,
                         float *__restrict__ Y, int ldy, int M, int N, int post, float scale, OutJobs jobs) {
  constexpr int NPL = 2, NPROD = 3, NW = 8;
  constexpr int kTileU4 = NPL * 2 * KS * 32;  // uint4 per staged W tile: [plane][octet][32 units]
  constexpr int kRing = OutRowsRing(KS), kLead = kRing - 1;  // tile seq + kLead is requested in step seq
  constexpr int kSideRing = 8;  // the 32 log priors of a tile (twice, one copy per half-wave), kept longer than the tile
  static_assert(kRing >= 4, "the W ring needs four tiles");
  constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
  uint4 *s_w = reinterpret_cast<uint4 *>(smem_q);
  float *s_lp = reinterpret_cast<float *>(s_w + kRing * kTileU4);
  float *s_lz = s_lp + kSideRing * 64;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, h = lane >> 5;
  int row0, nrows;
  const int b = blockIdx.x;
  if (b >= jobs.lead && b < jobs.lead + jobs.nbig) {
    row0 = (b - jobs.lead) * 256;
    nrows = 256;
  } else {
    row0 = jobs.nbig * 256 + (b < jobs.lead ? b : b - jobs.nbig) * jobs.small;
    nrows = min(jobs.small, M - row0);
  }
  if (nrows <= 0) return;
  const int ntile = Npad >> 5;
  const int total = 2 * ntile;  // W tiles to stream: both sweeps
  const bool has_prior = (post & kPostPrior) != 0;
  // (without priors the side ring is filled from the start of W: the values are read and dropped)
  const float *lp_src = has_prior ? logprior_pad : reinterpret_cast<const float *>(Wp);

  // The W stream, as in nnet_pnorm_rows_kernel: pieces of 1 KiB dealt round-robin to the 8 waves,
  // kMine per wave and tile, plus the tile's log priors (256 bytes, every wave the same ones):
  // kMine + 1 vector-memory loads per wave and tile.
  constexpr int kPieces = NPL * KS, kMine = (kPieces + NW - 1) / NW;
  const int rot = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 7u) % (unsigned)(NW * kMine)));
  unsigned piece_off[kMine];  // bytes from Wp + 512 tile
  int piece_at[kMine];        // uint4 index inside a ring slot
#pragma unroll
  for (int i = 0; i < kMine; ++i) {
    const int piece = min((w + NW * i + rot) % (NW * kMine), kPieces - 1);
    const int q = piece / KS, op = piece - q * KS;  // plane, octet pair
    piece_off[i] = (unsigned)(((size_t)q * wplane + (size_t)(2 * op + h) * Npad + li) * 16);
    piece_at[i] = __builtin_amdgcn_readfirstlane(piece * 64);
  }
  // load i of a tile's kMine + 1: W piece i, or (i == kMine) the tile's log priors
  auto request_piece = [&](int seq, auto i_tag) __attribute__((always_inline)) {
    constexpr int i = decltype(i_tag)::value;
    const int tile = seq % ntile;
    if constexpr (i < kMine) {
      const char *src0 = reinterpret_cast<const char *>(Wp) + (size_t)tile * 512;
      uint4 *dst = s_w + (seq % kRing) * kTileU4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + piece_off[i]),
                                       (__attribute__((address_space(3))) void *)(dst + piece_at[i]), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(lp_src + (tile << 5) + li),
                                       (__attribute__((address_space(3))) void *)(s_lp + (seq % kSideRing) * 64), 4, 0, 0);
    }
  };
  auto request = [&](int seq) __attribute__((always_inline)) {
    static_for<kMine + 1>([&](auto i_tag) __attribute__((always_inline)) { request_piece(seq, i_tag); });
  };

  constexpr int NMF = KS * NPROD;  // matrix instructions per W tile
  constexpr int kSkip = 3;         // matrix instructions of a step issued before the previous tile's values are read
  // A wave's 32 rows.  The job that ends the matrix may hold a wave with fewer: that wave takes the
  // last 32 rows of the matrix instead, recomputing some of its neighbour's (the same operands
  // through the same instructions: the same bits, written twice).  Every active wave is then whole.
  const int wbase = min(row0 + w * 32, M - 32);
  bf16x8_t xf[KS][NPL];
  {
    const size_t r = (size_t)(wbase + li);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        xf[ks][q] = __builtin_bit_cast(bf16x8_t, Xp[(size_t)q * xplane + (size_t)(2 * ks + h) * M + r]);
  }
  static_for<kLead>([&](auto t) __attribute__((always_inline)) { request(decltype(t)::value); });

  // One accumulator chain per tile (the register budget of two waves per SIMD, 128 ordinary
  // registers, does not hold two): a matrix instruction that continues the chain of the one just
  // before it issues a few cycles late when its wave is alone, and in the other wave's shadow otherwise.
  typedef f32x16 TileAcc;
  // Sweep 1.  Running softmax statistics of this lane's row over its half of the units, in log2
  // units: maximum, and sum of 2^(z log2e - max).  The work on a finished tile, in 27 units that
  // go between the matrix instructions of the next: the tile's maximum (8); the rescaling of the
  // running sum; a three-stage pipeline over the values (scaled difference, 2^x, accumulate), one
  // value entering per unit.
  float mx = -INFINITY, sm = 0.0f, tm = 0.0f, st3[3];
  constexpr int kStatUnits = 8 + 1 + 18;
  auto stats_unit = [&](const TileAcc &acc, auto u_tag) __attribute__((always_inline)) {
    constexpr int u = decltype(u_tag)::value;
    if constexpr (u < 8) {
      if constexpr (u == 0) tm = fmaxf(acc[0], acc[1]);
      else tm = fmaxf(tm, fmaxf(acc[2 * u], acc[2 * u + 1]));
      pin_order(tm);
    } else if constexpr (u == 8) {
      const float mn = fmaxf(mx, tm * kLog2e);
      sm *= __builtin_amdgcn_exp2f(mx - mn);  // 2^(-inf) = 0 on the first tile, where sm is 0 too
      mx = mn;
      pin_order(sm);
      pin_order(mx);
    } else {
      constexpr int i = u - 9;  // value i enters; i - 1 is exponentiated; i - 2 is added
      if constexpr (i < 16) {
        st3[i % 3] = fmaf(acc[i], kLog2e, -mx);
        pin_order(st3[i % 3]);
      }
      if constexpr (i >= 1 && i - 1 < 16) {
        st3[(i - 1) % 3] = __builtin_amdgcn_exp2f(st3[(i - 1) % 3]);
        pin_order(st3[(i - 1) % 3]);
      }
      if constexpr (i >= 2) {
        sm += st3[(i - 2) % 3];
        pin_order(sm);
      }
    }
  };
  // Sweep 2.  Accumulator register r of this lane is unit li of the tile for row 8 (r / 4) + 4 h +
  // r % 4 of the wave's 32: log softmax with that row's log Z, floor (or posterior), minus log
  // prior times scale.  Rows are ldy floats apart and ldy need not be a multiple of 32: a tile's 32
  // units of a row then straddle two 128-byte lines, and a partly written line costs the memory
  // about as much as two whole ones (measured: 3 TB/s against 5.5 TB/s for the same stores on
  // whole lines).  So a store instruction writes whole lines: the lanes at or past the row's line
  // boundary `b` carry the PREVIOUS tile's value (kept in `prev`), the lanes before it this
  // tile's, 128 bytes further on -- together the line that begins in the previous tile.  With ldy a
  // multiple of 8 the boundary depends on r % 4 only: four lane masks in scalar registers.  The
  // row's first and last partial lines are written by themselves (first_emit, the epilogue).
  f32x4 lz4[4];          // log Z of the 16 rows whose values this lane holds
  float lp_cur = 0.0f;   // this lane's unit's log prior, of the tile finished last (a hidden read)
  float nlp = 0.0f;      // -scale x log prior of the tile whose values are being written (0 without priors)
  const float pscale = has_prior ? scale : 1.0f, nscale = -scale;
  char *const yw = reinterpret_cast<char *>(Y + (size_t)wbase * ldy);
  const unsigned ldy4 = (unsigned)ldy * 4u;
  const unsigned off_old = (unsigned)(4 * h) * ldy4 + (unsigned)li * 4u, off_new = off_old + 128u;
  unsigned long long keep_old[4];  // per r % 4: the lanes (of both half-waves) at or past the line boundary
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned bnd = (32u - (((unsigned)(wbase + e) * (unsigned)ldy) & 31u)) & 31u;
    const unsigned m32 = 0xffffffffu << bnd;
    keep_old[e] = (unsigned long long)m32 | ((unsigned long long)m32 << 32);
  }
  float prev[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 0.0f;
  auto final_value = [&](const TileAcc &acc, auto r_tag) __attribute__((always_inline)) -> float {
    constexpr int r = decltype(r_tag)::value;
    float val = acc[r] - lz4[r / 4][r % 4];
    if constexpr (LOGM) val = fmaxf(val, -46.051701859880914f);  // log(1e-20)
    else val = floor_1e20(__expf(val));
    return fmaf(val, pscale, nlp);
  };
  // kind 2: the tile at n0 (its values in acc) and the one before it (prev): one whole line per row.
  // kind 3: the sweep's first tile: the lanes before the boundary, the row's first (partial) line.
  auto emit_unit = [&](const TileAcc &acc, int n0, auto r_tag, auto kind_tag) __attribute__((always_inline)) {
    constexpr int r = decltype(r_tag)::value, fr = 8 * (r / 4) + r % 4, kKind = decltype(kind_tag)::value;
    const float cur = final_value(acc, r_tag);
    const unsigned long long m = keep_old[r % 4];
    // the row's part of the address is wave-uniform and stays in scalar registers
    const char *base = yw + (size_t)(n0 - 32) * 4 + (size_t)(fr * ldy4);
    if constexpr (kKind == 2) {
#if AMLIKE_OUTROWS_ABL & 1  // (timing experiment: the whole-line stores of the second sweep are not issued)
      float keep = select_lanes(cur, prev[r], m);
      pin_order(keep);
#else
      store_f32_at(base, select_lanes(off_new, off_old, m), select_lanes(cur, prev[r], m));
#endif
    } else {
      if (!((m >> lane) & 1ull)) store_f32_at(base, off_new, cur);
    }
    prev[r] = cur;
    pin_order(prev[r]);
  };

  constexpr int kAhead = 2;
  static_assert(KS % 3 == 1, "the register slots below assume KS = 3 n + 1");
  f32x4 wf[4][NPL];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl)
#pragma unroll
    for (int q = 0; q < NPL; ++q) wf[sl][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const unsigned s_w_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint4 *)s_w +
                            (unsigned)(h * 32 + li) * 16u;
  const unsigned s_lp_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)s_lp + (unsigned)lane * 4u;
  auto load_frag = [&](int seq, auto ks_tag) __attribute__((always_inline)) {  // k-step ks of tile seq
    constexpr int ks = decltype(ks_tag)::value, slot = ks == KS - 1 ? 3 : ks % 3;
    const unsigned at = s_w_addr + (unsigned)(seq % kRing) * (unsigned)(kTileU4 * 16);
    static_for<NPL>([&](auto q_tag) __attribute__((always_inline)) {
      constexpr int q = decltype(q_tag)::value;
      lds_hidden_reload_at<(q * 2 * KS + 2 * ks) * 32 * 16>(wf[slot][q], at);
    });
  };
  const bool active = w * 32 < nrows;
  // One step: the matrix instructions of W tile seq into acc_new (from zero), with the vector work
  // on the previous tile (acc_old; kind 0: none, 1: sweep-1 statistics, 2: sweep-2 stores of tile
  // old_n0) between them.  SWEEP2: operands the other way round.  stores_counted: this step and
  // the kLead - 2 before it each issue all their 16 stores; those of this step and the one before
  // are younger than the loads waited for at the end and need not have finished.
  auto step = [&](int seq, TileAcc &acc_new, const TileAcc &acc_old, auto sweep2_tag, auto kind_tag,
                  int old_n0, bool stores_counted) __attribute__((always_inline)) {
    constexpr bool kSweep2 = decltype(sweep2_tag)::value;
    constexpr int kKind = decltype(kind_tag)::value;
    if (!active) {
      request(seq + kLead);
      wait_vector_memory_but<(kLead - 2) * (kMine + 1)>();
      barrier_only();
      return;
    }
    static_for<KS>([&](auto ks_tag) __attribute__((always_inline)) {
      constexpr int ks = decltype(ks_tag)::value;
      if constexpr (ks + kAhead < KS) load_frag(seq, std::integral_constant<int, (ks + kAhead) % KS>());
      else load_frag(seq + 1, std::integral_constant<int, (ks + kAhead) % KS>());
      // (past the end of the stream the request wraps around: harmless, and it keeps the count
      // of outstanding loads the same in every step)
      // one load every kSpread k-steps rather than all at the head of the step, where the eight
      // waves, fresh from the barrier, would queue 40 at once (measured: 1.5 % of the kernel)
      constexpr int kSpread = (KS - 1) / kMine;
      static_assert(kSpread >= 1 && kSpread * kMine < KS, "the tile's loads are issued within its step");
      if constexpr (ks % kSpread == 0 && ks / kSpread <= kMine)
        request_piece(seq + kLead, std::integral_constant<int, ks / kSpread>());
      constexpr int slot = ks == KS - 1 ? 3 : ks % 3;
      lds_hidden_wait_but<kAhead * NPL>(wf[slot][0], wf[slot][1]);  // all but the reads of the k-steps ahead
      static_for<NPROD>([&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value, mi = ks * NPROD + j;
        // order within a k-step: the two cross products, then the high parts
        if constexpr (!kSweep2) {
          if constexpr (j == 0) mfma_f16_asm_vacc<ks == 0>(acc_new, wf[slot][1], xf[ks][0]);
          else if constexpr (j == 1) mfma_f16_asm_vacc<false>(acc_new, wf[slot][0], xf[ks][1]);
          else mfma_f16_asm_vacc<false>(acc_new, wf[slot][0], xf[ks][0]);
        } else {
          if constexpr (j == 0) mfma_f16_asm_vacc_swapped<ks == 0>(acc_new, xf[ks][0], wf[slot][1]);
          else if constexpr (j == 1) mfma_f16_asm_vacc_swapped<false>(acc_new, xf[ks][1], wf[slot][0]);
          else mfma_f16_asm_vacc_swapped<false>(acc_new, xf[ks][0], wf[slot][0]);
        }
        if constexpr (kKind != 0 && mi >= kSkip) {
          constexpr int NU = kKind == 1 ? kStatUnits : 16;
          constexpr int u0 = ((mi - kSkip) * NU + NMF - kSkip - 1) / (NMF - kSkip),
                        u1 = ((mi - kSkip + 1) * NU + NMF - kSkip - 1) / (NMF - kSkip);
          static_for<u1 - u0>([&](auto d) __attribute__((always_inline)) {
            if constexpr (kKind == 1) stats_unit(acc_old, std::integral_constant<int, u0 + decltype(d)::value>());
            else emit_unit(acc_old, old_n0, std::integral_constant<int, u0 + decltype(d)::value>(), kind_tag);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    // this tile's log prior for the lane's unit (its values are written in the next step)
    if constexpr (kSweep2) lds_hidden_read32(lp_cur, s_lp_addr + (unsigned)(seq % kSideRing) * 256u);
    // every LDS read of this step has delivered before the step ends
    lds_hidden_wait_but<0>(wf[0][0], wf[0][1]);
    lds_hidden_wait_but<0>(wf[1][0], wf[1][1]);
    if constexpr (kSweep2) {
      lds_hidden_wait32(lp_cur);
      nlp = has_prior ? lp_cur * nscale : 0.0f;
    }
    // tile seq + 2 has landed; every wave has finished reading tile seq.  (The tile's last load was
    // issued near the end of step seq - 2, among that step's stores: younger than it are, at least,
    // the loads and stores of the two steps since.)
    if (kKind == 2 && stores_counted) wait_vector_memory_but<(kLead - 2) * (kMine + 1) + (kLead - 2) * 16>();
    else wait_vector_memory_but<(kLead - 2) * (kMine + 1)>();
    barrier_only();
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  TileAcc acc_a, acc_b;
  wait_vector_memory_but<(kLead - 2) * (kMine + 1)>();
  barrier_only();  // tiles 0 and 1 are in LDS
  if (active) static_for<kAhead>([&](auto j) __attribute__((always_inline)) { load_frag(0, j); });
#ifdef AMLIKE_OUTROWS_STAMPS
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // sweep 1
  step(0, acc_a, acc_b, std::false_type(), K0(), 0, false);
  bool in_a = true;  // which set holds the tile finished last
  for (int t = 1; t < ntile; t += 2) {
    step(t, acc_b, acc_a, std::false_type(), K1(), 0, false);
    in_a = false;
    if (t + 1 < ntile) {
      step(t + 1, acc_a, acc_b, std::false_type(), K1(), 0, false);
      in_a = true;
    }
  }
  if (active) {
    mfma_result_gap();
    static_for<kStatUnits>([&](auto u) __attribute__((always_inline)) {
      if (in_a) stats_unit(acc_a, u);
      else stats_unit(acc_b, u);
    });
    // the two half-waves hold different units of the same rows
    const float om = __shfl_xor(mx, 32), os = __shfl_xor(sm, 32);
    const float mn = fmaxf(mx, om);
    const float tot = sm * __builtin_amdgcn_exp2f(mx - mn) + os * __builtin_amdgcn_exp2f(om - mn);
    s_lz[w * 32 + li] = (mn + __log2f(tot)) * kLn2;  // (both half-waves write the same value)
#pragma unroll
    for (int part = 0; part < 4; ++part) {
      const float4 z = *reinterpret_cast<const float4 *>(s_lz + w * 32 + 8 * part + 4 * h);
      lz4[part] = (f32x4){z.x, z.y, z.z, z.w};
    }
  }
#ifdef AMLIKE_OUTROWS_STAMPS
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
#endif
  // sweep 2 (ntile >= 2): tile 0; tile 1 with the first partial lines of tile 0; then whole lines.
  // (Stores are counted once this step and the kLead - 2 before it are whole-line steps: a partial
  // one may be skipped where no lane has anything to write.)
  step(ntile, acc_a, acc_b, std::true_type(), K0(), 0, false);
  step(ntile + 1, acc_b, acc_a, std::true_type(), K3(), 0, false);
  in_a = false;
  for (int t = 2; t < ntile; t += 2) {
    step(ntile + t, acc_a, acc_b, std::true_type(), K2(), (t - 1) << 5, t >= kLead);
    in_a = true;
    if (t + 1 < ntile) {
      step(ntile + t + 1, acc_b, acc_a, std::true_type(), K2(), t << 5, t + 1 >= kLead);
      in_a = false;
    }
  }
  if (active) {
    // the last tile: its line with the tile before it, then what lies past its boundary; only here
    // can a lane's unit lie beyond the N the matrix holds (the padding of the last tile)
    mfma_result_gap();
    const int n0 = (ntile - 1) << 5;
    static_for<16>([&](auto r_tag) __attribute__((always_inline)) {
      constexpr int r = decltype(r_tag)::value, fr = 8 * (r / 4) + r % 4;
      const float cur = in_a ? final_value(acc_a, r_tag) : final_value(acc_b, r_tag);
      const bool old_lane = (keep_old[r % 4] >> lane) & 1ull;
      char *row = yw + (size_t)(fr * ldy4) + (size_t)(4 * h) * ldy4;
      if (old_lane) {
        *reinterpret_cast<float *>(row + (size_t)(n0 - 32 + li) * 4) = prev[r];
        if (n0 + li < N) *reinterpret_cast<float *>(row + (size_t)(n0 + li) * 4) = cur;
      } else if (n0 + li < N) {
        *reinterpret_cast<float *>(row + (size_t)(n0 + li) * 4) = cur;
      }
    });
  }
  wait_vector_memory_but<0>();  // no load may still be on its way into LDS when the block ends
#ifdef AMLIKE_OUTROWS_STAMPS
  if ((int)blockIdx.x == jobs.lead && threadIdx.x == 0) {
    g_outrows_stamps[0] = c1 - c0;
    g_outrows_stamps[1] = __builtin_readcyclecounter() - c1;
    g_outrows_stamps[2] = r1 - r0;
    g_outrows_stamps[3] = __builtin_amdgcn_s_memrealtime() - r1;
    g_outrows_stamps[4] = ntile;
  }
#endif
}

// Softmax for rows wider than the register form holds: three passes over the row.
__global__ void __launch_bounds__(256)
    nnet_softmax_wide_kernel(const float *__restrict__ in, int ld_in, int dim, int rows, int post,
                             const float *__restrict__ log_priors, float scale,
                             float *__restrict__ out, int ld_out, int zero_pad) {
  __shared__ float s_red[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *x = in + (size_t)r * ld_in;
    float mx = -INFINITY;
    for (int c = t; c < dim; c += 256) mx = fmaxf(mx, x[c]);
    mx = wave_max(mx);
    if (lane == 0) s_red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int c = t; c < dim; c += 256) sum += __expf(x[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) s_red[w] = sum;
    __syncthreads();
    sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    const float inv = 1.0f / sum;
    float *o = out + (size_t)r * ld_out;
    for (int c = t; c < ld_out; c += 256) {
      if (c < dim)
        o[c] = post_process(floor_1e20(__expf(x[c] - mx) * inv), post, log_priors, c, scale);
      else if (zero_pad) o[c] = 0.0f;
    }
  }
}

// SumGroupComponent: out[r][j] = sum of in[r][begin_j .. begin_{j+1}), plus post-processing.
__global__ void __launch_bounds__(256)
    nnet_sumgroup_kernel(const float *__restrict__ in, int ld_in, const int32_t *__restrict__ begins,
                         int out_dim, int rows, int post, const float *__restrict__ log_priors,
                         float scale, float *__restrict__ out, int ld_out, int zero_pad) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *x = in + (size_t)r * ld_in;
    float *o = out + (size_t)r * ld_out;
    for (int j = threadIdx.x; j < ld_out; j += 256) {
      if (j < out_dim) {
        float s = 0.0f;
        for (int c = begins[j]; c < begins[j + 1]; ++c) s += x[c];
        o[j] = post_process(s, post, log_priors, j, scale);
      } else if (zero_pad) {
        o[j] = 0.0f;
      }
    }
  }
}

// The elementwise components: Sigmoid / Tanh / RectifiedLinear (matrix/kaldi-vector.cc:937-951,
// :899-913; nnet-component.cc:782-788), SoftHinge (matrix/kaldi-matrix.cc:2434-2448), Power
// (matrix/kaldi-vector.cc:496-525 with include_sign = false), Scale (nnet-component.cc:844-850),
// FixedScale (:3103-3109) and FixedBias (:3171-3177) with their vector in `vec`, or a pure
// post-processing pass (kind = 0).
__global__ void __launch_bounds__(256)
    nnet_elementwise_kernel(const float *__restrict__ in, int ld_in, int dim, int rows, int kind,
                            const float *__restrict__ vec, float p, int post,
                            const float *__restrict__ log_priors, float scale,
                            float *__restrict__ out, int ld_out, int zero_pad) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *x = in + (size_t)r * ld_in;
    float *o = out + (size_t)r * ld_out;
    for (int c = threadIdx.x; c < ld_out; c += 256) {
      if (c < dim) {
        float v = x[c];
        if (kind == AMLIKE_SIGMOID) {
          v = v > 0.0f ? 1.0f / (1.0f + __expf(-v)) : __expf(v) / (__expf(v) + 1.0f);
        } else if (kind == AMLIKE_TANH) {
          const float e = __expf(-fabsf(v));
          const float th = -1.0f + 2.0f / (1.0f + e * e);
          v = v > 0.0f ? th : -th;
        } else if (kind == AMLIKE_RELU) {
          v = fmaxf(v, 0.0f);
        } else if (kind == AMLIKE_SOFTHINGE) {
          v = v > 10.0f ? v : log1pf(expf(v));
        } else if (kind == AMLIKE_POWER) {
          const float a = fabsf(v);
          if (p == 2.0f) v = v * v;
          else if (p == 0.5f) v = sqrtf(a);
          else if (p == 1.0f) v = a;
          else if (p < 0.0f) v = v == 0.0f ? 0.0f : powf(a, p);
          else v = powf(a, p);
        } else if (kind == AMLIKE_SCALE) {
          v *= p;
        } else if (kind == AMLIKE_FIXED_SCALE) {
          v *= vec[c];
        } else if (kind == AMLIKE_FIXED_BIAS) {
          v += vec[c];
        }
        o[c] = post_process(v, post, log_priors, c, scale);
      } else if (zero_pad) {
        o[c] = 0.0f;
      }
    }
  }
}

// Maxout (group > 0: out[j] = the largest of -1e20 and in[j * group .. (j + 1) * group),
// nnet-component.cc:443-454) and Permute (group == 0: out[c] = in[src[c]], :2262-2270).
__global__ void __launch_bounds__(256)
    nnet_regroup_kernel(const float *__restrict__ in, int ld_in, const int32_t *__restrict__ src,
                        int group, int out_dim, int rows, int post,
                        const float *__restrict__ log_priors, float scale, float *__restrict__ out,
                        int ld_out, int zero_pad) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *x = in + (size_t)r * ld_in;
    float *o = out + (size_t)r * ld_out;
    for (int j = threadIdx.x; j < ld_out; j += 256) {
      if (j < out_dim) {
        float v;
        if (group > 0) {
          v = -1.0e20f;
          for (int c = j * group; c < (j + 1) * group; ++c) v = fmaxf(v, x[c]);
        } else {
          v = x[src[j]];
        }
        o[j] = post_process(v, post, log_priors, j, scale);
      } else if (zero_pad) {
        o[j] = 0.0f;
      }
    }
  }
}

// ---------------------------------------------------------------------------
struct Stage {
  int kind = 0;
  int in_dim = 0, out_dim = 0;
  int left = 0, right = 0, const_dim = 0;  // splice
  float p = 2.0f;
  int normalize_after = 0;  // pnorm fused with the Normalize that follows
  float *d_w = nullptr;     // [out_dim][Kp]
  uint4 *d_wp = nullptr;    // the same split into bf16 planes: [3][out_dim][Kp] bf16
  uint4 *d_wpo = nullptr;   // the planes in k-octet-major order: [3][Kp/8][out_dim] x 16 bytes
  // affine followed by a p = 2 p-norm whose group size divides 80: the fused kernel's image
  uint4 *d_wpn = nullptr;   // rows permuted and padded to Npad, bias as column in_dim: [3][Kpn/8][Npad]
  uint4 *d_wpr = nullptr;   // the same in the row order of nnet_pnorm_rows_kernel (fp16x2, groups of 10, Kpn = 208)
  uint4 *d_wob = nullptr;   // the net's last affine stage before its softmax: fp16x2 planes [2][Kp/8][Nob] with the
  int Nob = 0;              //   bias as column in_dim and the units padded to whole tiles of 32 (nnet_out_rows_kernel)
  int fuse_gs = 0, Kpn = 0, Npad = 0;
  // 1: operands as fp16x2 (input behind a NormalizeComponent, weights inside the fp16 range);
  // 2: fp16x2 on trial (weights fit, input range unknown): d_wpo_safe / d_wpn_safe hold the
  //    bf16x3 images the fallback kernels read
  int f16 = 0;
  bool normalised_input = false;  // the stage reads the output of a NormalizeComponent
  uint4 *d_wpo_safe = nullptr, *d_wpn_safe = nullptr;
  std::vector<float> h_w, h_b;  // host copies kept until the next component is known
  float *d_bias = nullptr;
  int Kp = 0;
  int32_t *d_begins = nullptr;  // sumgroup prefix [out_dim + 1]
};

}  // namespace amlike

using namespace amlike;

struct amlike_nnet {
  int device = 0;
  std::vector<Stage> stages;  // after fusion
  int input_dim = 0, output_dim = 0, left = 0, right = 0;
  int num_priors = 0;
  bool gemm_f32 = false;  // AMLIKE_NNET_GEMM=f32: the exact-fp32 MFMA GEMM (A/B check of the bf16x3 one)
  bool gemm_inline_split = false;  // AMLIKE_NNET_GEMM=inline: bf16x3 GEMM that splits X in its k-loop
  int pnorm_rows = 1;  // AMLIKE_NNET_PNORM_ROWS=0: never the row-stationary affine + p-norm kernel (A/B aid)
  int pnorm_xcd_order = 1;  // AMLIKE_NNET_PNORM_XCD=0: launch-order block mapping in the affine + p-norm kernel (A/B aid)
  // tuning / testing aids, read from the environment once, when the handle is made
  int out_cus = 0;            // AMLIKE_NNET_OUT_CUS=<n>: plan the output kernel's jobs for a chip of n CUs
  bool out_no_stagger = false;  // AMLIKE_NNET_OUT_NO_STAGGER: no short jobs in front
  bool no_fused_out = false;    // AMLIKE_NNET_NO_FUSED_OUT: output layer and softmax as separate kernels
  bool no_direct_planes = false;  // AMLIKE_NNET_NO_DIRECT_PLANES: no stage writes its consumer's operand planes
  int rows_norm = 1;  // AMLIKE_NNET_ROWS_NORM=0: the normalisation behind the row-stationary kernel stays a kernel of its own (A/B aid)
  int out_rows = 1;  // AMLIKE_NNET_OUT_ROWS=0: never the 8-wave output-layer + softmax kernel (A/B aid)
  int first_rows = 1;  // AMLIKE_NNET_FIRST_ROWS=0: the wide-input (k = 368) layer stays on nnet_gemm_pnorm_kernel (A/B aid)
  int collapse = 1;  // AMLIKE_NNET_COLLAPSE=0: an affine stage right behind another stays a stage of its own (A/B aid)
  float *d_log_priors = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // submit / wait: the copies out, behind the kernels
  AsyncSlot slots[kAsyncSlots];
  std::vector<UttMap> slot_utts[kAsyncSlots];  // the row maps a submission's upload reads
  int32_t next_ticket = 0;
  DeviceBuffer ws_a, ws_b, ws_in, ws_out, ws_utts, ws_planes, ws_planes_b;
  int *d_flags = nullptr;  // one range flag per stage (fp16x2 on trial)
  std::vector<UttMap> h_utts;
  PinnedUploadRing uploads;  // h_utts on its way to ws_utts
  std::mutex mu;
};

namespace amlike {

int LaunchPlaneGemmBf16x3(const void *xp, size_t xplane, const void *wp, size_t wplane, size_t woct,
                          int Kp, const float *bias, float *y, int ldy, int M, int N, hipStream_t st) {
  if (M <= 0 || N <= 0 || (Kp & 15)) {
    SetError("LaunchPlaneGemmBf16x3: bad shape");
    return AMLIKE_EINVAL;
  }
  dim3 grid((N + 127) / 128, (M + 127) / 128);
  const bool big_out = (size_t)M * N * sizeof(float) > kStreamingBytes;
  hipLaunchKernelGGL((big_out ? nnet_gemm_bias_planes_kernel<0, true> : nnet_gemm_bias_planes_kernel<0, false>), grid, dim3(256), 0, st,
                     static_cast<const uint4 *>(xp), xplane, static_cast<const uint4 *>(wp), wplane, woct,
                     Kp, bias, y, ldy, M, N, nullptr, 0);
  AMLIKE_HIP(hipGetLastError());
  return AMLIKE_OK;
}

static void FreeStage(Stage &s) {
  (void)hipFree(s.d_w);
  (void)hipFree(s.d_wp);
  (void)hipFree(s.d_wpo);
  (void)hipFree(s.d_wpn);
  (void)hipFree(s.d_wpr);
  (void)hipFree(s.d_wob);
  (void)hipFree(s.d_wpo_safe);
  (void)hipFree(s.d_wpn_safe);
  (void)hipFree(s.d_bias);
  (void)hipFree(s.d_begins);
}

static void FreeNnet(amlike_nnet *n) {
  for (Stage &s : n->stages) FreeStage(s);
  (void)hipFree(n->d_log_priors);
  (void)hipFree(n->d_flags);
  n->ws_a.Free();
  n->ws_b.Free();
  n->ws_in.Free();
  n->ws_out.Free();
  n->ws_utts.Free();
  n->ws_planes.Free();
  n->ws_planes_b.Free();
  for (AsyncSlot &s : n->slots) s.Release();
  if (n->copy_stream) (void)hipStreamDestroy(n->copy_stream);
  if (n->stream) (void)hipStreamDestroy(n->stream);
  delete n;
}

// Runs the pipeline over a batch of utterances laid out back to back in d_feats.
// utt_rows[u] = input frames of utterance u.  Output rows are laid out back to back too
// (out_rows[u] = rows - left - right when !pad_input).  Returns total output rows.
// The operand image of nnet_gemm_pnorm_kernel: rows of W permuted so that, per block of
// 160 units, unit 80 hh + s (hh = 0, 1; s = 0 .. 79) sits in MFMA tile s / 16 at row
// (r & 3) + 8 (r >> 2) + 4 hh with r = s % 16 -- the row that accumulator register r of
// half-wave hh holds; the bias as column K; rows padded to a multiple of 160, columns to a
// multiple of 16; split into bf16x3 planes in k-octet-major order.
// x as NPL 16-bit parts: bf16x3 (exact) or fp16x2 (22 bits).
static inline void SplitParts(float x, int f16, uint16_t parts[3]) {
  if (!f16) {
    SplitBf16x3(x, parts);
    return;
  }
  const _Float16 h1 = static_cast<_Float16>(x);
  const _Float16 h2 = static_cast<_Float16>(x - static_cast<float>(h1));
  memcpy(&parts[0], &h1, 2);
  memcpy(&parts[1], &h2, 2);
  parts[2] = 0;
}

// Plain operand image of W for nnet_gemm_bias_planes_kernel: [NPL][Kp/8][N] x 16 bytes.
static std::vector<uint16_t> BuildPlaneWeights(const std::vector<float> &w, int Kp, int N, int f16) {
  const int npl = f16 ? 2 : 3;
  const size_t kp8 = Kp / 8, plane = (size_t)N * Kp;
  std::vector<uint16_t> out(npl * plane, 0);
  for (size_t r = 0; r < (size_t)N; ++r)
    for (int k = 0; k < Kp; ++k) {
      uint16_t parts[3];
      SplitParts(w[r * Kp + k], f16, parts);
      for (int q = 0; q < npl; ++q) out[q * plane + (((size_t)(k / 8) * N + r) * 8) + (k % 8)] = parts[q];
    }
  (void)kp8;
  return out;
}

static std::vector<uint16_t> BuildPnormWeights(const std::vector<float> &w, int ldw,
                                               const std::vector<float> &bias, int N, int K,
                                               int f16, int *Kpn, int *Npad) {
  *Kpn = RoundUp(K + 1, 16);
  *Npad = RoundUp(N, 160);
  const int npl = f16 ? 2 : 3;
  const size_t kp8 = *Kpn / 8, np = *Npad;
  std::vector<uint16_t> out(npl * np * *Kpn, 0);
  for (int u = 0; u < N; ++u) {
    const int wt = u / 160, hh = (u % 160) / 80, sl = u % 80;
    const int ti = sl / 16, r = sl % 16, rho = (r & 3) + 8 * (r >> 2) + 4 * hh;
    const size_t row = (size_t)wt * 160 + ti * 32 + rho;
    for (int k = 0; k <= K; ++k) {
      uint16_t parts[3];
      SplitParts(k < K ? w[(size_t)u * ldw + k] : bias[u], f16, parts);
      for (int q = 0; q < npl; ++q)
        out[q * np * *Kpn + (((size_t)(k / 8) * np + row) * 8) + (k % 8)] = parts[q];
    }
  }
  (void)kp8;
  return out;
}

// The operand image of nnet_pnorm_rows_kernel (groups of 10): unit 10 g + j of group g sits, with
// blk = g / 16, hh = (g % 16) / 8, in tile 5 blk + j / 2 at the matrix row that accumulator
// register r = g % 8 + 8 (j % 2) of half-wave hh holds, (r & 3) + 8 (r >> 2) + 4 hh; the bias as
// column K; rows padded to a multiple of 160, columns to a multiple of 16; fp16x2 planes in
// k-octet-major order.
static std::vector<uint16_t> BuildPnormRowsWeights(const std::vector<float> &w, int ldw,
                                                   const std::vector<float> &bias, int N, int K,
                                                   int *Kpn, int *Npad) {
  *Kpn = RoundUp(K + 1, 16);
  *Npad = RoundUp(N, 160);
  const size_t np = *Npad;
  std::vector<uint16_t> out(2 * np * *Kpn, 0);
  for (int u = 0; u < N; ++u) {
    const int g = u / 10, j = u % 10;
    const int blk = g / 16, hh = (g % 16) / 8, r = g % 8 + 8 * (j % 2);
    const int rho = (r & 3) + 8 * (r >> 2) + 4 * hh;
    const size_t row = (size_t)blk * 160 + (j / 2) * 32 + rho;
    for (int k = 0; k <= K; ++k) {
      uint16_t parts[3];
      SplitParts(k < K ? w[(size_t)u * ldw + k] : bias[u], 1, parts);
      for (int q = 0; q < 2; ++q)
        out[q * np * *Kpn + (((size_t)(k / 8) * np + row) * 8) + (k % 8)] = parts[q];
    }
  }
  return out;
}

static int LaunchPnormRows(const Stage &s, const uint4 *xp, size_t xplane, float *y, int ldy, int rows, int G,
                           hipStream_t st, Planes nout = Planes{nullptr, 0, 0}, int one_col = -1,
                           const int *flag = nullptr, int gmode = 0) {
  // Jobs: as many full rounds of whole jobs (256 rows, or 128 for the wide-input form) as the CUs
  // take evenly, then the rest in equal shares (multiples of 32 rows) so that the last round is
  // short everywhere.
  const int KS = s.Kpn / 16;
  const int job_rows = KS == 13 ? 256 : 128;
  const int cus = std::max(1, NumComputeUnits());
  RowJobs jobs;
  jobs.nbig = (rows / job_rows / cus) * cus;
  const int rest = rows - jobs.nbig * job_rows;
  jobs.small = job_rows;
  int nsmall = 0;
  if (rest > 0) {
    jobs.small = std::min(job_rows, RoundUp((rest + cus - 1) / cus, 32));
    nsmall = (rest + jobs.small - 1) / jobs.small;
  }
  const int grid = jobs.nbig + nsmall;
  const size_t wplane = (size_t)s.Npad * (s.Kpn / 8);
  const size_t smem = (size_t)PnormRowsRing(KS) * 2 * 2 * KS * 32 * sizeof(uint4);
#define AMLIKE_ROWS_LAUNCH(KSV, NWV)                                                                       \
  do {                                                                                                     \
    AMLIKE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nnet_pnorm_rows_kernel<KSV, NWV>),       \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                \
    hipLaunchKernelGGL((nnet_pnorm_rows_kernel<KSV, NWV>), dim3(grid), dim3(64 * NWV), smem, st, xp,       \
                       xplane, s.d_wpr, wplane, s.Npad, y, ldy, rows, G, jobs, nout, one_col, flag, gmode); \
  } while (0)
  if (KS == 13) AMLIKE_ROWS_LAUNCH(13, 8);
  else if (KS == 23) AMLIKE_ROWS_LAUNCH(23, 4);
  else {
    SetError("nnet2 forward: no row-stationary p-norm kernel for k = %d", s.Kpn);
    return AMLIKE_EINVAL;
  }
#undef AMLIKE_ROWS_LAUNCH
  AMLIKE_HIP(hipGetLastError());
#ifdef AMLIKE_ROWS_STAMPS
  {
    AMLIKE_HIP(hipStreamSynchronize(st));
    unsigned long long hs[16];
    if (hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_rows_stamps), sizeof(hs)) == hipSuccess && hs[3])
      for (int k = 0; k < 2; ++k)
        fprintf(stderr, "rows kernel job 0 wave %d: per step %.0f ticks = matrix part %.0f + load wait %.0f + barrier %.0f; "
                "whole job %.0f ticks in %.2f us (%.2f GHz), %llu steps\n", 4 * k,
                (double)(hs[8 * k] + hs[8 * k + 1] + hs[8 * k + 2]) / hs[8 * k + 3], (double)hs[8 * k] / hs[8 * k + 3],
                (double)hs[8 * k + 1] / hs[8 * k + 3], (double)hs[8 * k + 2] / hs[8 * k + 3], (double)hs[8 * k + 4],
                hs[8 * k + 5] * 0.01, hs[8 * k + 4] / (hs[8 * k + 5] * 10.0), hs[8 * k + 3]);
  }
#endif
  return AMLIKE_OK;
}

// The operand image of nnet_out_rows_kernel: fp16x2 planes [2][Kp/8][Npad] in k-octet-major order,
// the bias as column K, the units padded to whole tiles of 32 by units of weight 0 and bias
// -60000 (exact in fp16: their exponential is 0 whatever the row's maximum).
static std::vector<uint16_t> BuildOutRowsWeights(const std::vector<float> &w, int ldw,
                                                 const std::vector<float> &bias, int N, int K, int Kp,
                                                 int *Npad) {
  *Npad = RoundUp(N, 32);
  const size_t np = *Npad;
  std::vector<uint16_t> out(2 * np * Kp, 0);
  for (size_t u = 0; u < np; ++u)
    for (int k = 0; k <= K; ++k) {
      const float val = u < (size_t)N ? (k < K ? w[u * ldw + k] : bias[u]) : (k < K ? 0.0f : -60000.0f);
      uint16_t parts[3];
      SplitParts(val, 1, parts);
      for (int q = 0; q < 2; ++q) out[q * np * Kp + (((size_t)(k / 8) * np + u) * 8) + (k % 8)] = parts[q];
    }
  return out;
}

static int LaunchOutRows(const amlike_nnet *n, const Stage &s, const uint4 *xp, size_t xplane, const float *log_priors, float *y,
                         int ldy, int rows, int post, float scale, hipStream_t st) {
  if ((post & kPostPrior) && !log_priors) {
    SetError("amlike_nnet_compute: log-likelihoods asked for, but the net has no priors");
    return AMLIKE_EINVAL;
  }
  // Jobs as for nnet_out_softmax_kernel: full rounds of 256-row jobs, the rest in equal shares of a
  // multiple of 32 rows; half of the CUs start with a short job (when the short jobs are short),
  // so that the two halves of the chip are in different sweeps from then on.
  int cus = std::max(1, NumComputeUnits());
  if (n->out_cus) cus = n->out_cus;  // testing aid: job shapes of a smaller chip
  OutJobs jobs;
  jobs.nbig = (rows / 256 / cus) * cus;
  const int rest = rows - jobs.nbig * 256;
  int nsmall = 0;
  jobs.small = 256;
  if (rest > 0) {
    jobs.small = std::min(256, RoundUp((rest + cus - 1) / cus, 32));
    nsmall = (rest + jobs.small - 1) / jobs.small;
  }
  jobs.lead = jobs.nbig > 0 && jobs.small <= 128 ? std::min(nsmall, cus / 2) : 0;
  if (n->out_no_stagger) jobs.lead = 0;  // tuning aid
  const int grid = jobs.nbig + nsmall;
  constexpr int KS = 13;
  const size_t wplane = (size_t)s.Nob * (s.Kp / 8);
  const size_t smem = (size_t)OutRowsRing(KS) * 2 * 2 * KS * 32 * sizeof(uint4) + 8 * 256 + 8 * 128;
#define AMLIKE_OUTROWS_LAUNCH(LOGM)                                                                       \
  do {                                                                                                    \
    AMLIKE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nnet_out_rows_kernel<KS, LOGM>),        \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));               \
    hipLaunchKernelGGL((nnet_out_rows_kernel<KS, LOGM>), dim3(grid), dim3(512), smem, st, xp, xplane,     \
                       s.d_wob, wplane, s.Nob, (post & kPostPrior) ? log_priors : nullptr, y, ldy, rows,  \
                       s.out_dim, post, scale, jobs);                                                     \
  } while (0)
  if (post & kPostFloorLog) AMLIKE_OUTROWS_LAUNCH(true);
  else AMLIKE_OUTROWS_LAUNCH(false);
#undef AMLIKE_OUTROWS_LAUNCH
  AMLIKE_HIP(hipGetLastError());
#ifdef AMLIKE_OUTROWS_STAMPS
  {
    AMLIKE_HIP(hipStreamSynchronize(st));
    unsigned long long hs[8];
    if (hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_outrows_stamps), sizeof(hs)) == hipSuccess && hs[4])
      fprintf(stderr, "out rows job 0: sweep 1 %.0f ticks/tile (%.3f us), sweep 2 %.0f ticks/tile (%.3f us), %llu tiles\n",
              (double)hs[0] / hs[4], hs[2] * 0.01 / hs[4], (double)hs[1] / hs[4], hs[3] * 0.01 / hs[4], hs[4]);
  }
#endif
  return AMLIKE_OK;
}

// Does the fused output-layer + softmax kernel take stage si (an AffineComponent) and the
// SoftmaxComponent behind it?  It covers the nets' usual end: affine, softmax, nothing after;
// operands on fixed fp16x2 or bf16x3 planes; k padded to 208 (inputs of 193..208 dimensions)
// or 304; rows of the result 16-byte aligned.
static bool FusedOutputApplies(const amlike_nnet *n, size_t si, const float *d_out, int ldo, int post) {
  (void)post;
  if (n->no_fused_out || n->gemm_f32 || n->gemm_inline_split) return false;
  if (si + 2 != n->stages.size()) return false;
  const Stage &s = n->stages[si], &sm = n->stages[si + 1];
  if (s.kind != AMLIKE_AFFINE || s.fuse_gs || sm.kind != AMLIKE_SOFTMAX || s.f16 == 2) return false;
  if (s.Kp != 208 && !(s.Kp == 304 && s.f16)) return false;  // (bf16x3 fragments at k = 304 do not fit the registers)
  if ((s.out_dim & 7) || (ldo & 3) || (reinterpret_cast<uintptr_t>(d_out) & 15)) return false;
  if (n->num_priors != 0 && n->num_priors != s.out_dim) return false;
  return true;
}

static int LaunchOutSoftmax(const amlike_nnet *n, const Stage &s, const uint4 *xp, size_t xplane, const float *log_priors,
                            float *y, int ldy, int rows, int post, float scale, hipStream_t st) {
  if ((post & kPostPrior) && !log_priors) {
    SetError("amlike_nnet_compute: log-likelihoods asked for, but the net has no priors");
    return AMLIKE_EINVAL;
  }
  // Jobs: as many full rounds of 256-row jobs as the CUs take evenly, then the rest in equal
  // shares (multiples of 32 rows) so that the last round is short everywhere.
  int cus = std::max(1, NumComputeUnits());
  if (n->out_cus) cus = n->out_cus;  // testing aid: job shapes of a smaller chip
  OutJobs jobs;
  jobs.nbig = (rows / 256 / cus) * cus;
  const int rest = rows - jobs.nbig * 256;
  int nsmall = 0;
  jobs.small = 256;
  if (rest > 0) {
    const int per = RoundUp((rest + cus - 1) / cus, 32);
    jobs.small = per > 128 ? 256 : per;
    nsmall = (rest + jobs.small - 1) / jobs.small;
  }
  // The second sweep is bound by the writes, the first writes nothing: half of the CUs start
  // with a short job, so that from then on one half of the chip is in its first sweep while
  // the other is in its second.
  jobs.lead = jobs.nbig > 0 && jobs.small <= 128 ? std::min(nsmall, cus / 2) : 0;
  if (n->out_no_stagger) jobs.lead = 0;  // tuning aid
  const int grid = jobs.nbig + nsmall;
  const size_t wplane = (size_t)s.out_dim * (s.Kp / 8);
  const int npl = s.f16 ? 2 : 3;
  const size_t smem = (size_t)OutRingTiles(s.Kp / 16, s.f16 ? 1 : 0) * npl * 2 * (s.Kp / 16) * 32 * sizeof(uint4) +
                      kOutSideRing * 64 * sizeof(float) + kOutStageBytes;
#define AMLIKE_OUT_LAUNCH(KS, PREC)                                                                     \
  do {                                                                                                  \
    AMLIKE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nnet_out_softmax_kernel<KS, PREC>),   \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));             \
    hipLaunchKernelGGL((nnet_out_softmax_kernel<KS, PREC>), dim3(grid), dim3(256), smem, st, xp, xplane, \
                       s.d_wpo, wplane, s.d_bias, log_priors, y, ldy, rows, s.out_dim, post, scale, jobs); \
  } while (0)
  if (s.Kp == 208 && s.f16) AMLIKE_OUT_LAUNCH(13, 1);
  else if (s.Kp == 208) AMLIKE_OUT_LAUNCH(13, 0);
  else AMLIKE_OUT_LAUNCH(19, 1);
#undef AMLIKE_OUT_LAUNCH
  AMLIKE_HIP(hipGetLastError());
#ifdef AMLIKE_OUT_STAMPS
  {
    AMLIKE_HIP(hipStreamSynchronize(st));
    unsigned long long hs[8];
    if (hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_out_stamps), sizeof(hs)) == hipSuccess && hs[4])
      fprintf(stderr, "out+softmax job 0: sweep 1 %.0f ticks/tile (%.3f us), sweep 2 %.0f ticks/tile (%.3f us), %llu tiles\n",
              (double)hs[0] / hs[4], hs[2] * 0.01 / hs[4], (double)hs[1] / hs[4], hs[3] * 0.01 / hs[4], hs[4]);
    unsigned long long sl[16];
    if (hipMemcpyFromSymbol(sl, HIP_SYMBOL(g_out_stalls), sizeof(sl)) == hipSuccess && hs[4])
      for (int w = 0; w < 4; ++w)
        fprintf(stderr, "out+softmax job 0 wave %d, ticks/tile at the end of a step: sweep 1 wait %.0f barrier %.0f, sweep 2 wait %.0f barrier %.0f\n",
                w, (double)sl[4 * w] / hs[4], (double)sl[4 * w + 1] / hs[4], (double)sl[4 * w + 2] / hs[4],
                (double)sl[4 * w + 3] / hs[4]);
  }
#endif
  return AMLIKE_OK;
}

// The dense [out_dim x dim] matrix a DctComponent amounts to (nnet2/nnet-component.cc:2744-2830):
// the DCT matrix of matrix/matrix-functions.cc:592-608 (its first `keep` rows) applied to each
// block of dct_dim columns, between the two column reorderings of DctComponent::Reorder when
// `reorder` is set.  The first of those writes only keep * (dim / dct_dim) positions of a zeroed
// row, so with keep < dct_dim some inputs are dropped and blocks see zeros: restated as it is.
static bool BuildDctMatrix(int dim, int dct_dim, int out_dim, bool reorder, std::vector<float> *w) {
  if (dim <= 0 || dct_dim <= 0 || dim % dct_dim != 0) return false;
  const int chunks = dim / dct_dim;
  if (out_dim <= 0 || out_dim % chunks != 0) return false;
  const int keep = out_dim / chunks;
  if (keep > dct_dim) return false;
  std::vector<float> dct((size_t)keep * dct_dim);
  {
    float normalizer = (float)std::sqrt(1.0 / (double)(float)dct_dim);
    for (int n = 0; n < dct_dim; ++n) dct[n] = normalizer;
    normalizer = (float)std::sqrt(2.0 / (double)(float)dct_dim);
    for (int k = 1; k < keep; ++k)
      for (int n = 0; n < dct_dim; ++n)
        dct[(size_t)k * dct_dim + n] = (float)((double)normalizer * std::cos(M_PI / dct_dim * (n + 0.5) * k));
  }
  // src_of[q]: the input column that position q of the (reordered) row holds, or -1 for a zero
  std::vector<int> src_of(dim), dst_of(out_dim);
  for (int q = 0; q < dim; ++q) src_of[q] = reorder ? -1 : q;
  for (int m = 0; m < out_dim; ++m) dst_of[m] = m;
  if (reorder) {
    for (int b = 0; b < keep; ++b)
      for (int j = 0; j < chunks; ++j) src_of[j * keep + b] = b * chunks + j;
    for (int b = 0; b < chunks; ++b)
      for (int j = 0; j < keep; ++j) dst_of[b * keep + j] = j * chunks + b;
  }
  w->assign((size_t)out_dim * dim, 0.0f);
  for (int ch = 0; ch < chunks; ++ch)
    for (int k = 0; k < keep; ++k)
      for (int n = 0; n < dct_dim; ++n) {
        const int src = src_of[ch * dct_dim + n];
        if (src >= 0) (*w)[(size_t)dst_of[ch * keep + k] * dim + src] = dct[(size_t)k * dct_dim + n];
      }
  return true;
}

static int Forward(amlike_nnet *n, const float *d_feats, int ld, const int32_t *utt_rows,
                   int num_utts, int pad_input, int mode, float prob_scale, float *d_out, int ldo,
                   hipStream_t st, int64_t *total_out_rows) {
  const int ctx_l = n->left, ctx_r = n->right;
  // Row bookkeeping.  Stage 0 input rows are the caller's rows; the first Splice stage
  // does the padding as part of its gather.  If the net has no Splice, nothing changes.
  std::vector<UttMap> &um = n->h_utts;
  um.resize(num_utts);
  int64_t in_total = 0;
  for (int u = 0; u < num_utts; ++u) {
    if (utt_rows[u] <= 0 || (!pad_input && utt_rows[u] <= ctx_l + ctx_r)) {
      SetError("amlike_nnet_compute: utterance %d with %d frames would give empty output", u,
               utt_rows[u]);
      return AMLIKE_EINVAL;
    }
    um[u].in_off = (int32_t)in_total;
    um[u].in_rows = utt_rows[u];
    in_total += utt_rows[u];
  }
  if (in_total > (int64_t)1 << 30) {
    SetError("amlike_nnet_compute: batch too large");
    return AMLIKE_EINVAL;
  }
  // Padding (when asked) is applied once, at the first Splice: that stage sees the
  // utterance extended by the TOTAL context of the net on each side.
  int pad_l = pad_input ? ctx_l : 0, pad_r = pad_input ? ctx_r : 0;
  int post = 0;
  if (mode == AMLIKE_NNET_LOG) post = kPostFloorLog;
  if (mode == AMLIKE_NNET_LOGLIKES) post = kPostFloorLog | kPostPrior;
  if ((post & kPostPrior) && n->num_priors != n->output_dim) {
    SetError("Priors in neural network not set up.");
    return AMLIKE_EDIM;
  }
  // Worst-case activation buffer: rows (with padding) x widest stage.
  int64_t max_rows = in_total + (int64_t)num_utts * (pad_l + pad_r);
  int max_ld = RoundUp(n->input_dim, 16);
  for (const Stage &s : n->stages) max_ld = std::max(max_ld, RoundUp(s.out_dim, 16));
  int rc = n->ws_a.Reserve((size_t)max_rows * max_ld * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  rc = n->ws_b.Reserve((size_t)max_rows * max_ld * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  rc = n->ws_utts.Reserve((size_t)num_utts * sizeof(UttMap));
  if (rc != AMLIKE_OK) return rc;

  // Per-utterance row maps of every gather stage, built up front and uploaded once.
  // A net that does not start with a Splice gets a plain copy stage in front, so that
  // every later stage sees an aligned, k-padded buffer.
  const size_t ns = n->stages.size();
  const bool copy_first = n->stages[0].kind != AMLIKE_SPLICE && n->stages[0].kind != AMLIKE_SPLICE_MAX;
  std::vector<UttMap> &maps = n->h_utts;
  maps.clear();
  std::vector<int64_t> gather_rows;  // output rows of each gather, in order
  {
    std::vector<int32_t> rows(utt_rows, utt_rows + num_utts);
    bool pad_done = !(pad_l || pad_r);
    auto add_gather = [&](int lp, int rp, int shrink) {
      int64_t ioff = 0, ooff = 0;
      for (int u = 0; u < num_utts; ++u) {
        UttMap m;
        m.in_off = (int32_t)ioff;
        m.in_rows = rows[u];
        m.out_off = (int32_t)ooff;
        m.out_rows = rows[u] + lp + rp - shrink;
        ioff += rows[u];
        ooff += m.out_rows;
        rows[u] = m.out_rows;
        maps.push_back(m);
      }
      gather_rows.push_back(ooff);
    };
    if (copy_first) add_gather(0, 0, 0);
    for (const Stage &s : n->stages)
      if (s.kind == AMLIKE_SPLICE || s.kind == AMLIKE_SPLICE_MAX) {
        add_gather(pad_done ? 0 : pad_l, pad_done ? 0 : pad_r, s.left + s.right);
        pad_done = true;
      }
    if (!pad_done) {
      SetError("amlike_nnet_compute: pad_input needs a SpliceComponent to carry the context");
      return AMLIKE_EINVAL;
    }
  }
  rc = n->ws_utts.Reserve(maps.size() * sizeof(UttMap) + 16);
  if (rc != AMLIKE_OK) return rc;
  if (!maps.empty()) {
    rc = n->uploads.Upload(n->ws_utts.ptr, maps.data(), maps.size() * sizeof(UttMap), st);
    if (rc != AMLIKE_OK) return rc;
  }
  const UttMap *d_maps = static_cast<const UttMap *>(n->ws_utts.ptr);
  if (int crc = ClearWords(n->d_flags, (int)n->stages.size() + 1, st)) return crc;

  const float *cur = d_feats;
  int cur_ld = ld;
  int64_t cur_rows = in_total;
  bool padded = !(pad_l || pad_r);
  float *bufs[2] = {static_cast<float *>(n->ws_a.ptr), static_cast<float *>(n->ws_b.ptr)};
  int which = 0;
  size_t gi = 0;  // next gather
  auto launch_gather = [&](int lp, int nsplice, int dim, int const_dim, float *dst, int dst_ld,
                           int out_dim) -> int {
    int max_out = 0;
    for (int u = 0; u < num_utts; ++u) max_out = std::max(max_out, maps[gi * num_utts + u].out_rows);
    dim3 grid(std::min(max_out, 2048), num_utts);
    hipLaunchKernelGGL(nnet_pad_splice_kernel, grid, dim3(256), 0, st, cur, cur_ld,
                       d_maps + gi * num_utts, num_utts, lp, nsplice, dim, const_dim, dst, dst_ld,
                       out_dim);
    AMLIKE_HIP(hipGetLastError());
    return AMLIKE_OK;
  };
  if (copy_first) {
    float *dst = bufs[which];
    const int dst_ld = RoundUp(n->input_dim, 16);
    rc = launch_gather(0, 1, n->input_dim, 0, dst, dst_ld, n->input_dim);
    if (rc != AMLIKE_OK) return rc;
    cur = dst;
    cur_ld = dst_ld;
    cur_rows = gather_rows[gi++];
    which ^= 1;
  }
  // The operand planes in use: a stage that writes its consumer's planes while other blocks of its
  // own kernel still read its input planes writes them into the other buffer, which then takes over.
  DeviceBuffer *wsp = &n->ws_planes;
  // Set when the previous stage has already written this stage's operand planes into *wsp
  // (nnet_normalize_planes_kernel): the split pass is then skipped.
  int planes_ready = 0;
  for (size_t si = 0; si < ns; ++si) {
    const Stage &s = n->stages[si];
    const bool last = (si + 1 == ns);
    const int spost = last ? post : 0;
    float *dst = last ? d_out : bufs[which];
    const int dst_ld = last ? ldo : RoundUp(s.out_dim, 16);
    const int zero_pad = last ? 0 : 1;
    int64_t out_rows = cur_rows;
    if (s.kind == AMLIKE_SPLICE) {
      // The gather kernel writes whole rows of dst_ld floats; on the last stage it must
      // stay inside out_dim.
      const Stage *cons = !last ? &n->stages[si + 1] : nullptr;
      const bool direct = cons && cons->kind == AMLIKE_AFFINE && cons->in_dim == s.out_dim &&
                          (size_t)kSpliceStageRows * (s.in_dim | 1) * sizeof(float) <= 60 * 1024 && !n->gemm_f32 && !n->gemm_inline_split && !n->no_direct_planes;
      if (direct) {
        // the spliced rows go straight into the consumer's operand planes
        const bool cons_fused = cons->fuse_gs && si + 2 < ns;
        const int ckp = cons_fused ? cons->Kpn : cons->Kp;
        const int64_t total = gather_rows[gi];
        const size_t cstride = (size_t)total * ckp;
        rc = wsp->Reserve((cons->f16 == 2 ? 5 : 3) * cstride * sizeof(unsigned short));
        if (rc != AMLIKE_OK) return rc;
        unsigned short *pbase = static_cast<unsigned short *>(wsp->ptr);
        int *flag = n->d_flags + si + 1;
        const int one_col = (cons_fused || cons->d_wob) ? cons->in_dim : -1;
        const int sblocks = (int)((total + 63) / 64);
        const size_t ssmem = (size_t)kSpliceStageRows * (s.in_dim | 1) * sizeof(float);
        const int lp = padded ? 0 : pad_l, nspl = s.left + s.right + 1, sdim = s.in_dim - s.const_dim;
        auto splice_to = [&](int prec, unsigned short *dstp, int gmode) {
          Planes xp{dstp, cstride, ckp};
          if (prec)
            hipLaunchKernelGGL(nnet_splice_planes_kernel<1>, dim3(sblocks), dim3(256), ssmem, st, cur, cur_ld,
                               d_maps + gi * num_utts, num_utts, lp, nspl, sdim, s.out_dim, (int)total, xp,
                               one_col, flag, gmode);
          else
            hipLaunchKernelGGL(nnet_splice_planes_kernel<0>, dim3(sblocks), dim3(256), ssmem, st, cur, cur_ld,
                               d_maps + gi * num_utts, num_utts, lp, nspl, sdim, s.out_dim, (int)total, xp,
                               one_col, flag, gmode);
        };
        if (cons->f16 == 2) {
          splice_to(1, pbase, 1);
          AMLIKE_HIP(hipGetLastError());
          splice_to(0, pbase + 2 * cstride, 2);
        } else {
          splice_to(cons->f16, pbase, 0);
        }
        AMLIKE_HIP(hipGetLastError());
        planes_ready = 2;
        out_rows = gather_rows[gi++];
        padded = true;
        cur = nullptr;  // no fp32 copy exists
        cur_ld = dst_ld;
        cur_rows = out_rows;
        continue;
      }
      rc = launch_gather(padded ? 0 : pad_l, s.left + s.right + 1, s.in_dim - s.const_dim,
                         s.const_dim, dst, last ? s.out_dim : dst_ld, s.out_dim);
      if (rc != AMLIKE_OK) return rc;
      if (last && ldo != s.out_dim) {
        SetError("amlike_nnet_compute: a net ending in a SpliceComponent needs ldo == output dim");
        return AMLIKE_EINVAL;
      }
      out_rows = gather_rows[gi++];
      padded = true;
    } else if (s.kind == AMLIKE_SPLICE_MAX) {
      int max_out = 0;
      for (int u = 0; u < num_utts; ++u) max_out = std::max(max_out, maps[gi * num_utts + u].out_rows);
      dim3 grid(std::min(max_out, 2048), num_utts);
      hipLaunchKernelGGL(nnet_pad_splicemax_kernel, grid, dim3(256), 0, st, cur, cur_ld, d_maps + gi * num_utts,
                         num_utts, padded ? 0 : pad_l, s.left + s.right + 1, s.in_dim, dst, dst_ld, zero_pad);
      AMLIKE_HIP(hipGetLastError());
      out_rows = gather_rows[gi++];
      padded = true;
    } else if (s.kind == AMLIKE_AFFINE && s.fuse_gs && si + 1 < ns) {
      // affine + p-norm (+ normalize) : planes of the input with a 1.0 column for the bias,
      // the fused GEMM writing [rows x groups], then the row normalisation.
      const Stage &pn = n->stages[si + 1];
      const bool pn_last = (si + 2 == ns);
      const size_t pstride = (size_t)cur_rows * s.Kpn;
      // fp16x2 on trial keeps both plane images: [2 planes fp16][3 planes bf16]
      rc = wsp->Reserve((s.f16 == 2 ? 5 : 3) * pstride * sizeof(unsigned short));
      if (rc != AMLIKE_OK) return rc;
      unsigned short *pbase = static_cast<unsigned short *>(wsp->ptr);
      int *flag = n->d_flags + si;
      const int sblocks = (int)std::min<size_t>((pstride / 8 + 255) / 256, 16384);
      auto split = [&](int prec, unsigned short *dstp, int gmode) {
        if (planes_ready) return;  // the producing stage has written them
        Planes xp{dstp, pstride, s.Kpn};
        if (prec)
          hipLaunchKernelGGL(nnet_split_planes_kernel<1>, dim3(sblocks), dim3(256), 0, st, cur, cur_ld,
                             s.in_dim, (int)cur_rows, xp, s.in_dim, flag, gmode);
        else
          hipLaunchKernelGGL(nnet_split_planes_kernel<0>, dim3(sblocks), dim3(256), 0, st, cur, cur_ld,
                             s.in_dim, (int)cur_rows, xp, s.in_dim, flag, gmode);
      };
      const int G = pn.out_dim;
      // without a Normalize the p-norm output is this pair's result; with one it goes to the
      // other scratch buffer first
      const bool norm = pn.normalize_after != 0;
      float *pdst = norm ? bufs[which] : (pn_last ? d_out : bufs[which]);
      const int pld = norm ? RoundUp(G, 16) : (pn_last ? ldo : RoundUp(G, 16));
      dim3 fgrid(s.Npad / 160, (unsigned)((cur_rows + 127) / 128));
#define AMLIKE_PNORM_LAUNCH(GS, PREC, XPL, WPN, GMODE)                                            \
  hipLaunchKernelGGL((nnet_gemm_pnorm_kernel<GS, PREC>), fgrid, dim3(256), 0, st,                \
                     reinterpret_cast<const uint4 *>(XPL), pstride / 8, WPN, s.Kpn, pdst, pld,    \
                     (int)cur_rows, s.Npad, G, flag, GMODE, n->pnorm_xcd_order)
      auto fused = [&](int prec, const unsigned short *xpl, const uint4 *wpn, int gmode) {
        if (s.fuse_gs == 10 && prec) AMLIKE_PNORM_LAUNCH(10, 1, xpl, wpn, gmode);
        else if (s.fuse_gs == 10) AMLIKE_PNORM_LAUNCH(10, 0, xpl, wpn, gmode);
        else if (s.fuse_gs == 5 && prec) AMLIKE_PNORM_LAUNCH(5, 1, xpl, wpn, gmode);
        else if (s.fuse_gs == 5) AMLIKE_PNORM_LAUNCH(5, 0, xpl, wpn, gmode);
        else if (prec) AMLIKE_PNORM_LAUNCH(8, 1, xpl, wpn, gmode);
        else AMLIKE_PNORM_LAUNCH(8, 0, xpl, wpn, gmode);
      };
      if (s.f16 == 2) {
        split(1, pbase, 1);
        AMLIKE_HIP(hipGetLastError());
        const Stage *nx1 = (norm && si + 2 < ns) ? &n->stages[si + 2] : nullptr;
        if (s.d_wpr && cur_rows >= 64 && n->first_rows && nx1 && nx1->kind == AMLIKE_AFFINE && nx1->f16 == 1 &&
            nx1->in_dim == G && G <= 240 && n->rows_norm && !n->gemm_f32 && !n->gemm_inline_split &&
            !n->no_direct_planes) {
          // Rows stationary, one wave per SIMD, normalising its rows and writing the next stage's
          // planes (into the other planes buffer) when the fp16x2 attempt stands; otherwise the
          // bf16x3 twin on the older kernel, and the normalisation into the same planes by a kernel
          // of its own that runs only then.
          const bool nx_fused = nx1->fuse_gs && si + 3 < ns;
          const int ckp = nx_fused ? nx1->Kpn : nx1->Kp;
          const size_t cstride = (size_t)cur_rows * ckp;
          DeviceBuffer *other = wsp == &n->ws_planes ? &n->ws_planes_b : &n->ws_planes;
          rc = other->Reserve(3 * cstride * sizeof(unsigned short));
          if (rc != AMLIKE_OK) return rc;
          const Planes cp{static_cast<unsigned short *>(other->ptr), cstride, ckp};
          const int cons_one = (nx_fused || nx1->d_wob) ? nx1->in_dim : -1;
          rc = LaunchPnormRows(s, reinterpret_cast<const uint4 *>(pbase), pstride / 8, pdst, pld, (int)cur_rows, G, st,
                               cp, cons_one, flag, 1);
          if (rc != AMLIKE_OK) return rc;
          split(0, pbase + 2 * pstride, 2);
          AMLIKE_HIP(hipGetLastError());
          fused(0, pbase + 2 * pstride, s.d_wpn_safe, 2);
          AMLIKE_HIP(hipGetLastError());
          hipLaunchKernelGGL(nnet_normalize_planes_kernel<64>, dim3((unsigned)((cur_rows + 63) / 64)), dim3(256),
                             ((size_t)64 * (G | 1) + 256) * sizeof(float), st, pdst, pld, G, (int)cur_rows, cp,
                             cons_one, flag, 2);
          AMLIKE_HIP(hipGetLastError());
          wsp = other;
          planes_ready = 1;
          cur = nullptr;  // no fp32 copy of the normalised rows exists
          cur_ld = RoundUp(G, 16);
          which ^= 1;
          ++si;  // the p-norm stage is done
          continue;
        }
        if (s.d_wpr && cur_rows >= 64 && n->first_rows) {
          // rows stationary, one wave per SIMD (the fp16x2 attempt; the bf16x3 twin keeps the older kernel)
          rc = LaunchPnormRows(s, reinterpret_cast<const uint4 *>(pbase), pstride / 8, pdst, pld, (int)cur_rows, G, st,
                               Planes{nullptr, 0, 0}, -1, flag, 1);
          if (rc != AMLIKE_OK) return rc;
        } else {
          fused(1, pbase, s.d_wpn, 1);
        }
        AMLIKE_HIP(hipGetLastError());
        split(0, pbase + 2 * pstride, 2);
        AMLIKE_HIP(hipGetLastError());
        fused(0, pbase + 2 * pstride, s.d_wpn_safe, 2);
      } else if (s.d_wpr && cur_rows >= 64) {
        // rows stationary: X read once per job, the units streamed (nnet_pnorm_rows_kernel); the
        // kernel also normalises its rows and writes the next affine stage's planes when that
        // stage reads fp16x2 planes of this width
        split(s.f16, pbase, 0);
        AMLIKE_HIP(hipGetLastError());
        const Stage *nx = (norm && si + 2 < ns) ? &n->stages[si + 2] : nullptr;
        if (nx && nx->kind == AMLIKE_AFFINE && nx->f16 == 1 && nx->in_dim == G && G <= 480 && n->rows_norm &&
            !n->gemm_f32 && !n->gemm_inline_split && !n->no_direct_planes) {
          const bool nx_fused = nx->fuse_gs && si + 3 < ns;
          const int ckp = nx_fused ? nx->Kpn : nx->Kp;
          const size_t cstride = (size_t)cur_rows * ckp;
          DeviceBuffer *other = wsp == &n->ws_planes ? &n->ws_planes_b : &n->ws_planes;
          rc = other->Reserve(3 * cstride * sizeof(unsigned short));
          if (rc != AMLIKE_OK) return rc;
          rc = LaunchPnormRows(s, reinterpret_cast<const uint4 *>(pbase), pstride / 8, pdst, pld, (int)cur_rows, G, st,
                               Planes{static_cast<unsigned short *>(other->ptr), cstride, ckp},
                               (nx_fused || nx->d_wob) ? nx->in_dim : -1);
          if (rc != AMLIKE_OK) return rc;
          wsp = other;
          planes_ready = 1;
          cur = nullptr;  // no fp32 copy of the normalised rows exists
          cur_ld = RoundUp(G, 16);
          which ^= 1;
          ++si;  // the p-norm stage is done
          continue;
        }
        rc = LaunchPnormRows(s, reinterpret_cast<const uint4 *>(pbase), pstride / 8, pdst, pld, (int)cur_rows, G, st);
        if (rc != AMLIKE_OK) return rc;
      } else {
        split(s.f16, pbase, 0);
        AMLIKE_HIP(hipGetLastError());
        fused(s.f16, pbase, s.d_wpn, 0);
      }
      planes_ready = 0;
#undef AMLIKE_PNORM_LAUNCH
      AMLIKE_HIP(hipGetLastError());
      float *fin = pdst;
      int fin_ld = pld;
      // The consumer of the normalised rows, when it is an affine stage on unconditional
      // fp16x2 operands read from planes: write its planes directly.
      const Stage *cons = (norm && si + 2 < ns) ? &n->stages[si + 2] : nullptr;
      const bool direct = cons && cons->kind == AMLIKE_AFFINE && cons->f16 == 1 && cons->in_dim == G &&
                          G <= 480 && !n->gemm_f32 && !n->gemm_inline_split && !n->no_direct_planes;
      if (direct) {
        const bool cons_fused = cons->fuse_gs && si + 3 < ns;
        const int ckp = cons_fused ? cons->Kpn : cons->Kp;
        const size_t cstride = (size_t)cur_rows * ckp;
        rc = wsp->Reserve(3 * cstride * sizeof(unsigned short));  // what the consumer asks for
        if (rc != AMLIKE_OK) return rc;
        Planes cp{static_cast<unsigned short *>(wsp->ptr), cstride, ckp};
        // the 1.0 the consumer's bias column multiplies (fused affine + p-norm, 8-wave output layer)
        const int cons_one = (cons_fused || cons->d_wob) ? cons->in_dim : -1;
        const int rb = G <= 240 ? 64 : 32;
        const size_t nsmem = ((size_t)rb * (G | 1) + 256) * sizeof(float);
        if (rb == 64)
          hipLaunchKernelGGL(nnet_normalize_planes_kernel<64>, dim3((unsigned)((cur_rows + 63) / 64)), dim3(256),
                             nsmem, st, pdst, pld, G, (int)cur_rows, cp, cons_one, nullptr, 0);
        else
          hipLaunchKernelGGL(nnet_normalize_planes_kernel<32>, dim3((unsigned)((cur_rows + 31) / 32)), dim3(256),
                             nsmem, st, pdst, pld, G, (int)cur_rows, cp, cons_one, nullptr, 0);
        AMLIKE_HIP(hipGetLastError());
        planes_ready = 1;
        fin = nullptr;  // no fp32 copy exists
        fin_ld = RoundUp(G, 16);
      } else if (norm) {
        fin = pn_last ? d_out : bufs[which ^ 1];
        fin_ld = pn_last ? ldo : RoundUp(G, 16);
        const int blocks = (int)std::min<int64_t>((cur_rows + 3) / 4, 8192);
        hipLaunchKernelGGL(nnet_normalize_kernel, dim3(blocks), dim3(256), 0, st, pdst, pld, G,
                           (int)cur_rows, fin, fin_ld, pn_last ? 0 : 1);
        AMLIKE_HIP(hipGetLastError());
        which ^= 1;  // the result sits in the other buffer
      } else if (!pn_last && pld != G) {
        hipLaunchKernelGGL(nnet_zero_pad_kernel, dim3(256), dim3(256), 0, st, pdst, (int)cur_rows, pld, G);
        AMLIKE_HIP(hipGetLastError());
      }
      if (pn_last && post) {
        const int blocks = (int)std::min<int64_t>(cur_rows, 16384);
        hipLaunchKernelGGL(nnet_elementwise_kernel, dim3(blocks), dim3(256), 0, st, d_out, ldo, G,
                           (int)cur_rows, 0, nullptr, 0.0f, post, n->d_log_priors, prob_scale, d_out, ldo, 0);
        AMLIKE_HIP(hipGetLastError());
      }
      cur = fin;
      cur_ld = fin_ld;
      which ^= 1;
      ++si;  // the p-norm stage is done
      continue;
    } else if (s.kind == AMLIKE_AFFINE && FusedOutputApplies(n, si, d_out, ldo, post)) {
      // the net's last affine layer and its softmax in one kernel: [rows x N] is written once
      const Stage &sm = n->stages[si + 1];
      (void)sm;
      const size_t pstride = (size_t)cur_rows * s.Kp;
      rc = wsp->Reserve(3 * pstride * sizeof(unsigned short));
      if (rc != AMLIKE_OK) return rc;
      unsigned short *pbase = static_cast<unsigned short *>(wsp->ptr);
      if (!planes_ready) {
        if (cur_ld < s.Kp || (cur_ld & 3) || (reinterpret_cast<uintptr_t>(cur) & 15)) {
          SetError("amlike_nnet_compute: internal layout error before the output layer");
          return AMLIKE_EINVAL;
        }
        Planes xp{pbase, pstride, s.Kp};
        const int sblocks = (int)std::min<size_t>((pstride / 8 + 255) / 256, 16384);
        if (s.f16)
          hipLaunchKernelGGL(nnet_split_planes_kernel<1>, dim3(sblocks), dim3(256), 0, st, cur, cur_ld, s.in_dim,
                             (int)cur_rows, xp, s.d_wob ? s.in_dim : -1, nullptr, 0);
        else
          hipLaunchKernelGGL(nnet_split_planes_kernel<0>, dim3(sblocks), dim3(256), 0, st, cur, cur_ld, s.in_dim,
                             (int)cur_rows, xp, -1, nullptr, 0);
        AMLIKE_HIP(hipGetLastError());
      }
      planes_ready = 0;
      // (the planes carry 1.0 in column in_dim whenever d_wob exists: the 8-wave kernel's bias column)
      // (whole-line stores need the line boundary of a row to depend on the row modulo 4 only; two tiles)
      if (s.d_wob && n->out_rows && cur_rows >= 64 && (ldo & 7) == 0 && ldo < (1 << 24) && s.Nob >= 64)
        rc = LaunchOutRows(n, s, reinterpret_cast<const uint4 *>(pbase), pstride / 8, n->d_log_priors, d_out, ldo,
                           (int)cur_rows, post, prob_scale, st);
      else
        rc = LaunchOutSoftmax(n, s, reinterpret_cast<const uint4 *>(pbase), pstride / 8, n->d_log_priors, d_out, ldo,
                              (int)cur_rows, post, prob_scale, st);
      if (rc != AMLIKE_OK) return rc;
      cur = d_out;
      cur_ld = ldo;
      ++si;  // the softmax stage is done
      continue;
    } else if (s.kind == AMLIKE_AFFINE) {
      if (cur_ld < s.Kp || (cur_ld & 3) || (reinterpret_cast<uintptr_t>(cur) & 15)) {
        SetError("amlike_nnet_compute: internal layout error before an affine stage");
        return AMLIKE_EINVAL;
      }
      dim3 grid((s.out_dim + 127) / 128, (unsigned)((cur_rows + 127) / 128));
      if (n->gemm_f32)
        hipLaunchKernelGGL(nnet_gemm_bias_kernel, grid, dim3(256), 0, st, cur, cur_ld, s.d_w, s.Kp,
                           s.d_bias, dst, dst_ld, (int)cur_rows, s.out_dim);
      else if (n->gemm_inline_split)
        hipLaunchKernelGGL(nnet_gemm_bias_bf16x3_kernel, grid, dim3(256), 0, st, cur, cur_ld, s.d_wp,
                           s.Kp, s.d_bias, dst, dst_ld, (int)cur_rows, s.out_dim);
      else {
        // the input as bf16x3 planes (one cheap pass), then the conversion-free GEMM
        const bool big_out = (size_t)cur_rows * s.out_dim * sizeof(float) > kStreamingBytes;
        const size_t pstride = (size_t)cur_rows * s.Kp;
        rc = wsp->Reserve((s.f16 == 2 ? 5 : 3) * pstride * sizeof(unsigned short));
        if (rc != AMLIKE_OK) return rc;
        unsigned short *pbase = static_cast<unsigned short *>(wsp->ptr);
        int *flag = n->d_flags + si;
        const int sblocks = (int)std::min<size_t>((pstride / 8 + 255) / 256, 16384);
        auto split_gemm = [&](int prec, unsigned short *xpl, const uint4 *wpo, int gmode) {
          Planes xp{xpl, pstride, s.Kp};
          const uint4 *x4 = reinterpret_cast<const uint4 *>(xpl);
          const size_t wplane = (size_t)s.out_dim * (s.Kp / 8);
          if (prec) {
            if (!planes_ready)
              hipLaunchKernelGGL(nnet_split_planes_kernel<1>, dim3(sblocks), dim3(256), 0, st, cur, cur_ld,
                                 s.in_dim, (int)cur_rows, xp, -1, flag, gmode);
            hipLaunchKernelGGL((big_out ? nnet_gemm_bias_planes_kernel<1, true> : nnet_gemm_bias_planes_kernel<1, false>),
                               grid, dim3(256), 0, st, x4, pstride / 8, wpo, wplane, (size_t)s.out_dim, s.Kp,
                               s.d_bias, dst, dst_ld, (int)cur_rows, s.out_dim, flag, gmode);
          } else {
            if (!planes_ready)
              hipLaunchKernelGGL(nnet_split_planes_kernel<0>, dim3(sblocks), dim3(256), 0, st, cur, cur_ld,
                                 s.in_dim, (int)cur_rows, xp, -1, flag, gmode);
            hipLaunchKernelGGL((big_out ? nnet_gemm_bias_planes_kernel<0, true> : nnet_gemm_bias_planes_kernel<0, false>),
                               grid, dim3(256), 0, st, x4, pstride / 8, wpo, wplane, (size_t)s.out_dim, s.Kp,
                               s.d_bias, dst, dst_ld, (int)cur_rows, s.out_dim, flag, gmode);
          }
        };
        if (s.f16 == 2) {
          split_gemm(1, pbase, s.d_wpo, 1);
          AMLIKE_HIP(hipGetLastError());
          split_gemm(0, pbase + 2 * pstride, s.d_wpo_safe, 2);
        } else {
          split_gemm(s.f16, pbase, s.d_wpo, 0);
        }
        planes_ready = 0;
      }
      AMLIKE_HIP(hipGetLastError());
      if (!last && dst_ld != s.out_dim) {
        hipLaunchKernelGGL(nnet_zero_pad_kernel, dim3(256), dim3(256), 0, st, dst, (int)cur_rows,
                           dst_ld, s.out_dim);
        AMLIKE_HIP(hipGetLastError());
      }
    } else if (s.kind == AMLIKE_PNORM) {
      const int blocks = (int)std::min<int64_t>((cur_rows + 3) / 4, 8192);
      const size_t smem = (size_t)4 * s.in_dim * sizeof(float);
      if (s.normalize_after)
        hipLaunchKernelGGL(nnet_pnorm_norm_kernel<true>, dim3(blocks), dim3(256), smem, st, cur, cur_ld,
                           s.in_dim, (int)cur_rows, s.out_dim, s.p, dst, dst_ld, zero_pad);
      else
        hipLaunchKernelGGL(nnet_pnorm_norm_kernel<false>, dim3(blocks), dim3(256), smem, st, cur, cur_ld,
                           s.in_dim, (int)cur_rows, s.out_dim, s.p, dst, dst_ld, zero_pad);
      AMLIKE_HIP(hipGetLastError());
    } else if (s.kind == AMLIKE_NORMALIZE) {
      const int blocks = (int)std::min<int64_t>((cur_rows + 3) / 4, 8192);
      hipLaunchKernelGGL(nnet_normalize_kernel, dim3(blocks), dim3(256), 0, st, cur, cur_ld,
                         s.in_dim, (int)cur_rows, dst, dst_ld, zero_pad);
      AMLIKE_HIP(hipGetLastError());
    } else if (s.kind == AMLIKE_SOFTMAX) {
      const int blocks = (int)std::min<int64_t>(cur_rows, 16384);
      const bool vec4 = (s.in_dim & 3) == 0 && (cur_ld & 3) == 0 && (dst_ld & 3) == 0 &&
                        ((reinterpret_cast<uintptr_t>(cur) | reinterpret_cast<uintptr_t>(dst) |
                          reinterpret_cast<uintptr_t>(n->d_log_priors)) & 15) == 0;
      if (vec4 && s.in_dim <= 256 * 4 * 4)
        hipLaunchKernelGGL(nnet_softmax4_kernel<4>, dim3(blocks), dim3(256), 0, st, cur, cur_ld, s.in_dim,
                           (int)cur_rows, spost, n->d_log_priors, prob_scale, dst, dst_ld, zero_pad);
      else if (vec4 && s.in_dim <= 256 * 4 * 10)
        hipLaunchKernelGGL(nnet_softmax4_kernel<10>, dim3(blocks), dim3(256), 0, st, cur, cur_ld, s.in_dim,
                           (int)cur_rows, spost, n->d_log_priors, prob_scale, dst, dst_ld, zero_pad);
      else if (s.in_dim <= 256 * kSoftmaxRegs)
        hipLaunchKernelGGL(nnet_softmax_kernel, dim3(blocks), dim3(256), 0, st, cur, cur_ld, s.in_dim,
                           (int)cur_rows, spost, n->d_log_priors, prob_scale, dst, dst_ld, zero_pad);
      else
        hipLaunchKernelGGL(nnet_softmax_wide_kernel, dim3(blocks), dim3(256), 0, st, cur, cur_ld,
                           s.in_dim, (int)cur_rows, spost, n->d_log_priors, prob_scale, dst, dst_ld,
                           zero_pad);
      AMLIKE_HIP(hipGetLastError());
    } else if (s.kind == AMLIKE_SUMGROUP) {
      const int blocks = (int)std::min<int64_t>(cur_rows, 16384);
      hipLaunchKernelGGL(nnet_sumgroup_kernel, dim3(blocks), dim3(256), 0, st, cur, cur_ld, s.d_begins,
                         s.out_dim, (int)cur_rows, spost, n->d_log_priors, prob_scale, dst, dst_ld,
                         zero_pad);
      AMLIKE_HIP(hipGetLastError());
    } else if (s.kind == AMLIKE_MAXOUT || s.kind == AMLIKE_PERMUTE) {
      const int blocks = (int)std::min<int64_t>(cur_rows, 16384);
      hipLaunchKernelGGL(nnet_regroup_kernel, dim3(blocks), dim3(256), 0, st, cur, cur_ld, s.d_begins,
                         s.kind == AMLIKE_MAXOUT ? s.in_dim / s.out_dim : 0, s.out_dim, (int)cur_rows,
                         spost, n->d_log_priors, prob_scale, dst, dst_ld, zero_pad);
      AMLIKE_HIP(hipGetLastError());
    } else {
      const int blocks = (int)std::min<int64_t>(cur_rows, 16384);
      hipLaunchKernelGGL(nnet_elementwise_kernel, dim3(blocks), dim3(256), 0, st, cur, cur_ld,
                         s.in_dim, (int)cur_rows, s.kind, s.d_bias, s.p, spost, n->d_log_priors,
                         prob_scale, dst, dst_ld, zero_pad);
      AMLIKE_HIP(hipGetLastError());
    }
    const bool fused_post = (s.kind == AMLIKE_SOFTMAX || s.kind == AMLIKE_SUMGROUP ||
                             s.kind >= AMLIKE_SIGMOID);
    if (last && post && !fused_post) {
      // The last stage has no fused post-processing (e.g. an affine output): one more pass.
      const int blocks = (int)std::min<int64_t>(out_rows, 16384);
      hipLaunchKernelGGL(nnet_elementwise_kernel, dim3(blocks), dim3(256), 0, st, d_out, ldo,
                         s.out_dim, (int)out_rows, 0, nullptr, 0.0f, post, n->d_log_priors, prob_scale,
                         d_out, ldo, 0);
      AMLIKE_HIP(hipGetLastError());
    }
    cur = dst;
    cur_ld = dst_ld;
    cur_rows = out_rows;
    which ^= 1;
  }
  if (total_out_rows) *total_out_rows = cur_rows;
  return AMLIKE_OK;
}

}  // namespace amlike

extern "C" {

int amlike_nnet_create(const amlike_component_t *comps, int32_t num, const float *priors,
                       int32_t num_priors, amlike_nnet_t **out) {
  if (int brc = BindThread()) return brc;
  if (!comps || num <= 0 || !out) {
    SetError("amlike_nnet_create: invalid argument");
    return AMLIKE_EINVAL;
  }
  if (CurrentDevice() < 0) {
    int rc = amlike_init(0);
    if (rc != AMLIKE_OK) return rc;
  }
  amlike_nnet *n = new amlike_nnet;
  n->device = CurrentDevice();
  if (const char *e = getenv("AMLIKE_NNET_GEMM")) {
    n->gemm_f32 = strcmp(e, "f32") == 0;
    n->gemm_inline_split = strcmp(e, "inline") == 0;
  }
  if (const char *e = getenv("AMLIKE_NNET_PNORM_XCD")) n->pnorm_xcd_order = e[0] != '0';
  if (const char *e = getenv("AMLIKE_NNET_PNORM_ROWS")) n->pnorm_rows = e[0] != '0';
  if (const char *e = getenv("AMLIKE_NNET_COLLAPSE")) n->collapse = e[0] != '0';
  if (const char *e = getenv("AMLIKE_NNET_OUT_ROWS")) n->out_rows = e[0] != '0';
  if (const char *e = getenv("AMLIKE_NNET_FIRST_ROWS")) n->first_rows = e[0] != '0';
  if (const char *e = getenv("AMLIKE_NNET_ROWS_NORM")) n->rows_norm = e[0] != '0';
  if (const char *e = getenv("AMLIKE_NNET_OUT_CUS")) n->out_cus = std::max(1, atoi(e));
  n->out_no_stagger = getenv("AMLIKE_NNET_OUT_NO_STAGGER") != nullptr;
  n->no_fused_out = getenv("AMLIKE_NNET_NO_FUSED_OUT") != nullptr;
  n->no_direct_planes = getenv("AMLIKE_NNET_NO_DIRECT_PLANES") != nullptr;
  n->input_dim = comps[0].input_dim;
  int dim = comps[0].input_dim;
  int rc = AMLIKE_OK;
  for (int i = 0; i < num && rc == AMLIKE_OK; ++i) {
    const amlike_component_t &c = comps[i];
    if (c.input_dim != dim) {
      SetError("amlike_nnet_create: component %d expects input dim %d, previous output is %d", i,
               c.input_dim, dim);
      rc = AMLIKE_EDIM;
      break;
    }
    Stage s;
    s.kind = c.kind;
    s.in_dim = c.input_dim;
    s.out_dim = c.output_dim;
    // What the affine stage is built from: the caller's matrix, or the one a DctComponent
    // amounts to; no bias stands for zeros (FixedLinearComponent).
    const float *lin = c.linear, *lin_bias = c.bias;
    std::vector<float> own_lin, zero_bias;
    if (c.kind == AMLIKE_DCT) {
      if (!BuildDctMatrix(c.input_dim, c.const_dim, c.output_dim, c.left_context != 0, &own_lin)) {
        SetError("amlike_nnet_create: bad DctComponent %d (dim %d, dct dim %d, output dim %d)", i,
                 c.input_dim, c.const_dim, c.output_dim);
        rc = AMLIKE_EINVAL;
        break;
      }
      lin = own_lin.data();
      lin_bias = nullptr;
      s.kind = AMLIKE_AFFINE;
    }
    if (s.kind == AMLIKE_AFFINE && lin && !lin_bias && c.output_dim > 0) {
      zero_bias.assign(c.output_dim, 0.0f);
      lin_bias = zero_bias.data();
    }
    switch (s.kind) {
      case AMLIKE_SPLICE: {
        s.left = c.left_context;
        s.right = c.right_context;
        s.const_dim = c.const_dim;
        const int want = (c.input_dim - c.const_dim) * (1 + c.left_context + c.right_context) + c.const_dim;
        if (c.left_context < 0 || c.right_context < 0 || c.const_dim < 0 ||
            c.const_dim >= c.input_dim || c.output_dim != want) {
          SetError("amlike_nnet_create: bad SpliceComponent %d", i);
          rc = AMLIKE_EINVAL;
        }
        n->left += s.left;
        n->right += s.right;
        break;
      }
      case AMLIKE_SPLICE_MAX:
        s.left = c.left_context;
        s.right = c.right_context;
        if (c.left_context < 0 || c.right_context < 0 || c.input_dim <= 0 || c.output_dim != c.input_dim) {
          SetError("amlike_nnet_create: bad SpliceMaxComponent %d", i);
          rc = AMLIKE_EINVAL;
        }
        n->left += s.left;
        n->right += s.right;
        break;
      case AMLIKE_AFFINE: {
        if (!lin || !lin_bias || c.input_dim <= 0 || c.output_dim <= 0) {
          SetError("amlike_nnet_create: affine component %d without parameters", i);
          rc = AMLIKE_EINVAL;
          break;
        }
        // An affine stage right behind another one (FixedAffine then Affine: the LDA-like input
        // transform of the p-norm recipes) becomes one stage, W = W2 W1, b = W2 b1 + b2, as
        // Nnet::Collapse does for nnet-am-copy --collapse (nnet2/nnet-nnet.cc:624-663,
        // nnet2/nnet-component.cc:1292-1306), when that costs no more arithmetic per frame.
        // Sums in double, rounded to float once.
        int in_dim = c.input_dim;
        s.normalised_input = i > 0 && comps[i - 1].kind == AMLIKE_NORMALIZE;
        std::vector<float> both_w, both_b;
        if (n->collapse && !n->stages.empty()) {
          Stage &pv = n->stages.back();
          if (pv.kind == AMLIKE_AFFINE && !pv.fuse_gs && !pv.h_w.empty() && pv.out_dim == c.input_dim &&
              (int64_t)c.output_dim * pv.in_dim <=
                  (int64_t)c.output_dim * c.input_dim + (int64_t)pv.out_dim * pv.in_dim) {
            in_dim = pv.in_dim;
            both_w.resize((size_t)c.output_dim * in_dim);
            both_b.resize(c.output_dim);
            std::vector<double> row(in_dim);
            for (int o = 0; o < c.output_dim; ++o) {
              std::fill(row.begin(), row.end(), 0.0);
              double bsum = lin_bias[o];
              const float *w2 = lin + (size_t)o * c.input_dim;
              for (int m = 0; m < c.input_dim; ++m) {
                const double f = w2[m];
                const float *w1 = &pv.h_w[(size_t)m * pv.Kp];
                for (int k = 0; k < in_dim; ++k) row[k] += f * w1[k];
                bsum += f * pv.h_b[m];
              }
              for (int k = 0; k < in_dim; ++k) both_w[(size_t)o * in_dim + k] = (float)row[k];
              both_b[o] = (float)bsum;
            }
            s.normalised_input = pv.normalised_input;
            FreeStage(pv);
            n->stages.pop_back();
            lin = both_w.data();
            lin_bias = both_b.data();
            s.in_dim = in_dim;
          }
        }
        s.Kp = RoundUp(in_dim, 16);
        std::vector<float> w((size_t)c.output_dim * s.Kp, 0.0f);
        for (int r = 0; r < c.output_dim; ++r)
          memcpy(&w[(size_t)r * s.Kp], lin + (size_t)r * in_dim, in_dim * sizeof(float));
        s.h_w = w;
        s.h_b.assign(lin_bias, lin_bias + c.output_dim);
        std::vector<uint16_t> wp(3 * w.size());
        for (size_t idx = 0; idx < w.size(); ++idx) {
          uint16_t parts[3];
          SplitBf16x3(w[idx], parts);
          for (int q = 0; q < 3; ++q) wp[q * w.size() + idx] = parts[q];
        }
        // fp16x2 operands where the input is the output of a NormalizeComponent (every value
        // within sqrt(dim)) and the parameters are well inside the fp16 range
        {
          float wmax = 0.0f;
          for (float v : w) wmax = std::max(wmax, std::fabs(v));
          for (int r = 0; r < c.output_dim; ++r) wmax = std::max(wmax, std::fabs(lin_bias[r]));
          const bool w_ok = std::isfinite(wmax) && wmax <= 1024.0f && !getenv("AMLIKE_NNET_NOF16");
          s.f16 = !w_ok ? 0 : s.normalised_input ? 1 : (getenv("AMLIKE_NNET_NOF16TRY") ? 0 : 2);
        }
        std::vector<uint16_t> wpo = BuildPlaneWeights(w, s.Kp, c.output_dim, s.f16);
        hipError_t e = hipMalloc(&s.d_w, w.size() * sizeof(float));
        if (e == hipSuccess && s.f16 == 2) {  // the bf16x3 image for the fallback
          std::vector<uint16_t> safe = BuildPlaneWeights(w, s.Kp, c.output_dim, 0);
          e = hipMalloc(&s.d_wpo_safe, safe.size() * sizeof(uint16_t));
          if (e == hipSuccess)
            e = hipMemcpy(s.d_wpo_safe, safe.data(), safe.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) e = hipMalloc(&s.d_wpo, wpo.size() * sizeof(uint16_t));
        if (e == hipSuccess)
          e = hipMemcpy(s.d_wpo, wpo.data(), wpo.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s.d_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&s.d_wp, wp.size() * sizeof(uint16_t));
        if (e == hipSuccess)
          e = hipMemcpy(s.d_wp, wp.data(), wp.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        {
          // padded to whole 32-unit tiles with -inf: a padding unit then has score -inf in the
          // fused output-layer kernel (no share in the softmax sum); the other kernels stop at out_dim
          std::vector<float> bpad(RoundUp(c.output_dim, 32), -INFINITY);
          std::copy(lin_bias, lin_bias + c.output_dim, bpad.begin());
          if (e == hipSuccess) e = hipMalloc(&s.d_bias, bpad.size() * sizeof(float));
          if (e == hipSuccess)
            e = hipMemcpy(s.d_bias, bpad.data(), bpad.size() * sizeof(float), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
          SetError("amlike_nnet_create: %s", hipGetErrorString(e));
          rc = e == hipErrorOutOfMemory ? AMLIKE_EOOM : AMLIKE_EHIP;
        }
        break;
      }
      case AMLIKE_PNORM:
        s.p = c.p;
        if (c.output_dim <= 0 || c.input_dim % c.output_dim != 0 || c.p < 0 ||
            (size_t)4 * c.input_dim * sizeof(float) > 60 * 1024) {
          SetError("amlike_nnet_create: bad PnormComponent %d", i);
          rc = AMLIKE_EINVAL;
        } else if (!n->stages.empty() && n->stages.back().kind == AMLIKE_AFFINE && c.p == 2.0f &&
                   !n->gemm_f32 && !n->gemm_inline_split && !getenv("AMLIKE_NNET_NOFUSE")) {
          // affine + p-norm in one kernel when the group size is one the kernel is built for
          Stage &a = n->stages.back();
          const int gs = c.input_dim / c.output_dim;
          if ((gs == 10 || gs == 5 || gs == 8) && a.out_dim == c.input_dim && !a.h_w.empty()) {
            std::vector<uint16_t> img = BuildPnormWeights(a.h_w, a.Kp, a.h_b, a.out_dim, a.in_dim,
                                                          a.f16, &a.Kpn, &a.Npad);
            hipError_t e = hipMalloc(&a.d_wpn, img.size() * sizeof(uint16_t));
            if (e == hipSuccess)
              e = hipMemcpy(a.d_wpn, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
            if (e == hipSuccess && a.f16 == 2) {
              int kpn2 = 0, npad2 = 0;
              std::vector<uint16_t> safe = BuildPnormWeights(a.h_w, a.Kp, a.h_b, a.out_dim, a.in_dim, 0,
                                                             &kpn2, &npad2);
              e = hipMalloc(&a.d_wpn_safe, safe.size() * sizeof(uint16_t));
              if (e == hipSuccess)
                e = hipMemcpy(a.d_wpn_safe, safe.data(), safe.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
            }
            // the row-stationary kernel's image: fp16x2 operands behind a NormalizeComponent,
            // groups of 10, k of 13 steps with the bias column (AMLIKE_NNET_PNORM_ROWS=0: A/B aid)
            if (e == hipSuccess && gs == 10 && n->pnorm_rows &&
                ((a.f16 == 1 && a.Kpn == 208) || (a.f16 == 2 && a.Kpn == 368))) {
              int kpn2 = 0, npad2 = 0;
              std::vector<uint16_t> rows_img = BuildPnormRowsWeights(a.h_w, a.Kp, a.h_b, a.out_dim, a.in_dim,
                                                                     &kpn2, &npad2);
              e = hipMalloc(&a.d_wpr, rows_img.size() * sizeof(uint16_t));
              if (e == hipSuccess)
                e = hipMemcpy(a.d_wpr, rows_img.data(), rows_img.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
            }
            if (e != hipSuccess) {
              SetError("amlike_nnet_create: %s", hipGetErrorString(e));
              rc = e == hipErrorOutOfMemory ? AMLIKE_EOOM : AMLIKE_EHIP;
            } else {
              a.fuse_gs = gs;
            }
          }
        }
        break;
      case AMLIKE_SUMGROUP: {
        if (!c.sizes) {
          SetError("amlike_nnet_create: SumGroupComponent %d without sizes", i);
          rc = AMLIKE_EINVAL;
          break;
        }
        std::vector<int32_t> begins(c.output_dim + 1, 0);
        for (int j = 0; j < c.output_dim; ++j) begins[j + 1] = begins[j] + c.sizes[j];
        if (begins[c.output_dim] != c.input_dim) {
          SetError("amlike_nnet_create: SumGroupComponent %d sizes do not add up", i);
          rc = AMLIKE_EINVAL;
          break;
        }
        hipError_t e = hipMalloc(&s.d_begins, begins.size() * sizeof(int32_t));
        if (e == hipSuccess)
          e = hipMemcpy(s.d_begins, begins.data(), begins.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
          SetError("amlike_nnet_create: %s", hipGetErrorString(e));
          rc = AMLIKE_EHIP;
        }
        break;
      }
      case AMLIKE_FIXED_SCALE:
      case AMLIKE_FIXED_BIAS: {
        if (!c.bias || c.input_dim != c.output_dim || c.input_dim <= 0) {
          SetError("amlike_nnet_create: component %d needs a vector of its dimension", i);
          rc = AMLIKE_EINVAL;
          break;
        }
        hipError_t e = hipMalloc(&s.d_bias, c.input_dim * sizeof(float));
        if (e == hipSuccess)
          e = hipMemcpy(s.d_bias, c.bias, c.input_dim * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
          SetError("amlike_nnet_create: %s", hipGetErrorString(e));
          rc = AMLIKE_EHIP;
        }
        break;
      }
      case AMLIKE_MAXOUT:
        if (c.output_dim <= 0 || c.input_dim <= 0 || c.input_dim % c.output_dim != 0) {
          SetError("amlike_nnet_create: bad MaxoutComponent %d", i);
          rc = AMLIKE_EINVAL;
        }
        break;
      case AMLIKE_PERMUTE: {
        // out(r, c) = in(r, reverse_reorder[c]) with reverse_reorder[reorder[i]] = i
        std::vector<int32_t> src(std::max(c.input_dim, 0), -1);
        bool ok = c.sizes && c.input_dim == c.output_dim && c.input_dim > 0;
        for (int k = 0; ok && k < c.input_dim; ++k) {
          ok = c.sizes[k] >= 0 && c.sizes[k] < c.input_dim && src[c.sizes[k]] < 0;
          if (ok) src[c.sizes[k]] = k;
        }
        if (!ok) {
          SetError("amlike_nnet_create: PermuteComponent %d needs a permutation of 0..dim-1", i);
          rc = AMLIKE_EINVAL;
          break;
        }
        hipError_t e = hipMalloc(&s.d_begins, src.size() * sizeof(int32_t));
        if (e == hipSuccess)
          e = hipMemcpy(s.d_begins, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
          SetError("amlike_nnet_create: %s", hipGetErrorString(e));
          rc = AMLIKE_EHIP;
        }
        break;
      }
      case AMLIKE_POWER:
      case AMLIKE_SCALE:
        s.p = c.p;
        [[fallthrough]];
      case AMLIKE_NORMALIZE:
      case AMLIKE_SOFTMAX:
      case AMLIKE_SIGMOID:
      case AMLIKE_TANH:
      case AMLIKE_RELU:
      case AMLIKE_SOFTHINGE:
        if (c.input_dim != c.output_dim) {
          SetError("amlike_nnet_create: component %d must keep the dimension", i);
          rc = AMLIKE_EINVAL;
        }
        break;
      default:
        SetError("Unknown component type %d", c.kind);
        rc = AMLIKE_EINVAL;
    }
    if (rc != AMLIKE_OK) {
      (void)hipFree(s.d_w);
      (void)hipFree(s.d_wp);
      (void)hipFree(s.d_wpo);
      (void)hipFree(s.d_wpo_safe);
      (void)hipFree(s.d_bias);
      (void)hipFree(s.d_begins);
      break;
    }
    // Fuse Normalize into the Pnorm before it.
    if (c.kind == AMLIKE_NORMALIZE && !n->stages.empty() && n->stages.back().kind == AMLIKE_PNORM &&
        !n->stages.back().normalize_after && c.input_dim <= 64 * kPnormKeep) {
      n->stages.back().normalize_after = 1;
    } else {
      n->stages.push_back(s);
    }
    dim = c.output_dim;
  }
  n->output_dim = dim;
  // the net's last affine stage, behind a NormalizeComponent and in front of the closing softmax,
  // with room for the bias among its padded k columns: the image of nnet_out_rows_kernel
  if (rc == AMLIKE_OK && n->stages.size() >= 2 && n->stages.back().kind == AMLIKE_SOFTMAX) {
    Stage &a = n->stages[n->stages.size() - 2];
    if (a.kind == AMLIKE_AFFINE && !a.fuse_gs && a.f16 == 1 && a.Kp == 208 && a.in_dim < a.Kp && !a.h_w.empty() &&
        n->out_rows) {
      std::vector<uint16_t> img = BuildOutRowsWeights(a.h_w, a.Kp, a.h_b, a.out_dim, a.in_dim, a.Kp, &a.Nob);
      hipError_t e = hipMalloc(&a.d_wob, img.size() * sizeof(uint16_t));
      if (e == hipSuccess) e = hipMemcpy(a.d_wob, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        SetError("amlike_nnet_create: %s", hipGetErrorString(e));
        rc = e == hipErrorOutOfMemory ? AMLIKE_EOOM : AMLIKE_EHIP;
      }
    }
  }
  for (Stage &st : n->stages) {
    std::vector<float>().swap(st.h_w);
    std::vector<float>().swap(st.h_b);
  }
  if (rc == AMLIKE_OK && priors && num_priors > 0) {
    std::vector<float> lp(RoundUp(num_priors, 32), 0.0f);  // padded to whole 32-unit tiles
    for (int i = 0; i < num_priors; ++i) lp[i] = logf(priors[i]);
    hipError_t e = hipMalloc(&n->d_log_priors, lp.size() * sizeof(float));
    if (e == hipSuccess)
      e = hipMemcpy(n->d_log_priors, lp.data(), lp.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      SetError("amlike_nnet_create: %s", hipGetErrorString(e));
      rc = AMLIKE_EHIP;
    }
    n->num_priors = num_priors;
  }
  if (rc == AMLIKE_OK) {
    hipError_t e = hipMalloc(&n->d_flags, (n->stages.size() + 1) * sizeof(int));
    if (e != hipSuccess) {
      SetError("amlike_nnet_create: %s", hipGetErrorString(e));
      rc = AMLIKE_EHIP;
    }
  }
  if (rc == AMLIKE_OK) {
    hipError_t e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      SetError("amlike_nnet_create: %s", hipGetErrorString(e));
      rc = AMLIKE_EHIP;
    }
  }
  if (rc != AMLIKE_OK) {
    FreeNnet(n);
    return rc;
  }
  *out = n;
  return AMLIKE_OK;
}

int amlike_nnet_destroy(amlike_nnet_t *n) {
  if (int brc = BindHandle(n)) return brc;
  if (n) FreeNnet(n);
  return AMLIKE_OK;
}
int32_t amlike_nnet_input_dim(const amlike_nnet_t *n) { return n ? n->input_dim : 0; }
int32_t amlike_nnet_device(const amlike_nnet_t *n) { return n ? n->device : -1; }
int32_t amlike_nnet_output_dim(const amlike_nnet_t *n) { return n ? n->output_dim : 0; }
int32_t amlike_nnet_left_context(const amlike_nnet_t *n) { return n ? n->left : 0; }
int32_t amlike_nnet_right_context(const amlike_nnet_t *n) { return n ? n->right : 0; }

int amlike_nnet_compute_batch_dev(amlike_nnet_t *n, const float *d_feats, int32_t ld,
                                  const int32_t *utt_rows, int32_t num_utts, int32_t pad_input,
                                  int32_t mode, float prob_scale, float *d_out, int32_t ldo,
                                  void *stream, int64_t *total_out_rows) {
  if (int brc = BindHandle(n)) return brc;
  if (!n || !d_feats || !d_out || !utt_rows || num_utts <= 0 || ld < n->input_dim ||
      ldo < n->output_dim || mode < 0 || mode > 2) {
    SetError("amlike_nnet_compute: invalid argument");
    return AMLIKE_EINVAL;
  }
  return Forward(n, d_feats, ld, utt_rows, num_utts, pad_input, mode, prob_scale, d_out, ldo,
                 static_cast<hipStream_t>(stream), total_out_rows);
}

int amlike_nnet_compute_dev(amlike_nnet_t *n, const float *d_feats, int32_t T, int32_t ld,
                            int32_t pad_input, int32_t mode, float prob_scale, float *d_out,
                            int32_t ldo, void *stream) {
  if (int brc = BindHandle(n)) return brc;
  return amlike_nnet_compute_batch_dev(n, d_feats, ld, &T, 1, pad_input, mode, prob_scale, d_out,
                                       ldo, stream, nullptr);
}

int amlike_nnet_compute(amlike_nnet_t *n, const float *feats, int32_t T, int32_t ld,
                        int32_t pad_input, int32_t mode, float prob_scale, float *out, int32_t ldo) {
  if (int brc = BindHandle(n)) return brc;
  if (!n || !feats || !out || T <= 0 || ld < n->input_dim || ldo < n->output_dim) {
    SetError("amlike_nnet_compute: invalid argument");
    return AMLIKE_EINVAL;
  }
  std::lock_guard<std::mutex> lock(n->mu);
  const int t_out = pad_input ? T : T - n->left - n->right;
  if (t_out <= 0) {
    SetError("amlike_nnet_compute: %d frames give empty output", T);
    return AMLIKE_EINVAL;
  }
  // Features travel as one contiguous copy that keeps the caller's row stride (a strided
  // 2-D copy from pageable memory degenerates into one transfer per 160-byte row).
  const bool feats_1d = (size_t)ld <= (size_t)2 * n->input_dim + 8;
  const int dld = feats_1d ? ld : n->input_dim;
  int rc = n->ws_in.Reserve((size_t)T * dld * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  rc = n->ws_out.Reserve((size_t)t_out * n->output_dim * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  float *din = static_cast<float *>(n->ws_in.ptr), *dout = static_cast<float *>(n->ws_out.ptr);
  if (feats_1d)
    AMLIKE_HIP(hipMemcpyAsync(din, feats, ((size_t)(T - 1) * ld + n->input_dim) * sizeof(float),
                              hipMemcpyHostToDevice, n->stream));
  else
    AMLIKE_HIP(hipMemcpy2DAsync(din, (size_t)n->input_dim * sizeof(float), feats,
                                (size_t)ld * sizeof(float), (size_t)n->input_dim * sizeof(float), T,
                                hipMemcpyHostToDevice, n->stream));
  rc = amlike_nnet_compute_dev(n, din, T, dld, pad_input, mode, prob_scale, dout,
                               n->output_dim, n->stream);
  if (rc != AMLIKE_OK) return rc;
  return CopyMatrixToHost(out, (size_t)ldo, dout, t_out, n->output_dim, n->stream);
}

int amlike_nnet_compute_submit(amlike_nnet_t *n, const float *feats, int32_t T, int32_t ld,
                               int32_t pad_input, int32_t mode, float prob_scale, float *out,
                               int32_t ldo, int32_t *ticket) {
  if (int brc = BindHandle(n)) return brc;
  if (!n || !feats || !out || !ticket || T <= 0 || ld < n->input_dim || ldo < n->output_dim) {
    SetError("amlike_nnet_compute_submit: invalid argument");
    return AMLIKE_EINVAL;
  }
  const int t_out = pad_input ? T : T - n->left - n->right;
  if (t_out <= 0) {
    SetError("amlike_nnet_compute: %d frames give empty output", T);
    return AMLIKE_EINVAL;
  }
  if ((size_t)t_out * n->output_dim > ((size_t)1 << 30)) {
    SetError("amlike_nnet_compute_submit: more than 2^30 outputs in one submission");
    return AMLIKE_EINVAL;
  }
  std::lock_guard<std::mutex> lock(n->mu);
  int si = -1;
  for (int i = 0; i < kAsyncSlots; ++i)
    if (n->slots[i].ticket < 0 && si < 0) si = i;
  if (si < 0) {
    SetError("amlike_nnet_compute_submit: %d submissions in flight, wait for one first", kAsyncSlots);
    return AMLIKE_EBUSY;
  }
  AsyncSlot *s = &n->slots[si];
  if (!n->copy_stream) AMLIKE_HIP(hipStreamCreateWithFlags(&n->copy_stream, hipStreamNonBlocking));
  int rc = s->Prepare();
  if (rc != AMLIKE_OK) return rc;
  const bool feats_1d = (size_t)ld <= (size_t)2 * n->input_dim + 8;
  const int dld = feats_1d ? ld : n->input_dim;
  rc = s->in.Reserve((size_t)T * dld * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  rc = s->out.Reserve((size_t)t_out * n->output_dim * sizeof(float));
  if (rc != AMLIKE_OK) return rc;
  float *din = static_cast<float *>(s->in.ptr), *dout = static_cast<float *>(s->out.ptr);
  if (feats_1d)
    AMLIKE_HIP(hipMemcpyAsync(din, feats, ((size_t)(T - 1) * ld + n->input_dim) * sizeof(float),
                              hipMemcpyHostToDevice, n->stream));
  else
    AMLIKE_HIP(hipMemcpy2DAsync(din, (size_t)n->input_dim * sizeof(float), feats,
                                (size_t)ld * sizeof(float), (size_t)n->input_dim * sizeof(float), T,
                                hipMemcpyHostToDevice, n->stream));
  // the row maps are uploaded from host memory that must outlive the copy: each slot keeps
  // its own
  std::swap(n->h_utts, n->slot_utts[si]);
  rc = amlike_nnet_compute_dev(n, din, T, dld, pad_input, mode, prob_scale, dout, n->output_dim,
                               n->stream);
  std::swap(n->h_utts, n->slot_utts[si]);
  if (rc != AMLIKE_OK) return rc;
  AMLIKE_HIP(hipEventRecord(s->computed, n->stream));
  AMLIKE_HIP(hipStreamWaitEvent(n->copy_stream, s->computed, 0));
  if (ldo == n->output_dim)
    AMLIKE_HIP(hipMemcpyAsync(out, dout, (size_t)t_out * n->output_dim * sizeof(float),
                              hipMemcpyDeviceToHost, n->copy_stream));
  else
    AMLIKE_HIP(hipMemcpy2DAsync(out, (size_t)ldo * sizeof(float), dout,
                                (size_t)n->output_dim * sizeof(float),
                                (size_t)n->output_dim * sizeof(float), t_out, hipMemcpyDeviceToHost,
                                n->copy_stream));
  AMLIKE_HIP(hipEventRecord(s->done, n->copy_stream));
  s->ticket = n->next_ticket;
  n->next_ticket = (n->next_ticket + 1) & 0x3fffffff;
  *ticket = s->ticket;
  return AMLIKE_OK;
}

int amlike_nnet_compute_wait(amlike_nnet_t *n, int32_t ticket) {
  if (int brc = BindHandle(n)) return brc;
  if (!n || ticket < 0) {
    SetError("amlike_nnet_compute_wait: invalid argument");
    return AMLIKE_EINVAL;
  }
  AsyncSlot *s = nullptr;
  {
    std::lock_guard<std::mutex> lock(n->mu);
    for (AsyncSlot &c : n->slots)
      if (c.ticket == ticket) s = &c;
  }
  if (!s) {
    SetError("amlike_nnet_compute_wait: ticket %d is not in flight", ticket);
    return AMLIKE_EINVAL;
  }
  AMLIKE_HIP(hipEventSynchronize(s->done));
  std::lock_guard<std::mutex> lock(n->mu);
  s->ticket = -1;
  return AMLIKE_OK;
}

int amlike_nnet_time_compute_dev(amlike_nnet_t *n, const float *d_feats, int32_t T, int32_t ld,
                                 int32_t pad_input, int32_t mode, float prob_scale, float *d_out,
                                 int32_t ldo, void *stream, int32_t iters, float *ms_per_pass) {
  if (int brc = BindHandle(n)) return brc;
  if (!ms_per_pass || iters <= 0) return AMLIKE_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  AMLIKE_HIP(hipEventCreate(&e0));
  AMLIKE_HIP(hipEventCreate(&e1));
  AMLIKE_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) {
    int rc = amlike_nnet_compute_dev(n, d_feats, T, ld, pad_input, mode, prob_scale, d_out, ldo, stream);
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
  *ms_per_pass = ms / iters;
  return AMLIKE_OK;
}

}  // extern "C"
