// Which fp16 matrix-instruction shape suits the GMM contraction (K = 80 per partial product, three
// partial products, a 64-Gaussian x 32-frame output tile per wave) when the chip is power-limited?
// Every arrangement computes the same output tile from pseudo-random operands held in registers:
//   A  32x32x16: 2 tiles x 5 k-steps x 3 products               = 30 instructions of 32 cycles
//   B  16x16x32, K padded 80 -> 96 per product: 8 tiles x 3 x 3  = 72 instructions of 16 cycles
//   C  16x16x32, the three products laid end to end along K (240 -> 256): 8 tiles x 8 = 64
//   D  16x16x32 twice + 16x16x16 once per product (K = 80 exactly): 8 tiles x 3 x 3  = 72
//   E  32x32x16 twice + 32x32x8 once per 40 of K ... (the older half-depth shape): 2 x 3 x (4 + 2)
// and prints time per tile, the useful TFLOP/s (2 * 81 * 64 * 32 FLOP per tile, as the benchmark
// counts them) and the clock implied by the instruction cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_shapes tools/micro/mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__device__ __forceinline__ f16x8 rnd8(unsigned &r) {
  f16x8 v;
  for (int i = 0; i < 8; ++i) {
    r = r * 1664525u + 1013904223u;
    v[i] = (_Float16)(((int)(r >> 8 & 0xffff) - 32768) * (1.0f / 16384.0f));
  }
  return v;
}

template <int ARR, int THREADS>
__global__ void __launch_bounds__(THREADS) shapes(float *out, int iters) {
  unsigned r = 2654435761u * (blockIdx.x * THREADS + threadIdx.x + 1);
  // model operands (stationary): up to 16 fragments hi + 16 lo; feature operands: 8 hi + 8 lo
  f16x8 ah[10], al[10], bh[5], bl[5];
  for (int i = 0; i < 10; ++i) { ah[i] = rnd8(r); al[i] = rnd8(r) * (_Float16)0.001f; }
  for (int i = 0; i < 5; ++i) { bh[i] = rnd8(r); bl[i] = rnd8(r) * (_Float16)0.001f; }
  float s = 0.0f;
  if (ARR == 0) {
    f32x16 c[2] = {{0}, {0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t * 5 + k], bh[k], c[t], 0, 0, 0);
          c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t * 5 + k], bl[k], c[t], 0, 0, 0);
          c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t * 5 + k], bh[k], c[t], 0, 0, 0);
        }
      asm volatile("" : "+v"(bh[0]), "+v"(bl[0]));
    }
    for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i];
  } else if (ARR == 1 || ARR == 2) {
    // 16x16x32: 8 tiles of 16 Gaussians x 16 frames... (4 Gaussian tiles x 2 frame halves)
    constexpr int KS = ARR == 1 ? 9 : 8;  // instructions per tile
    f32x4 c[8];
    for (int t = 0; t < 8; ++t) c[t] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const f16x8 a = (k % 3 == 0) ? al[(t / 2 * 3 + k) % 10] : ah[(t / 2 * 3 + k) % 10];
          const f16x8 b = (k % 3 == 1) ? bl[(k + t % 2) % 5] : bh[(k + t % 2) % 5];
          c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[t], 0, 0, 0);
        }
      asm volatile("" : "+v"(bh[0]), "+v"(bl[0]));
    }
    for (int t = 0; t < 8; ++t) s += c[t][0] + c[t][1] + c[t][2] + c[t][3];
  } else if (ARR == 3) {
    f32x4 c[8];
    for (int t = 0; t < 8; ++t) c[t] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const f16x8 a0 = p == 0 ? al[t] : ah[t], a1 = p == 0 ? al[(t + 1) % 10] : ah[(t + 1) % 10];
          const f16x8 b0 = p == 1 ? bl[t % 2] : bh[t % 2], b1 = p == 1 ? bl[2 + t % 2] : bh[2 + t % 2];
          c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c[t], 0, 0, 0);
          c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[t], 0, 0, 0);
          const f16x8 a2 = p == 0 ? al[(t + 2) % 10] : ah[(t + 2) % 10];
          const f16x8 b2 = p == 1 ? bl[4] : bh[4];
          c[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(a2, a2, 0, 1, 2, 3),
                                                       __builtin_shufflevector(b2, b2, 0, 1, 2, 3), c[t], 0, 0, 0);
        }
      asm volatile("" : "+v"(bh[0]), "+v"(bl[0]));
    }
    for (int t = 0; t < 8; ++t) s += c[t][0] + c[t][1] + c[t][2] + c[t][3];
  } else {
    // 32x32x8 (the half-depth shape) only: 2 tiles x 10 k-steps x 3 products
    f32x16 c[2] = {{0}, {0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f16x4 a_l = __builtin_shufflevector(al[t * 5 + k / 2], al[t * 5 + k / 2], 0, 1, 2, 3);
          const f16x4 a_h = __builtin_shufflevector(ah[t * 5 + k / 2], ah[t * 5 + k / 2], 4, 5, 6, 7);
          const f16x4 b_h = __builtin_shufflevector(bh[k / 2], bh[k / 2], 0, 1, 2, 3);
          const f16x4 b_l = __builtin_shufflevector(bl[k / 2], bl[k / 2], 4, 5, 6, 7);
          c[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a_l, b_h, c[t], 0, 0, 0);
          c[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a_h, b_l, c[t], 0, 0, 0);
          c[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(a_h, b_h, c[t], 0, 0, 0);
        }
      asm volatile("" : "+v"(bh[0]), "+v"(bl[0]));
    }
    for (int i = 0; i < 16; ++i) s += c[0][i] + c[1][i];
  }
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.0f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int ARR, int THREADS>
static void run(float *d_out, int cus, int iters, const char *name, int n_instr, int cyc_per_instr) {
  const double ms = time_ms([&] { hipLaunchKernelGGL((shapes<ARR, THREADS>), dim3(cus), dim3(THREADS), 0, 0, d_out, iters); }, 4);
  const double tiles = (double)cus * (THREADS / 64) * iters;
  const double useful = tiles * 2.0 * 81 * 64 * 32 / (ms * 1e-3) / 1e12;
  const double ns_per_tile = ms * 1e6 / ((double)iters * (THREADS / 256));  // per SIMD
  const double ghz = (double)n_instr * cyc_per_instr / ns_per_tile;
  printf("%-44s %d waves/SIMD: %7.1f ns per tile per SIMD, %6.1f useful TFLOP/s, %.2f GHz if %d cycles x %d\n", name,
         THREADS / 256, ns_per_tile, useful, ghz, cyc_per_instr, n_instr);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *d_out;
  CHECK(hipMalloc(&d_out, (size_t)cus * 1024 * sizeof(float)));
  const int iters = 40000;  // ~20-40 ms per launch: long enough for the clock to settle
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 256>(d_out, cus, iters, "A 32x32x16 x30 (K=80)", 30, 32);
    run<0, 512>(d_out, cus, iters, "A 32x32x16 x30 (K=80)", 30, 32);
    run<1, 256>(d_out, cus, iters, "B 16x16x32 x72 (K padded to 96)", 72, 16);
    run<1, 512>(d_out, cus, iters, "B 16x16x32 x72 (K padded to 96)", 72, 16);
    run<2, 256>(d_out, cus, iters, "C 16x16x32 x64 (products end to end, K=256)", 64, 16);
    run<2, 512>(d_out, cus, iters, "C 16x16x32 x64 (products end to end, K=256)", 64, 16);
    run<3, 256>(d_out, cus, iters, "D 16x16x32 x48 + 16x16x16 x24 (K=80)", 72, 16);
    run<3, 512>(d_out, cus, iters, "D 16x16x32 x48 + 16x16x16 x24 (K=80)", 72, 16);
    run<4, 256>(d_out, cus, iters, "E 32x32x8 x60 (K=80)", 60, 16);
    run<4, 512>(d_out, cus, iters, "E 32x32x8 x60 (K=80)", 60, 16);
  }
  CHECK(hipFree(d_out));
  return 0;
}
