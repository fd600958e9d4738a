// Microbenchmark: how VALU work interleaves with fp32 and bf16 MFMA on gfx950.
// Each wave runs ITER x { G MFMAs (one dependent chain) then G*NV independent VALU ops }.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int G, int NV, int THREADS>
__global__ void __launch_bounds__(THREADS) k(float *out, int iters, float a0, float b0) {
  f32x16 acc = {0};
  float a = a0 + threadIdx.x, b = b0;
  bf16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (short)(threadIdx.x + i); bh[i] = (short)(i * 3); }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a0 * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NV * G; ++i)
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(b));
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int G, int NV, int THREADS>
void run() {
  float *d; (void)hipMalloc(&d, 1024 * 1024 * 4);
  int iters = 20000 / G;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, G, NV, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, 100, 1.0f, 2.0f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, G, NV, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, iters, 1.0f, 2.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double per = ms * 1e6 / (iters * G);
  printf("%s G=%2d NV=%2d waves/SIMD=%d : %6.1f ns per MFMA (+NV valu) = %4.0f cyc@2.4GHz\n",
         KIND ? "bf16 32x32x16" : "f32  32x32x2 ", G, NV, THREADS / 256, per, per * 2.4);
  (void)hipFree(d);
}

int main() {
  run<0, 1, 0, 256>(); run<0, 1, 4, 256>(); run<0, 1, 8, 256>(); run<0, 4, 4, 256>(); run<0, 4, 8, 256>();
  run<0, 8, 4, 256>(); run<0, 8, 8, 256>(); run<0, 40, 4, 256>();
  run<0, 1, 0, 512>(); run<0, 1, 4, 512>(); run<0, 1, 8, 512>(); run<0, 8, 4, 512>(); run<0, 8, 8, 512>(); run<0, 40, 4, 512>();
  run<1, 1, 0, 256>(); run<1, 1, 2, 256>(); run<1, 1, 4, 256>(); run<1, 1, 6, 256>(); run<1, 1, 8, 256>(); run<1, 4, 4, 256>(); run<1, 4, 8, 256>();
  run<1, 1, 0, 512>(); run<1, 1, 2, 512>(); run<1, 1, 4, 512>(); run<1, 1, 6, 512>(); run<1, 1, 8, 512>(); run<1, 4, 4, 512>(); run<1, 4, 8, 512>();
  return 0;
}
