// Microbenchmark: a model of the GMM scoring kernel's per-tile instruction stream, to compare
// two waves per SIMD holding 2 accumulator tiles each against one wave per SIMD holding 4.
// Per tile a wave issues NT*15 matrix instructions (NT independent chains), each followed by one
// v_exp_f32 and one v_fma_f32, then a tail of TAILV vector + TAILS scalar instructions with no
// matrix work (the segment loop and stores of the real kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NT, int THREADS, int TAILV, int TAILS, bool EXP, int PHASE = 0>
__global__ void __launch_bounds__(THREADS, 1) k(float *out, int tiles, float a0) {
  f32x16 acc[NT];
  f16x8 a[NT], b;
  for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.001f * i);
  for (int n = 0; n < NT; ++n) {
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;
    for (int i = 0; i < 8; ++i) a[n][i] = (_Float16)(0.002f * (i + n));
  }
  float s = a0, e = 0.5f;
  int sc = 0;
  // PHASE: the second wave of every SIMD starts PHASE x 64 cycles late, so that its tail falls
  // into the other wave's matrix-instruction stretch
  if (PHASE > 0 && (threadIdx.x >> 8))
    for (int i = 0; i < PHASE; ++i) __builtin_amdgcn_s_sleep(1);
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int m = 0; m < 15; ++m) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[n], b, acc[n], 0, 0, 0);
#if __HIP_DEVICE_COMPILE__
        asm volatile("" : "+v"(acc[n]));
        if (EXP) {
          asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(s));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(a0), "v"(e));
        }
#endif
      }
    }
#if __HIP_DEVICE_COMPILE__
#pragma unroll
    for (int i = 0; i < TAILV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s) : "v"(a0));
#endif
#if __HIP_DEVICE_COMPILE__  // the host pass cannot parse scalar-register constraints
#pragma unroll
    for (int i = 0; i < TAILS; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) : : "scc");
#endif
  }
  float r = s + sc;
  for (int n = 0; n < NT; ++n)
    for (int i = 0; i < 16; ++i) r += acc[n][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NT, int THREADS, int TAILV, int TAILS, bool EXP, int PHASE = 0>
void run(const char *what) {
  float *d;
  (void)hipMalloc(&d, 1024 * 1024 * 4);
  const int tiles = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NT, THREADS, TAILV, TAILS, EXP, PHASE>), dim3(256), dim3(THREADS), 0, 0, d, 50, 1.0e-3f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NT, THREADS, TAILV, TAILS, EXP, PHASE>), dim3(256), dim3(THREADS), 0, 0, d, tiles, 1.0e-3f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) printf("launch failed\n");
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)tiles * NT * 15 * (THREADS / 256);
  printf("%-44s NT=%d waves/SIMD=%d tail=%3dv+%3ds exp=%d : %6.2f ns per MFMA per SIMD\n", what, NT,
         THREADS / 256, TAILV, TAILS, (int)EXP, ms * 1e6 / mfma_per_simd);
  (void)hipFree(d);
}

int main() {
  run<2, 512, 0, 0, false>("two waves x 2 tiles, MFMA only");
  run<4, 256, 0, 0, false>("one wave x 4 tiles, MFMA only");
  run<2, 512, 0, 0, true>("two waves x 2 tiles, + exp/fma");
  run<4, 256, 0, 0, true>("one wave x 4 tiles, + exp/fma");
  run<2, 512, 60, 60, true>("two waves x 2 tiles, + exp/fma + tail");
  run<4, 256, 90, 90, true>("one wave x 4 tiles, + exp/fma + tail");
  run<2, 512, 120, 120, true>("two waves x 2 tiles, + exp/fma + long tail");
  run<4, 256, 180, 180, true>("one wave x 4 tiles, + exp/fma + long tail");
  run<2, 512, 120, 120, true, 8>("two waves, long tail, second wave 512 cycles late");
  run<2, 512, 120, 120, true, 16>("two waves, long tail, second wave 1024 cycles late");
  run<2, 512, 120, 120, true, 24>("two waves, long tail, second wave 1536 cycles late");
  run<2, 512, 60, 60, true, 16>("two waves, tail, second wave 1024 cycles late");
  return 0;
}
