// Peak micro-kernels for the roofline denominators, measured on the box the benchmark runs on:
//   1. v_mfma_f32_32x32x16_f16, dense, four independent accumulators per wave, 1 / 2 / 4 waves per
//      SIMD on every CU: TFLOP/s and the clock it implies (32 cycles per instruction per SIMD);
//   2. HBM: streaming write, read and copy of a buffer far larger than the caches, 16 bytes per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/peaks tools/micro/peaks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

// RANDOM = false: the same small operands in every instruction (little switching activity);
// RANDOM = true: four different pseudo-random operand pairs of mixed sign and magnitude per wave,
// so that consecutive instructions toggle the multiplier inputs the way real data does (the
// clock the chip sustains depends on it).
template <int THREADS, bool RANDOM>
__global__ void __launch_bounds__(THREADS) mfma_peak(float *out, int iters, float seed) {
  f16x8 a[4], b[4];
  unsigned r = 2654435761u * (blockIdx.x * THREADS + threadIdx.x + 1);
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 8; ++i) {
      if (RANDOM) {
        r = r * 1664525u + 1013904223u;
        a[q][i] = (_Float16)(((int)(r >> 8 & 0xffff) - 32768) * (1.0f / 16384.0f));
        r = r * 1664525u + 1013904223u;
        b[q][i] = (_Float16)(((int)(r >> 8 & 0xffff) - 32768) * (1.0f / 16384.0f));
      } else {
        a[q][i] = (_Float16)(seed * (threadIdx.x + i));
        b[q][i] = (_Float16)(seed * (i + 1));
      }
    }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[2], c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[3], c3, 0, 0, 0);
  }
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) hbm_write(f32x4 *dst, size_t n, float v) {
  const f32x4 x = {v, v, v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = x;
}

__global__ void __launch_bounds__(256) hbm_read(const f32x4 *src, size_t n, float *out) {
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;  // keeps the loads alive
}

__global__ void __launch_bounds__(256) hbm_copy(const f32x4 *src, f32x4 *dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
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
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return ms / reps;
}

template <int THREADS, bool RANDOM>
static void run_mfma(float *d_out, int cus, int iters) {
  const double ms = time_ms([&] { hipLaunchKernelGGL((mfma_peak<THREADS, RANDOM>), dim3(cus), dim3(THREADS), 0, 0, d_out, iters, 0.001f); }, 5);
  const double n_mfma = (double)cus * (THREADS / 64) * iters * 4.0;
  const double tflops = n_mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  // every SIMD issues (THREADS/256) waves' instructions back to back, 32 cycles each at full rate
  const double ghz = (double)(THREADS / 256) * iters * 4.0 * 32.0 / (ms * 1e-3) / 1e9;
  printf("mfma_f32_32x32x16_f16  %s operands, %d waves/SIMD on %d CUs: %8.1f TFLOP/s  (%.2f ms per launch; %.2f GHz if 32 cycles per instruction)\n",
         RANDOM ? "random  " : "constant", THREADS / 256, cus, tflops, ms, ghz);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs, clockRate %.2f GHz\n", prop.gcnArchName, cus, prop.clockRate / 1e6);
  float *d_out;
  CHECK(hipMalloc(&d_out, (size_t)cus * 1024 * sizeof(float)));
  // short launches (about 1-5 ms) and launches as long as the benchmark's kernel (20+ ms)
  for (int iters : {20000, 150000}) {
    run_mfma<256, false>(d_out, cus, iters);
    run_mfma<512, false>(d_out, cus, iters);
    run_mfma<1024, false>(d_out, cus, iters);
    run_mfma<512, true>(d_out, cus, iters);
    run_mfma<1024, true>(d_out, cus, iters);
  }

  const size_t bytes = (size_t)8 << 30, n = bytes / 16;
  f32x4 *a, *b;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 0, bytes));
  CHECK(hipMemset(b, 0, bytes));
  for (int blocks : {cus * 8, cus * 32}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(hbm_write, dim3(blocks), dim3(256), 0, 0, a, n, 1.0f); }, 5);
    printf("HBM write 8 GiB, %5d blocks: %6.2f TB/s (%.2f ms)\n", blocks, bytes / (ms * 1e-3) / 1e12, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(hbm_read, dim3(blocks), dim3(256), 0, 0, a, n, d_out); }, 5);
    printf("HBM read  8 GiB, %5d blocks: %6.2f TB/s (%.2f ms)\n", blocks, bytes / (ms * 1e-3) / 1e12, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(hbm_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 5);
    printf("HBM copy  8 GiB, %5d blocks: %6.2f TB/s read+write (%.2f ms)\n", blocks, 2.0 * bytes / (ms * 1e-3) / 1e12, ms);
  }
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  CHECK(hipFree(d_out));
  return 0;
}
