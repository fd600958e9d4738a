// How fast can a kernel write a [rows x 8600] fp32 matrix when every wave owns 64 rows and walks
// the columns 32 at a time (the store pattern of the second sweep of the fused output-layer kernel)?
// Each step a wave issues 8 stores of 1 KiB: 8 lanes x 16 bytes cover 128 bytes of one row, 8 rows
// per instruction.  Varied: waves resident per CU (through the LDS a block asks for), a delay
// between steps (s_sleep, standing in for the matrix work), rows of the matrix or consecutive
// addresses.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/store_pattern tools/micro/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

template <int SLEEP>
__global__ void __launch_bounds__(256) row_store(float *y, int ldy, int rows, int ntile, int linear) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + w) * 64;
  if (row0 + 64 > rows) return;
  const f32x4 v = {1.0f, 2.0f, 3.0f, (float)lane};
  if (threadIdx.x == 1000) lds[0] = 1;  // keeps the allocation
  for (int t = 0; t < ntile; ++t) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int r = row0 + 8 * c + (lane >> 3);
      float *p = linear ? y + ((size_t)(blockIdx.x * 4 + w) * ntile + t) * 2048 + c * 256 + lane * 4
                        : y + (size_t)r * ldy + t * 32 + 4 * (lane & 7);
      *reinterpret_cast<f32x4 *>(p) = v;
    }
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

// The same with SEG lanes (SEG x 16 bytes) per row segment: a step covers 4 SEG columns of the wave's
// 64 rows in SEG store instructions.
template <int SEG>
__global__ void __launch_bounds__(256) row_store_seg(float *y, int ldy, int rows, int ncols, int) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + w) * 64;
  if (row0 + 64 > rows) return;
  const f32x4 v = {1.0f, 2.0f, 3.0f, (float)lane};
  constexpr int RPI = 64 / SEG;  // rows per instruction
  for (int c0 = 0; c0 + 4 * SEG <= ncols; c0 += 4 * SEG) {
#pragma unroll
    for (int c = 0; c < 64 / RPI; ++c) {
      const int r = row0 + RPI * c + lane / SEG;
      *reinterpret_cast<f32x4 *>(y + (size_t)r * ldy + c0 + 4 * (lane % SEG)) = v;
    }
  }
}

// The pattern of a unit-stationary second sweep: a block's 8 waves own 8 neighbouring tiles of 32
// units (1 KiB of every row) and walk the rows of a row range 32 at a time; a step of a wave is 16
// 4-byte stores per lane, each instruction writing the 128 bytes of two rows.  The blocks of one
// XCD (block id modulo 8) take the strips of one row range after the other (XCD_AWARE), so that at
// any time the XCD's CUs write neighbouring strips of the same rows through their common L2; or
// the strips of a range are dealt over all XCDs (XCD_AWARE = 0).
// LINES = 1: every row's 128 bytes start at a 128-byte boundary (the window of units is shifted per row):
// the same volume written as whole cache lines whatever the row pitch.  LINES = 2: the same lines,
// each filled by four 16-byte stores of the two lanes that own the row (quarter lines per instruction).
template <int SLEEP, int XCD_AWARE, int LINES = 0>
__global__ void __launch_bounds__(512) strip_store(float *y, int ldy, int rows, int ntile, int ranges) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 31, h = lane >> 5;
  const int strips = (ntile + 7) / 8;
  int range, strip;
  if (XCD_AWARE) {
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
    range = xcd * (ranges / 8) + j / strips;
    strip = j % strips;
    if (j / strips >= ranges / 8) return;
  } else {
    range = blockIdx.x / strips;
    strip = blockIdx.x % strips;
    if (range >= ranges) return;
  }
  const int tile = strip * 8 + w;
  if (tile >= ntile) return;
  const int per = rows / ranges, r0 = range * per;
  const unsigned ldy4 = ldy * 4u, off = 4u * h * ldy4 + 4u * li;
  for (int r = r0; r + 32 <= r0 + per; r += 32) {
    char *base = reinterpret_cast<char *>(y + (size_t)r * ldy + tile * 32);
    if (LINES == 2) {
      // a lane owns a row: four 16-byte stores per lane, each instruction 32 bytes of every row's
      // line-aligned 128 bytes (lanes li and li + 32), the four together the whole line
      char *line = reinterpret_cast<char *>(reinterpret_cast<size_t>(base + (size_t)li * ldy4) & ~(size_t)127);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4 *>(line + (8 * j + 4 * h) * 4) = make_float4((float)j, 1.0f, 2.0f, 3.0f);
      if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
      continue;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int fr = 8 * (k / 4) + k % 4;
      char *p = base + (size_t)fr * ldy4 + off;
      if (LINES) p = reinterpret_cast<char *>((reinterpret_cast<size_t>(base + (size_t)fr * ldy4 + 4u * h * ldy4) & ~(size_t)127) + 4u * li);
      *reinterpret_cast<float *>(p) = (float)k;
    }
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

int main(int argc, char **argv) {
  const int ldy = argc > 1 ? atoi(argv[1]) : 8600, rows = 256 * 256 * 2, ntile = 268;  // 268 * 32 = 8576 <= 8600
  float *y;
  const size_t bytes = (size_t)rows * ldy * 4;
  printf("row pitch %d floats (%d bytes, %s to 128 bytes)\n", ldy, ldy * 4, ldy % 32 ? "not aligned" : "aligned");
  CHECK(hipMalloc(&y, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double written = (double)rows * ntile * 128.0;
  auto run = [&](const char *what, auto kernel, size_t smem, int linear) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(rows / 256), dim3(256), smem, 0, y, ldy, rows, ntile, linear);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("%-64s %7.3f ms  %5.2f TB/s\n", what, best, written / best / 1e9);
  };
  // LDS per block decides how many blocks (of 4 waves) a CU holds: 160 KB -> 1, 80 -> 2, 40 -> 4, 20 -> 8
  run("rows, 4 waves per CU, no delay", row_store<0>, 160 * 1024, 0);
  run("rows, 8 waves per CU, no delay", row_store<0>, 80 * 1024, 0);
  run("rows, 16 waves per CU, no delay", row_store<0>, 40 * 1024, 0);
  run("rows, 32 waves per CU, no delay", row_store<0>, 20 * 1024, 0);
  run("consecutive addresses, 4 waves per CU, no delay", row_store<0>, 160 * 1024, 1);
  run("consecutive addresses, 32 waves per CU, no delay", row_store<0>, 20 * 1024, 1);
  run("rows, 4 waves per CU, s_sleep 16 per step (~1 us)", row_store<16>, 160 * 1024, 0);
  run("rows, 4 waves per CU, s_sleep 32 per step (~2 us)", row_store<32>, 160 * 1024, 0);
  run("rows, 8 waves per CU, s_sleep 32 per step", row_store<32>, 80 * 1024, 0);
  auto run_seg = [&](const char *what, auto kernel, int seg) {
    float best = 1e9f;
    const int ncols = 8576;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(rows / 256), dim3(256), 0, 0, y, ldy, rows, ncols, 0);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double wr = (double)rows * (ncols / (4 * seg)) * (4 * seg) * 4.0;
    printf("%-64s %7.3f ms  %5.2f TB/s\n", what, best, wr / best / 1e9);
  };
  run_seg("rows, segments of 128 bytes (8 lanes)", row_store_seg<8>, 8);
  run_seg("rows, segments of 256 bytes (16 lanes)", row_store_seg<16>, 16);
  run_seg("rows, segments of 512 bytes (32 lanes)", row_store_seg<32>, 32);
  run_seg("rows, segments of 1024 bytes (64 lanes)", row_store_seg<64>, 64);
  auto run_strip = [&](const char *what, auto kernel, int ranges) {
    const int strips = (ntile + 7) / 8;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(strips * ranges), dim3(512), 0, 0, y, ldy, rows, ntile, ranges);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double wr = (double)(rows / ranges / 32 * 32) * ranges * ntile * 128.0;
    printf("%-64s %7.3f ms  %5.2f TB/s\n", what, best, wr / best / 1e9);
  };
  run_strip("strips of 8 tiles, 64 row ranges, one range per XCD at a time", strip_store<0, 1>, 64);
  run_strip("strips of 8 tiles, 64 row ranges, strips dealt over the XCDs", strip_store<0, 0>, 64);
  run_strip("strips, one range per XCD at a time, s_sleep 16 per step", strip_store<16, 1>, 64);
  run_strip("strips, one range per XCD at a time, s_sleep 24 per step", strip_store<24, 1>, 64);
  run_strip("strips dealt over the XCDs, s_sleep 24 per step", strip_store<24, 0>, 64);
  run_strip("strips, 16 row ranges, one range per XCD at a time", strip_store<0, 1>, 16);
  run_strip("strips dealt over the XCDs, every row's 128 bytes line-aligned", strip_store<0, 0, 1>, 64);
  run_strip("the same, s_sleep 24 per step", strip_store<24, 0, 1>, 64);
  run_strip("line-aligned, a lane per row: 4 x 16-byte stores fill a line", strip_store<0, 0, 2>, 64);
  run_strip("the same, s_sleep 24 per step", strip_store<24, 0, 2>, 64);
  CHECK(hipFree(y));
  return 0;
}
