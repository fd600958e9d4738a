// Issue cost of the instructions the scoring kernels are made of, in shader cycles per instruction
// (s_memtime around an unrolled run of independent instructions), with 1 and with 2 waves resident
// on a SIMD: per wave, and for the SIMD (both waves' work from the first start to the last end).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rates tools/micro/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8 independent chains, 8 instructions per REP
#define REP8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int WHAT>
__global__ void __launch_bounds__(512) rate(unsigned long long *out, float seed, int iters) {
  float v[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) {
    v[i] = seed * (threadIdx.x + i + 1);
    p[i] = f32x2{v[i], v[i] + 1.0f};
  }
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(seed * (i + 1));
    b[i] = (_Float16)(seed * (threadIdx.x & 7));
  }
  f32x16 c0 = {0}, c1 = {0}, c2;
  for (int i = 0; i < 16; ++i) c2[i] = seed * i;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 ld4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  __shared__ float lds[512 * 8 + 64 + 2048];
  const unsigned lds_addr = (unsigned)(size_t)((__attribute__((address_space(3))) float *)lds) + threadIdx.x * 4;
  // store target: lane l of wave w of block b writes row (b * 8 + w) * 64 + l, rows 34400 bytes apart
  unsigned long long gaddr = (unsigned long long)(out + 4096) +
                             (unsigned long long)((blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63)) * 34400ull;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (WHAT == 0) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if (WHAT == 1) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if (WHAT == 2) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if (WHAT == 3) {
#define S(i) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if (WHAT == 4) {
#define S(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if (WHAT == 5) {  // matrix instructions alone, two accumulators
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
      }
    } else if (WHAT == 6) {  // each matrix instruction followed by one v_exp_f32 and one v_fma_f32
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[k & 7]), "+v"(v[(k + 3) & 7]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[(k + 1) & 7]), "+v"(v[(k + 4) & 7]));
      }
    } else if (WHAT == 7) {  // ... by two of each
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                     : "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      }
    } else if (WHAT == 8) {  // each matrix instruction followed by six v_fma_f32
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#define S(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
        S(0) S(1) S(2) S(3) S(4) S(5)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
        S(6) S(7) S(0) S(1) S(2) S(3)
#undef S
      }
    } else if (WHAT == 9) {  // register read by index through the index mode (on, move, off)
      const int idx = it & 3;
#define S(i) asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, %0\n\ts_set_gpr_idx_off" : "+v"(v[i]) : "s"(idx));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if (WHAT == 10) {  // each matrix instruction followed by two indexed reads
      const int idx = it & 3;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
#define S(i) asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, %0\n\ts_set_gpr_idx_off" : "+v"(v[i]) : "s"(idx));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
        S(0) S(1)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
        S(2) S(3)
#undef S
      }
    } else if (WHAT == 11) {  // LDS write of one register per lane
#define S(i) asm volatile("ds_write_b32 %1, %0 offset:%2" : : "v"(v[i]), "v"(lds_addr), "n"(i * 256) : "memory");
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (WHAT == 12) {  // each matrix instruction followed by 1 exp, 1 fma and one LDS write
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tds_write_b32 %2, %1" : "+v"(v[k & 7]), "+v"(v[(k + 3) & 7]) : "v"(lds_addr) : "memory");
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tds_write_b32 %2, %1 offset:256" : "+v"(v[(k + 1) & 7]), "+v"(v[(k + 4) & 7]) : "v"(lds_addr) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (WHAT == 14 || WHAT == 15 || WHAT == 16) {
      // waves 0-3 issue matrix instructions only, waves 4-7 (their SIMD partners) vector work only
      if ((threadIdx.x >> 6) < 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
        }
      } else if (WHAT == 14) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
        REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
      } else if (WHAT == 15) {
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
      } else {
        const int idx = it & 3;
#define S(i) asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\tv_mov_b32 %0, %0\n\ts_set_gpr_idx_off" : "+v"(v[i]) : "s"(idx));
        REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
      }
    } else if (WHAT == 17) {  // as WHAT == 6 with the accumulators in vector registers
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[k & 7]), "+v"(v[(k + 3) & 7]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[(k + 1) & 7]), "+v"(v[(k + 4) & 7]));
      }
    } else if (WHAT == 18) {  // as WHAT == 17, and the exponentials read registers of a third accumulator set
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %2\n\tv_fma_f32 %1, %1, %1, %0" : "=&v"(v[k & 7]), "+v"(v[(k + 3) & 7]) : "v"(c2[k]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %2\n\tv_fma_f32 %1, %1, %1, %0" : "=&v"(v[(k + 1) & 7]), "+v"(v[(k + 4) & 7]) : "v"(c2[(k + 8) & 15]));
      }
    } else if (WHAT == 19) {  // as WHAT == 17 with a 16-byte LDS read per three matrix instructions
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[k & 7]), "+v"(v[(k + 3) & 7]));
        if (k % 3 == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld4[0]) : "v"(lds_addr * 4) : "memory");
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[(k + 1) & 7]), "+v"(v[(k + 4) & 7]));
        if (k % 3 == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(ld4[1]) : "v"(lds_addr * 4) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (WHAT == 13) {  // scattered 4-byte stores: 64 lanes, 64 rows 34400 bytes apart
#define S(i) asm volatile("global_store_dword %0, %1, off offset:%2" : : "v"(gaddr), "v"(v[i]), "n"(i * 4) : "memory");
      REP8(S)
#undef S
      gaddr += 32;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i];
  s += ld4[0][0] + ld4[1][1];
  if (s == 12345.678f) out[1000 + threadIdx.x] = 1;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
    out[2 * (threadIdx.x >> 6)] = t0;
    out[2 * (threadIdx.x >> 6) + 1] = t1;
  }
}

int main() {
  unsigned long long *d, h[16];
  CHECK(hipMalloc(&d, 4096 * sizeof(unsigned long long) + (size_t)256 * 8 * 64 * 34400 + (1 << 20)));
  const int iters = 2000;
  auto run = [&](const char *what, auto kernel, int per_iter) {
    for (int threads : {256, 512}) {  // 1 or 2 waves per SIMD
      hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), 0, 0, d, 1e-3f, iters);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      // waves w and w + 4 share SIMD w: wave 0 alone, wave 4 alone, and both from first start to last end
      const int nw = threads / 64;
      unsigned long long lo = h[0], hi = h[1];
      if (nw == 8) {
        lo = h[0] < h[8] ? h[0] : h[8];
        hi = h[1] > h[9] ? h[1] : h[9];
      }
      printf("%-50s %d wave(s)/SIMD: wave 0 %6.2f", what, threads / 256, (double)(h[1] - h[0]) / iters / per_iter);
      if (nw == 8) printf("  wave 4 %6.2f  SIMD %6.2f per instruction of either wave", (double)(h[9] - h[8]) / iters / per_iter,
                          (double)(hi - lo) / iters / per_iter / 2);
      printf("\n");
    }
  };
  run("v_exp_f32", rate<0>, 32);
  run("v_fma_f32", rate<1>, 32);
  run("v_pk_fma_f32", rate<2>, 32);
  run("v_log_f32", rate<3>, 32);
  run("v_permlane32_swap_b32", rate<4>, 32);
  run("v_mfma_f32_32x32x16_f16 (per matrix instruction)", rate<5>, 32);
  run("mfma + 1 exp + 1 fma (per matrix instruction)", rate<6>, 32);
  run("mfma + 2 exp + 2 fma (per matrix instruction)", rate<7>, 32);
  run("mfma + 6 fma (per matrix instruction)", rate<8>, 32);
  run("indexed register read (idx_on, v_mov, idx_off)", rate<9>, 32);
  run("mfma + 2 indexed reads (per matrix instruction)", rate<10>, 32);
  run("ds_write_b32", rate<11>, 32);
  run("mfma + 1 exp + 1 fma + 1 ds_write_b32 (per mfma)", rate<12>, 32);
  run("global_store_dword, 64 rows per instruction", rate<13>, 8);
  run("mfma (acc in VGPRs) + 1 exp + 1 fma", rate<17>, 32);
  run("... exp reading a third accumulator set", rate<18>, 32);
  run("... + one ds_read_b128 per 3 mfma", rate<19>, 32);
  run("wave 0: mfma only; wave 4: v_fma_f32 only", rate<14>, 32);
  run("wave 0: mfma only; wave 4: v_exp_f32 only", rate<15>, 32);
  run("wave 0: mfma only; wave 4: indexed reads only", rate<16>, 32);
  printf("(s_memtime ticks; the counter runs at a fixed 100 MHz-derived rate on some parts: compare the rows, and the\n"
         " matrix instruction's known 32 shader cycles (8 passes), to convert)\n");
  CHECK(hipFree(d));
  return 0;
}
