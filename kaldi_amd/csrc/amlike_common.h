// Shared host-side helpers of libamlike (error text, HIP call checking).
#ifndef KALDI_AMD_CSRC_AMLIKE_COMMON_H_
#define KALDI_AMD_CSRC_AMLIKE_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/amlike.h"

namespace amlike {

// Per-thread last error text (what KALDI_ERR would have printed).
void SetError(const char *fmt, ...);
const char *LastError();

// Devices.  amlike_init(d) may be called for several devices (from one thread or from one
// thread per device).  A handle lives on the device that was current for its creating thread:
// the device of that thread's latest amlike_init, else of the process's latest.  Every entry
// point that takes a handle makes the handle's device current on the calling thread first
// (HIP's current device is per thread and starts at 0), so handles of different devices can
// be driven from any threads.
int CurrentDevice();     // the calling thread's device as described above, -1 before any amlike_init
int NumComputeUnits();   // of the device current in HIP on the calling thread
int BindDevice(int device);  // makes `device` current in HIP (no-op for -1)
int BindThread();            // BindDevice(CurrentDevice())
// BindDevice(the handle's device), or BindThread() for a null handle (the caller reports it)
template <typename H>
int BindHandle(const H *h) {
  return h ? BindDevice(h->device) : BindThread();
}
// Looks up, once per device, the address of the built-in scratch array of the entry point that
// allocates nothing (gmm_score.hip); called by amlike_init with `device` current.
int SetupNoAllocScratch(int device);

// Control words in device memory (overflow counts, "out of range" and "not finite" flags): a
// kernel raises them, later kernels of the same call decide by them what to run, the host reads
// them back.  The chip's compute dies each have an L2 of their own, agent-scope atomics carry no
// cache-scope bit, and an ordinary load may be served from a die's copy of the line taken before
// another die's atomic changed the word.  These few accesses therefore go at system scope, past
// the L2s, for readers and writers alike.
#if defined(__HIPCC__)
__device__ __forceinline__ int ctl_load(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ctl_store(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int ctl_add(int *p, int v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ctl_or(int *p, int v) {
  (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
// Zeroes n control words by a kernel on `st` (the same kind of store as above; a fill
// by the copy engine or a fill node of a graph is not seen by every later atomic).
int ClearWords(int *d_words, int n, hipStream_t st);

// Growable device buffer owned by a handle.  Never shrinks.
struct DeviceBuffer {
  void *ptr = nullptr;
  size_t bytes = 0;
  int Reserve(size_t want);  // AMLIKE_OK / AMLIKE_EOOM
  void Free();
};

// Copies a dense device matrix [rows x cols] (row stride cols) to host rows of stride ldo,
// enqueued on `st` behind the kernels that produce it, and waits for it.  Page-locked
// destinations get one asynchronous copy.  Ordinary (pageable) destinations -- what a caller
// holding a plain malloc'ed matrix has -- go through page-locked staging buffers in 32 MB
// chunks, the link copy of chunk i+1 and i+2 overlapping a multi-threaded host copy of chunk i:
// about twice the rate of the runtime's own pageable path for results of a gigabyte.
int CopyMatrixToHost(float *out, size_t ldo, const float *d_src, int rows, int cols, hipStream_t st);

// Small host-to-device uploads of tables the caller's next call rewrites (row maps, offsets): the
// bytes are first copied into one of a few page-locked buffers, each guarded by an event, so the
// source may change as soon as Upload returns and the copy itself is truly asynchronous.
class PinnedUploadRing {
 public:
  ~PinnedUploadRing();
  int Upload(void *d_dst, const void *src, size_t bytes, hipStream_t st);

 private:
  static constexpr int kSlots = 4;
  struct Slot {
    void *buf = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    bool used = false;
  } slots_[kSlots];
  int next_ = 0;
};

// One submission of the asynchronous host-pointer entry points (*_submit / *_wait): its own
// device input and output buffers, the event the output copy waits for, the event the
// caller waits for, and a pinned word the non-finite flag lands in.
struct AsyncSlot {
  hipEvent_t computed = nullptr, done = nullptr;
  int *h_flag = nullptr;
  DeviceBuffer in, out;
  int32_t ticket = -1;  // -1: free
  int Prepare();        // creates the events and the flag word on first use
  void Release();
};
constexpr int kAsyncSlots = 2;

// Y[M x N] = X W^T + bias on the bf16x3 matrix-instruction GEMM of the nnet2 path
// (nnet_forward.hip).  Operands are three bf16 planes (x = p1 + p2 + p3 exactly) of 16-byte
// elements holding 8 consecutive k of one row, k-octet-major: element (plane q, octet o, row r)
// of X sits at q * xplane + o * M + r; of W at q * wplane + o * woct + r, so that W may be a
// row range of a larger image.  Kp = padded k, a multiple of 16.
int LaunchPlaneGemmBf16x3(const void *xp, size_t xplane, const void *wp, size_t wplane, size_t woct,
                          int Kp, const float *bias, float *y, int ldy, int M, int N, hipStream_t st);

// fp32 -> bf16, round to nearest even (inf / nan: truncated).
static inline uint16_t Bf16Rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float Bf16ToFloat(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}
// x = p[0] + p[1] + p[2] exactly (24 mantissa bits over three bf16).
static inline void SplitBf16x3(float x, uint16_t p[3]) {
  p[0] = Bf16Rne(x);
  float r = x - Bf16ToFloat(p[0]);
  p[1] = Bf16Rne(r);
  r -= Bf16ToFloat(p[1]);
  p[2] = Bf16Rne(r);
}

}  // namespace amlike

// CU_SAFE_CALL analogue (/root/reference/src/cudamatrix/cu-common.h:38-45)
// without the device synchronisation the reference adds after every call.
#define AMLIKE_HIP(call)                                                          \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      amlike::SetError("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                       __FILE__, __LINE__);                                       \
      return (e_ == hipErrorOutOfMemory) ? AMLIKE_EOOM : AMLIKE_EHIP;             \
    }                                                                             \
  } while (0)

#endif  // KALDI_AMD_CSRC_AMLIKE_COMMON_H_
