// Shared host-side helpers of libamlike (error text, HIP call checking).
#ifndef KALDI_AMD_CSRC_AMLIKE_COMMON_H_
#define KALDI_AMD_CSRC_AMLIKE_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/amlike.h"

namespace amlike {

// Per-thread last error text (what KALDI_ERR would have printed).
void SetError(const char *fmt, ...);
const char *LastError();

// The device every handle of this process is bound to (amlike_init).
int CurrentDevice();
int NumComputeUnits();

// Growable device buffer owned by a handle.  Never shrinks.
struct DeviceBuffer {
  void *ptr = nullptr;
  size_t bytes = 0;
  int Reserve(size_t want);  // AMLIKE_OK / AMLIKE_EOOM
  void Free();
};

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
