// Process-level state of libamlike: device binding, error text, buffers.
#include "amlike_common.h"

#include <cstring>

namespace amlike {

static thread_local char g_err[512] = "";
static int g_device = -1;
static int g_num_cu = 0;

void SetError(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char *LastError() { return g_err; }
int CurrentDevice() { return g_device; }
int NumComputeUnits() { return g_num_cu; }

int BindThread() {
  if (g_device < 0) return AMLIKE_OK;
  int cur = -1;
  AMLIKE_HIP(hipGetDevice(&cur));
  if (cur != g_device) AMLIKE_HIP(hipSetDevice(g_device));
  return AMLIKE_OK;
}

int DeviceBuffer::Reserve(size_t want) {
  if (want <= bytes) return AMLIKE_OK;
  if (ptr) {
    AMLIKE_HIP(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
  }
  size_t grow = want + want / 4;  // amortise regrowth across calls of rising size
  hipError_t e = hipMalloc(&ptr, grow);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&ptr, want);
    grow = want;
  }
  if (e != hipSuccess) {
    ptr = nullptr;
    SetError("hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
    return AMLIKE_EOOM;
  }
  bytes = grow;
  return AMLIKE_OK;
}

void DeviceBuffer::Free() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
}

int AsyncSlot::Prepare() {
  if (!computed) AMLIKE_HIP(hipEventCreateWithFlags(&computed, hipEventDisableTiming));
  if (!done) AMLIKE_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  if (!h_flag) {
    AMLIKE_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_flag), sizeof(int), hipHostMallocDefault));
    *h_flag = 0;
  }
  return AMLIKE_OK;
}

void AsyncSlot::Release() {
  if (ticket >= 0 && done) (void)hipEventSynchronize(done);
  if (computed) (void)hipEventDestroy(computed);
  if (done) (void)hipEventDestroy(done);
  if (h_flag) (void)hipHostFree(h_flag);
  computed = done = nullptr;
  h_flag = nullptr;
  in.Free();
  out.Free();
  ticket = -1;
}

}  // namespace amlike

extern "C" {

int amlike_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int amlike_init(int device) {
  int n = amlike_device_count();
  if (n <= 0 || device < 0 || device >= n) {
    amlike::SetError("amlike_init: device %d not available (%d visible)", device, n);
    return AMLIKE_ENODEV;
  }
  AMLIKE_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  AMLIKE_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    amlike::SetError("amlike_init: device %d is %s, this library is built for gfx950 only",
                     device, prop.gcnArchName);
    return AMLIKE_ENODEV;
  }
  amlike::g_device = device;
  amlike::g_num_cu = prop.multiProcessorCount;
  return AMLIKE_OK;
}

const char *amlike_last_error(void) { return amlike::LastError(); }

int amlike_host_alloc(size_t bytes, void **ptr) {
  if (!ptr || bytes == 0) {
    amlike::SetError("amlike_host_alloc: invalid argument");
    return AMLIKE_EINVAL;
  }
  *ptr = nullptr;
  if (int brc = amlike::BindThread()) return brc;
  hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *ptr = nullptr;
    amlike::SetError("hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return AMLIKE_EOOM;
  }
  return AMLIKE_OK;
}

int amlike_host_free(void *ptr) {
  if (ptr) AMLIKE_HIP(hipHostFree(ptr));
  return AMLIKE_OK;
}

}  // extern "C"
