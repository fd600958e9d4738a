// Process-level state of libamlike: device binding, error text, buffers.
#include "amlike_common.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace amlike {

static thread_local char g_err[512] = "";
constexpr int kMaxDevices = 64;
static std::atomic<int> g_default_device{-1};   // the device of the latest amlike_init of the process
static thread_local int t_device = -1;          // the device of the calling thread's latest amlike_init
static int g_num_cu[kMaxDevices] = {0};

void SetError(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char *LastError() { return g_err; }
int CurrentDevice() { return t_device >= 0 ? t_device : g_default_device.load(); }
int NumComputeUnits() {
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur < 0 || cur >= kMaxDevices) return 0;
  return g_num_cu[cur];
}

int BindDevice(int device) {
  if (device < 0) return AMLIKE_OK;
  int cur = -1;
  AMLIKE_HIP(hipGetDevice(&cur));
  if (cur != device) AMLIKE_HIP(hipSetDevice(device));
  return AMLIKE_OK;
}

int BindThread() { return BindDevice(CurrentDevice()); }

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

namespace {

// A few persistent host threads that split large memcpys between them.
class HostCopyPool {
 public:
  static HostCopyPool &Get() {
    static HostCopyPool pool;
    return pool;
  }
  // dst[r * dpitch .. + width) = src[r * spitch .. + width) for r < rows (bytes), in parallel.
  void Copy2D(char *dst, size_t dpitch, const char *src, size_t spitch, size_t width, size_t rows) {
    const int parts = (int)std::min<size_t>(workers_.size() + 1, std::max<size_t>(1, rows * width >> 20));
    if (parts <= 1) {
      CopyRows(dst, dpitch, src, spitch, width, 0, rows);
      return;
    }
    // one split copy at a time (callers on different devices share the pool and its counter)
    std::lock_guard<std::mutex> one_call(call_mu_);
    {
      std::unique_lock<std::mutex> lock(mu_);
      pending_ = parts - 1;
      for (int p = 1; p < parts; ++p) {
        const size_t r0 = rows * p / parts, r1 = rows * (p + 1) / parts;
        tasks_.push_back([=]() { CopyRows(dst, dpitch, src, spitch, width, r0, r1); });
      }
    }
    cv_.notify_all();
    CopyRows(dst, dpitch, src, spitch, width, 0, rows / parts);  // the caller takes part 0
    std::unique_lock<std::mutex> lock(mu_);
    done_.wait(lock, [&]() { return pending_ == 0; });
  }

 private:
  HostCopyPool() {
    const int n = std::max(1, std::min(7, (int)std::thread::hardware_concurrency() / 2 - 1));
    for (int i = 0; i < n; ++i) workers_.emplace_back([this]() { Work(); });
  }
  ~HostCopyPool() {
    {
      std::unique_lock<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread &t : workers_) t.join();
  }
  static void CopyRows(char *dst, size_t dpitch, const char *src, size_t spitch, size_t width,
                       size_t r0, size_t r1) {
    if (dpitch == width && spitch == width) {
      memcpy(dst + r0 * width, src + r0 * width, (r1 - r0) * width);
      return;
    }
    for (size_t r = r0; r < r1; ++r) memcpy(dst + r * dpitch, src + r * spitch, width);
  }
  void Work() {
    for (;;) {
      std::function<void()> task;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [&]() { return stop_ || !tasks_.empty(); });
        if (stop_ && tasks_.empty()) return;
        task = std::move(tasks_.back());
        tasks_.pop_back();
      }
      task();
      std::unique_lock<std::mutex> lock(mu_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> workers_;
  std::vector<std::function<void()>> tasks_;
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  int pending_ = 0;
  bool stop_ = false;
};

constexpr size_t kStageBytes = (size_t)32 << 20;
constexpr int kStageBuffers = 3;
struct Staging {
  std::mutex mu;  // one pipelined copy at a time per device
  char *buf[kStageBuffers] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[kStageBuffers] = {nullptr, nullptr, nullptr};
  bool ready = false;
};
Staging g_staging_tab[kMaxDevices];

bool IsPageLocked(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

}  // namespace

__global__ void clear_words_kernel(int *words, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) ctl_store(words + i, 0);
}

int ClearWords(int *d_words, int n, hipStream_t st) {
  if (!d_words || n <= 0) {
    SetError("ClearWords: bad argument");
    return AMLIKE_EINVAL;
  }
  hipLaunchKernelGGL(clear_words_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_words, n);
  AMLIKE_HIP(hipGetLastError());
  return AMLIKE_OK;
}

int CopyMatrixToHost(float *out, size_t ldo, const float *d_src, int rows, int cols, hipStream_t st) {
  const size_t row_bytes = (size_t)cols * sizeof(float), total = row_bytes * rows;
  const bool staged = total >= 2 * kStageBytes && !IsPageLocked(out) && !getenv("AMLIKE_NO_STAGED_COPY");
  if (!staged) {
    if (ldo == (size_t)cols)
      AMLIKE_HIP(hipMemcpyAsync(out, d_src, total, hipMemcpyDeviceToHost, st));
    else
      AMLIKE_HIP(hipMemcpy2DAsync(out, ldo * sizeof(float), d_src, row_bytes, row_bytes, rows,
                                  hipMemcpyDeviceToHost, st));
    AMLIKE_HIP(hipStreamSynchronize(st));
    return AMLIKE_OK;
  }
  int dev = 0;
  AMLIKE_HIP(hipGetDevice(&dev));  // the handle's device (bound by the caller)
  Staging &g_staging = g_staging_tab[std::max(0, std::min(dev, kMaxDevices - 1))];
  std::lock_guard<std::mutex> lock(g_staging.mu);
  if (!g_staging.ready) {  // first use on this device
    for (int i = 0; i < kStageBuffers; ++i) {
      AMLIKE_HIP(hipHostMalloc(reinterpret_cast<void **>(&g_staging.buf[i]), kStageBytes, hipHostMallocDefault));
      AMLIKE_HIP(hipEventCreateWithFlags(&g_staging.ev[i], hipEventDisableTiming));
    }
    g_staging.ready = true;
  }
  const int chunk_rows = (int)std::max<size_t>(1, kStageBytes / row_bytes);
  const int nchunks = (rows + chunk_rows - 1) / chunk_rows;
  auto drain = [&](int c) -> int {
    const int r0 = c * chunk_rows, rn = std::min(chunk_rows, rows - r0);
    AMLIKE_HIP(hipEventSynchronize(g_staging.ev[c % kStageBuffers]));
    HostCopyPool::Get().Copy2D(reinterpret_cast<char *>(out + (size_t)r0 * ldo), ldo * sizeof(float),
                               g_staging.buf[c % kStageBuffers], row_bytes, row_bytes, (size_t)rn);
    return AMLIKE_OK;
  };
  for (int c = 0; c < nchunks; ++c) {
    const int r0 = c * chunk_rows, rn = std::min(chunk_rows, rows - r0);
    AMLIKE_HIP(hipMemcpyAsync(g_staging.buf[c % kStageBuffers], d_src + (size_t)r0 * cols,
                              (size_t)rn * row_bytes, hipMemcpyDeviceToHost, st));
    AMLIKE_HIP(hipEventRecord(g_staging.ev[c % kStageBuffers], st));
    if (c >= kStageBuffers - 1) {
      int rc = drain(c - (kStageBuffers - 1));
      if (rc != AMLIKE_OK) return rc;
    }
  }
  for (int c = std::max(0, nchunks - (kStageBuffers - 1)); c < nchunks; ++c) {
    int rc = drain(c);
    if (rc != AMLIKE_OK) return rc;
  }
  return AMLIKE_OK;
}

PinnedUploadRing::~PinnedUploadRing() {
  for (Slot &s : slots_) {
    if (s.ev) {
      if (s.used) (void)hipEventSynchronize(s.ev);
      (void)hipEventDestroy(s.ev);
    }
    if (s.buf) (void)hipHostFree(s.buf);
  }
}

int PinnedUploadRing::Upload(void *d_dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return AMLIKE_OK;
  Slot &s = slots_[next_];
  next_ = (next_ + 1) % kSlots;
  if (!s.ev) AMLIKE_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  if (s.used) AMLIKE_HIP(hipEventSynchronize(s.ev));  // the copy that last read this buffer is done
  if (bytes > s.bytes) {
    if (s.buf) AMLIKE_HIP(hipHostFree(s.buf));
    s.buf = nullptr;
    s.bytes = 0;
    const size_t grow = std::max<size_t>(4096, bytes + bytes / 2);
    AMLIKE_HIP(hipHostMalloc(&s.buf, grow, hipHostMallocDefault));
    s.bytes = grow;
  }
  memcpy(s.buf, src, bytes);
  AMLIKE_HIP(hipMemcpyAsync(d_dst, s.buf, bytes, hipMemcpyHostToDevice, st));
  AMLIKE_HIP(hipEventRecord(s.ev, st));
  s.used = true;
  return AMLIKE_OK;
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
  if (n <= 0 || device < 0 || device >= n || device >= amlike::kMaxDevices) {
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
  if (int rc = amlike::SetupNoAllocScratch(device)) return rc;
  amlike::g_num_cu[device] = prop.multiProcessorCount;
  amlike::t_device = device;
  amlike::g_default_device.store(device);
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
