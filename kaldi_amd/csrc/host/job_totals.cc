#include "job_totals.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>
#include <algorithm>

namespace kaldi_amd {

// RCCL is loaded only when a job of several processes asks for its totals: a run of one process
// does not pay for it.
struct JobTotals::Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;

  Rccl() {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) throw std::runtime_error(std::string("job totals: cannot load librccl: ") + dlerror());
    Resolve(&GetUniqueId, "ncclGetUniqueId");
    Resolve(&CommInitRank, "ncclCommInitRank");
    Resolve(&AllReduce, "ncclAllReduce");
    Resolve(&CommDestroy, "ncclCommDestroy");
    Resolve(&GetErrorString, "ncclGetErrorString");
  }
  template <typename F>
  void Resolve(F *fn, const char *symbol) {
    *fn = reinterpret_cast<F>(dlsym(handle, symbol));
    if (!*fn) throw std::runtime_error(std::string("job totals: librccl has no ") + symbol);
  }
  void Check(ncclResult_t r, const char *what) const {
    if (r != ncclSuccess)
      throw std::runtime_error(std::string("job totals: ") + what + ": " + GetErrorString(r));
  }
};

namespace {

void CheckHip(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("job totals: ") + what + ": " + hipGetErrorString(e));
}

void PublishId(const std::string &path, const ncclUniqueId &id) {
  const std::string tmp = path + ".tmp." + std::to_string(static_cast<long long>(getpid()));
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) throw std::runtime_error("job totals: cannot write " + tmp);
  const size_t n = fwrite(&id, 1, sizeof(id), f);
  const bool ok = fclose(f) == 0 && n == sizeof(id);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
    unlink(tmp.c_str());
    throw std::runtime_error("job totals: cannot publish " + path);
  }
}

void AwaitId(const std::string &path, ncclUniqueId *id, int timeout_s) {
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
  for (;;) {
    struct stat st;
    if (stat(path.c_str(), &st) == 0 && st.st_size == static_cast<off_t>(sizeof(*id))) {
      FILE *f = fopen(path.c_str(), "rb");
      if (f) {
        const size_t n = fread(id, 1, sizeof(*id), f);
        fclose(f);
        if (n == sizeof(*id)) return;
      }
    }
    if (std::chrono::steady_clock::now() > deadline)
      throw std::runtime_error("job totals: process 0 did not publish " + path + " within " +
                               std::to_string(timeout_s) + " s");
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
}

}  // namespace

JobTotals::JobTotals(const JobTotalsOptions &opts, int device) : device_(device) {
  static_assert(sizeof(long long) == 8, "counts travel as 64-bit integers");
  try {
    if (opts.num_procs < 1 || opts.proc_id < 0 || opts.proc_id >= opts.num_procs)
      throw std::runtime_error("job totals: --proc-id must lie in [0, --num-procs)");
    if (opts.rendezvous.empty()) throw std::runtime_error("job totals: --rendezvous is needed with --num-procs");
    rccl_ = new Rccl;
    CheckHip(hipSetDevice(device), "hipSetDevice");
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    if (opts.proc_id == 0) {
      rccl_->Check(rccl_->GetUniqueId(&id), "ncclGetUniqueId");
      PublishId(opts.rendezvous, id);
    } else {
      AwaitId(opts.rendezvous, &id, opts.rendezvous_timeout);
    }
    ncclComm_t comm = nullptr;
    const ncclResult_t joined = rccl_->CommInitRank(&comm, opts.num_procs, id, opts.proc_id);
    if (opts.proc_id == 0) unlink(opts.rendezvous.c_str());  // every process has read it by now, or the job failed
    rccl_->Check(joined, "ncclCommInitRank");
    comm_ = comm;
  } catch (const std::exception &e) {
    error_ = e.what();
    comm_ = nullptr;
  }
}

JobTotals::~JobTotals() {
  if (comm_ && rccl_) (void)rccl_->CommDestroy(static_cast<ncclComm_t>(comm_));
  delete rccl_;
}

bool JobTotals::AllReduce(long long *counts, int n) {
  if (!comm_) {
    if (error_.empty()) error_ = "job totals: this process has not joined the job";
    return false;
  }
  long long *d_counts = nullptr;
  hipStream_t stream = nullptr;
  bool ok = true;
  try {
    CheckHip(hipSetDevice(device_), "hipSetDevice");
    CheckHip(hipMalloc(reinterpret_cast<void **>(&d_counts), sizeof(long long) * n), "hipMalloc");
    CheckHip(hipStreamCreate(&stream), "hipStreamCreate");
    CheckHip(hipMemcpyAsync(d_counts, counts, sizeof(long long) * n, hipMemcpyHostToDevice, stream), "copy up");
    rccl_->Check(rccl_->AllReduce(d_counts, d_counts, static_cast<size_t>(n), ncclInt64, ncclSum,
                                  static_cast<ncclComm_t>(comm_), stream),
                 "ncclAllReduce");
    std::vector<long long> sums(n);
    CheckHip(hipMemcpyAsync(sums.data(), d_counts, sizeof(long long) * n, hipMemcpyDeviceToHost, stream), "copy down");
    CheckHip(hipStreamSynchronize(stream), "hipStreamSynchronize");
    std::copy(sums.begin(), sums.end(), counts);  // the caller's own counts stay intact on failure
  } catch (const std::exception &e) {
    error_ = e.what();
    ok = false;
  }
  if (stream) (void)hipStreamDestroy(stream);
  if (d_counts) (void)hipFree(d_counts);
  return ok;
}

}  // namespace kaldi_amd
