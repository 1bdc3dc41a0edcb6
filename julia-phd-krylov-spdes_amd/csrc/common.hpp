// Context, error plumbing, device buffers and the run-time RCCL binding of libmi355schur.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is bound with dlopen at run time (no link dependency)

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mi355schur.h"

namespace mi {

// ------------------------------------------------------------------ errors
inline std::string &last_error() {
  static thread_local std::string s;
  return s;
}
inline int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}
struct Error {
  int code;
};
[[noreturn]] inline void raise(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  throw Error{code};
}
#define MI_HIP(expr)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      ::mi::raise(MI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                  __LINE__);                                                                     \
  } while (0)

// Every extern "C" body runs inside this guard: no C++ exception crosses the ABI.
template <class F>
inline int guarded(F &&f) {
  try {
    return f();
  } catch (const Error &e) {
    return e.code;
  } catch (const std::bad_alloc &) {
    return fail(MI_ERR_HIP, "host allocation failed");
  } catch (...) {
    return fail(MI_ERR_HIP, "unexpected exception");
  }
}

// ------------------------------------------------------------------ device buffers
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    MI_HIP(hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)));
  }
  void ensure(size_t count) {
    if (count > n) alloc(count);
  }
  void upload(const T *src, size_t count, hipStream_t s) {
    ensure(count);
    if (count) MI_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void upload(const std::vector<T> &v, hipStream_t s) {
    upload(v.data(), v.size(), s);
    MI_HIP(hipStreamSynchronize(s));  // the std::vector may die right after
  }
  void zero(hipStream_t s) {
    if (n) MI_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

// ------------------------------------------------------------------ RCCL, bound at run time
struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  static Rccl &get();  // throws MI_ERR_COMM when librccl cannot be bound
};
#define MI_NCCL(expr)                                                                       \
  do {                                                                                      \
    ncclResult_t r_ = (expr);                                                               \
    if (r_ != ncclSuccess)                                                                  \
      ::mi::raise(MI_ERR_COMM, "%s failed: %s", #expr, ::mi::Rccl::get().GetErrorString(r_)); \
  } while (0)

struct Operator;
struct SolverWorkspace;

}  // namespace mi

// ------------------------------------------------------------------ the context
namespace mi {
// Pinned (device-visible) host buffer: kernels read / write it directly over PCIe, so a host-pointer call needs no
// separate copy operations on the stream.
struct PinnedBuf {
  double *p = nullptr;
  size_t n = 0;
  void ensure(size_t m) {
    if (m <= n) return;
    release();
    MI_HIP(hipHostMalloc((void **)&p, (m ? m : 1) * sizeof(double)));
    n = m;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr; n = 0;
  }
  ~PinnedBuf() { release(); }
};
}  // namespace mi

namespace mi {
// In-process stand-in for the RCCL communicator: N contexts ("ranks") of ONE process on one GPU, each driven by its own
// host thread, sum their buffers through device memory with host-side rendezvous. It exists to exercise the sharded
// (multi-GPU) code paths — slot-table union, non-local subdomains, replicated vectors — on a single-GPU box, where RCCL
// refuses two ranks on the same device. Never captured into graphs (it synchronises with the host).
struct LoopGroup {
  int n;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long long generation = 0;
  std::vector<const double *> send;
  std::vector<hipEvent_t> ready;
  explicit LoopGroup(int n_) : n(n_), send(n_, nullptr), ready(n_, nullptr) {}
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long long gen = generation;
    if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != gen; });
  }
};
void loop_allreduce(LoopGroup &g, int rank, const double *send, double *recv, size_t n, hipStream_t s);
}  // namespace mi

struct mi_ctx_s {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  int ptr_mode = MI_PTR_HOST;
  int chunk = 8;
  ncclComm_t comm = nullptr;
  mi::LoopGroup *loop = nullptr;  // in-process test communicator (see LoopGroup); mutually exclusive with comm
  int rank = 0, n_ranks = 1;
  bool has_comm() const { return comm != nullptr || loop != nullptr; }
  bool no_graph = false;  // set when a captured collective could not be instantiated: eager launches from then on
  // scratch for BLAS-1 entry points and reductions
  mi::DevBuf<double> scratch_a, scratch_b, partials, scalar;
  mi::PinnedBuf pin_b, pin_x;  // host-pointer solves: b, x0 in / x out without stream copies
  // solver workspaces keyed by problem size; graphs keyed inside
  std::map<int64_t, std::unique_ptr<mi::SolverWorkspace>> workspaces;
  void use() const { MI_HIP(hipSetDevice(device)); }
  void allreduce(double *buf, size_t n) { allreduce(buf, buf, n); }
  void allreduce(const double *send, double *recv, size_t n) {
    if (loop) { mi::loop_allreduce(*loop, rank, send, recv, n, stream); return; }
    if (comm)  // also with n_ranks == 1, so that a single-GPU box exercises the captured collective
      MI_NCCL(mi::Rccl::get().AllReduce(send, recv, n, ncclDouble, ncclSum, comm, stream));
  }
};

struct mi_event_s {
  hipEvent_t ev;
};
