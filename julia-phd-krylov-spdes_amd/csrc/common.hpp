// Context, error plumbing, device buffers and the run-time RCCL binding of libmi355schur.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is bound with dlopen at run time (no link dependency)

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
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

// ------------------------------------------------------------------ blocking copies / fills that stay off the legacy stream
// `hipMemset` / `hipMemcpy` run on the legacy (NULL) stream, which the runtime refuses while ANY stream of the process is
// being captured ("operation would make the legacy stream depend on a capturing blocking stream") and which also tears
// that capture down: with several contexts driven by several host threads (in-process ranks, tests/test_gpu_multirank.py)
// one thread's set-up would kill another thread's graph capture. Everything blocking goes through one non-blocking
// utility stream per device instead.
struct UtilStream {   // ONE per process and device, serialised: in-process ranks need a hardware queue each (exchange.hpp), streams are not free
  std::mutex mu;
  std::map<int, hipStream_t> streams;
  static UtilStream &get() { static UtilStream *u = new UtilStream(); return *u; }
};
inline void memset_sync(void *p, int v, size_t bytes) {
  if (!bytes) return;
  UtilStream &u = UtilStream::get();
  std::lock_guard<std::mutex> lk(u.mu);
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  hipStream_t &s = u.streams[dev];
  if (!s) MI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  MI_HIP(hipMemsetAsync(p, v, bytes, s));
  MI_HIP(hipStreamSynchronize(s));
}
inline void memcpy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (!bytes) return;
  UtilStream &u = UtilStream::get();
  std::lock_guard<std::mutex> lk(u.mu);
  int dev = 0;
  MI_HIP(hipGetDevice(&dev));
  hipStream_t &s = u.streams[dev];
  if (!s) MI_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  MI_HIP(hipMemcpyAsync(dst, src, bytes, kind, s));
  MI_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------ device buffers
// Freed device memory is kept in a per-process cache (size classes, per device) and handed out again instead of going
// back to the runtime: `hipFree` waits for EVERY stream of the device, so a context that frees a temporary while a kernel
// of another context is waiting for it (in-process ranks joined by device-side flags, exchange.hpp) would stop both until
// the wait expires — and a free in a per-solve path costs a device-wide synchronisation on any workload. The cache is
// bounded (MI355_POOL_MB, default 4096: beyond it blocks are really freed).
struct DevPool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void *> blocks;   // (device, capacity) -> free block
  size_t cached = 0, limit;
  DevPool() {
    const char *e = std::getenv("MI355_POOL_MB");
    limit = (size_t)(e && *e ? std::atoll(e) : 4096) << 20;
  }
  ~DevPool() { /* process exit: the runtime may already be gone; blocks die with the process */ }
  static DevPool &get() { static DevPool *p = new DevPool(); return *p; }
  static size_t size_class(size_t bytes) {
    const size_t g = bytes <= (64u << 10) ? 256 : bytes <= (4u << 20) ? 4096 : (1u << 20);
    return (std::max<size_t>(bytes, 1) + g - 1) / g * g;
  }
  void *take(size_t bytes, size_t *cap, int *dev) {
    *cap = size_class(bytes);
    MI_HIP(hipGetDevice(dev));
    void *reuse = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = blocks.find({*dev, *cap});
      if (it != blocks.end()) { void *p = it->second; blocks.erase(it); cached -= *cap; reuse = p; }
    }
    if (reuse) { memset_sync(reuse, 0, *cap); return reuse; }   // as clean as a fresh allocation usually is (no stale NaN behind a 0 * x)
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, *cap);
    if (e != hipSuccess) {   // out of memory: give the cache back and try once more
      (void)hipGetLastError();
      trim();
      MI_HIP(hipMalloc(&p, *cap));
    }
    return p;
  }
  void give(void *p, size_t cap, int dev) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (cached + cap <= limit) { blocks.insert({{dev, cap}, p}); cached += cap; return; }
    }
    (void)hipFree(p);
  }
  void trim() {
    std::multimap<std::pair<int, size_t>, void *> drop;
    { std::lock_guard<std::mutex> lk(mu); drop.swap(blocks); cached = 0; }
    for (auto &kv : drop) (void)hipFree(kv.second);
  }
};

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  size_t cap_bytes = 0;
  int dev = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), cap_bytes(o.cap_bytes), dev(o.dev) { o.p = nullptr; o.n = 0; o.cap_bytes = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; cap_bytes = o.cap_bytes; dev = o.dev; o.p = nullptr; o.n = 0; o.cap_bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) DevPool::get().give(p, cap_bytes, dev);
    p = nullptr;
    n = 0;
    cap_bytes = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    p = (T *)DevPool::get().take((count ? count : 1) * sizeof(T), &cap_bytes, &dev);
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
  int mode = 0;                       // 0: device-side peer exchange when the runtime has a hardware queue per rank; 1: host rendezvous
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long long generation = 0;
  std::vector<const double *> send;
  std::vector<hipEvent_t> ready;
  std::vector<void *> arena;          // peer-exchange arenas of the ranks (same process: plain pointers)
  int selftest_failed = 0;            // ranks whose trial exchanges expired (streams sharing a hardware queue): everybody falls back
  explicit LoopGroup(int n_) : n(n_), send(n_, nullptr), ready(n_, nullptr), arena(n_, nullptr) {}
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long long gen = generation;
    if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != gen; });
  }
};
void loop_allreduce(LoopGroup &g, int rank, const double *send, double *recv, size_t n, hipStream_t s);
struct PeerComm;  // exchange.hpp: the one-shot peer exchange (device-side flags, no host in the loop)
void peer_allreduce(PeerComm &p, const double *send, double *recv, size_t n, hipStream_t s);
}  // namespace mi

struct mi_ctx_s {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  int ptr_mode = MI_PTR_HOST;
  int chunk = 8;
  ncclComm_t comm = nullptr;
  mi::LoopGroup *loop = nullptr;  // in-process test communicator (see LoopGroup); mutually exclusive with comm
  mi::PeerComm *peer = nullptr;   // peer exchange (exchange.hpp): beside RCCL (table exchanges) or alone (also the all-reduce)
  bool peer_on = false;           // every arena imported; mi_ctx_set_exchange can switch it off (RCCL for everything)
  bool peer_inwait = false;       // mi_ctx_set_exchange(ctx, 2): the folded launches wait for the flags themselves (one GPU per rank)
  int rank = 0, n_ranks = 1;
  bool has_comm() const { return comm != nullptr || loop != nullptr || (peer != nullptr && peer_on); }
  bool use_peer() const { return peer != nullptr && peer_on; }
  bool no_graph = false;  // set when a captured collective could not be instantiated: eager launches from then on
  long long n_replays = 0;  // hipGraphLaunch calls of the solvers on this context (mi_ctx_query)
  // scratch for BLAS-1 entry points and reductions
  mi::DevBuf<double> scratch_a, scratch_b, partials, scalar;
  mi::PinnedBuf pin_b, pin_x;  // host-pointer solves: b, x0 in / x out without stream copies
  // solver workspaces keyed by problem size; graphs keyed inside
  std::map<int64_t, std::unique_ptr<mi::SolverWorkspace>> workspaces;
  void use() const { MI_HIP(hipSetDevice(device)); }
  void allreduce(double *buf, size_t n) { allreduce(buf, buf, n); }
  void allreduce(const double *send, double *recv, size_t n) {
    if (use_peer() && !comm) { mi::peer_allreduce(*peer, send, recv, n, stream); return; }  // (with RCCL attached the generic sum stays RCCL's)
    if (loop) { mi::loop_allreduce(*loop, rank, send, recv, n, stream); return; }
    if (comm)  // also with n_ranks == 1, so that a single-GPU box exercises the captured collective
      MI_NCCL(mi::Rccl::get().AllReduce(send, recv, n, ncclDouble, ncclSum, comm, stream));
  }
};

struct mi_event_s {
  hipEvent_t ev;
};
