// extern "C" surface of libmi355schur (see include/mi355schur.h for the contract of every symbol).
#include <dlfcn.h>

#include <thread>
#include <tuple>

#include "eig_solvers.hpp"
#include "setup_gj.hpp"

namespace mi {

Rccl &Rccl::get() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // Prefer an RCCL that is already in the process (e.g. the one torch loaded), else the system one.
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
      r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (r.handle) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    }
  }
  if (!r.handle || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString)
    raise(MI_ERR_COMM, "librccl could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
  return r;
}

// Input vector in the caller's pointer mode -> device pointer valid on ctx->stream.
struct In {
  const double *dev;
  In(mi_ctx_s *c, const double *p, size_t n, DevBuf<double> &stage) {
    if (c->ptr_mode == MI_PTR_DEVICE || n == 0) { dev = p; return; }
    stage.ensure(n);
    MI_HIP(hipMemcpyAsync(stage.p, p, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    dev = stage.p;
  }
};
// In/out vector: staged on entry, copied back by finish().
struct InOut {
  mi_ctx_s *c; double *user; double *dev; size_t n; bool staged;
  InOut(mi_ctx_s *c_, double *p, size_t n_, DevBuf<double> &stage, bool read) : c(c_), user(p), n(n_) {
    staged = c->ptr_mode != MI_PTR_DEVICE && n > 0;
    if (!staged) { dev = p; return; }
    stage.ensure(n);
    dev = stage.p;
    if (read) MI_HIP(hipMemcpyAsync(dev, p, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  void finish() {
    if (staged) {
      MI_HIP(hipMemcpyAsync(user, dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      MI_HIP(hipStreamSynchronize(c->stream));
    }
  }
};

static double reduce_to_host(mi_ctx_s *c, int g, int take_sqrt) {
  c->scalar.ensure(1);
  hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(NT), 0, c->stream, c->partials.p, g, c->scalar.p, take_sqrt);
  MI_HIP(hipGetLastError());
  double out = 0.0;
  MI_HIP(hipMemcpyAsync(&out, c->scalar.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MI_HIP(hipStreamSynchronize(c->stream));
  return out;
}

// out[i] = Σ_r in[r][i], r ascending (the loopback communicator's reduction)
__global__ __launch_bounds__(NT) void k_loop_sum(size_t n, int nr, const double *const *in, double *__restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    double s = in[0][i];
    for (int r = 1; r < nr; ++r) s += in[r][i];
    out[i] = s;
  }
}

void loop_allreduce(LoopGroup &g, int rank, const double *send, double *recv, size_t n, hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  if (cap != hipStreamCaptureStatusNone) raise(MI_ERR_COMM, "the loopback communicator cannot be captured into a graph");
  MI_HIP(hipStreamSynchronize(s));               // this rank's buffer is complete before it is published
  g.send[rank] = send;
  g.barrier();                                   // every rank has published a complete buffer
  DevBuf<const double *> ptrs;
  ptrs.upload(g.send.data(), g.send.size(), s);
  DevBuf<double> tmp(n);                         // in-place calls (recv == send) must not be overwritten while others read
  const int grid = (int)std::max<size_t>(1, std::min<size_t>((n + NT - 1) / NT, 1024));
  hipLaunchKernelGGL(k_loop_sum, dim3(grid), dim3(NT), 0, s, n, g.n, ptrs.p, tmp.p);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));               // done reading everyone's buffer
  g.barrier();                                   // nobody overwrites a send buffer before all ranks have read it
  MI_HIP(hipMemcpyAsync(recv, tmp.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  MI_HIP(hipStreamSynchronize(s));
}

static int run_solver(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit,
                      double eps, double *res_norm, int64_t res_cap, int64_t *it, bool want_M, bool want_W) {
  if (!A || !A->impl || !b || !x || !it || res_cap < 0 || (res_cap > 0 && !res_norm) || maxit < 0)
    return fail(MI_ERR_BAD_ARG, "solver: NULL or negative argument");
  if (want_M && (!M || !M->impl)) return fail(MI_ERR_BAD_ARG, "solver: preconditioner handle is NULL");
  if (want_W && (nvec < 0 || (nvec > 0 && !W) || nvec > 1024)) return fail(MI_ERR_BAD_ARG, "solver: bad W / nvec");
  Operator *a = A->impl.get(), *m = want_M ? M->impl.get() : nullptr;
  if (m && (m->n != a->n || m->ctx != a->ctx)) return fail(MI_ERR_BAD_ARG, "solver: A and M differ in size or context");
  mi_ctx_s *c = a->ctx;
  return guarded([&]() -> int {
    c->use();
    const size_t n = (size_t)a->n;
    DevBuf<double> wstage;
    In wi(c, W, want_W ? n * (size_t)nvec : 0, wstage);
    Krylov k(c, a, m, want_W ? (int)nvec : 0);
    if (c->ptr_mode == MI_PTR_DEVICE) return k.solve(b, x, wi.dev, maxit, eps, res_norm, res_cap, it);
    // Host pointers (Julia arrays): b and x0 go through pinned buffers that the solve's entry kernel reads and its
    // exit kernel writes directly — no copy operations on the stream, one wait per solve.
    c->pin_b.ensure(n); c->pin_x.ensure(n);
    std::memcpy(c->pin_b.p, b, n * sizeof(double));
    std::memcpy(c->pin_x.p, x, n * sizeof(double));
    const int rc = k.solve(c->pin_b.p, c->pin_x.p, wi.dev, maxit, eps, res_norm, res_cap, it);  // synchronises before returning
    std::memcpy(x, c->pin_x.p, n * sizeof(double));
    return rc;
  });
}

static int run_eig_solver(EigKind kind, mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec,
                          int64_t spdim, int64_t maxit, double eps, double *res_norm, int64_t res_cap, int64_t *it,
                          double *V_out) {
  const bool want_M = kind == EIGPCG || kind == EIGDEFPCG, want_W = kind == EIGDEFCG || kind == EIGDEFPCG;
  if (!A || !A->impl || !b || !x || !it || res_cap < 0 || (res_cap > 0 && !res_norm) || maxit < 0)
    return fail(MI_ERR_BAD_ARG, "solver: NULL or negative argument");
  if (want_M && (!M || !M->impl)) return fail(MI_ERR_BAD_ARG, "solver: preconditioner handle is NULL");
  if (nvec < 1 || nvec > 512 || spdim > 4096 || (want_W && !W)) return fail(MI_ERR_BAD_ARG, "solver: bad W / nvec / spdim");
  if (spdim < 2 * nvec + 1)
    return fail(MI_ERR_BOUNDS, "spdim = %lld < 2 nvec + 1 = %lld: V[:, nev + 1] can fall outside the search space (BoundsError)",
                (long long)spdim, (long long)(2 * nvec + 1));
  Operator *a = A->impl.get(), *m = want_M ? M->impl.get() : nullptr;
  if (m && (m->n != a->n || m->ctx != a->ctx)) return fail(MI_ERR_BAD_ARG, "solver: A and M differ in size or context");
  mi_ctx_s *c = a->ctx;
  return guarded([&]() -> int {
    c->use();
    const size_t n = (size_t)a->n;
    In bi(c, b, n, c->scratch_a);
    InOut xi(c, x, n, c->scratch_b, true);
    DevBuf<double> wstage, vstage;
    In wi(c, W, want_W ? n * (size_t)nvec : 0, wstage);
    InOut vi(c, V_out, V_out ? n * (size_t)nvec : 0, vstage, false);
    EigKrylov k(c, a, m, kind, (int)nvec, (int)spdim);
    const int rc = k.solve(bi.dev, xi.dev, wi.dev, maxit, eps, res_norm, res_cap, it, V_out ? vi.dev : nullptr);
    xi.finish();
    vi.finish();
    return rc;
  });
}

// initcg / initpcg (initcg.jl:28-75, 106-160): x += W (WtAW \ W'(b - A x)), then cg / pcg from that guess.
static int run_init_solver(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit,
                           double eps, double *res_norm, int64_t res_cap, int64_t *it, bool want_M) {
  if (!A || !A->impl || !b || !x || !it || res_cap < 0 || (res_cap > 0 && !res_norm) || maxit < 0)
    return fail(MI_ERR_BAD_ARG, "solver: NULL or negative argument");
  if (want_M && (!M || !M->impl)) return fail(MI_ERR_BAD_ARG, "solver: preconditioner handle is NULL");
  if (nvec < 1 || !W || nvec > 1024) return fail(MI_ERR_BAD_ARG, "solver: bad W / nvec");
  Operator *a = A->impl.get(), *m = want_M ? M->impl.get() : nullptr;
  if (m && (m->n != a->n || m->ctx != a->ctx)) return fail(MI_ERR_BAD_ARG, "solver: A and M differ in size or context");
  mi_ctx_s *c = a->ctx;
  return guarded([&]() -> int {
    c->use();
    const size_t n = (size_t)a->n;
    In bi(c, b, n, c->scratch_a);
    InOut xi(c, x, n, c->scratch_b, true);
    DevBuf<double> wstage;
    In wi(c, W, n * (size_t)nvec, wstage);
    {
      Krylov guess(c, a, nullptr, (int)nvec, /*generic=*/true);
      int64_t mx = maxit, cap = 0;
      double e = eps;
      guess.begin(bi.dev, xi.dev, wi.dev, mx, e, cap);   // leaves the deflated guess in the workspace's x
      MI_HIP(hipMemcpyAsync(xi.dev, guess.ws.x, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    Krylov k(c, a, m, 0);
    const int rc = k.solve(bi.dev, xi.dev, nullptr, maxit, eps, res_norm, res_cap, it);
    xi.finish();
    return rc;
  });
}

void peer_allreduce(PeerComm &p, const double *send, double *recv, size_t n, hipStream_t s) { p.allreduce(send, recv, n, s); }
}  // namespace mi

using namespace mi;

extern "C" {

int mi_version(void) { return 200; }  // 0.2.0
const char *mi_last_error(void) { return last_error().c_str(); }

int mi_device_count(int *count) {
  if (!count) return fail(MI_ERR_BAD_ARG, "count is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return MI_OK;
}

int mi_ctx_create(int device, mi_ctx_t *ctx) {
  if (!ctx) return fail(MI_ERR_BAD_ARG, "ctx is NULL");
  *ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(MI_ERR_NO_DEVICE, "no HIP device visible: libmi355schur has no CPU fallback");
  if (device < 0 || device >= n) return fail(MI_ERR_BAD_ARG, "device %d out of range [0,%d)", device, n);
  return guarded([&]() -> int {
    hipDeviceProp_t prop;
    MI_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !std::getenv("MI355_ALLOW_ANY_ARCH"))
      return fail(MI_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    std::unique_ptr<mi_ctx_s> c(new mi_ctx_s);
    c->device = device;
    c->use();
    MI_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    c->chunk = env_int("MI355_CHUNK", 8);
    c->partials.alloc(MAX_PARTS);
    c->scalar.alloc(1);
    *ctx = c.release();
    return MI_OK;
  });
}

int mi_ctx_destroy(mi_ctx_t ctx) {
  if (!ctx) return MI_OK;
  return guarded([&]() -> int {
    ctx->use();
    (void)hipStreamSynchronize(ctx->stream);
    ctx->workspaces.clear();
    if (ctx->comm) (void)Rccl::get().CommDestroy(ctx->comm);
    delete ctx->peer;
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MI_OK;
  });
}

int mi_ctx_set_pointer_mode(mi_ctx_t ctx, int mode) {
  if (!ctx || (mode != MI_PTR_HOST && mode != MI_PTR_DEVICE)) return fail(MI_ERR_BAD_ARG, "bad ctx or pointer mode");
  ctx->ptr_mode = mode;
  return MI_OK;
}

int mi_ctx_set_stream(mi_ctx_t ctx, void *hip_stream) {
  if (!ctx) return fail(MI_ERR_BAD_ARG, "ctx is NULL");
  return guarded([&]() -> int {
    ctx->use();
    MI_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->workspaces) kv.second->drop_graphs();  // graphs are tied to the capture stream
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return MI_OK;
  });
}

int mi_ctx_get_stream(mi_ctx_t ctx, void **hip_stream) {
  if (!ctx || !hip_stream) return fail(MI_ERR_BAD_ARG, "NULL argument");
  *hip_stream = (void *)ctx->stream;
  return MI_OK;
}

int mi_ctx_synchronize(mi_ctx_t ctx) {
  if (!ctx) return fail(MI_ERR_BAD_ARG, "ctx is NULL");
  return guarded([&]() -> int { ctx->use(); MI_HIP(hipStreamSynchronize(ctx->stream)); return MI_OK; });
}

int mi_ctx_set_chunk(mi_ctx_t ctx, int iterations_per_graph) {
  if (!ctx || iterations_per_graph < 0 || iterations_per_graph > 4096) return fail(MI_ERR_BAD_ARG, "bad ctx or chunk");
  ctx->chunk = iterations_per_graph;
  return MI_OK;
}

// ---------------------------------------------------------------- multi-GPU
int mi_comm_unique_id(void *id_out) {
  if (!id_out) return fail(MI_ERR_BAD_ARG, "id_out is NULL");
  return guarded([&]() -> int {
    static_assert(sizeof(ncclUniqueId) == MI_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    MI_NCCL(Rccl::get().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return MI_OK;
  });
}

int mi_ctx_comm_init(mi_ctx_t ctx, const void *id, int rank, int n_ranks) {
  if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MI_ERR_BAD_ARG, "bad communicator arguments");
  return guarded([&]() -> int {
    ctx->use();
    if (ctx->comm) { MI_NCCL(Rccl::get().CommDestroy(ctx->comm)); ctx->comm = nullptr; }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    MI_NCCL(Rccl::get().CommInitRank(&ctx->comm, n_ranks, uid, rank));
    ctx->rank = rank; ctx->n_ranks = n_ranks;
    for (auto &kv : ctx->workspaces) kv.second->drop_graphs();
    // Warm-up collective outside any stream capture: RCCL sets up its channels lazily on first use,
    // which must not happen while an iteration graph is being captured.
    ctx->scalar.ensure(1);
    MI_HIP(hipMemsetAsync(ctx->scalar.p, 0, sizeof(double), ctx->stream));
    ctx->allreduce(ctx->scalar.p, 1);
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI_OK;
  });
}

// In-process "ranks" for single-GPU testing of the sharded paths (see LoopGroup in common.hpp).
int mi_loopback_group_create(int n_ranks, void **group) {
  if (!group || n_ranks < 1 || n_ranks > 64) return fail(MI_ERR_BAD_ARG, "bad loopback group arguments");
  return guarded([&]() -> int { *group = new LoopGroup(n_ranks); return MI_OK; });
}
int mi_loopback_group_destroy(void *group) {
  if (!group) return MI_OK;
  LoopGroup *g = static_cast<LoopGroup *>(group);
  for (auto e : g->ready) if (e) (void)hipEventDestroy(e);
  for (void *a : g->arena) if (a) (void)hipFree(a);
  delete g;
  return MI_OK;
}
static size_t peer_arena_default() {
  const int mb = env_int("MI355_PEER_ARENA_MB", 64);
  return (size_t)std::max(4, mb) << 20;
}
int mi_ctx_loopback_init(mi_ctx_t ctx, void *group, int rank) {
  LoopGroup *g = static_cast<LoopGroup *>(group);
  if (!ctx || !g || rank < 0 || rank >= g->n || ctx->comm) return fail(MI_ERR_BAD_ARG, "bad loopback arguments");
  return guarded([&]() -> int {
    ctx->use();
    ctx->loop = g; ctx->rank = rank; ctx->n_ranks = g->n;
    for (auto &kv : ctx->workspaces) kv.second->drop_graphs();
    // Device-side joining needs every rank's stream on a hardware queue of its own (a wait kernel at the head of a queue
    // it shares with the kernel it waits for never ends; measured: 8 streams on the default 4 queues expire every wait,
    // GPU_MAX_HW_QUEUES=16 runs them). Same answer on every rank: the environment is the process's.
    const bool device_side = g->mode == 0 && g->n <= XCHG_MAX_RANKS && env_int("GPU_MAX_HW_QUEUES", 4) >= g->n &&
                             !env_int("MI355_LOOPBACK_HOST", 0);
    if (!device_side) {
      ctx->no_graph = true;  // the host-rendezvous collective synchronises with the host
      return MI_OK;
    }
    delete ctx->peer;
    ctx->peer = new PeerComm();
    ctx->peer->init(ctx->device, rank, g->n, peer_arena_default());
    ctx->peer->owns_arena = false;                  // freed with the group: a rank that leaves early must not pull memory from under its peers' stores
    g->arena[rank] = ctx->peer->arena;
    g->barrier();                                   // every rank has published its arena
    for (int q = 0; q < g->n; ++q) ctx->peer->import_peer(q, nullptr, g->arena[q]);
    ctx->peer->finish();
    ctx->peer_on = true;
    ctx->no_graph = false;
    // Trial exchanges with a short bound: the ranks lined up by the host, then STAGGERED, each exchange between an
    // asynchronous copy in and one out — if two ranks' streams share a hardware queue (more live streams in the process
    // than GPU_MAX_HW_QUEUES), or the runtime puts the ranks' copies on one engine queue (a copy that waits for a
    // spinning kernel then blocks the next rank's copy in: measured with copies above GPU_FORCE_BLIT_COPY_SIZE), a wait
    // expires here instead of in a solve, and the whole group falls back to the host rendezvous.
    const long long keep = ctx->peer->peers.timeout;
    ctx->peer->set_timeout_ms(env_int("MI355_PEER_SELFTEST_MS", 400));
    const size_t trial = 16384;   // doubles: 128 KiB, above the runtime's default blit threshold
    ctx->pin_b.ensure(trial);
    DevBuf<double> tbuf(trial);
    ctx->peer->reserve_stage(trial);
    std::memset(ctx->pin_b.p, 0, trial * sizeof(double));
    for (int k = 0; k < 3; ++k) {
      g->barrier();
      std::this_thread::sleep_for(std::chrono::milliseconds(3 * ((rank + k) % g->n)));
      MI_HIP(hipMemcpyAsync(tbuf.p, ctx->pin_b.p, trial * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      ctx->peer->allreduce(tbuf.p, tbuf.p, trial, ctx->stream);
      MI_HIP(hipMemcpyAsync(ctx->pin_b.p, tbuf.p, trial * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      MI_HIP(hipStreamSynchronize(ctx->stream));
    }
    const int bad = ctx->peer->take_error(ctx->stream);
    { std::lock_guard<std::mutex> lk(g->mu); g->selftest_failed += bad; }
    g->barrier();
    ctx->peer->set_timeout_ticks(keep);
    if (g->selftest_failed) {
      ctx->peer_on = false;
      ctx->no_graph = true;
    }
    g->barrier();
    return MI_OK;
  });
}
int mi_loopback_group_set_mode(void *group, int mode) {
  if (!group || mode < 0 || mode > 1) return fail(MI_ERR_BAD_ARG, "bad loopback mode");
  static_cast<LoopGroup *>(group)->mode = mode;
  return MI_OK;
}

// ---- the peer exchange (exchange.hpp)
int mi_ctx_peer_init(mi_ctx_t ctx, int rank, int n_ranks, int64_t arena_bytes) {
  if (!ctx || arena_bytes < 0) return fail(MI_ERR_BAD_ARG, "bad peer-exchange arguments");
  return guarded([&]() -> int {
    ctx->use();
    MI_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->workspaces) kv.second->drop_graphs();
    delete ctx->peer;
    ctx->peer = nullptr; ctx->peer_on = false;
    ctx->peer = new PeerComm();
    ctx->peer->init(ctx->device, rank, n_ranks, arena_bytes ? (size_t)arena_bytes : peer_arena_default());
    if (!ctx->comm) { ctx->rank = rank; ctx->n_ranks = n_ranks; }
    else if (ctx->rank != rank || ctx->n_ranks != n_ranks) raise(MI_ERR_BAD_ARG, "peer exchange: rank %d of %d differs from the communicator's %d of %d", rank, n_ranks, ctx->rank, ctx->n_ranks);
    return MI_OK;
  });
}
int mi_ctx_peer_export(mi_ctx_t ctx, void *handle_out, void **base_out) {
  if (!ctx || !ctx->peer || (!handle_out && !base_out)) return fail(MI_ERR_BAD_ARG, "peer exchange not initialised, or nothing to export into");
  return guarded([&]() -> int {
    ctx->use();
    if (handle_out) ctx->peer->export_handle(handle_out);
    if (base_out) *base_out = ctx->peer->arena;
    return MI_OK;
  });
}
int mi_ctx_peer_import(mi_ctx_t ctx, int rank, const void *handle, void *same_process_base) {
  if (!ctx || !ctx->peer) return fail(MI_ERR_BAD_ARG, "peer exchange not initialised");
  return guarded([&]() -> int { ctx->use(); ctx->peer->import_peer(rank, handle, same_process_base); return MI_OK; });
}
int mi_ctx_peer_ready(mi_ctx_t ctx) {
  if (!ctx || !ctx->peer) return fail(MI_ERR_BAD_ARG, "peer exchange not initialised");
  return guarded([&]() -> int { ctx->use(); ctx->peer->finish(); ctx->peer_on = true; return MI_OK; });
}
int mi_ctx_query(mi_ctx_t ctx, int what, int64_t *out) {
  if (!ctx || !out) return fail(MI_ERR_BAD_ARG, "bad query arguments");
  return guarded([&]() -> int {
    ctx->use();
    switch (what) {
      case MI_QUERY_NO_GRAPH: *out = ctx->no_graph ? 1 : 0; break;
      case MI_QUERY_PEER_EXCHANGE: *out = ctx->use_peer() ? (ctx->peer_inwait ? 3 : ctx->peer->fine_grained ? 2 : 1) : 0; break;
      case MI_QUERY_GRAPH_REPLAYS: *out = ctx->n_replays; break;
      case MI_QUERY_SPECTRAL_PINV: *out = spectral_pinv_calls().load(); break;
      case MI_QUERY_EXPERIMENTAL:
#ifdef MI355_EXPERIMENTAL
        *out = 1;
#else
        *out = 0;
#endif
        break;
      case MI_QUERY_EXCHANGES: {
        unsigned long long e = 0;
        if (ctx->peer) {
          MI_HIP(hipStreamSynchronize(ctx->stream));
          memcpy_sync(&e, &ctx->peer->st->epoch, sizeof e, hipMemcpyDeviceToHost);
        }
        *out = (int64_t)e;
        break;
      }
      default: return fail(MI_ERR_BAD_ARG, "unknown query %d", what);
    }
    return MI_OK;
  });
}
int mi_ctx_set_exchange(mi_ctx_t ctx, int use_peer_exchange) {
  if (!ctx) return fail(MI_ERR_BAD_ARG, "ctx is NULL");
  if (use_peer_exchange && (!ctx->peer || !ctx->peer->ready)) return fail(MI_ERR_BAD_ARG, "peer exchange not ready");
  return guarded([&]() -> int {
    ctx->use();
    MI_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->workspaces) kv.second->drop_graphs();
    ctx->peer_on = use_peer_exchange != 0;
    // (in-launch waits read the tables with no cache maintenance of their own: only with a fine-grained arena)
    ctx->peer_inwait = use_peer_exchange == 2 && ctx->peer && ctx->peer->fine_grained;
    return MI_OK;
  });
}

int mi_ctx_comm_destroy(mi_ctx_t ctx) {
  if (!ctx) return fail(MI_ERR_BAD_ARG, "ctx is NULL");
  return guarded([&]() -> int {
    ctx->use();
    MI_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->workspaces) kv.second->drop_graphs();
    if (ctx->comm) MI_NCCL(Rccl::get().CommDestroy(ctx->comm));
    ctx->comm = nullptr; ctx->rank = 0; ctx->n_ranks = 1;
    return MI_OK;
  });
}

int mi_ctx_allreduce_sum(mi_ctx_t ctx, double *buf, int64_t n) {
  if (!ctx || !buf || n < 0) return fail(MI_ERR_BAD_ARG, "bad allreduce arguments");
  return guarded([&]() -> int {
    ctx->use();
    InOut v(ctx, buf, (size_t)n, ctx->scratch_a, true);
    ctx->allreduce(v.dev, (size_t)n);  // RCCL, the peer exchange or the in-process group's host rendezvous; a no-op without a communicator
    v.finish();
    if (ctx->use_peer() && !ctx->comm && ctx->peer->take_error(ctx->stream))
      raise(MI_ERR_COMM, "peer exchange: a wait for the other ranks expired (MI355_PEER_TIMEOUT_MS): %s", ctx->peer->err_text.c_str());
    return MI_OK;
  });
}

// ---------------------------------------------------------------- operators
#define MI_NEW_OP(ctx, op, expr)                                   \
  if (!(ctx) || !(op)) return fail(MI_ERR_BAD_ARG, "ctx/op is NULL"); \
  *(op) = nullptr;                                                 \
  return guarded([&]() -> int {                                    \
    (ctx)->use();                                                  \
    std::unique_ptr<mi_op_s> h(new mi_op_s);                       \
    h->impl.reset(expr);                                           \
    MI_HIP(hipStreamSynchronize((ctx)->stream));                   \
    *(op) = h.release();                                           \
    return MI_OK;                                                  \
  })

int mi_csr_create(mi_ctx_t ctx, int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int64_t *colidx,
                  const double *val, int index_base, mi_op_t *op) {
  MI_NEW_OP(ctx, op, new CsrOp(ctx, host_csr(n_rows, n_cols, rowptr, colidx, val, index_base)));
}

int mi_diag_create(mi_ctx_t ctx, int64_t n, const double *dinv, mi_op_t *op) {
  if (n < 0 || n >= INT32_MAX) return fail(MI_ERR_BAD_ARG, "bad n");
  MI_NEW_OP(ctx, op, new DiagOp(ctx, n, dinv));
}

int mi_schur_assembled_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d,
                              const int64_t *const *gather_idx, const double *const *Sd, int index_base,
                              int64_t dom_begin, int64_t dom_end, mi_op_t *op) {
  MI_NEW_OP(ctx, op, new DenseBlockOp(ctx, ndom, n_gamma, n_gamma_d, gather_idx, Sd, nullptr, index_base, dom_begin, dom_end));
}

int mi_nn_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d, const int64_t *const *gather_idx,
                 const double *const *PiSd, const int64_t *node_gamma_cnt, int index_base, int64_t dom_begin,
                 int64_t dom_end, mi_op_t *op) {
  if (!node_gamma_cnt) return fail(MI_ERR_BAD_ARG, "node_gamma_cnt is NULL");
  MI_NEW_OP(ctx, op, new DenseBlockOp(ctx, ndom, n_gamma, n_gamma_d, gather_idx, PiSd, node_gamma_cnt, index_base, dom_begin, dom_end));
}

int mi_schur_matfree_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d, const int64_t *n_i,
                            const int64_t *const *gather_idx, const int64_t *const *ig_colptr,
                            const int64_t *const *ig_rowval, const double *const *ig_nzval,
                            const int64_t *const *gg_colptr, const int64_t *const *gg_rowval,
                            const double *const *gg_nzval, mi_interior_solve_fn solve, void *user, int index_base,
                            int64_t dom_begin, int64_t dom_end, mi_op_t *op) {
  MI_NEW_OP(ctx, op, new MatfreeSchurOp(ctx, ndom, n_gamma, n_gamma_d, n_i, gather_idx, ig_colptr, ig_rowval, ig_nzval,
                                        gg_colptr, gg_rowval, gg_nzval, solve, user, index_base, dom_begin, dom_end));
}

int mi_schur_matfree_device_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d,
                                   const int64_t *n_i, const int64_t *const *gather_idx,
                                   const int64_t *const *ii_colptr, const int64_t *const *ii_rowval,
                                   const double *const *ii_nzval, const int64_t *const *ig_colptr,
                                   const int64_t *const *ig_rowval, const double *const *ig_nzval,
                                   const int64_t *const *gg_colptr, const int64_t *const *gg_rowval,
                                   const double *const *gg_nzval, double reltol, int index_base, int64_t dom_begin,
                                   int64_t dom_end, mi_op_t *op) {
  if (!ii_colptr) return fail(MI_ERR_BAD_ARG, "A_II arrays are NULL");
  MI_NEW_OP(ctx, op, new MatfreeSchurOp(ctx, ndom, n_gamma, n_gamma_d, n_i, gather_idx, ig_colptr, ig_rowval, ig_nzval,
                                        gg_colptr, gg_rowval, gg_nzval, nullptr, nullptr, index_base, dom_begin, dom_end,
                                        ii_colptr, ii_rowval, ii_nzval, reltol));
}

int mi_schur_global_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_i,
                           const int64_t *const *ig_colptr, const int64_t *const *ig_rowval,
                           const double *const *ig_nzval, const int64_t *gg_colptr, const int64_t *gg_rowval,
                           const double *gg_nzval, mi_interior_solve_fn solve, void *user, int index_base, mi_op_t *op) {
  MI_NEW_OP(ctx, op, new GlobalSchurOp(ctx, ndom, n_gamma, n_i, ig_colptr, ig_rowval, ig_nzval, gg_colptr, gg_rowval,
                                       gg_nzval, solve, user, index_base));
}

int mi_schur_global_device_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_i,
                                  const int64_t *const *ii_colptr, const int64_t *const *ii_rowval,
                                  const double *const *ii_nzval, const int64_t *const *ig_colptr,
                                  const int64_t *const *ig_rowval, const double *const *ig_nzval,
                                  const int64_t *gg_colptr, const int64_t *gg_rowval, const double *gg_nzval,
                                  double reltol, int index_base, mi_op_t *op) {
  if (!ii_colptr) return fail(MI_ERR_BAD_ARG, "mi_schur_global_device_create: A_II arrays are NULL");
  MI_NEW_OP(ctx, op, new GlobalSchurOp(ctx, ndom, n_gamma, n_i, ig_colptr, ig_rowval, ig_nzval, gg_colptr, gg_rowval,
                                       gg_nzval, nullptr, nullptr, index_base, ii_colptr, ii_rowval, ii_nzval, reltol));
}

int mi_op_size(mi_op_t op, int64_t *n) {
  if (!op || !op->impl || !n) return fail(MI_ERR_BAD_ARG, "NULL argument");
  *n = op->impl->n;
  return MI_OK;
}

int mi_op_apply(mi_op_t op, const double *x, double *y) {
  if (!op || !op->impl || !x || !y || x == y) return fail(MI_ERR_BAD_ARG, "mi_op_apply: NULL or aliased argument");
  return guarded([&]() -> int {
    mi_ctx_s *c = op->impl->ctx;
    c->use();
    const size_t n = (size_t)op->impl->n;
    if (c->ptr_mode == MI_PTR_DEVICE) {
      op->impl->apply(x, y, nullptr);
      return MI_OK;
    }
    // Host pointers (the reference's own Julia loop calling mul! / \ every iteration): through pinned buffers, so both
    // copies are plain DMA transfers instead of the runtime's staged pageable copies.
    static const bool trace = env_int("MI355_TRACE", 0) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto tr = [&](const char *what) {
      if (trace) std::fprintf(stderr, "[trace rank %d] mi_op_apply %-22s +%.3f s\n", c->rank, what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    c->pin_b.ensure(n); c->pin_x.ensure(n);
    tr("pinned buffers");
    op->hx.ensure(n); op->hy.ensure(n);
    tr("device buffers");
    std::memcpy(c->pin_b.p, x, n * sizeof(double));
    MI_HIP(hipMemcpyAsync(op->hx.p, c->pin_b.p, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    tr("H2D enqueued");
    if (c->has_comm() || !op->impl->writes_y_once()) {  // a collective or read-modify-write kernels touch y: keep it in device memory
      op->impl->apply(op->hx.p, op->hy.p, nullptr);
      tr("apply enqueued");
      MI_HIP(hipMemcpyAsync(c->pin_x.p, op->hy.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      tr("D2H enqueued");
    } else {
      op->impl->apply(op->hx.p, c->pin_x.p, nullptr);  // the last kernel of the apply writes y straight into pinned memory
    }
    MI_HIP(hipStreamSynchronize(c->stream));
    tr("synchronised");
    std::memcpy(y, c->pin_x.p, n * sizeof(double));
    return MI_OK;
  });
}

int mi_op_bytes(mi_op_t op, int64_t *bytes_apply, int64_t *bytes_dominant_kernel) {
  if (!op || !op->impl) return fail(MI_ERR_BAD_ARG, "op is NULL");
  int64_t a = 0, d = 0;
  op->impl->bytes(&a, &d);
  if (bytes_apply) *bytes_apply = a;
  if (bytes_dominant_kernel) *bytes_dominant_kernel = d;
  return MI_OK;
}

// Launch the dominant kernel `reps` times. With `us_per_launch` the launches are replayed from one graph
// (eager launches of a ~5 us kernel are bound by the host's launch rate) between two HIP events recorded on the
// context's stream, after a warm-up replay; the average duration per launch is returned.
static int dominant_impl(mi_op_t op, const double *x, int reps, double *us_per_launch) {
  if (!op || !op->impl || !x || reps < 0) return fail(MI_ERR_BAD_ARG, "bad argument");
  return guarded([&]() -> int {
    mi_ctx_s *c = op->impl->ctx;
    c->use();
    In xi(c, x, (size_t)op->impl->n, op->hx);
    if (!us_per_launch) {
      for (int i = 0; i < reps; ++i) op->impl->apply_dominant(xi.dev);
      return MI_OK;
    }
    *us_per_launch = 0.0;
    if (reps == 0) return MI_OK;
    hipStream_t s = c->stream;
    hipGraph_t gr = nullptr;
    MI_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try {
      for (int i = 0; i < reps; ++i) op->impl->apply_dominant(xi.dev);
    } catch (...) {
      (void)hipStreamEndCapture(s, &gr);
      if (gr) (void)hipGraphDestroy(gr);
      throw;
    }
    MI_HIP(hipStreamEndCapture(s, &gr));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gr);
    if (e != hipSuccess) raise(MI_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipGraphLaunch(ex, s);           // warm-up replay
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipEventRecord(e0, s);
    if (e == hipSuccess) e = hipGraphLaunch(ex, s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ex);
    if (e != hipSuccess) raise(MI_ERR_HIP, "timed replay failed: %s", hipGetErrorString(e));
    *us_per_launch = (double)ms * 1e3 / reps;
    return MI_OK;
  });
}

int mi_op_apply_dominant(mi_op_t op, const double *x, int reps) { return dominant_impl(op, x, reps, nullptr); }

int mi_op_time_dominant(mi_op_t op, const double *x, int reps, double *us_per_launch) {
  if (!us_per_launch) return fail(MI_ERR_BAD_ARG, "us_per_launch is NULL");
  return dominant_impl(op, x, reps, us_per_launch);
}

int mi_op_destroy(mi_op_t op) {
  if (!op) return MI_OK;
  return guarded([&]() -> int {
    if (op->impl) {
      mi_ctx_s *c = op->impl->ctx;
      c->use();
      (void)hipStreamSynchronize(c->stream);
      for (auto &kv : c->workspaces) kv.second->drop_graphs_of(op->impl.get());
    }
    delete op;
    return MI_OK;
  });
}

// ---------------------------------------------------------------- BLAS-1
int mi_dot(mi_ctx_t ctx, int64_t n, const double *x, const double *y, double *result) {
  if (!ctx || n < 0 || n >= INT32_MAX || !x || !y || !result) return fail(MI_ERR_BAD_ARG, "mi_dot: bad argument");
  return guarded([&]() -> int {
    ctx->use();
    In xi(ctx, x, (size_t)n, ctx->scratch_a), yi(ctx, y, (size_t)n, ctx->scratch_b);
    const int g = vec_grid(n);
    hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(NT), 0, ctx->stream, (int)n, xi.dev, yi.dev, ctx->partials.p,
                       (const int *)nullptr);
    *result = reduce_to_host(ctx, g, 0);
    return MI_OK;
  });
}

int mi_norm2(mi_ctx_t ctx, int64_t n, const double *x, double *result) {
  if (!ctx || n < 0 || n >= INT32_MAX || !x || !result) return fail(MI_ERR_BAD_ARG, "mi_norm2: bad argument");
  return guarded([&]() -> int {
    ctx->use();
    In xi(ctx, x, (size_t)n, ctx->scratch_a);
    const int g = vec_grid(n);
    hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(NT), 0, ctx->stream, (int)n, xi.dev, xi.dev, ctx->partials.p,
                       (const int *)nullptr);
    *result = reduce_to_host(ctx, g, 1);
    return MI_OK;
  });
}

int mi_axpy(mi_ctx_t ctx, int64_t n, double a, const double *x, double *y) {
  if (!ctx || n < 0 || n >= INT32_MAX || !x || !y) return fail(MI_ERR_BAD_ARG, "mi_axpy: bad argument");
  return guarded([&]() -> int {
    ctx->use();
    In xi(ctx, x, (size_t)n, ctx->scratch_a);
    InOut yo(ctx, y, (size_t)n, ctx->scratch_b, true);
    hipLaunchKernelGGL(k_axpy, dim3(vec_grid(n)), dim3(NT), 0, ctx->stream, (int)n, a, xi.dev, yo.dev);
    MI_HIP(hipGetLastError());
    yo.finish();
    return MI_OK;
  });
}

int mi_axpby(mi_ctx_t ctx, int64_t n, double a, const double *x, double b, double *y) {
  if (!ctx || n < 0 || n >= INT32_MAX || !x || !y) return fail(MI_ERR_BAD_ARG, "mi_axpby: bad argument");
  return guarded([&]() -> int {
    ctx->use();
    In xi(ctx, x, (size_t)n, ctx->scratch_a);
    InOut yo(ctx, y, (size_t)n, ctx->scratch_b, true);
    hipLaunchKernelGGL(k_axpby, dim3(vec_grid(n)), dim3(NT), 0, ctx->stream, (int)n, a, xi.dev, b, yo.dev);
    MI_HIP(hipGetLastError());
    yo.finish();
    return MI_OK;
  });
}

// ---------------------------------------------------------------- solvers
int mi_cg(mi_op_t A, const double *b, double *x, int64_t maxit, double eps, double *res_norm, int64_t res_cap,
          int64_t *it) {
  return run_solver(A, nullptr, b, x, nullptr, 0, maxit, eps, res_norm, res_cap, it, false, false);
}
int mi_pcg(mi_op_t A, mi_op_t M, const double *b, double *x, int64_t maxit, double eps, double *res_norm,
           int64_t res_cap, int64_t *it) {
  return run_solver(A, M, b, x, nullptr, 0, maxit, eps, res_norm, res_cap, it, true, false);
}
int mi_defcg(mi_op_t A, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit, double eps,
             double *res_norm, int64_t res_cap, int64_t *it) {
  return run_solver(A, nullptr, b, x, W, nvec, maxit, eps, res_norm, res_cap, it, false, true);
}
int mi_defpcg(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit,
              double eps, double *res_norm, int64_t res_cap, int64_t *it) {
  return run_solver(A, M, b, x, W, nvec, maxit, eps, res_norm, res_cap, it, true, true);
}

int mi_eigcg(mi_op_t A, const double *b, double *x, int64_t nvec, int64_t spdim, int64_t maxit, double eps,
             double *res_norm, int64_t res_cap, int64_t *it, double *V_out) {
  return run_eig_solver(EIGCG, A, nullptr, b, x, nullptr, nvec, spdim, maxit, eps, res_norm, res_cap, it, V_out);
}
int mi_eigpcg(mi_op_t A, mi_op_t M, const double *b, double *x, int64_t nvec, int64_t spdim, int64_t maxit, double eps,
              double *res_norm, int64_t res_cap, int64_t *it, double *V_out) {
  return run_eig_solver(EIGPCG, A, M, b, x, nullptr, nvec, spdim, maxit, eps, res_norm, res_cap, it, V_out);
}
int mi_eigdefcg(mi_op_t A, const double *b, double *x, const double *W, int64_t nvec, int64_t spdim, int64_t maxit,
                double eps, double *res_norm, int64_t res_cap, int64_t *it, double *V_out) {
  return run_eig_solver(EIGDEFCG, A, nullptr, b, x, W, nvec, spdim, maxit, eps, res_norm, res_cap, it, V_out);
}
int mi_eigdefpcg(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec, int64_t spdim,
                 int64_t maxit, double eps, double *res_norm, int64_t res_cap, int64_t *it, double *V_out) {
  return run_eig_solver(EIGDEFPCG, A, M, b, x, W, nvec, spdim, maxit, eps, res_norm, res_cap, it, V_out);
}
int mi_initcg(mi_op_t A, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit, double eps,
              double *res_norm, int64_t res_cap, int64_t *it) {
  return run_init_solver(A, nullptr, b, x, W, nvec, maxit, eps, res_norm, res_cap, it, false);
}
int mi_initpcg(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit, double eps,
               double *res_norm, int64_t res_cap, int64_t *it) {
  return run_init_solver(A, M, b, x, W, nvec, maxit, eps, res_norm, res_cap, it, true);
}

// ---------------------------------------------------------------- on-device block assembly
int mi_assembly_plan_create(mi_ctx_t ctx, int64_t nel, int64_t n_node, const int64_t *cells, int index_base, const double *G,
                            const double *area, const double *ue, const double *be, int64_t n_entries,
                            int64_t n_matrix_entries, const int64_t *cptr, const int64_t *ccode, mi_plan_t *plan) {
  if (!plan) return fail(MI_ERR_BAD_ARG, "plan is NULL");
  *plan = nullptr;
  if (!ctx || nel < 0 || n_node < 0 || n_entries < 0 || n_matrix_entries < 0 || n_matrix_entries > n_entries || !cptr ||
      (nel && (!cells || !G || !area || !ue || !be)) || (index_base != 0 && index_base != 1))
    return fail(MI_ERR_BAD_ARG, "mi_assembly_plan_create: bad argument");
  if (12 * nel >= INT32_MAX) return fail(MI_ERR_BAD_ARG, "mi_assembly_plan_create: 12*nel must fit 32 bits");
  return guarded([&]() -> int {
    ctx->use();
    const int64_t nc = cptr[n_entries];
    if (cptr[0] != 0 || nc < 0 || (nc && !ccode)) return fail(MI_ERR_BAD_ARG, "mi_assembly_plan_create: bad cptr / ccode");
    std::vector<long long> cp((size_t)n_entries + 1);
    for (int64_t k = 0; k <= n_entries; ++k) {
      if (cptr[k] < 0 || cptr[k] > nc || (k && cptr[k] < cptr[k - 1])) return fail(MI_ERR_BAD_ARG, "cptr is not monotone");
      cp[k] = cptr[k];
    }
    std::vector<int> cc((size_t)nc), cl((size_t)3 * nel);
    for (int64_t c = 0; c < nc; ++c) {
      if (ccode[c] < 0 || ccode[c] >= 12 * nel) return fail(MI_ERR_BAD_ARG, "contribution code %lld out of range", (long long)ccode[c]);
      cc[c] = (int)ccode[c];
    }
    for (int64_t k = 0; k < 3 * nel; ++k) cl[k] = to_i32(cells[k] - index_base, 0, n_node, "cells");
    std::unique_ptr<mi_plan_s> p(new mi_plan_s);
    p->ctx = ctx; p->nel = (int)nel; p->n_node = n_node; p->n_entries = n_entries; p->n_matrix = n_matrix_entries; p->n_contrib = nc;
    hipStream_t s = ctx->stream;
    p->cells.upload(cl, s); p->ccode.upload(cc, s); p->cptr.upload(cp, s);
    p->G.upload(G, (size_t)9 * nel, s); p->area.upload(area, (size_t)nel, s);
    p->ue.upload(ue, (size_t)3 * nel, s); p->be.upload(be, (size_t)3 * nel, s);
    p->da.alloc((size_t)nel + 1);
    MI_HIP(hipStreamSynchronize(s));
    *plan = p.release();
    return MI_OK;
  });
}
int mi_assembly_run(mi_plan_t plan, const double *a_nodal, double *values) {
  if (!plan || !a_nodal || (plan->n_entries && !values)) return fail(MI_ERR_BAD_ARG, "mi_assembly_run: NULL argument");
  mi_ctx_s *c = plan->ctx;
  return guarded([&]() -> int {
    c->use();
    In ai(c, a_nodal, (size_t)plan->n_node, plan->a_stage);
    InOut vo(c, values, (size_t)plan->n_entries, plan->out_stage, false);
    auto grid = [](int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + NT - 1) / NT, 1 << 20)); };
    if (plan->nel)
      hipLaunchKernelGGL(k_elem_coeff, dim3(grid(plan->nel)), dim3(NT), 0, c->stream, plan->nel, plan->cells.p, ai.dev, plan->da.p);
    if (plan->n_entries)
      hipLaunchKernelGGL(k_assemble_plan, dim3(grid(plan->n_entries)), dim3(NT), 0, c->stream, (long long)plan->n_entries,
                         (long long)plan->n_matrix, plan->nel, plan->cptr.p, plan->ccode.p, plan->da.p, plan->G.p, plan->area.p,
                         plan->ue.p, plan->be.p, vo.dev);
    MI_HIP(hipGetLastError());
    vo.finish();
    return MI_OK;
  });
}
int mi_assembly_plan_destroy(mi_plan_t plan) {
  if (!plan) return MI_OK;
  return guarded([&]() -> int {
    plan->ctx->use();
    (void)hipStreamSynchronize(plan->ctx->stream);
    delete plan;
    return MI_OK;
  });
}
static MatfreeSchurOp *as_matfree(mi_op_t op) {
  return op && op->impl ? dynamic_cast<MatfreeSchurOp *>(op->impl.get()) : nullptr;
}
int mi_schur_matfree_set_values(mi_op_t op, const double *ii_val, const double *ig_val, const double *gg_val) {
  MatfreeSchurOp *m = as_matfree(op);
  if (!m) return fail(MI_ERR_BAD_ARG, "mi_schur_matfree_set_values: not a matrix-free local-Schur operator");
  mi_ctx_s *c = m->ctx;
  return guarded([&]() -> int {
    c->use();
    DevBuf<double> s1, s2, s3;
    In a(c, ii_val, ii_val && m->icg ? (size_t)m->icg->A.nnz : 0, s1), b(c, ig_val, ig_val ? (size_t)m->A_GI.nnz : 0, s2),
        g(c, gg_val, gg_val ? (size_t)m->A_GG.nnz : 0, s3);
    m->set_values(ii_val ? a.dev : nullptr, ig_val ? b.dev : nullptr, gg_val ? g.dev : nullptr);
    MI_HIP(hipStreamSynchronize(c->stream));  // staging buffers go out of scope
    return MI_OK;
  });
}
static GlobalSchurOp *as_global(mi_op_t op) {
  return op && op->impl ? dynamic_cast<GlobalSchurOp *>(op->impl.get()) : nullptr;
}
int mi_schur_interior_precond(mi_op_t op, int kind) {
  MatfreeSchurOp *m = as_matfree(op);
  GlobalSchurOp *gl = as_global(op);
  InteriorCg *icg = m ? m->icg.get() : gl ? gl->icg.get() : nullptr;
  if (!icg) return fail(MI_ERR_BAD_ARG, "mi_schur_interior_precond: not a Schur operator with the interior solve on the device");
  if (kind != 0 && kind != 1) return fail(MI_ERR_BAD_ARG, "mi_schur_interior_precond: kind must be 0 (none) or 1 (diagonal)");
  return guarded([&]() -> int {
    op->impl->ctx->use();
    icg->set_jacobi(kind == 1);
    return MI_OK;
  });
}
int mi_schur_interior_iterations(mi_op_t op, int64_t *iterations) {
  MatfreeSchurOp *m = as_matfree(op);
  GlobalSchurOp *gl = as_global(op);
  InteriorCg *icg = m ? m->icg.get() : gl ? gl->icg.get() : nullptr;
  if (!icg || !iterations) return fail(MI_ERR_BAD_ARG, "mi_schur_interior_iterations: not a Schur operator with the interior solve on the device");
  *iterations = icg->total_iterations;
  return MI_OK;
}
int mi_schur_matfree_rhs(mi_op_t op, const double *b_I, const double *b_gamma, double *b_schur) {
  MatfreeSchurOp *m = as_matfree(op);
  GlobalSchurOp *gl = as_global(op);
  const int64_t ni_tot = m ? m->ni_tot : gl ? gl->ni_tot : 0;
  if ((!m && !gl) || !b_gamma || !b_schur || (ni_tot && !b_I))
    return fail(MI_ERR_BAD_ARG, "mi_schur_matfree_rhs: not a matrix-free (local or global) Schur operator, or NULL argument");
  mi_ctx_s *c = op->impl->ctx;
  return guarded([&]() -> int {
    c->use();
    DevBuf<double> s1;
    In bi(c, b_I, (size_t)ni_tot, s1), bg(c, b_gamma, (size_t)op->impl->n, c->scratch_a);
    InOut out(c, b_schur, (size_t)op->impl->n, c->scratch_b, false);
    if (m) m->schur_rhs(bi.dev, bg.dev, out.dev); else gl->schur_rhs(bi.dev, bg.dev, out.dev);
    out.finish();
    MI_HIP(hipStreamSynchronize(c->stream));
    return MI_OK;
  });
}

int mi_schur_matfree_interior_solutions(mi_op_t op, const double *u_gamma, const double *b_I, double *u_I) {
  MatfreeSchurOp *m = as_matfree(op);
  GlobalSchurOp *gl = as_global(op);
  const int64_t ni_tot = m ? m->ni_tot : gl ? gl->ni_tot : 0;
  if ((!m && !gl) || !u_gamma || (ni_tot && (!b_I || !u_I)))
    return fail(MI_ERR_BAD_ARG, "mi_schur_matfree_interior_solutions: not a matrix-free (local or global) Schur operator, or NULL argument");
  mi_ctx_s *c = op->impl->ctx;
  return guarded([&]() -> int {
    c->use();
    DevBuf<double> s1, s2;
    In ug(c, u_gamma, (size_t)op->impl->n, c->scratch_a), bi(c, b_I, (size_t)ni_tot, s1);
    InOut out(c, u_I, (size_t)ni_tot, s2, false);
    if (m) m->interior_solutions(ug.dev, bi.dev, out.dev); else gl->interior_solutions(ug.dev, bi.dev, out.dev);
    out.finish();
    MI_HIP(hipStreamSynchronize(c->stream));
    return MI_OK;
  });
}

// ---------------------------------------------------------------- set-up of the assembled mode on the device
int mi_schur_setup_create(mi_ctx_t ctx, int64_t ndom, const int64_t *n_gamma_d, const int64_t *n_i,
                          const int64_t *const *ii_colptr, const int64_t *const *ii_rowval, const int64_t *const *ig_colptr,
                          const int64_t *const *ig_rowval, const int64_t *const *gg_colptr, const int64_t *const *gg_rowval,
                          int index_base, mi_setup_t *plan) {
  if (!plan) return fail(MI_ERR_BAD_ARG, "plan is NULL");
  *plan = nullptr;
  if (!ctx || ndom <= 0 || !n_gamma_d || !n_i || !ii_colptr || !ii_rowval || !ig_colptr || !ig_rowval || !gg_colptr || !gg_rowval ||
      (index_base != 0 && index_base != 1))
    return fail(MI_ERR_BAD_ARG, "mi_schur_setup_create: bad argument");
  return guarded([&]() -> int {
    ctx->use();
    std::unique_ptr<mi_setup_s> p(new mi_setup_s);
    setup_plan_build(*p, ctx, ndom, n_gamma_d, n_i, ii_colptr, ii_rowval, ig_colptr, ig_rowval, gg_colptr, gg_rowval, index_base);
    *plan = p.release();
    return MI_OK;
  });
}
int mi_schur_setup_run(mi_setup_t plan, const double *ii_val, const double *ig_val, const double *gg_val, const double *b_I,
                       double *Sd, double *w) {
  if (!plan || !ig_val || !gg_val || !Sd || (plan->n_ii && !ii_val)) return fail(MI_ERR_BAD_ARG, "mi_schur_setup_run: NULL argument");
  mi_ctx_s *c = plan->ctx;
  return guarded([&]() -> int {
    c->use();
    In a(c, ii_val, (size_t)plan->n_ii, plan->st_ii), b(c, ig_val, (size_t)plan->n_ig, plan->st_ig), g(c, gg_val, (size_t)plan->n_gg, plan->st_gg),
        bi(c, b_I, b_I ? (size_t)plan->n_bi : 0, plan->st_bi);
    InOut so(c, Sd, (size_t)plan->n_s, plan->st_S, false), wo(c, w, w ? (size_t)plan->n_w : 0, plan->st_w, false);
#ifdef MI355_EXPERIMENTAL
    if (!plan->lanes.empty()) setup_plan_run(*plan, a.dev, b.dev, g.dev, b_I ? bi.dev : nullptr, so.dev, w ? wo.dev : nullptr);   // MI355_SETUP_LIB=1
    else
#endif
    gj_run(*plan, a.dev, b.dev, g.dev, b_I ? bi.dev : nullptr, so.dev, w ? wo.dev : nullptr);
    so.finish();
    wo.finish();
    if (c->ptr_mode != MI_PTR_DEVICE) {   // host mode is synchronous: a break-down is reported
      // Gauss-Jordan route (the default): no pivoting, so an interior block that is not positive definite shows as Inf / NaN
      // in S_d (the reference's CHOLMOD would throw PosDefException). Device-pointer mode stays asynchronous: there the
      // next mi_nn_pinv reports non-finite blocks.
      for (size_t i = 0; i < (size_t)plan->n_s; ++i)
        if (!std::isfinite(Sd[i])) return fail(MI_ERR_SINGULAR, "mi_schur_setup_run: S_d is not finite (an interior block is singular or not positive definite)");
      for (auto &l : plan->lanes) {
        int info[2] = {0, 0};
        memcpy_sync(info, l.info.p, sizeof info, hipMemcpyDeviceToHost);
        if (info[0] || info[1]) return fail(MI_ERR_SINGULAR, "mi_schur_setup_run: an interior block is not positive definite (potrf info %d / %d)", info[0], info[1]);
      }
    }
    return MI_OK;
  });
}
int mi_schur_setup_keep_levels(mi_setup_t plan, int on) {
  if (!plan) return fail(MI_ERR_BAD_ARG, "plan is NULL");
  return guarded([&]() -> int {
    plan->ctx->use();
#ifdef MI355_EXPERIMENTAL
    if (!plan->lanes.empty()) return fail(MI_ERR_BAD_ARG, "level solves belong to the Gauss-Jordan set-up (not MI355_SETUP_LIB=1)");
#endif
    gj_set_keep(*plan, on != 0);
    return MI_OK;
  });
}
int mi_schur_setup_interior_solve(mi_setup_t plan, const double *f, double *u) {
  if (!plan || !f || !u) return fail(MI_ERR_BAD_ARG, "mi_schur_setup_interior_solve: NULL argument");
  mi_ctx_s *c = plan->ctx;
  return guarded([&]() -> int {
    c->use();
    DevBuf<double> s1, s2;
    In fi(c, f, (size_t)plan->n_bi, s1);
    InOut uo(c, u, (size_t)plan->n_bi, s2, false);
    gj_level_solve(*plan, fi.dev, uo.dev);
    uo.finish();
    if (c->ptr_mode != MI_PTR_DEVICE) MI_HIP(hipStreamSynchronize(c->stream));
    return MI_OK;
  });
}
int mi_schur_matfree_interior_levels(mi_op_t op, mi_setup_t plan) {
  if (!op || !op->impl) return fail(MI_ERR_BAD_ARG, "op is NULL");
  return guarded([&]() -> int {
    MatfreeSchurOp *mf = dynamic_cast<MatfreeSchurOp *>(op->impl.get());
    GlobalSchurOp *gs = dynamic_cast<GlobalSchurOp *>(op->impl.get());
    if (!mf && !gs) return fail(MI_ERR_BAD_ARG, "mi_schur_matfree_interior_levels: not a matrix-free Schur operator");
    std::function<void(const double *, double *)> fn;
    if (plan) {
      if (plan->ctx != op->impl->ctx) return fail(MI_ERR_BAD_ARG, "plan and operator live on different contexts");
      const size_t ni_tot = mf ? mf->ni_tot : gs->ni_tot;
      const int nd = mf ? mf->maps.ndl : gs->ndom;
      if ((long long)ni_tot != plan->n_bi || nd != plan->ndom)
        return fail(MI_ERR_BAD_ARG, "plan and operator differ in their subdomains (%d with %lld interior nodes vs %d with %lld)", plan->ndom, plan->n_bi, nd, (long long)ni_tot);
      fn = [plan](const double *f, double *u) { gj_level_solve(*plan, f, u); };
    }
    if (mf) mf->level_solver = fn; else gs->level_solver = fn;
    return MI_OK;
  });
}
int mi_schur_setup_destroy(mi_setup_t plan) {
  if (!plan) return MI_OK;
  return guarded([&]() -> int {
    plan->ctx->use();
    (void)hipStreamSynchronize(plan->ctx->stream);
    delete plan;
    return MI_OK;
  });
}
int mi_nn_pinv(mi_ctx_t ctx, int64_t ndom, const int64_t *n_gamma_d, const double *Sd, double rtol, double *PiSd) {
  if (!ctx || ndom <= 0 || !n_gamma_d || !Sd || !PiSd) return fail(MI_ERR_BAD_ARG, "mi_nn_pinv: bad argument");
  return guarded([&]() -> int {
    ctx->use();
    size_t tot = 0;
    for (int64_t d = 0; d < ndom; ++d) {
      if (n_gamma_d[d] < 0 || n_gamma_d[d] >= 46000) return fail(MI_ERR_BAD_ARG, "mi_nn_pinv: bad block size");
      tot += (size_t)n_gamma_d[d] * n_gamma_d[d];
    }
    if (rtol <= 0.0) rtol = std::sqrt(2.220446049250313e-16);   // sqrt(eps(Float64)), EPDD.jl:1211
    DevBuf<double> s1, s2;
    In si(ctx, Sd, tot, s1);
    InOut po(ctx, PiSd, tot, s2, false);
    pinv_blocks_fast(ctx, (int)ndom, n_gamma_d, si.dev, rtol, po.dev);
    po.finish();
    return MI_OK;
  });
}
int mi_dense_set_blocks(mi_op_t op, const double *blocks) {
  DenseBlockOp *dop = op && op->impl ? op->impl->as_dense() : nullptr;
  if (!dop || !blocks) return fail(MI_ERR_BAD_ARG, "mi_dense_set_blocks: not an assembled-Schur / Neumann-Neumann operator, or NULL blocks");
  mi_ctx_s *c = dop->ctx;
  return guarded([&]() -> int {
    c->use();
    size_t tot = 0;
    for (int dl = 0; dl < dop->maps.ndl; ++dl) if (dop->owned_h[dl]) tot += (size_t)dop->maps.nd[dl] * dop->maps.nd[dl];
    DevBuf<double> st;
    In bi(c, blocks, tot, st);
    dop->set_blocks(bi.dev);
    if (c->ptr_mode != MI_PTR_DEVICE) MI_HIP(hipStreamSynchronize(c->stream));   // the staging buffer goes out of scope
    return MI_OK;
  });
}

// ---------------------------------------------------------------- events
int mi_event_create(mi_event_t *ev) {
  if (!ev) return fail(MI_ERR_BAD_ARG, "ev is NULL");
  return guarded([&]() -> int {
    std::unique_ptr<mi_event_s> e(new mi_event_s);
    MI_HIP(hipEventCreate(&e->ev));
    *ev = e.release();
    return MI_OK;
  });
}
int mi_event_record(mi_ctx_t ctx, mi_event_t ev) {
  if (!ctx || !ev) return fail(MI_ERR_BAD_ARG, "NULL argument");
  return guarded([&]() -> int { ctx->use(); MI_HIP(hipEventRecord(ev->ev, ctx->stream)); return MI_OK; });
}
int mi_event_elapsed_ms(mi_event_t start, mi_event_t stop, double *ms) {
  if (!start || !stop || !ms) return fail(MI_ERR_BAD_ARG, "NULL argument");
  return guarded([&]() -> int {
    MI_HIP(hipEventSynchronize(stop->ev));
    float t = 0.f;
    MI_HIP(hipEventElapsedTime(&t, start->ev, stop->ev));
    *ms = (double)t;
    return MI_OK;
  });
}
int mi_event_destroy(mi_event_t ev) {
  if (!ev) return MI_OK;
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return MI_OK;
}

}  // extern "C"
