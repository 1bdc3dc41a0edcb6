// Γ-sum across ranks without a host in the loop: the one-shot peer exchange.
//
// The reference's own sketch of the distributed Schur apply is a `(+)` reduction over subdomains
// (`Fem/EllipticPdePllDomainDecomposition.jl:10-14`: `@sync @distributed (+) for idom`). Here every rank owns an ARENA of
// peer-visible device memory (fine-grained when the runtime grants it); the arenas' base addresses are exchanged once
// (same process: plain pointers; other processes of the node: `hipIpcGetMemHandle` / `hipIpcOpenMemHandle`, carried by
// whatever transport the host code has — bench.py uses torch.distributed). Tables live at the SAME offset in every arena
// (collective, deterministic bump allocation: sharded operators are created in the same order with the same sizes on all
// ranks), so a rank addresses a peer's table as `base[q] + offset`.
//
// One exchange = every rank WRITES its entries into every rank's table (xGMI peer stores; its own copy included), then
// releases them at system scope and stores its exchange number `e` into flag `rank` of every arena; a rank continues
// when all flags of its own arena have reached `e`. Nothing is added on the way for the dense operators: their slot
// tables are disjoint unions over the ranks (operators.hpp), so the result has the bits of the single-GPU loop.
// Tables are double-buffered by the parity of `e`: copy e&1 is rewritten by exchange e+2, which a rank can only start
// after every rank has signalled e+1, i.e. (stream order) after every rank has finished reading copy e&1.
// Waits are bounded (wall clock): an expired wait sets `err` in the context's exchange state, the solve then fails with
// MI_ERR_COMM instead of hanging.
//
// What this file cannot know (no multi-GPU box was available to any round): the cost and ordering of the peer stores
// and flag stores on real xGMI. On one GPU ("in-process ranks": several contexts of one process, one host thread and one
// stream each) every arena is local memory and the protocol — epochs, parity, bounded waits, graph capture — is what runs.
#pragma once
#include <chrono>

#include "common.hpp"
#include "exchange_dev.hpp"

namespace mi {

// ---- table exchange: this rank's entries of `src` (list own_idx, ascending) into copy (e & 1) of the table in EVERY arena,
// signal, wait. One launch of (chunks × destinations) workgroups: workgroup (c, q) stores chunk c of the entries into
// arena q — the stores are 8-byte scatters (slots of 32-byte rows), one workgroup issues about 1.4 of them per ns
// (tools/probes/xchg_probe.hip), and there are n_own × n ≈ the whole table of them whatever the number of ranks. Each
// workgroup publishes its stores (barrier, then ONE system-scope fence: 16 waves fencing cost 4 µs more, same probe)
// and counts itself in; the last one to arrive signals the flags and waits for the peers'.
constexpr int XCHG_PUSH_CHUNK = 4096;
__global__ __launch_bounds__(1024) void k_xchg_push(XchgPeers P, XchgState *st, size_t table_off, size_t copy_doubles,
                                                    const double *__restrict__ src, const int *__restrict__ own_idx, int n_own,
                                                    const int *done) {
  if (done && *done) return;
  __shared__ unsigned long long e_sh;
  __shared__ int last_sh;
  const unsigned long long e_next = st->epoch + 1;       // (the last arriver advances it, after every workgroup has read it)
  const size_t par = (size_t)(e_next & 1) * copy_doubles;
  {
    double *dst = reinterpret_cast<double *>(P.arena[blockIdx.y] + table_off) + par;
    const int i0 = blockIdx.x * XCHG_PUSH_CHUNK + threadIdx.x;
    int id[4];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = i0 + k * 1024; id[k] = i < n_own ? own_idx[i] : -1; }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = id[k] >= 0 ? src[id[k]] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (id[k] >= 0) dst[id[k]] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned int total = gridDim.x * gridDim.y;
    const unsigned int prev = total > 1 ? __hip_atomic_fetch_add(&st->arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    last_sh = prev == total - 1;
    if (last_sh) {
      __hip_atomic_store(&st->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      e_sh = xchg_signal(st, P);
    }
  }
  __syncthreads();
  if (last_sh && threadIdx.x < 64) { xchg_wait(st, P, e_sh); __atomic_thread_fence(__ATOMIC_ACQUIRE); }
}

// ---- generic all-reduce (sum) through per-rank staging slots [2][n_ranks][cap] in every arena: copy, signal + wait, sum
__global__ __launch_bounds__(256) void k_xchg_stage(XchgPeers P, const XchgState *st, size_t stage_off, size_t cap,
                                                    const double *__restrict__ send, size_t n, const int *done) {
  if (done && *done) return;
  const size_t slot = ((size_t)((st->epoch + 1) & 1) * P.n + P.rank) * cap;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double v = send[i];
    for (int q = 0; q < P.n; ++q) reinterpret_cast<double *>(P.arena[q] + stage_off)[slot + i] = v;
  }
  __threadfence_system();
}
__global__ __launch_bounds__(64) void k_xchg_signal_wait(XchgPeers P, XchgState *st, const int *done) {
  if (done && *done) return;
  __shared__ unsigned long long e_sh;
  __threadfence_system();
  if (threadIdx.x == 0) e_sh = xchg_signal(st, P);
  __syncthreads();
  xchg_wait(st, P, e_sh);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
__global__ __launch_bounds__(256) void k_xchg_sum(XchgPeers P, const XchgState *st, size_t stage_off, size_t cap,
                                                  double *__restrict__ recv, size_t n, const int *done) {
  if (done && *done) return;
  const double *stage = reinterpret_cast<const double *>(P.arena[P.rank] + stage_off) + (size_t)(st->epoch & 1) * P.n * cap;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double s = stage[i];
    for (int q = 1; q < P.n; ++q) s += stage[(size_t)q * cap + i];   // rank order, as the loopback group and a ring's result
    recv[i] = s;
  }
}
// copy (epoch & 1) of a table -> a fixed local buffer (consumers whose views carry no parity)
__global__ __launch_bounds__(256) void k_xchg_settle(const XchgState *st, const double *__restrict__ table, size_t copy_doubles,
                                                     double *__restrict__ dst, size_t n, const int *done) {
  if (done && *done) return;
  const double *src = table + (size_t)(st->epoch & 1) * copy_doubles;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// the wait half of an exchange whose stores and flags came out of the producing launch itself (k_gemv_pcg): one wave
// (4.8 us with the stop flag, the counter, the error word and the flag read one after the other; all four are requested at
// once here and the clock is only read when a flag is not there yet)
__global__ __launch_bounds__(64) void k_xchg_wait_advance(XchgPeers P, XchgState *st, const int *done) {
  const int q = threadIdx.x;
  const int d = done ? *done : 0;
  const unsigned long long e = st->epoch + 1;
  const int err = st->err;
  const unsigned long long seen = q < P.n ? __hip_atomic_load(xchg_flag(P, P.rank, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0ull;
  asm volatile("" ::"v"(d), "v"(err), "v"(seen), "s"(e));
  if (d) return;                 // (the producing launch took the same early exit: nothing was signalled)
  if (err || seen < e) xchg_wait(st, P, e);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (threadIdx.x == 0) st->epoch = e;
}

// in-launch waits (PcgFold::x_inwait): the folded loop of a solve starts from the exchange number as it stands now
__global__ void k_xchg_begin(XchgState *st) { st->xep[0] = st->epoch; st->xep[1] = st->epoch; }
__global__ void k_xchg_set_abort(XchgState *st, int *flag) { st->abort_done = flag; }

// ------------------------------------------------------------------ host side
struct PeerComm {
  int n = 1, rank = 0, device = 0;
  char *arena = nullptr;
  size_t arena_bytes = 0, bump = 0;
  bool fine_grained = false, ready = false;
  bool owns_arena = true;            // false: an in-process group keeps the arena alive until every rank is gone
  std::vector<char *> base;          // arena of every rank (base[rank] == arena)
  std::vector<void *> ipc_opened;    // mappings to close
  XchgState *st = nullptr;
  XchgPeers peers{};
  size_t stage_off = 0, stage_cap = 0;

  static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
  void init(int device_, int rank_, int n_, size_t bytes) {
    if (n_ < 1 || n_ > XCHG_MAX_RANKS || rank_ < 0 || rank_ >= n_) raise(MI_ERR_BAD_ARG, "peer exchange: bad rank %d of %d (at most %d ranks)", rank_, n_, XCHG_MAX_RANKS);
    device = device_; rank = rank_; n = n_;
    arena_bytes = align_up(std::max<size_t>(bytes, XCHG_FLAG_BYTES + (1u << 20)));
    const char *fg = std::getenv("MI355_PEER_FINEGRAINED");
    if (!fg || std::atoi(fg) != 0) {
      void *p = nullptr;
      if (hipExtMallocWithFlags(&p, arena_bytes, hipDeviceMallocFinegrained) == hipSuccess) { arena = (char *)p; fine_grained = true; }
      else (void)hipGetLastError();
    }
    if (!arena) MI_HIP(hipMalloc((void **)&arena, arena_bytes));
    memset_sync(arena, 0, arena_bytes);
    MI_HIP(hipMalloc((void **)&st, sizeof(XchgState)));
    memset_sync(st, 0, sizeof(XchgState));
    base.assign(n, nullptr);
    base[rank] = arena;
    bump = XCHG_FLAG_BYTES;
  }
  void export_handle(void *handle64) const {
    static_assert(sizeof(hipIpcMemHandle_t) <= MI_PEER_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof h);
    MI_HIP(hipIpcGetMemHandle(&h, arena));
    std::memset(handle64, 0, MI_PEER_HANDLE_BYTES);
    std::memcpy(handle64, &h, sizeof h);
  }
  void import_peer(int q, const void *handle64, void *same_process_base) {
    if (q < 0 || q >= n) raise(MI_ERR_BAD_ARG, "peer exchange: rank %d outside [0, %d)", q, n);
    if (q == rank) return;
    if (same_process_base) { base[q] = (char *)same_process_base; return; }
    if (!handle64) raise(MI_ERR_BAD_ARG, "peer exchange: no handle for rank %d", q);
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    void *p = nullptr;
    MI_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));   // (the flag enables peer access between the two devices as needed)
    ipc_opened.push_back(p);
    // a mapping this device's kernels cannot reach would fault inside a launch — a refusal here lets the caller fall back
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.device >= 0 && at.device != device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, device, at.device) == hipSuccess && !can)
        raise(MI_ERR_COMM, "peer exchange: device %d cannot access the memory of device %d (rank %d)", device, at.device, q);
    } else {
      (void)hipGetLastError();
    }
    base[q] = (char *)p;
  }
  XchgPeers *peers_dev = nullptr;    // the same structure in device memory (the folded launches read it through a pointer)
  void set_timeout_ms(long long ms) { set_timeout_ticks(ms * 100000ll); }   // 100 MHz ticks
  void set_timeout_ticks(long long ticks) {
    peers.timeout = ticks;
    if (peers_dev) memcpy_sync(peers_dev, &peers, sizeof peers, hipMemcpyHostToDevice);
  }
  void finish() {
    for (int q = 0; q < n; ++q) if (!base[q]) raise(MI_ERR_COMM, "peer exchange: the arena of rank %d was never imported", q);
    peers = XchgPeers{};
    peers.n = n; peers.rank = rank;
    const char *t = std::getenv("MI355_PEER_TIMEOUT_MS");
    set_timeout_ms(t && *t ? std::atoll(t) : 60000);   // generous: ranks reach their first exchange seconds apart (host-side set-up)
    for (int q = 0; q < n; ++q) peers.arena[q] = base[q];
    if (!peers_dev) MI_HIP(hipMalloc((void **)&peers_dev, sizeof(XchgPeers)));
    memcpy_sync(peers_dev, &peers, sizeof peers, hipMemcpyHostToDevice);
    ready = true;
  }
  // Collective and deterministic: every rank calls it in the same order with the same size. Returns the offset.
  size_t alloc(size_t bytes) {
    const size_t off = bump, need = align_up(bytes);
    if (off + need > arena_bytes)
      raise(MI_ERR_COMM, "peer exchange: arena exhausted (%zu of %zu bytes in use, %zu more asked; MI355_PEER_ARENA_MB)", off, arena_bytes, need);
    bump = off + need;
    return off;
  }
  double *local(size_t off) const { return reinterpret_cast<double *>(arena + off); }
  // staging of the generic all-reduce: grows by bump allocation (collective: every rank reduces the same sizes in the same order)
  void reserve_stage(size_t n_doubles) {
    if (n_doubles <= stage_cap) return;
    const size_t cap = std::max<size_t>(n_doubles, 4096);
    stage_off = alloc(2 * (size_t)n * cap * sizeof(double));
    stage_cap = cap;
  }
  void allreduce(const double *send, double *recv, size_t cnt, hipStream_t s, const int *done = nullptr) {
    if (!ready) raise(MI_ERR_COMM, "peer exchange used before every arena was imported");
    if (cnt == 0) return;
    if (cnt > stage_cap) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(s, &cap);
      if (cap != hipStreamCaptureStatusNone) raise(MI_ERR_COMM, "peer exchange: staging for %zu doubles must be reserved before stream capture", cnt);
      reserve_stage(cnt);
    }
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((cnt + 255) / 256, 256));
    hipLaunchKernelGGL(k_xchg_stage, dim3(grid), dim3(256), 0, s, peers, st, stage_off, stage_cap, send, cnt, done);
    hipLaunchKernelGGL(k_xchg_signal_wait, dim3(1), dim3(64), 0, s, peers, st, done);
    hipLaunchKernelGGL(k_xchg_sum, dim3(grid), dim3(256), 0, s, peers, st, stage_off, stage_cap, recv, cnt, done);
    MI_HIP(hipGetLastError());
  }
  void set_abort_flag(int *done_flag, hipStream_t s) {   // stream-ordered: around the launches of one solve
    hipLaunchKernelGGL(k_xchg_set_abort, dim3(1), dim3(1), 0, s, st, done_flag);
    MI_HIP(hipGetLastError());
  }
  void push(size_t table_off, size_t copy_doubles, const double *src, const int *own_idx, int n_own, hipStream_t s, const int *done) {
    if (!ready) raise(MI_ERR_COMM, "peer exchange used before every arena was imported");
    const int chunks = std::max(1, (n_own + XCHG_PUSH_CHUNK - 1) / XCHG_PUSH_CHUNK);   // (a rank without entries still signals and waits)
    hipLaunchKernelGGL(k_xchg_push, dim3(chunks, n), dim3(1024), 0, s, peers, st, table_off, copy_doubles, src, own_idx, n_own, done);
    MI_HIP(hipGetLastError());
  }
  void begin_inwait(hipStream_t s) {
    hipLaunchKernelGGL(k_xchg_begin, dim3(1), dim3(1), 0, s, st);
    MI_HIP(hipGetLastError());
  }
  void wait_advance(hipStream_t s, const int *done) {
    if (!ready) raise(MI_ERR_COMM, "peer exchange used before every arena was imported");
    hipLaunchKernelGGL(k_xchg_wait_advance, dim3(1), dim3(64), 0, s, peers, st, done);
    MI_HIP(hipGetLastError());
  }
  // 0, or 1 when a wait has expired since the last call (cleared); synchronises the stream
  std::string err_text;
  int take_error(hipStream_t s) {
    XchgState h{};
    MI_HIP(hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (!h.err) return 0;
    char buf[256];
    std::snprintf(buf, sizeof buf, "rank %d of %d waited for rank %d at exchange %llu (its flag stood at %llu; this rank has signalled %llu)", rank, n,
                  h.err_rank, h.err_epoch, h.err_seen, h.epoch);
    err_text = buf;
    MI_HIP(hipMemsetAsync(&st->err, 0, sizeof(int), s));
    MI_HIP(hipStreamSynchronize(s));
    return 1;
  }
  ~PeerComm() {
    for (void *p : ipc_opened) (void)hipIpcCloseMemHandle(p);
    if (arena && owns_arena) (void)hipFree(arena);
    if (st) (void)hipFree(st);
    if (peers_dev) (void)hipFree(peers_dev);
  }
};

}  // namespace mi
