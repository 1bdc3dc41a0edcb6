// cg / pcg / defcg / defpcg with the whole loop on the device (RecyclingKrylovSolvers/cg.jl, defcg.jl).
//
// One iteration is a fixed sequence of kernel launches whose scalars (alpha, beta, r'r, r'z, it,
// the stop flag) live in HBM (`SolverState`), so the host never reads a scalar inside the loop.
// `chunk` iterations are captured once into a hipGraph and replayed; after every replay the host
// looks at the stop flag of the PREVIOUS replay (one replay of run-ahead). Launches that follow the
// iteration at which the stop rule fired return immediately (`done`), so the iterates, `it` and
// `res_norm` are exactly those of the reference's `while (it < maxit) && (res_norm[it] > tol)`.
#pragma once
#include <cmath>
#include <tuple>

#include "operators.hpp"

namespace mi {

struct PinnedFlags {
  long long it;
  int done;
  int overflow;
};

struct GraphKey {
  const Operator *A, *M;
  int nvec, chunk;
  bool operator<(const GraphKey &o) const {
    return std::tie(A, M, nvec, chunk) < std::tie(o.A, o.M, o.nvec, o.chunk);
  }
};

struct SolverWorkspace {
  int64_t n = 0;
  int g = 1;  // workgroups of the vector kernels = number of partials per dot
  DevBuf<double> r, z, p, Ap, x, b;
  DevBuf<double> part_pAp, part_rr, part_rz, part_bb, res_norm;
  DevBuf<SolverState> st;
  // deflation
  DevBuf<double> W, AW, LU, mu, part_mu, gram;
  DevBuf<int> piv;
  int nvec_cap = 0;
  PinnedFlags *flags = nullptr;  // 2 slots, pinned
  hipEvent_t ev[2] = {nullptr, nullptr};
  std::map<GraphKey, hipGraphExec_t> graphs;

  explicit SolverWorkspace(int64_t n_) : n(n_), g(vec_grid(n_)) {
    const size_t m = (size_t)n + 2;
    r.alloc(m); z.alloc(m); p.alloc(m); Ap.alloc(m); x.alloc(m); b.alloc(m);
    part_pAp.alloc(MAX_PARTS); part_rr.alloc(MAX_PARTS); part_rz.alloc(MAX_PARTS); part_bb.alloc(MAX_PARTS);
    st.alloc(1);
    MI_HIP(hipHostMalloc((void **)&flags, 2 * sizeof(PinnedFlags)));
    for (auto &e : ev) MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ~SolverWorkspace() {
    drop_graphs();
    if (flags) (void)hipHostFree(flags);
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
  }
  void drop_graphs() {
    for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
  }
  void drop_graphs_of(const Operator *op) {
    for (auto it = graphs.begin(); it != graphs.end();)
      if (it->first.A == op || it->first.M == op) { (void)hipGraphExecDestroy(it->second); it = graphs.erase(it); }
      else ++it;
  }
  void ensure_deflation(int nvec) {
    if (nvec <= nvec_cap) return;
    drop_graphs();  // buffers move
    W.alloc((size_t)n * nvec); AW.alloc((size_t)n * nvec);
    LU.alloc((size_t)nvec * nvec); gram.alloc((size_t)nvec * nvec);
    mu.alloc(nvec); piv.alloc(nvec); part_mu.alloc((size_t)nvec * MAX_PARTS);
    nvec_cap = nvec;
  }
};

inline SolverWorkspace &workspace(mi_ctx_s *ctx, int64_t n) {
  auto &slot = ctx->workspaces[n];
  if (!slot) slot.reset(new SolverWorkspace(n));
  return *slot;
}

// LAPACK getrf (unblocked, partial pivoting) on a column-major copy. Returns 0 or k+1 for U[k,k]==0.
inline int host_lu(int n, std::vector<double> &a, std::vector<int> &piv) {
  piv.resize(n);
  for (int k = 0; k < n; ++k) {
    int p = k;
    double mx = std::fabs(a[k + (size_t)k * n]);
    for (int i = k + 1; i < n; ++i) {
      const double v = std::fabs(a[i + (size_t)k * n]);
      if (v > mx) { mx = v; p = i; }
    }
    piv[k] = p;
    if (a[p + (size_t)k * n] == 0.0 || !std::isfinite(a[p + (size_t)k * n])) return k + 1;
    if (p != k)
      for (int j = 0; j < n; ++j) std::swap(a[k + (size_t)j * n], a[p + (size_t)j * n]);
    const double piv_inv = 1.0 / a[k + (size_t)k * n];
    for (int i = k + 1; i < n; ++i) a[i + (size_t)k * n] *= piv_inv;
    for (int j = k + 1; j < n; ++j) {
      const double akj = a[k + (size_t)j * n];
      for (int i = k + 1; i < n; ++i) a[i + (size_t)j * n] -= a[i + (size_t)k * n] * akj;
    }
  }
  return 0;
}

struct Krylov {
  mi_ctx_s *ctx;
  Operator *A, *M;  // M == nullptr: unpreconditioned (z is r)
  SolverWorkspace &ws;
  int nvec;
  int n, g;
  hipStream_t s;

  Krylov(mi_ctx_s *c, Operator *A_, Operator *M_, int nvec_)
      : ctx(c), A(A_), M(M_), ws(workspace(c, A_->n)), nvec(nvec_), n((int)A_->n), g(ws.g), s(c->stream) {}

  const int *done() const { return &ws.st.p->done; }

  void dot_partial(const double *x, const double *y, double *part, const int *dn) {
    hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(NT), 0, s, n, x, y, part, dn);
    MI_HIP(hipGetLastError());
  }
  // mu = WtAW \ (V' v), V = AW (loop) or W (set-up)
  void project(const double *V, const double *v, const int *dn) {
    hipLaunchKernelGGL(k_multi_dot_partial, dim3(g, nvec), dim3(NT), 0, s, n, V, v, ws.part_mu.p, dn);
    hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(64), sizeof(double) * nvec, s, nvec, ws.LU.p, ws.piv.p, ws.part_mu.p, g,
                       ws.mu.p, dn);
    MI_HIP(hipGetLastError());
  }

  // One loop iteration (cg.jl:35-47 / 92-106; defcg.jl:68-80 / 291-305), enqueued on the stream.
  void iteration() {
    const int *dn = done();
    const int pre = M != nullptr;
    A->apply(ws.p.p, ws.Ap.p, dn);                                   // mul!(Ap, A, p)
    dot_partial(ws.p.p, ws.Ap.p, ws.part_pAp.p, dn);                 // d = dot(p, Ap)
    hipLaunchKernelGGL(k_update_xr, dim3(g), dim3(NT), 0, s, n, ws.st.p, ws.part_pAp.p, g, ws.p.p, ws.Ap.p, ws.x.p,
                       ws.r.p, ws.part_rr.p, pre);                   // alpha; x += alpha p; r -= alpha Ap; r'r
    const double *zz = ws.r.p;
    if (pre) {
      M->apply(ws.r.p, ws.z.p, dn);                                  // z .= M \ r
      dot_partial(ws.r.p, ws.z.p, ws.part_rz.p, dn);                 // rTz = dot(r, z)
      zz = ws.z.p;
    }
    if (nvec > 0) project(ws.AW.p, zz, dn);                          // mu .= WtAW \ (WtA * z)
    hipLaunchKernelGGL(k_update_p, dim3(g), dim3(NT), 0, s, n, ws.st.p, ws.part_rr.p, ws.part_rz.p, g, zz, ws.p.p,
                       nvec > 0 ? ws.W.p : (const double *)nullptr, nvec > 0 ? ws.mu.p : (const double *)nullptr, nvec,
                       ws.res_norm.p, pre);                          // beta; p; it += 1; res_norm[it]; stop rule
    MI_HIP(hipGetLastError());
  }

  hipGraphExec_t graph(int chunk) {
    GraphKey key{A, M, nvec, chunk};
    auto it = ws.graphs.find(key);
    if (it != ws.graphs.end()) return it->second;
    hipGraph_t gr = nullptr;
    MI_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try {
      for (int k = 0; k < chunk; ++k) iteration();
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
    ws.graphs[key] = ex;
    return ex;
  }

  void fetch_flags(int slot) {
    // it/done/overflow are contiguous at the end of SolverState
    MI_HIP(hipMemcpyAsync(&ws.flags[slot].it, &ws.st.p->it, sizeof(long long), hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(&ws.flags[slot].done, &ws.st.p->done, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipEventRecord(ws.ev[slot], s));
  }

  // b_in / x_io / W_in are device pointers here (host mode is staged by the caller).
  int solve(const double *b_in, double *x_io, const double *W_in, int64_t maxit, double eps, double *res_host,
            int64_t res_cap, int64_t *it_out) {
    if (eps <= 0.0) eps = 1e-7;           // RecyclingKrylovSolvers.jl:21
    if (maxit == 0) maxit = n;            // cg.jl:25
    const int64_t cap_dev = std::min<int64_t>(maxit, (int64_t)n) + 1;  // reference: res_norm has n entries
    ws.res_norm.ensure((size_t)cap_dev);
    if (nvec > 0) ws.ensure_deflation(nvec);
    const size_t vb = sizeof(double) * (size_t)n;
    MI_HIP(hipMemcpyAsync(ws.b.p, b_in, vb, hipMemcpyDeviceToDevice, s));
    MI_HIP(hipMemcpyAsync(ws.x.p, x_io, vb, hipMemcpyDeviceToDevice, s));
    const int pre = M != nullptr;

    if (nvec > 0) {
      // defcg.jl:40-54 / 260-275
      MI_HIP(hipMemcpyAsync(ws.W.p, W_in, vb * nvec, hipMemcpyDeviceToDevice, s));
      for (int v = 0; v < nvec; ++v) A->apply(ws.W.p + (size_t)v * n, ws.AW.p + (size_t)v * n, nullptr);  // WtA[v,:] = A*W[:,v]
      hipLaunchKernelGGL(k_small_gram, dim3(nvec, nvec), dim3(NT), 0, s, n, ws.AW.p, ws.W.p, ws.gram.p, nvec);  // WtAW
      MI_HIP(hipGetLastError());
      std::vector<double> lu((size_t)nvec * nvec);
      MI_HIP(hipMemcpyAsync(lu.data(), ws.gram.p, sizeof(double) * lu.size(), hipMemcpyDeviceToHost, s));
      MI_HIP(hipStreamSynchronize(s));
      std::vector<int> piv;
      const int info = host_lu(nvec, lu, piv);
      if (info) raise(MI_ERR_SINGULAR, "WtAW is singular: U[%d,%d] == 0 (SingularException(%d))", info, info, info);
      ws.LU.upload(lu.data(), lu.size(), s);
      ws.piv.upload(piv.data(), piv.size(), s);
      MI_HIP(hipStreamSynchronize(s));
      A->apply(ws.x.p, ws.Ap.p, nullptr);                                    // r .= b .- A*x
      hipLaunchKernelGGL(k_residual, dim3(g), dim3(NT), 0, s, n, ws.b.p, ws.Ap.p, ws.r.p, ws.part_rr.p, ws.part_bb.p);
      project(ws.W.p, ws.r.p, nullptr);                                      // mu = WtAW \ (W'r)
      hipLaunchKernelGGL(k_add_Wmu, dim3(g), dim3(NT), 0, s, n, ws.x.p, ws.W.p, ws.mu.p, nvec);  // x .+= W*mu
      MI_HIP(hipGetLastError());
    }
    // cg.jl:27-32 / 82-89; defcg.jl:58-66 / 279-288
    A->apply(ws.x.p, ws.Ap.p, nullptr);
    hipLaunchKernelGGL(k_residual, dim3(g), dim3(NT), 0, s, n, ws.b.p, ws.Ap.p, ws.r.p, ws.part_rr.p, ws.part_bb.p);
    const double *zz = ws.r.p;
    if (pre) {
      M->apply(ws.r.p, ws.z.p, nullptr);
      dot_partial(ws.r.p, ws.z.p, ws.part_rz.p, nullptr);
      zz = ws.z.p;
    }
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(NT), 0, s, ws.st.p, ws.part_rr.p, ws.part_bb.p,
                       pre ? ws.part_rz.p : (const double *)nullptr, g, eps, (long long)maxit, (long long)cap_dev,
                       ws.res_norm.p);
    if (nvec > 0) project(ws.AW.p, zz, nullptr);
    hipLaunchKernelGGL(k_init_p, dim3(g), dim3(NT), 0, s, n, zz, ws.p.p, nvec > 0 ? ws.W.p : (const double *)nullptr,
                       nvec > 0 ? ws.mu.p : (const double *)nullptr, nvec);
    MI_HIP(hipGetLastError());

    // ---- the loop
    const bool use_graph = ctx->chunk > 0 && A->graph_safe() && (!M || M->graph_safe());
    const int64_t max_launch = use_graph ? (maxit + ctx->chunk - 1) / ctx->chunk + 2 : maxit + 2;
    if (use_graph) {
      hipGraphExec_t ex = graph(ctx->chunk);
      int slot = 0;
      bool have_prev = false, stop = false;
      fetch_flags(slot);  // state after set-up (covers maxit <= 1 and an already converged x)
      have_prev = true;
      for (int64_t l = 0; l < max_launch && !stop; ++l) {
        MI_HIP(hipGraphLaunch(ex, s));
        const int prev = slot;
        slot ^= 1;
        fetch_flags(slot);
        if (have_prev) {
          MI_HIP(hipEventSynchronize(ws.ev[prev]));
          stop = ws.flags[prev].done != 0;
        }
      }
    } else {
      for (int64_t l = 0; l < max_launch; ++l) {
        fetch_flags(0);
        MI_HIP(hipEventSynchronize(ws.ev[0]));
        if (ws.flags[0].done) break;
        iteration();
      }
    }
    MI_HIP(hipStreamSynchronize(s));
    fetch_flags(0);
    MI_HIP(hipStreamSynchronize(s));
    const long long it = ws.flags[0].it;
    if (!ws.flags[0].done) raise(MI_ERR_HIP, "internal: Krylov loop ended without the stop flag (it=%lld)", it);
    MI_HIP(hipMemcpyAsync(x_io, ws.x.p, vb, hipMemcpyDeviceToDevice, s));
    const int64_t ncopy = std::min<int64_t>(std::min<int64_t>(it, res_cap), cap_dev);
    if (res_host && ncopy > 0)
      MI_HIP(hipMemcpyAsync(res_host, ws.res_norm.p, sizeof(double) * ncopy, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (it_out) *it_out = it;
    if (ws.flags[0].overflow || it > res_cap)
      return fail(MI_ERR_RES_CAPACITY, "res_norm capacity %lld < it = %lld", (long long)res_cap, it);
    return MI_OK;
  }
};

}  // namespace mi
