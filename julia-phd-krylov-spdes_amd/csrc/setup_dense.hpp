// Set-up of the assembled mode on the device, without torch:
//   * `assemble_local_schurs` (EPDD.jl:667-695): dense S_d = A_ΓΓdd - A_IΓdd' A_IIdd^{-1} A_IΓdd for every subdomain,
//     symmetrised from the upper triangle (`Symmetric(Array(...))`, :692), plus the condensed right-hand side
//     w_d = A_IΓdd' A_IIdd^{-1} b_Id that `get_schur_rhs` subtracts from b_Γ (:853-861);
//   * `prepare_neumann_neumann_schur_precond(Sd, ...)` (EPDD.jl:1201-1220): ΠS_d = pinv(S_d, rtol = sqrt(eps)).
//
// The reference builds S_d by applying `apply_local_schur` (interior CG to reltol 1e-9) to every unit vector: n_Γd
// iterative solves with ~124 k unknowns each at config 3. Here the interior is eliminated EXACTLY (documented deviation
// N2 of DESIGN.md §2: it only tightens S_d): the interior nodes are ordered by breadth-first levels L_0, L_1, ... grown
// from the nodes adjacent to Γ_d; A_II is block tridiagonal in that order, so
//     T_m = A_mm,   T_k = A_kk - A_{k+1,k}' T_{k+1}^{-1} A_{k+1,k},   g_k = b_k - A_{k+1,k}' T_{k+1}^{-1} g_{k+1},
//     S_d = A_ΓΓ - A_{0Γ}' T_0^{-1} A_{0Γ},   w_d = A_{0Γ}' T_0^{-1} g_0
// is a chain of dense Cholesky / triangular solve / symmetric rank-k update steps on blocks a few hundred wide. Those are
// plain library BLAS-3 / LAPACK calls (rocBLAS dtrsm / dsyrk / dgemv, rocSOLVER dpotrf / dsyevd, bound with dlopen like
// RCCL); the sparse-to-dense scatter, symmetrisation and the pseudo-inverse's spectral filter are kernels of this file.
// The symbolic half (levels, where every stored entry of the CSC blocks lands) is a plan built once per mesh /
// partition; a realization (Example07:162-199) is then `run` with the block values where mi_assembly_run left them.
#pragma once
#include <dlfcn.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <cmath>

#include "operators.hpp"

namespace mi {

struct RocLa {
  void *hb = nullptr, *hs = nullptr;
  rocblas_status (*create_handle)(rocblas_handle *) = nullptr;
  rocblas_status (*destroy_handle)(rocblas_handle) = nullptr;
  rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
  rocblas_status (*set_pointer_mode)(rocblas_handle, rocblas_pointer_mode) = nullptr;
  rocblas_status (*dtrsm)(rocblas_handle, rocblas_side, rocblas_fill, rocblas_operation, rocblas_diagonal, rocblas_int, rocblas_int,
                          const double *, const double *, rocblas_int, double *, rocblas_int) = nullptr;
  rocblas_status (*dsyrk)(rocblas_handle, rocblas_fill, rocblas_operation, rocblas_int, rocblas_int, const double *, const double *,
                          rocblas_int, const double *, double *, rocblas_int) = nullptr;
  rocblas_status (*dgemv)(rocblas_handle, rocblas_operation, rocblas_int, rocblas_int, const double *, const double *, rocblas_int,
                          const double *, rocblas_int, const double *, double *, rocblas_int) = nullptr;
  rocblas_status (*dgemm)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const double *,
                          const double *, rocblas_int, const double *, rocblas_int, const double *, double *, rocblas_int) = nullptr;
  rocblas_status (*dpotrf)(rocblas_handle, const rocblas_fill, const rocblas_int, double *, const rocblas_int, rocblas_int *) = nullptr;
  rocblas_status (*dsyevd)(rocblas_handle, const rocblas_evect, const rocblas_fill, const rocblas_int, double *, const rocblas_int,
                           double *, double *, rocblas_int *) = nullptr;
  static RocLa &get() {
    static RocLa r;
    static bool tried = false;
    if (!tried) {
      tried = true;
      // By SONAME, never by path: when a framework in the process has already loaded its own rocBLAS / rocSOLVER pair
      // (torch ships one with the same SONAMEs), that pair is reused — a second rocBLAS copy would be bound to the first
      // one's rocSOLVER and would page in another 0.6 GB of kernel libraries; otherwise the system pair is loaded.
      const char *nb[] = {"librocblas.so.5", "librocblas.so"};
      const char *ns[] = {"librocsolver.so.0", "librocsolver.so"};
      for (const char *n : nb) if ((r.hb = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
      if (!r.hb) for (const char *n : nb) if ((r.hb = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      for (const char *n : ns) if ((r.hs = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
      if (!r.hs) for (const char *n : ns) if ((r.hs = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      if (r.hb && r.hs) {
#define MI_SYM(h, field, name) r.field = (decltype(r.field))dlsym(h, name)
        MI_SYM(r.hb, create_handle, "rocblas_create_handle"); MI_SYM(r.hb, destroy_handle, "rocblas_destroy_handle");
        MI_SYM(r.hb, set_stream, "rocblas_set_stream"); MI_SYM(r.hb, set_pointer_mode, "rocblas_set_pointer_mode");
        MI_SYM(r.hb, dtrsm, "rocblas_dtrsm"); MI_SYM(r.hb, dsyrk, "rocblas_dsyrk"); MI_SYM(r.hb, dgemv, "rocblas_dgemv");
        MI_SYM(r.hb, dgemm, "rocblas_dgemm");
        MI_SYM(r.hs, dpotrf, "rocsolver_dpotrf"); MI_SYM(r.hs, dsyevd, "rocsolver_dsyevd");
#undef MI_SYM
      }
    }
    if (!r.create_handle || !r.destroy_handle || !r.set_stream || !r.dtrsm || !r.dsyrk || !r.dgemv || !r.dgemm || !r.dpotrf || !r.dsyevd)
      raise(MI_ERR_HIP, "rocBLAS / rocSOLVER could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
    return r;
  }
};
#define MI_ROC(expr)                                                                                       \
  do {                                                                                                     \
    rocblas_status s_ = (expr);                                                                            \
    if (s_ != rocblas_status_success) ::mi::raise(MI_ERR_HIP, "%s failed with rocblas_status %d", #expr, (int)s_); \
  } while (0)

// dense[dst[k]] = vals[src[k]] for the stored entries of one sparse block (the buffer was zeroed before)
__global__ __launch_bounds__(NT) void k_scatter_entries(long long cnt, const int *__restrict__ src, const int *__restrict__ dst,
                                                        const double *__restrict__ vals, double *__restrict__ dense) {
  for (long long k = blockIdx.x * (long long)NT + threadIdx.x; k < cnt; k += (long long)gridDim.x * NT) dense[dst[k]] = vals[src[k]];
}
// v[i] = src[perm[i]]
__global__ __launch_bounds__(NT) void k_gather_perm(int n, const int *__restrict__ perm, const double *__restrict__ src, double *__restrict__ v) {
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) v[i] = src[perm[i]];
}
// out (n x n, column-major) = upper triangle of S mirrored: `Symmetric(Array(S))` (EPDD.jl:692)
__global__ __launch_bounds__(NT) void k_symmetrize_upper(int n, const double *__restrict__ S, int lds, double *__restrict__ out) {
  const long long tot = (long long)n * n;
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < tot; e += (long long)gridDim.x * NT) {
    const int i = (int)(e % n), j = (int)(e / n);
    out[e] = i <= j ? S[i + (long long)j * lds] : S[j + (long long)i * lds];
  }
}
// B[:, j] = V[:, j] * (|lam_j| > tol ? 1 / lam_j : 0), tol = rtol * max_k |lam_k|  (pinv's filter on singular values |lam|)
__global__ __launch_bounds__(NT) void k_pinv_scale(int n, const double *__restrict__ V, const double *__restrict__ lam, double rtol,
                                                   double *__restrict__ B) {
  const double tol = rtol * fmax(fabs(lam[0]), fabs(lam[n - 1]));   // syevd returns the eigenvalues in ascending order
  const long long tot = (long long)n * n;
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < tot; e += (long long)gridDim.x * NT) {
    const double l = lam[e / n];
    B[e] = fabs(l) > tol ? V[e] / l : 0.0;
  }
}

inline int grid_for(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + NT - 1) / NT, 4096)); }

// ------------------------------------------------------------------ the plan
struct SetupDom {
  int n_g = 0, n_i = 0, nlev = 0;
  std::vector<int> lev_off;              // [nlev + 1] offsets of the levels in the permuted interior
  // per block: range [e0, e1) of the entry lists below. Blocks: D_k (nlev), E_k (nlev - 1: rows level k+1, cols level k),
  // B (rows level 0, cols Γ_d), G (A_ΓΓ)
  std::vector<long long> d_e0, e_e0;     // [nlev + 1], [nlev]
  long long b_e0 = 0, b_e1 = 0, g_e0 = 0, g_e1 = 0;
  long long ii_val_off = 0, ig_val_off = 0, gg_val_off = 0, bi_off = 0, s_off = 0, w_off = 0;
  int max_lev = 0;
  // CSC form of the coupling blocks (hand-written elimination): for level k the columns of C_k = A_{k+1,k}, as offsets into
  // the plan-wide arrays c_ptr / c_row / c_src; [nlev - 1] entries; then B = A_IΓ[L_0, :] by Γ_d column
  std::vector<int> cptr_off;
  int bptr_off = 0;
};

struct GjState;
}  // namespace mi

struct mi_setup_s {
  mi_ctx_s *ctx = nullptr;
  int ndom = 0;
  std::vector<mi::SetupDom> dom;
  mi::DevBuf<int> src, dst, perm;        // entry lists of all blocks of all subdomains; interior permutation (concatenated)
  long long n_ii = 0, n_ig = 0, n_gg = 0, n_bi = 0, n_s = 0, n_w = 0;
  // per-stream work space (subdomains are eliminated round-robin on a few streams)
  struct Lane {
    hipStream_t s = nullptr;
    rocblas_handle h = nullptr;
    mi::DevBuf<double> T, Tn, Y, g, gn, S;
    mi::DevBuf<int> info;
  };
  std::vector<Lane> lanes;
  mi::DevBuf<double> st_ii, st_ig, st_gg, st_bi, st_S, st_w;   // staging for host-pointer calls
  // hand-written elimination (block Gauss-Jordan, batched over the subdomains, replayed from a hipGraph)
  std::vector<int> c_ptr_h, c_row_h, c_src_h;
  mi::DevBuf<int> c_ptr, c_row, c_src;
  std::unique_ptr<mi::GjState> gj;
  ~mi_setup_s();
  void release_lanes() {
    for (auto &l : lanes) {
      if (l.h) (void)mi::RocLa::get().destroy_handle(l.h);
      if (l.s) (void)hipStreamDestroy(l.s);
    }
  }
};

namespace mi {

inline void setup_plan_build(mi_setup_s &P, mi_ctx_s *c, int64_t ndom, const int64_t *n_gamma_d, const int64_t *n_i,
                             const int64_t *const *ii_ptr, const int64_t *const *ii_idx, const int64_t *const *ig_ptr,
                             const int64_t *const *ig_idx, const int64_t *const *gg_ptr, const int64_t *const *gg_idx, int base) {
  P.ctx = c; P.ndom = (int)ndom;
  std::vector<int> src, dst, perm_all;
  size_t max_T = 1, max_Y = 1, max_S = 1, max_g = 1;
  for (int64_t d = 0; d < ndom; ++d) {
    SetupDom D;
    D.n_g = (int)n_gamma_d[d]; D.n_i = (int)n_i[d];
    const int ng = D.n_g, ni = D.n_i;
    if (ng < 0 || ni < 0) raise(MI_ERR_BAD_ARG, "setup plan: negative size");
    const int64_t *ip = ii_ptr[d], *ix = ii_idx[d], *gp = ig_ptr[d], *gx = ig_idx[d], *sp = gg_ptr[d], *sx = gg_idx[d];
    if ((ni && (!ip || !ix)) || !gp || !sp) raise(MI_ERR_BAD_ARG, "setup plan: NULL block arrays for subdomain %lld", (long long)d);
    const int64_t nnz_ii = ni ? ip[ni] - base : 0, nnz_ig = gp[ng] - base, nnz_gg = sp[ng] - base;
    if (nnz_ii < 0 || nnz_ig < 0 || nnz_gg < 0 || nnz_ii >= INT32_MAX) raise(MI_ERR_BAD_ARG, "setup plan: bad colptr");
    D.ii_val_off = P.n_ii; D.ig_val_off = P.n_ig; D.gg_val_off = P.n_gg; D.bi_off = P.n_bi; D.s_off = P.n_s; D.w_off = P.n_w;
    if (P.n_ii + nnz_ii >= INT32_MAX || P.n_ig + nnz_ig >= INT32_MAX) raise(MI_ERR_BAD_ARG, "setup plan: more than 2^31 stored entries");
    // breadth-first levels from the interior nodes adjacent to Γ_d (rows of A_IΓdd that hold an entry)
    std::vector<int> level(ni, -1), order;
    order.reserve(ni);
    std::vector<int> frontier;
    for (int64_t k = 0; k < nnz_ig; ++k) {
      const int r = to_i32(gx[k] - base, 0, ni, "A_IΓ rowval");
      if (level[r] < 0) { level[r] = 0; frontier.push_back(r); }
    }
    std::sort(frontier.begin(), frontier.end());
    D.lev_off.push_back(0);
    while (!frontier.empty()) {
      for (int v : frontier) order.push_back(v);
      D.lev_off.push_back((int)order.size());
      std::vector<int> next;
      const int lv = (int)D.lev_off.size() - 1;
      for (int v : frontier)
        for (int64_t k = ip[v] - base; k < ip[v + 1] - base; ++k) {
          const int u = to_i32(ix[k] - base, 0, ni, "A_II rowval");
          if (level[u] < 0) { level[u] = lv; next.push_back(u); }
        }
      std::sort(next.begin(), next.end());
      frontier.swap(next);
    }
    D.nlev = (int)D.lev_off.size() - 1;
    std::vector<int> pos(ni, -1);   // position inside its level
    for (int l = 0; l < D.nlev; ++l)
      for (int q = D.lev_off[l]; q < D.lev_off[l + 1]; ++q) pos[order[q]] = q - D.lev_off[l];
    for (int l = 0; l < D.nlev; ++l) D.max_lev = std::max(D.max_lev, D.lev_off[l + 1] - D.lev_off[l]);
    // entry lists: per block a list of (index into the value family, position in the dense block, column-major)
    std::vector<std::vector<std::pair<int, int>>> Dl(D.nlev), El(std::max(0, D.nlev - 1));
    for (int col = 0; col < ni; ++col) {
      const int lc = level[col];
      if (lc < 0) continue;   // not connected to Γ_d: cannot influence S_d
      for (int64_t k = ip[col] - base; k < ip[col + 1] - base; ++k) {
        const int row = (int)(ix[k] - base), lr = level[row];
        const int srcv = (int)(P.n_ii + k);
        if (lr == lc) {
          const int nl = D.lev_off[lc + 1] - D.lev_off[lc];
          Dl[lc].push_back({srcv, pos[row] + pos[col] * nl});
        } else if (lr == lc + 1) {   // E_lc: rows level lc+1, cols level lc
          const int nr = D.lev_off[lr + 1] - D.lev_off[lr];
          El[lc].push_back({srcv, pos[row] + pos[col] * nr});
        }                            // lr == lc - 1: the transposed twin, not stored (symmetric); |lr - lc| > 1 cannot occur
        else if (lr >= 0 && lr != lc - 1) raise(MI_ERR_BAD_ARG, "setup plan: A_II is not block tridiagonal over the BFS levels (not symmetric?)");
      }
    }
    auto flush = [&](std::vector<std::pair<int, int>> &v) {
      const long long e0 = (long long)src.size();
      for (auto &p : v) { src.push_back(p.first); dst.push_back(p.second); }
      return e0;
    };
    // CSC (by column position in the level) of every coupling block, for the pick kernels
    for (int l = 0; l + 1 < D.nlev; ++l) {
      const int nr = D.lev_off[l + 2] - D.lev_off[l + 1], nc = D.lev_off[l + 1] - D.lev_off[l];
      std::vector<std::vector<std::pair<int, int>>> cols(nc);
      for (auto &e : El[l]) cols[e.second / nr].push_back({e.second % nr, e.first});
      D.cptr_off.push_back((int)P.c_ptr_h.size());
      for (int j = 0; j < nc; ++j) {
        P.c_ptr_h.push_back((int)P.c_row_h.size());
        std::sort(cols[j].begin(), cols[j].end());
        for (auto &e : cols[j]) { P.c_row_h.push_back(e.first); P.c_src_h.push_back(e.second); }
      }
      P.c_ptr_h.push_back((int)P.c_row_h.size());
    }
    for (int l = 0; l < D.nlev; ++l) D.d_e0.push_back(flush(Dl[l]));
    D.d_e0.push_back((long long)src.size());
    for (int l = 0; l + 1 < D.nlev; ++l) D.e_e0.push_back(flush(El[l]));
    D.e_e0.push_back((long long)src.size());
    // B = A_IΓ[L_0, :] (n_0 x n_Γd) from the CSC arrays of A_IΓdd
    const int n0 = D.nlev ? D.lev_off[1] : 0;
    D.b_e0 = (long long)src.size();
    for (int col = 0; col < ng; ++col)
      for (int64_t k = gp[col] - base; k < gp[col + 1] - base; ++k) {
        const int row = (int)(gx[k] - base);
        src.push_back((int)(P.n_ig + k)); dst.push_back(pos[row] + col * std::max(n0, 1));
      }
    D.b_e1 = (long long)src.size();
    D.bptr_off = (int)P.c_ptr_h.size();
    for (int col = 0; col < ng; ++col) {
      P.c_ptr_h.push_back((int)P.c_row_h.size());
      for (int64_t k = gp[col] - base; k < gp[col + 1] - base; ++k) {
        P.c_row_h.push_back(pos[(int)(gx[k] - base)]); P.c_src_h.push_back((int)(P.n_ig + k));
      }
    }
    P.c_ptr_h.push_back((int)P.c_row_h.size());
    D.g_e0 = (long long)src.size();
    for (int col = 0; col < ng; ++col)
      for (int64_t k = sp[col] - base; k < sp[col + 1] - base; ++k) {
        const int row = to_i32(sx[k] - base, 0, ng, "A_ΓΓ rowval");
        src.push_back((int)(P.n_gg + k)); dst.push_back(row + col * std::max(ng, 1));
      }
    D.g_e1 = (long long)src.size();
    for (int v : order) perm_all.push_back((int)(P.n_bi + v));
    for (int q = (int)order.size(); q < ni; ++q) perm_all.push_back(0);   // keep one slot per interior node (unreached: unused)
    max_T = std::max(max_T, (size_t)D.max_lev * D.max_lev);
    max_Y = std::max(max_Y, (size_t)D.max_lev * (std::max(D.max_lev, ng) + 1));
    max_S = std::max(max_S, (size_t)ng * ng);
    max_g = std::max(max_g, (size_t)std::max(D.max_lev, ng) + 1);
    P.n_ii += nnz_ii; P.n_ig += nnz_ig; P.n_gg += nnz_gg; P.n_bi += ni; P.n_s += (long long)ng * ng; P.n_w += ng;
    P.dom.push_back(std::move(D));
  }
  if (src.size() >= (size_t)INT32_MAX) raise(MI_ERR_BAD_ARG, "setup plan: too many entries");
  hipStream_t s = c->stream;
  P.src.upload(src, s); P.dst.upload(dst, s); P.perm.upload(perm_all, s);
  P.c_ptr.upload(P.c_ptr_h, s); P.c_row.upload(P.c_row_h, s); P.c_src.upload(P.c_src_h, s);
#ifndef MI355_EXPERIMENTAL
  return;                                       // the library path (rocBLAS / rocSOLVER chains) exists in EXPERIMENTAL builds only
#else
  if (!env_int("MI355_SETUP_LIB", 0)) return;   // ... and there on request (`make EXPERIMENTAL=1`, MI355_SETUP_LIB=1)
  RocLa &la = RocLa::get();
  // One chain at a time by default: with several rocSOLVER/rocBLAS handles running potrf / trsm chains concurrently on
  // different streams some S_d came out wrong at the 1e-7 level (tools/setup_probe.py, config 3), with one stream every
  // block matches the host elimination to rounding — and the chains are bound by the host's launch rate anyway
  // (~250 levels x a few dozen library kernels per subdomain; capturing the chain into a hipGraph is not an option:
  // rocSOLVER's potrf faults under stream capture on this ROCm).
  const int nl = std::max(1, std::min<int>((int)ndom, env_int("MI355_SETUP_STREAMS", 1)));
  P.lanes.resize(nl);
  for (auto &l : P.lanes) {
    MI_HIP(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
    MI_ROC(la.create_handle(&l.h));
    MI_ROC(la.set_stream(l.h, l.s));
    l.T.alloc(max_T); l.Tn.alloc(max_T); l.Y.alloc(max_Y); l.g.alloc(max_g); l.gn.alloc(max_g); l.S.alloc(max_S); l.info.alloc(2);
    MI_HIP(hipMemsetAsync(l.info.p, 0, 2 * sizeof(int), s));
  }
  MI_HIP(hipStreamSynchronize(s));
#endif
}

#ifdef MI355_EXPERIMENTAL
// One realization: values (device pointers; the concatenations over subdomains of the CSC nzval arrays and of b_Id) ->
// Sd (concatenated column-major blocks) and, with bI != nullptr, w (concatenated). Enqueued on the plan's streams, which
// first wait for the context's stream and are joined back into it at the end.
inline void setup_plan_run(mi_setup_s &P, const double *ii_val, const double *ig_val, const double *gg_val, const double *bI,
                           double *Sd, double *w) {
  RocLa &la = RocLa::get();
  mi_ctx_s *c = P.ctx;
  const double one = 1.0, mone = -1.0;
  const bool same_stream = P.lanes.size() == 1 && P.lanes[0].s == c->stream;   // (the captured form)
  hipEvent_t ev0 = nullptr;
  if (!same_stream) {
    MI_HIP(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
    MI_HIP(hipEventRecord(ev0, c->stream));
    for (auto &l : P.lanes) MI_HIP(hipStreamWaitEvent(l.s, ev0, 0));
  }
  auto fill = [&](hipStream_t s, double *dense, size_t count, long long e0, long long e1, const double *vals) {
    MI_HIP(hipMemsetAsync(dense, 0, count * sizeof(double), s));
    if (e1 > e0) hipLaunchKernelGGL(k_scatter_entries, dim3(grid_for(e1 - e0)), dim3(NT), 0, s, e1 - e0, P.src.p + e0, P.dst.p + e0, vals, dense);
  };
  for (int d = 0; d < P.ndom; ++d) {
    const SetupDom &D = P.dom[d];
    mi_setup_s::Lane &L = P.lanes[d % P.lanes.size()];
    hipStream_t s = L.s;
    const int ng = D.n_g;
    double *Sout = Sd + D.s_off;
    // S <- dense(A_ΓΓ)
    fill(s, L.S.p, (size_t)std::max(1, ng) * std::max(1, ng), D.g_e0, D.g_e1, gg_val);
    if (D.nlev > 0 && ng > 0) {
      auto nlev_of = [&](int l) { return D.lev_off[l + 1] - D.lev_off[l]; };
      const int m = D.nlev - 1;
      double *T = L.T.p, *Tn = L.Tn.p;
      fill(s, T, (size_t)nlev_of(m) * nlev_of(m), D.d_e0[m], D.d_e0[m + 1], ii_val);
      if (bI) hipLaunchKernelGGL(k_gather_perm, dim3(grid_for(nlev_of(m))), dim3(NT), 0, s, nlev_of(m), P.perm.p + D.bi_off + D.lev_off[m], bI, L.g.p);
      double *g = L.g.p, *gn = L.gn.p;
      for (int k = m - 1; k >= 0; --k) {
        const int n1 = nlev_of(k + 1), n0 = nlev_of(k);
        MI_ROC(la.dpotrf(L.h, rocblas_fill_lower, n1, T, n1, L.info.p));                       // T_{k+1} = L L'
        fill(s, L.Y.p, (size_t)n1 * n0, D.e_e0[k], D.e_e0[k + 1], ii_val);                      // C = A_{k+1,k}
        if (bI) MI_HIP(hipMemcpyAsync(L.Y.p + (size_t)n1 * n0, g, sizeof(double) * n1, hipMemcpyDeviceToDevice, s));
        MI_ROC(la.dtrsm(L.h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n1,
                        n0 + (bI ? 1 : 0), &one, T, n1, L.Y.p, n1));                           // Y = L \ [C | g]
        fill(s, Tn, (size_t)n0 * n0, D.d_e0[k], D.d_e0[k + 1], ii_val);                          // A_kk
        MI_ROC(la.dsyrk(L.h, rocblas_fill_lower, rocblas_operation_transpose, n0, n1, &mone, L.Y.p, n1, &one, Tn, n0));  // - Yc' Yc
        if (bI) {
          hipLaunchKernelGGL(k_gather_perm, dim3(grid_for(n0)), dim3(NT), 0, s, n0, P.perm.p + D.bi_off + D.lev_off[k], bI, gn);
          MI_ROC(la.dgemv(L.h, rocblas_operation_transpose, n1, n0, &mone, L.Y.p, n1, L.Y.p + (size_t)n1 * n0, 1, &one, gn, 1));
          std::swap(g, gn);
        }
        std::swap(T, Tn);
      }
      const int n0 = nlev_of(0);
      MI_ROC(la.dpotrf(L.h, rocblas_fill_lower, n0, T, n0, L.info.p + 1));
      fill(s, L.Y.p, (size_t)n0 * ng, D.b_e0, D.b_e1, ig_val);                                   // B = A_IΓ[L_0, :]
      if (bI) MI_HIP(hipMemcpyAsync(L.Y.p + (size_t)n0 * ng, g, sizeof(double) * n0, hipMemcpyDeviceToDevice, s));
      MI_ROC(la.dtrsm(L.h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n0,
                      ng + (bI ? 1 : 0), &one, T, n0, L.Y.p, n0));
      MI_ROC(la.dsyrk(L.h, rocblas_fill_upper, rocblas_operation_transpose, ng, n0, &mone, L.Y.p, n0, &one, L.S.p, ng));  // S -= Yc' Yc (upper)
      if (bI && w) {
        MI_HIP(hipMemsetAsync(w + D.w_off, 0, sizeof(double) * ng, s));
        MI_ROC(la.dgemv(L.h, rocblas_operation_transpose, n0, ng, &one, L.Y.p, n0, L.Y.p + (size_t)n0 * ng, 1, &one, w + D.w_off, 1));
      }
    } else if (bI && w && ng > 0) {
      MI_HIP(hipMemsetAsync(w + D.w_off, 0, sizeof(double) * ng, s));
    }
    if (ng > 0) hipLaunchKernelGGL(k_symmetrize_upper, dim3(grid_for((long long)ng * ng)), dim3(NT), 0, s, ng, L.S.p, ng, Sout);
    MI_HIP(hipGetLastError());
  }
  if (same_stream) return;
  for (auto &l : P.lanes) {
    hipEvent_t e = nullptr;
    MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    MI_HIP(hipEventRecord(e, l.s));
    MI_HIP(hipStreamWaitEvent(c->stream, e, 0));
    (void)hipEventDestroy(e);
  }
  (void)hipEventDestroy(ev0);
}
#endif   // MI355_EXPERIMENTAL

// ΠS_d = pinv(S_d, rtol) for the concatenated symmetric blocks (device pointers): S = V diag(lam) V', singular values |lam|.
inline void pinv_blocks(mi_ctx_s *c, int ndom, const int64_t *n_gamma_d, const double *Sd, double rtol, double *Pi) {
  RocLa &la = RocLa::get();
  rocblas_handle h = nullptr;
  MI_ROC(la.create_handle(&h));
  struct Guard { RocLa &la; rocblas_handle h; ~Guard() { (void)la.destroy_handle(h); } } guard{la, h};
  MI_ROC(la.set_stream(h, c->stream));
  size_t nmax = 1;
  for (int d = 0; d < ndom; ++d) nmax = std::max<size_t>(nmax, (size_t)n_gamma_d[d]);
  DevBuf<double> V(nmax * nmax), B(nmax * nmax), lam(nmax), E(nmax);
  DevBuf<int> info(1);
  const double one = 1.0, zero = 0.0;
  size_t off = 0;
  std::vector<int> infos(ndom, 0);
  for (int d = 0; d < ndom; ++d) {
    const int n = (int)n_gamma_d[d];
    if (n == 0) continue;
    MI_HIP(hipMemcpyAsync(V.p, Sd + off, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, c->stream));
    MI_ROC(la.dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, V.p, n, lam.p, E.p, info.p));
    hipLaunchKernelGGL(k_pinv_scale, dim3(grid_for((long long)n * n)), dim3(NT), 0, c->stream, n, V.p, lam.p, rtol, B.p);
    MI_HIP(hipGetLastError());
    MI_ROC(la.dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, B.p, n, V.p, n, &zero, Pi + off, n));
    MI_HIP(hipMemcpyAsync(&infos[d], info.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    off += (size_t)n * n;
  }
  MI_HIP(hipStreamSynchronize(c->stream));
  for (int d = 0; d < ndom; ++d)
    if (infos[d] != 0) raise(MI_ERR_HIP, "pinv: the symmetric eigensolver did not converge on block %d (info = %d)", d, infos[d]);
}

}  // namespace mi
