// eigcg / eigpcg (RecyclingKrylovSolvers/eigcg.jl:27-123, 143-290) and eigdefcg / eigdefpcg (defcg.jl:111-223,
// 337-473): CG / PCG / Def-CG / Def-PCG that also extract approximate least-dominant eigenvectors (Stathopoulos &
// Orginos' eigCG window: spdim search directions, thick restart keeping <= 2*nvec Ritz vectors). SURVEY.md §8 f1.
//
// Device / host split: the iteration (the same kernels as cg/pcg/defcg/defpcg, multi-workgroup form) and the
// per-iteration Lanczos bookkeeping (eig_kernels.hpp) run on the device; all iterations up to the next restart are
// ONE graph replay. At a restart the host pulls the spdim x spdim projected matrix, does the tiny dense eigen/SVD work
// (dense_small.hpp) and pushes back the rotation; the n x spdim basis never leaves HBM.
//
// x, it and res_norm are those of cg/pcg/defcg/defpcg to the usual bar (eigdefpcg additionally re-orthogonalises r
// against W each iteration, defcg.jl:411). The returned V[:, 1:nvec] is determined up to the sign/rotation freedom of
// eigenvectors; `eigen(H)` of the cg variants (a general solve in Julia unless H is exactly symmetric) is restated
// with the symmetric solver on H's upper triangle.
#pragma once
#include "dense_small.hpp"
#include "solvers.hpp"

namespace mi {

enum EigKind { EIGCG = 0, EIGPCG = 1, EIGDEFCG = 2, EIGDEFPCG = 3 };

struct EigKrylov {
  Krylov k;
  EigKind kind;
  int nvec, spdim;  // nvec: vectors returned (and, for the deflated kinds, columns of W)
  bool pre, deflated, has_tvec;
  SolverWorkspace &ws;
  hipStream_t s;
  int n, g;
  bool first_restart = true;

  EigKrylov(mi_ctx_s *c, Operator *A, Operator *M, EigKind kind_, int nvec_, int spdim_)
      : k(c, A, M, (kind_ == EIGDEFCG || kind_ == EIGDEFPCG) ? nvec_ : 0, /*generic=*/kind_ == EIGDEFPCG, /*allow_fold=*/false),
        kind(kind_), nvec(nvec_),
        spdim(spdim_), pre(M != nullptr), deflated(kind_ == EIGDEFCG || kind_ == EIGDEFPCG), has_tvec(!deflated), ws(k.ws),
        s(c->stream), n(k.n), g(k.g) {}

  struct Snapshot {
    SolverState st;
    EigState es;
  };
  Snapshot fetch() {
    Snapshot h;
    MI_HIP(hipMemcpyAsync(&h.st, ws.st, sizeof(SolverState), hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(&h.es, ws.ees.p, sizeof(EigState), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    return h;
  }
  std::vector<double> pull_T() {
    std::vector<double> T((size_t)spdim * spdim);
    MI_HIP(hipMemcpyAsync(T.data(), ws.eT.p, sizeof(double) * T.size(), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    return T;
  }

  // VtAV[1:nvec, nvec+1:m] = WtA * V[:, nvec+1:m]  (defcg.jl:186-189 / 422-425, 455-457)
  void deflated_block(int m) {
    if (m > nvec)
      hipLaunchKernelGGL(k_gram_rect, dim3(nvec, m - nvec), dim3(NT), 0, s, n, ws.AW.p, ws.eV.p + (size_t)nvec * n,
                         ws.eT.p + (size_t)nvec * spdim, spdim);
    MI_HIP(hipGetLastError());
  }
  // V[:, 1:nev] = V[:, 1:m] * (Q*Z)
  void rotate(const dense::Ritz &R, int m) {
    if (R.nev == 0) return;
    ws.eG.upload(R.G.data(), R.G.size(), s);
    hipLaunchKernelGGL(k_eig_rotate, dim3(g, R.nev), dim3(NT), sizeof(double) * m, s, n, m, ws.eV.p, ws.eG.p, ws.eVtmp.p);
    MI_HIP(hipGetLastError());
    MI_HIP(hipMemcpyAsync(ws.eV.p, ws.eVtmp.p, sizeof(double) * (size_t)n * R.nev, hipMemcpyDeviceToDevice, s));
  }

  // The `if ivec == spdim` block (eigcg.jl:87-109 / 232-263; defcg.jl:185-210 / 421-442), after the iteration ran.
  void restart(const Snapshot &h) {
    if (has_tvec) {  // tvec .= -beta*Ap first: the A-applies below may reuse the slots the Ap view points at
      hipLaunchKernelGGL(k_eig_seed, dim3(g), dim3(NT), 0, s, n, ws.st, (int)pre, k.z_view(), (double *)nullptr, ws.etvec.p,
                         k.Ap_view());
      MI_HIP(hipGetLastError());
    }
    if (kind == EIGPCG) {
      k.A->apply_multi(ws.eV.p, n, spdim, ws.eAV.p, n);                     // AV[:, j] = A * V[:, j]
      hipLaunchKernelGGL(k_gram_rect, dim3(spdim, spdim), dim3(NT), 0, s, n, ws.eV.p, ws.eAV.p, ws.eT.p, spdim);  // VtAV .= V'AV
      MI_HIP(hipGetLastError());
    } else if (deflated && first_restart) {
      deflated_block(spdim);
      first_restart = false;
    }
    std::vector<double> T = pull_T();
    const dense::Ritz R = dense::ritz_restart(T.data(), spdim, spdim, nvec);
    if (R.nev + 1 > spdim) raise(MI_ERR_BOUNDS, "eig restart: nev + 1 = %d exceeds spdim = %d (BoundsError)", R.nev + 1, spdim);
    rotate(R, spdim);
    const int ivec = R.nev;  // 0-based column of the new Lanczos vector
    hipLaunchKernelGGL(k_eig_seed, dim3(g), dim3(NT), 0, s, n, ws.st, (int)pre, k.z_view(), ws.eV.p + (size_t)ivec * n,
                       (double *)nullptr, k.Ap_view());
    MI_HIP(hipGetLastError());
    std::fill(T.begin(), T.end(), 0.0);
    for (int j = 0; j < R.nev; ++j) T[j + (size_t)j * spdim] = R.vals[j];
    T[ivec + (size_t)ivec * spdim] = h.st.beta / h.st.alpha;
    ws.eT.upload(T.data(), T.size(), s);
    EigState es{};
    es.rec_it = h.st.it;
    es.ivec = ivec;
    es.nev = R.nev;
    es.just_restarted = 1;
    es.restart_pending = 0;
    es.hlpr = has_tvec ? std::sqrt(pre ? h.st.rTz : h.st.rTr) : 0.0;
    ws.ees.upload(&es, 1, s);
    MI_HIP(hipStreamSynchronize(s));  // T, G, es are host stack/heap buffers
  }

  // The block after the loop of the preconditioned kinds (eigcg.jl:269-287 / defcg.jl:451-470).
  int final_extraction(const Snapshot &h) {
    if (!pre || h.es.just_restarted) return MI_OK;
    const int ivec1 = h.es.ivec + 1;  // the reference's 1-based ivec
    if (ivec1 <= nvec) return MI_OK;  // "Less CG iterations than the number of eigenvectors wanted": Lanczos vectors returned
    const int m = ivec1 - 1;
    if (deflated && first_restart) deflated_block(m);
    if (m - 1 < nvec)
      return fail(MI_ERR_BOUNDS, "final Ritz extraction: eigvecs(Tm[1:%d,1:%d])[:, 1:%d] is out of bounds (BoundsError)", m - 1,
                  m - 1, nvec);
    std::vector<double> T = pull_T();
    const dense::Ritz R = dense::ritz_restart(T.data(), spdim, m, nvec);
    rotate(R, m);
    MI_HIP(hipStreamSynchronize(s));
    return MI_OK;
  }

  int solve(const double *b_in, double *x_io, const double *W_in, int64_t maxit, double eps, double *res_host,
            int64_t res_cap, int64_t *it_out, double *V_out) {
    k.eig.tag = 1 + (int)kind + 8 * spdim;
    k.eig.spdim = spdim;
    k.eig.has_tvec = has_tvec;
    k.eig.project_r = kind == EIGDEFPCG;
    ws.ensure_eig(spdim, nvec, kind == EIGPCG);
    int64_t cap_dev = 0;
    k.begin(b_in, x_io, W_in, maxit, eps, cap_dev);
    if (kind == EIGDEFPCG) {  // WtW .= W'W (defcg.jl:369), factored once
      hipLaunchKernelGGL(k_gram_rect, dim3(nvec, nvec), dim3(NT), 0, s, n, ws.W.p, ws.W.p, ws.gram.p, nvec);
      MI_HIP(hipGetLastError());
      k.factor(ws.gram.p, ws.eLUw, ws.epivw, nullptr, "WtW");
    }
    k.setup_tail();
    // VtAV[1:nvec,1:nvec] = WtAW; V[:,1:nvec] = W; ivec = nvec + 1; V[:, ivec] = z / sqrt(rTz)  (defcg.jl:158-163 / 391-396)
    const int ivec0 = deflated ? nvec : 0;
    std::vector<double> T((size_t)spdim * spdim, 0.0);
    if (deflated) {
      for (int j = 0; j < nvec; ++j)
        for (int i = 0; i < nvec; ++i) T[i + (size_t)j * spdim] = k.gram_host[i + (size_t)j * nvec];
      MI_HIP(hipMemcpyAsync(ws.eV.p, ws.W.p, sizeof(double) * (size_t)n * nvec, hipMemcpyDeviceToDevice, s));
    }
    ws.eT.upload(T.data(), T.size(), s);
    MI_HIP(hipMemsetAsync(ws.etvec.p, 0, sizeof(double) * (size_t)n, s));
    hipLaunchKernelGGL(k_eig_seed, dim3(g), dim3(NT), 0, s, n, ws.st, (int)pre, k.z_view(), ws.eV.p + (size_t)ivec0 * n,
                       (double *)nullptr, k.Ap_view());
    MI_HIP(hipGetLastError());
    EigState es0{};
    es0.rec_it = 1;
    es0.ivec = ivec0;
    ws.ees.upload(&es0, 1, s);
    Snapshot h = fetch();

    const bool use_graph = k.ctx->chunk > 0 && k.A->graph_safe() && (!k.M || k.M->graph_safe()) && !k.ctx->no_graph && !k.ctx->has_comm();
    // A replay runs to the next thick restart; launches behind the stop rule return at once but still cost a launch each
    // (~4.5 us x 7-10 kernels per iteration). When the previous solve with these operators took `predicted` iterations, the
    // replay that would overshoot it is cut there (one more replay follows if this solve needs more).
    const GraphKey pk{k.A, k.M, nvec, -1000 - (int)kind};
    const int predicted = ws.predicted.count(pk) ? ws.predicted[pk] : 0;
    for (int64_t guard = 0; guard < maxit + 4; ++guard) {
      if (h.es.restart_pending) {
        restart(h);
        h.es = fetch().es;
      }
      if (h.st.done) break;
      int seg = spdim - h.es.ivec;  // iterations up to and including the one that fills the window
      if (predicted > 0 && h.st.it < predicted && h.st.it + seg > predicted) seg = (int)(predicted - h.st.it);
      if (use_graph) {
        if (ws.graphs.size() > 48) ws.drop_graphs();
        MI_HIP(hipGraphLaunch(k.graph(seg), s));
      } else {
        for (int i = 0; i < seg; ++i) k.iteration();
      }
      h = fetch();
    }
    if (!h.st.done) raise(MI_ERR_HIP, "internal: eig Krylov loop ended without the stop flag (it=%lld)", h.st.it);
    ws.predicted[pk] = (int)std::max<long long>(1, h.st.it);
    const int rc_extract = final_extraction(h);

    const size_t vb = sizeof(double) * (size_t)n;
    MI_HIP(hipMemcpyAsync(x_io, ws.x, vb, hipMemcpyDeviceToDevice, s));
    if (V_out) MI_HIP(hipMemcpyAsync(V_out, ws.eV.p, vb * nvec, hipMemcpyDeviceToDevice, s));
    const long long it = h.st.it;
    const int64_t ncopy = std::min<int64_t>(it, std::min<int64_t>(res_cap, cap_dev));
    if (res_host && ncopy > 0)
      MI_HIP(hipMemcpyAsync(res_host, ws.res_norm.p, sizeof(double) * ncopy, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (it_out) *it_out = it;
    if (h.st.overflow || it > res_cap)
      return fail(MI_ERR_RES_CAPACITY, "res_norm capacity %lld < it = %lld", (long long)res_cap, it);
    return rc_extract;
  }
};

}  // namespace mi
