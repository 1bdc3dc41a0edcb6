// Device side of the eigCG family (eigcg.jl, defcg.jl:111-223, 337-473): the Lanczos bookkeeping that rides on the
// CG iteration — the search-space columns V, the helper vector tvec and the projected matrix VtAV — kept in HBM so
// that iterations between two Ritz restarts replay as one hipGraph. The restart itself (tiny dense eigen/SVD work)
// is done on the host (dense_small.hpp).
#pragma once
#include "kernels.hpp"

namespace mi {

struct EigState {
  long long rec_it;     // value of SolverState::it when the last iteration was recorded
  int ivec;             // 0-based column of V holding the newest Lanczos vector
  int nev;              // columns kept by the last restart
  int just_restarted;   // eigcg.jl:45 — the iteration after a restart computes the coupling column
  int restart_pending;  // the iteration with ivec == spdim ran: the host must restart before the next one
  double hlpr;          // sqrt(rTz) (eigpcg, eigcg.jl:212-214) / res_norm[it-1] (eigcg, :83) at the restart
};

// V[:, col] = z / sqrt(rTz)  (pcg variants)  or  r / res_norm[it]  (cg variants); after a restart also
// tvec = -beta * Ap (eigcg.jl:108 / 262). Scalars come from the state block. Launched eagerly by the host
// (set-up and restarts), never inside a graph.
// z and Ap are "views" (kernels.hpp AsmView): plain vectors, or the contribution slots of the dense operators whose
// Γ-sum is taken on the fly (the fused loop never materialises z and Ap).
__global__ __launch_bounds__(NT) void k_eig_seed(int n, const SolverState *st, int pre, AsmView z, double *__restrict__ vcol,
                                                 double *__restrict__ tvec, AsmView Ap) {
  const double scale = sqrt(pre ? st->rTz : st->rTr);
  const double mbeta = -st->beta;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    if (vcol) vcol[i] = view_load(z, i) / scale;
    if (tvec) tvec[i] = mbeta * view_load(Ap, i);
  }
}

// Vector part of one recorded iteration (runs after the iteration's p-update; no-op when the iteration did not run):
//   if ivec == spdim: tvec .-= beta*Ap   (eigcg.jl:71-73 / 217-219)
//   if just_restarted: tvec .+= Ap       (:80 / 226)
//   if ivec != spdim:  V[:, ivec+1] = z/sqrt(rTz) | r/res_norm[it]   (:112 / 265; defcg.jl:213 / 446)
__global__ __launch_bounds__(NT) void k_eig_vec(int n, const SolverState *st, const EigState *es, int pre, int spdim,
                                                AsmView z, AsmView Ap, double *__restrict__ V, double *__restrict__ tvec) {
  if (st->it <= es->rec_it) return;
  const int ivec = es->ivec, jr = es->just_restarted;
  const bool last = ivec == spdim - 1;
  const double scale = sqrt(pre ? st->rTz : st->rTr);
  const double beta = st->beta;
  double *vnew = V + (long long)(ivec + 1) * n;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    if (tvec && (last || jr)) {
      double t = tvec[i];
      const double a = view_load(Ap, i);
      if (last) t = t - beta * a;
      if (jr) t = t + a;
      tvec[i] = t;
    }
    if (!last) vnew[i] = view_load(z, i) / scale;
  }
}

// Coupling column after a restart: part[j*gx + g] = partial of V[:, j] . (tvec / hlpr), j < nev (eigcg.jl:82-83 / 228-229).
// grid (gx, 2*nvec): the number of kept columns is device state, rows beyond it return at once.
__global__ __launch_bounds__(NT) void k_eig_coupling(int n, const SolverState *st, const EigState *es,
                                                     const double *__restrict__ V, const double *__restrict__ tvec,
                                                     double *__restrict__ part) {
  if (st->it <= es->rec_it || !es->just_restarted || (int)blockIdx.y >= es->nev) return;
  __shared__ double sm[NT / 64 + 1];
  const double h = es->hlpr;
  const double *v = V + (long long)blockIdx.y * n;
  double s = 0.0;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) s += v[i] * (tvec[i] / h);
  s = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// Scalar part of one recorded iteration: VtAV entries and the column counter (eigcg.jl:78-85, 110-115; defcg.jl:183, 443-448).
__global__ __launch_bounds__(64) void k_eig_state(const SolverState *st, EigState *es, double *__restrict__ T, int spdim,
                                                  const double *__restrict__ part, int gx, int has_tvec) {
  if (st->it <= es->rec_it) return;
  const int ivec = es->ivec;
  const double alpha = st->alpha, beta = st->beta;
  if (has_tvec && es->just_restarted)
    for (int j = threadIdx.x; j < es->nev; j += 64) {
      double s = 0.0;
      for (int g = 0; g < gx; ++g) s += part[j * gx + g];
      T[j + (long long)ivec * spdim] = s;
    }
  if (threadIdx.x == 0) {
    T[ivec + (long long)ivec * spdim] += 1. / alpha;
    es->just_restarted = 0;
    if (ivec == spdim - 1) {
      es->restart_pending = 1;
    } else {
      T[ivec + (long long)(ivec + 1) * spdim] = -sqrt(beta) / alpha;
      T[(ivec + 1) + (long long)(ivec + 1) * spdim] = beta / alpha;
      es->ivec = ivec + 1;
    }
    es->rec_it = st->it;
  }
}

// out[:, j] = V[:, 0:m] * G[:, j]   (V[:, 1:nev] = V * (Q*Z), eigcg.jl:99 / 253); grid (gx, nev). G is m x nev.
__global__ __launch_bounds__(NT) void k_eig_rotate(int n, int m, const double *__restrict__ V, const double *__restrict__ G,
                                                   double *__restrict__ out) {
  extern __shared__ double gcol[];
  for (int k = threadIdx.x; k < m; k += NT) gcol[k] = G[k + (long long)blockIdx.y * m];
  __syncthreads();
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += V[(long long)k * n + i] * gcol[k];
    out[(long long)blockIdx.y * n + i] = s;
  }
}

// C[i + j*ldc] = A[:, i] . B[:, j]; grid (na, nb), one workgroup per entry (V'AV of eigpcg, WtA*V of eigdef*, W'W).
__global__ __launch_bounds__(NT) void k_gram_rect(int n, const double *__restrict__ A, const double *__restrict__ B,
                                                  double *__restrict__ C, int ldc) {
  __shared__ double sm[NT / 64 + 1];
  const double *a = A + (long long)blockIdx.x * n;
  const double *b = B + (long long)blockIdx.y * n;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += NT) s += a[i] * b[i];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) C[blockIdx.x + (long long)blockIdx.y * ldc] = s;
}

// r .-= W * mu; partial r'r  (eigdefpcg: r .-= W * (WtW \ (W' * r)), defcg.jl:411, followed by rTr = dot(r, r))
__global__ __launch_bounds__(NT) void k_project_r(int n, double *__restrict__ r, const double *__restrict__ W,
                                                  const double *__restrict__ mu, int nvec, double *__restrict__ part_rr,
                                                  const int *done) {
  if (done && *done) return;
  __shared__ double sm[NT / 64 + 1];
  double srr = 0.0;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    const double ri = r[i] - w_times_mu(W, n, i, mu, nvec);
    r[i] = ri;
    srr += ri * ri;
  }
  srr = block_sum(srr, sm);
  if (threadIdx.x == 0) part_rr[blockIdx.x] = srr;
}

}  // namespace mi
