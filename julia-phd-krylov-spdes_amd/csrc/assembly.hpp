// On-device numeric assembly of the local Schur blocks: the element loop of
// `prepare_local_schurs` (Fem/EllipticPdeDomainDecomposition.jl:389-546) for a fixed mesh / partition / f / uexact and
// a NEW nodal coefficient vector — what Example07's realization loop (:162-171) redoes on the host per draw.
//
// The index half (which element contributes to which stored entry, in which order) is prepared once by the host
// (`fem.make_assembly_plan`, or the Julia equivalent); this file is the numeric half:
//   k_elem_coeff:     Δa[e] = (a1 + a2 + a3)/3                                  (EPDD.jl:436-445)
//   k_assemble_plan:  one thread per stored entry sums its contributions in the reference's element order
//                     ΔK_ij = Δa*G_ij/4/Area (:470); right-hand sides: -(ΔK_ij*uexact_i) (:498-507), Δb_i (:516-525)
// Same operations in the same order as the host path (library built with -ffp-contract=off): BIT-EXACT.
// HBM-bound gather: per contribution 4 B code + 8 B G + (cached) 8 B Δa + 8 B Area; no atomics.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace mi {

__global__ __launch_bounds__(NT) void k_elem_coeff(int nel, const int *__restrict__ cells, const double *__restrict__ a,
                                                   double *__restrict__ da) {
  for (int e = blockIdx.x * NT + threadIdx.x; e < nel; e += gridDim.x * NT) {
    double s = 0.0;
    s = s + a[cells[e]];
    s = s + a[cells[nel + e]];
    s = s + a[cells[2 * (long long)nel + e]];
    da[e] = s / 3.0;
  }
}

__global__ __launch_bounds__(NT) void k_assemble_plan(long long n_entries, long long n_matrix, int nel,
                                                      const long long *__restrict__ cptr, const int *__restrict__ ccode,
                                                      const double *__restrict__ da, const double *__restrict__ G,
                                                      const double *__restrict__ area, const double *__restrict__ ue,
                                                      const double *__restrict__ be, double *__restrict__ out) {
  for (long long k = blockIdx.x * (long long)NT + threadIdx.x; k < n_entries; k += (long long)gridDim.x * NT) {
    const long long c0 = cptr[k], c1 = cptr[k + 1];
    const bool rhs = k >= n_matrix;
    double s = 0.0;
    for (long long c = c0; c < c1; ++c) {
      const int code = ccode[c];
      const int e = code / 12, cc = code - 12 * e;
      double t;
      if (cc < 9) {
        t = da[e] * G[(long long)cc * nel + e] / 4 / area[e];
        if (rhs) t = -(t * ue[(long long)(cc / 3) * nel + e]);
      } else {
        t = be[(long long)(cc - 9) * nel + e];
      }
      s = c == c0 ? t : s + t;
    }
    out[k] = s;
  }
}

// dst[k] = src[perm[k]]
__global__ __launch_bounds__(NT) void k_permute(long long n, const int *__restrict__ perm, const double *__restrict__ src,
                                                double *__restrict__ dst) {
  for (long long k = blockIdx.x * (long long)NT + threadIdx.x; k < n; k += (long long)gridDim.x * NT) dst[k] = src[perm[k]];
}

}  // namespace mi

struct mi_plan_s {
  mi_ctx_s *ctx = nullptr;
  int nel = 0;
  int64_t n_node = 0, n_entries = 0, n_matrix = 0, n_contrib = 0;
  mi::DevBuf<int> cells, ccode;
  mi::DevBuf<long long> cptr;
  mi::DevBuf<double> G, area, ue, be, da, a_stage, out_stage;
};
