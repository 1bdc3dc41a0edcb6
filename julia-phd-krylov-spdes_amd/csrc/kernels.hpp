// Device kernels of libmi355schur (gfx950 / CDNA4, wave64). All arithmetic fp64, built with
// -ffp-contract=off so that element-wise updates and the CSR row sums reproduce the
// reference's mul-then-add sequence bit for bit; only tree reductions reorder sums.
//
// Every kernel that runs inside the Krylov loop takes `const int *done`: once the stop rule
// (cg.jl:34,91) has fired, the remaining launches of a captured iteration chunk return at once.
#pragma once
#include <hip/hip_runtime.h>

#include "exchange_dev.hpp"

namespace mi {

constexpr int NT = 256;        // threads per workgroup for all streaming kernels (4 waves)
constexpr int MAX_PARTS = 512; // max workgroups of a vector kernel = max partial sums per dot

// Scalars of one Krylov solve, resident in HBM (read by every workgroup, written by one).
struct SolverState {
  double rTr, rTz;            // current r'r and r'z
  double rTr_prev, rTz_prev;  // values before this iteration's update (for beta = (1/old)*new)
  double d, alpha, beta;      // diagnostics (each workgroup recomputes them from the partials)
  double tol, bnorm, eps;
  long long it, maxit, res_cap;
  long long it_nxt;           // folded PCG: iteration counter written one launch ahead of `it`
  int done;
  int overflow;               // res_norm capacity hit (BoundsError in the reference)
  int x0_zero;                // the initial guess of this solve is identically zero: `A*x0` need not stream A (set-up only)
};

// What the host reads back after a replay (pinned memory, written by k_solve_end or copied by fetch_flags)
struct PinnedFlags {
  long long it;
  int done;
  int overflow;
  int respec;              // whole-solve graph built for x0 == 0 met a non-zero x0: nothing was done, replay the general one
  int x0z;                 // the initial guess of this solve was identically zero
  long long t_entry, t_exit;  // wall_clock64() (100 MHz) at the entry kernel's start and at the hand-over (MI355_SOLVE_STATS)
  unsigned long long seq;  // whole-solve graph: the solve number, stored LAST by whoever publishes (the host spins on it)
};
// Per-call arguments of a solve whose entry and exit kernels are part of the replayed graph: the host fills this block
// (pinned memory) before the replay, k_solve_begin_g reads it with system-scope loads and leaves a device copy for
// k_solve_end_g.
struct SolveArgs {
  const double *b_in, *x_in;
  double *x_out, *res_stage;
  double eps;
  long long maxit, res_cap, ncap;
  unsigned long long seq;
};

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0 holds the sum
}
// Deterministic workgroup sum, result broadcast to all threads: per-wave shuffle tree, the NTH/64 wave results through
// LDS, then the same shuffle tree over them in every wave (no serial chain). `sm` has NTH/64 doubles (callers size it
// NTH/64 + 1); the trailing barrier lets the next call reuse it.
template <int NTH>
__device__ __forceinline__ double block_sum_t(double v, double *sm) {
  constexpr int NW = NTH / 64;
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double t = lane < NW ? sm[lane] : 0.0;
#pragma unroll
  for (int off = NW / 2; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  t = __shfl(t, 0, 64);
  __syncthreads();
  return t;
}
__device__ __forceinline__ double block_sum(double v, double *sm) { return block_sum_t<NT>(v, sm); }
// Two workgroup sums with ONE barrier: per-wave shuffle trees, then every wave reduces the NTH/64 wave results with a
// second shuffle tree (no serial LDS chain, no second barrier). Fixed tree: deterministic, identical in every workgroup.
// `sm` has 2*(NTH/64) doubles and must not be reused before the next barrier.
template <int NTH>
__device__ __forceinline__ void block_sum2_t(double &a, double &b, double *sm) {
  constexpr int NW = NTH / 64;
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sm[w] = a; sm[NW + w] = b; }
  __syncthreads();
  double ta = lane < NW ? sm[lane] : 0.0, tb = lane < NW ? sm[NW + lane] : 0.0;
#pragma unroll
  for (int off = NW / 2; off > 0; off >>= 1) {
    ta += __shfl_down(ta, off, 64);
    tb += __shfl_down(tb, off, 64);
  }
  a = __shfl(ta, 0, 64);
  b = __shfl(tb, 0, 64);
}
// Sum of `g` per-workgroup partials, in a fixed order, identical in every workgroup.
__device__ __forceinline__ double sum_partials(const double *part, int g, double *sm) {
  double v = 0.0;
  for (int i0 = threadIdx.x; i0 < g; i0 += 8 * NT) {  // eight loads in flight per thread (a plain loop pays one round trip each)
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = i0 + k * NT < g ? part[i0 + k * NT] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += t[k];
  }
  return block_sum(v, sm);
}

// (W*mu)[i] in column order (wm += W[i,k]*mu[k], k ascending — the order of a column-axpy gemv), with the loads of
// eight columns issued together: a plain runtime loop serialises one memory round trip per column.
__device__ __forceinline__ double w_times_mu(const double *__restrict__ W, long long n, long long i,
                                             const double *__restrict__ mu, int nvec) {
  double wm = 0.0;
  for (int q0 = 0; q0 < nvec; q0 += 8) {
    double w[8], m[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = q0 + u < nvec;
      w[u] = ok ? W[(long long)(q0 + u) * n + i] : 0.0;
      m[u] = ok ? mu[q0 + u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (q0 + u < nvec) wm += w[u] * m[u];
  }
  return wm;
}

// ------------------------------------------------------------------ BLAS-1 building blocks
__global__ __launch_bounds__(NT) void k_dot_partial(int n, const double *__restrict__ x,
                                                    const double *__restrict__ y, double *__restrict__ part,
                                                    const int *done) {
  if (done && *done) return;
  __shared__ double sm[NT / 64 + 1];
  double s = 0.0;
  const int stride = gridDim.x * NT;
  for (int i0 = blockIdx.x * NT + threadIdx.x; i0 < n; i0 += 4 * stride) {  // four elements in flight per thread, summed in order
    double a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * stride;
      a[k] = i < n ? x[i] : 0.0;
      b[k] = i < n ? y[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k * stride < n) s += a[k] * b[k];
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// out[0] = sum(part[0..g)) (or its square root)
__global__ __launch_bounds__(NT) void k_finish_sum(const double *part, int g, double *out, int take_sqrt) {
  __shared__ double sm[NT / 64 + 1];
  double s = sum_partials(part, g, sm);
  if (threadIdx.x == 0) out[0] = take_sqrt ? sqrt(s) : s;
}
__global__ __launch_bounds__(NT) void k_axpy(int n, double a, const double *__restrict__ x, double *__restrict__ y) {
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) y[i] = y[i] + a * x[i];
}
__global__ __launch_bounds__(NT) void k_axpby(int n, double a, const double *__restrict__ x, double b,
                                              double *__restrict__ y) {
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) y[i] = a * x[i] + b * y[i];
}
// z = dinv .* r (Jacobi) or z = r (identity)
__global__ __launch_bounds__(NT) void k_diag_apply(int n, const double *__restrict__ dinv,
                                                   const double *__restrict__ r, double *__restrict__ z,
                                                   const int *done) {
  if (done && *done) return;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) z[i] = dinv ? dinv[i] * r[i] : r[i];
}

// ------------------------------------------------------------------ CSR SpMV ("CSR-stream")
// One workgroup owns a block of consecutive rows holding <= SPMV_TILE non-zeros. Phase 1 streams
// the block's column indices and values with coalesced loads, gathers x and parks the products in
// LDS; phase 2 gives each row to one thread, which adds its products left to right — the order
// of the reference's CSC scatter SpMV for a symmetric matrix (stdlib SparseArrays `A*x`,
// `y[rowval[k]] += nzval[k]*x[j]`, j ascending), so y is bit-identical to it.
// Row blocks are dealt to XCDs in contiguous ranges (blockIdx % 8 selects the range) so that each
// XCD's L2 caches one slice of x instead of all of it.
#ifndef MI355_SPMV_TILE
#define MI355_SPMV_TILE 1024   // 512..4096 swept on MI355X at 250k DoF: 1024 gives the shortest launch (profiles/)
#endif
constexpr int SPMV_TILE = MI355_SPMV_TILE;

struct SpmvBlock {  // one record per row block: rows [r0, r1), non-zeros [k0, k1)
  int r0, r1, k0, k1;
};

template <int MODE, bool DOT>  // MODE 0: y = A x, 1: y = yin - A x;  DOT: also part[b] = Σ_{rows of block b} w[r]*y[r]
__global__ __launch_bounds__(NT) void k_spmv_csr(int nblocks, const SpmvBlock *__restrict__ blk,
                                                 const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 const double *__restrict__ val, const double *__restrict__ x,
                                                 const double *yin, double *y, const double *__restrict__ w,
                                                 double *__restrict__ part, const int *done) {
  if (done && *done) return;
  __shared__ __attribute__((aligned(16))) double prod[SPMV_TILE];
  __shared__ double sm[NT / 64 + 1];
  const int per = (nblocks + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (b >= nblocks) return;
  const SpmvBlock bi = blk[b];
  const int r0 = bi.r0, r1 = bi.r1, k0 = bi.k0;
  const int nnz = bi.k1 - k0;
  double wy = 0.0;
  if (nnz <= SPMV_TILE) {
    // row bounds of this thread's first two rows: independent of phase 1, so issue the loads now
    const int ra = r0 + threadIdx.x, rbb = ra + NT;
    int a0 = 0, e0 = 0, a1 = 0, e1 = 0;
    double w0 = 0.0, w1 = 0.0;  // the dot's weights for those rows too: one fewer round trip in the epilogue
    if (ra < r1) { a0 = rowptr[ra] - k0; e0 = rowptr[ra + 1] - k0; if (DOT) w0 = w[ra]; }
    if (rbb < r1) { a1 = rowptr[rbb] - k0; e1 = rowptr[rbb + 1] - k0; if (DOT) w1 = w[rbb]; }
    if ((k0 & 1) == 0) {  // 16-byte aligned values, 8-byte aligned indices: two non-zeros per load
      const int npair = (nnz + 1) >> 1;
      for (int q = threadIdx.x; q < npair; q += NT) {
        const int k = 2 * q;
        if (k + 1 < nnz) {
          const int2 c = *reinterpret_cast<const int2 *>(col + k0 + k);
          const double2 v = *reinterpret_cast<const double2 *>(val + k0 + k);
          *reinterpret_cast<double2 *>(&prod[k]) = make_double2(v.x * x[c.x], v.y * x[c.y]);
        } else {
          prod[k] = val[k0 + k] * x[col[k0 + k]];
        }
      }
    } else {
      for (int k = threadIdx.x; k < nnz; k += NT) prod[k] = val[k0 + k] * x[col[k0 + k]];
    }
    __syncthreads();
    for (int r = ra, i = 0; r < r1; r += NT, ++i) {
      int a, e;
      if (i == 0) { a = a0; e = e0; }
      else if (i == 1) { a = a1; e = e1; }
      else { a = rowptr[r] - k0; e = rowptr[r + 1] - k0; }
      double sum = 0.0;
      for (int k = a; k < e; ++k) sum += prod[k];
      const double yr = MODE ? yin[r] - sum : sum;
      y[r] = yr;
      if (DOT) wy += (i == 0 ? w0 : i == 1 ? w1 : w[r]) * yr;
    }
  } else {
    // a single row longer than the tile (never the case for P1-FEM blocks): tile by tile,
    // thread 0 keeps the running left-to-right sum.
    double sum = 0.0;
    for (int t0 = 0; t0 < nnz; t0 += SPMV_TILE) {
      const int m = min(SPMV_TILE, nnz - t0);
      for (int k = threadIdx.x; k < m; k += NT) prod[k] = val[k0 + t0 + k] * x[col[k0 + t0 + k]];
      __syncthreads();
      if (threadIdx.x == 0)
        for (int k = 0; k < m; ++k) sum += prod[k];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double yr = MODE ? yin[r0] - sum : sum;
      y[r0] = yr;
      if (DOT) wy = w[r0] * yr;
    }
  }
  if (DOT) {
    wy = block_sum(wy, sm);
    if (threadIdx.x == 0) part[b] = wy;
  }
}

// ------------------------------------------------------------------ pcg on a sparse matrix in 2 launches per iteration
// Config 2 (`pcg(A, b, x, M)` on the full matrix, M diagonal or absent; cg.jl:67-109 / 14-50). The three launches of the
// generic loop (SpMV + p'Ap, x/r/z update, p update) become two, and every vector element is written on the XCD that
// reads it next:
//   k_spmv_pcg     : r'r, r'z from the partials -> it += 1, res_norm[it], stop rule, beta (cg.jl:91, 102-106); then the
//                    CSR-stream SpMV of the NEW direction without materialising it first: a gathered entry is
//                    beta*p_old[c] + z[c], computed on the fly from the interleaved pair (p_old[c], z[c]) — one 16-byte
//                    gather per non-zero, the same arithmetic (hence bits) in every workgroup; the row owner stores
//                    p_new[r]; Ap[r]; per-block partial p'Ap (cg.jl:93-94).
//   k_update_xr_blk: alpha = r'z / p'Ap; x += alpha p; r -= alpha Ap; z = M \ r (diagonal); per-block partials r'r, r'z
//                    (cg.jl:95-101), one workgroup per row block with the SpMV's block -> XCD mapping.
// The pairs live in two buffers used alternately (parity of `it`), so no launch reads what it writes; scalars follow the
// it / it_nxt, rTz / rTz_prev hand-off of the folded Schur loop. Start-up: it = 0, p = 0, rTz_prev = 1, so the first
// k_spmv_pcg produces it = 1, res_norm[1] = ||r_0||, p = z_0.
__device__ __forceinline__ int spmv_block_of(int nblocks) {
  const int per = (nblocks + 7) >> 3;
  return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}
__global__ __launch_bounds__(NT, 8) void k_spmv_pcg(int nblocks, const SpmvBlock *__restrict__ blk, const int *__restrict__ rowptr,
                                                 const int *__restrict__ col, const double *__restrict__ val, SolverState *st,
                                                 const double *part_rr, const double *part_rz, int g_vec, double *pz0, double *pz1,
                                                 double *__restrict__ Ap, double *__restrict__ part_pAp, double *res_norm,
                                                 int precond) {
  __shared__ __attribute__((aligned(16))) double prod[SPMV_TILE];
  __shared__ double sm[NT / 64 + 1];
  const int b = spmv_block_of(nblocks);
  const bool has = b < nblocks;
  // block record, state block and stop flag in one memory round trip (the empty asm keeps the compiler from sinking the
  // loads below the early exit)
  const SpmvBlock bi = blk[has ? b : 0];
  const int done0 = st->done;
  const long long it0 = st->it, maxit = st->maxit, cap = st->res_cap;
  const double tol = st->tol, old = precond ? st->rTz_prev : st->rTr_prev;
  asm volatile("" ::"s"(bi.r0), "s"(bi.r1), "s"(bi.k0), "s"(bi.k1), "s"(done0), "s"(it0), "s"(maxit), "s"(cap), "s"(tol), "s"(old));
  // the partial sums of the previous launch are requested FIRST (they need no index): results return in issue order, so the
  // reduction to beta below waits for these two loads only while the non-zeros and their gathered pairs are still in flight
  double rr = (int)threadIdx.x < g_vec ? part_rr[threadIdx.x] : 0.0;
  double rz = (precond && (int)threadIdx.x < g_vec) ? part_rz[threadIdx.x] : 0.0;
  if (done0) return;
  const int r0 = has ? bi.r0 : 0, r1 = has ? bi.r1 : 0, k0 = has ? bi.k0 : 0, nnz = has ? bi.k1 - bi.k0 : 0;
  const double2 *pz_old = reinterpret_cast<const double2 *>((it0 & 1) ? pz1 : pz0);
  double2 *pz_new = reinterpret_cast<double2 *>((it0 & 1) ? pz0 : pz1);
  // everything that does not depend on beta is requested before the reduction: this thread's non-zeros (SPMV_TILE / NT
  // of them) with their gathered pairs, and the bounds and pairs of its first two rows
  constexpr int KPT = SPMV_TILE / NT;
  double vv[KPT];
  double2 gp[KPT];
  const bool fits = nnz <= SPMV_TILE;
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const int k = q * NT + (int)threadIdx.x;
    vv[q] = 0.0; gp[q] = make_double2(0.0, 0.0);
    if (fits && k < nnz) { vv[q] = val[k0 + k]; gp[q] = pz_old[col[k0 + k]]; }
  }
  const int ra = r0 + threadIdx.x;
  int a0 = 0, e0 = 0;
  double2 o0 = make_double2(0.0, 0.0);
  if (ra < r1) { a0 = rowptr[ra] - k0; e0 = rowptr[ra + 1] - k0; o0 = pz_old[ra]; }
  // ---- scalars (identical in every workgroup): g_vec <= NT partials per sum, one load per thread (above), one barrier for both
  __shared__ double sm2[2 * (NT / 64)];
  block_sum2_t<NT>(rr, rz, sm2);
  if (!precond) rz = rr;
  double beta = 1. / old;
  beta *= rz;
  const long long it_new = it0 + 1;
  const double res = sqrt(rr);
  const bool over = it_new > cap;
  const bool stop = over || !((it_new < maxit) && (res > tol));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->rTr = rr; st->rTz = rz; st->beta = beta;
    st->it_nxt = it_new;
    if (!over) res_norm[it_new - 1] = res; else st->overflow = 1;
    if (stop) st->done = 1;
  }
  if (stop || !has) return;
  double wy = 0.0;
  if (fits) {
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      const int k = q * NT + (int)threadIdx.x;
      if (k < nnz) prod[k] = vv[q] * (beta * gp[q].x + gp[q].y);       // A[r,c] * (beta p + z)[c]
    }
    __syncthreads();
    for (int r = ra, i = 0; r < r1; r += NT, ++i) {
      int a, e;
      double2 o;
      if (i == 0) { a = a0; e = e0; o = o0; }
      else { a = rowptr[r] - k0; e = rowptr[r + 1] - k0; o = pz_old[r]; }
      double sum = 0.0;
      for (int k = a; k < e; ++k) sum += prod[k];                      // left to right: the CSC scatter order (bit-exact)
      const double pn = beta * o.x + o.y;                              // axpby!(1, z, beta, p)
      Ap[r] = sum;
      pz_new[r].x = pn;
      wy += pn * sum;
    }
  } else {  // one row longer than the tile: tile by tile, thread 0 keeps the running sum
    double sum = 0.0;
    for (int t0 = 0; t0 < nnz; t0 += SPMV_TILE) {
      const int m = min(SPMV_TILE, nnz - t0);
      for (int k = threadIdx.x; k < m; k += NT) {
        const double2 g = pz_old[col[k0 + t0 + k]];
        prod[k] = val[k0 + t0 + k] * (beta * g.x + g.y);
      }
      __syncthreads();
      if (threadIdx.x == 0)
        for (int k = 0; k < m; ++k) sum += prod[k];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double2 o = pz_old[r0];
      const double pn = beta * o.x + o.y;
      Ap[r0] = sum;
      pz_new[r0].x = pn;
      wy = pn * sum;
    }
  }
  wy = block_sum(wy, sm);
  if (threadIdx.x == 0) part_pAp[b] = wy;
}
// Grid = 8 * nj workgroups: workgroup k works on XCD k & 7 (blocks are dealt round-robin over the XCDs) and takes the
// (k >> 3)-th of nj equal slices of the rows whose SpMV row blocks run on that XCD (`xcd_row[x] .. xcd_row[x + 1]`), so
// that z, r, x are written into the L2 that k_spmv_pcg's gathers and row sums read them from.
__global__ __launch_bounds__(NT) void k_update_xr_blk(const int *__restrict__ xcd_row, SolverState *st, const double *part_pAp,
                                                      int nblocks, double *pz0, double *pz1, const double *__restrict__ Ap,
                                                      double *__restrict__ x, double *__restrict__ r,
                                                      const double *__restrict__ dinv, int diag, int precond,
                                                      double *__restrict__ part_rr, double *__restrict__ part_rz) {
  __shared__ double sm[NT / 64 + 1];
  const int xc = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
  const long long R0 = xcd_row[xc], R1 = xcd_row[xc + 1];
  const int done0 = st->done;
  const long long it_n = st->it_nxt;
  const double rTr0 = st->rTr, rTz0 = st->rTz;
  asm volatile("" ::"s"(R0), "s"(R1), "s"(done0), "s"(it_n), "s"(rTr0), "s"(rTz0));   // one round trip, then the exit test
  if (done0) return;
  const int lo = (int)(R0 + (R1 - R0) * j / nj), hi = (int)(R0 + (R1 - R0) * (j + 1) / nj);
  const double num = precond ? rTz0 : rTr0;
  double2 *pz = reinterpret_cast<double2 *>((it_n & 1) ? pz1 : pz0);     // the pairs k_spmv_pcg has just written p into
  // the partial sums of p'Ap first (up to eight per thread in flight; more in the loop below), then the first four rows of this
  // thread: everything is requested ahead of the reduction, and the reduction waits for the partials only
  double dpart = 0.0;
  double t8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { const int i = (int)threadIdx.x + k * NT; t8[k] = i < nblocks ? part_pAp[i] : 0.0; }
  double pv[4], av[4], xv[4], rv[4], dv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rw = lo + k * NT + (int)threadIdx.x;
    const bool ok = rw < hi;
    pv[k] = ok ? pz[rw].x : 0.0; av[k] = ok ? Ap[rw] : 0.0; xv[k] = ok ? x[rw] : 0.0; rv[k] = ok ? r[rw] : 0.0;
    dv[k] = ok && diag == 2 ? dinv[rw] : 1.0;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) dpart += t8[k];                            // (the order sum_partials uses: bit-identical alpha)
  for (int i0 = (int)threadIdx.x + 8 * NT; i0 < nblocks; i0 += 8 * NT) {
    double u8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) u8[k] = i0 + k * NT < nblocks ? part_pAp[i0 + k * NT] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) dpart += u8[k];
  }
  const double d = block_sum(dpart, sm);
  const double alpha = num / d;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->d = d; st->alpha = alpha;
    st->rTr_prev = rTr0; st->rTz_prev = rTz0;
    st->it = it_n;
  }
  double srr = 0.0, srz = 0.0;
  for (int rw = lo + (int)threadIdx.x, i = 0; rw < hi; rw += NT, ++i) {
    double p_, a_, x_, r_, d_;
    if (i < 4) { p_ = pv[i]; a_ = av[i]; x_ = xv[i]; r_ = rv[i]; d_ = dv[i]; }
    else { p_ = pz[rw].x; a_ = Ap[rw]; x_ = x[rw]; r_ = r[rw]; d_ = diag == 2 ? dinv[rw] : 1.0; }
    x[rw] = x_ + alpha * p_;                     // axpy!(alpha, p, x)
    const double ri = r_ + (-alpha) * a_;        // axpy!(-alpha, Ap, r)
    r[rw] = ri;
    const double zi = diag == 2 ? d_ * ri : ri;  // z .= M \ r
    pz[rw].y = zi;
    srr += ri * ri;
    srz += ri * zi;
  }
  srr = block_sum(srr, sm);
  srz = block_sum(srz, sm);
  if (threadIdx.x == 0) { part_rr[blockIdx.x] = srr; part_rz[blockIdx.x] = srz; }
}
// Start-up of that loop after r_0 (and z_0) exist: pairs (0, z_0) into buffer 0; the g_vec partials = (r'r, r'z, 0, 0, ...);
// scalars as k_fused_residual<., true> leaves them for the folded Schur loop.
__global__ __launch_bounds__(NT) void k_csrfold_start(int n, int g_vec, SolverState *st, const double *part_rr_in,
                                                      const double *part_bb_in, const double *part_rz_in, int g_in,
                                                      const double *__restrict__ z, double *__restrict__ pz0,
                                                      double *__restrict__ part_rr, double *__restrict__ part_rz) {
  __shared__ double sm[NT / 64 + 1];
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    pz0[2 * (long long)i] = 0.0;
    pz0[2 * (long long)i + 1] = z[i];
  }
  for (int i = blockIdx.x * NT + threadIdx.x; i < g_vec; i += gridDim.x * NT)
    if (i > 0) { part_rr[i] = 0.0; part_rz[i] = 0.0; }
  if (blockIdx.x == 0) {
    const double eps = st->eps;
    const double rr = sum_partials(part_rr_in, g_in, sm);
    const double bb = sum_partials(part_bb_in, g_in, sm);
    const double rz = part_rz_in ? sum_partials(part_rz_in, g_in, sm) : rr;
    if (threadIdx.x == 0) {
      part_rr[0] = rr; part_rz[0] = rz;
      st->rTr = rr; st->rTz = rz; st->rTr_prev = 1.0; st->rTz_prev = 1.0;
      st->bnorm = sqrt(bb); st->tol = eps * st->bnorm;
      st->it = 0; st->it_nxt = 0; st->done = 0; st->overflow = 0;
      st->d = 0.0; st->alpha = 0.0; st->beta = 0.0;
    }
  }
}

// ------------------------------------------------------------------ interior CG on the device
// The reference's matrix-free Schur applies solve A_IIdd v = A_IΓdd xd with `IterativeSolvers.cg(A, b; reltol)`
// (EPDD.jl:648-650; third-party, restated from its published iteration, IterativeSolvers.jl cg.jl `CGIterable`):
//     x = 0, r = b, u = 0, residual = ||r||, prev_residual = 1, tol = reltol*residual, maxiter = n
//     while residual > tol and iteration < maxiter:
//         beta = residual^2 / prev_residual^2;  u = r + beta u;  c = A u
//         alpha = residual^2 / (u'c);  x += alpha u;  r -= alpha c;  prev_residual = residual;  residual = ||r||
// All local subdomains are solved at once on the block-diagonal A_II with per-subdomain scalars; row blocks never
// straddle subdomains. One iteration = k_spmv_csr<0,true> (c = A u with the u'c partials) + k_icg_update +
// k_icg_direction. Per-subdomain scalars are double-buffered (cur/nxt) so no launch reads what it writes;
// a converged subdomain is frozen (its workgroups return), the others keep iterating.
// A "piece" of the interior CG's vector kernels: a run of rows of ONE subdomain inside one XCD's share of the SpMV row blocks
struct IcgPiece {
  int lo, hi, dom, slot;  // rows [lo, hi) of subdomain dom; partial slot (consecutive within the subdomain)
  int b0, b1, lead, pad;  // row-block range of the subdomain (partials of u'c); first piece of the subdomain
  int p0, p1;             // partial-slot range of the subdomain (all of its pieces)
};
struct IcgMeta {
  const SpmvBlock *blk;
  const int *blk_dom;          // subdomain of every row block
  const int *dom_b0, *dom_b1;  // row-block range of every subdomain
  double *res_cur, *res_nxt;   // residual norms
  double *tol;
  int *done_cur, *done_nxt, *iters;
  int maxiter_cap;             // 0: maxiter = n_i of the subdomain
  // `Pl` = Diagonal(A): c = Pl \ r, rho = dot(c, r), beta = rho / rho_prev, u = c + beta u, alpha = rho / u'c
  const double *dinv;          // nullptr: unpreconditioned (rho = residual^2)
  double *rho_cur, *rho_nxt;
  double *p_rz;                // per-piece partials of r'z
  const int *dom_p0, *dom_p1;  // partial-slot range of every subdomain (pieces of the vector kernels)
};
__device__ __forceinline__ double icg_dom_sum(const double *part, int b0, int b1, double *sm) {
  double v = 0.0;
  for (int i0 = b0 + (int)threadIdx.x; i0 < b1; i0 += 8 * NT) {  // eight loads in flight per thread, added in order
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = i0 + k * NT < b1 ? part[i0 + k * NT] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += t[k];
  }
  return block_sum(v, sm);
}
// The vector kernels work on "pieces" (IcgPiece: a run of ~2 000 rows of ONE subdomain inside one XCD's share of the SpMV
// row blocks, ~512 workgroups in all, first rows requested ahead of the reduction) instead of one small workgroup per SpMV
// row block (6 800 workgroups of ~146 rows at 1 M DoF, each a chain of round trips): 46 -> 41 us per iteration.
// r = rhs; x = 0; u = z_0 (beta*0); partials of r'r (and r'z)
__global__ __launch_bounds__(NT) void k_icg_init(IcgMeta m, const IcgPiece *__restrict__ pieces, const double *__restrict__ rhs,
                                                 double *__restrict__ x, double *__restrict__ r, double *__restrict__ u,
                                                 double *__restrict__ p_rr) {
  __shared__ double sm[NT / 64 + 1];
  const IcgPiece pc = pieces[blockIdx.x];
  if (pc.hi <= pc.lo) return;
  double s = 0.0, sz = 0.0;
  for (int i = pc.lo + threadIdx.x; i < pc.hi; i += NT) {
    const double v = rhs[i];
    const double z = m.dinv ? m.dinv[i] * v : v;
    r[i] = v; u[i] = z; x[i] = 0.0;
    s += v * v;
    sz += v * z;
  }
  s = block_sum(s, sm);
  if (m.dinv) sz = block_sum(sz, sm);
  if (threadIdx.x == 0) { p_rr[pc.slot] = s; if (m.dinv) m.p_rz[pc.slot] = sz; }
}
// one workgroup per subdomain: residual = ||b||, tol = reltol*residual, flags
__global__ __launch_bounds__(NT) void k_icg_start(IcgMeta m, const double *p_rr, double reltol) {
  __shared__ double sm[NT / 64 + 1];
  const int d = blockIdx.x;
  const double rr = icg_dom_sum(p_rr, m.dom_p0[d], m.dom_p1[d], sm);
  const double rz = m.dinv ? icg_dom_sum(m.p_rz, m.dom_p0[d], m.dom_p1[d], sm) : 0.0;
  if (threadIdx.x == 0) {
    const double res = sqrt(rr);
    m.res_cur[d] = res; m.res_nxt[d] = res;
    const double rho = m.dinv ? rz : res * res;
    m.rho_cur[d] = rho; m.rho_nxt[d] = rho;
    m.tol[d] = reltol * res;
    m.iters[d] = 0;
    const int dn = res <= m.tol[d];
    m.done_cur[d] = dn; m.done_nxt[d] = dn;
  }
}
// alpha = rho / (u'c); x += alpha u; r -= alpha c; partials of r'r (and r'z)
__global__ __launch_bounds__(NT) void k_icg_update(IcgMeta m, const IcgPiece *__restrict__ pieces, const double *__restrict__ p_uc,
                                                   const double *__restrict__ u, const double *__restrict__ c, double *__restrict__ x,
                                                   double *__restrict__ r, double *__restrict__ p_rr) {
  __shared__ double sm[NT / 64 + 1];
  const IcgPiece pc = pieces[blockIdx.x];
  if (pc.hi <= pc.lo) return;
  const int d = pc.dom;
  const bool lead = pc.lead && threadIdx.x == 0;
  const int dn = m.done_nxt[d];
  const double res = m.res_nxt[d], rho_n = m.rho_nxt[d];
  // the first four rows of this thread: requested together with the scalars, ahead of the reduction
  double uv[4], cv[4], xv[4], rv[4], dv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rw = pc.lo + k * NT + (int)threadIdx.x;
    const bool ok = rw < pc.hi;
    uv[k] = ok ? u[rw] : 0.0; cv[k] = ok ? c[rw] : 0.0; xv[k] = ok ? x[rw] : 0.0; rv[k] = ok ? r[rw] : 0.0;
    dv[k] = ok && m.dinv ? m.dinv[rw] : 1.0;
  }
  if (dn) { if (lead) m.done_cur[d] = 1; return; }
  const double uc = icg_dom_sum(p_uc, pc.b0, pc.b1, sm);
  const double rho = m.dinv ? rho_n : res * res;
  const double alpha = rho / uc;
  double s = 0.0, sz = 0.0;
  for (int rw = pc.lo + (int)threadIdx.x, i = 0; rw < pc.hi; rw += NT, ++i) {
    double u_, c_, x_, r_, d_;
    if (i < 4) { u_ = uv[i]; c_ = cv[i]; x_ = xv[i]; r_ = rv[i]; d_ = dv[i]; }
    else { u_ = u[rw]; c_ = c[rw]; x_ = x[rw]; r_ = r[rw]; d_ = m.dinv ? m.dinv[rw] : 1.0; }
    x[rw] = x_ + alpha * u_;
    const double ri = r_ - alpha * c_;
    r[rw] = ri;
    s += ri * ri;
    if (m.dinv) sz += ri * (d_ * ri);
  }
  s = block_sum(s, sm);
  if (m.dinv) sz = block_sum(sz, sm);
  if (threadIdx.x == 0) { p_rr[pc.slot] = s; if (m.dinv) m.p_rz[pc.slot] = sz; }
  if (lead) { m.res_cur[d] = res; m.rho_cur[d] = rho; m.done_cur[d] = 0; }
}
// residual = ||r||; beta = residual^2 / prev_residual^2 (rho / rho_prev with Pl); u = z + beta u; iteration count and stop test
__global__ __launch_bounds__(NT) void k_icg_direction(IcgMeta m, const IcgPiece *__restrict__ pieces, const double *__restrict__ p_rr,
                                                      const double *__restrict__ r, double *__restrict__ u,
                                                      const int *__restrict__ n_i) {
  __shared__ double sm[NT / 64 + 1];
  const IcgPiece pc = pieces[blockIdx.x];
  if (pc.hi <= pc.lo) return;
  const int d = pc.dom;
  const int dc = m.done_cur[d];
  const double prev = m.res_cur[d], rho_c = m.rho_cur[d];
  double rv[4], uv[4], dv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rw = pc.lo + k * NT + (int)threadIdx.x;
    const bool ok = rw < pc.hi;
    rv[k] = ok ? r[rw] : 0.0; uv[k] = ok ? u[rw] : 0.0; dv[k] = ok && m.dinv ? m.dinv[rw] : 1.0;
  }
  if (dc) return;
  const double rr = icg_dom_sum(p_rr, pc.p0, pc.p1, sm);
  const double res = sqrt(rr);
  double beta = (res * res) / (prev * prev), rho = 0.0;
  if (m.dinv) {
    rho = icg_dom_sum(m.p_rz, pc.p0, pc.p1, sm);
    beta = rho / rho_c;
  }
  for (int rw = pc.lo + (int)threadIdx.x, i = 0; rw < pc.hi; rw += NT, ++i) {
    double r_, u_, d_;
    if (i < 4) { r_ = rv[i]; u_ = uv[i]; d_ = dv[i]; }
    else { r_ = r[rw]; u_ = u[rw]; d_ = m.dinv ? m.dinv[rw] : 1.0; }
    u[rw] = (m.dinv ? d_ * r_ : r_) + beta * u_;
  }
  if (pc.lead && threadIdx.x == 0) {
    const int it = m.iters[d] + 1;
    m.iters[d] = it;
    m.res_nxt[d] = res;
    if (m.dinv) m.rho_nxt[d] = rho;
    const int maxiter = m.maxiter_cap > 0 ? m.maxiter_cap : n_i[d];
    m.done_nxt[d] = (res <= m.tol[d]) || (it >= maxiter);
  }
}

// ---- the same interior CG in 2 launches per iteration (round 2): the config-2 loop above with per-subdomain scalars.
//   k_icg_spmv  : per subdomain: r'r (and r'z) from the partials -> residual, stop test, beta; the CSR-stream SpMV of the NEW
//                 direction u = z + beta u_old computed on the fly from the gathered (u_old, z) pairs; the row owner stores
//                 u_new, c = A u_new, per-block partial u'c.
//   k_icg_update: alpha = rho / u'c; x += alpha u; r -= alpha c; z = r (or dinv .* r: `Pl` = diagonal); per-piece partials.
// A "piece" is a contiguous run of rows of ONE subdomain that also lies inside one XCD's share of the SpMV row blocks, so
// every vector entry is written into the L2 that gathers it next; pieces of a subdomain own consecutive partial slots.
// Subdomain scalars are double-buffered (cur: written by k_icg_update, read by k_icg_spmv; nxt: the other way round);
// a converged subdomain is frozen: both kernels return at once for its workgroups.
struct IcgDomState {
  double rho_prev, tol, res;
  int it, done;
};
struct IcgBlkInfo {       // per SpMV row block, next to its SpmvBlock record: one load instead of a chain of index lookups
  int dom, p0, np, lead;  // subdomain; its partial-slot range (pieces); first block of the subdomain
  int maxit, pad0, pad1, pad2;
};
struct IcgFold {
  const SpmvBlock *blk;
  const IcgBlkInfo *binfo;
  IcgDomState *cur, *nxt;
  double *ur0, *ur1;                     // interleaved (u, z) pairs, two buffers (parity of the subdomain's `it`)
  double *c, *r;
  double *part_uc, *part_rr, *part_rz;
  const double *dinv;                    // nullptr: unpreconditioned
  double reltol;
};
__global__ __launch_bounds__(NT, 8) void k_icg_spmv(int nblocks, IcgFold m, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 const double *__restrict__ val) {
  __shared__ __attribute__((aligned(16))) double prod[SPMV_TILE];
  __shared__ double sm[NT / 64 + 1];
  __shared__ double sm2[2 * (NT / 64)];
  const int b = spmv_block_of(nblocks);
  if (b >= nblocks) return;
  const IcgBlkInfo inf = m.binfo[b];
  const SpmvBlock bi = m.blk[b];
  const int d = inf.dom;
  // the subdomain's partials are requested together with its scalars (both addresses come from the block's own record)
  const int p0 = inf.p0, np = inf.np;
  double rr = (int)threadIdx.x < np ? m.part_rr[p0 + threadIdx.x] : 0.0;
  double rz = (m.dinv && (int)threadIdx.x < np) ? m.part_rz[p0 + threadIdx.x] : 0.0;
  const IcgDomState S = m.cur[d];
  if (S.done) return;
  const int r0 = bi.r0, r1 = bi.r1, k0 = bi.k0, nnz = bi.k1 - k0;
  const double2 *ur_old = reinterpret_cast<const double2 *>((S.it & 1) ? m.ur1 : m.ur0);
  double2 *ur_new = reinterpret_cast<double2 *>((S.it & 1) ? m.ur0 : m.ur1);
  constexpr int KPT = SPMV_TILE / NT;
  double vv[KPT];
  double2 gp[KPT];
  const bool fits = nnz <= SPMV_TILE;
#pragma unroll
  for (int q = 0; q < KPT; ++q) {
    const int k = q * NT + (int)threadIdx.x;
    vv[q] = 0.0; gp[q] = make_double2(0.0, 0.0);
    if (fits && k < nnz) { vv[q] = val[k0 + k]; gp[q] = ur_old[col[k0 + k]]; }
  }
  const int ra = r0 + threadIdx.x;
  int a0 = 0, e0 = 0;
  double2 o0 = make_double2(0.0, 0.0);
  if (ra < r1) { a0 = rowptr[ra] - k0; e0 = rowptr[ra + 1] - k0; o0 = ur_old[ra]; }
  // ---- scalars of this subdomain (identical in all of its workgroups): at most NT pieces per subdomain
  block_sum2_t<NT>(rr, rz, sm2);
  const double res = sqrt(rr);
  const double rho = m.dinv ? rz : res * res;          // IterativeSolvers: residual^2, resp. dot(c, r) with Pl
  const double tol = S.it == 0 ? m.reltol * res : S.tol;
  const bool stop = !(res > tol) || S.it >= inf.maxit;  // `while residual > tol && iteration < maxiter`, maxiter = size(A, 2)
  if (inf.lead && threadIdx.x == 0) {
    IcgDomState N;
    N.rho_prev = rho; N.tol = tol; N.res = res; N.it = S.it + (stop ? 0 : 1); N.done = stop;
    m.nxt[d] = N;
  }
  if (stop) return;
  const double beta = rho / S.rho_prev;
  double wy = 0.0;
  if (fits) {
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      const int k = q * NT + (int)threadIdx.x;
      if (k < nnz) prod[k] = vv[q] * (beta * gp[q].x + gp[q].y);
    }
    __syncthreads();
    for (int r = ra, i = 0; r < r1; r += NT, ++i) {
      int a, e;
      double2 o;
      if (i == 0) { a = a0; e = e0; o = o0; }
      else { a = rowptr[r] - k0; e = rowptr[r + 1] - k0; o = ur_old[r]; }
      double sum = 0.0;
      for (int k = a; k < e; ++k) sum += prod[k];
      const double un = beta * o.x + o.y;              // u = c + beta u
      m.c[r] = sum;
      ur_new[r].x = un;
      wy += un * sum;
    }
  } else {
    double sum = 0.0;
    for (int t0 = 0; t0 < nnz; t0 += SPMV_TILE) {
      const int mm = min(SPMV_TILE, nnz - t0);
      for (int k = threadIdx.x; k < mm; k += NT) {
        const double2 g = ur_old[col[k0 + t0 + k]];
        prod[k] = val[k0 + t0 + k] * (beta * g.x + g.y);
      }
      __syncthreads();
      if (threadIdx.x == 0)
        for (int k = 0; k < mm; ++k) sum += prod[k];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double2 o = ur_old[r0];
      const double un = beta * o.x + o.y;
      m.c[r0] = sum;
      ur_new[r0].x = un;
      wy = un * sum;
    }
  }
  wy = block_sum(wy, sm);
  if (threadIdx.x == 0) m.part_uc[b] = wy;
}
// One workgroup per piece (pieces[blockIdx.x]; hi <= lo: padding of the XCD interleave).
__global__ __launch_bounds__(NT) void k_icg_update_blk(IcgFold m, const IcgPiece *__restrict__ pieces, double *__restrict__ x) {
  __shared__ double sm[NT / 64 + 1];
  const IcgPiece pc = pieces[blockIdx.x];
  if (pc.hi <= pc.lo) return;
  const int d = pc.dom;
  const IcgDomState N = m.nxt[d];
  const bool lead = pc.lead && threadIdx.x == 0;
  if (N.done) {
    if (lead) { IcgDomState C = N; m.cur[d] = C; }
    return;
  }
  double2 *ur = reinterpret_cast<double2 *>((N.it & 1) ? m.ur1 : m.ur0);   // the pairs k_icg_spmv has just written u into
  double uv[4], cv[4], xv[4], rv[4], dv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rw = pc.lo + k * NT + (int)threadIdx.x;
    const bool ok = rw < pc.hi;
    uv[k] = ok ? ur[rw].x : 0.0; cv[k] = ok ? m.c[rw] : 0.0; xv[k] = ok ? x[rw] : 0.0; rv[k] = ok ? m.r[rw] : 0.0;
    dv[k] = ok && m.dinv ? m.dinv[rw] : 1.0;
  }
  const double uc = icg_dom_sum(m.part_uc, pc.b0, pc.b1, sm);
  const double alpha = N.rho_prev / uc;                 // nxt.rho_prev holds this iteration's rho
  if (lead) { IcgDomState C = N; C.done = 0; m.cur[d] = C; }
  double srr = 0.0, srz = 0.0;
  for (int rw = pc.lo + (int)threadIdx.x, i = 0; rw < pc.hi; rw += NT, ++i) {
    double u_, c_, x_, r_, d_;
    if (i < 4) { u_ = uv[i]; c_ = cv[i]; x_ = xv[i]; r_ = rv[i]; d_ = dv[i]; }
    else { u_ = ur[rw].x; c_ = m.c[rw]; x_ = x[rw]; r_ = m.r[rw]; d_ = m.dinv ? m.dinv[rw] : 1.0; }
    x[rw] = x_ + alpha * u_;
    const double ri = r_ - alpha * c_;
    m.r[rw] = ri;
    const double zi = m.dinv ? d_ * ri : ri;
    ur[rw].y = zi;
    srr += ri * ri;
    srz += ri * zi;
  }
  srr = block_sum(srr, sm);
  srz = block_sum(srz, sm);
  if (threadIdx.x == 0) { m.part_rr[pc.slot] = srr; m.part_rz[pc.slot] = srz; }
}
// x = 0, r = rhs, pairs (0, z_0) into buffer 0, partials of r'r / r'z; the lead piece of a subdomain resets its scalars
__global__ __launch_bounds__(NT) void k_icg_fold_init(IcgFold m, const IcgPiece *__restrict__ pieces, const double *__restrict__ rhs,
                                                      double *__restrict__ x) {
  __shared__ double sm[NT / 64 + 1];
  const IcgPiece pc = pieces[blockIdx.x];
  if (pc.hi <= pc.lo) return;
  double2 *ur = reinterpret_cast<double2 *>(m.ur0);
  double srr = 0.0, srz = 0.0;
  for (int rw = pc.lo + (int)threadIdx.x; rw < pc.hi; rw += NT) {
    const double ri = rhs[rw];
    const double zi = m.dinv ? m.dinv[rw] * ri : ri;
    x[rw] = 0.0; m.r[rw] = ri;
    ur[rw] = make_double2(0.0, zi);
    srr += ri * ri; srz += ri * zi;
  }
  srr = block_sum(srr, sm);
  srz = block_sum(srz, sm);
  if (threadIdx.x == 0) {
    m.part_rr[pc.slot] = srr; m.part_rz[pc.slot] = srz;
    if (pc.lead) {
      IcgDomState C;
      C.rho_prev = 1.0; C.tol = 0.0; C.res = 0.0; C.it = 0; C.done = 0;
      m.cur[pc.dom] = C; m.nxt[pc.dom] = C;
    }
  }
}

// dinv[r] = 1 / A[r,r] of a CSR matrix (1 if the row stores no diagonal entry)
__global__ __launch_bounds__(NT) void k_csr_inv_diag(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const double *__restrict__ val, double *__restrict__ dinv) {
  const int r = blockIdx.x * NT + threadIdx.x;
  if (r >= n) return;
  double dg = 1.0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
    if (col[k] == r) dg = 1.0 / val[k];
  dinv[r] = dg;
}

// ------------------------------------------------------------------ batched dense GEMV with fused gather
// y_d = M_d * (D_d R_d x)  [* D_d], for all subdomains d of this rank in one launch.
//   S-apply  (SCALE=false): M_d = S_d,  EPDD.jl:775-778  (gather, `Sd[idom]*xd`)
//   NN-apply (SCALE=true) : M_d = ΠS_d, EPDD.jl:1373-1381 (gather r/cnt, `ΠSd[idom]*rd`, result /cnt)
// Layout: every M_d is stored row-major with its leading dimension padded to a multiple of 16
// doubles (128 B), so each row is a contiguous, line-aligned stream. A workgroup owns RPW rows per
// wave (WAVES*RPW rows, 32 by default) described by ONE 32-byte tile record. The operand gather
// (index -> x, two dependent loads) is requested first and the first group of matrix loads right
// behind it — vector-memory results return in issue order — so the stream is in flight while x_d is
// staged into LDS (padded with zeros); then one group of 4 x 16 B per lane and row at a time, 16 waves
// per CU keeping 128 KB in flight. Each lane multiplies against the LDS copy of x_d and the row sum is
// finished with a wave shuffle tree.
// Output: row r of subdomain d goes to out_pos[...] = g*W + j, the j-th contribution slot of Γ node
// g (slots in ascending subdomain order, W = max multiplicity; unused slots stay 0). The
// scatter-add over subdomains `Sx[lΓ] += Sdxd[lΓd]` (EPDD.jl:779-781 / 1379-1381) is then a
// contiguous W-term sum per Γ node, taken by the consumer kernel in the reference's idom order.
struct GemvTile {
  long long mat_off;  // element offset of the subdomain block
  int n, ld;          // n_Γd and padded leading dimension
  int loc_off, row0;  // offset of the block in the local index space, first row of this tile
  int active, nrows;  // active 0: the block of this subdomain lives on another rank (multi-GPU): nothing to stream here;
                      // otherwise 1 + the slot of this tile's partial dot products in the folded launches.
                      // nrows: rows [row0, row0 + nrows) whose owner duties this tile performs in the folded launches
                      // (active: the WAVES*RPW streamed rows; inactive: up to one row per thread)
};
struct DenseMeta {
  const double *M;        // all blocks of this rank, row-major, padded
  const GemvTile *tiles;  // [ntiles]
  const int *gidx;        // [nloc] Γ index of every local row/column
  const double *cnt;      // [nloc] node_Γ_cnt as double (NN only)
  const int *out_pos;     // [nloc] slot g*W + j of every local row
};
constexpr int GEMV_PANEL = 2048;  // doubles of x_d staged per pass (16 KiB LDS)
#ifndef MI355_OPERAND_FIRST
#define MI355_OPERAND_FIRST 1   // 0: matrix stream first (measured 7 % slower in the folded PCG launches)
#endif
#ifndef MI355_GEMV_GU
#define MI355_GEMV_GU 4        // 16-byte loads per lane and row in one group (2 and 8 measured slower, profiles/)
#endif
constexpr int GU = MI355_GEMV_GU;
// Default-policy loads on purpose: the 136 MB working set of a PCG iteration is re-read from the Infinity
// Cache every iteration; non-temporal loads made the stand-alone GEMV 3 % faster and the solve 4 % slower.
__device__ __forceinline__ double2 gemv_ld(const double *p) { return *reinterpret_cast<const double2 *>(p); }
template <int RPW>
__device__ __forceinline__ void gemv_load_group(double2 (&mv)[RPW][GU], const double *const (&rowp)[RPW], int col0,
                                                int cb, int pw, int lane) {
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    const int c = cb + u * 128 + lane * 2;
#pragma unroll
    for (int k = 0; k < RPW; ++k)
      mv[k][u] = (c < pw) ? gemv_ld(rowp[k] + col0 + c) : make_double2(0.0, 0.0);
  }
}
template <int RPW>
__device__ __forceinline__ void gemv_fma_group(double (&acc)[RPW], const double2 (&mv)[RPW][GU], const double *xs, int cb,
                                               int pw, int lane) {
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    const int c = cb + u * 128 + lane * 2;
    const double2 xv = (c < pw) ? *reinterpret_cast<const double2 *>(&xs[c]) : make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      acc[k] += mv[k][u].x * xv.x;
      acc[k] += mv[k][u].y * xv.y;
    }
  }
}
// Row streamer of one tile: RPW rows per wave, 16 B per lane and row per load, groups of 4 loads.
// `begin` issues the first group of matrix loads (so the stream is in flight while the caller
// stages the column values into LDS), `panel` consumes one staged panel, `finish` reduces.
template <int RPW>
struct GemvRows {
  const double *rowp[RPW];
  double acc[RPW];
  double2 buf[RPW][GU];
  int lane, ld;
  __device__ __forceinline__ void begin(const DenseMeta &m, const GemvTile &t) {
    lane = threadIdx.x & 63;
    ld = t.ld;
    const int row_base = t.row0 + (threadIdx.x >> 6) * RPW;
    const double *Md = m.M + t.mat_off;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int r = min(row_base + k, t.n - 1);  // clamp: tail rows re-read a valid row, result dropped
      rowp[k] = Md + (long long)r * ld;
      acc[k] = 0.0;
    }
    gemv_load_group<RPW>(buf, rowp, 0, 0, min(GEMV_PANEL, ld), lane);
  }
  // xs holds columns [c0, c0 + pw) of the operand; for c0 > 0 the first group is loaded here
  __device__ __forceinline__ void panel(const double *xs, int c0, int pw) {
    if (c0) gemv_load_group<RPW>(buf, rowp, c0, 0, pw, lane);
    for (int cb = 0; cb < pw; cb += 128 * GU) {
      if (cb) gemv_load_group<RPW>(buf, rowp, c0, cb, pw, lane);
      gemv_fma_group<RPW>(acc, buf, xs, cb, pw, lane);
    }
  }
  __device__ __forceinline__ void finish(double (&sum)[RPW]) {
#pragma unroll
    for (int k = 0; k < RPW; ++k) sum[k] = wave_sum(acc[k]);  // valid in lane 0
  }
};

template <int RPW, bool SCALE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gemv_batched(DenseMeta m, const double *__restrict__ x,
                                                             double *__restrict__ yslots, const int *done,
                                                             const int *zero_x) {
  constexpr int NTH = 64 * WAVES;
  // tile record and both flags in ONE memory round trip (tested one after the other they cost three, 1.2-2 us each)
  const GemvTile t = m.tiles[blockIdx.x];
  const int done0 = done ? *done : 0, zero0 = zero_x ? *zero_x : 0;
  asm volatile("" ::"s"(t.mat_off), "s"(t.n), "s"(t.ld), "s"(t.loc_off), "s"(t.row0), "s"(t.active), "s"(done0), "s"(zero0));
  if (done0) return;
  __shared__ __attribute__((aligned(16))) double xs[GEMV_PANEL];
  if (!t.active) return;
  const int off = t.loc_off, n = t.n;
  if (zero0) {  // x is identically zero (set-up of a solve from x0 = 0): the products are +0, nothing to stream
    const int row_base = t.row0 + (threadIdx.x >> 6) * RPW;
    if ((threadIdx.x & 63) == 0)
      for (int k = 0; k < RPW; ++k)
        if (row_base + k < n) yslots[m.out_pos[off + row_base + k]] = 0.0;
    return;
  }
  GemvRows<RPW> rows;
#if !MI355_OPERAND_FIRST
  rows.begin(m, t);
#endif
  for (int c0 = 0; c0 < t.ld; c0 += GEMV_PANEL) {
    const int pw = min(GEMV_PANEL, t.ld - c0);  // multiple of 16
    if (c0) __syncthreads();
#if MI355_OPERAND_FIRST
    // Vector-memory results return in issue order: the operand gather (index -> value, two dependent loads) is issued
    // BEFORE the first group of the matrix stream, so staging waits for its own loads only, not for 128 KB of matrix.
    constexpr int XPT = (GEMV_PANEL + NTH - 1) / NTH;
    int gi[XPT];
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      const int j = c0 + q * NTH + (int)threadIdx.x;
      gi[q] = (q * NTH + (int)threadIdx.x < pw && j < n) ? m.gidx[off + j] : -1;
    }
    double xv[XPT];
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      const int j = c0 + q * NTH + (int)threadIdx.x;
      xv[q] = 0.0;
      if (gi[q] >= 0) {
        xv[q] = x[gi[q]];
        if (SCALE) xv[q] = xv[q] / m.cnt[off + j];
      }
    }
    if (c0 == 0) rows.begin(m, t);
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      const int l = q * NTH + (int)threadIdx.x;
      if (l < pw) xs[l] = xv[q];
    }
#else
    for (int l = threadIdx.x; l < pw; l += NTH) {
      const int j = c0 + l;
      double v = 0.0;
      if (j < n) {
        v = x[m.gidx[off + j]];
        if (SCALE) v = v / m.cnt[off + j];
      }
      xs[l] = v;
    }
#endif
    __syncthreads();
    rows.panel(xs, c0, pw);
  }
  double sum[RPW];
  rows.finish(sum);
  const int row_base = t.row0 + (threadIdx.x >> 6) * RPW;
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int r = row_base + k;
    if (rows.lane == 0 && r < n) yslots[m.out_pos[off + r]] = SCALE ? sum[k] / m.cnt[off + r] : sum[k];
  }
}

// The same GEMV for KV operand vectors at once: S_d is streamed ONCE for up to KV columns of X (`WtA[i,:] = A*W[:,i]`
// of the deflated solvers, defcg.jl:41-44 / 261-264; `AV[:, j] = A*V[:, j]` of eigpcg's restart, eigcg.jl:233-240) instead of once
// per column. Per (row, vector) the products are accumulated in the order of k_gemv_batched: results are bit-identical
// to KV single applies. X is column-major with leading dimension ldx; vector v writes its slot table at v*slot_stride.
template <int RPW, int KV, bool SCALE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gemv_multi(DenseMeta m, const double *__restrict__ X, long long ldx, int kv,
                                                           double *__restrict__ yslots, long long slot_stride) {
  constexpr int NTH = 64 * WAVES;
  __shared__ __attribute__((aligned(16))) double xs[KV][GEMV_PANEL];
  const GemvTile t = m.tiles[blockIdx.x];
  if (!t.active) return;
  const int off = t.loc_off, n = t.n, lane = threadIdx.x & 63;
  const int row_base = t.row0 + (threadIdx.x >> 6) * RPW;
  const double *rowp[RPW];
  double acc[RPW][KV];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    rowp[k] = m.M + t.mat_off + (long long)min(row_base + k, n - 1) * t.ld;
#pragma unroll
    for (int v = 0; v < KV; ++v) acc[k][v] = 0.0;
  }
  for (int c0 = 0; c0 < t.ld; c0 += GEMV_PANEL) {
    const int pw = min(GEMV_PANEL, t.ld - c0);
    if (c0) __syncthreads();
    for (int l = threadIdx.x; l < pw; l += NTH) {
      const int j = c0 + l;
      const int gi = j < n ? m.gidx[off + j] : -1;
      const double cn = (SCALE && j < n) ? m.cnt[off + j] : 1.0;
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        double xv = 0.0;
        if (gi >= 0 && v < kv) {
          xv = X[(long long)v * ldx + gi];
          if (SCALE) xv = xv / cn;
        }
        xs[v][l] = xv;
      }
    }
    __syncthreads();
    double2 buf[RPW][GU];
    for (int cb = 0; cb < pw; cb += 128 * GU) {
      gemv_load_group<RPW>(buf, rowp, c0, cb, pw, lane);
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int c = cb + u * 128 + lane * 2;
        if (c < pw) {
#pragma unroll
          for (int v = 0; v < KV; ++v) {
            const double2 xv = *reinterpret_cast<const double2 *>(&xs[v][c]);
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
              acc[k][v] += buf[k][u].x * xv.x;
              acc[k][v] += buf[k][u].y * xv.y;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int r = row_base + k;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const double sum = wave_sum(acc[k][v]);
      if (lane == 0 && r < n && v < kv)
        yslots[(long long)v * slot_stride + m.out_pos[off + r]] = SCALE ? sum / m.cnt[off + r] : sum;
    }
  }
}
// Γ-sum of the slot tables of k_gemv_multi: grid (x, kv); vector v -> Y[:, v] (leading dimension ldy)
__global__ __launch_bounds__(NT) void k_assemble_slots_multi(int n, int width, const double *__restrict__ yslots,
                                                             long long slot_stride, double *__restrict__ Y, long long ldy) {
  const double *ys = yslots + (long long)blockIdx.y * slot_stride;
  double *y = Y + (long long)blockIdx.y * ldy;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    double s = 0.0;
    for (int j = 0; j < width; ++j) s += ys[(long long)i * width + j];
    y[i] = s;
  }
}

// ------------------------------------------------------------------ PCG folded into the two GEMVs (2 launches / iteration)
// For pcg(S, b, x, ΠSnn) with both operators dense and built on the same subdomain maps, the vector
// work of an iteration is folded into the prologue/epilogue of the two GEMV launches:
//
//   PHASE 1 (ΠS GEMV): d = Σ partial(p'Ap); alpha = r'z / d (cg.jl:94-95); r_new = r - alpha Ap for this
//            workgroup's columns (Ap = Γ-sum of the S contributions) -> LDS, scaled by 1/cnt; the workgroup that
//            owns a Γ node (first contributing subdomain) stores r_new, x += alpha p (cg.jl:97-98); GEMV;
//            epilogue: z contributions, partial r'z and partial r'r (cg.jl:99-101).
//   PHASE 0 (S GEMV):  r'r, r'z from the partials; it += 1; res_norm[it] = sqrt(r'r); stop rule (cg.jl:91,
//            104-105) — evaluated identically by every workgroup; beta = (1/old)*new; p_new = beta p + z for this
//            workgroup's columns (z = Γ-sum of the ΠS contributions) -> LDS (cg.jl:102-103); owners store p_new;
//            GEMV; epilogue: S contributions, partial p'Ap (cg.jl:93-94).
//
// A dot product is therefore summed per GEMV row and then over tiles (Σ_d Σ_rows p_g y_row = Σ_g p_g Σ_d y),
// a re-association of the same terms.
// Everything the prologue reads is kept in LOCAL order (one entry per (subdomain, Γ_d slot)), so the
// loads of a workgroup are contiguous and need no index chain: a producer writes its row result
// into the contribution row of every subdomain that shares the Γ node (`tgt`, <= W tiny stores), and
// the owner of a node writes the updated vector entry into every sharing subdomain's copy (`peer`).
// r and p have a "current" copy (read by everyone) and a "next" copy (written by owners); the owners
// of the following launch copy next -> current, so no launch reads a buffer it writes. Scalars follow
// the same rule (it/it_nxt, rTz/rTz_prev).
// Start-up: it_nxt = 0 marks the first PHASE 1 launch (alpha = 0, so r_new = r_0 and x is untouched;
// p = 0 and rTz_prev = 1 make the first PHASE 0 produce p = z_0, it = 1, res_norm[1] = ||r_0||).
struct PcgFold {
  SolverState *st;
  const double *con_in;     // [nloc*W] contributions to sum: S (PHASE 1) / ΠS (PHASE 0), local order
  double *con_out;          // [nloc*W]
  const double *part_in0;   // PHASE 1: partial p'Ap         PHASE 0: partial r'r
  const double *part_in1;   //                               PHASE 0: partial r'z
  int n_in;
  double *part_out0;        // PHASE 1: partial r'r          PHASE 0: partial p'Ap
  double *part_out1;        // PHASE 1: partial r'z
  double *r_cur, *r_nxt, *p_cur, *p_nxt;  // [nloc] local-order copies
  double *x;                // [n_Γ]
  const double *r_gamma;    // [n_Γ] r_0 in Γ order: the first PHASE 1 launch gathers it (no separate scatter pass)
  double *res_norm;
  const int *tgt;           // [nloc*W] where this row's result goes in each sharing subdomain's contribution row (-1 pad)
  const int *peer;          // [nloc*W] local positions of the same Γ node in the sharing subdomains (-1 pad)
  const int *jrank;         // [nloc] rank of this subdomain among the contributors of the node (0 = owner)
  int W;
  // inputs that come out of a peer exchange (exchange.hpp) are double-buffered by the parity of the exchange number:
  // con_in / part_in0 / part_in1 point at copy 0, copy (*in_epoch & 1) is in_stride doubles further
  const unsigned long long *in_epoch;
  long long in_stride;
  // outputs of a launch that is sharded over ranks and exchanged by peer stores (exchange.hpp): con_out / part_out0 /
  // part_out1 then point at copy 0 of the table in the OWN arena; every streamed tile stores its results into copy
  // ((epoch + 1) & 1) of the table in EVERY arena, publishes them and counts itself in; the last of the n_arrive tiles
  // stores exchange number epoch + 1 into this rank's flag of every arena. The wait (and the advance of `epoch`) is a
  // one-wave kernel behind the launch — a streaming launch never spins.
  const XchgPeers *xp;      // device copy (null: outputs stay local)
  XchgState *xst;
  long long out_stride;
  unsigned int n_arrive;
  // x_inwait (every rank on a GPU of its own — a waiting launch keeps its compute units): no kernel between the launches.
  // A launch whose inputs come out of an exchange (in_stride != 0) waits for the flags itself, with its first matrix loads
  // already in flight. The exchange number is then carried like it / it_nxt: a PHASE p launch reads xst->xep[p] and its
  // lead thread writes xst->xep[1 - p] (the launches alternate), so no workgroup reads a word its own launch writes.
  const XchgPeers *xpw;     // peers to wait for (device copy)
  int x_inwait;
  // deflation (defcg.jl:291-305; nvec == 0: plain pcg). PHASE 1 also leaves per-tile partials of WtA*z; k_defl_mu turns
  // them into mu = WtAW \ (WtA*z) and (W*mu) in local order; PHASE 0 subtracts that from beta*p + z.
  int nvec;
  long long n_gamma;
  const double *AW;         // [nvec * n_Γ] WtA[v, :] = A*W[:, v], Γ order
  double *part_mu;          // [nvec * ntiles(ΠS)] layout v * ntiles + tile
  const double *wm_loc;     // [nloc] (W*mu)[gidx[loc]]
  // whole-solve graphs: the PHASE 0 launch that meets the stop rule hands the results to the host itself (block 0),
  // so that the host is released before this launch and the graph's tail have drained
  const SolveArgs *exit_args;
  PinnedFlags *exit_flags;
  long long *dbg;           // MI355_FOLD_DEBUG: wall-clock stamps of workgroup dbg_wg, 8 per launch (tools/fold_stamps.py)
  int dbg_wg;
};
#define MI_FSTAMP(i) \
  do { if (f.dbg && (int)blockIdx.x == f.dbg_wg && threadIdx.x == 0) f.dbg[dbg_row * 8 + (i)] = wall_clock64(); } while (0)
__device__ __forceinline__ double slot_sum(const double *slots, int g, int W) {
  double s = 0.0;
  if (W == 4) {
    const double4 q = *reinterpret_cast<const double4 *>(slots + 4ll * g);
    s += q.x; s += q.y; s += q.z; s += q.w;
  } else if (W == 2) {
    const double2 q = *reinterpret_cast<const double2 *>(slots + 2ll * g);
    s += q.x; s += q.y;
  } else {
    for (int j = 0; j < W; ++j) s += slots[(long long)g * W + j];
  }
  return s;
}

// FOLD_CPT = columns per thread staged in registers: the launch needs max n_Γd <= FOLD_CPT * 256 (<= GEMV_PANEL)
// FOLD_CPT = columns per thread staged in registers: the launch needs max n_Γd <= FOLD_CPT * 64 * WAVES (<= GEMV_PANEL)
// XCHG = false compiles every peer-exchange branch out: the single-GPU launches are the kernel they were before.
template <int RPW, int PHASE, int FOLD_CPT, int WAVES, bool XCHG>
__global__ __launch_bounds__(64 * WAVES) void k_gemv_pcg(DenseMeta m, PcgFold f) {
  constexpr int NTH = 64 * WAVES, NR = WAVES * RPW;  // threads and rows per workgroup
  SolverState *st = f.st;
  const bool x_push = XCHG && f.xp != nullptr, x_inwait = XCHG && f.x_inwait != 0;
  // The tile record and the state block are requested together, before the stop flag is looked at (one memory round trip
  // instead of two at the top of every launch; the empty asm takes them all as inputs — without it the compiler sinks
  // every load but `done` below the early exit).
  const GemvTile t = m.tiles[blockIdx.x];
  const int done0 = st->done;
  const long long it0 = st->it, it_nxt0 = st->it_nxt, maxit = st->maxit, cap = st->res_cap;
  const double tol = st->tol, rTz0 = st->rTz, old = st->rTz_prev;
  const unsigned long long xo = XCHG && f.xst ? (x_inwait ? f.xst->xep[PHASE] : f.xst->epoch) : 0ull;   // (same round trip as the state block)
  const unsigned long long xe = x_inwait ? xo : (f.in_epoch ? *f.in_epoch : 0ull);
  asm volatile("" ::"s"(t.mat_off), "s"(t.n), "s"(t.ld), "s"(t.loc_off), "s"(t.row0), "s"(t.active), "s"(t.nrows), "s"(it0),
               "s"(it_nxt0), "s"(maxit), "s"(cap), "s"(tol), "s"(rTz0), "s"(old), "s"(done0), "s"(xe), "s"(xo));
  if (done0) return;
  GemvRows<RPW> rows;
  if (x_inwait && f.in_stride) {
    // the tables this launch reads are complete when every rank's flag in the own arena has reached the exchange number
    if (t.active) rows.begin(m, t);      // (the matrix does not depend on them: its first loads travel while the flags are polled)
    if (threadIdx.x < 64) xchg_wait(f.xst, *f.xpw, xo);
    __syncthreads();
    // (the tables sit in fine-grained memory — no level of cache keeps their lines across the peers' stores; the vector
    // caches were emptied when this launch began and have not seen the tables since)
    if (__hip_atomic_load(&f.xst->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // expired: the solve fails with MI_ERR_COMM
  }
  const bool streaming = x_inwait && f.in_stride;
  const long long xoff = (long long)(xe & 1ull) * f.in_stride;
  const double *con_in = f.con_in + xoff, *part_in0 = f.part_in0 + xoff, *part_in1 = PHASE == 0 ? f.part_in1 + xoff : nullptr;
  __shared__ __attribute__((aligned(16))) double xs[GEMV_PANEL];
  __shared__ double sm[2 * (NTH / 64)];
  __shared__ double rowv[NR], rowc0[NR], rowc1[NR];
  __shared__ double rowy[NR];   // deflation: the rows' z-contributions
  __shared__ int rowg[NR];      //            and their Γ indices
  const int off = t.loc_off, W = f.W, n = t.n;
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  if (threadIdx.x < NR) { rowc0[threadIdx.x] = 0.0; rowv[threadIdx.x] = 0.0; rowy[threadIdx.x] = 0.0; rowg[threadIdx.x] = 0; }  // visible after the barrier of the sums
#if !MI355_OPERAND_FIRST
  if (t.active && !streaming) rows.begin(m, t);  // matrix stream in flight from here on
#endif

  // ---- every load of the prologue is issued before the first barrier (one memory round trip), all contiguous
  const bool first1 = PHASE == 1 && it_nxt0 == 0;  // very first launch of a solve: r_0 comes from the Γ-ordered vector, p = 0
  const int dbg_row = (int)((PHASE == 1 ? 2 * it_nxt0 : 2 * it0 + 1) & 63);
  MI_FSTAMP(0);
  double pa = 0.0, pb = 0.0;
  for (int i0 = threadIdx.x; i0 < f.n_in; i0 += 8 * NTH) {  // up to eight partials in flight per thread, added in order
    double ta[8], tb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * NTH;
      ta[k] = i < f.n_in ? part_in0[i] : 0.0;
      tb[k] = PHASE == 0 && i < f.n_in ? part_in1[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { pa += ta[k]; if (PHASE == 0) pb += tb[k]; }
  }
  double cv[FOLD_CPT], cs[FOLD_CPT], cc[FOLD_CPT];  // vector value, Γ-sum of the contributions, cnt (PHASE 0: W*mu) — per column
#pragma unroll
  for (int q = 0; q < FOLD_CPT; ++q) {
    const int j = q * NTH + threadIdx.x;
    cv[q] = cs[q] = 0.0; cc[q] = PHASE == 1 ? 1.0 : 0.0;
    if (j < n) {
      const int loc = off + j;
      cs[q] = slot_sum(con_in, loc, W);
      if (PHASE == 1) { cv[q] = first1 ? f.r_gamma[m.gidx[loc]] : f.r_cur[loc]; cc[q] = m.cnt[loc]; }
      else { cv[q] = f.p_cur[loc]; if (f.nvec > 0) cc[q] = f.wm_loc[loc]; }
    }
  }
  // The thread whose column j is also a row of this tile serves that row (at most one column per thread:
  // the rows of a tile are consecutive and fewer than the threads). Its owner duties need p/r of the node, x[g] and the peer list: load now.
  int o_q = -1, o_g = 0;
  bool o_own = false;
  double o_a = 0.0, o_x = 0.0;  // PHASE 1: p (next copy), x[g]    PHASE 0: r (next copy)
  int o_peer[4] = {-1, -1, -1, -1};
#pragma unroll
  for (int q = 0; q < FOLD_CPT; ++q) {
    const int j = q * NTH + threadIdx.x, ri = j - t.row0;
    if (j < n && ri >= 0 && ri < t.nrows) {
      const int loc = off + j;
      o_q = q;
      // everything an owner needs is requested unconditionally (a load behind `if (owner)` would wait for jrank first:
      // a second memory round trip on the way to the operand barrier); x[g] — a dependent load — is consumed only in the
      // epilogue, after the matrix stream
      o_own = f.jrank[loc] == 0;
      if (PHASE == 1) { o_a = first1 ? 0.0 : f.p_nxt[loc]; o_g = m.gidx[loc]; }
      else o_a = f.r_nxt[loc];
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < W) o_peer[k] = f.peer[loc * W + k];
    }
  }
  if (PHASE == 1 && o_q >= 0 && o_own) o_x = f.x[o_g];
  // lane 0 of every wave stores the results of its RPW rows into the contribution rows of all sharing subdomains
  int e_tgt[RPW][4];
  double e_cnt[RPW];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int r = t.row0 + (int)(threadIdx.x >> 6) * RPW + k;
    e_cnt[k] = 1.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) e_tgt[k][q] = -1;
    if ((threadIdx.x & 63) == 0 && r < n) {
      const int loc = off + r;
      if (PHASE == 1) e_cnt[k] = m.cnt[loc];
#pragma unroll
      for (int q = 0; q < 4; ++q) if (q < W) e_tgt[k][q] = f.tgt[loc * W + q];
    }
  }

  // peer stores of the results: thread i < 4*NR serves slot (i & 3) of row (i >> 2) of this tile, in every arena
  int x_tgt = -1;
  if (x_push && t.active && threadIdx.x < 4 * NR) {
    const int r = t.row0 + (int)(threadIdx.x >> 2), k = threadIdx.x & 3;
    if (r < n && k < W) x_tgt = f.tgt[(off + r) * W + k];
  }
#if MI355_OPERAND_FIRST
  if (t.active && !streaming) rows.begin(m, t);  // matrix stream in flight from here on: issued AFTER the prologue's loads (results return in issue order)
#endif
  MI_FSTAMP(1);   // all prologue loads and the first matrix group issued
  // ---- scalars
  double coef;  // alpha (PHASE 1) or beta (PHASE 0)
  if (PHASE == 1) {
    const bool first = first1;
    block_sum2_t<NTH>(pa, pb, sm);
    const double d = pa;
    coef = first ? 0.0 : rTz0 / d;
    if (lead) {
      st->d = d; st->alpha = coef;
      st->rTz_prev = first ? 1.0 : rTz0;
      st->it = it_nxt0;
      if (x_inwait) { f.xst->xep[0] = xo + (x_push ? 1 : 0); f.xst->epoch = xo; }
    }
  } else {
    const long long it_new = it0 + 1;
    block_sum2_t<NTH>(pa, pb, sm);
    const double rr = pa, rz = pb;
    const double res = sqrt(rr);
    const bool stop = !((it_new < maxit) && (res > tol));
    coef = 1. / old;
    coef *= rz;
    if (lead) {
      st->rTr = rr; st->rTz = rz; st->beta = coef;
      st->it_nxt = it_new;
      if (it_new <= cap) f.res_norm[it_new - 1] = res; else st->overflow = 1;
      if (stop || it_new > cap) st->done = 1;
      if (x_inwait) { f.xst->xep[1] = xo + (x_push && !(stop || it_new > cap) ? 1 : 0); f.xst->epoch = xo; }
    }
    if (stop || it_new > cap) {        // same decision in every workgroup (it_new > cap: the reference's BoundsError)
      if (f.exit_args && blockIdx.x == 0) {
        // x is final (the ΠS launch before this one stored it): results to the caller, then ONE store of the solve number
        const SolveArgs q = *f.exit_args;
        for (long long i = threadIdx.x; i < f.n_gamma; i += NTH) q.x_out[i] = f.x[i];
        if (q.res_stage) {
          const long long mres = it_new < q.ncap ? it_new : q.ncap;
          for (long long i = threadIdx.x; i < mres; i += NTH) q.res_stage[i] = i == it_new - 1 ? res : f.res_norm[i];
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
          PinnedFlags *fl = f.exit_flags;
          fl->it = it_new; fl->done = 1; fl->overflow = it_new > cap; fl->respec = 0; fl->x0z = st->x0_zero;
          fl->t_exit = wall_clock64();
          __threadfence_system();
          __hip_atomic_store(&fl->seq, q.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      return;
    }
  }

  MI_FSTAMP(2);   // partial sums reduced: alpha / beta known
  // ---- operand of this GEMV into LDS; rows of this tile: value for the epilogue dot, owners' stores
#pragma unroll
  for (int q = 0; q < FOLD_CPT; ++q) {
    const int j = q * NTH + threadIdx.x;
    if (j < t.ld) {
      double v = 0.0, vs = 0.0;
      if (j < n) {
        // (the first launch of a solve takes r_0 as it is: the contribution rows still hold the previous solve's
        // values, and 0 * NaN left there by a solve that ended non-finite would poison every later solve)
        v = PHASE == 1 ? (first1 ? cv[q] : cv[q] + (-coef) * cs[q])   // r - alpha*Ap
                       : coef * cv[q] + cs[q];                        // beta*p + z
        if (PHASE == 0 && f.nvec > 0) v = v - cc[q];                  // ... - W*mu (defcg.jl:303)
        vs = PHASE == 1 ? v / cc[q] : v;
      }
      xs[j] = vs;
      if (q == o_q) {
        const int ri = j - t.row0;
        if (t.active) rowv[ri] = v;
        if (PHASE == 1 && f.nvec > 0) rowg[ri] = o_g;
        if (o_own) {                                      // owner of this Γ node
          if (PHASE == 1) {
            if (t.active) rowc0[ri] = v * v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int pl = o_peer[k];
              if (pl >= 0) { f.r_nxt[pl] = v; f.p_cur[pl] = o_a; }
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int pl = o_peer[k];
              if (pl >= 0) { f.p_nxt[pl] = v; f.r_cur[pl] = o_a; }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  // Multi-GPU, sharded operator: a tile whose block lives on another rank has done its share of the vector work (scalars,
  // owner stores of p / r, x) and stops here; its contribution rows and partial dots come from the owning rank through the
  // exchange that follows the launch (its own entries stay zero).
  if (!t.active) {
    if (PHASE == 1 && o_q >= 0 && o_own) f.x[o_g] = o_x + coef * o_a;  // x + alpha*p (cg.jl:97)
    return;
  }
  MI_FSTAMP(3);   // operand staged
  rows.panel(xs, 0, t.ld);
  MI_FSTAMP(4);   // stream consumed
  double sum[RPW];
  rows.finish(sum);
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int ri = w * RPW + k, r = t.row0 + ri;
    if (rows.lane == 0) {
      double y = 0.0;
      if (r < n) {
        y = PHASE == 1 ? sum[k] / e_cnt[k] : sum[k];
        if (!x_push) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int tg = e_tgt[k][q];
            if (tg >= 0) f.con_out[tg] = y;
          }
        }
      }
      rowc1[ri] = (r < n ? rowv[ri] : 0.0) * y;    // r_g * z-contribution  /  p_g * Ap-contribution
      if (PHASE == 1 || x_push) rowy[ri] = y;
    }
  }
  if (PHASE == 1 && o_q >= 0 && o_own) f.x[o_g] = o_x + coef * o_a;  // x + alpha*p (cg.jl:97), off the critical path
  __syncthreads();
  MI_FSTAMP(5);   // results scattered
  const long long xpo = (long long)((xo + 1) & 1ull) * f.out_stride;
  if (x_push && x_tgt >= 0) {
    const XchgPeers &P = *f.xp;
    const double y = rowy[threadIdx.x >> 2];
    const char *own = P.arena[P.rank];
    for (int q = 0; q < P.n; ++q)
      xchg_store(reinterpret_cast<double *>(reinterpret_cast<char *>(f.con_out) + (P.arena[q] - own)) + xpo + x_tgt, y);
  }
  if (PHASE == 1 && f.nvec > 0 && (int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + f.nvec) {
    // per-tile partial of WtA*z (defcg.jl:301): sum over this tile's rows of WtA[v, g(row)] * (z-contribution of the row);
    // the second wave does it while the first one reduces the dot products (nvec <= 64)
    const int v = (int)threadIdx.x - 64;
    const double *aw = f.AW + (long long)v * f.n_gamma;
    double s = 0.0;
    for (int i0 = 0; i0 < NR; i0 += 8) {
      double a[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = i0 + k < NR ? aw[rowg[i0 + k]] : 0.0;   // rows beyond the tile: rowy = 0, rowg = 0
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k < NR) s += a[k] * rowy[i0 + k];
    }
    f.part_mu[(long long)v * gridDim.x + blockIdx.x] = s;
  }
  if (threadIdx.x < 64) {  // per-tile partials of the next dot products: one shuffle tree over the NR row terms
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < NR; i += 64) { a += rowc1[i]; b += rowc0[i]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
      const int ps = t.active - 1;   // the tile number on one GPU; a tiling-independent slot when the launch is sharded over ranks
      if (!x_push) {
        if (PHASE == 1) { f.part_out1[ps] = a; f.part_out0[ps] = b; }
        else f.part_out0[ps] = a;
      } else {
        const XchgPeers &P = *f.xp;
        const char *own = P.arena[P.rank];
        for (int q = 0; q < P.n; ++q) {
          const long long d = P.arena[q] - own;
          xchg_store(reinterpret_cast<double *>(reinterpret_cast<char *>(f.part_out0) + d) + xpo + ps, PHASE == 1 ? b : a);
          if (PHASE == 1) xchg_store(reinterpret_cast<double *>(reinterpret_cast<char *>(f.part_out1) + d) + xpo + ps, a);
        }
      }
    }
  }
  if (x_push) {
    // publish. The results went out as system-scope write-through stores (xchg_store): nothing of them stays in an L2, so
    // no cache write-back is needed — a system-scope fence per tile walks the L2 and cost 8 us per launch with ~290 tiles
    // (profiles/NOTES.md). Every thread waits for the acknowledgement of its own stores, the barrier collects the
    // workgroup, and the last tile to count itself in stores the flags: all stores of the launch are complete by then.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int prev = __hip_atomic_fetch_add(&f.xst->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == f.n_arrive - 1) {
        __hip_atomic_store(&f.xst->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const XchgPeers &P = *f.xp;
        for (int q = 0; q < P.n; ++q) __hip_atomic_store(xchg_flag(P, q, P.rank), xo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  MI_FSTAMP(6);
}

// y[i] = sum of the local contributions to Γ node i, in ascending subdomain order
// (`Sx[lΓ] += Sdxd[lΓd]` for idom = 1..ndom, EPDD.jl:779-781 / 1379-1381); indexed form, used by the
// matrix-free operator whose local results are produced in local order by the SpMV kernel.
__global__ __launch_bounds__(NT) void k_assemble(int n, const int *__restrict__ aptr, const int *__restrict__ apos,
                                                 const double *__restrict__ yloc, double *__restrict__ y,
                                                 const int *done) {
  if (done && *done) return;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    double s = 0.0;
    for (int k = aptr[i]; k < aptr[i + 1]; ++k) s += yloc[apos[k]];
    y[i] = s;
  }
}
// The same sum over the contiguous contribution slots written by k_gemv_batched.
__global__ __launch_bounds__(NT) void k_assemble_slots(int n, int width, const double *__restrict__ yslots,
                                                       double *__restrict__ y, const int *done) {
  if (done && *done) return;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    double s = 0.0;
    for (int j = 0; j < width; ++j) s += yslots[(long long)i * width + j];
    y[i] = s;
  }
}
// xcat[slot] = x[gidx[slot]] for every local slot (matrix-free path: xd[lΓd] = x[lΓ], EPDD.jl:728-730)
__global__ __launch_bounds__(NT) void k_gather(int nloc, const int *__restrict__ gidx, const double *__restrict__ x,
                                               double *__restrict__ xcat) {
  for (int i = blockIdx.x * NT + threadIdx.x; i < nloc; i += gridDim.x * NT) xcat[i] = x[gidx[i]];
}

// ------------------------------------------------------------------ Krylov loop kernels
// Set-up:  r = b - Ap; partial r'r and b'b.
__global__ __launch_bounds__(NT) void k_residual(int n, const double *__restrict__ b, const double *__restrict__ Ap,
                                                 double *__restrict__ r, double *__restrict__ part_rr,
                                                 double *__restrict__ part_bb) {
  __shared__ double sm[NT / 64 + 1];
  double srr = 0.0, sbb = 0.0;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    const double bi = b[i];
    const double ri = bi - Ap[i];
    r[i] = ri;
    srr += ri * ri;
    sbb += bi * bi;
  }
  srr = block_sum(srr, sm);
  sbb = block_sum(sbb, sm);
  if (threadIdx.x == 0) {
    part_rr[blockIdx.x] = srr;
    if (part_bb) part_bb[blockIdx.x] = sbb;
  }
}
// Start-up of the folded Schur loop when n_Γ exceeds the single-workgroup kernels (k_fused_residual<., true>): after
// k_residual, one workgroup sums its partials and leaves the scalars exactly as that kernel does.
__global__ __launch_bounds__(NT) void k_fold_start(SolverState *st, const double *part_rr, const double *part_bb, int g) {
  __shared__ double sm[NT / 64 + 1];
  const double eps = st->eps;
  const double rr = sum_partials(part_rr, g, sm);
  const double bb = sum_partials(part_bb, g, sm);
  if (threadIdx.x == 0) {
    st->rTr = rr; st->bnorm = sqrt(bb);
    st->tol = eps * st->bnorm;
    st->it = 0; st->it_nxt = 0; st->done = 0; st->overflow = 0;
    st->rTz = 0.0; st->rTz_prev = 1.0; st->rTr_prev = rr;
    st->d = 0.0; st->alpha = 0.0; st->beta = 0.0;
  }
}
// Set-up: it = 1; res_norm[1] = sqrt(r'r); tol = eps*norm2(b) (cg.jl:26-32 / 81-89). eps, maxit and
// res_cap were written into the state block by the host.
__global__ __launch_bounds__(NT) void k_init_state(SolverState *st, const double *part_rr, const double *part_bb,
                                                   const double *part_rz, int g, double *res_norm) {
  __shared__ double sm[NT / 64 + 1];
  const double eps = st->eps;
  const long long maxit = st->maxit, res_cap = st->res_cap;
  const double rr = sum_partials(part_rr, g, sm);
  const double bb = sum_partials(part_bb, g, sm);
  const double rz = part_rz ? sum_partials(part_rz, g, sm) : rr;
  if (threadIdx.x == 0) {
    st->rTr = rr; st->rTz = rz; st->rTr_prev = rr; st->rTz_prev = rz;
    st->bnorm = sqrt(bb);
    st->tol = eps * st->bnorm;
    st->it = 1;
    st->d = 0.0; st->alpha = 0.0; st->beta = 0.0;
    const double res = sqrt(rr);
    st->overflow = 0;
    if (res_cap >= 1) res_norm[0] = res; else st->overflow = 1;
    st->done = !((1 < maxit) && (res > st->tol));
  }
}
// Loop step 1:  d = p'Ap (from partials); alpha = num/d; x += alpha p; r -= alpha Ap; partial r'r.
// num = r'z (pcg, cg.jl:95) or r'r (cg, cg.jl:38). With a diagonal preconditioner (diag 1: identity,
// 2: Jacobi) `z .= M \ r` and the partial r'z (cg.jl:100-101) are produced in the same pass.
__global__ __launch_bounds__(NT) void k_update_xr(int n, SolverState *st, const double *part_pAp, int g_in,
                                                  const double *__restrict__ p, const double *__restrict__ Ap,
                                                  double *__restrict__ x, double *__restrict__ r,
                                                  double *__restrict__ part_rr, int precond, int diag,
                                                  const double *__restrict__ dinv, double *__restrict__ z,
                                                  double *__restrict__ part_rz) {
  if (st->done) return;
  __shared__ double sm[NT / 64 + 1];
  // The element loads do not depend on alpha: the first four elements of every thread (all of them unless the grid is
  // capped) are requested BEFORE the partial sums are reduced, so the kernel pays one memory round trip, not two.
  const int stride = gridDim.x * NT;
  int i0 = blockIdx.x * NT + threadIdx.x;
  double pv[4], xv[4], rv[4], av[4], dv[4];
  auto load = [&](int base) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + k * stride;
      const bool ok = i < n;
      pv[k] = ok ? p[i] : 0.0; xv[k] = ok ? x[i] : 0.0; rv[k] = ok ? r[i] : 0.0; av[k] = ok ? Ap[i] : 0.0;
      dv[k] = ok && diag == 2 ? dinv[i] : 1.0;
    }
  };
  load(i0);
  const double d = sum_partials(part_pAp, g_in, sm);
  const double num = precond ? st->rTz : st->rTr;
  const double alpha = num / d;
  double srr = 0.0, srz = 0.0;
  for (;;) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * stride;
      if (i < n) {
        x[i] = xv[k] + alpha * pv[k];              // axpy!(alpha, p, x)
        const double ri = rv[k] + (-alpha) * av[k];  // axpy!(-alpha, Ap, r)
        r[i] = ri;
        srr += ri * ri;
        if (diag) {
          const double zi = diag == 2 ? dv[k] * ri : ri;
          z[i] = zi;
          srz += ri * zi;
        }
      }
    }
    i0 += 4 * stride;
    if (i0 >= n) break;
    load(i0);
  }
  srr = block_sum(srr, sm);
  if (diag) srz = block_sum(srz, sm);
  if (threadIdx.x == 0) {
    part_rr[blockIdx.x] = srr;
    if (diag) part_rz[blockIdx.x] = srz;
    if (blockIdx.x == 0) {
      st->d = d; st->alpha = alpha;
      st->rTr_prev = st->rTr; st->rTz_prev = st->rTz;
    }
  }
}
// Loop step 2:  r'r, r'z from partials; beta = (1/old)*new (cg.jl:39,44 / 96,102);
// p = beta p + z [- W mu] (cg.jl:45 / 103; defcg.jl:78 / 303); it += 1; res_norm[it] = sqrt(r'r); stop rule.
__global__ __launch_bounds__(NT) void k_update_p(int n, SolverState *st, const double *part_rr,
                                                 const double *part_rz, int g_in, const double *__restrict__ z,
                                                 double *__restrict__ p, const double *__restrict__ W,
                                                 const double *__restrict__ mu, int nvec, double *res_norm,
                                                 int precond) {
  __shared__ double sm[NT / 64 + 1];
  __shared__ int was_done;  // thread 0 of workgroup 0 sets st->done below: take one uniform snapshot
  if (threadIdx.x == 0) was_done = st->done;
  __syncthreads();
  if (was_done) return;
  const int stride = gridDim.x * NT;
  int i0 = blockIdx.x * NT + threadIdx.x;
  double pv[4], zv[4];
  auto load = [&](int base) {  // independent of beta: requested before the reductions (see k_update_xr)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + k * stride;
      pv[k] = i < n ? p[i] : 0.0;
      zv[k] = i < n ? z[i] : 0.0;
    }
  };
  load(i0);
  const double rr = sum_partials(part_rr, g_in, sm);
  const double rz = precond ? sum_partials(part_rz, g_in, sm) : rr;
  const double old = precond ? st->rTz_prev : st->rTr_prev;
  double beta = 1. / old;
  beta *= rz;
  for (;;) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * stride;
      if (i < n) {
        double v = beta * pv[k] + zv[k];     // axpby!(1, z, beta, p)
        if (nvec > 0) v = v - w_times_mu(W, n, i, mu, nvec);   // (W*mu)[i], column-axpy order
        p[i] = v;
      }
    }
    i0 += 4 * stride;
    if (i0 >= n) break;
    load(i0);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->rTr = rr; st->rTz = rz; st->beta = beta;
    const long long it = st->it + 1;
    st->it = it;
    const double res = sqrt(rr);
    const bool over = it > st->res_cap;  // the reference's `res_norm[it] = ...` throws BoundsError here: stop
    if (!over) res_norm[it - 1] = res; else st->overflow = 1;
    st->done = over || !((it < st->maxit) && (res > st->tol));
  }
}
// Entry and exit of a solve, one launch each instead of a handful of small copies (each a separate blit on the stream):
// in:  b, x0 -> workspace; eps / maxit / res_cap -> state block.
__global__ __launch_bounds__(NT) void k_solve_begin(int n, const double *__restrict__ b_in, const double *__restrict__ x_in,
                                                    double *__restrict__ b, double *__restrict__ x, SolverState *st, double eps,
                                                    long long maxit, long long res_cap) {
  int nz = 0;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    const double xi = x_in[i];
    b[i] = b_in[i]; x[i] = xi;
    nz |= xi != 0.0;   // NaN counts as non-zero
  }
  // x0_zero was left at 1 by the previous solve's k_solve_end (0 after allocation): it survives only if every entry is 0
  if (__syncthreads_or(nz) && threadIdx.x == 0) st->x0_zero = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st->eps = eps; st->maxit = maxit; st->res_cap = res_cap; st->done = 0; }
}
// The same entry / exit pair as graph nodes: arguments through the pinned block instead of kernel parameters, and the
// exit kernel ends with a release of everything it wrote followed by ONE store of the solve number, which is what the
// host waits for (no stream synchronisation on the critical path of a solve).
__device__ __forceinline__ unsigned long long sys_load_u64(const void *p) {
  return __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(NT) void k_solve_begin_g(int n, const SolveArgs *pin, double *__restrict__ b, double *__restrict__ x,
                                                      SolverState *st, SolveArgs *dev, PinnedFlags *flags) {
  __shared__ unsigned long long a[sizeof(SolveArgs) / 8];
  if (blockIdx.x == 0 && threadIdx.x == 0) flags->t_entry = wall_clock64();
  static_assert(sizeof(SolveArgs) % 8 == 0 && sizeof(SolveArgs) / 8 <= 64, "SolveArgs is a block of 8-byte fields");
  if (threadIdx.x < sizeof(SolveArgs) / 8) a[threadIdx.x] = sys_load_u64((const unsigned long long *)pin + threadIdx.x);
  __syncthreads();
  const SolveArgs &q = *reinterpret_cast<const SolveArgs *>(a);
  const double *b_in = q.b_in, *x_in = q.x_in;
  int nz = 0;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    const double xi = x_in[i];
    b[i] = b_in[i]; x[i] = xi;
    nz |= xi != 0.0;
  }
  if (__syncthreads_or(nz) && threadIdx.x == 0) st->x0_zero = 0;
  if (blockIdx.x == 0) {
    if (threadIdx.x < sizeof(SolveArgs) / 8) ((unsigned long long *)dev)[threadIdx.x] = a[threadIdx.x];
    if (threadIdx.x == 0) { st->eps = q.eps; st->maxit = q.maxit; st->res_cap = q.res_cap; st->done = 0; }
  }
}
__global__ __launch_bounds__(NT) void k_solve_end_g(int n, SolverState *st, const SolveArgs *dev, int fold, const double *__restrict__ x,
                                                    const double *__restrict__ res_norm, PinnedFlags *flags, int *x0_zero,
                                                    unsigned *end_count) {
  __shared__ int last;
  const long long it = fold ? st->it_nxt : st->it;
  const SolveArgs q = *dev;
  // a loop launch may have handed the results over already (k_gemv_pcg's stop branch): same answer in every workgroup,
  // since that launch precedes this one on the stream and this kernel stores the number only after all of them counted in
  if (sys_load_u64(&flags->seq) == q.seq) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *x0_zero = 1;   // "assume zero" for the next solve's entry kernel to refute
    return;
  }
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) q.x_out[i] = x[i];
  if (q.res_stage) {
    const long long m = it < q.ncap ? it : q.ncap;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < m; i += (long long)gridDim.x * NT) q.res_stage[i] = res_norm[i];
  }
  __threadfence_system();   // this thread's stores are out (device memory written back, pinned memory delivered) ...
  __syncthreads();          // ... for the whole workgroup before it counts itself as finished
  if (threadIdx.x == 0) last = atomicAdd(end_count, 1u) == gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x == 0) {
    *end_count = 0;
    flags->it = it; flags->done = st->done; flags->overflow = st->overflow; flags->respec = 0; flags->x0z = *x0_zero;
    flags->t_exit = wall_clock64();
    *x0_zero = 1;
    __threadfence_system();
    __hip_atomic_store(&flags->seq, q.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// out: x -> caller's vector (device), `it` / done / overflow and the first min(it, ncap) residual norms -> pinned host memory.
__global__ __launch_bounds__(NT) void k_solve_end(int n, const SolverState *st, int fold, const double *__restrict__ x,
                                                  double *__restrict__ x_out, const double *__restrict__ res_norm,
                                                  double *__restrict__ res_stage, long long ncap, PinnedFlags *flags,
                                                  int *x0_zero) {
  const long long it = fold ? st->it_nxt : st->it;
  if (blockIdx.x == 0 && threadIdx.x == 0) *x0_zero = 1;  // "assume zero" for the next solve's k_solve_begin to refute
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) x_out[i] = x[i];
  if (res_stage) {
    const long long m = it < ncap ? it : ncap;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < m; i += (long long)gridDim.x * NT) res_stage[i] = res_norm[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { flags->it = it; flags->done = st->done; flags->overflow = st->overflow; }
}

// ------------------------------------------------------------------ fused single-workgroup kernels (n <= 8192)
// On the Schur system the Γ-vectors are tiny (n_Γ ≈ 4 k): one 1024-thread workgroup keeps its
// share of every vector in registers and does Γ-sum + dot + update in one launch, so a PCG
// iteration is 4 launches (S GEMV, this, NN GEMV, this) instead of 8 latency-bound ones.
// A vector operand is a "view": width == 0: plain vector src[i]; width == W: the Γ-sum
// y[i] = Σ_{j<W} src[i*W + j] over the contribution slots of k_gemv_batched (deferred scatter-add).
// All loads of a kernel are issued up front (one memory round trip), scalars included.
constexpr int NTF = 1024;
constexpr int FUSED_MAX_N = NTF * 8;

struct AsmView {
  const double *src;
  int width;
};
__device__ __forceinline__ double view_load(const AsmView &v, int e) {
  if (v.width == 0) return v.src[e];
  double s = 0.0;
  if (v.width == 4) {
    const double4 q = *reinterpret_cast<const double4 *>(v.src + 4ll * e);
    s += q.x; s += q.y; s += q.z; s += q.w;
  } else if (v.width == 2) {
    const double2 q = *reinterpret_cast<const double2 *>(v.src + 2ll * e);
    s += q.x; s += q.y;
  } else {
    for (int j = 0; j < v.width; ++j) s += v.src[(long long)e * v.width + j];
  }
  return s;
}
// Deterministic sum over the 1024-thread workgroup, broadcast. `sm` has NTF/64 + 1 doubles.
__device__ __forceinline__ double block_sum_f(double v, double *sm) { return block_sum_t<NTF>(v, sm); }
// Ap = view; d = p'Ap; alpha = num/d; x += alpha p; r -= alpha Ap; r'r   (cg.jl:36-43 / 93-99)
template <int EPT>
__global__ __launch_bounds__(NTF) void k_fused_xr(int n, SolverState *st, AsmView vAp, const double *__restrict__ p,
                                                  double *__restrict__ x, double *__restrict__ r, int precond) {
  __shared__ double sm[NTF / 64 + 1];
  const int done0 = st->done;   // tested after the operands have been requested: one memory round trip, not two
  const double rTr0 = st->rTr, rTz0 = st->rTz;
  double pe[EPT], ae[EPT], xe[EPT], re[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    pe[k] = ae[k] = xe[k] = re[k] = 0.0;
    if (e < n) { ae[k] = view_load(vAp, e); pe[k] = p[e]; xe[k] = x[e]; re[k] = r[e]; }
  }
  asm volatile("" ::"s"(done0), "v"(pe[0]), "v"(ae[0]), "v"(xe[0]), "v"(re[0]));
  if (done0) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) s += pe[k] * ae[k];
  const double d = block_sum_f(s, sm);
  const double alpha = (precond ? rTz0 : rTr0) / d;
  double srr = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    if (e < n) {
      x[e] = xe[k] + alpha * pe[k];
      const double ri = re[k] + (-alpha) * ae[k];
      r[e] = ri;
      srr += ri * ri;
    }
  }
  srr = block_sum_f(srr, sm);
  if (threadIdx.x == 0) {
    st->d = d; st->alpha = alpha;
    st->rTr_prev = rTr0; st->rTz_prev = rTz0;
    st->rTr = srr;
  }
}
// mu = WtAW \ rhs by ONE wave, in registers: lane i holds b[i]; row swaps, unit-lower forward and upper backward
// substitution in the order of LAPACK getrs (and of k_lu_solve), values exchanged with wave shuffles. nvec <= 64.
__device__ __forceinline__ double wave_lu_solve(int nvec, const double *__restrict__ LU, const int *__restrict__ piv,
                                                double b) {
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < nvec; ++k) {
    const int pk = piv[k];
    const double vk = __shfl(b, k, 64), vp = __shfl(b, pk, 64);
    if (pk != k) { if (lane == k) b = vp; else if (lane == pk) b = vk; }
  }
  for (int k = 0; k < nvec; ++k) {
    const double bk = __shfl(b, k, 64);
    if (lane > k && lane < nvec) b -= bk * LU[lane + (long long)k * nvec];
  }
  for (int k = nvec - 1; k >= 0; --k) {
    if (lane == k) b /= LU[k + (long long)k * nvec];
    const double bk = __shfl(b, k, 64);
    if (lane < k) b -= bk * LU[lane + (long long)k * nvec];
  }
  return b;
}
// The same solve with this lane's row of the factors in registers and the exchanged values read with v_readlane
// (uniform lane index): a step of the 3*nvec-long dependency chain costs a few cycles instead of an LDS round trip.
__device__ __forceinline__ double lane_read(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
template <int NV>
__device__ __forceinline__ double wave_lu_solve_reg(int nvec, const double *lu_s, const int *piv_s, double b) {
  const int lane = threadIdx.x & 63;
  double row[NV];                                    // LU[lane, 0:nvec]
#pragma unroll
  for (int k = 0; k < NV; ++k) row[k] = (k < nvec && lane < nvec) ? lu_s[lane + k * nvec] : 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (k < nvec) {
      const int pk = __builtin_amdgcn_readfirstlane(piv_s[k]);
      const double vk = lane_read(b, k), vp = lane_read(b, pk);
      if (pk != k) { if (lane == k) b = vp; else if (lane == pk) b = vk; }
    }
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (k < nvec) {
      const double bk = lane_read(b, k);
      if (lane > k && lane < nvec) b -= bk * row[k];
    }
#pragma unroll
  for (int k = NV - 1; k >= 0; --k)
    if (k < nvec) {
      if (lane == k) b /= row[k];
      const double bk = lane_read(b, k);
      if (lane < k) b -= bk * row[k];
    }
  return b;
}

// W_loc[v * nloc + loc] = W[v * n + gidx[loc]]: the deflation vectors in the local order of the dense blocks, once per solve
// (k_defl_mu then needs no Γ index before it can request its W entries: one memory round trip less per iteration)
__global__ __launch_bounds__(NT) void k_gather_w_loc(int nvec, long long n_gamma, int nloc, const int *__restrict__ gidx,
                                                     const double *__restrict__ W, double *__restrict__ W_loc) {
  const int loc = blockIdx.x * NT + threadIdx.x;
  if (loc >= nloc) return;
  const long long g = gidx[loc];
  const int v = blockIdx.y;
  W_loc[(long long)v * nloc + loc] = W[(long long)v * n_gamma + g];
}

// Folded Def-PCG, between the ΠS and the S launch: mu = WtAW \ (WtA*z) from the per-tile partials (every workgroup
// solves the same nvec x nvec system with the LU factors staged in LDS, nvec <= 64), then
// wm_loc[loc] = (W*mu)[gidx[loc]] in column-axpy order (`W * mu`, defcg.jl:303) for this workgroup's slice of the
// local positions, where the S launch reads it contiguously. 1024 threads: 16 per deflation vector for the sums.
__global__ __launch_bounds__(1024) void k_defl_mu(const SolverState *st, int nvec, int ntiles, const double *__restrict__ part_mu,
                                                  const double *__restrict__ LU, const int *__restrict__ piv,
                                                  const double *__restrict__ W, long long n_gamma, int nloc,
                                                  const int *__restrict__ gidx, double *__restrict__ wm_loc,
                                                  double *__restrict__ mu_out, const double *__restrict__ W_loc) {
  // One memory round trip for everything whose address is known at launch (stop flag, LU factors, pivots, Γ index, the
  // first batch of partials), a second one for the W entries behind the Γ index: the kernel is nothing but latency.
  __shared__ double lu_s[64 * 64];
  __shared__ int piv_s[64];
  __shared__ double mu_s[64];
  const int done0 = st->done;
  double lu_r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = (int)threadIdx.x + 1024 * k; lu_r[k] = i < nvec * nvec ? LU[i] : 0.0; }
  const int piv_r = (int)threadIdx.x < nvec ? piv[threadIdx.x] : 0;
  const int loc = blockIdx.x * 1024 + threadIdx.x;
  const long long g = (loc < nloc && !W_loc) ? gidx[loc] : 0;
  // rhs[v] = sum over tiles: TPV = 1024 / (nvec rounded up to a power of two) threads per vector, every thread's
  // partials requested in one batch, then a TPV-lane shuffle tree (fixed order: deterministic)
  int vpad = 1;
  while (vpad < nvec) vpad <<= 1;
  const int tpv = 1024 / vpad;                          // 16 .. 1024, a power of two
  const int v = threadIdx.x / tpv, l16 = threadIdx.x % tpv;
  double a0[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a0[k] = (v < nvec && l16 + tpv * k < ntiles) ? part_mu[(long long)v * ntiles + l16 + tpv * k] : 0.0;
  // W entries of this thread's local position: independent of mu, in flight during the sums and the solve (from the
  // pre-gathered W_loc they need no Γ index: requested with everything else, ahead of the exit test)
  double w0[20];                                        // the first 20 columns of W at g
#pragma unroll
  for (int u = 0; u < 20; ++u)
    w0[u] = (u < nvec && loc < nloc) ? (W_loc ? W_loc[(long long)u * nloc + loc] : W[(long long)u * n_gamma + g]) : 0.0;
  asm volatile("" ::"s"(done0));
  if (done0) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = (int)threadIdx.x + 1024 * k; if (i < nvec * nvec) lu_s[i] = lu_r[k]; }
  if ((int)threadIdx.x < nvec) piv_s[threadIdx.x] = piv_r;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a0[k];
  if (v < nvec)
    for (int i0 = l16 + tpv * 16; i0 < ntiles; i0 += tpv * 16) {
      double a[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = i0 + tpv * k < ntiles ? part_mu[(long long)v * ntiles + i0 + tpv * k] : 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += a[k];
    }
  if (tpv > 64) {                                       // few vectors: finish across waves through LDS
    s = wave_sum(s);
    __shared__ double wsum[16];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (l16 == 0 && v < nvec) { s = 0.0; for (int k = 0; k < tpv / 64; ++k) s += wsum[v * (tpv / 64) + k]; }
  } else {
    for (int o = tpv / 2; o > 0; o >>= 1) s += __shfl_down(s, o, tpv);
  }
  if (l16 == 0 && v < nvec) mu_s[v] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    const double rv = (int)threadIdx.x < nvec ? mu_s[threadIdx.x] : 0.0;
    const double m = nvec <= 12 ? wave_lu_solve_reg<12>(nvec, lu_s, piv_s, rv)
                   : nvec <= 20 ? wave_lu_solve_reg<20>(nvec, lu_s, piv_s, rv) : wave_lu_solve(nvec, lu_s, piv_s, rv);
    mu_s[threadIdx.x] = m;
    if (blockIdx.x == 0 && (int)threadIdx.x < nvec) mu_out[threadIdx.x] = m;
  }
  __syncthreads();
  if (loc < nloc) {
    double wm = 0.0;
#pragma unroll
    for (int u = 0; u < 20; ++u)
      if (u < nvec) wm += w0[u] * mu_s[u];
    for (int q0 = 20; q0 < nvec; q0 += 16) {      // beyond the preloaded columns: sixteen in flight per thread
      double w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        w[u] = q0 + u < nvec ? (W_loc ? W_loc[(long long)(q0 + u) * nloc + loc] : W[(long long)(q0 + u) * n_gamma + g]) : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (q0 + u < nvec) wm += w[u] * mu_s[q0 + u];
    }
    wm_loc[loc] = wm;
  }
}

// z = view; r'z; beta = (1/old)*new; p = beta p + z [- W mu]; it += 1; res_norm[it]; stop rule
// (cg.jl:44-47 / 100-106; defcg.jl:76-80 / 299-305). r'r was stored by k_fused_xr. Deflation: mu is either
// given, or (LU != nullptr, nvec <= 64) solved here from rhs[v] = WtA[v,:].z by the first wave.
template <int EPT>
__global__ __launch_bounds__(NTF) void k_fused_p(int n, SolverState *st, AsmView vz, const double *__restrict__ r,
                                                 double *__restrict__ p, const double *__restrict__ W,
                                                 const double *__restrict__ mu, int nvec, double *res_norm,
                                                 int precond, const double *__restrict__ LU,
                                                 const int *__restrict__ piv, const double *__restrict__ rhs) {
  if (st->done) return;  // written only by thread 0 at the very end, behind the barriers below
  __shared__ double sm[NTF / 64 + 1];
  __shared__ double mu_s[64];
  __shared__ double lu_s[64 * 64];  // the factors in LDS: the substitution is a chain of 3*nvec dependent steps, and a
  __shared__ int piv_s[64];         // global load inside every step costs a memory round trip each
  const bool wave_lu = nvec > 0 && LU;
  if (wave_lu) {
    for (int i = threadIdx.x; i < nvec * nvec; i += NTF) lu_s[i] = LU[i];
    if ((int)threadIdx.x < nvec) piv_s[threadIdx.x] = piv[threadIdx.x];
  }
  const double rr = st->rTr;
  const double old = precond ? st->rTz_prev : st->rTr_prev;
  const double tol = st->tol;
  const long long it0 = st->it, maxit = st->maxit, cap = st->res_cap;
  double ze[EPT], pe[EPT];
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    ze[k] = pe[k] = 0.0;
    if (e < n) {
      ze[k] = view_load(vz, e);
      pe[k] = p[e];
      if (precond) s += r[e] * ze[k];
    }
  }
  if (wave_lu) {
    const double rv = threadIdx.x < (unsigned)nvec ? rhs[threadIdx.x] : 0.0;
    __syncthreads();  // lu_s, piv_s
    if (threadIdx.x < 64) mu_s[threadIdx.x] = wave_lu_solve(nvec, lu_s, piv_s, rv);
  }
  const double rz = precond ? block_sum_f(s, sm) : rr;
  if (!precond) __syncthreads();  // mu_s visible (block_sum_f has barriers of its own)
  double beta = 1. / old;
  beta *= rz;
  double wm[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) wm[k] = 0.0;
  if (nvec > 0) {  // (W*mu)[e] in column order; the loads of 8 columns x EPT elements go out together (one round trip per 8 columns)
    const double *mup = LU ? mu_s : mu;
    constexpr int QB = EPT >= 8 ? 4 : 8;  // EPT*QB doubles in flight per thread (<= 64 VGPRs)
    for (int q0 = 0; q0 < nvec; q0 += QB) {
      double w[EPT][QB];
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int e = k * NTF + threadIdx.x;
#pragma unroll
        for (int u = 0; u < QB; ++u) w[k][u] = (e < n && q0 + u < nvec) ? W[(long long)(q0 + u) * n + e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < QB; ++u)
        if (q0 + u < nvec) {
          const double m = mup[q0 + u];
#pragma unroll
          for (int k = 0; k < EPT; ++k) wm[k] += w[k][u] * m;
        }
    }
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    if (e < n) {
      double v = beta * pe[k] + ze[k];
      if (nvec > 0) v = v - wm[k];
      p[e] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->rTz = rz; st->beta = beta;
    const long long it = it0 + 1;
    st->it = it;
    const double res = sqrt(rr);
    if (it <= cap) res_norm[it - 1] = res; else st->overflow = 1;
    st->done = it > cap || !((it < maxit) && (res > tol));  // it > cap: BoundsError in the reference, the loop ends there
  }
}
// Unpreconditioned cg (cg.jl:35-47): both halves of the iteration in ONE single-workgroup launch — z is r,
// so nothing but two workgroup reductions separates alpha from beta:
//   d = p'Ap; alpha = r'r/d; x += alpha p; r -= alpha Ap; r'r; beta = (1/old)*new; p = beta p + r; it; stop rule.
template <int EPT>
__global__ __launch_bounds__(NTF) void k_fused_cg(int n, SolverState *st, AsmView vAp, double *__restrict__ p,
                                                  double *__restrict__ x, double *__restrict__ r, double *res_norm) {
  __shared__ double sm[NTF / 64 + 1];
  const int done0 = st->done;   // tested after the operands have been requested: one memory round trip, not two
  const double rTr0 = st->rTr, tol = st->tol;
  const long long it0 = st->it, maxit = st->maxit, cap = st->res_cap;
  double pe[EPT], ae[EPT], xe[EPT], re[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    pe[k] = ae[k] = xe[k] = re[k] = 0.0;
    if (e < n) { ae[k] = view_load(vAp, e); pe[k] = p[e]; xe[k] = x[e]; re[k] = r[e]; }
  }
  asm volatile("" ::"s"(done0), "v"(pe[0]), "v"(ae[0]), "v"(xe[0]), "v"(re[0]));
  if (done0) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) s += pe[k] * ae[k];
  const double d = block_sum_f(s, sm);
  const double alpha = rTr0 / d;
  double srr = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    if (e < n) {
      x[e] = xe[k] + alpha * pe[k];
      re[k] = re[k] + (-alpha) * ae[k];
      r[e] = re[k];
      srr += re[k] * re[k];
    }
  }
  const double rr = block_sum_f(srr, sm);
  double beta = 1. / rTr0;
  beta *= rr;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    if (e < n) p[e] = beta * pe[k] + re[k];
  }
  if (threadIdx.x == 0) {
    st->d = d; st->alpha = alpha; st->beta = beta;
    st->rTr_prev = rTr0; st->rTz_prev = rTr0;
    st->rTr = rr; st->rTz = rr;
    const long long it = it0 + 1;
    st->it = it;
    const double res = sqrt(rr);
    if (it <= cap) res_norm[it - 1] = res; else st->overflow = 1;
    st->done = it > cap || !((it < maxit) && (res > tol));  // it > cap: BoundsError in the reference, the loop ends there
  }
}

// Set-up, first half: r = b - A*x (Ap as a view); r'r and b'b   (cg.jl:28-29,32 / 83-84,88).
// FOLD: also the scalars the folded PCG launches (k_gemv_pcg) expect at start-up: tol, it_nxt = 0,
// rTz_prev = 1, flags cleared (the first PHASE 1 launch gathers r_0 and treats p as 0 by itself).
template <int EPT, bool FOLD>
__global__ __launch_bounds__(NTF) void k_fused_residual(int n, SolverState *st, AsmView vAp,
                                                        const double *__restrict__ b, double *__restrict__ r) {
  __shared__ double sm[NTF / 64 + 1];
  const double eps = st->eps;
  double srr = 0.0, sbb = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    if (e < n) {
      const double bi = b[e];
      const double ri = bi - view_load(vAp, e);
      r[e] = ri;
      srr += ri * ri;
      sbb += bi * bi;
    }
  }
  srr = block_sum_f(srr, sm);
  sbb = block_sum_f(sbb, sm);
  if (threadIdx.x == 0) {
    st->rTr = srr; st->bnorm = sqrt(sbb);
    if (FOLD) {
      st->tol = eps * st->bnorm;
      st->it = 0; st->it_nxt = 0; st->done = 0; st->overflow = 0;
      st->rTz = 0.0; st->rTz_prev = 1.0; st->rTr_prev = srr;
      st->d = 0.0; st->alpha = 0.0; st->beta = 0.0;
    }
  }
}
// Entry of a whole-solve graph of the folded loop that was built for x0 == 0 (the usual call, `pcg(S, b, zeros, M)`): the
// three launches of the general entry (k_solve_begin_g, the skipped `S*x0`, k_fused_residual<., true>) in one — b and x0
// in, r_0 = b, the scalars exactly as k_fused_residual<., true> leaves them (b - 0 = b bit for bit, same sum order).
// A non-zero x0 is reported instead (respec): nothing else of the graph runs (done = 1) and the host replays the
// general form.
template <int EPT>
__global__ __launch_bounds__(NTF) void k_entry_zero(int n, const SolveArgs *pin, double *__restrict__ b, double *__restrict__ x,
                                                    double *__restrict__ r, SolverState *st, SolveArgs *dev, PinnedFlags *flags) {
  __shared__ unsigned long long a[sizeof(SolveArgs) / 8];
  __shared__ double sm[NTF / 64 + 1];
  if (threadIdx.x == 0) flags->t_entry = wall_clock64();
  if (threadIdx.x < sizeof(SolveArgs) / 8) a[threadIdx.x] = sys_load_u64((const unsigned long long *)pin + threadIdx.x);
  __syncthreads();
  const SolveArgs &q = *reinterpret_cast<const SolveArgs *>(a);
  const double *b_in = q.b_in, *x_in = q.x_in;
  double bv[EPT], xv[EPT];
  int nz = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    bv[k] = 0.0; xv[k] = 0.0;
    if (e < n) { bv[k] = b_in[e]; xv[k] = x_in[e]; nz |= xv[k] != 0.0; }
  }
  nz = __syncthreads_or(nz);
  if (threadIdx.x < sizeof(SolveArgs) / 8) ((unsigned long long *)dev)[threadIdx.x] = a[threadIdx.x];
  if (nz) {
    if (threadIdx.x == 0) {
      st->done = 1; st->x0_zero = 0;
      flags->it = 0; flags->done = 0; flags->overflow = 0; flags->respec = 1; flags->x0z = 0;
      __threadfence_system();
      __hip_atomic_store(&flags->seq, q.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  double srr = 0.0, sbb = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    if (e < n) {
      const double bi = bv[k];
      const double ri = bi - 0.0;
      b[e] = bi; x[e] = xv[k]; r[e] = ri;
      srr += ri * ri;
      sbb += bi * bi;
    }
  }
  srr = block_sum_f(srr, sm);
  sbb = block_sum_f(sbb, sm);
  if (threadIdx.x == 0) {
    st->eps = q.eps; st->maxit = q.maxit; st->res_cap = q.res_cap; st->x0_zero = 1;
    st->rTr = srr; st->bnorm = sqrt(sbb);
    st->tol = q.eps * st->bnorm;
    st->it = 0; st->it_nxt = 0; st->done = 0; st->overflow = 0;
    st->rTz = 0.0; st->rTz_prev = 1.0; st->rTr_prev = srr;
    st->d = 0.0; st->alpha = 0.0; st->beta = 0.0;
  }
}
// Set-up, second half: z = view; r'z; p = z; it = 1; res_norm[1]; tol; stop rule (cg.jl:26-33 / 81-89).
// eps, maxit, res_cap arrive through st->tol / st->maxit / st->res_cap (written by the host before the
// launch), so that the launch can live in a replayable graph.
template <int EPT>
__global__ __launch_bounds__(NTF) void k_fused_start(int n, SolverState *st, AsmView vz, const double *__restrict__ r,
                                                     double *__restrict__ p, double *res_norm, int precond) {
  __shared__ double sm[NTF / 64 + 1];
  const double rr = st->rTr, bnorm = st->bnorm, eps = st->eps;
  const long long maxit = st->maxit, cap = st->res_cap;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    if (e < n) {
      const double ze = view_load(vz, e);
      p[e] = ze;
      if (precond) s += r[e] * ze;
    }
  }
  const double rz = precond ? block_sum_f(s, sm) : rr;
  if (threadIdx.x == 0) {
    st->rTz = rz; st->rTr_prev = rr; st->rTz_prev = rz;
    const double tol = eps * bnorm;
    st->tol = tol;
    st->it = 1;
    st->d = 0.0; st->alpha = 0.0; st->beta = 0.0;
    const double res = sqrt(rr);
    st->overflow = 0;
    if (cap >= 1) res_norm[0] = res; else st->overflow = 1;
    st->done = !((1 < maxit) && (res > tol));
  }
}

// Set-up of the deflated solvers: p = z - W mu (defcg.jl:61 / 285); plain copy when nvec == 0.
__global__ __launch_bounds__(NT) void k_init_p(int n, const double *__restrict__ z, double *__restrict__ p,
                                               const double *__restrict__ W, const double *__restrict__ mu, int nvec) {
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    double v = z[i];
    if (nvec > 0) v = v - w_times_mu(W, n, i, mu, nvec);
    p[i] = v;
  }
}
// x += W mu (defcg.jl:54 / 275)
__global__ __launch_bounds__(NT) void k_add_Wmu(int n, double *__restrict__ x, const double *__restrict__ W,
                                                const double *__restrict__ mu, int nvec) {
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
    x[i] = x[i] + w_times_mu(W, n, i, mu, nvec);
  }
}
// part[v*gx + g] = partial of V[:,v] . z   (V = AW for `WtA*z`, V = W for `W'r`); grid (gx, nvec)
__global__ __launch_bounds__(NT) void k_multi_dot_partial(int n, const double *__restrict__ V,
                                                          const double *__restrict__ z, double *__restrict__ part,
                                                          const int *done) {
  if (done && *done) return;
  __shared__ double sm[NT / 64 + 1];
  const double *v = V + (long long)blockIdx.y * n;
  double s = 0.0;
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) s += v[i] * z[i];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = s;
}
// part[v] = V[:,v] . z with z given as a view (small systems: one 1024-thread workgroup per vector, all loads of a
// thread issued together); grid (nvec)
template <int EPT>
__global__ __launch_bounds__(NTF) void k_multi_dot_view(int n, const double *__restrict__ V, AsmView vz,
                                                        double *__restrict__ part, const int *done) {
  if (done && *done) return;
  __shared__ double sm[NTF / 64 + 1];
  const double *v = V + (long long)blockIdx.x * n;
  double a[EPT], b[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = k * NTF + threadIdx.x;
    a[k] = b[k] = 0.0;
    if (e < n) { a[k] = v[e]; b[k] = view_load(vz, e); }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) s += a[k] * b[k];
  s = block_sum_f(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// C[i + j*nvec] = partial-free small gemm entry AW[:,i] . W[:,j]; grid (nvec, nvec), one workgroup each.
__global__ __launch_bounds__(NT) void k_small_gram(int n, const double *__restrict__ AW, const double *__restrict__ W,
                                                   double *__restrict__ C, int nvec) {
  __shared__ double sm[NT / 64 + 1];
  const double *a = AW + (long long)blockIdx.x * n;
  const double *b = W + (long long)blockIdx.y * n;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += NT) s += a[i] * b[i];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) C[blockIdx.x + (long long)blockIdx.y * nvec] = s;
}
// mu = WtAW \ rhs with rhs[v] = sum of partials; LU (unit lower L, U) and pivots from the host
// factorisation (LAPACK getrf/getrs order: all row swaps, forward, backward). One wave.
__global__ __launch_bounds__(64) void k_lu_solve(int nvec, const double *__restrict__ LU, const int *__restrict__ piv,
                                                 const double *__restrict__ part, int gx, double *__restrict__ mu,
                                                 const int *done) {
  if (done && *done) return;
  extern __shared__ double bsh[];
  if (nvec <= 64) {  // one wave, factors staged in LDS (launched with nvec + nvec*nvec + nvec/2 + 1 doubles of LDS)
    double *lu_s = bsh + nvec;
    int *piv_s = reinterpret_cast<int *>(lu_s + nvec * nvec);
    for (int i = threadIdx.x; i < nvec * nvec; i += 64) lu_s[i] = LU[i];
    if ((int)threadIdx.x < nvec) piv_s[threadIdx.x] = piv[threadIdx.x];
    double s = 0.0;
    if ((int)threadIdx.x < nvec)
      for (int g = 0; g < gx; ++g) s += part[threadIdx.x * gx + g];
    __syncthreads();
    s = wave_lu_solve(nvec, lu_s, piv_s, s);
    if ((int)threadIdx.x < nvec) mu[threadIdx.x] = s;
    return;
  }
  for (int v = threadIdx.x; v < nvec; v += 64) {
    double s = 0.0;
    for (int g = 0; g < gx; ++g) s += part[v * gx + g];
    bsh[v] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int k = 0; k < nvec; ++k) {
      const int pk = piv[k];
      if (pk != k) { const double t = bsh[k]; bsh[k] = bsh[pk]; bsh[pk] = t; }
    }
  __syncthreads();
  for (int k = 0; k < nvec; ++k) {
    const double bk = bsh[k];
    for (int i = k + 1 + threadIdx.x; i < nvec; i += 64) bsh[i] -= bk * LU[i + (long long)k * nvec];
    __syncthreads();
  }
  for (int k = nvec - 1; k >= 0; --k) {
    if (threadIdx.x == 0) bsh[k] /= LU[k + (long long)k * nvec];
    __syncthreads();
    const double bk = bsh[k];
    for (int i = threadIdx.x; i < k; i += 64) bsh[i] -= bk * LU[i + (long long)k * nvec];
    __syncthreads();
  }
  for (int v = threadIdx.x; v < nvec; v += 64) mu[v] = bsh[v];
}
// y = a - b (matrix-free: Sdxd .-= A_IΓdd' * v, EPDD.jl:652)
__global__ __launch_bounds__(NT) void k_sub(int n, const double *__restrict__ a, const double *__restrict__ b,
                                            double *__restrict__ y) {
  for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) y[i] = a[i] - b[i];
}

}  // namespace mi
