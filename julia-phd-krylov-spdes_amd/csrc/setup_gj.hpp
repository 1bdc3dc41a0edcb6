// The level elimination of setup_dense.hpp with kernels of this library only, batched over the subdomains and replayed
// from one hipGraph per realization (the library path issues ~250 levels x dozens of small rocSOLVER / rocBLAS kernels
// per subdomain from the host: ~3 s per realization at config 3, bound by the host's launch rate; rocSOLVER cannot be
// captured into a graph).
//
// Formulation. With Z_k = T_k^{-1} kept explicitly (dense, symmetric) the recursion of setup_dense.hpp reads
//     T_k = A_kk - C_k' Z_{k+1} C_k,     g_k = b_k - C_k' (Z_{k+1} g_{k+1}),     Z_k = T_k^{-1},
//     S_d = A_ΓΓ - B' Z_0 B,             w_d = B' (Z_0 g_0),
// where C_k = A_{k+1,k} and B = A_IΓ[L_0, :] are SPARSE (a P1 node has <= 3-4 neighbours in the adjacent level), so every
// entry of C' Z C is a sum of a dozen picked entries of Z ("pick" kernels, one thread per entry) and the only dense
// O(n^3) work per level is the inversion of the SPD matrix T_k: an in-place-style block Gauss-Jordan inversion without
// pivoting (block size 32; the k-th pivot block of an SPD matrix is its k-th Schur complement, SPD again): the 32 x 32
// pivot block is inverted by one wave (rows in registers) — for the first block of a level in a launch of its own, for the
// others inside the previous step's update launch (look-ahead) —, and one launch per block step updates the whole
// matrix from the previous copy (ping-pong buffers: no launch reads what it writes) —
//     row panel   P A_Kj,   column panel   -A_iK P,   trailing   A_ij - A_iK (P A_Kj),   pivot block   P.
// All subdomains advance together (grid.z), aligned so that they reach level 0 in the same step. The launch sequence
// depends only on the plan: it is captured once and replayed per realization on plan-owned buffers.
#pragma once
#include <atomic>

#include "setup_dense.hpp"

namespace mi {

#ifndef MI355_GJ_B
#define MI355_GJ_B 64
#endif
constexpr int GJ_B = MI355_GJ_B;   // pivot block: 64 (one full-matrix pass per 64 eliminated rows; its inverse from two 32-wide
                                   // register inversions and 32 x 32 matrix-core products) or 32 (round 2's form)
constexpr int GJ_H = 32;    // rows one wave inverts in registers
constexpr int GJ_T = 64;    // tile of the update launch (256 threads, 4 x 4 outputs each)
static_assert(GJ_B == 32 || GJ_B == 64, "pivot block");

struct GjStep {             // one per (step, subdomain); n0 == 0: the subdomain is not active in this step
  int n1, n0;               // size of the deeper level (dimension of Z_in; 0 at the subdomain's first step) and of this level
  int zin;                  // which ping-pong buffer holds Z_in (0 / 1)
  int cptr;                 // offset of C_k's column pointers (n0 + 1 entries)
  long long d_e0, d_e1;     // entry list of A_kk
  int b_off;                // offset of this level's nodes in `perm` (gather of b_k)
  int nb;                   // GJ block steps of this level = ceil(n0 / GJ_B)
  // level solves (the inverses Z_k kept after a run): where this level's Z lives in `zstore`, and the ROW form of the
  // coupling block C = A_{this level, next shallower level} (rows = this level's nodes) for the back-substitution
  long long zoff;
  int rptr;
};
struct GjDom {              // per subdomain
  double *T, *Z[2], *y, *g[2], *P;
  int ng, n_last, zfin;     // n_Γd; size of level 0; buffer holding Z_0
  int bptr;                 // offset of B's column pointers (ng + 1 entries)
  long long g_e0, g_e1;     // entry list of A_ΓΓ
  long long s_off, w_off;
  int nlev;
};

struct GjState {
  int nsteps = 0, ndom = 0, nmax = 0, ngmax = 0;
  std::vector<GjStep> steps_h;
  std::vector<GjDom> dom_h;
  std::vector<int> nb_step;          // GJ block steps to launch per step (max over the active subdomains)
  std::vector<int> n_step;           // largest level of the step (grid sizes)
  DevBuf<GjStep> steps;
  DevBuf<GjDom> doms;
  DevBuf<double> pool;               // all work buffers
  DevBuf<double> in_ii, in_ig, in_gg, in_bi, out_S, out_w;   // plan-owned I/O of the captured graph
  hipGraphExec_t graph[2] = {nullptr, nullptr};               // without / with right-hand side
  bool graph_failed = false;
  // level solves: A_IIdd \ f for all subdomains by forward / backward sweeps over the kept level inverses
  bool keep = false;
  DevBuf<double> zstore, gstore, ustore, sv_in, sv_out;
  DevBuf<int> r_ptr, r_col, r_src;
  hipGraphExec_t solve_graph = nullptr;
  bool have_levels = false;                                   // a run with keep == true has filled zstore
  ~GjState() {
    for (auto &g : graph) if (g) (void)hipGraphExecDestroy(g);
    if (solve_graph) (void)hipGraphExecDestroy(solve_graph);
  }
};

#pragma clang fp contract(fast)

// (a runtime index into an array member of a by-value struct sends the whole struct to scratch memory: 120 bytes per thread
// in every kernel below; a select keeps it in scalar registers)
#define GJ_Z(dm, k) ((k) ? (dm).Z[1] : (dm).Z[0])
#define GJ_G(dm, k) ((k) ? (dm).g[1] : (dm).g[0])

// T[i,j] = -(C' Z C)[i,j]  (0 at a subdomain's first step); the entries of A_kk are added by k_gj_scatter.
// One thread per entry of the UPPER triangle (the mirror image is stored with it: C' Z C is symmetric, and the two
// summation orders differ in the last bit only). A column of C holds a handful of entries (a P1 node has <= 3-4
// neighbours in the adjacent level): up to PK of them are fetched into registers first, so that all picked entries of Z
// are requested together instead of one dependent chain per entry; longer columns take the plain loops.
__global__ __launch_bounds__(256) void k_gj_pick(int step, int ndom, const GjStep *__restrict__ steps, const GjDom *__restrict__ doms,
                                                 const int *__restrict__ c_ptr, const int *__restrict__ c_row,
                                                 const int *__restrict__ c_src, const double *__restrict__ ii_val) {
  constexpr int PK = 4;
  if (blockIdx.y < blockIdx.x) return;
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  const int n0 = st.n0;
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= n0 || j >= n0 || i > j) return;
  const GjDom dm = doms[blockIdx.z];
  double acc = 0.0;
  if (st.n1 > 0) {
    const double *Z = GJ_Z(dm, st.zin);
    const int *cp = c_ptr + st.cptr;
    const int ia = cp[i], ib = cp[i + 1], ja = cp[j], jb = cp[j + 1];
    const int na = ib - ia, nb = jb - ja;
    if (na <= PK && nb <= PK) {
      int ra[PK], rb[PK], sa[PK], sb[PK];
      double va[PK], vb[PK];
#pragma unroll
      for (int k = 0; k < PK; ++k) {
        ra[k] = k < na ? c_row[ia + k] : 0; sa[k] = k < na ? c_src[ia + k] : 0;
        rb[k] = k < nb ? c_row[ja + k] : 0; sb[k] = k < nb ? c_src[ja + k] : 0;
      }
#pragma unroll
      for (int k = 0; k < PK; ++k) { va[k] = k < na ? ii_val[sa[k]] : 0.0; vb[k] = k < nb ? ii_val[sb[k]] : 0.0; }
      double z[PK][PK];
#pragma unroll
      for (int p_ = 0; p_ < PK; ++p_)
#pragma unroll
        for (int q = 0; q < PK; ++q) z[p_][q] = (p_ < na && q < nb) ? Z[(size_t)ra[p_] + (size_t)rb[q] * st.n1] : 0.0;
#pragma unroll
      for (int p_ = 0; p_ < PK; ++p_)
        if (p_ < na) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < PK; ++q)
            if (q < nb) s += z[p_][q] * vb[q];
          acc += va[p_] * s;
        }
    } else {
      for (int p_ = ia; p_ < ib; ++p_) {
        const double ci = ii_val[c_src[p_]];
        const double *zr = Z + (size_t)c_row[p_];           // Z[a, :] read as Z[a + b*n1] (symmetric)
        double s = 0.0;
        for (int q = ja; q < jb; ++q) s += zr[(size_t)c_row[q] * st.n1] * ii_val[c_src[q]];
        acc += ci * s;
      }
    }
  }
  dm.T[i + (size_t)j * n0] = -acc;
  if (i != j) dm.T[j + (size_t)i * n0] = -acc;
}
__global__ __launch_bounds__(256) void k_gj_scatter(int step, int ndom, const GjStep *__restrict__ steps, const GjDom *__restrict__ doms,
                                                    const int *__restrict__ src, const int *__restrict__ dst,
                                                    const double *__restrict__ ii_val) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  if (st.n0 == 0) return;
  double *T = doms[blockIdx.z].T;
  for (long long e = st.d_e0 + blockIdx.x * 256ll + threadIdx.x; e < st.d_e1; e += (long long)gridDim.x * 256) T[dst[e]] += ii_val[src[e]];
}
// y = Z g for 64 rows per workgroup (Z symmetric, column-major: a wave reads 64 consecutive rows of one column); the
// columns are dealt to the four waves and the partial sums meet in LDS
__device__ __forceinline__ void gj_gemv64(const double *__restrict__ Z, int n, const double *__restrict__ g, double *__restrict__ y) {
  __shared__ double part[4][64];
  const int r = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  double s = 0.0;
  if (r < n)
    for (int b0 = wv; b0 < n; b0 += 64) {      // sixteen columns in flight per thread (four: 0.4 TB/s — ~100 workgroups cannot fill the chip otherwise); same order of the sum
      double z[16], gg[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int b = b0 + 4 * k;
        z[k] = b < n ? Z[r + (size_t)b * n] : 0.0;
        gg[k] = b < n ? g[b] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) s += z[k] * gg[k];
    }
  part[wv][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && r < n) y[r] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_gj_zg(int step, int ndom, const GjStep *__restrict__ steps, const GjDom *__restrict__ doms) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  if (st.n0 == 0 || (int)blockIdx.x * 64 >= st.n1) return;
  const GjDom dm = doms[blockIdx.z];
  gj_gemv64(GJ_Z(dm, st.zin), st.n1, GJ_G(dm, (step + 1) & 1), dm.y);
}
// g_k = b_k - C_k' y
__global__ __launch_bounds__(256) void k_gj_g(int step, int ndom, const GjStep *__restrict__ steps, const GjDom *__restrict__ doms,
                                              const int *__restrict__ c_ptr, const int *__restrict__ c_row, const int *__restrict__ c_src,
                                              const double *__restrict__ ii_val, const int *__restrict__ perm, const double *__restrict__ bI) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= st.n0) return;
  const GjDom dm = doms[blockIdx.z];
  double v = bI[perm[st.b_off + j]];
  if (st.n1 > 0) {
    const int *cp = c_ptr + st.cptr;
    double s = 0.0;
    for (int p = cp[j]; p < cp[j + 1]; ++p) s += ii_val[c_src[p]] * dm.y[c_row[p]];
    v -= s;
  }
  GJ_G(dm, step & 1)[j] = v;
}

// ---- block Gauss-Jordan inversion of T (n0 x n0, SPD), block step kb: source = T (kb == 0) or Z[kb & 1], destination Z[(kb + 1) & 1]
__device__ __forceinline__ const double *gj_src(const GjDom &dm, int kb) { return kb == 0 ? dm.T : GJ_Z(dm, kb & 1); }
// P = (pivot block)^{-1}: ONE wave per subdomain, lane r holds row r of the block in registers; a Gauss-Jordan step
// broadcasts the scaled pivot row with v_readlane (no LDS round trips, no barriers in the 32-step dependency chain).
// row[c] = lane r's row of a 32 x 32 SPD block (identity beyond the block's size) -> its inverse, in place
__device__ __forceinline__ void gj_invert_rows(double (&row)[GJ_H], int r) {
#pragma unroll
  for (int p = 0; p < GJ_H; ++p) {
    const double piv = 1.0 / lane_read(row[p], p);
    if (r == p) {
#pragma unroll
      for (int c = 0; c < GJ_H; ++c) row[c] = c == p ? piv : row[c] * piv;
    }
    const double f = row[p];
#pragma unroll
    for (int c = 0; c < GJ_H; ++c) {
      const double rp = lane_read(row[c], p);          // the (scaled) pivot row, uniform
      if (r != p && c != p) row[c] -= f * rp;
    }
    if (r != p) row[p] = -f * piv;
  }
}
typedef double gj_d4 __attribute__((ext_vector_type(4)));
// in-place inverse of the 32 x 32 SPD block at M[o.., o..] by ONE wave, all 64 lanes: lane l holds half a row (row l & 31,
// columns 16 (l >> 5) .. + 15) in registers. A Gauss-Jordan step passes the pivot row and the pivot column through 64
// doubles of LDS (`scr`; broadcast reads, no v_readlane chain, no second wave): ~70 instructions per step instead of
// ~200 with whole rows per lane and 64 v_readlanes — the 32-step chain is the critical path of the whole level elimination
// (every block step waits for it), so this is the kernel that sets the set-up's time.
// In-place Gauss-Jordan inverse of the 32 x 32 block at (o, o) of an LDS matrix, by 32 lanes of wave 0: lane r keeps row r
// in registers (32 doubles); a step broadcasts the pivot row with v_readlane (uniform values, straight into the FMAs'
// scalar operand) — no LDS round trip and no barrier inside the 32-step chain, which is the serial part of every block
// step of the elimination (the LDS version with half rows per lane took 0.4 us per step, this one 0.1).
__device__ __forceinline__ double gj_readlane(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void gj_inv32(double *M, int LD, int o, double *scr) {
  (void)scr;
  if (threadIdx.x < 64) {
    const int r = threadIdx.x & 31;          // (lanes 32..63 mirror lanes 0..31: the wave stays convergent, only the lower half stores)
    double a[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = M[(o + r) * LD + o + c];
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const double piv = 1.0 / gj_readlane(a[p], p);
      const double f = a[p];
      const double fp = f * piv;
      const bool me = r == p;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const double pr = gj_readlane(a[c], p);        // row p before this step
        if (c == p) a[c] = me ? piv : -fp;
        else a[c] = me ? pr * piv : a[c] - fp * pr;
      }
    }
    if (threadIdx.x < 32) {
#pragma unroll
      for (int c = 0; c < 32; ++c) M[(o + r) * LD + o + c] = a[c];
    }
  }
}
#ifndef MI355_GJ_HEAD
#define MI355_GJ_HEAD 0            // 0 (default): look-ahead — the tile holding the NEXT pivot block inverts it at the END of the launch (161.5 ms per
                                   // config-3 realization); 1: the pivot block of a step is inverted at the HEAD of its own launch (167.3 ms)
#endif
#ifndef MI355_GJ_INV_RECURSIVE
#define MI355_GJ_INV_RECURSIVE 0   // 1: the 2 x 2 block recursion over two 32-step register inversions (25 us per pivot block instead of 21)
#endif
// inverse of a 4 x 4 matrix (row-major m) whose leading 2 x 2 block and its Schur complement are non-singular (pivot blocks of
// an SPD matrix are): [A B; C D]^{-1} = [Ai + X Si V, -X Si; -Si V, Si],  Ai = A^{-1}, X = Ai B, V = C Ai, S = D - C X, Si = S^{-1}
__device__ __forceinline__ void gj_inv4(const double (&m)[16], double (&p)[16]) {
  const double da = 1.0 / (m[0] * m[5] - m[1] * m[4]);
  const double a00 = m[5] * da, a01 = -m[1] * da, a10 = -m[4] * da, a11 = m[0] * da;
  const double x00 = a00 * m[2] + a01 * m[6], x01 = a00 * m[3] + a01 * m[7], x10 = a10 * m[2] + a11 * m[6], x11 = a10 * m[3] + a11 * m[7];
  const double v00 = m[8] * a00 + m[9] * a10, v01 = m[8] * a01 + m[9] * a11, v10 = m[12] * a00 + m[13] * a10, v11 = m[12] * a01 + m[13] * a11;
  const double s00 = m[10] - (m[8] * x00 + m[9] * x10), s01 = m[11] - (m[8] * x01 + m[9] * x11);
  const double s10 = m[14] - (m[12] * x00 + m[13] * x10), s11 = m[15] - (m[12] * x01 + m[13] * x11);
  const double ds = 1.0 / (s00 * s11 - s01 * s10);
  const double i00 = s11 * ds, i01 = -s01 * ds, i10 = -s10 * ds, i11 = s00 * ds;
  const double y00 = x00 * i00 + x01 * i10, y01 = x00 * i01 + x01 * i11, y10 = x10 * i00 + x11 * i10, y11 = x10 * i01 + x11 * i11;   // X Si
  const double z00 = i00 * v00 + i01 * v10, z01 = i00 * v01 + i01 * v11, z10 = i10 * v00 + i11 * v10, z11 = i10 * v01 + i11 * v11;   // Si V
  p[0] = a00 + (x00 * z00 + x01 * z10); p[1] = a01 + (x00 * z01 + x01 * z11); p[2] = -y00; p[3] = -y01;
  p[4] = a10 + (x10 * z00 + x11 * z10); p[5] = a11 + (x10 * z01 + x11 * z11); p[6] = -y10; p[7] = -y11;
  p[8] = -z00; p[9] = -z01; p[10] = i00; p[11] = i01;
  p[12] = -z10; p[13] = -z11; p[14] = i10; p[15] = i11;
}
// In-place inverse of a 64 x 64 SPD block in LDS (row stride LD) by 256 threads: block Gauss-Jordan with 4 x 4 pivots —
// 16 block steps instead of 64 scalar ones. A step: every thread inverts the pivot block M[K, K] in registers (redundantly:
// no broadcast of the result needed); thread (k, j) forms one entry of the row panel R = P M[K, :] (columns K: P itself);
// then D = C - M[:, K] R is ONE 16 x 16 x 4 matrix-core op per 16 x 16 block (C = M, zero in columns K, so those become the
// column panel -M[:, K] P); rows K take R. Two barriers per step. `Rb`: 4 x 68 doubles of scratch.
// (The scalar chain this replaces — two 32-step register inversions — took 24 of the 25 us of a pivot-block inversion; this
// one takes 21, 1.3 us per block step of which 0.7-0.9 are the four matrix-core ops with their operand moves. A variant that
// keeps the matrix in registers and moves only the pivot rows / columns through LDS measured the same 21 us: the tiles'
// LDS traffic is not what a step waits for. tools/probes/inv_probe.hip.)
#ifndef GJ_STAMP
#define GJ_STAMP(i)
#endif
__device__ __forceinline__ void gj_invert_block64(double *M, int LD, double *Rb) {
  constexpr int RL = 68;
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = l & 15, lk = l >> 4;
#pragma unroll 1
  for (int s = 0; s < 16; ++s) {
    const int k0 = 4 * s;
    double m[16], p[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) m[4 * a + b] = M[(k0 + a) * LD + k0 + b];
    GJ_STAMP(0);
    gj_inv4(m, p);
    GJ_STAMP(1);
    {
      double q0, q1, q2, q3;                       // row wv of P (wave-uniform choice)
      if (wv == 0) { q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3]; }
      else if (wv == 1) { q0 = p[4]; q1 = p[5]; q2 = p[6]; q3 = p[7]; }
      else if (wv == 2) { q0 = p[8]; q1 = p[9]; q2 = p[10]; q3 = p[11]; }
      else { q0 = p[12]; q1 = p[13]; q2 = p[14]; q3 = p[15]; }
      const int j = l, jj = j & 3;
      const bool inK = (j >> 2) == s;
      // (unconditional loads, then the choice: a load under `?:` becomes a branch with its own wait — four serial LDS round trips)
      const double w0 = M[(k0 + 0) * LD + j], w1 = M[(k0 + 1) * LD + j], w2 = M[(k0 + 2) * LD + j], w3 = M[(k0 + 3) * LD + j];
      const double v0 = inK ? (jj == 0 ? 1.0 : 0.0) : w0, v1 = inK ? (jj == 1 ? 1.0 : 0.0) : w1;
      const double v2 = inK ? (jj == 2 ? 1.0 : 0.0) : w2, v3 = inK ? (jj == 3 ? 1.0 : 0.0) : w3;
      Rb[wv * RL + j] = ((q0 * v0 + q1 * v1) + q2 * v2) + q3 * v3;
    }
    GJ_STAMP(2);
    __syncthreads();
    GJ_STAMP(3);
    const double aop = M[(16 * wv + lc) * LD + k0 + lk];        // M[row block wv][K]: A[row = lc][k = lk]
    const bool krows = wv == (s >> 2);
    const int vk = s & 3;
    double rb[4];
    gj_d4 d[4];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) {                            // every operand first (one LDS round trip), then the matrix cores, then the stores
      const int col = 16 * bj + lc;
      rb[bj] = Rb[lk * RL + col];
#pragma unroll
      for (int v = 0; v < 4; ++v) d[bj][v] = M[(16 * wv + lk + 4 * v) * LD + col];
    }
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) {
      const bool cK = ((16 * bj + lc) >> 2) == s;
#pragma unroll
      for (int v = 0; v < 4; ++v) d[bj][v] = cK ? 0.0 : d[bj][v];
      d[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, -rb[bj], d[bj], 0, 0, 0);
    }
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) {
      const int col = 16 * bj + lc;
#pragma unroll
      for (int v = 0; v < 4; ++v) M[(16 * wv + lk + 4 * v) * LD + col] = (krows && v == vk) ? rb[bj] : d[bj][v];   // row 16 wv + lk + 4 v is pivot row k0 + lk exactly then
    }
    GJ_STAMP(4);
    __syncthreads();
    GJ_STAMP(5);
  }
  // symmetric to the last bit (the update kernels read P as symmetric): mirror the upper triangle
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int i = e >> 6, j = e & 63;
    if (j > i) M[j * LD + i] = M[i * LD + j];
  }
  __syncthreads();
}
// in-place inverse of the SPD GJ_B x GJ_B block in LDS (row stride LD; identity beyond the matrix' end), 256 threads. GJ_B = 64:
//   [A B; B' D]^{-1} = [P + Y X', -Y; -Y', S^{-1}],  P = A^{-1},  X = P B,  S = D - B' X,  Y = X S^{-1}
// two 32-step register inversions (the serial part) and five 32 x 32 x 32 products; `W` = 32 x LD doubles of scratch, `scr` = 64 more.
__device__ __forceinline__ void gj_invert_block(double *M, int LD, double *W, double *scr) {
  if (GJ_B == 64 && !MI355_GJ_INV_RECURSIVE) { gj_invert_block64(M, LD, W); return; }
  if (GJ_B == 32) {
    gj_inv32(M, LD, 0, scr);
    __syncthreads();
    return;
  }
  gj_inv32(M, LD, 0, scr);                                              // P = A^{-1}                        (M11)
  __syncthreads();
  // X = P B -> W (32 x 32)
  {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = l & 15, lk = l >> 4, bi = wv >> 1, bj = wv & 1;
    gj_d4 d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4)
      d = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(16 * bi + lc) * LD + kk + lk], M[(kk + lk) * LD + 32 + 16 * bj + lc], d, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) W[(16 * bi + lk + 4 * v) * LD + 16 * bj + lc] = d[v];
  }
  __syncthreads();
  // S = D - B' X   (B' = M21 as stored: the block is symmetric; read M12 transposed)      -> M22
  {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = l & 15, lk = l >> 4, bi = wv >> 1, bj = wv & 1;
    gj_d4 d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4)
      d = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(kk + lk) * LD + 32 + 16 * bi + lc], W[(kk + lk) * LD + 16 * bj + lc], d, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) M[(32 + 16 * bi + lk + 4 * v) * LD + 32 + 16 * bj + lc] -= d[v];
  }
  __syncthreads();
  gj_inv32(M, LD, 32, scr);                                             // S^{-1}                            (M22)
  __syncthreads();
  // Y = X S^{-1} -> M12 (B is no longer needed), then M21 = -Y', M12 = -Y, M11 = P + Y X'
  {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = l & 15, lk = l >> 4, bi = wv >> 1, bj = wv & 1;
    gj_d4 d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4)
      d = __builtin_amdgcn_mfma_f64_16x16x4f64(W[(16 * bi + lc) * LD + kk + lk], M[(32 + kk + lk) * LD + 32 + 16 * bj + lc], d, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) M[(16 * bi + lk + 4 * v) * LD + 32 + 16 * bj + lc] = d[v];      // Y (sign below)
  }
  __syncthreads();
  {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = l & 15, lk = l >> 4, bi = wv >> 1, bj = wv & 1;
    gj_d4 d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; kk += 4)                                   // Y X' : A = Y[i][k], B = X'[k][j] = X[j][k]
      d = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(16 * bi + lc) * LD + 32 + kk + lk], W[(16 * bj + lc) * LD + kk + lk], d, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) M[(16 * bi + lk + 4 * v) * LD + 16 * bj + lc] += d[v];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += 256) {
    const int i = e >> 5, j = e & 31;
    const double y = M[i * LD + 32 + j];
    M[i * LD + 32 + j] = -y;
    M[(32 + j) * LD + i] = -y;
  }
  __syncthreads();
}
// The pivot inverse of block step kb lives in P + (kb & 1) * GJ_B^2: the update launch of step kb reads it while one of its
// workgroups writes the inverse for step kb + 1 into the other half (look-ahead, below). This kernel serves kb = 0 only.
__global__ __launch_bounds__(256) void k_gj_pivot(int step, int kb, int ndom, const GjStep *__restrict__ steps,
                                                  const GjDom *__restrict__ doms) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  if (kb >= st.nb) return;
  const GjDom dm = doms[blockIdx.z];
  const int n = st.n0, k0 = kb * GJ_B, bs = min(GJ_B, n - k0);
  const double *A = gj_src(dm, kb);
  constexpr int LD = GJ_B + 1;
  __shared__ double Mb[GJ_B * LD];
  __shared__ double Wb[GJ_H * LD + 64];
  for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256) {
    const int r = e % GJ_B, c = e / GJ_B;
    Mb[r * LD + c] = (r < bs && c < bs) ? A[(k0 + r) + (size_t)(k0 + c) * n] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  gj_invert_block(Mb, LD, Wb, Wb + GJ_H * LD);
  double *Pd = dm.P + (kb & 1) * (GJ_B * GJ_B);
  for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256) Pd[e] = Mb[(e % GJ_B) * LD + e / GJ_B];   // column-major GJ_B x GJ_B
}
// one 64 x 64 tile of the updated matrix from the previous copy, on the fp64 matrix cores (v_mfma_f64_16x16x4_f64:
// A[l & 15][k = l >> 4], B[k = l >> 4][l & 15], D: col = l & 15, row = (l >> 4) + 4 reg). R = P A[K, J] (32 x 64) first, then
// the trailing update computed TRANSPOSED — the instruction's row index runs over the tile's columns j, its column index
// over the tile's rows i — so that a lane's four results sit in 16-lane groups of consecutive i: loads and stores of the
// column-major matrix are 128-byte segments. Wave v owns rows 16 v .. 16 v + 15 of the tile.
__device__ __forceinline__ unsigned *gj_ready_flag(const GjDom &dm) { return reinterpret_cast<unsigned *>(dm.P + 2 * GJ_B * GJ_B); }
__global__ void k_gj_reset(int ndom, const GjDom *__restrict__ doms) {
  if ((int)threadIdx.x < ndom) *gj_ready_flag(doms[threadIdx.x]) = 0u;
}
template <bool HEAD_T>
__global__ __launch_bounds__(256, 3) void k_gj_update(int step, int kb, int ndom, const GjStep *__restrict__ steps,
                                                   const GjDom *__restrict__ doms, unsigned seq, int tiles_x) {
  // HEAD (MI355_GJ_HEAD=1, measured and not the default): the pivot block of THIS step is inverted at the head of the launch
  // by its own workgroup while every other tile loads its operands (the bandwidth-bound 7-10 us of a launch); the others
  // then wait for a per-subdomain flag and fetch P; no separate first-pivot launch per level. Small levels gain (37 us per
  // launch against ~45), the large ones lose more: their waiting tiles keep compute units that the look-ahead form uses.
  constexpr bool HEAD = HEAD_T && GJ_B == GJ_T;
  // HEAD: a 1-D grid whose FIRST ndom workgroups are the pivot tiles of the subdomains — dispatched before any tile that
  // will wait for them, whatever the number of resident workgroups (with the pivot tile merely first in its subdomain's
  // slab, slabs beyond the resident set started their inversion only when earlier slabs had finished: 128 us launches).
  // `tiles_x` = tiles per row of the (square) tile grid.
  int dz, gbx, gby;
  if (HEAD) {
    const int b = blockIdx.x;
    if (b < ndom) { dz = b; gbx = gby = kb; }
    else { const int t = b - ndom; gbx = t % tiles_x; gby = (t / tiles_x) % tiles_x; dz = t / (tiles_x * tiles_x); }
  } else { dz = blockIdx.z; gbx = blockIdx.x; gby = blockIdx.y; }
  const GjStep st = steps[(size_t)step * ndom + dz];
  const int n = st.n0;
  if (kb >= st.nb || gbx * GJ_T >= n || gby * GJ_T >= n) return;   // (the swap below stays inside the grid: td * GJ_T < n)
  if (HEAD && (int)blockIdx.x >= ndom && gbx == kb && gby == kb) return;   // the pivot tile is one of the first ndom workgroups
  const GjDom dm = doms[dz];
  const int k0 = kb * GJ_B, bs = min(GJ_B, n - k0);
  const double *A = gj_src(dm, kb);
  double *O = GJ_Z(dm, (kb + 1) & 1);
  // Look-ahead: the tile that holds the NEXT pivot block (diagonal tile td) also inverts it once it has updated it, so that
  // the next block step needs no pivot launch of its own (16 us of a serial 32-step chain per step, 5 400 steps per
  // realization at config 3). That tile is dealt first (swapped with tile (0, 0)) so that its longer life stays inside the launch.
  const int k1 = k0 + GJ_B, td = HEAD ? kb : k1 / GJ_T;
  const bool ahead = HEAD || kb + 1 < st.nb;
  int bx = gbx, by = gby;
  if (ahead && !HEAD) {
    if (bx == 0 && by == 0) bx = by = td;
    else if (bx == td && by == td) bx = by = 0;
  }
  const bool special = HEAD ? (int)blockIdx.x < ndom : (ahead && bx == td && by == td);
  // SYMMETRY. With σ(i) = -1 for the indices already swept (i < k0) and +1 otherwise, the matrix between two block steps
  // satisfies M[j,i] = σ(i) σ(j) M[i,j] (the inverse pivot block and the trailing part are symmetric, the row panel P A_Kj
  // and the column panel -A_iK P are each other's negative transpose; induction over the steps, P symmetric). Only the
  // tiles on and above the diagonal are computed; an off-diagonal tile also writes its mirror image (transposed through
  // LDS so that both stores are 128-byte segments). Storage stays full: the pick kernels and the next step's panel loads
  // read any entry.
  if (bx > by) return;
  const int i0 = bx * GJ_T, j0 = by * GJ_T;
#ifdef MI355_GJ_STAMPS
  const bool dbg = step == 120 && kb == 2 && threadIdx.x == 0 && (special || ((bx * 7 + by * 3 + dz) % 29) == 0);
  long long ts[6] = {0, 0, 0, 0, 0, 0};
  if (dbg) ts[0] = wall_clock64();
#define GJ_KSTAMP(i) do { if (dbg) ts[i] = wall_clock64(); } while (0)
#else
#define GJ_KSTAMP(i)
#endif
  // LDS: A[K, J] / R (GJ_B x 65) and A[I, K] (64 x (GJ_B + 1)): 33.4 KB with a 32-wide pivot block (three workgroups per CU),
  // 66.6 KB with a 64-wide one (two). The pivot inverse P is NOT staged: every lane keeps the GJ_B / 4 entries it feeds to
  // the matrix cores in registers (one request per entry, served by L2: every tile reads the same 32 KB).
  // With a pivot block as wide as a tile A[I, K] is not staged either: every lane holds the 16 entries of its row that it
  // feeds to the matrix cores (CC_REGS). The second buffer is then only the scratch of the look-ahead inversion: 50 KB per
  // workgroup, THREE workgroups per CU — the 530-620 tiles of a config-3 launch are resident at once instead of running
  // a second round for the last few (measured: the tile loop alone 44 -> see profiles/NOTES.md).
  constexpr bool CC_REGS = GJ_B == GJ_T;
  constexpr int LDS_R = GJ_B * (GJ_T + 1), LDS_CC = CC_REGS ? GJ_H * (GJ_T + 1) + 64 : GJ_T * (GJ_B + 1);
  static_assert(LDS_R + LDS_CC >= GJ_T * (GJ_T + 1), "the mirror image of a tile is staged in the R and Cc buffers");
  __shared__ double lds[LDS_R + LDS_CC];
  double (&R)[GJ_B][GJ_T + 1] = *reinterpret_cast<double (*)[GJ_B][GJ_T + 1]>(lds);           // A[K, J] first, then R = P * A[K, J], then (look-ahead tile) the next pivot block
  double (&Cc)[GJ_T][GJ_B + 1] = *reinterpret_cast<double (*)[GJ_T][GJ_B + 1]>(lds + LDS_R);  // A[I, K]
  double (&Ak)[GJ_B][GJ_T + 1] = R;
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, lc = l & 15, lk = l >> 4;
  const double *Pcur = dm.P + (kb & 1) * (GJ_B * GJ_B);   // column-major, symmetric
  // R = P A[K, J]: (GJ_B / 16) x 4 blocks of 16 x 16; wave wv owns t-block(s) and j-blocks as below
  constexpr int TBW = GJ_B == 64 ? 1 : 1, NQ = GJ_B == 64 ? 4 : 2;      // blocks per wave: 4 (64: tb = wv, all j) or 2 (32: tb = wv & 1, j = 2 (wv >> 1) + q)
  (void)TBW;
  const int tb = GJ_B == 64 ? wv : (wv & 1), jb0 = GJ_B == 64 ? 0 : 2 * (wv >> 1);
  double pa[GJ_B / 4];                                                   // P[16 tb + lc][kk + lk], kk = 0, 4, ...
  if (HEAD && special) {
    // the pivot block of this step: load, invert, publish (write-through stores, acknowledged, then the flag), and store
    // this tile of the result — the inverse itself
    for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256) {
      const int r = e % GJ_B, c = e / GJ_B;
      R[r][c] = (r < bs && c < bs) ? A[(k0 + r) + (size_t)(k0 + c) * n] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    GJ_KSTAMP(1);
    gj_invert_block(&R[0][0], GJ_T + 1, lds + LDS_R, lds + LDS_R + (GJ_B == 32 ? 0 : GJ_H * (GJ_T + 1)));
    GJ_KSTAMP(2);
    double *Pw = dm.P + (kb & 1) * (GJ_B * GJ_B);
    for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256) __hip_atomic_store(&Pw[e], R[e % GJ_B][e / GJ_B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(gj_ready_flag(dm), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    GJ_KSTAMP(3);
#ifdef MI355_GJ_STAMPS
    if (dbg) printf("PIVOT tile %2d %2d dom %d: start %lld loaded +%lld inverted +%lld published +%lld (n = %d)\n", bx, by, dz, ts[0], ts[1] - ts[0], ts[2] - ts[0], ts[3] - ts[0], n);
#endif
    for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256) {
      const int r = e % GJ_B, c = e / GJ_B;
      if (r < bs && c < bs) O[(k0 + r) + (size_t)(k0 + c) * n] = R[r][c];
    }
    return;
  }
  if (!HEAD) {
#pragma unroll
    for (int q = 0; q < GJ_B / 4; ++q) pa[q] = Pcur[(16 * tb + lc) + (size_t)(4 * q + lk) * GJ_B];
  }
  for (int e = threadIdx.x; e < GJ_B * GJ_T; e += 256) {
    const int t = e % GJ_B, c = e / GJ_B;        // A[k0 + t, j0 + c]: consecutive threads walk down a column
    Ak[t][c] = (t < bs && j0 + c < n) ? A[(k0 + t) + (size_t)(j0 + c) * n] : 0.0;
  }
  double cb[GJ_B / 4];                           // CC_REGS: A[i0 + 16 wv + lc, k0 + 4 q + lk]
  if (CC_REGS) {
    const int ir = i0 + 16 * wv + lc;
#pragma unroll
    for (int q = 0; q < GJ_B / 4; ++q) cb[q] = (4 * q + lk < bs && ir < n) ? A[ir + (size_t)(k0 + 4 * q + lk) * n] : 0.0;
  } else {
    for (int t = wv; t < GJ_B; t += 4)           // A[i0 + l, k0 + t]
      Cc[l][t] = (t < bs && i0 + l < n) ? A[(i0 + l) + (size_t)(k0 + t) * n] : 0.0;
  }
  // the old entries this lane will update (D layout of the transposed product): i = i0 + 16 wv + lc, j = j0 + 16 jt + lk + 4 v
  const int i = i0 + 16 * wv + lc;
  gj_d4 acc[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = j0 + 16 * jt + lk + 4 * v;
      acc[jt][v] = (i < n && j < n) ? A[i + (size_t)j * n] : 0.0;
    }
  // With a pivot block as wide as a tile the tiles of block column K (by == kb, above the diagonal) hold only column-panel
  // entries -(A_iK P): the SAME matrix-core loop as the trailing update with P in place of R and a zero start
  // (D[j][i] = Σ_u P[u][j] (-A[i][k0 + u])); the diagonal tile (kb, kb) becomes P itself.
  const bool kcol = GJ_B == GJ_T && by == kb;
  GJ_KSTAMP(4);   // operand loads issued
  if (HEAD) {
    // P of this step: wait for the pivot tile's flag (bounded: a tile that gives up poisons its results, the finite check
    // of the run then fails), then fetch it past the caches it was written through
    __shared__ int gave_up;
    if (threadIdx.x == 0) {
      const unsigned *fl = gj_ready_flag(dm);
      long long spins = 0;
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq && spins < (1ll << 22)) { __builtin_amdgcn_s_sleep(32); ++spins; }
      gave_up = spins >= (1ll << 22);
    }
    __syncthreads();
    // Plain loads: no L2 holds a line of this P yet — the launch began with the caches' acquire, nobody has read P since
    // (every reader waits for the flag), and the pivot tile wrote it through to memory before the flag. (Loads that bypass
    // the L2 instead — 600 tiles x 32 KB from the same few channels — made a launch 20 us longer.)
    const double poison = gave_up ? __builtin_nan("") : 0.0;
#pragma unroll
    for (int q = 0; q < GJ_B / 4; ++q) pa[q] = Pcur[(16 * tb + lc) + (size_t)(4 * q + lk) * GJ_B] + poison;
  }
  __syncthreads();
  GJ_KSTAMP(1);   // operands loaded
  if (!kcol) {
    gj_d4 d[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int jbk = jb0 + q;
      d[q] = gj_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < GJ_B; kk += 4)
        d[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kk / 4], Ak[kk + lk][16 * jbk + lc], d[q], 0, 0, 0);
    }
    __syncthreads();                             // every wave has read A[K, J]: the buffer becomes R
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int jbk = jb0 + q;
#pragma unroll
      for (int v = 0; v < 4; ++v) R[16 * tb + lk + 4 * v][16 * jbk + lc] = d[q][v];
    }
  } else {
    __syncthreads();
    for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256)    // P[u][jl] (columns beyond bs: identity, times a zero panel)
      R[e % GJ_B][e / GJ_B] = Pcur[e];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = gj_d4{0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();
  // acc[jt] (rows j, cols i) -= R[K, J_jt]' * Cc[I, K]'  ==  (A_ij - A_iK (P A_Kj))'
#pragma unroll
  for (int kk = 0; kk < GJ_B; kk += 4) {
    const double bneg = CC_REGS ? -cb[kk / 4] : -Cc[16 * wv + lc][kk + lk];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(R[kk + lk][16 * jt + lc], bneg, acc[jt], 0, 0, 0);
  }
  const bool ik = i >= k0 && i < k0 + bs;
  const int bs1 = min(GJ_B, n - k1);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int jl = 16 * jt + lk + 4 * v, j = j0 + jl;
      if (i >= n || j >= n) continue;
      const bool jk = j >= k0 && j < k0 + bs;
      double val = acc[jt][v];                                                 // A_ij - A_iK (P A_Kj);  block column K: -(A_iK P)[i, j - k0]
      if (ik && jk) val = Pcur[(i - k0) + (size_t)(j - k0) * GJ_B];
      else if (ik) val = R[i - k0][jl];                                        // (P A_Kj)[i - k0, j]
      else if (!CC_REGS && jk && !kcol) {                                      // -(A_iK P)[i, j - k0]   (32-wide pivot block inside a 64-wide tile)
        double s2 = 0.0;
        for (int u = 0; u < GJ_B; ++u) s2 += Cc[16 * wv + lc][u] * Pcur[u + (size_t)(j - k0) * GJ_B];
        val = -s2;
      }
      O[i + (size_t)j * n] = val;
      acc[jt][v] = val;
    }
  GJ_KSTAMP(2);   // tile updated and stored
#ifdef MI355_GJ_STAMPS
  if (dbg && !special) printf("tile %2d %2d dom %d: start %lld loaded +%lld updated +%lld (loads issued +%lld)\n", bx, by, dz, ts[0], ts[1] - ts[0], ts[2] - ts[0], ts[4] - ts[0]);
#endif
  if (bx != by) {                                // the mirror image: M[j,i] = σ'(i) σ'(j) M[i,j], σ' = -1 below k1 (swept after this step)
    __syncthreads();                             // R, Cc are read for the last time above
    double *T = lds;                             // [GJ_T][GJ_T + 1]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int jl = 16 * jt + lk + 4 * v, j = j0 + jl;
        const double sg = ((i < k1) != (j < k1)) ? -1.0 : 1.0;
        T[jl * (GJ_T + 1) + 16 * wv + lc] = sg * acc[jt][v];
      }
    __syncthreads();
    const int r = j0 + 16 * wv + lc;             // row of the mirror tile (a column index of this tile)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int cl = 16 * jt + lk + 4 * v, cg = i0 + cl;
        if (r < n && cg < n) O[r + (size_t)cg * n] = T[(16 * wv + lc) * (GJ_T + 1) + cl];
      }
    return;
  }
  if (HEAD || !special) return;
  // Look-ahead: this diagonal tile holds the next pivot block (rows / columns k1 .. k1 + bs1 of the updated matrix); its inverse
  // goes to the other half of P. R becomes the landing zone (row stride GJ_T + 1), Cc the scratch of the block inversion.
  __syncthreads();
  for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256) { const int rr = e / GJ_B, cc = e % GJ_B; R[rr][cc] = rr == cc ? 1.0 : 0.0; }   // identity beyond the matrix' end
  __syncthreads();
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = j0 + 16 * jt + lk + 4 * v;
      if (i >= k1 && i < k1 + bs1 && j >= k1 && j < k1 + bs1) R[i - k1][j - k1] = acc[jt][v];   // (trailing entries: K1 != K)
    }
  __syncthreads();
  static_assert(LDS_CC >= GJ_H * (GJ_T + 1) + 64 || GJ_B == 32, "scratch of the block inversion lives in the Cc buffer");
  gj_invert_block(&R[0][0], GJ_T + 1, lds + LDS_R, lds + LDS_R + (GJ_B == 32 ? 0 : GJ_H * (GJ_T + 1)));
  GJ_KSTAMP(3);   // pivot block inverted
  double *Pn = dm.P + ((kb + 1) & 1) * (GJ_B * GJ_B);
  for (int e = threadIdx.x; e < GJ_B * GJ_B; e += 256) Pn[e] = R[e % GJ_B][e / GJ_B];
#ifdef MI355_GJ_STAMPS
  if (dbg) printf("LOOK-AHEAD tile %2d %2d dom %d: start %lld loaded +%lld updated +%lld inverted +%lld (n = %d)\n", bx, by, dz, ts[0], ts[1] - ts[0], ts[2] - ts[0], ts[3] - ts[0], n);
#endif
}
// ---- the end of a subdomain's chain: S (upper triangle mirrored) = A_ΓΓ - B' Z_0 B; w = B' (Z_0 g_0)
__global__ __launch_bounds__(256) void k_gj_final_pick(int ndom, const GjDom *__restrict__ doms, const int *__restrict__ c_ptr,
                                                       const int *__restrict__ c_row, const int *__restrict__ c_src,
                                                       const double *__restrict__ ig_val) {
  const GjDom dm = doms[blockIdx.z];
  const int ng = dm.ng;
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= ng || j >= ng) return;
  double acc = 0.0;
  if (dm.n_last > 0) {
    const double *Z = GJ_Z(dm, dm.zfin);
    const int *cp = c_ptr + dm.bptr;
    for (int p = cp[i]; p < cp[i + 1]; ++p) {
      const double ci = ig_val[c_src[p]];
      const double *zr = Z + (size_t)c_row[p];
      double s = 0.0;
      for (int q = cp[j]; q < cp[j + 1]; ++q) s += zr[(size_t)c_row[q] * dm.n_last] * ig_val[c_src[q]];
      acc += ci * s;
    }
  }
  dm.T[i + (size_t)j * ng] = -acc;
}
__global__ __launch_bounds__(256) void k_gj_final_scatter(int ndom, const GjDom *__restrict__ doms, const int *__restrict__ src,
                                                          const int *__restrict__ dst, const double *__restrict__ gg_val) {
  const GjDom dm = doms[blockIdx.z];
  for (long long e = dm.g_e0 + blockIdx.x * 256ll + threadIdx.x; e < dm.g_e1; e += (long long)gridDim.x * 256) dm.T[dst[e]] += gg_val[src[e]];
}
__global__ __launch_bounds__(256) void k_gj_final_sym(int ndom, const GjDom *__restrict__ doms, double *__restrict__ Sd) {
  const GjDom dm = doms[blockIdx.z];
  const int ng = dm.ng;
  const long long tot = (long long)ng * ng;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
    const int i = (int)(e % ng), j = (int)(e / ng);
    Sd[dm.s_off + e] = i <= j ? dm.T[i + (size_t)j * ng] : dm.T[j + (size_t)i * ng];   // `Symmetric(Array(...))`, EPDD.jl:692
  }
}
__global__ __launch_bounds__(256) void k_gj_final_zg(int ndom, int last_step, const GjDom *__restrict__ doms) {
  const GjDom dm = doms[blockIdx.z];
  if ((int)blockIdx.x * 64 >= dm.n_last) return;
  gj_gemv64(GJ_Z(dm, dm.zfin), dm.n_last, GJ_G(dm, last_step & 1), dm.y);
}
__global__ __launch_bounds__(256) void k_gj_final_w(int ndom, const GjDom *__restrict__ doms, const int *__restrict__ c_ptr,
                                                    const int *__restrict__ c_row, const int *__restrict__ c_src,
                                                    const double *__restrict__ ig_val, double *__restrict__ w) {
  const GjDom dm = doms[blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= dm.ng) return;
  double s = 0.0;
  if (dm.n_last > 0) {
    const int *cp = c_ptr + dm.bptr;
    for (int p = cp[j]; p < cp[j + 1]; ++p) s += ig_val[c_src[p]] * dm.y[c_row[p]];
  }
  w[dm.w_off + j] = s;
}
// ---- level solves. After a run with `keep`, zstore holds Z_k = T_k^{-1} of every level of every subdomain, and
//     A_IIdd \ f   =   forward:  g_m = f_m,  g_k = f_k - C_k' (Z_{k+1} g_{k+1})        (deepest level first: the run's own recursion)
//                      backward: u_0 = Z_0 g_0,  u_{k+1} = Z_{k+1} (g_{k+1} - C_k u_k)  (C_k = A_{k+1,k})
// is an EXACT interior solve (block LDL' over the breadth-first levels): two streams of the kept inverses instead of the
// thousands of A_II SpMVs of the reference's `IterativeSolvers.cg` (EPDD.jl:648-650).
__global__ __launch_bounds__(256) void k_gj_keep(int step, int ndom, const GjStep *__restrict__ steps, const GjDom *__restrict__ doms,
                                                 double *__restrict__ zstore) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  if (st.n0 == 0) return;
  const GjDom dm = doms[blockIdx.z];
  const double *src = GJ_Z(dm, st.nb & 1);
  double *dst = zstore + st.zoff;
  const long long tot = (long long)st.n0 * st.n0;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) dst[e] = src[e];
}
// y = Z_{k+1} g_{k+1} (the deeper level, handled one step earlier)
__global__ __launch_bounds__(256) void k_lv_zg(int step, int ndom, const GjStep *__restrict__ steps, const GjDom *__restrict__ doms,
                                               const double *__restrict__ zstore, const double *__restrict__ gstore) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  if (st.n0 == 0 || st.n1 == 0 || (int)blockIdx.x * 64 >= st.n1) return;
  const GjStep sp = steps[(size_t)(step - 1) * ndom + blockIdx.z];
  gj_gemv64(zstore + sp.zoff, st.n1, gstore + sp.b_off, doms[blockIdx.z].y);
}
// g_k = f_k - C_k' y  (f in the caller's interior order, g in level order)
__global__ __launch_bounds__(256) void k_lv_g(int step, int ndom, const GjStep *__restrict__ steps, const GjDom *__restrict__ doms,
                                              const int *__restrict__ c_ptr, const int *__restrict__ c_row, const int *__restrict__ c_src,
                                              const double *__restrict__ ii_val, const int *__restrict__ perm, const double *__restrict__ f,
                                              double *__restrict__ gstore) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= st.n0) return;
  double v = f[perm[st.b_off + j]];
  if (st.n1 > 0) {
    const double *y = doms[blockIdx.z].y;
    const int *cp = c_ptr + st.cptr;
    double s2 = 0.0;
    for (int p = cp[j]; p < cp[j + 1]; ++p) s2 += ii_val[c_src[p]] * y[c_row[p]];
    v -= s2;
  }
  gstore[st.b_off + j] = v;
}
// u = Z (g - C u_shallower) for the level of `step` (64 rows per workgroup; the operand is formed in LDS first: every
// workgroup needs all of it). last == 1: the shallowest level, u_0 = Z_0 g_0.
constexpr int LV_MAX = 2048;
__global__ __launch_bounds__(256) void k_lv_back(int step, int last, int ndom, const GjStep *__restrict__ steps,
                                                 const double *__restrict__ zstore, const double *__restrict__ gstore,
                                                 const int *__restrict__ r_ptr, const int *__restrict__ r_col, const int *__restrict__ r_src,
                                                 const double *__restrict__ ii_val, double *__restrict__ ustore) {
  const GjStep st = steps[(size_t)step * ndom + blockIdx.z];
  const int n = st.n0;
  if (n == 0 || (int)blockIdx.x * 64 >= n) return;
  __shared__ double t[LV_MAX];
  __shared__ double part[4][64];
  const double *g = gstore + st.b_off;
  if (last) {
    for (int i = threadIdx.x; i < n; i += 256) t[i] = g[i];
  } else {
    const GjStep sn = steps[(size_t)(step + 1) * ndom + blockIdx.z];      // the shallower level: its u is known
    const double *us = ustore + sn.b_off;
    const int *rp = r_ptr + st.rptr;
    for (int i = threadIdx.x; i < n; i += 256) {
      double s2 = 0.0;
      for (int p = rp[i]; p < rp[i + 1]; ++p) s2 += ii_val[r_src[p]] * us[r_col[p]];
      t[i] = g[i] - s2;
    }
  }
  __syncthreads();
  const double *Z = zstore + st.zoff;
  const int r = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  double s2 = 0.0;
  if (r < n)
    for (int b0 = wv; b0 < n; b0 += 64) {
      double z[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) { const int b = b0 + 4 * k; z[k] = b < n ? Z[r + (size_t)b * n] : 0.0; }
#pragma unroll
      for (int k = 0; k < 16; ++k) { const int b = b0 + 4 * k; if (b < n) s2 += z[k] * t[b]; }
    }
  part[wv][threadIdx.x & 63] = s2;
  __syncthreads();
  if (threadIdx.x < 64 && r < n) ustore[st.b_off + r] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_lv_scatter(long long n, const int *__restrict__ perm, const double *__restrict__ ustore, double *__restrict__ u) {
  for (long long q = blockIdx.x * 256ll + threadIdx.x; q < n; q += (long long)gridDim.x * 256) u[perm[q]] = ustore[q];
}
#pragma clang fp contract(off)

inline void gj_build(mi_setup_s &P) {
  std::unique_ptr<GjState> G(new GjState);
  G->ndom = P.ndom;
  int smax = 0;
  for (auto &D : P.dom) { smax = std::max(smax, D.nlev); G->nmax = std::max(G->nmax, D.max_lev); G->ngmax = std::max(G->ngmax, D.n_g); }
  G->nsteps = smax;
  G->steps_h.assign((size_t)std::max(1, smax) * P.ndom, GjStep{});
  G->nb_step.assign(std::max(1, smax), 0);
  G->n_step.assign(std::max(1, smax), 1);
  // work buffers: T also receives the final n_Γd x n_Γd pick
  size_t tot = 0;
  std::vector<size_t> off_T(P.ndom), off_Z0(P.ndom), off_Z1(P.ndom), off_y(P.ndom), off_g0(P.ndom), off_g1(P.ndom), off_P(P.ndom);
  for (int d = 0; d < P.ndom; ++d) {
    const SetupDom &D = P.dom[d];
    const size_t nm = (size_t)std::max(1, D.max_lev), nt = (size_t)std::max<int>(std::max(1, D.max_lev), D.n_g);
    auto take = [&](size_t cnt) { const size_t o = tot; tot += (cnt + 31) / 32 * 32; return o; };
    off_T[d] = take(nt * nt); off_Z0[d] = take(nm * nm); off_Z1[d] = take(nm * nm);
    off_y[d] = take(nm); off_g0[d] = take(nm); off_g1[d] = take(nm); off_P[d] = take(2 * GJ_B * GJ_B + 16);   // (+ the step's ready flag)
  }
  G->pool.alloc(tot + 32);
  memset_sync(G->pool.p, 0, sizeof(double) * (tot + 32));
  G->dom_h.resize(P.ndom);
  for (int d = 0; d < P.ndom; ++d) {
    const SetupDom &D = P.dom[d];
    GjDom &q = G->dom_h[d];
    q.T = G->pool.p + off_T[d]; q.Z[0] = G->pool.p + off_Z0[d]; q.Z[1] = G->pool.p + off_Z1[d];
    q.y = G->pool.p + off_y[d]; q.g[0] = G->pool.p + off_g0[d]; q.g[1] = G->pool.p + off_g1[d]; q.P = G->pool.p + off_P[d];
    q.ng = D.n_g; q.nlev = D.nlev; q.bptr = D.bptr_off; q.g_e0 = D.g_e0; q.g_e1 = D.g_e1; q.s_off = D.s_off; q.w_off = D.w_off;
    q.n_last = D.nlev ? D.lev_off[1] - D.lev_off[0] : 0;
    q.zfin = 0;
    int zin = 0;
    for (int k = D.nlev - 1; k >= 0; --k) {           // level k is handled in step smax - 1 - k
      const int step = smax - 1 - k;
      GjStep &st = G->steps_h[(size_t)step * P.ndom + d];
      st.n0 = D.lev_off[k + 1] - D.lev_off[k];
      st.n1 = k + 1 < D.nlev ? D.lev_off[k + 2] - D.lev_off[k + 1] : 0;
      st.zin = zin;
      st.cptr = k + 1 < D.nlev ? D.cptr_off[k] : 0;
      st.d_e0 = D.d_e0[k]; st.d_e1 = D.d_e0[k + 1];
      st.b_off = (int)D.bi_off + D.lev_off[k];
      st.nb = (st.n0 + GJ_B - 1) / GJ_B;
      zin = st.nb & 1;                                 // block step kb writes Z[(kb + 1) & 1]: the inverse ends in Z[nb & 1]
      G->nb_step[step] = std::max(G->nb_step[step], st.nb);
      G->n_step[step] = std::max(G->n_step[step], std::max(st.n0, st.n1));
    }
    q.zfin = zin;
  }
  hipStream_t s = P.ctx->stream;
  G->steps.upload(G->steps_h, s);
  G->doms.upload(G->dom_h, s);
  G->in_ii.alloc((size_t)P.n_ii + 1); G->in_ig.alloc((size_t)P.n_ig + 1); G->in_gg.alloc((size_t)P.n_gg + 1);
  G->in_bi.alloc((size_t)P.n_bi + 1); G->out_S.alloc((size_t)P.n_s + 1); G->out_w.alloc((size_t)P.n_w + 1);
  P.gj = std::move(G);
}

// enqueue the whole elimination on `s` (plain launches: capturable)
inline void gj_enqueue(mi_setup_s &P, hipStream_t s, const double *ii, const double *ig, const double *gg, const double *bI, double *Sd,
                       double *w) {
  GjState &G = *P.gj;
  const int nd = P.ndom;
  const GjStep *st = G.steps.p;
  const GjDom *dm = G.doms.p;
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  int nm = std::max(1, G.nmax);
  // levels of at most `head_maxn` nodes invert the step's pivot block at the head of the launch, larger ones look ahead
  static const int head_maxn = GJ_B == GJ_T ? env_int("MI355_GJ_HEAD_MAXN", MI355_GJ_HEAD ? (1 << 30) : 0) : 0;
  unsigned seq = 0;
  if (nd > 1024) raise(MI_ERR_BAD_ARG, "device set-up: more than 1024 subdomains in one plan");
  hipLaunchKernelGGL(k_gj_reset, dim3(1), dim3(1024), 0, s, nd, dm);
  for (int step = 0; step < G.nsteps; ++step) {
    nm = G.n_step[step];   // grids cover the largest level of this step only
    if (bI && step > 0) hipLaunchKernelGGL(k_gj_zg, dim3(cdiv(nm, 64), 1, nd), dim3(256), 0, s, step, nd, st, dm);
    hipLaunchKernelGGL(k_gj_pick, dim3(cdiv(nm, 16), cdiv(nm, 16), nd), dim3(256), 0, s, step, nd, st, dm, P.c_ptr.p, P.c_row.p, P.c_src.p, ii);
    hipLaunchKernelGGL(k_gj_scatter, dim3(8, 1, nd), dim3(256), 0, s, step, nd, st, dm, P.src.p, P.dst.p, ii);
    if (bI) hipLaunchKernelGGL(k_gj_g, dim3(cdiv(nm, 256), 1, nd), dim3(256), 0, s, step, nd, st, dm, P.c_ptr.p, P.c_row.p, P.c_src.p, ii, P.perm.p, bI);
    const bool gj_head = nm <= head_maxn;
    for (int kb = 0; kb < G.nb_step[step]; ++kb) {
      if (kb == 0 && !gj_head) hipLaunchKernelGGL(k_gj_pivot, dim3(1, 1, nd), dim3(256), 0, s, step, kb, nd, st, dm);   // later pivots: look-ahead in the update
      const int tx = cdiv(nm, GJ_T);
      if (gj_head) hipLaunchKernelGGL(k_gj_update<true>, dim3(nd + tx * tx * nd), dim3(256), 0, s, step, kb, nd, st, dm, ++seq, tx);
      else hipLaunchKernelGGL(k_gj_update<false>, dim3(tx, tx, nd), dim3(256), 0, s, step, kb, nd, st, dm, ++seq, tx);
    }
    if (G.keep) hipLaunchKernelGGL(k_gj_keep, dim3(std::min(1024, cdiv(nm * nm, 1024)), 1, nd), dim3(256), 0, s, step, nd, st, dm, G.zstore.p);
  }
  const int ngm = std::max(1, G.ngmax);
  nm = std::max(1, G.nmax);
  hipLaunchKernelGGL(k_gj_final_pick, dim3(cdiv(ngm, 16), cdiv(ngm, 16), nd), dim3(256), 0, s, nd, dm, P.c_ptr.p, P.c_row.p, P.c_src.p, ig);
  hipLaunchKernelGGL(k_gj_final_scatter, dim3(8, 1, nd), dim3(256), 0, s, nd, dm, P.src.p, P.dst.p, gg);
  hipLaunchKernelGGL(k_gj_final_sym, dim3(256, 1, nd), dim3(256), 0, s, nd, dm, Sd);
  if (bI && w) {
    hipLaunchKernelGGL(k_gj_final_zg, dim3(cdiv(nm, 64), 1, nd), dim3(256), 0, s, nd, G.nsteps - 1, dm);
    hipLaunchKernelGGL(k_gj_final_w, dim3(cdiv(ngm, 256), 1, nd), dim3(256), 0, s, nd, dm, P.c_ptr.p, P.c_row.p, P.c_src.p, ig, w);
  }
  MI_HIP(hipGetLastError());
}

// Keep the level inverses of every following run (mi_schur_setup_keep_levels): storage for all Z_k (Σ n_level² doubles: 6.2 GB
// at config 3 — this is what 288 GB of HBM are for), the row form of the coupling blocks for the back-substitution, and
// new graphs (the run now ends every step with a copy of the level's inverse).
inline void gj_set_keep(mi_setup_s &P, bool on) {
  if (!P.gj) gj_build(P);
  GjState &G = *P.gj;
  if (G.keep == on) return;
  MI_HIP(hipStreamSynchronize(P.ctx->stream));
  for (auto &g : G.graph) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
  if (G.solve_graph) { (void)hipGraphExecDestroy(G.solve_graph); G.solve_graph = nullptr; }
  G.keep = on; G.have_levels = false;
  if (!on) { G.zstore.release(); G.gstore.release(); G.ustore.release(); G.sv_in.release(); G.sv_out.release(); return; }
  if (G.nmax > LV_MAX) raise(MI_ERR_BAD_ARG, "level solves: a level of %d nodes exceeds the %d the back-substitution kernel stages", G.nmax, LV_MAX);
  const int nd = P.ndom;
  long long ztot = 0;
  std::vector<int> rp, rc, rs;
  for (int d = 0; d < nd; ++d) {
    const SetupDom &D = P.dom[d];
    if (D.nlev == 0) continue;
    if (D.lev_off[D.nlev] != D.n_i) raise(MI_ERR_BAD_ARG, "level solves: subdomain %d has interior nodes that are not connected to its interface", d);
    for (int k = 0; k < D.nlev; ++k) {
      GjStep &st = G.steps_h[(size_t)(G.nsteps - 1 - k) * nd + d];
      st.zoff = ztot;
      ztot += ((long long)st.n0 * st.n0 + 31) / 32 * 32;
      st.rptr = 0;
      if (k == 0) continue;
      // rows of C_{k-1} = A_{k, k-1}: from its column form (columns = nodes of level k-1)
      const int nr = st.n0, nc = D.lev_off[k] - D.lev_off[k - 1];
      const int *cp = P.c_ptr_h.data() + D.cptr_off[k - 1];
      std::vector<std::vector<std::pair<int, int>>> rows(nr);
      for (int j = 0; j < nc; ++j)
        for (int p = cp[j]; p < cp[j + 1]; ++p) rows[P.c_row_h[p]].push_back({j, P.c_src_h[p]});
      st.rptr = (int)rp.size();
      for (int i = 0; i < nr; ++i) {
        rp.push_back((int)rc.size());
        for (auto &e : rows[i]) { rc.push_back(e.first); rs.push_back(e.second); }
      }
      rp.push_back((int)rc.size());
    }
  }
  hipStream_t s = P.ctx->stream;
  G.steps.upload(G.steps_h, s);
  G.r_ptr.upload(rp, s); G.r_col.upload(rc, s); G.r_src.upload(rs, s);
  G.zstore.alloc((size_t)ztot + 32);
  G.gstore.alloc((size_t)P.n_bi + 1); G.ustore.alloc((size_t)P.n_bi + 1);
  G.sv_in.alloc((size_t)P.n_bi + 1); G.sv_out.alloc((size_t)P.n_bi + 1);
}
inline void gj_level_enqueue(mi_setup_s &P, hipStream_t s, const double *f, double *u) {
  GjState &G = *P.gj;
  const int nd = P.ndom;
  const GjStep *st = G.steps.p;
  const GjDom *dm = G.doms.p;
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  const double *ii = G.in_ii.p;   // the values of the last run
  for (int step = 0; step < G.nsteps; ++step) {
    const int nm = G.n_step[step];
    if (step > 0) hipLaunchKernelGGL(k_lv_zg, dim3(cdiv(nm, 64), 1, nd), dim3(256), 0, s, step, nd, st, dm, G.zstore.p, G.gstore.p);
    hipLaunchKernelGGL(k_lv_g, dim3(cdiv(nm, 256), 1, nd), dim3(256), 0, s, step, nd, st, dm, P.c_ptr.p, P.c_row.p, P.c_src.p, ii, P.perm.p, f, G.gstore.p);
  }
  for (int step = G.nsteps - 1; step >= 0; --step) {
    const int nm = G.n_step[step];
    hipLaunchKernelGGL(k_lv_back, dim3(cdiv(nm, 64), 1, nd), dim3(256), 0, s, step, step == G.nsteps - 1 ? 1 : 0, nd, st, G.zstore.p, G.gstore.p,
                       G.r_ptr.p, G.r_col.p, G.r_src.p, ii, G.ustore.p);
  }
  hipLaunchKernelGGL(k_lv_scatter, dim3(std::min<long long>(4096, std::max<long long>(1, (P.n_bi + 255) / 256))), dim3(256), 0, s, (long long)P.n_bi,
                     P.perm.p, G.ustore.p, u);
  MI_HIP(hipGetLastError());
}
// u = A_II \ f for all subdomains (concatenated interior vectors in the caller's order), device pointers
inline void gj_level_solve(mi_setup_s &P, const double *f, double *u) {
  if (!P.gj || !P.gj->keep || !P.gj->have_levels) raise(MI_ERR_BAD_ARG, "level solves need mi_schur_setup_keep_levels(plan, 1) and a run after it");
  GjState &G = *P.gj;
  hipStream_t s = P.ctx->stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  if (cap != hipStreamCaptureStatusNone || env_int("MI355_SETUP_NO_GRAPH", 0)) { gj_level_enqueue(P, s, f, u); return; }   // inside somebody's capture: plain launches
  if (!G.solve_graph) {
    hipGraph_t gr = nullptr;
    MI_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try { gj_level_enqueue(P, s, G.sv_in.p, G.sv_out.p); } catch (...) { (void)hipStreamEndCapture(s, &gr); if (gr) (void)hipGraphDestroy(gr); throw; }
    MI_HIP(hipStreamEndCapture(s, &gr));
    hipError_t e = hipGraphInstantiate(&G.solve_graph, gr, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gr);
    if (e != hipSuccess) { G.solve_graph = nullptr; raise(MI_ERR_HIP, "level solves: hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
  }
  MI_HIP(hipMemcpyAsync(G.sv_in.p, f, sizeof(double) * (size_t)P.n_bi, hipMemcpyDeviceToDevice, s));
  MI_HIP(hipGraphLaunch(G.solve_graph, s));
  MI_HIP(hipMemcpyAsync(u, G.sv_out.p, sizeof(double) * (size_t)P.n_bi, hipMemcpyDeviceToDevice, s));
}

inline void gj_run(mi_setup_s &P, const double *ii_val, const double *ig_val, const double *gg_val, const double *bI, double *Sd, double *w) {
  if (!P.gj) gj_build(P);
  GjState &G = *P.gj;
  if (G.keep) G.have_levels = true;
  hipStream_t s = P.ctx->stream;
  const bool rhs = bI != nullptr && w != nullptr;
  auto cp = [&](double *dst, const double *src, long long cnt) {
    if (cnt > 0) MI_HIP(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToDevice, s));
  };
  if (G.graph_failed || env_int("MI355_SETUP_NO_GRAPH", 0)) {
    if (G.keep) cp(G.in_ii.p, ii_val, P.n_ii);   // the level solves read the coupling values from the plan's copy
    gj_enqueue(P, s, ii_val, ig_val, gg_val, rhs ? bI : nullptr, Sd, w);
    return;
  }
  hipGraphExec_t &ex = G.graph[rhs ? 1 : 0];
  if (!ex) {
    hipGraph_t gr = nullptr;
    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      try {
        gj_enqueue(P, s, G.in_ii.p, G.in_ig.p, G.in_gg.p, rhs ? G.in_bi.p : nullptr, G.out_S.p, rhs ? G.out_w.p : nullptr);
      } catch (const Error &) { ok = false; }
      if (hipStreamEndCapture(s, &gr) != hipSuccess) ok = false;
    }
    if (ok && gr) ok = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) == hipSuccess;
    if (gr) (void)hipGraphDestroy(gr);
    if (!ok) {
      (void)hipGetLastError();
      ex = nullptr; G.graph_failed = true;
      if (G.keep) cp(G.in_ii.p, ii_val, P.n_ii);
      gj_enqueue(P, s, ii_val, ig_val, gg_val, rhs ? bI : nullptr, Sd, w);
      return;
    }
  }
  cp(G.in_ii.p, ii_val, P.n_ii); cp(G.in_ig.p, ig_val, P.n_ig); cp(G.in_gg.p, gg_val, P.n_gg);
  if (rhs) cp(G.in_bi.p, bI, P.n_bi);
  MI_HIP(hipGraphLaunch(ex, s));
  cp(Sd, G.out_S.p, P.n_s);
  if (rhs) cp(w, G.out_w.p, P.n_w);
}

// ---------------------------------------------------------------- pinv of well-conditioned blocks = their inverse
// `pinv(S_d, rtol)` keeps the singular values above rtol * σ_max. If 1 / ||S^{-1}||_inf > rtol * ||S||_inf then
// σ_min >= 1 / ||S^{-1}||_2 >= 1 / ||S^{-1}||_inf > rtol ||S||_inf >= rtol σ_max: nothing is dropped and the pseudo-inverse IS the
// inverse — computed by the block Gauss-Jordan kernels above in ~1 ms instead of an eigen-decomposition (~30 ms per block
// through rocSOLVER).
// FLOATING subdomains (no Dirichlet node: pure Neumann problem, `S_d 1 = 0` — at the reference's own partitions, 80-500
// subdomains, almost every block) fail that test by construction. For a symmetric positive semi-definite S whose kernel is
// span{u}, u = 1/sqrt(n):   S^+ = (S + α u u')^{-1} - (1/α) u u'   for any α > 0 (S + α u u' has the eigenpairs of S with
// the zero eigenvalue replaced by α). With α = ||S||_inf the shifted matrix is as well conditioned as S on its range, the
// SAME kernels invert it, and the same certificate on it shows that no OTHER singular value lies below the cut-off. Taken
// when ||S 1||_inf <= rtol ||S||_inf (the constant vector is in the numerical kernel pinv would drop). Only blocks that fail
// both tests (rank deficiency > 1, or a kernel that is not the constants) go to the spectral route (rocSOLVER dsyevd).
template <bool SIGNED>
__global__ __launch_bounds__(256) void k_rowsum_max_t(int n, const double *__restrict__ A, double *__restrict__ out) {
  __shared__ double sm[NT / 64 + 1];
  double m = 0.0;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    double s2 = 0.0;
    for (int c = 0; c < n; ++c) s2 += SIGNED ? A[r + (size_t)c * n] : fabs(A[r + (size_t)c * n]);   // symmetric: column sums = row sums, coalesced this way
    s2 = fabs(s2);
    m = fmax(m, isfinite(s2) ? s2 : INFINITY);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
__global__ __launch_bounds__(256) void k_add_const(long long n, double *__restrict__ A, double c) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (long long)gridDim.x * 256) A[e] += c;
}
// dst = src + c (element-wise)
__global__ __launch_bounds__(256) void k_copy_add_const(long long n, const double *__restrict__ src, double *__restrict__ dst, double c) {
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (long long)gridDim.x * 256) dst[e] = src[e] + c;
}
inline std::atomic<long long> &spectral_pinv_calls() { static std::atomic<long long> n{0}; return n; }   // blocks sent to the eigen-decomposition (mi_ctx_query)

inline void pinv_blocks_fast(mi_ctx_s *c, int ndom, const int64_t *n_gamma_d, const double *Sd, double rtol, double *Pi) {
  hipStream_t s = c->stream;
  if (env_int("MI355_PINV_EIG", 0)) { spectral_pinv_calls() += ndom; pinv_blocks(c, ndom, n_gamma_d, Sd, rtol, Pi); return; }
  std::vector<GjStep> st(ndom);
  std::vector<GjDom> dm(ndom);
  size_t tot = 0;
  std::vector<size_t> oT(ndom), o0(ndom), o1(ndom), oP(ndom), off(ndom);
  size_t run = 0;
  for (int d = 0; d < ndom; ++d) {
    const size_t n = (size_t)n_gamma_d[d], nn = std::max<size_t>(1, n * n);
    off[d] = run; run += n * n;
    auto take = [&](size_t cnt) { const size_t o = tot; tot += (cnt + 31) / 32 * 32; return o; };
    oT[d] = take(nn); o0[d] = take(nn); o1[d] = take(nn); oP[d] = take(2 * GJ_B * GJ_B + 16);
  }
  DevBuf<double> pool(tot + 32), norms((size_t)3 * ndom * 8);
  for (int d = 0; d < ndom; ++d) {
    const int n = (int)n_gamma_d[d];
    st[d] = GjStep{}; st[d].n0 = n; st[d].nb = (n + GJ_B - 1) / GJ_B;
    dm[d] = GjDom{}; dm[d].T = pool.p + oT[d]; dm[d].Z[0] = pool.p + o0[d]; dm[d].Z[1] = pool.p + o1[d]; dm[d].P = pool.p + oP[d];
  }
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  DevBuf<GjStep> std_; DevBuf<GjDom> dmd;
  // One batch: T_d = S_d + shift_d (every entry) for the blocks `ds`, inverted together; norms of S_d (abs row sums), of the
  // inverse, and of S_d 1 (signed row sums) come back to the host.
  auto batch = [&](const std::vector<int> &ds, const std::vector<double> &shift, std::vector<double> &nS, std::vector<double> &nZ,
                   std::vector<double> &nS1) {
    std::vector<GjStep> sb; std::vector<GjDom> db;
    int nmax = 1, nbmax = 0;
    for (size_t k = 0; k < ds.size(); ++k) {
      const int d = ds[k], n = (int)n_gamma_d[d];
      sb.push_back(st[d]); db.push_back(dm[d]);
      nmax = std::max(nmax, n); nbmax = std::max(nbmax, st[d].nb);
      if (n) hipLaunchKernelGGL(k_copy_add_const, dim3(std::min(4096, cdiv(n * n, 256))), dim3(256), 0, s, (long long)n * n, Sd + off[d], dm[d].T, shift[k]);
    }
    std_.upload(sb, s); dmd.upload(db, s);
    const int nb_ = (int)ds.size();
    constexpr bool gj_head = false;   // (pinv batches: look-ahead form)
    if (nb_ > 1024) raise(MI_ERR_BAD_ARG, "pinv: more than 1024 blocks in one batch");
    hipLaunchKernelGGL(k_gj_reset, dim3(1), dim3(1024), 0, s, nb_, dmd.p);
    for (int kb = 0; kb < nbmax; ++kb) {
      if (kb == 0 && !gj_head) hipLaunchKernelGGL(k_gj_pivot, dim3(1, 1, nb_), dim3(256), 0, s, 0, kb, nb_, std_.p, dmd.p);   // later pivots: look-ahead in the update
      const int tx = cdiv(nmax, GJ_T);
      hipLaunchKernelGGL(k_gj_update<false>, dim3(tx, tx, nb_), dim3(256), 0, s, 0, kb, nb_, std_.p, dmd.p, (unsigned)(kb + 1), tx);
    }
    for (size_t k = 0; k < ds.size(); ++k) {
      const int d = ds[k], n = (int)n_gamma_d[d];
      if (!n) continue;
      hipLaunchKernelGGL(k_rowsum_max_t<false>, dim3(8), dim3(256), 0, s, n, Sd + off[d], norms.p + (size_t)24 * d);
      hipLaunchKernelGGL(k_rowsum_max_t<false>, dim3(8), dim3(256), 0, s, n, dm[d].Z[st[d].nb & 1], norms.p + (size_t)24 * d + 8);
      hipLaunchKernelGGL(k_rowsum_max_t<true>, dim3(8), dim3(256), 0, s, n, Sd + off[d], norms.p + (size_t)24 * d + 16);
    }
    MI_HIP(hipGetLastError());
    std::vector<double> nh((size_t)24 * ndom);
    MI_HIP(hipMemcpyAsync(nh.data(), norms.p, sizeof(double) * nh.size(), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    nS.assign(ndom, 0.0); nZ.assign(ndom, 0.0); nS1.assign(ndom, 0.0);
    for (int d : ds)
      for (int k = 0; k < 8; ++k) {
        nS[d] = std::max(nS[d], nh[(size_t)24 * d + k]); nZ[d] = std::max(nZ[d], nh[(size_t)24 * d + 8 + k]);
        nS1[d] = std::max(nS1[d], nh[(size_t)24 * d + 16 + k]);
      }
  };
  std::vector<int> all;
  for (int d = 0; d < ndom; ++d) all.push_back(d);
  std::vector<double> nS, nZ, nS1, zero(ndom, 0.0);
  batch(all, zero, nS, nZ, nS1);
  std::vector<int> floating, slow;
  std::vector<double> shift;
  for (int d = 0; d < ndom; ++d) {
    const int n = (int)n_gamma_d[d];
    if (!n) continue;
    if (!std::isfinite(nS[d])) raise(MI_ERR_SINGULAR, "mi_nn_pinv: block %d is not finite (mi_schur_setup_run met a singular or indefinite interior block)", d);
    const bool inv_ok = std::isfinite(nS[d]) && std::isfinite(nZ[d]) && nZ[d] > 0.0 && 1.0 / nZ[d] > rtol * nS[d];
    if (inv_ok) MI_HIP(hipMemcpyAsync(Pi + off[d], dm[d].Z[st[d].nb & 1], sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, s));
    else if (std::isfinite(nS[d]) && nS[d] > 0.0 && nS1[d] <= rtol * nS[d] && !env_int("MI355_PINV_NO_SHIFT", 0)) {
      floating.push_back(d);
      shift.push_back(nS[d] / n);                  // α u u' with α = ||S||_inf, u = 1/sqrt(n): every entry + α / n
    } else slow.push_back(d);
  }
  if (!floating.empty()) {
    MI_HIP(hipStreamSynchronize(s));               // the copies out of Z above
    std::vector<double> nS2, nZ2, nS12;
    batch(floating, shift, nS2, nZ2, nS12);
    for (size_t k = 0; k < floating.size(); ++k) {
      const int d = floating[k], n = (int)n_gamma_d[d];
      const double alpha = shift[k] * n;
      const bool ok = std::isfinite(nZ2[d]) && nZ2[d] > 0.0 && 1.0 / nZ2[d] > rtol * nS[d];   // every eigenvalue of S + α u u' above the cut-off
      if (ok) hipLaunchKernelGGL(k_copy_add_const, dim3(std::min(4096, cdiv(n * n, 256))), dim3(256), 0, s, (long long)n * n,
                                 (const double *)dm[d].Z[st[d].nb & 1], Pi + off[d], -1.0 / (alpha * n));   // - (1/α) u u'
      else slow.push_back(d);
    }
    MI_HIP(hipGetLastError());
  }
  for (int d : slow) { ++spectral_pinv_calls(); pinv_blocks(c, 1, n_gamma_d + d, Sd + off[d], rtol, Pi + off[d]); }
  MI_HIP(hipStreamSynchronize(s));   // the work buffers go out of scope
}

}  // namespace mi

inline mi_setup_s::~mi_setup_s() { release_lanes(); }
