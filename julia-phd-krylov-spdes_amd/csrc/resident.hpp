// pcg(S, b, x, ΠSnn) as ONE persistent launch with the dense blocks held on chip (cg.jl:67-109 on the operators of
// EPDD.jl:761-785 and :1361-1386).
//
// The folded loop (kernels.hpp k_gemv_pcg) re-reads S_d and ΠS_d — 2 x 67.9 MB at config 3 — from HBM / Infinity Cache
// in every iteration: that stream IS the iteration time (2 x 13.4 us). An MI355X has 256 CUs x (512 KiB of VGPRs +
// 160 KiB of LDS) = 168 MB of on-chip storage, more than the 136 MB the two operators take. This kernel therefore
// launches one 512-thread workgroup per CU that stays resident for the whole solve:
//   * every workgroup owns the same few rows of S_d and of ΠS_d of one subdomain and loads them ONCE: most of them into
//     registers (2 waves per SIMD leave 256 VGPRs per lane: ~100 doubles of matrix per lane, 400 KiB per CU), the next
//     ones into LDS, and only what does not fit (a few % at config 3, everything beyond the chip's capacity for larger
//     problems) is streamed from memory in every iteration like before;
//   * the PCG iteration is the two folded phases of k_gemv_pcg (ΠS phase: alpha, x, r, z-contributions, r'r, r'z; S phase:
//     stop rule, beta, p, Ap-contributions, p'Ap) separated by grid-wide barriers instead of kernel boundaries;
//   * r_d and p_d of the workgroup's subdomain live in LDS and are updated redundantly by every workgroup of that
//     subdomain (same inputs, same arithmetic, same bits), so the only data that crosses workgroups per phase are the
//     row results (contribution rows, <= 4 stores per row) and two or three partial sums per workgroup.
// Cross-workgroup visibility follows MI355X_MICROARCH.md (inter-workgroup visibility, sc1 hand-off): every handed-off
// byte is stored and loaded with agent-scope relaxed atomics (global_store/load ... sc1), every storing wave waits for
// its stores (s_waitcnt vmcnt(0)) before the workgroup barrier behind which ONE lane publishes the workgroup's epoch
// flag; consumers poll all flags with sc1 loads, join a workgroup barrier, then load. No L2 write-back / invalidate.
// Every spin loop is bounded: a workgroup that waits too long raises `abort`, everybody leaves, and the host falls back
// to the folded graph loop (the grid always drains).
//
// Arithmetic: row sums use the per-lane column order and the shuffle tree of k_gemv_batched (bit-identical rows);
// vector updates are the reference's mul-then-add sequence; dot products are summed per workgroup and then over
// workgroups in a fixed order (deterministic; a re-association of the oracle's left-to-right sums).
#pragma once
#include "kernels.hpp"

namespace mi {

constexpr int RES_NTH = 512, RES_WAVES = 8;
#ifndef MI355_RES_REG_DOUBLES
#define MI355_RES_REG_DOUBLES 84    // doubles of matrix per lane in VGPRs, streaming slot included (of 128 at 2 waves per SIMD)
#endif
constexpr int RES_REG_DOUBLES = MI355_RES_REG_DOUBLES;
constexpr int RES_MAX_SLOTS = 12;       // row slots per wave in registers (accumulators live at once)
constexpr int RES_LDS_BYTES = 160 * 1024 - 512;
constexpr int RES_MAX_U = 16;           // leading dimension <= 16 * 128 = GEMV_PANEL doubles
constexpr unsigned RES_SPIN_LIMIT = 1u << 22;
// code variants of the kernel: a tile's U = ceil(ld / 128) is rounded up to the next of these
__host__ __device__ constexpr int res_variant(int u) { return u <= 2 ? 2 : (u + 1) / 2 * 2; }
// LDS bytes of everything but the resident matrix rows (must match the carve-up in resident_pcg)
__host__ __device__ constexpr size_t res_lds_fixed(int U, int max_rows) {
  return (size_t)(3 * U * 128 + 2 * RES_WAVES + 4 * max_rows) * 8 + (size_t)16 * max_rows + 16;
}

// Register slots per wave (one slot = one matrix row spread over the wave's 64 lanes = 2U doubles per lane): ONE of them
// is the streaming slot, re-loaded every phase with a row that is not resident — its loads are issued right after the
// previous phase's rows have been consumed and complete during the wait for the other workgroups' partial sums —, the
// others hold resident rows of S (half, rounded down) and of ΠS.
__host__ __device__ constexpr int res_total_slots(int U) {
  return RES_REG_DOUBLES / (2 * U) > RES_MAX_SLOTS + 1 ? RES_MAX_SLOTS + 1 : (RES_REG_DOUBLES / (2 * U) < 1 ? 1 : RES_REG_DOUBLES / (2 * U));
}
__host__ __device__ constexpr int res_slots(int U) { return res_total_slots(U) - 1; }          // resident slots
__host__ __device__ constexpr int res_slots_S(int U) { return res_slots(U) / 2; }
__host__ __device__ constexpr int res_slots_P(int U) { return res_slots(U) - res_slots(U) / 2; }
// rows of one operator a workgroup holds without LDS: resident register rows + the streaming slot's rows
__host__ __device__ constexpr int res_reg_rows_S(int U) { return RES_WAVES * (res_slots_S(U) + 1); }
__host__ __device__ constexpr int res_reg_rows_P(int U) { return RES_WAVES * (res_slots_P(U) + 1); }

struct ResTile {          // one per workgroup: rows [row0, row0 + nrows) of subdomain block d, in S and in ΠS
  long long matS, matP;   // element offsets of the block in the operators' matrix buffers (row-major, ld)
  int n, ld;              // n_Γd, padded leading dimension (same in both operators)
  int loc_off;            // offset of the subdomain in the local index space
  int row0, nrows;
  int U;                  // ceil(ld / 128): code variant
  int ldsS, ldsP;         // rows kept in LDS (after the register rows) per operator; the rest is streamed
};

struct ResArgs {
  const double *MS, *MP;
  const ResTile *tiles;
  const int *gidx;        // [nloc] Γ index of every local row / column
  const double *cnt;      // [nloc] node_Γ_cnt as double
  const int *tgt;         // [nloc*W] where a row result goes in each sharing subdomain's contribution row (-1 pad)
  const int *jrank;       // [nloc] 0: this subdomain owns the Γ node
  double *conS, *conP;    // [nloc*W] contribution rows, local order
  struct ResRec *recs;    // [4*G] published partial sums {value, tag}: p'Ap | r'r | r'z | b'b
  int *abort;
  SolverState *st;
  double *res_norm;
  double *x;              // [n_Γ] in: x0, out: solution
  const double *b;        // [n_Γ]
  unsigned long long epoch0;  // records carry epochs <= epoch0 on entry
  int W, G, max_rows;
  long long *dbg;         // diagnostic (MI355_RES_DEBUG=1): wall-clock stamps of workgroup `dbg_wg`, 8 per iteration
  int dbg_wg;
};

__device__ __forceinline__ double res_ld(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void res_st(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Γ-sum of the W contribution slots of local position `loc`, in ascending subdomain order (EPDD.jl:779-781 / 1379-1381)
__device__ __forceinline__ double slot_sum_sc1(const double *con, int loc, int W) {
  double s = 0.0;
  if (W == 4) {
    const double q0 = res_ld(con + 4ll * loc), q1 = res_ld(con + 4ll * loc + 1), q2 = res_ld(con + 4ll * loc + 2),
                 q3 = res_ld(con + 4ll * loc + 3);
    s += q0; s += q1; s += q2; s += q3;
  } else {
    for (int j = 0; j < W; ++j) s += res_ld(con + (long long)loc * W + j);
  }
  return s;
}

// A value that is the same in every lane, moved to scalar registers (the loop scalars would otherwise occupy VGPRs
// that the resident matrix rows need).
__device__ __forceinline__ double res_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Cross-workgroup hand-off without a separate barrier: at the end of a phase every workgroup publishes its partial sums as
// 16-byte records {value, tag} with tag = epoch ^ mix(value); at the start of the next phase every workgroup polls all G
// records of that kind (one per thread) until every tag matches the epoch it expects — the values it then holds ARE the
// partial sums, so the wait and the reduction's loads are one memory round trip. A record is written and read with one
// 16-byte sc1 access; should such an access ever tear (value of one epoch, tag of another) the tag does not match and the
// record is simply polled again. A workgroup publishes only after all its contribution stores have completed
// (s_waitcnt vmcnt(0) in every wave, then the workgroup barrier), so fresh records imply visible contributions.
// The global dot products make every phase depend on every workgroup's previous phase, so no workgroup can run a full
// phase ahead of another: buffers are never overwritten before their readers are done (no double buffering needed).
struct alignas(16) ResRec {
  double v;
  unsigned long long tag;
};
typedef unsigned int res_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long res_tag(double v, unsigned long long epoch) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return epoch ^ (b * 0x9E3779B97F4A7C15ull) ^ (b >> 29);
}
__device__ __forceinline__ void res_publish(ResRec *p, double v, unsigned long long epoch) {
  const unsigned long long t = res_tag(v, epoch), b = (unsigned long long)__double_as_longlong(v);
  res_u4 d;
  d.x = (unsigned)b; d.y = (unsigned)(b >> 32); d.z = (unsigned)t; d.w = (unsigned)(t >> 32);
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ bool res_fetch(const ResRec *p, unsigned long long epoch, double &v) {
  res_u4 d;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(p) : "memory");
  v = __longlong_as_double((long long)(((unsigned long long)d.y << 32) | d.x));
  return ((((unsigned long long)d.w << 32) | d.z)) == res_tag(v, epoch);
}
// Poll NA record arrays (G records each, G <= RES_NTH) for `epoch`; on success sum[k] = Σ over workgroups in a fixed
// order, identical in every workgroup. false: the solve was aborted (bounded wait expired somewhere).
template <int NA>
__device__ __forceinline__ bool res_wait_sums(const ResArgs &a, const ResRec *base, int first, unsigned long long epoch,
                                              double (&sum)[NA], double *red) {
  double v[NA];
  const bool mine = (int)threadIdx.x < a.G;
  for (unsigned spin = 0;; ++spin) {
    int stale = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      v[k] = 0.0;
      if (mine) stale |= !res_fetch(base + (size_t)(first + k) * a.G + threadIdx.x, epoch, v[k]);
    }
    if (!__syncthreads_or(stale)) break;
    if ((spin & 63u) == 63u) {
      int ab = 0;
      if (threadIdx.x == 0) {
        ab = __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ab && spin >= RES_SPIN_LIMIT) { __hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ab = 1; }
      }
      if (__syncthreads_or(ab)) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (NA >= 2) {
    block_sum2_t<RES_NTH>(v[0], v[1], red);
    __syncthreads();
  } else {
    v[0] = block_sum_t<RES_NTH>(v[0], red);
  }
  if (NA >= 3) v[2] = block_sum_t<RES_NTH>(v[2], red);
#pragma unroll
  for (int k = 0; k < NA; ++k) sum[k] = res_uniform(v[k]);
  return true;
}

template <int U>
__device__ __forceinline__ void resident_pcg(const ResArgs &a, const ResTile &t, double *lds) {
  constexpr int RRS = res_slots_S(U), RRP = res_slots_P(U);
  constexpr int LDW = U * 128;                                  // padded row width
  constexpr int CPT = (LDW + RES_NTH - 1) / RES_NTH;            // columns per thread in the vector work
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave id: scalar (row bases stay in SGPRs)
  const int n = t.n, ld = t.ld, off = t.loc_off, W = a.W, G = a.G, nrows = t.nrows, row0 = t.row0;
  SolverState *st = a.st;
  // ---- LDS carve-up
  double *xs = lds, *rd = xs + LDW, *pd = rd + LDW, *red = pd + LDW;            // operand, r_d, p_d, reduction scratch
  double *rowv = red + 2 * RES_WAVES, *rowc0 = rowv + a.max_rows, *rowc1 = rowc0 + a.max_rows, *rcnt = rowc1 + a.max_rows;
  int *rtgt = reinterpret_cast<int *>(rcnt + a.max_rows);                        // [max_rows*4]
  double *ldsS = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(rtgt + 4 * a.max_rows) + 15) & ~(uintptr_t)15);
  double *ldsP = ldsS + (size_t)t.ldsS * LDW;

  // ---- residents: registers first (issued now, consumed at first use), then LDS
  double2 mS[RRS > 0 ? RRS : 1][U], mP[RRP > 0 ? RRP : 1][U];
  const double *MS = a.MS + t.matS, *MP = a.MP + t.matP;
  const int rmax = n > 0 ? n - 1 : 0;   // (clamped addresses + select instead of predicated loads: no branch per element)
  double2 mT[U];   // the streaming slot: row RRo*8 + w of the operator of the NEXT phase
#define RES_LOAD_STREAM(MGLOB, RRO)                                                              \
  do {                                                                                           \
    const int i_ = (RRO) * RES_WAVES + w;                                                        \
    /* no masking, no clamping of columns: a group past the row's end reads the next row or the zeroed tail of the  \
       buffer and meets a zero of the operand; a clamped row's result is dropped. Scalar row base + lane offset. */   \
    const double2 *rowp_ = reinterpret_cast<const double2 *>((MGLOB) + (long long)min(row0 + i_, rmax) * ld);        \
    _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) mT[u_] = rowp_[u_ * 64 + lane];                                 \
  } while (0)
#pragma unroll
  for (int s = 0; s < RRP; ++s) {
    const int i = s * RES_WAVES + w;
    const double2 *rowp = reinterpret_cast<const double2 *>(MP + (long long)min(row0 + i, rmax) * ld);
#pragma unroll
    for (int u = 0; u < U; ++u) mP[s][u] = rowp[u * 64 + lane];   // (unmasked, unclamped columns: see RES_LOAD_STREAM)
  }
#pragma unroll
  for (int s = 0; s < RRS; ++s) {
    const int i = s * RES_WAVES + w;
    const double2 *rowp = reinterpret_cast<const double2 *>(MS + (long long)min(row0 + i, rmax) * ld);
#pragma unroll
    for (int u = 0; u < U; ++u) mS[s][u] = rowp[u * 64 + lane];   // (unmasked, unclamped columns: see RES_LOAD_STREAM)
  }
  for (int k = w; k < t.ldsP; k += RES_WAVES) {
    const double *rowp = MP + (long long)(row0 + (RRP + 1) * RES_WAVES + k) * ld;
#pragma unroll 2
    for (int u = 0; u < U; ++u) {
      const int c = u * 128 + lane * 2;
      const double2 v = *reinterpret_cast<const double2 *>(rowp + min(c, ld - 2));
      *reinterpret_cast<double2 *>(&ldsP[(size_t)k * LDW + c]) = c < ld ? v : make_double2(0.0, 0.0);
    }
  }
  for (int k = w; k < t.ldsS; k += RES_WAVES) {
    const double *rowp = MS + (long long)(row0 + (RRS + 1) * RES_WAVES + k) * ld;
#pragma unroll 2
    for (int u = 0; u < U; ++u) {
      const int c = u * 128 + lane * 2;
      const double2 v = *reinterpret_cast<const double2 *>(rowp + min(c, ld - 2));
      *reinterpret_cast<double2 *>(&ldsS[(size_t)k * LDW + c]) = c < ld ? v : make_double2(0.0, 0.0);
    }
  }
  // per-row constants of this workgroup's rows: contribution targets and 1/cnt divisor
  for (int i = threadIdx.x; i < nrows; i += RES_NTH) {
    const int loc = off + row0 + i;
    rcnt[i] = a.cnt[loc];
    for (int q = 0; q < 4; ++q) rtgt[4 * i + q] = q < W ? a.tgt[loc * W + q] : -1;
  }
  // The thread whose column j is one of this workgroup's rows serves that row's vector entry (at most one column per
  // thread: the rows are consecutive and fewer than the threads); the owner of the Γ node also carries x[g].
  int my_q = -1, my_ri = -1, my_g = 0;
  bool my_own = false;
  double ox = 0.0;
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int j = q * RES_NTH + threadIdx.x, ri = j - row0;
    if (j < n && ri >= 0 && ri < nrows) {
      my_q = q; my_ri = ri;
      my_own = a.jrank[off + j] == 0;
      my_g = a.gidx[off + j];
      if (my_own) ox = a.x[my_g];
    }
  }
  const double eps = st->eps;
  const long long maxit = st->maxit, cap = st->res_cap;
  const bool x0_zero = st->x0_zero != 0;
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  enum { REC_PAP = 0, REC_RR = 1, REC_RZ = 2, REC_BB = 3 };
  unsigned long long epoch = a.epoch0;   // epoch of the phase whose records are published next

  // y = M_o * xs for this workgroup's rows: register rows, LDS rows, streamed rows. SCALE: ΠS (result / cnt).
  // Row result -> contribution rows of the sharing subdomains (sc1), rowc1[i] = rowv[i] * y for the next dot product.
#define RES_EMIT(I, SUM, CON, SCALE)                                   \
  do {                                                                 \
    const int ie_ = (I);                                               \
    if (lane == 0 && ie_ < nrows) {                                    \
      const double y_ = (SCALE) ? (SUM) / rcnt[ie_] : (SUM);           \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {               \
        const int tg_ = rtgt[4 * ie_ + q_];                            \
        if (tg_ >= 0) res_st((CON) + tg_, y_);                         \
      }                                                                \
      rowc1[ie_] = rowv[ie_] * y_;                                     \
    }                                                                  \
  } while (0)
#define RES_GEMV(MREG, RRO, MGLOB, LDSROWS, NLDS, CON, SCALE)                                                        \
  do {                                                                                                               \
    double acc_[RRO > 0 ? RRO : 1], acct_ = 0.0;                                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < RRO; ++s_) acc_[s_] = 0.0;                                               \
    _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) {                                                               \
      const double2 xv_ = *reinterpret_cast<const double2 *>(&xs[u_ * 128 + lane * 2]);                              \
      _Pragma("unroll") for (int s_ = 0; s_ < RRO; ++s_) {                                                           \
        acc_[s_] += MREG[s_][u_].x * xv_.x;                                                                          \
        acc_[s_] += MREG[s_][u_].y * xv_.y;                                                                          \
      }                                                                                                              \
      acct_ += mT[u_].x * xv_.x;                                                                                     \
      acct_ += mT[u_].y * xv_.y;                                                                                     \
      __builtin_amdgcn_sched_barrier(0); /* one operand pair in flight at a time: the registers hold the matrix */   \
    }                                                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < RRO; ++s_) {                                                             \
      const double sum_ = wave_sum(acc_[s_]);                                                                        \
      RES_EMIT(s_ * RES_WAVES + w, sum_, CON, SCALE);                                                                \
    }                                                                                                                \
    {                                                                                                                \
      const double sum_ = wave_sum(acct_);                                                                           \
      RES_EMIT(RRO * RES_WAVES + w, sum_, CON, SCALE);                                                               \
    }                                                                                                                \
    for (int k_ = w; k_ < (NLDS); k_ += RES_WAVES) {                                                                 \
      asm volatile("" ::: "memory"); /* the operand is re-read per row: hoisted out of the loop it would pin 4U VGPRs */ \
      double a_ = 0.0;                                                                                               \
      const double *row_ = (LDSROWS) + (size_t)k_ * LDW;                                                             \
      _Pragma("unroll 2") for (int u_ = 0; u_ < U; ++u_) {                                                           \
        const int c_ = u_ * 128 + lane * 2;                                                                          \
        const double2 mv_ = *reinterpret_cast<const double2 *>(&row_[c_]);                                           \
        const double2 xv_ = *reinterpret_cast<const double2 *>(&xs[c_]);                                             \
        a_ += mv_.x * xv_.x;                                                                                         \
        a_ += mv_.y * xv_.y;                                                                                         \
      }                                                                                                              \
      const double sum_ = wave_sum(a_);                                                                              \
      RES_EMIT((RRO + 1) * RES_WAVES + k_, sum_, CON, SCALE);                                                        \
    }                                                                                                                \
    for (int i_ = (RRO + 1) * RES_WAVES + (NLDS) + w; i_ < nrows; i_ += RES_WAVES) {   /* does not fit on chip */    \
      asm volatile("" ::: "memory");                                                                                 \
      double a_ = 0.0;                                                                                               \
      const double *row_ = (MGLOB) + (long long)(row0 + i_) * ld;                                                    \
      _Pragma("unroll 1") for (int u0_ = 0; u0_ < U; u0_ += 2) {                                                     \
        double2 mv_[2];                                                                                              \
        _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {                                                           \
          const int c_ = (u0_ + k_) * 128 + lane * 2;                                                                \
          const double2 ld_ = *reinterpret_cast<const double2 *>(row_ + min(c_, ld - 2));                            \
          const bool ok_ = u0_ + k_ < U && c_ < ld;                                                                  \
          mv_[k_] = make_double2(ok_ ? ld_.x : 0.0, ok_ ? ld_.y : 0.0);                                              \
        }                                                                                                            \
        _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {                                                           \
          if (u0_ + k_ < U) {                                                                                        \
            const double2 xv_ = *reinterpret_cast<const double2 *>(&xs[(u0_ + k_) * 128 + lane * 2]);               \
            a_ += mv_[k_].x * xv_.x;                                                                                 \
            a_ += mv_[k_].y * xv_.y;                                                                                 \
          }                                                                                                          \
        }                                                                                                            \
      }                                                                                                              \
      const double sum_ = wave_sum(a_);                                                                              \
      RES_EMIT(i_, sum_, CON, SCALE);                                                                                \
    }                                                                                                                \
  } while (0)
  // per-workgroup partials of the next dot products: rows' rowc1 (and rowc0) summed by the first wave, stored sc1
  // end of a phase: every wave waits for its contribution stores, then the first wave sums the rows' terms of the next dot
  // products and lane 0 publishes them under the phase's epoch (fresh records imply visible contributions)
#define RES_PARTIALS(K1, HAS0, K0, HAS2, K2, V2)                                                    \
  do {                                                                                              \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    __syncthreads();                                                                                \
    if (threadIdx.x < 64) {                                                                         \
      double a1_ = 0.0, a0_ = 0.0;                                                                  \
      for (int i_ = threadIdx.x; i_ < nrows; i_ += 64) { a1_ += rowc1[i_]; a0_ += rowc0[i_]; }      \
      a1_ = wave_sum(a1_);                                                                          \
      a0_ = wave_sum(a0_);                                                                          \
      if (threadIdx.x == 0) {                                                                       \
        ++epoch_pub;                                                                                \
        res_publish(a.recs + (size_t)(K1) * G + blockIdx.x, a1_, epoch_pub);                        \
        if (HAS0) res_publish(a.recs + (size_t)(K0) * G + blockIdx.x, a0_, epoch_pub);              \
        if (HAS2) res_publish(a.recs + (size_t)(K2) * G + blockIdx.x, (V2), epoch_pub);             \
      }                                                                                             \
    }                                                                                               \
  } while (0)
  unsigned long long epoch_pub = a.epoch0;   // (only thread 0's copy is used)

  int dbg_n = 0;
#define RES_STAMP()                                                                                   \
  do {                                                                                                \
    if (a.dbg && (int)blockIdx.x == a.dbg_wg && threadIdx.x == 0 && dbg_n < 512) a.dbg[dbg_n++] = wall_clock64(); \
  } while (0)
  RES_STAMP();
  __syncthreads();  // rcnt, rtgt, LDS rows
  RES_STAMP();
  // ---- r_0 = b - S x_0 (cg.jl:83). x_0 = 0: r_0 = b without touching S.
  if (!x0_zero) {
#pragma unroll 1
    for (int q = 0; q < CPT; ++q) {
      const int j = q * RES_NTH + threadIdx.x;
      if (j < LDW) xs[j] = j < n ? a.x[a.gidx[off + j]] : 0.0;
      if (q == my_q) { rowv[my_ri] = 0.0; rowc0[my_ri] = 0.0; }
    }
    RES_LOAD_STREAM(MS, RRS);
    __syncthreads();
    RES_GEMV(mS, RRS, MS, ldsS, t.ldsS, a.conS, false);
    RES_PARTIALS(REC_PAP, false, REC_RR, false, REC_BB, 0.0);           // (the value is not used: the records are the barrier)
    double dummy[1];
    if (!res_wait_sums<1>(a, a.recs, REC_PAP, ++epoch, dummy, red)) return;
  }
  // ---- set-up: r_0 into LDS; first ΠS phase produces z_0 contributions, r_0'r_0, r_0'z_0, b'b
  RES_LOAD_STREAM(MP, RRP);
  double bb_loc = 0.0;
#pragma unroll 1
  for (int q = 0; q < CPT; ++q) {
    const int j = q * RES_NTH + threadIdx.x;
    if (j < LDW) {
      double r = 0.0, vs = 0.0;
      if (j < n) {
        const int loc = off + j;
        const double bg = a.b[a.gidx[loc]];
        r = bg;
        if (!x0_zero) {
          double cs = 0.0;
          for (int k = 0; k < W; ++k) cs += res_ld(a.conS + (long long)loc * W + k);
          r = bg - cs;
        }
        vs = r / a.cnt[loc];
        if (q == my_q) {
          rowv[my_ri] = r;
          rowc0[my_ri] = my_own ? r * r : 0.0;
          if (my_own) bb_loc += bg * bg;
        }
      }
      rd[j] = r; pd[j] = 0.0; xs[j] = vs;
    }
  }
  const double bb_wg = res_uniform(block_sum_t<RES_NTH>(bb_loc, red));   // (barriers inside: xs, rd, rowv visible afterwards)
  RES_STAMP();
  RES_GEMV(mP, RRP, MP, ldsP, t.ldsP, a.conP, true);
  RES_STAMP();
  RES_PARTIALS(REC_RZ, true, REC_RR, true, REC_BB, bb_wg);
  ++epoch;
  RES_LOAD_STREAM(MS, RRS);   // in flight while the other workgroups' records are awaited
  RES_STAMP();

  long long it = 0;
  double rTz_prev = 1.0, tol = 0.0;
  bool overflow = false;
  for (;;) {
    // ================= S phase: stop rule, beta, p = beta p + z, Ap contributions, p'Ap   (cg.jl:91, 102-106, 93-94)
    double rr, rz;
    if (it == 0) {
      double sm3[3];
      if (!res_wait_sums<3>(a, a.recs, REC_RR, epoch, sm3, red)) return;   // r'r, r'z, b'b of the set-up phase
      rr = sm3[0]; rz = sm3[1];
      tol = eps * sqrt(sm3[2]);
      if (lead) { st->bnorm = sqrt(sm3[2]); st->tol = tol; }
    } else {
      double sm2[2];
      if (!res_wait_sums<2>(a, a.recs, REC_RR, epoch, sm2, red)) return;
      rr = sm2[0]; rz = sm2[1];
    }
    RES_STAMP();   // [0] partial sums in
    const long long it_new = it + 1;
    const double res = sqrt(rr);
    const bool stop = !((it_new < maxit) && (res > tol));
    double beta = 1. / rTz_prev;
    beta *= rz;
    if (it_new > cap) overflow = true;
    if (lead) {
      st->rTr = rr; st->rTz = rz; st->beta = beta;
      if (it_new <= cap) a.res_norm[it_new - 1] = res;
    }
    it = it_new;
    if (stop || overflow) break;
#pragma unroll 1
    for (int q = 0; q < CPT; ++q) {
      const int j = q * RES_NTH + threadIdx.x;
      if (j < n) {
        const int loc = off + j;
        const double z = slot_sum_sc1(a.conP, loc, W);
        const double p = beta * pd[j] + z;                     // axpby!(1, z, beta, p)
        pd[j] = p; xs[j] = p;
        if (q == my_q) rowv[my_ri] = p;
      }
    }
    __syncthreads();
    RES_STAMP();   // [1] operand staged
    RES_GEMV(mS, RRS, MS, ldsS, t.ldsS, a.conS, false);
    RES_STAMP();   // [2] S rows done
    RES_PARTIALS(REC_PAP, false, REC_RR, false, REC_BB, 0.0);
    ++epoch;
    RES_LOAD_STREAM(MP, RRP);
    RES_STAMP();   // [3] published
    // ================= ΠS phase: alpha, x += alpha p, r -= alpha Ap, z contributions, r'r, r'z   (cg.jl:94-101)
    double sd[1];
    if (!res_wait_sums<1>(a, a.recs, REC_PAP, epoch, sd, red)) return;
    const double d = sd[0];
    const double alpha = rz / d;
    RES_STAMP();   // [4] partial sum in
    if (lead) { st->d = d; st->alpha = alpha; st->rTz_prev = rz; }
    rTz_prev = rz;
#pragma unroll 1
    for (int q = 0; q < CPT; ++q) {
      const int j = q * RES_NTH + threadIdx.x;
      if (j < n) {
        const int loc = off + j;
        const double Ap = slot_sum_sc1(a.conS, loc, W);
        const double r = rd[j] + (-alpha) * Ap;                // axpy!(-alpha, Ap, r)
        rd[j] = r; xs[j] = r / a.cnt[loc];
        if (q == my_q) {
          rowv[my_ri] = r;
          rowc0[my_ri] = my_own ? r * r : 0.0;
          if (my_own) ox = ox + alpha * pd[j];                 // axpy!(alpha, p, x)
        }
      }
    }
    __syncthreads();
    RES_STAMP();   // [5] operand staged
    RES_GEMV(mP, RRP, MP, ldsP, t.ldsP, a.conP, true);
    RES_STAMP();   // [6] ΠS rows done
    RES_PARTIALS(REC_RZ, true, REC_RR, false, REC_BB, 0.0);
    ++epoch;
    RES_LOAD_STREAM(MS, RRS);
    RES_STAMP();   // [7] published
  }
#undef RES_STAMP
#undef RES_LOAD_STREAM
#undef RES_GEMV
#undef RES_EMIT
#undef RES_PARTIALS
  if (my_q >= 0 && my_own) a.x[my_g] = ox;
  if (lead) {
    st->it = it; st->it_nxt = it;
    st->overflow = overflow ? 1 : 0;
    st->done = 1;
  }
}

__global__ __launch_bounds__(RES_NTH) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pcg_resident(ResArgs a) {
  extern __shared__ __attribute__((aligned(16))) double res_lds[];
  const ResTile t = a.tiles[blockIdx.x];
  switch (t.U) {   // code variant = padded row width / 128 (the host rounds up to the next one that exists)
#define RES_CASE(UU) case UU: resident_pcg<UU>(a, t, res_lds); break;
#ifdef MI355_RES_ONLY_U   // build-time probe of one variant's register allocation
    RES_CASE(MI355_RES_ONLY_U)
#else
    RES_CASE(2) RES_CASE(4) RES_CASE(6) RES_CASE(8) RES_CASE(10) RES_CASE(12) RES_CASE(14) RES_CASE(16)
#endif
#undef RES_CASE
    default: break;  // never: the host only emits the variants above
  }
}

}  // namespace mi
