// Row streamer of the dense S_d / ΠS_d applies as an LDS-DMA ring (gfx950).
//
// A tile is a run of consecutive rows of one block: a contiguous byte range of the padded row-major matrix. It is cut
// into UNITS of one wave-instruction each: unit (row, piece) = 128 consecutive doubles (1 KiB) of a row, moved by ONE
// `global_load_lds_dwordx4` straight into LDS (no VGPR destination). The first SW waves of a workgroup are STREAM waves:
// wave w owns the contiguous unit range [U*w/SW, U*(w+1)/SW) of the tile (balanced to one unit whatever the number of
// rows) and a private ring of D KiB-slots, keeps D pieces in flight from the first instruction of the launch to its last
// unit (counted `s_waitcnt vmcnt`, no cross-wave hand-shake, no barrier in the loop), multiplies a landed piece with the
// operand panel (two `ds_read_b128`, two FMAs per lane) and re-issues the slot. The remaining waves are VECTOR waves:
// they do everything that needs ordinary loads (operand gather, and in the folded PCG launches the whole prologue) while
// the stream waves' DMAs are in flight — a wave's vector-memory results return in issue order, so a wave that has DMAs
// outstanding must not wait for an ordinary load, and a wave that waits for ordinary loads must not issue DMAs.
// Row sums: a wave flushes its lane sums (shuffle tree) whenever its units leave a row, into part[wave][k]; after the
// closing barrier thread r adds the partials of row r in wave order — a fixed order, no atomics.
#pragma once
#include <hip/hip_runtime.h>

namespace mi {

#ifdef MI355_RING_LAB
__device__ long long *g_ring_dbg = nullptr;   // lab: 8 wall-clock stamps per workgroup (wave 0, lane 0)
#define RING_STAMP(i) do { if (g_ring_dbg && threadIdx.x == 0) g_ring_dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define RING_STAMP(i) do { } while (0)
#endif

constexpr int RING_PIECE = 128;       // doubles per unit (64 lanes x 16 B)
constexpr int RING_XS = 2048 + 128;   // operand panel incl. the zero tail of the last piece
constexpr int RING_MAXROWS = 64;      // rows per tile (host guarantees)

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)p; }  // LDS byte address of a __shared__ object

// one LDS-DMA wave-instruction: lane l's 16 bytes at gsrc -> LDS[lds_dst + 16*l]. M0 is written in the same statement
// that reads it and restored (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same behind `s_waitcnt lgkmcnt(0)`: the slot is re-used, this wave's reads of it must have returned
__device__ __forceinline__ void glds16_reuse(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// s_waitcnt vmcnt(k) for a wave-uniform run-time k (the field is an immediate)
__device__ __forceinline__ void wait_vmcnt(int k) {
  switch (k) {
#define MI_VM(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
    MI_VM(0) MI_VM(1) MI_VM(2) MI_VM(3) MI_VM(4) MI_VM(5) MI_VM(6) MI_VM(7) MI_VM(8) MI_VM(9) MI_VM(10) MI_VM(11)
    MI_VM(12) MI_VM(13) MI_VM(14) MI_VM(15) MI_VM(16) MI_VM(17) MI_VM(18) MI_VM(19) MI_VM(20) MI_VM(21) MI_VM(22) MI_VM(23)
    MI_VM(24) MI_VM(25) MI_VM(26) MI_VM(27) MI_VM(28) MI_VM(29) MI_VM(30) MI_VM(31)
#undef MI_VM
    default: break;  // k >= 32: nothing to wait for yet (D <= 32)
  }
}

// first unit of wave w of nw: U*w/nw in 32-bit unsigned arithmetic (U <= 64 rows x 17 pieces, nw <= 16; a 64-bit
// division is a few hundred instructions on this target)
__device__ __forceinline__ int ring_split(int U, int w, int nw) { return (int)((unsigned)U * (unsigned)w / (unsigned)nw); }

// The stream of one wave. `begin` sets the unit range, `fill` puts the first pieces in flight (D of them before `run`),
// `run` consumes all units against the staged operand.
template <int D>
struct RingStream {
  const double *Md;      // block base + row0*ld
  int ld, ppr;           // leading dimension, pieces per row
  int u0, u1;            // unit range of this wave
  int issued;            // units issued so far (relative to u0)
  int ir, ip;            // row / piece of the next unit to issue
  unsigned ring;         // LDS byte address of this wave's ring
  int lane2;
  __device__ __forceinline__ const double *src(int r, int p) const {
    return Md + (long long)r * ld + min(p * RING_PIECE + lane2, ld - 2);  // lanes past the row end re-read its last 16 bytes (operand 0 there)
  }
  __device__ __forceinline__ void begin(const double *Mtile, int ld_, int nrows, int w, int sw, unsigned ring_addr) {
    Md = Mtile; ld = ld_;
    ppr = (ld_ + RING_PIECE - 1) / RING_PIECE;
    const int U = nrows * ppr;
    u0 = ring_split(U, w, sw);
    u1 = ring_split(U, w + 1, sw);
    ring = ring_addr;
    lane2 = (threadIdx.x & 63) * 2;
    ir = (int)((unsigned)u0 / (unsigned)ppr); ip = u0 - ir * ppr;
    issued = 0;
  }
  // issue pieces until `upto` are in flight (or the wave's units are exhausted); only before `run`
  __device__ __forceinline__ void fill(int upto) {
    const int total = u1 - u0;
#pragma unroll 1
    for (; issued < upto && issued < total; ++issued) {
      glds16(src(ir, ip), ring + (unsigned)issued * 1024u);
      if (++ip == ppr) { ip = 0; ++ir; }
    }
  }
  // part[k] receives the sum of this wave's units in its k-th row (k = row - first row of the wave); xs = operand panel.
  // CONSUME = false: the stream without the arithmetic (DMA floor measurement).
  template <bool CONSUME>
  __device__ __forceinline__ void run(const double *xs, const double *ring_p, double *part) {
    const int total = u1 - u0;
    int cr = (int)((unsigned)u0 / (unsigned)ppr), cp = u0 - cr * ppr;
    const int r_first = cr;
    double acc = 0.0;
    int slot = 0, i = 0;
    auto consume = [&]() {
      if (CONSUME) {
        const double2 mv = *reinterpret_cast<const double2 *>(ring_p + slot * RING_PIECE + lane2);
        const double2 xv = *reinterpret_cast<const double2 *>(xs + cp * RING_PIECE + lane2);
        acc += mv.x * xv.x;
        acc += mv.y * xv.y;
      }
    };
    auto leave = [&]() {
      if (++slot == D) slot = 0;
      ++i;
      if (++cp == ppr || i == total) {   // leaving the row: flush
        const double s = wave_sum(acc);
        if (lane2 == 0) part[cr - r_first] = s;
        acc = 0.0; cp = 0; ++cr;
      }
    };
    // steady state: D pieces outstanding, the oldest is the one consumed; its slot is re-issued at once
#pragma unroll 1
    while (issued < total) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(D - 1) : "memory");
      consume();
      glds16_reuse(src(ir, ip), ring + (unsigned)slot * 1024u);
      ++issued;
      if (++ip == ppr) { ip = 0; ++ir; }
      leave();
    }
    // tail: nothing left to issue, the count of younger pieces shrinks
#pragma unroll 1
    while (i < total) {
      wait_vmcnt(total - i - 1);
      consume();
      leave();
    }
  }
};

// rows per wave bound: a wave's ceil(U/SW) units touch at most that many / ppr + 2 rows
template <int SW> constexpr int ring_max_rpw() { return RING_MAXROWS / SW + 2; }

// Sum of the partials of tile row r over the stream waves, in wave order. wrow[2w], wrow[2w+1] = first and last row of
// wave w's units (last < first: the wave has none), left in LDS by the waves themselves (no divisions here: an integer
// division by a run-time value is ~40 instructions, and this loop ran 3 of them per wave on one wave per workgroup).
template <int SW>
__device__ __forceinline__ double ring_row_sum(const double *part, const int *wrow, int r) {
  constexpr int MRW = ring_max_rpw<SW>();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < SW; ++w) {
    const int rf = wrow[2 * w], rl = wrow[2 * w + 1];
    if (r >= rf && r <= rl) s += part[w * MRW + (r - rf)];
  }
  return s;
}

// Plain S-apply / NN-apply GEMV (`apply_local_schurs` EPDD.jl:761-785, `apply_neumann_neumann_schur` :1361-1386):
// the ring form of k_gemv_batched. mode 1 = stream without consuming (measurement only).
template <int WAVES, int SW, int D, bool SCALE>
__global__ __launch_bounds__(64 * WAVES) void k_gemv_ring(DenseMeta m, const double *__restrict__ x, double *__restrict__ yslots,
                                                          const int *done, const int *zero_x, int mode, int thin) {
  constexpr int VW = WAVES - SW, NV = 64 * VW, MRW = ring_max_rpw<SW>();
  static_assert(VW >= 1 && D <= 32, "need a vector wave; vmcnt switch covers 32 slots");
  const GemvTile t = m.tiles[blockIdx.x];
  const int done0 = done ? *done : 0, zero0 = zero_x ? *zero_x : 0;
  asm volatile("" ::"s"(t.mat_off), "s"(t.n), "s"(t.ld), "s"(t.loc_off), "s"(t.row0), "s"(t.active), "s"(t.nrows), "s"(done0), "s"(zero0));
  if (done0 || !t.active) return;
  RING_STAMP(0);
  __shared__ __attribute__((aligned(1024))) double ring[SW * D * RING_PIECE];
  __shared__ __attribute__((aligned(16))) double xs[RING_XS];
  __shared__ double part[SW * MRW];
  __shared__ int wrow[2 * SW];
  const int off = t.loc_off, n = t.n;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (zero0) {
    if ((int)threadIdx.x < t.nrows) yslots[m.out_pos[off + t.row0 + threadIdx.x]] = 0.0;
    return;
  }
  const int ppr = (t.ld + RING_PIECE - 1) / RING_PIECE;
  if (w < SW) {
    // The operand gather of the vector waves is two dependent loads (index, value). Loads issued behind a full ring wait
    // for 128 KB of matrix requests in the CU's queue, so the stream starts thin and is topped up once the vector waves
    // have issued their second hop (barrier B): the ring then covers the rest of the gather and the staging.
    RingStream<D> rs;
    rs.begin(m.M + t.mat_off + (long long)t.row0 * t.ld, t.ld, t.nrows, w, SW, lds_addr(ring) + (unsigned)w * (D * 1024u));
    if ((threadIdx.x & 63) == 0) {
      wrow[2 * w] = (int)((unsigned)rs.u0 / (unsigned)rs.ppr);
      wrow[2 * w + 1] = rs.u1 > rs.u0 ? (int)((unsigned)(rs.u1 - 1) / (unsigned)rs.ppr) : -1;
    }
    __builtin_amdgcn_s_barrier();   // A: index loads issued
    rs.fill(thin);
    __builtin_amdgcn_s_barrier();   // B: value loads issued
    rs.fill(D);
    RING_STAMP(1);
    __builtin_amdgcn_s_barrier();   // C: operand staged (this wave's DMAs stay in flight)
    RING_STAMP(2);
    if (mode == 0) rs.template run<true>(xs, ring + w * (D * RING_PIECE), part + w * MRW);
    else rs.template run<false>(xs, ring + w * (D * RING_PIECE), part + w * MRW);
    RING_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RING_STAMP(4);
  } else {
    const int vt = (int)threadIdx.x - 64 * SW;
    const int plen = ppr * RING_PIECE;
    constexpr int XPT = (RING_XS + NV - 1) / NV;
    int gi[XPT];
    double xv[XPT], cn[XPT];
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      const int j = q * NV + vt;
      gi[q] = j < n ? m.gidx[off + j] : -1;
      cn[q] = SCALE && j < n ? m.cnt[off + j] : 1.0;
    }
    __builtin_amdgcn_s_barrier();   // A
#pragma unroll
    for (int q = 0; q < XPT; ++q) xv[q] = gi[q] >= 0 ? x[gi[q]] : 0.0;
    __builtin_amdgcn_s_barrier();   // B
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
      const int j = q * NV + vt;
      if (j < plen) xs[j] = SCALE ? xv[q] / cn[q] : xv[q];
    }
    __syncthreads();                // C
    __syncthreads();
  }
  // every wave: rows of the tile, one thread each
  if ((int)threadIdx.x < t.nrows) {
    const int r = t.row0 + threadIdx.x;
    const double s = ring_row_sum<SW>(part, wrow, threadIdx.x);
    yslots[m.out_pos[off + r]] = SCALE ? s / m.cnt[off + r] : s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same unit stream through REGISTERS, software-pipelined: every wave keeps NB one-KiB loads in flight (NB x 4 VGPRs),
// consumes the oldest and re-issues its buffer at once. All code is the compiler's (it counts its own loads: the waits
// come out as vmcnt(NB-1)); units past the end of the wave's range load one broadcast address (a single 64-byte request)
// so that the loop stays branch-free around the loads and the count exact.
template <int NB>
struct PipeStream {
  const double *Md;
  int ld, ppr, u0, total, lane2;
  double2 buf[NB];
  int ir, ip;   // row / piece of the next unit to issue
  int nissued;
  __device__ __forceinline__ const double *src_next() {
    const double *p = nissued < total ? Md + (long long)ir * ld + min(ip * RING_PIECE + lane2, ld - 2) : Md;
    ++nissued;
    if (++ip == ppr) { ip = 0; ++ir; }
    return p;
  }
  __device__ __forceinline__ void begin(const double *Mtile, int ld_, int nrows, int w, int nw) {
    Md = Mtile; ld = ld_;
    ppr = (ld_ + RING_PIECE - 1) / RING_PIECE;
    const int U = nrows * ppr;
    u0 = ring_split(U, w, nw);
    total = ring_split(U, w + 1, nw) - u0;
    lane2 = (threadIdx.x & 63) * 2;
    ir = (int)((unsigned)u0 / (unsigned)ppr); ip = u0 - ir * ppr;
    nissued = 0;
  }
  // buffers [K0, K1) into flight (all of [0, NB) before `run`)
  template <int K0, int K1>
  __device__ __forceinline__ void fill() {
#pragma unroll
    for (int k = K0; k < K1; ++k) buf[k] = *reinterpret_cast<const double2 *>(src_next());
  }
  __device__ __forceinline__ void run(const double *xs, double *part) {
    int cr = (int)((unsigned)u0 / (unsigned)ppr), cp = u0 - cr * ppr;
    const int r_first = cr;
    double acc = 0.0;
#pragma unroll 1
    for (int i = 0; i < total; i += NB) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const double2 mv = buf[k];
        buf[k] = *reinterpret_cast<const double2 *>(src_next());
        if (i + k < total) {
          const double2 xv = *reinterpret_cast<const double2 *>(xs + cp * RING_PIECE + lane2);
          acc += mv.x * xv.x;
          acc += mv.y * xv.y;
          if (++cp == ppr || i + k + 1 == total) {
            const double s = wave_sum(acc);
            if (lane2 == 0) part[cr - r_first] = s;
            acc = 0.0; cp = 0; ++cr;
          }
        }
      }
    }
  }
};

template <int WAVES, int NB, int THIN, bool SCALE>
__global__ __launch_bounds__(64 * WAVES) void k_gemv_pipe(DenseMeta m, const double *__restrict__ x, double *__restrict__ yslots,
                                                          const int *done, const int *zero_x) {
  constexpr int NTH = 64 * WAVES, MRW = ring_max_rpw<WAVES>();
  static_assert(THIN <= NB, "thin start is part of the pipeline");
  const GemvTile t = m.tiles[blockIdx.x];
  const int done0 = done ? *done : 0, zero0 = zero_x ? *zero_x : 0;
  asm volatile("" ::"s"(t.mat_off), "s"(t.n), "s"(t.ld), "s"(t.loc_off), "s"(t.row0), "s"(t.active), "s"(t.nrows), "s"(done0), "s"(zero0));
  if (done0 || !t.active) return;
  RING_STAMP(0);
  __shared__ __attribute__((aligned(16))) double xs[RING_XS];
  __shared__ double part[WAVES * MRW];
  __shared__ int wrow[2 * WAVES];
  const int off = t.loc_off, n = t.n;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (zero0) {
    if ((int)threadIdx.x < t.nrows) yslots[m.out_pos[off + t.row0 + threadIdx.x]] = 0.0;
    return;
  }
  const int ppr = (t.ld + RING_PIECE - 1) / RING_PIECE, plen = ppr * RING_PIECE;
  constexpr int XPT = (RING_XS + NTH - 1) / NTH;
  int gi[XPT];
  double xv[XPT], cn[XPT];
  // Requests are served in issue order, by the CU and by the memory system behind it, and under a chip-wide stream a
  // dependent hop costs (bytes in flight chip-wide) / (7.8 TB/s): the operand gather (index, then value) runs beside a
  // THIN stream, and the pipeline is filled only when the last dependent load of every wave has been issued.
#pragma unroll
  for (int q = 0; q < XPT; ++q) {
    const int j = q * NTH + (int)threadIdx.x;
    gi[q] = j < n ? m.gidx[off + j] : -1;
    cn[q] = SCALE && j < n ? m.cnt[off + j] : 1.0;
  }
  PipeStream<NB> ps;
  ps.begin(m.M + t.mat_off + (long long)t.row0 * t.ld, t.ld, t.nrows, w, WAVES);
  if ((threadIdx.x & 63) == 0) {
    wrow[2 * w] = (int)((unsigned)ps.u0 / (unsigned)ps.ppr);
    wrow[2 * w + 1] = ps.total > 0 ? (int)((unsigned)(ps.u0 + ps.total - 1) / (unsigned)ps.ppr) : -1;
  }
  ps.template fill<0, THIN>();
#pragma unroll
  for (int q = 0; q < XPT; ++q) xv[q] = gi[q] >= 0 ? x[gi[q]] : 0.0;
  __builtin_amdgcn_s_barrier();   // every wave's value loads are in the queue
  ps.template fill<THIN, NB>();
#pragma unroll
  for (int q = 0; q < XPT; ++q) {
    const int j = q * NTH + (int)threadIdx.x;
    if (j < plen) xs[j] = SCALE ? xv[q] / cn[q] : xv[q];
  }
  RING_STAMP(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();    // raw: the matrix loads stay in flight across it
  RING_STAMP(2);
  ps.run(xs, part + w * MRW);
  RING_STAMP(3);
  __syncthreads();
  RING_STAMP(4);
  if ((int)threadIdx.x < t.nrows) {
    const int r = t.row0 + threadIdx.x;
    const double s = ring_row_sum<WAVES>(part, wrow, threadIdx.x);
    yslots[m.out_pos[off + r]] = SCALE ? s / m.cnt[off + r] : s;
  }
}

}  // namespace mi
