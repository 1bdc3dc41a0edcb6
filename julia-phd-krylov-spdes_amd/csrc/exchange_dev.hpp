// Device-side view of the one-shot peer exchange (exchange.hpp has the protocol and the host side): the structures the
// folded PCG launches (kernels.hpp) and the exchange kernels share.
#pragma once
#include <hip/hip_runtime.h>

namespace mi {

constexpr int XCHG_MAX_RANKS = 16;
constexpr int XCHG_FLAG_STRIDE = 16;                                  // unsigned long long per flag: a 128-byte line each
constexpr size_t XCHG_FLAG_BYTES = XCHG_MAX_RANKS * XCHG_FLAG_STRIDE * 8;  // flags at offset 0 of every arena

struct XchgState {            // device resident, one per context
  unsigned long long epoch;   // exchanges this rank has signalled
  int err;                    // a bounded wait expired
  int err_rank;               // ... waiting for this rank
  unsigned long long err_epoch, err_seen;  // ... at this exchange; the flag stood at err_seen
  int *abort_done;            // the running solve's stop flag (or null): an expired wait ends the loop instead of iterating on garbage
  unsigned int arrived;       // workgroups of the running table exchange that have published their stores
  unsigned long long xep[2];  // exchange number as the folded launches carry it when they wait themselves (kernels.hpp, PcgFold::x_inwait)
};
struct XchgPeers {            // by-value kernel argument
  int n, rank;
  long long timeout;          // wall-clock ticks (100 MHz) a wait may take
  char *arena[XCHG_MAX_RANKS];
};

__device__ __forceinline__ unsigned long long *xchg_flag(const XchgPeers &P, int arena_of, int flag_of) {
  return reinterpret_cast<unsigned long long *>(P.arena[arena_of]) + (size_t)flag_of * XCHG_FLAG_STRIDE;
}

// thread 0 of a workgroup whose stores are all behind a system-scope fence and a barrier: next exchange number to all peers
__device__ __forceinline__ unsigned long long xchg_signal(XchgState *st, const XchgPeers &P) {
  const unsigned long long e = st->epoch + 1;
  st->epoch = e;
  for (int q = 0; q < P.n; ++q) __hip_atomic_store(xchg_flag(P, q, P.rank), e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  return e;
}
// lanes 0..n-1 of one wave: until flag q of the own arena has reached e (bounded). Relaxed polls that bypass the caches —
// an acquire load per poll invalidates the L2 each time, and with every workgroup of a launch polling that cost 60 us per
// launch; the CALLER issues the one acquire fence its readers need. The clock and the error word are only looked at
// when the flag is not there at the first look.
__device__ __forceinline__ void xchg_wait(XchgState *st, const XchgPeers &P, unsigned long long e) {
  const int q = threadIdx.x;
  if (q >= P.n) return;
  const unsigned long long *f = xchg_flag(P, P.rank, q);
  if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= e) return;
  const long long t0 = wall_clock64();
  if (__hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {   // a wait has expired before: fall through, the solve fails anyway
    if (st->abort_done) *st->abort_done = 1;
    return;
  }
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
    if (wall_clock64() - t0 > P.timeout) {
      if (atomicExch(&st->err, 1) == 0) { st->err_rank = q; st->err_epoch = e; st->err_seen = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      if (st->abort_done) *st->abort_done = 1;
      return;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

// a result on its way to a peer (or to the own arena): write-through to the system coherence point, nothing cached
__device__ __forceinline__ void xchg_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

}  // namespace mi
