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
};
struct XchgPeers {            // by-value kernel argument
  int n, rank;
  long long timeout;          // wall-clock ticks (100 MHz) a wait may take
  char *arena[XCHG_MAX_RANKS];
};

__device__ __forceinline__ unsigned long long *xchg_flag(const XchgPeers &P, int arena_of, int flag_of) {
  return reinterpret_cast<unsigned long long *>(P.arena[arena_of]) + (size_t)flag_of * XCHG_FLAG_STRIDE;
}

// a result on its way to a peer (or to the own arena): write-through to the system coherence point, nothing cached
__device__ __forceinline__ void xchg_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

}  // namespace mi
