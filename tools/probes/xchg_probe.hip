// What does one table exchange cost on ONE GPU, and which part of it?  (The exchange of exchange.hpp: own entries of a
// table into every arena, system-scope release of the flags, bounded wait on the own arena's flags.)
// Variants of the publishing order, same data movement; arenas are fine-grained device memory as in the product.
//   0  every thread: plain stores, __threadfence_system(); barrier; one thread stores the flags (release, system); acquire poll
//   1  every thread: relaxed system-scope stores; barrier (workgroup fence); one thread: release flags; acquire poll
//   2  as 1, poll with relaxed loads, one acquire fence after the loop
//   3  plain stores; barrier; ONE thread __threadfence_system() + relaxed flag stores; poll relaxed + one acquire fence
//   4  relaxed system-scope (write-through) stores; barrier; relaxed flag stores — no L2 write-back anywhere; poll as 3
// Each variant alone, and behind a kernel that streams 64 MB and writes 1 MB (the state a GEMV launch leaves the L2s in).
// plus: the consumer's side — 256 workgroups reading the table from a fine-grained arena against ordinary device memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(2); } } while (0)

struct Peers { int n; char *arena[16]; };
struct State { unsigned long long epoch; int err; };

__device__ __forceinline__ unsigned long long *flag_of(const Peers &P, int q, int r) { return reinterpret_cast<unsigned long long *>(P.arena[q]) + r * 16; }

template <int V>
__global__ __launch_bounds__(1024) void k_push(Peers P, State *st, size_t table_off, size_t copy_doubles, const double *__restrict__ src,
                                               const int *__restrict__ own_idx, int n_own) {
  __shared__ unsigned long long e_sh;
  const unsigned long long e_next = st->epoch + 1;
  const size_t par = (size_t)(e_next & 1) * copy_doubles;
  for (int i0 = threadIdx.x; i0 < n_own; i0 += 4 * 1024) {
    int id[4];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = i0 + k * 1024; id[k] = i < n_own ? own_idx[i] : -1; }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = id[k] >= 0 ? src[id[k]] : 0.0;
    for (int q = 0; q < P.n; ++q) {
      double *dst = reinterpret_cast<double *>(P.arena[q] + table_off) + par;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (id[k] >= 0) {
          if (V == 1 || V == 2 || V == 4) __hip_atomic_store(&dst[id[k]], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else dst[id[k]] = v[k];
        }
    }
  }
  if (V == 0) __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    st->epoch = e_next;
    if (V == 3 || V == 4) {
      if (V == 3) __threadfence_system();
      for (int q = 0; q < P.n; ++q) __hip_atomic_store(flag_of(P, q, 0), e_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      for (int q = 0; q < P.n; ++q) __hip_atomic_store(flag_of(P, q, 0), e_next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    e_sh = e_next;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long *f = flag_of(P, 0, 0);
    long long spins = 0;
    if (V == 0 || V == 1) {
      while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e_sh) { if (++spins > 1000000) { st->err = 1; break; } __builtin_amdgcn_s_sleep(4); }
    } else {
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < e_sh) { if (++spins > 1000000) { st->err = 1; break; } __builtin_amdgcn_s_sleep(4); }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
  }
}
__global__ void k_empty() {}
__global__ __launch_bounds__(1024) void k_stream(const double2 *__restrict__ a, size_t n2, double *__restrict__ w) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)1024 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 1024) { const double2 v = a[i]; s += v.x + v.y; }
  if (threadIdx.x < 512) w[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(1024) void k_consume(const double *__restrict__ table, int n, double *out) {
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 1024) s += table[i];
  if (s == 12345.678) out[blockIdx.x] = s;
}

template <class F>
static float time_graph(hipStream_t s, int reps, F body) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < reps; ++i) body();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3f / reps;
}

int main(int argc, char **argv) {
  const int n_own = argc > 1 ? atoi(argv[1]) : 12000;
  const int reps = 200;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t table_off = 4096, copy_doubles = 16384, arena_bytes = table_off + 2 * copy_doubles * 8;
  std::vector<int> idx(n_own);
  for (int i = 0; i < n_own; ++i) idx[i] = i;
  int *own_idx; double *src, *out, *plain; State *st;
  CK(hipMalloc(&own_idx, n_own * 4)); CK(hipMemcpy(own_idx, idx.data(), n_own * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&src, copy_doubles * 8)); CK(hipMemset(src, 0, copy_doubles * 8));
  CK(hipMalloc(&plain, copy_doubles * 8)); CK(hipMemset(plain, 0, copy_doubles * 8));
  CK(hipMalloc(&out, 4096 * 8));
  CK(hipMalloc(&st, sizeof(State)));
  double2 *big; double *wout; const size_t big_n2 = (size_t)4 << 20;
  CK(hipMalloc(&big, big_n2 * 16)); CK(hipMemset(big, 0, big_n2 * 16)); CK(hipMalloc(&wout, 256 * 512 * 8));
  const float t_stream = time_graph(s, reps, [&] { hipLaunchKernelGGL(k_stream, dim3(256), dim3(1024), 0, s, big, big_n2, wout); });
  printf("streaming kernel alone %7.2f us\n", t_stream);
  printf("table of %d doubles; per-launch time over a graph of %d launches\n", n_own, reps);
  printf("  empty kernel                      %7.2f us\n", time_graph(s, reps, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }));
  for (int fine = 1; fine >= 0; --fine)
    for (int np : {1, 2, 4, 8}) {
      Peers P{}; P.n = np;
      for (int q = 0; q < np; ++q) {
        if (fine) CK(hipExtMallocWithFlags((void **)&P.arena[q], arena_bytes, hipDeviceMallocFinegrained));
        else CK(hipMalloc((void **)&P.arena[q], arena_bytes));
        CK(hipMemset(P.arena[q], 0, arena_bytes));
      }
      float t[5], tb[5];
      for (int v = 0; v < 5; ++v) {
        CK(hipMemset(st, 0, sizeof(State)));
        for (int q = 0; q < np; ++q) CK(hipMemset(P.arena[q], 0, 4096));
        CK(hipDeviceSynchronize());
        auto run = [&] {
          switch (v) {
            case 0: hipLaunchKernelGGL(k_push<0>, dim3(1), dim3(1024), 0, s, P, st, table_off, copy_doubles, src, own_idx, n_own); break;
            case 1: hipLaunchKernelGGL(k_push<1>, dim3(1), dim3(1024), 0, s, P, st, table_off, copy_doubles, src, own_idx, n_own); break;
            case 2: hipLaunchKernelGGL(k_push<2>, dim3(1), dim3(1024), 0, s, P, st, table_off, copy_doubles, src, own_idx, n_own); break;
            case 3: hipLaunchKernelGGL(k_push<3>, dim3(1), dim3(1024), 0, s, P, st, table_off, copy_doubles, src, own_idx, n_own); break;
            default: hipLaunchKernelGGL(k_push<4>, dim3(1), dim3(1024), 0, s, P, st, table_off, copy_doubles, src, own_idx, n_own);
          }
        };
        t[v] = time_graph(s, reps, run);
        tb[v] = time_graph(s, reps, [&] { hipLaunchKernelGGL(k_stream, dim3(256), dim3(1024), 0, s, big, big_n2, wout); run(); }) - t_stream;
      }
      State h; CK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
      printf("  %s arenas, %d destination(s):  v0 %6.2f  v1 %6.2f  v2 %6.2f  v3 %6.2f  v4 %6.2f us | behind a streaming kernel: %6.2f %6.2f %6.2f %6.2f %6.2f  (err %d)\n",
             fine ? "fine-grained" : "ordinary    ", np, t[0], t[1], t[2], t[3], t[4], tb[0], tb[1], tb[2], tb[3], tb[4], h.err);
      if (np == 1) {
        const double *tab = reinterpret_cast<const double *>(P.arena[0] + table_off);
        printf("    256 workgroups reading the table: from this arena %6.2f us, from ordinary memory %6.2f us\n",
               time_graph(s, reps, [&] { hipLaunchKernelGGL(k_consume, dim3(256), dim3(1024), 0, s, tab, n_own, out); }),
               time_graph(s, reps, [&] { hipLaunchKernelGGL(k_consume, dim3(256), dim3(1024), 0, s, plain, n_own, out); }));
      }
      for (int q = 0; q < np; ++q) CK(hipFree(P.arena[q]));
    }
  return 0;
}
