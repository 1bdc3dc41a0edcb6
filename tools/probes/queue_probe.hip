// Do N streams of one process make progress independently when kernels of one spin on flags set by kernels of another?
// stream i: [signal_i: flag[i] = e] [wait_i: until all flags >= e], e = 1..E, enqueued stream by stream (stream 0 first).
// A bounded spin reports how many waits expired. Run with and without GPU_MAX_HW_QUEUES.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(2); } } while (0)
__global__ void k_signal(unsigned long long *flags, int i, unsigned long long e) {
  __hip_atomic_store(&flags[i * 16], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait(const unsigned long long *flags, int n, unsigned long long e, int *expired, long long max_spin) {
  const int q = threadIdx.x;
  if (q >= n) return;
  long long spins = 0;
  while (__hip_atomic_load(&flags[q * 16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
    if (++spins > max_spin) { atomicAdd(expired, 1); return; }
    __builtin_amdgcn_s_sleep(8);
  }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8, E = 20;
  std::vector<hipStream_t> s(n);
  for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  unsigned long long *flags; int *expired;
  CK(hipMalloc(&flags, n * 16 * 8)); CK(hipMemset(flags, 0, n * 16 * 8));
  CK(hipMalloc(&expired, 4)); CK(hipMemset(expired, 0, 4));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s[0]));
  for (int e = 1; e <= E; ++e)
    for (int i = 0; i < n; ++i) {
      hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s[i], flags, i, (unsigned long long)e);
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s[i], flags, n, (unsigned long long)e, expired, 2000000ll);
    }
  for (int i = 1; i < n; ++i) CK(hipStreamSynchronize(s[i]));
  CK(hipEventRecord(e1, s[0]));
  CK(hipStreamSynchronize(s[0]));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  int h; CK(hipMemcpy(&h, expired, 4, hipMemcpyDeviceToHost));
  printf("streams=%d exchanges=%d expired_waits=%d total %.3f ms (%.2f us per exchange)\n", n, E, h, ms, ms * 1e3 / E);
  return 0;
}
