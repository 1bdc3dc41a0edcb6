// Which host-side call stalls behind another stream's spinning kernel? N host threads ("ranks"), one non-blocking stream each:
//   [optional host op] [big kernel] [signal] [wait for all signals, bounded] — per variant, wall time per thread.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <barrier>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(2); } } while (0)
__global__ void k_big(double *x, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = x[i] * 1.0000001 + 1.0;
}
// a workgroup that needs a whole CU: 1024 threads, ~128 VGPRs per lane, 16 KiB LDS (the shape of the dense GEMV kernels)
__global__ __launch_bounds__(1024) void k_heavy(double *x, int n) {
  __shared__ double sm[2048];
  double r[56];
#pragma unroll
  for (int k = 0; k < 56; ++k) r[k] = x[(threadIdx.x + k * 1024 + blockIdx.x * 7) % n];
  sm[threadIdx.x] = r[0]; sm[threadIdx.x + 1024] = r[1];
  __syncthreads();
  double s = sm[(threadIdx.x * 7) % 2048];
#pragma unroll
  for (int k = 0; k < 56; ++k) s += r[k] * (double)(k + 1);
  x[(blockIdx.x * 1024 + threadIdx.x) % n] = s;
}
__global__ void k_signal(unsigned long long *flags, int i, unsigned long long e) {
  __hip_atomic_store(&flags[i * 16], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait(const unsigned long long *flags, int n, unsigned long long e, int *expired, long long ticks) {
  const int q = threadIdx.x;
  if (q >= n) return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(&flags[q * 16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
    if (wall_clock64() - t0 > ticks) { atomicAdd(expired, 1); return; }
    __builtin_amdgcn_s_sleep(8);
  }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  unsigned long long *flags; int *expired;
  CK(hipMalloc(&flags, 64 * 16 * 8)); CK(hipMemset(flags, 0, 64 * 16 * 8));
  CK(hipMalloc(&expired, 4)); CK(hipMemset(expired, 0, 4));
  hipStream_t util; CK(hipStreamCreateWithFlags(&util, hipStreamNonBlocking));
  double *scratch; CK(hipMalloc(&scratch, 1 << 20));
  CK(hipDeviceSynchronize());
  const char *names[] = {"nothing", "hipMalloc 64MB", "pageable H2D 1MB + sync", "memset on a shared util stream + sync", "hipHostMalloc 1MB", "hipMalloc+hipFree 1MB",
                         "hipEventCreate+Record+Sync", "hipMemcpyAsync D2H pageable 1MB + sync", "hipMalloc 32KB (kept)", "hipMalloc 1MB (kept)",
                         "hipHostMalloc+hipHostFree 1MB", "hipExtMallocWithFlags finegrained 1MB (kept)", "graph capture+instantiate+launch",
                         "hipStreamCreate+Destroy", "hipEventCreate+Destroy", "hipFree of a block allocated before",
                         "whole-CU workgroups instead of k_big", "whole-CU workgroups, wait kernel of 1024 threads",
                         "pinned 32KB H2D before, D2H after the wait (async)", "pinned 4KB H2D before, D2H after the wait (async)"};
  unsigned long long epoch = 0;
  for (int variant = (argc > 2 ? atoi(argv[2]) : 0); variant < 20; ++variant) {
    ++epoch;
    std::vector<std::thread> th;
    std::vector<double> took(n);
    std::barrier bar(n);
    std::vector<hipStream_t> st(n);
    std::vector<double *> buf(n);
    for (int i = 0; i < n; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipMalloc(&buf[i], 8 << 20)); }
    std::vector<char> host(1 << 20, 1);
    std::vector<void *> pre(n, nullptr);
    for (int i = 0; i < n; ++i) CK(hipMalloc(&pre[i], 1 << 20));
    for (int i = 0; i < n; ++i)
      th.emplace_back([&, i]() {
        bar.arrive_and_wait();
        std::this_thread::sleep_for(std::chrono::milliseconds(20 * i));   // ranks arrive one after the other: the early ones are already spinning
        const auto t0 = std::chrono::steady_clock::now();
        void *p = nullptr; hipEvent_t ev;
        std::vector<char> mine(1 << 20, 2);
        switch (variant) {
          case 1: CK(hipMalloc(&p, 64 << 20)); break;
          case 2: CK(hipMemcpyAsync(buf[i], mine.data(), 1 << 20, hipMemcpyHostToDevice, st[i])); CK(hipStreamSynchronize(st[i])); break;
          case 3: CK(hipMemsetAsync(scratch, 0, 1 << 20, util)); CK(hipStreamSynchronize(util)); break;
          case 4: CK(hipHostMalloc(&p, 1 << 20)); break;
          case 5: CK(hipMalloc(&p, 1 << 20)); CK(hipFree(p)); break;
          case 6: CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventRecord(ev, st[i])); CK(hipEventSynchronize(ev)); break;
          case 7: CK(hipMemcpyAsync(mine.data(), buf[i], 1 << 20, hipMemcpyDeviceToHost, st[i])); CK(hipStreamSynchronize(st[i])); break;
          case 8: CK(hipMalloc(&p, 32 << 10)); break;
          case 9: CK(hipMalloc(&p, 1 << 20)); break;
          case 10: CK(hipHostMalloc(&p, 1 << 20)); CK(hipHostFree(p)); break;
          case 11: CK(hipExtMallocWithFlags(&p, 1 << 20, hipDeviceMallocFinegrained)); break;
          case 12: {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal));
            hipLaunchKernelGGL(k_big, dim3(256), dim3(1024), 0, st[i], buf[i], 1 << 20);
            CK(hipStreamEndCapture(st[i], &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st[i]));
            break;
          }
          case 13: { hipStream_t t; CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking)); CK(hipStreamDestroy(t)); break; }
          case 14: CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventDestroy(ev)); break;
          case 15: CK(hipFree(pre[i])); break;
          default: break;
        }
        void *pin = nullptr;
        const size_t cb = variant == 18 ? (32 << 10) : (4 << 10);
        if (variant >= 18) { CK(hipHostMalloc(&pin, 64 << 10)); CK(hipMemcpyAsync(buf[i], pin, cb, hipMemcpyHostToDevice, st[i])); }
        if (variant >= 16 && variant < 18) hipLaunchKernelGGL(k_heavy, dim3(256), dim3(1024), 0, st[i], buf[i], 1 << 20);
        else hipLaunchKernelGGL(k_big, dim3(256), dim3(1024), 0, st[i], buf[i], 1 << 20);
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st[i], flags, i, epoch);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(variant == 17 ? 1024 : 64), 0, st[i], flags, n, epoch, expired, 150000000ll);   // 1.5 s
        if (variant >= 18) CK(hipMemcpyAsync(pin, buf[i], cb, hipMemcpyDeviceToHost, st[i]));
        CK(hipStreamSynchronize(st[i]));
        took[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      });
    for (auto &t : th) t.join();
    int h; CK(hipMemcpy(&h, expired, 4, hipMemcpyDeviceToHost)); CK(hipMemset(expired, 0, 4));
    double mx = 0; for (double v : took) mx = v > mx ? v : mx;
    printf("variant %d (%-40s): expired waits %d, slowest rank %.3f s\n", variant, names[variant], h, mx);
    fflush(stdout);
    for (int i = 0; i < n; ++i) { CK(hipStreamDestroy(st[i])); }
  }
  return 0;
}
