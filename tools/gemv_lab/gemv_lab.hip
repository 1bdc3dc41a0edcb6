// GEMV lab: stand-alone timing of the dense S-apply stream (config 3 block sizes) for kernel variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o gemv_lab gemv_lab.hip
// Variants: the library's k_gemv_batched (baseline, included from csrc/kernels.hpp) and k_gemv_ring
// (wave-specialised LDS-DMA ring, byte-balanced tiles). Each timing replays a graph of launches that
// alternates two operators (S and ΠS: 2 x 68 MB, the working set of a PCG iteration).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#define MI355_RING_LAB 1
#include "../../julia-phd-krylov-spdes_amd/csrc/kernels.hpp"
#include "gemv_ring.hpp"

using namespace mi;

#define CK(e)                                                                              \
  do {                                                                                     \
    hipError_t r_ = (e);                                                                   \
    if (r_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s: %s (%s:%d)\n", #e, hipGetErrorString(r_), __FILE__, __LINE__);  \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)

struct Op {
  std::vector<int> nd, ld, loc_off;
  std::vector<long long> moff;
  int nloc = 0;
  long long tot = 0;
  double *M = nullptr;
  int *gidx = nullptr, *out_pos = nullptr;
  double *cnt = nullptr;
  std::vector<double> Mh;
  std::vector<int> gidx_h;
  GemvTile *tiles = nullptr;
  int ntiles = 0;
  long long alg_bytes = 0;
};

static void build_op(Op &o, const std::vector<int> &sizes, int n_gamma, unsigned seed) {
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int n : sizes) {
    int l = (n + 15) / 16 * 16;
    if (l % 256 == 0 && l != 2048) l += 16;
    o.nd.push_back(n); o.ld.push_back(l); o.loc_off.push_back(o.nloc); o.moff.push_back(o.tot);
    o.nloc += n; o.tot += (long long)n * l;
    o.alg_bytes += 8ll * n * n + 20ll * n;
  }
  o.Mh.assign((size_t)o.tot + 4096, 0.0);
  for (size_t d = 0; d < sizes.size(); ++d)
    for (int i = 0; i < o.nd[d]; ++i)
      for (int j = 0; j < o.nd[d]; ++j) o.Mh[o.moff[d] + (long long)i * o.ld[d] + j] = u(g);
  o.gidx_h.resize(o.nloc);
  for (int s = 0; s < o.nloc; ++s) o.gidx_h[s] = (int)(((long long)s * 7919) % n_gamma);
  std::vector<int> op(o.nloc);
  for (int s = 0; s < o.nloc; ++s) op[s] = s;
  std::vector<double> cnt(o.nloc, 1.0);
  CK(hipMalloc(&o.M, o.Mh.size() * 8));
  CK(hipMemcpy(o.M, o.Mh.data(), o.Mh.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&o.gidx, o.nloc * 4)); CK(hipMemcpy(o.gidx, o.gidx_h.data(), o.nloc * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&o.out_pos, o.nloc * 4)); CK(hipMemcpy(o.out_pos, op.data(), o.nloc * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&o.cnt, o.nloc * 8)); CK(hipMemcpy(o.cnt, cnt.data(), o.nloc * 8, hipMemcpyHostToDevice));
}

// rows-per-tile tiling (the library's): step rows per tile
static void tile_fixed(Op &o, int step) {
  std::vector<GemvTile> tv;
  for (size_t d = 0; d < o.nd.size(); ++d)
    for (int r = 0; r < o.nd[d]; r += step)
      tv.push_back(GemvTile{o.moff[d], o.nd[d], o.ld[d], o.loc_off[d], r, 1, std::min(step, o.nd[d] - r)});
  if (o.tiles) CK(hipFree(o.tiles));
  o.ntiles = (int)tv.size();
  CK(hipMalloc(&o.tiles, tv.size() * sizeof(GemvTile)));
  CK(hipMemcpy(o.tiles, tv.data(), tv.size() * sizeof(GemvTile), hipMemcpyHostToDevice));
}
// byte-balanced tiling: `want` tiles in all, apportioned to the blocks by bytes (largest remainder), rows split evenly
static void tile_balanced(Op &o, int want, int max_rows) {
  const size_t nb = o.nd.size();
  std::vector<double> share(nb);
  double totb = 0;
  for (size_t d = 0; d < nb; ++d) totb += (double)o.nd[d] * o.ld[d];
  std::vector<int> cnt(nb);
  int used = 0;
  std::vector<std::pair<double, int>> rem;
  for (size_t d = 0; d < nb; ++d) {
    share[d] = (double)o.nd[d] * o.ld[d] / totb * want;
    cnt[d] = std::max(1, (int)std::floor(share[d]));
    cnt[d] = std::max(cnt[d], (o.nd[d] + max_rows - 1) / max_rows);
    used += cnt[d];
    rem.push_back({share[d] - std::floor(share[d]), (int)d});
  }
  std::sort(rem.rbegin(), rem.rend());
  for (size_t k = 0; used < want && k < rem.size(); ++k) { cnt[rem[k].second]++; used++; }
  std::vector<GemvTile> tv;
  for (size_t d = 0; d < nb; ++d) {
    const int c = std::min(cnt[d], o.nd[d]);
    for (int k = 0; k < c; ++k) {
      const int r0 = (int)((long long)o.nd[d] * k / c), r1 = (int)((long long)o.nd[d] * (k + 1) / c);
      tv.push_back(GemvTile{o.moff[d], o.nd[d], o.ld[d], o.loc_off[d], r0, 1, r1 - r0});
    }
  }
  if (o.tiles) CK(hipFree(o.tiles));
  o.ntiles = (int)tv.size();
  CK(hipMalloc(&o.tiles, tv.size() * sizeof(GemvTile)));
  CK(hipMemcpy(o.tiles, tv.data(), tv.size() * sizeof(GemvTile), hipMemcpyHostToDevice));
}

static std::vector<double> ref_apply(const Op &o, const std::vector<double> &x) {
  std::vector<double> y(o.nloc);
  for (size_t d = 0; d < o.nd.size(); ++d)
    for (int i = 0; i < o.nd[d]; ++i) {
      long double s = 0;
      for (int j = 0; j < o.nd[d]; ++j) s += (long double)o.Mh[o.moff[d] + (long long)i * o.ld[d] + j] * x[o.gidx_h[o.loc_off[d] + j]];
      y[o.loc_off[d] + i] = (double)s;
    }
  return y;
}

template <class F>
static double time_graph(hipStream_t s, int launches, int replays, F &&launch_one) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < launches; ++i) launch_one(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, (double)ms * 1e3 / (launches * replays));
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best;
}

static double check(const Op &o, const double *ydev, const std::vector<double> &yref, const char *name) {
  std::vector<double> y(o.nloc);
  CK(hipMemcpy(y.data(), ydev, o.nloc * 8, hipMemcpyDeviceToHost));
  double err = 0, mx = 0;
  for (int i = 0; i < o.nloc; ++i) { err = std::max(err, std::fabs(y[i] - yref[i])); mx = std::max(mx, std::fabs(yref[i])); }
  if (!(err <= 1e-11 * mx)) printf("!! %s: max err %.3e (max |y| %.3e)\n", name, err, mx);
  return err / mx;
}

// one launch with stamps: per-workgroup times relative to the earliest workgroup start
static long long *g_dbg_dev = nullptr;
template <class F>
static void stamp_report(hipStream_t s, int ntiles, F &&launch) {
  if (!g_dbg_dev) CK(hipMalloc(&g_dbg_dev, 4096 * 8 * 8));
  CK(hipMemset(g_dbg_dev, 0, 4096 * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(mi::g_ring_dbg), &g_dbg_dev, sizeof(void *)));
  launch(); launch(); launch();
  CK(hipStreamSynchronize(s));
  std::vector<long long> h(ntiles * 8);
  CK(hipMemcpy(h.data(), g_dbg_dev, h.size() * 8, hipMemcpyDeviceToHost));
  long long *nul = nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(mi::g_ring_dbg), &nul, sizeof(void *)));
  long long t0 = h[0];
  for (int b = 0; b < ntiles; ++b) t0 = std::min(t0, h[b * 8]);
  double avg[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < ntiles; ++b)
    for (int k = 0; k < 5; ++k) {
      const double us = (h[b * 8 + k] - t0) * 0.01;
      avg[k] += us / ntiles; mx[k] = std::max(mx[k], us);
    }
  printf("      stamps us (avg/max over WGs, from first WG start): start %.2f/%.2f  pre-barrier %.2f/%.2f  staged %.2f/%.2f  streamed %.2f/%.2f  end %.2f/%.2f\n",
         avg[0], mx[0], avg[1], mx[1], avg[2], mx[2], avg[3], mx[3], avg[4], mx[4]);
}

template <int WAVES, int SW, int D>
static void run_ring(const char *tag, hipStream_t s, Op (&ops)[2], const double *x, double *y, const std::vector<double> (&yref)[2],
                     int mode, int thin = 2) {
  auto kern = k_gemv_ring<WAVES, SW, D, false>;
  for (int k = 0; k < 2; ++k) {
    CK(hipMemsetAsync(y, 0, ops[k].nloc * 8, s));
    DenseMeta m{ops[k].M, ops[k].tiles, ops[k].gidx, ops[k].cnt, ops[k].out_pos};
    hipLaunchKernelGGL(kern, dim3(ops[k].ntiles), dim3(64 * WAVES), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr, mode, thin);
    CK(hipStreamSynchronize(s));
    if (mode == 0) check(ops[k], y, yref[k], tag);
  }
  const double us = time_graph(s, 40, 10, [&](int i) {
    const Op &o = ops[i & 1];
    DenseMeta m{o.M, o.tiles, o.gidx, o.cnt, o.out_pos};
    hipLaunchKernelGGL(kern, dim3(o.ntiles), dim3(64 * WAVES), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr, mode, thin);
  });
  printf("%-40s thin=%2d tiles=%4d  %7.2f us  %6.2f TB/s  frac %.3f%s\n", tag, thin, ops[0].ntiles, us, ops[0].alg_bytes / us / 1e6,
         ops[0].alg_bytes / us / 1e6 / 8.0, mode ? "  (no consume: DMA floor)" : "");
  {
    int i = 0;
    stamp_report(s, ops[0].ntiles, [&]() {
      const Op &o = ops[(i++) & 1];
      DenseMeta m{o.M, o.tiles, o.gidx, o.cnt, o.out_pos};
      hipLaunchKernelGGL(kern, dim3(o.ntiles), dim3(64 * WAVES), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr, mode, thin);
    });
  }
  fflush(stdout);
}

template <int WAVES, int NB, int THIN>
static void run_pipe(const char *tag, hipStream_t s, Op (&ops)[2], const double *x, double *y, const std::vector<double> (&yref)[2]) {
  auto kern = k_gemv_pipe<WAVES, NB, THIN, false>;
  for (int k = 0; k < 2; ++k) {
    CK(hipMemsetAsync(y, 0, ops[k].nloc * 8, s));
    DenseMeta m{ops[k].M, ops[k].tiles, ops[k].gidx, ops[k].cnt, ops[k].out_pos};
    hipLaunchKernelGGL(kern, dim3(ops[k].ntiles), dim3(64 * WAVES), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr);
    CK(hipStreamSynchronize(s));
    check(ops[k], y, yref[k], tag);
  }
  const double us = time_graph(s, 40, 10, [&](int i) {
    const Op &o = ops[i & 1];
    DenseMeta m{o.M, o.tiles, o.gidx, o.cnt, o.out_pos};
    hipLaunchKernelGGL(kern, dim3(o.ntiles), dim3(64 * WAVES), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr);
  });
  printf("%-48s tiles=%4d  %7.2f us  %6.2f TB/s  frac %.3f\n", tag, ops[0].ntiles, us, ops[0].alg_bytes / us / 1e6,
         ops[0].alg_bytes / us / 1e6 / 8.0);
  {
    int i = 0;
    stamp_report(s, ops[0].ntiles, [&]() {
      const Op &o = ops[(i++) & 1];
      DenseMeta m{o.M, o.tiles, o.gidx, o.cnt, o.out_pos};
      hipLaunchKernelGGL(kern, dim3(o.ntiles), dim3(64 * WAVES), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr);
    });
  }
  fflush(stdout);
}

int main(int argc, char **argv) {
  std::vector<int> sizes = {749, 1249, 1249, 748, 748, 1247, 1247, 747};
  if (argc > 1 && std::string(argv[1]) == "equal") sizes.assign(8, 1000);
  if (argc > 1 && std::string(argv[1]) == "many") sizes.assign(160, 130);
  int n_gamma = 3989;
  if (sizes.size() > 8) n_gamma = 9417;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Op ops[2];
  build_op(ops[0], sizes, n_gamma, 1);
  build_op(ops[1], sizes, n_gamma, 2);
  std::vector<double> xh(n_gamma);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (auto &v : xh) v = u(g);
  double *x, *y;
  CK(hipMalloc(&x, n_gamma * 8)); CK(hipMemcpy(x, xh.data(), n_gamma * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&y, ops[0].nloc * 8 + 64));
  std::vector<double> yref[2] = {ref_apply(ops[0], xh), ref_apply(ops[1], xh)};
  printf("blocks:");
  for (int n : sizes) printf(" %d", n);
  printf("  alg bytes/launch %.3f MB\n", ops[0].alg_bytes / 1e6);

  // ---- baseline: the library's kernel, 32-row tiles
  for (int k = 0; k < 2; ++k) tile_fixed(ops[k], 32);
  {
    for (int k = 0; k < 2; ++k) {
      CK(hipMemsetAsync(y, 0, ops[k].nloc * 8, s));
      DenseMeta m{ops[k].M, ops[k].tiles, ops[k].gidx, ops[k].cnt, ops[k].out_pos};
      hipLaunchKernelGGL((k_gemv_batched<2, false, 16>), dim3(ops[k].ntiles), dim3(1024), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr);
      CK(hipStreamSynchronize(s));
      check(ops[k], y, yref[k], "baseline");
    }
    const double us = time_graph(s, 40, 10, [&](int i) {
      const Op &o = ops[i & 1];
      DenseMeta m{o.M, o.tiles, o.gidx, o.cnt, o.out_pos};
      hipLaunchKernelGGL((k_gemv_batched<2, false, 16>), dim3(o.ntiles), dim3(1024), 0, s, m, x, y, (const int *)nullptr, (const int *)nullptr);
    });
    printf("%-44s tiles=%4d  %7.2f us  %6.2f TB/s  frac %.3f\n", "baseline k_gemv_batched<2,16> 32-row tiles", ops[0].ntiles, us,
           ops[0].alg_bytes / us / 1e6, ops[0].alg_bytes / us / 1e6 / 8.0);
    // empty-grid cost: the same grid returning at once
    int *one;
    CK(hipMalloc(&one, 4));
    int h1 = 1;
    CK(hipMemcpy(one, &h1, 4, hipMemcpyHostToDevice));
    const double us0 = time_graph(s, 40, 10, [&](int i) {
      const Op &o = ops[i & 1];
      DenseMeta m{o.M, o.tiles, o.gidx, o.cnt, o.out_pos};
      hipLaunchKernelGGL((k_gemv_batched<2, false, 16>), dim3(o.ntiles), dim3(1024), 0, s, m, x, y, (const int *)one, (const int *)nullptr);
    });
    printf("%-44s tiles=%4d  %7.2f us\n", "same grid, early exit", ops[0].ntiles, us0);
  }
  // ---- register pipeline, old tiles
  run_pipe<16, 8, 8>("pipe W16 NB8 T8, 32-row tiles", s, ops, x, y, yref);
  int ncu = 256;
  {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    ncu = p.multiProcessorCount;
  }
  for (int k = 0; k < 2; ++k) tile_balanced(ops[k], ncu, 64);
  run_pipe<16, 2, 2>("pipe W16 NB2 T2, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 4, 0>("pipe W16 NB4 T0, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 4, 1>("pipe W16 NB4 T1, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 4, 4>("pipe W16 NB4 T4, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 6, 1>("pipe W16 NB6 T1, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 8, 0>("pipe W16 NB8 T0, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 8, 1>("pipe W16 NB8 T1, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 8, 2>("pipe W16 NB8 T2, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 8, 8>("pipe W16 NB8 T8, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 12, 1>("pipe W16 NB12 T1, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<16, 16, 1>("pipe W16 NB16 T1, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<8, 4, 1>("pipe W8 NB4 T1, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<8, 8, 2>("pipe W8 NB8 T2, balanced 1/CU", s, ops, x, y, yref);
  run_pipe<8, 16, 2>("pipe W8 NB16 T2, balanced 1/CU", s, ops, x, y, yref);
  run_ring<16, 8, 8>("ring W16 S8 D8, balanced 1/CU", s, ops, x, y, yref, 0, 2);
  run_ring<16, 8, 16>("ring W16 S8 D16, balanced 1/CU", s, ops, x, y, yref, 0, 2);
  run_ring<16, 8, 16>("ring W16 S8 D16, balanced 1/CU", s, ops, x, y, yref, 1, 2);
  for (int k = 0; k < 2; ++k) tile_balanced(ops[k], 2 * ncu, 64);
  run_pipe<8, 4, 1>("pipe W8 NB4 T1, balanced 2/CU", s, ops, x, y, yref);
  run_pipe<8, 8, 2>("pipe W8 NB8 T2, balanced 2/CU", s, ops, x, y, yref);
  return 0;
}
