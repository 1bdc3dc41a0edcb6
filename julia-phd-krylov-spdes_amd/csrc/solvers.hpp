// cg / pcg / defcg / defpcg with the whole loop on the device (RecyclingKrylovSolvers/cg.jl, defcg.jl).
//
// One iteration is a fixed sequence of kernel launches whose scalars (alpha, beta, r'r, r'z, it,
// the stop flag) live in HBM (`SolverState`), so the host never reads a scalar inside the loop.
// Iterations are captured into hipGraphs and replayed; launches that follow the iteration at which
// the stop rule fired return immediately (`done`), so the iterates, `it` and `res_norm` are exactly
// those of the reference's `while (it < maxit) && (res_norm[it] > tol)` whatever the replay sizes.
//
// Replay policy: the first replay holds the set-up (r = b - A x, z = M \ r, p = z, it = 1) plus as
// many iterations as the previous solve with the same (A, M) needed, so a repeated or similar solve
// (Example07's realization loop) is ONE graph launch and ONE host wait; result copies are enqueued
// speculatively behind it. If the stop flag is not up, `chunk`-sized replays follow, the host
// looking at the flag of the PREVIOUS replay (one replay of run-ahead).
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <tuple>

#include "eig_kernels.hpp"
#include "operators.hpp"
#ifdef MI355_EXPERIMENTAL
#include "resident.hpp"   // persistent on-chip PCG: measured and not adopted (profiles/NOTES.md); `make EXPERIMENTAL=1`
#endif

namespace mi {


constexpr int64_t RES_STAGE = 8192;
constexpr int CF_VEC_GRID = 256;  // workgroups of the vector half of the 2-launch sparse pcg (8 XCDs x 32) = partials per dot (<= NT)

struct GraphKey {
  const Operator *A, *M;
  int nvec, chunk;  // chunk > 0: that many iterations; chunk < 0: set-up + (-chunk) iterations
  int tag = 0;      // 4*(0: cg/pcg/defcg/defpcg; eigCG family: 1 + kind + 8*spdim) + loop form (fused, folded)
  bool operator<(const GraphKey &o) const {
    return std::tie(A, M, nvec, chunk, tag) < std::tie(o.A, o.M, o.nvec, o.chunk, o.tag);
  }
};

#ifdef MI355_EXPERIMENTAL
// Host side of the persistent on-chip PCG (resident.hpp): which rows of which subdomain every workgroup owns, how many
// of them fit into registers / LDS, and the hand-off buffers of the grid barrier.
struct ResidentPlan {
  int G = 0, max_rows = 0;
  double resident_frac = 0.0;   // share of the matrix elements held in registers or LDS or passing through the streaming slot
  bool usable = false;
  unsigned long long epoch = 16;   // records are zero-initialised: epochs start above 0
  DevBuf<ResTile> tiles;
  DevBuf<ResRec> recs;
  DevBuf<int> abort;
  std::vector<ResTile> tiles_h;

  ResidentPlan(mi_ctx_s *c, const DenseBlockOp &A, const DenseBlockOp &M) {
    hipDeviceProp_t prop;
    MI_HIP(hipGetDeviceProperties(&prop, c->device));
    G = env_int("MI355_RES_G", prop.multiProcessorCount);
    const int ndl = A.maps.ndl;
    if (G < 1 || ndl < 1 || ndl > G || G > RES_NTH) return;   // one published record per thread when polling
    // workgroups per subdomain in proportion to its elements, at least one each, G in all
    std::vector<double> wgt(ndl);
    double tot = 0.0;
    for (int d = 0; d < ndl; ++d) { wgt[d] = (double)A.maps.nd[d] * A.ld_h[d]; tot += wgt[d]; }
    if (tot <= 0.0) return;
    std::vector<int> gd(ndl);
    int used = 0;
    for (int d = 0; d < ndl; ++d) { gd[d] = std::max(1, (int)(G * wgt[d] / tot)); used += gd[d]; }
    while (used > G) {   // over-subscribed by the "at least one" rule: take from the subdomain with the fewest rows per workgroup
      int best = -1; double v = 1e300;
      for (int d = 0; d < ndl; ++d) if (gd[d] > 1 && wgt[d] / gd[d] < v) { v = wgt[d] / gd[d]; best = d; }
      if (best < 0) return;
      --gd[best]; --used;
    }
    while (used < G) {   // hand the spare workgroups to the subdomains with the most elements per workgroup
      int best = 0; double v = -1.0;
      for (int d = 0; d < ndl; ++d) if (wgt[d] / gd[d] > v) { v = wgt[d] / gd[d]; best = d; }
      ++gd[best]; ++used;
    }
    for (int d = 0; d < ndl; ++d) {
      const int n_d = A.maps.nd[d], ld = A.ld_h[d];
      if (ld != M.ld_h[d] || (ld + 127) / 128 > RES_MAX_U) return;
      const int rpw = (n_d + gd[d] - 1) / gd[d];
      for (int k = 0; k < gd[d]; ++k) {
        ResTile t{};
        t.matS = A.moff_h[d]; t.matP = M.moff_h[d];
        t.n = n_d; t.ld = ld; t.loc_off = A.maps.loc_off[d];
        t.row0 = std::min(k * rpw, n_d);
        t.nrows = std::max(0, std::min(rpw, n_d - t.row0));
        t.U = res_variant(std::max(1, (ld + 127) / 128));
        max_rows = std::max(max_rows, t.nrows);
        tiles_h.push_back(t);
      }
    }
    max_rows = std::max(8, (max_rows + 7) / 8 * 8);
    double res_el = 0.0, all_el = 0.0;
    for (auto &t : tiles_h) {
      const long long fixed = (long long)res_lds_fixed(t.U, max_rows), rowb = (long long)t.U * 128 * 8;
      if (fixed > RES_LDS_BYTES) return;
      const int cap = (int)((RES_LDS_BYTES - fixed) / rowb);
      const int needS = std::max(0, t.nrows - res_reg_rows_S(t.U)), needP = std::max(0, t.nrows - res_reg_rows_P(t.U));
      t.ldsS = std::min(needS, cap / 2);
      t.ldsP = std::min(needP, cap - t.ldsS);
      t.ldsS = std::min(needS, cap - t.ldsP);
      res_el += (double)(std::min(t.nrows, res_reg_rows_S(t.U) + t.ldsS) + std::min(t.nrows, res_reg_rows_P(t.U) + t.ldsP)) * t.ld;
      all_el += 2.0 * t.nrows * t.ld;
    }
    resident_frac = all_el > 0 ? res_el / all_el : 0.0;
    tiles.upload(tiles_h, c->stream);
    recs.alloc((size_t)4 * G); recs.zero(c->stream);
    abort.alloc(1); abort.zero(c->stream);
    static bool attr_set = false;
    if (!attr_set) {
      MI_HIP(hipFuncSetAttribute((const void *)k_pcg_resident, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_BYTES));
      attr_set = true;
    }
    MI_HIP(hipStreamSynchronize(c->stream));
    usable = resident_frac * 100.0 >= env_int("MI355_RES_MIN_PCT", 50);
  }
  void reset(hipStream_t s) {  // after an aborted launch
    abort.zero(s);
    epoch += 1u << 20;
  }
};
#else
struct ResidentPlan { bool usable = false; };   // (EXPERIMENTAL builds only)
#endif

struct SolverWorkspace {
  int64_t n = 0;
  int g = 1;  // workgroups of the multi-workgroup vector kernels = number of partials per dot
  // One slab for the state, the partials and the six work vectors: the latency-bound loop kernels
  // then touch a handful of pages instead of one per allocation.
  DevBuf<char> slab;
  SolverState *st = nullptr;
  double *part_pAp = nullptr, *part_rr = nullptr, *part_rz = nullptr, *part_bb = nullptr;
  double *r = nullptr, *z = nullptr, *p = nullptr, *Ap = nullptr, *x = nullptr, *b = nullptr;
  DevBuf<double> res_norm;
  // deflation
  DevBuf<double> W, AW, LU, mu, part_mu, gram;
  DevBuf<double> fold_mu, fold_wm;  // folded Def-PCG: per-tile partials of WtA*z, (W*mu) in local order
  DevBuf<double> fold_wloc;         //                  W in the local order of the blocks (nvec x nloc), gathered per solve
  DevBuf<long long> fold_dbg;       // MI355_FOLD_DEBUG=1: 64 launches x 8 stamps of one workgroup of the folded launches
  DevBuf<double> cf_pz, cf_rr, cf_rz;  // 2-launch sparse pcg: two buffers of interleaved (p, z) pairs, per-block partials
  DevBuf<int> piv;
  int nvec_cap = 0;
  // eigCG family (eig_solvers.hpp): search space V (n x spdim), A*V (eigpcg), rotation target, tvec, VtAV, ...
  DevBuf<double> eV, eAV, eVtmp, etvec, eT, eG, epart, eLUw, emu2;
  DevBuf<int> epivw;
  DevBuf<EigState> ees;
  int e_spdim = 0, e_nvec = 0;
  bool e_av = false;
  PinnedFlags *flags = nullptr;    // 2 slots, pinned
  SolveArgs *args = nullptr;       // pinned: per-call arguments of a whole-solve graph (k_solve_begin_g)
  DevBuf<SolveArgs> args_dev;      // its device copy for k_solve_end_g, + the exit kernel's workgroup counter
  DevBuf<unsigned> end_count;
  unsigned long long seq = 0;
  // MI355_SOLVE_STATS=1: every 64 whole-graph solves, the average device span (entry kernel's start .. hand-over) and the
  // average host period between hand-overs go to stderr (period - span = host turnaround + launch latency)
  bool stats_on = env_int("MI355_SOLVE_STATS", 0) != 0;
  long long stat_span = 0;
  int stat_n = 0;
  std::chrono::steady_clock::time_point stat_t0;
  void stat(long long span_ticks) {
    const auto now = std::chrono::steady_clock::now();
    if (stat_n == 0) stat_t0 = now;
    else stat_span += span_ticks;
    if (++stat_n == 65) {
      const double period = std::chrono::duration<double, std::micro>(now - stat_t0).count() / 64.0;
      std::fprintf(stderr, "[mi355 solve stats] n=%lld: device span %.2f us, host period %.2f us\n", (long long)n, stat_span * 0.01 / 64.0, period);
      stat_n = 0; stat_span = 0;
    }
  }
  double *res_stage = nullptr;     // pinned landing zone for short residual histories
  hipEvent_t ev[2] = {nullptr, nullptr};
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::map<GraphKey, int> predicted;  // loop iterations the last solve with this (A, M, nvec) took
  std::map<GraphKey, bool> zero_x0;   // ... and whether its initial guess was identically zero (whole-solve graphs)
  std::map<std::pair<const Operator *, const Operator *>, std::unique_ptr<ResidentPlan>> resident;  // per (S, ΠS) pair

  explicit SolverWorkspace(int64_t n_) : n(n_), g(vec_grid(n_)) {
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t vec = pad(sizeof(double) * ((size_t)n + 2)), part = pad(sizeof(double) * MAX_PARTS);
    const size_t total = pad(sizeof(SolverState)) + 4 * part + 6 * vec;
    slab.alloc(total);
    memset_sync(slab.p, 0, total);
    char *q = slab.p;
    st = (SolverState *)q; q += pad(sizeof(SolverState));
    part_pAp = (double *)q; q += part; part_rr = (double *)q; q += part;
    part_rz = (double *)q; q += part; part_bb = (double *)q; q += part;
    r = (double *)q; q += vec; z = (double *)q; q += vec; p = (double *)q; q += vec;
    Ap = (double *)q; q += vec; x = (double *)q; q += vec; b = (double *)q; q += vec;
    MI_HIP(hipHostMalloc((void **)&flags, 2 * sizeof(PinnedFlags)));
    std::memset(flags, 0, 2 * sizeof(PinnedFlags));
    MI_HIP(hipHostMalloc((void **)&args, sizeof(SolveArgs)));
    if (env_int("MI355_FOLD_DEBUG", 0)) { fold_dbg.alloc(512); memset_sync(fold_dbg.p, 0, 512 * sizeof(long long)); }
    args_dev.alloc(1); end_count.alloc(1);
    memset_sync(end_count.p, 0, sizeof(unsigned));
    MI_HIP(hipHostMalloc((void **)&res_stage, RES_STAGE * sizeof(double)));
    for (auto &e : ev) MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ~SolverWorkspace() {
    drop_graphs();
    if (flags) (void)hipHostFree(flags);
    if (args) (void)hipHostFree(args);
    if (res_stage) (void)hipHostFree(res_stage);
    for (auto &e : ev) if (e) (void)hipEventDestroy(e);
  }
  void drop_graphs() {
    for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
  }
  void drop_graphs_of(const Operator *op) {
    for (auto it = graphs.begin(); it != graphs.end();)
      if (it->first.A == op || it->first.M == op) { (void)hipGraphExecDestroy(it->second); it = graphs.erase(it); }
      else ++it;
    for (auto it = predicted.begin(); it != predicted.end();)
      if (it->first.A == op || it->first.M == op) it = predicted.erase(it);
      else ++it;
    for (auto it = zero_x0.begin(); it != zero_x0.end();)
      if (it->first.A == op || it->first.M == op) it = zero_x0.erase(it);
      else ++it;
    for (auto it = resident.begin(); it != resident.end();)
      if (it->first.first == op || it->first.second == op) it = resident.erase(it);
      else ++it;
  }
  void ensure_deflation(int nvec) {
    if (nvec <= nvec_cap) return;
    drop_graphs();  // buffers move
    W.alloc((size_t)n * nvec); AW.alloc((size_t)n * nvec);
    LU.alloc((size_t)nvec * nvec); gram.alloc((size_t)nvec * nvec);
    mu.alloc(nvec); piv.alloc(nvec); part_mu.alloc((size_t)nvec * MAX_PARTS);
    nvec_cap = nvec;
  }
  void ensure_eig(int spdim, int nvec, bool need_av) {
    if (spdim <= e_spdim && nvec <= e_nvec && (!need_av || e_av)) return;
    drop_graphs();
    spdim = std::max(spdim, e_spdim); nvec = std::max(nvec, e_nvec); need_av = need_av || e_av;
    eV.alloc((size_t)n * spdim); eVtmp.alloc((size_t)n * 2 * nvec); etvec.alloc((size_t)n);
    if (need_av) eAV.alloc((size_t)n * spdim);
    eT.alloc((size_t)spdim * spdim); eG.alloc((size_t)spdim * 2 * nvec); epart.alloc((size_t)2 * nvec * MAX_PARTS);
    eLUw.alloc((size_t)nvec * nvec); emu2.alloc(nvec); epivw.alloc(nvec); ees.alloc(1);
    e_spdim = spdim; e_nvec = nvec; e_av = need_av;
  }
};

inline SolverWorkspace &workspace(mi_ctx_s *ctx, int64_t n) {
  auto &slot = ctx->workspaces[n];
  if (!slot) slot.reset(new SolverWorkspace(n));
  return *slot;
}

// LAPACK getrf (unblocked, partial pivoting) on a column-major copy. Returns 0 or k+1 for U[k,k]==0.
inline int host_lu(int n, std::vector<double> &a, std::vector<int> &piv) {
  piv.resize(n);
  for (int k = 0; k < n; ++k) {
    int p = k;
    double mx = std::fabs(a[k + (size_t)k * n]);
    for (int i = k + 1; i < n; ++i) {
      const double v = std::fabs(a[i + (size_t)k * n]);
      if (v > mx) { mx = v; p = i; }
    }
    piv[k] = p;
    if (a[p + (size_t)k * n] == 0.0 || !std::isfinite(a[p + (size_t)k * n])) return k + 1;
    if (p != k)
      for (int j = 0; j < n; ++j) std::swap(a[k + (size_t)j * n], a[p + (size_t)j * n]);
    const double piv_inv = 1.0 / a[k + (size_t)k * n];
    for (int i = k + 1; i < n; ++i) a[i + (size_t)k * n] *= piv_inv;
    for (int j = k + 1; j < n; ++j) {
      const double akj = a[k + (size_t)j * n];
      for (int i = k + 1; i < n; ++i) a[i + (size_t)j * n] -= a[i + (size_t)k * n] * akj;
    }
  }
  return 0;
}

struct Krylov {
  mi_ctx_s *ctx;
  Operator *A, *M;  // M == nullptr: unpreconditioned (z is r)
  SolverWorkspace &ws;
  int nvec;
  int n, g;
  hipStream_t s;
  bool fused;  // single-workgroup loop kernels (small Γ systems)
  bool fold;   // pcg with both operators dense on the same maps: vector work folded into the two GEMVs
  CsrDev *Ac = nullptr;  // cg / pcg with a diagonal M on a plain sparse matrix: the 2-launch loop (k_spmv_pcg, k_update_xr_blk)
  bool csrfold() const { return Ac != nullptr && !eig.tag; }
  DenseBlockOp *Ad = nullptr, *Md = nullptr;
  int capture_whole = 0;  // while a whole-solve graph is being captured: 1 general entry, 2 entry for x0 == 0 (folded loop)
  // eigCG family: recording kernels after every iteration (eig_solvers.hpp fills this in)
  struct EigHook {
    int tag = 0, spdim = 0;
    bool has_tvec = false;   // eigcg / eigpcg keep tvec and the coupling column
    bool project_r = false;  // eigdefpcg: r .-= W * (WtW \ (W' * r)) before rTr (defcg.jl:411)
  } eig;
  std::vector<double> gram_host;  // WtAW as computed (before LU), kept for VtAV[1:nvec,1:nvec] (defcg.jl:158 / 391)

  Krylov(mi_ctx_s *c, Operator *A_, Operator *M_, int nvec_, bool generic = false, bool allow_fold = true)
      : ctx(c), A(A_), M(M_), ws(workspace(c, A_->n)), nvec(nvec_), n((int)A_->n), g(ws.g), s(c->stream),
        fused(!generic && A_->n <= FUSED_MAX_N && !env_int("MI355_NO_FUSED", 0)), fold(false) {
    // (the folded launches themselves are multi-workgroup: beyond FUSED_MAX_N only the start-up differs — the
    // reference's own partitions, 80-500 subdomains, have n_Γ of 10-40 k; undeflated solves only there)
    const bool big_fold = !fused && !generic && nvec_ == 0 && !env_int("MI355_NO_FUSED", 0) && !env_int("MI355_NO_BIG_FOLD", 0);
    if ((fused || big_fold) && allow_fold && M && nvec <= 64 && !env_int("MI355_NO_FOLD", 0) && !(nvec > 0 && env_int("MI355_NO_FOLD_DEFL", 0))) {
      Ad = A->as_dense(); Md = M->as_dense();
      // multi-GPU: S may be sharded (built on the maps of all subdomains, inactive tiles for the other ranks' blocks)
      // while the Neumann-Neumann blocks are replicated: the S launch is then followed by one all-reduce
      // ... or sharded as well (both on the maps of all subdomains): then the ΠS launch is followed by an exchange too
      fold = Ad && Md && (!Ad->reduce_over_ranks || (Ad->full_maps && nvec == 0)) &&
             (!Md->reduce_over_ranks || (Md->full_maps && Md->fold_p1_off && nvec == 0 && !env_int("MI355_NO_FOLD_SHARDED_NN", 0))) &&
             !Ad->scale && Md->scale && Ad->same_maps(*Md) && Ad->ntiles > 0 && Ad->max_ld <= GEMV_PANEL && Ad->maps.slot_width <= 4 &&
             (Ad->max_ld + 64 * Ad->waves - 1) / (64 * Ad->waves) <= 8 && (Md->max_ld + 64 * Md->waves - 1) / (64 * Md->waves) <= 8;
    }
    const double *dv = nullptr;
    if (!fused && nvec == 0 && A->as_csr() && A->as_csr()->nblocks > 0 && (!M || M->diag_kind(&dv) != 0) && !env_int("MI355_NO_CSRFOLD", 0))
      Ac = A->as_csr();
  }
  // pcg on one GPU with both operators dense on the same maps: the whole solve as one persistent launch with the blocks
  // held in registers / LDS (resident.hpp), when enough of them fit on the chip. nullptr: not applicable.
  ResidentPlan *resident_plan() {
#ifndef MI355_EXPERIMENTAL
    return nullptr;
#else
    if (!fold || nvec > 0 || eig.tag || ctx->has_comm() || Ad->reduce_over_ranks || !env_int("MI355_RESIDENT", 0) || env_int("MI355_NO_RESIDENT", 0)) return nullptr;
    auto &slot = ws.resident[{A, M}];
    if (!slot) slot.reset(new ResidentPlan(ctx, *Ad, *Md));
    return slot->usable ? slot.get() : nullptr;
#endif
  }
  // peer exchange with the waits inside the launches: every sharded operator of the loop stores into the arenas itself
  bool inwait() const {
    return fold && ctx->peer_inwait && ctx->use_peer() && (Ad->reduce_over_ranks || Md->reduce_over_ranks) &&
           (!Ad->reduce_over_ranks || (Ad->xt_on() && Ad->xt_direct)) && (!Md->reduce_over_ranks || (Md->xt_on() && Md->xt_direct));
  }
  PcgFold fold_args(int phase) const {
    PcgFold f{};
    const size_t nl = (size_t)Ad->maps.nloc;
    f.st = ws.st; f.W = Ad->maps.slot_width; f.res_norm = ws.res_norm.p; f.x = ws.x; f.r_gamma = ws.r;
    f.r_cur = Ad->fold_vec.p; f.r_nxt = f.r_cur + nl; f.p_cur = f.r_nxt + nl; f.p_nxt = f.p_cur + nl;
    f.tgt = Ad->maps.tgt.p; f.peer = Ad->maps.peer.p; f.jrank = Ad->maps.jrank.p;
    // partial-dot arrays of the OTHER operator's launch (tilings may differ); a sharded S writes one product per row
    f.nvec = nvec; f.n_gamma = n;
    if (nvec > 0) { f.AW = ws.AW.p; f.part_mu = ws.fold_mu.p; f.wm_loc = ws.fold_wm.p; }
    if (capture_whole && phase == 0) { f.exit_args = ws.args_dev.p; f.exit_flags = &ws.flags[0]; }
    if (ws.fold_dbg.p) { f.dbg = ws.fold_dbg.p; f.dbg_wg = env_int("MI355_FOLD_DEBUG_WG", 0); }
    // a sharded launch's outputs are exchanged over the ranks before the other launch reads them (reduced copy of the pack;
    // with the peer exchange a double-buffered table whose parity the reading launch takes from the exchange counter)
    const bool redA = Ad->reduce_over_ranks, redM = Md->reduce_over_ranks;
    f.n_in = phase ? Ad->part_total : Md->part_total;
    if (phase) {  // ΠS launch: reads S contributions + partial p'Ap, writes ΠS contributions + partial r'r, r'z
      f.con_in = Ad->fold_con(redA);
      f.part_in0 = Ad->fold_part0(redA); f.part_in1 = nullptr;
      Md->fold_outputs(f);
      if (redA && Ad->xt_on()) { f.in_epoch = &ctx->peer->st->epoch; f.in_stride = (long long)Ad->xt_copy; }
    } else {      // S launch: reads ΠS contributions + partial r'r, r'z, writes S contributions + partial p'Ap
      f.con_in = Md->fold_con(redM);
      f.part_in0 = Md->fold_part0(redM); f.part_in1 = Md->fold_part1p(redM);
      Ad->fold_outputs(f);
      if (redM && Md->xt_on()) { f.in_epoch = &ctx->peer->st->epoch; f.in_stride = (long long)Md->xt_copy; }
    }
    if (inwait()) { f.x_inwait = 1; f.xst = ctx->peer->st; f.xpw = ctx->peer->peers_dev; }
    return f;
  }

  const int *done() const { return &ws.st->done; }

  void dot_partial(const double *x, const double *y, double *part, const int *dn) {
    hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(NT), 0, s, n, x, y, part, dn);
    MI_HIP(hipGetLastError());
  }
  // mu = WtAW \ (V' v), V = AW (loop) or W (set-up)
  void project(const double *V, const double *v, const int *dn) { project(V, v, dn, ws.LU.p, ws.piv.p, ws.mu.p); }
  void project(const double *V, const double *v, const int *dn, const double *LU, const int *piv, double *mu) {
    hipLaunchKernelGGL(k_multi_dot_partial, dim3(g, nvec), dim3(NT), 0, s, n, V, v, ws.part_mu.p, dn);
    hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(64), sizeof(double) * (nvec <= 64 ? nvec + nvec * nvec + nvec / 2 + 1 : nvec), s, nvec, LU, piv,
                       ws.part_mu.p, g, mu, dn);
    MI_HIP(hipGetLastError());
  }
  // z (what p is updated with) and Ap of the current iteration as the loop in use leaves them: slot views in the fused
  // loop, plain vectors in the multi-workgroup one
  AsmView z_view() const {
    if (!M) return AsmView{ws.r, 0};
    return fused && nvec == 0 ? M->view_of(ws.z) : AsmView{ws.z, 0};
  }
  AsmView Ap_view() const { return fused ? A->view_of(ws.Ap) : AsmView{ws.Ap, 0}; }
  // Lanczos bookkeeping of the eigCG family for the iteration just enqueued
  void eig_record() {
    const int pre = M != nullptr;
    double *tv = eig.has_tvec ? ws.etvec.p : nullptr;
    hipLaunchKernelGGL(k_eig_vec, dim3(g), dim3(NT), 0, s, n, ws.st, ws.ees.p, pre, eig.spdim, z_view(), Ap_view(), ws.eV.p, tv);
    if (eig.has_tvec)
      hipLaunchKernelGGL(k_eig_coupling, dim3(g, 2 * ws.e_nvec), dim3(NT), 0, s, n, ws.st, ws.ees.p, ws.eV.p, ws.etvec.p,
                         ws.epart.p);
    hipLaunchKernelGGL(k_eig_state, dim3(1), dim3(64), 0, s, ws.st, ws.ees.p, ws.eT.p, eig.spdim, ws.epart.p, g,
                       (int)eig.has_tvec);
    MI_HIP(hipGetLastError());
  }

#define MI_EPT_DISPATCH(CALL)            \
  do {                                   \
    if (n <= NTF) { CALL(1); }           \
    else if (n <= 2 * NTF) { CALL(2); }  \
    else if (n <= 4 * NTF) { CALL(4); }  \
    else { CALL(8); }                    \
  } while (0)

  // One loop iteration (cg.jl:35-47 / 92-106; defcg.jl:68-80 / 291-305), enqueued on the stream.
  void iteration() {
    const int *dn = done();
    const int pre = M != nullptr;
    const double *Wp = nvec > 0 ? ws.W.p : nullptr, *mup = nvec > 0 ? ws.mu.p : nullptr;
    if (fold) {
      // 2 launches per iteration: (alpha, x, r, z, r'z, r'r) in the ΠS GEMV; (stop rule, beta, p, Ap, p'Ap) in the S GEMV
      Md->gemv_pcg(1, fold_args(1));
      if (Md->reduce_over_ranks) Md->reduce_fold(dn, inwait());  // ΠS contributions + partial r'r, r'z: union over the ranks
      if (nvec > 0) {  // mu = WtAW \ (WtA * z); W*mu in local order for the S launch (defcg.jl:301-303)
        hipLaunchKernelGGL(k_defl_mu, dim3((Ad->maps.nloc + 1023) / 1024), dim3(1024), 0, s, ws.st, nvec, Md->ntiles, ws.fold_mu.p,
                           ws.LU.p, ws.piv.p, ws.W.p, (long long)n, Ad->maps.nloc, Ad->maps.gidx.p, ws.fold_wm.p, ws.mu.p, ws.fold_wloc.p);
        MI_HIP(hipGetLastError());
      }
      Ad->gemv_pcg(0, fold_args(0));
      if (Ad->reduce_over_ranks) Ad->reduce_fold(dn, inwait());  // S contributions + partial p'Ap: union over the ranks
      return;
    }
    if (fused) {
      // small Γ systems: 4 launches per iteration (GEMV, fused, GEMV, fused)
      const AsmView vAp = A->apply_view(ws.p, ws.Ap, dn);             // mul!(Ap, A, p), Γ-sum deferred
      if (!pre && nvec == 0) {                                        // cg: 2 launches per iteration
#define MI_CALL(E) hipLaunchKernelGGL((k_fused_cg<E>), dim3(1), dim3(NTF), 0, s, n, ws.st, vAp, ws.p, ws.x, ws.r, ws.res_norm.p)
        MI_EPT_DISPATCH(MI_CALL);
#undef MI_CALL
        if (eig.tag) eig_record();
        MI_HIP(hipGetLastError());
        return;
      }
#define MI_CALL(E) hipLaunchKernelGGL((k_fused_xr<E>), dim3(1), dim3(NTF), 0, s, n, ws.st, vAp, ws.p, ws.x, ws.r, pre)
      MI_EPT_DISPATCH(MI_CALL);                                       // alpha; x += alpha p; r -= alpha Ap; r'r
#undef MI_CALL
      AsmView vz{ws.r, 0};
      if (pre) vz = M->apply_view(ws.r, ws.z, dn);                    // z .= M \ r, Γ-sum deferred
      const bool wave_lu = nvec > 0 && nvec <= 64;                    // mu solved inside k_fused_p by one wave
      if (wave_lu) {
#define MI_CALL(E) hipLaunchKernelGGL((k_multi_dot_view<E>), dim3(nvec), dim3(NTF), 0, s, n, ws.AW.p, vz, ws.part_mu.p, dn)
        MI_EPT_DISPATCH(MI_CALL);                                     // WtA * z
#undef MI_CALL
      } else if (nvec > 0) {
        if (vz.width) {  // materialise z for the generic projection kernels
          hipLaunchKernelGGL(k_assemble_slots, dim3(vec_grid(n)), dim3(NT), 0, s, n, vz.width, vz.src, ws.z, dn);
          vz = AsmView{ws.z, 0};
        }
        project(ws.AW.p, vz.src, dn);                                 // mu .= WtAW \ (WtA * z)
      }
      const double *lup = wave_lu ? ws.LU.p : nullptr;
      const int *pivp = wave_lu ? ws.piv.p : nullptr;
#define MI_CALL(E) hipLaunchKernelGGL((k_fused_p<E>), dim3(1), dim3(NTF), 0, s, n, ws.st, vz, ws.r, ws.p, Wp, mup, nvec, ws.res_norm.p, pre, lup, pivp, ws.part_mu.p)
      MI_EPT_DISPATCH(MI_CALL);                                       // beta; [mu;] p; it += 1; res_norm[it]; stop rule
#undef MI_CALL
      if (eig.tag) eig_record();
      MI_HIP(hipGetLastError());
      return;
    }
    if (csrfold()) {
      // sparse A, diagonal or no M: 2 launches per iteration (kernels.hpp, "pcg on a sparse matrix in 2 launches")
      const double *dinv = nullptr;
      const int diag = pre ? M->diag_kind(&dinv) : 0;
      const int grid = ((Ac->nblocks + 7) / 8) * 8;
      double *pz0 = ws.cf_pz.p, *pz1 = pz0 + 2 * (size_t)n;
      CsrOp *co = static_cast<CsrOp *>(A);
      hipLaunchKernelGGL(k_spmv_pcg, dim3(grid), dim3(NT), 0, s, Ac->nblocks, Ac->blk.p, Ac->rowptr.p, Ac->col.p, Ac->val.p, ws.st,
                         ws.cf_rr.p, ws.cf_rz.p, CF_VEC_GRID, pz0, pz1, ws.Ap, co->dot_part.p, ws.res_norm.p, pre);
      hipLaunchKernelGGL(k_update_xr_blk, dim3(CF_VEC_GRID), dim3(NT), 0, s, Ac->xcd_row.p, ws.st, co->dot_part.p, Ac->nblocks, pz0,
                         pz1, ws.Ap, ws.x, ws.r, dinv, diag, pre, ws.cf_rr.p, ws.cf_rz.p);
      MI_HIP(hipGetLastError());
      return;
    }
    // large systems (full A): SpMV with the p'Ap partials in its epilogue; r-update with a diagonal M folded in
    const double *part = ws.part_pAp;
    int npart = g;
    if (!A->apply_dot(ws.p, ws.Ap, ws.p, &part, &npart, dn)) {       // mul!(Ap, A, p); d = dot(p, Ap)
      A->apply(ws.p, ws.Ap, dn);
      dot_partial(ws.p, ws.Ap, ws.part_pAp, dn);
      part = ws.part_pAp; npart = g;
    }
    const double *dinv = nullptr;
    const int diag = pre && !eig.project_r ? M->diag_kind(&dinv) : 0;
    hipLaunchKernelGGL(k_update_xr, dim3(g), dim3(NT), 0, s, n, ws.st, part, npart, ws.p, ws.Ap, ws.x, ws.r,
                       ws.part_rr, pre, diag, dinv, ws.z, ws.part_rz);  // alpha; x += alpha p; r -= alpha Ap; r'r [; z; r'z]
    if (eig.project_r) {                                                // r .-= W * (WtW \ (W' * r)); rTr = dot(r, r)
      project(ws.W.p, ws.r, dn, ws.eLUw.p, ws.epivw.p, ws.emu2.p);
      hipLaunchKernelGGL(k_project_r, dim3(g), dim3(NT), 0, s, n, ws.r, ws.W.p, ws.emu2.p, nvec, ws.part_rr, dn);
    }
    const double *zz = ws.r;
    if (pre) {
      if (!diag) {
        M->apply(ws.r, ws.z, dn);                                     // z .= M \ r
        dot_partial(ws.r, ws.z, ws.part_rz, dn);                      // rTz = dot(r, z)
      }
      zz = ws.z;
    }
    if (nvec > 0) project(ws.AW.p, zz, dn);                           // mu .= WtAW \ (WtA * z)
    hipLaunchKernelGGL(k_update_p, dim3(g), dim3(NT), 0, s, n, ws.st, ws.part_rr, ws.part_rz, g, zz, ws.p, Wp, mup, nvec,
                       ws.res_norm.p, pre);                           // beta; p; it += 1; res_norm[it]; stop rule
    if (eig.tag) eig_record();
    MI_HIP(hipGetLastError());
  }

  // r = b - A x; z = M \ r; p = z [- W mu]; it = 1; res_norm[1]; tol (cg.jl:26-33 / 81-89; defcg.jl:56-66 / 277-288).
  // eps / maxit / res_cap are read from the state block (uploaded by solve()), so this is graph-replayable.
  void setup_tail() {
    const int pre = M != nullptr;
    struct Hint {  // `A*x0` may skip streaming A when the device flag says x0 == 0 (k_solve_begin)
      Operator *op;
      Hint(Operator *o, const int *flag) : op(o) { op->zero_hint = flag; }
      ~Hint() { op->zero_hint = nullptr; }
    };
    if (fold && !fused) {   // n_Γ > FUSED_MAX_N: multi-workgroup residual, then the scalars of the folded loop
      { Hint h(A, &ws.st->x0_zero); A->apply(ws.x, ws.Ap, nullptr); }
      hipLaunchKernelGGL(k_residual, dim3(g), dim3(NT), 0, s, n, ws.b, ws.Ap, ws.r, ws.part_rr, ws.part_bb);
      hipLaunchKernelGGL(k_fold_start, dim3(1), dim3(NT), 0, s, ws.st, ws.part_rr, ws.part_bb, g);
      MI_HIP(hipGetLastError());
      if (inwait()) ctx->peer->begin_inwait(s);
      return;
    }
    if (fold) {
      AsmView vAp;
      { Hint h(A, nvec == 0 ? &ws.st->x0_zero : nullptr); vAp = A->apply_view(ws.x, ws.Ap, nullptr); }  // deflated: x0 was updated by W*mu
#define MI_CALL(E) hipLaunchKernelGGL((k_fused_residual<E, true>), dim3(1), dim3(NTF), 0, s, n, ws.st, vAp, ws.b, ws.r)
      MI_EPT_DISPATCH(MI_CALL);
#undef MI_CALL
      MI_HIP(hipGetLastError());
      if (inwait()) ctx->peer->begin_inwait(s);
      return;
    }
    if (fused && nvec == 0) {
      AsmView vAp;
      { Hint h(A, &ws.st->x0_zero); vAp = A->apply_view(ws.x, ws.Ap, nullptr); }
#define MI_CALL(E) hipLaunchKernelGGL((k_fused_residual<E, false>), dim3(1), dim3(NTF), 0, s, n, ws.st, vAp, ws.b, ws.r)
      MI_EPT_DISPATCH(MI_CALL);
#undef MI_CALL
      const AsmView vz = pre ? M->apply_view(ws.r, ws.z, nullptr) : AsmView{ws.r, 0};
#define MI_CALL(E) hipLaunchKernelGGL((k_fused_start<E>), dim3(1), dim3(NTF), 0, s, n, ws.st, vz, ws.r, ws.p, ws.res_norm.p, pre)
      MI_EPT_DISPATCH(MI_CALL);
#undef MI_CALL
      MI_HIP(hipGetLastError());
      return;
    }
    { Hint h(A, nvec == 0 ? &ws.st->x0_zero : nullptr); A->apply(ws.x, ws.Ap, nullptr); }  // deflated: x0 was updated by W*mu
    hipLaunchKernelGGL(k_residual, dim3(g), dim3(NT), 0, s, n, ws.b, ws.Ap, ws.r, ws.part_rr, ws.part_bb);
    if (csrfold()) {
      const double *dinv = nullptr;
      const int diag = pre ? M->diag_kind(&dinv) : 0;
      if (diag == 2) {
        M->apply(ws.r, ws.z, nullptr);
        dot_partial(ws.r, ws.z, ws.part_rz, nullptr);
      }
      hipLaunchKernelGGL(k_csrfold_start, dim3(g), dim3(NT), 0, s, n, CF_VEC_GRID, ws.st, ws.part_rr, ws.part_bb,
                         diag == 2 ? ws.part_rz : (const double *)nullptr, g, diag == 2 ? ws.z : ws.r, ws.cf_pz.p, ws.cf_rr.p,
                         ws.cf_rz.p);
      MI_HIP(hipGetLastError());
      return;
    }
    const double *zz = ws.r;
    if (pre) {
      M->apply(ws.r, ws.z, nullptr);
      dot_partial(ws.r, ws.z, ws.part_rz, nullptr);
      zz = ws.z;
    }
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(NT), 0, s, ws.st, ws.part_rr, ws.part_bb,
                       pre ? ws.part_rz : (const double *)nullptr, g, ws.res_norm.p);
    if (nvec > 0) project(ws.AW.p, zz, nullptr);
    hipLaunchKernelGGL(k_init_p, dim3(g), dim3(NT), 0, s, n, zz, ws.p, nvec > 0 ? ws.W.p : (const double *)nullptr,
                       nvec > 0 ? ws.mu.p : (const double *)nullptr, nvec);
    MI_HIP(hipGetLastError());
  }

  // chunk > 0: `chunk` iterations; chunk < 0: set-up tail + (-chunk) iterations; whole: the entry kernel in front of
  // and the exit kernel behind them (arguments through ws.args), so that a solve is ONE replay.
  // (Splitting a whole-solve graph into a short head and the rest, so that the GPU starts while the host still enqueues,
  // was measured: 432 vs 427 us per solve — one replay is cheaper than two.)
  hipGraphExec_t graph(int chunk, int whole = 0) {
    GraphKey key{A, M, nvec, chunk, eig.tag * 8 + (fused ? 1 : 0) + (fold ? 2 : 0) + (csrfold() ? 4 : 0) + 1024 * whole};  // the loop form is part of the graph
    auto it = ws.graphs.find(key);
    if (it != ws.graphs.end()) return it->second;
    hipGraph_t gr = nullptr;
    MI_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try {
      capture_whole = whole;
      if (whole == 2) {   // folded loop, x0 == 0: one entry launch
#define MI_CALL(E) hipLaunchKernelGGL((k_entry_zero<E>), dim3(1), dim3(NTF), 0, s, n, ws.args, ws.b, ws.x, ws.r, ws.st, ws.args_dev.p, &ws.flags[0])
        MI_EPT_DISPATCH(MI_CALL);
#undef MI_CALL
        MI_HIP(hipGetLastError());
      } else {
        if (whole) {
          hipLaunchKernelGGL(k_solve_begin_g, dim3(g), dim3(NT), 0, s, n, ws.args, ws.b, ws.x, ws.st, ws.args_dev.p, &ws.flags[0]);
          MI_HIP(hipGetLastError());
        }
        if (chunk < 0) setup_tail();
      }
      for (int k = 0; k < std::abs(chunk); ++k) iteration();
      capture_whole = 0;
      if (whole) {
        hipLaunchKernelGGL(k_solve_end_g, dim3(g), dim3(NT), 0, s, n, ws.st, ws.args_dev.p, (int)(fold || csrfold()), ws.x, ws.res_norm.p,
                           &ws.flags[0], &ws.st->x0_zero, ws.end_count.p);
        MI_HIP(hipGetLastError());
      }
    } catch (...) {
      capture_whole = 0;
      (void)hipStreamEndCapture(s, &gr);
      if (gr) (void)hipGraphDestroy(gr);
      throw;
    }
    MI_HIP(hipStreamEndCapture(s, &gr));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gr);
    if (e != hipSuccess) raise(MI_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    ws.graphs[key] = ex;
    return ex;
  }

  // Wait for the exit kernel of a whole-solve graph: it stores the solve number after everything else it wrote has been
  // released to system scope. A bounded spin (a replay that takes longer, or a fault, ends in a stream synchronisation).
  void wait_seq(unsigned long long seq) {
    volatile unsigned long long *f = &ws.flags[0].seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned k = 1;; ++k) {
      if (*f == seq) { std::atomic_thread_fence(std::memory_order_acquire); return; }
      if ((k & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
      __builtin_ia32_pause();
    }
    MI_HIP(hipStreamSynchronize(s));
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*f != seq) raise(MI_ERR_HIP, "internal: the solve's exit kernel did not report (seq %llu, expected %llu)", (unsigned long long)*f, seq);
  }

  void fetch_flags(int slot) {
    MI_HIP(hipMemcpyAsync(&ws.flags[slot].it, (fold || csrfold()) ? &ws.st->it_nxt : &ws.st->it, sizeof(long long),
                          hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(&ws.flags[slot].done, &ws.st->done, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    MI_HIP(hipEventRecord(ws.ev[slot], s));
  }

  // Common entry of every solver: b, x to the workspace; eps / maxit / res_cap to the state block; deflated
  // solvers: W, WtA, WtAW (LU on the host) and the deflated initial guess (defcg.jl:40-54 / 260-275).
  void begin(const double *b_in, const double *x_in, const double *W_in, int64_t &maxit, double &eps, int64_t &cap_dev) {
    prepare(maxit, eps, cap_dev);
    const size_t vb = sizeof(double) * (size_t)n;
    hipLaunchKernelGGL(k_solve_begin, dim3(g), dim3(NT), 0, s, n, b_in, x_in, ws.b, ws.x, ws.st, eps, (long long)maxit,
                       (long long)cap_dev);
    MI_HIP(hipGetLastError());
    if (nvec > 0) {
      MI_HIP(hipMemcpyAsync(ws.W.p, W_in, vb * nvec, hipMemcpyDeviceToDevice, s));
      if (fold) {
        hipLaunchKernelGGL(k_gather_w_loc, dim3((Ad->maps.nloc + NT - 1) / NT, nvec), dim3(NT), 0, s, nvec, (long long)n, Ad->maps.nloc,
                           Ad->maps.gidx.p, ws.W.p, ws.fold_wloc.p);
        MI_HIP(hipGetLastError());
      }
      A->apply_multi(ws.W.p, n, nvec, ws.AW.p, n);                           // WtA[v,:] = A*W[:,v]
      hipLaunchKernelGGL(k_small_gram, dim3(nvec, nvec), dim3(NT), 0, s, n, ws.AW.p, ws.W.p, ws.gram.p, nvec);  // WtAW
      MI_HIP(hipGetLastError());
      factor(ws.gram.p, ws.LU, ws.piv, &gram_host, "WtAW");
      A->apply(ws.x, ws.Ap, nullptr);                                        // r .= b .- A*x
      hipLaunchKernelGGL(k_residual, dim3(g), dim3(NT), 0, s, n, ws.b, ws.Ap, ws.r, ws.part_rr, ws.part_bb);
      project(ws.W.p, ws.r, nullptr);                                        // mu = WtAW \ (W'r)
      hipLaunchKernelGGL(k_add_Wmu, dim3(g), dim3(NT), 0, s, n, ws.x, ws.W.p, ws.mu.p, nvec);  // x .+= W*mu
      MI_HIP(hipGetLastError());
    }
  }
  // host half of the entry: defaults, capacities, buffers
  void prepare(int64_t &maxit, double &eps, int64_t &cap_dev) {
    if (eps <= 0.0) eps = 1e-7;           // RecyclingKrylovSolvers.jl:21
    if (maxit == 0) maxit = n;            // cg.jl:25
    // reference: res_norm has n entries and `res_norm[it] = ...` throws BoundsError at it = n + 1 (cg.jl:23,47), which
    // only maxit > n can reach: the device loop stops there too (overflow flag -> MI_ERR_RES_CAPACITY), with x updated
    // exactly as often as the reference had updated it when it threw
    cap_dev = std::max<int64_t>(1, std::min<int64_t>(maxit, (int64_t)n));
    if ((size_t)cap_dev + 1 > ws.res_norm.n) { ws.drop_graphs(); ws.res_norm.alloc((size_t)cap_dev + 1); }
    if (nvec > 0) ws.ensure_deflation(nvec);
    if (nvec > 0 && fold) {
      const size_t need_mu = (size_t)nvec * Md->ntiles + 1, need_wm = (size_t)Ad->maps.nloc + 1;
      const size_t need_wl = (size_t)nvec * Ad->maps.nloc + 1;
      if (ws.fold_mu.n < need_mu || ws.fold_wm.n < need_wm || ws.fold_wloc.n < need_wl) {
        ws.drop_graphs();  // buffers move
        ws.fold_mu.alloc(need_mu); ws.fold_wm.alloc(need_wm); ws.fold_wloc.alloc(need_wl);
      }
    }
    if (csrfold()) {
      const size_t need_pz = 4 * (size_t)n + 2, need_p = (size_t)CF_VEC_GRID + 1;
      if (ws.cf_pz.n < need_pz || ws.cf_rr.n < need_p) {
        ws.drop_graphs();  // buffers move
        ws.cf_pz.alloc(need_pz); ws.cf_rr.alloc(need_p); ws.cf_rz.alloc(need_p);
      }
    }
  }
  // LU (host, LAPACK getrf order) of the nvec x nvec device matrix `gram`; factors and pivots back to the device.
  void factor(const double *gram, DevBuf<double> &LU, DevBuf<int> &piv, std::vector<double> *keep, const char *name) {
    std::vector<double> lu((size_t)nvec * nvec);
    MI_HIP(hipMemcpyAsync(lu.data(), gram, sizeof(double) * lu.size(), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    if (keep) *keep = lu;
    std::vector<int> pv;
    const int info = host_lu(nvec, lu, pv);
    if (info) raise(MI_ERR_SINGULAR, "%s is singular: U[%d,%d] == 0 (SingularException(%d))", name, info, info, info);
    LU.upload(lu.data(), lu.size(), s);
    piv.upload(pv.data(), pv.size(), s);
    MI_HIP(hipStreamSynchronize(s));
  }

  // b_in / x_io / W_in are device pointers here (host mode is staged by the caller).
  int solve(const double *b_in, double *x_io, const double *W_in, int64_t maxit, double eps, double *res_host,
            int64_t res_cap, int64_t *it_out) {
    int64_t cap_dev = 0;
    struct AbortFlag {   // peer exchange: a wait that expires during this solve sets the stop flag (no iterating on garbage up to maxit)
      mi_ctx_s *c; hipStream_t s;
      AbortFlag(mi_ctx_s *c_, hipStream_t s_, int *flag) : c(c_), s(s_) { if (c->use_peer()) c->peer->set_abort_flag(flag, s); }
      ~AbortFlag() { if (c->use_peer()) { c->peer->set_abort_flag(nullptr, s); (void)hipStreamSynchronize(s); } }
    } abort_flag(ctx, s, &ws.st->done);
    bool use_graph = ctx->chunk > 0 && A->graph_safe() && (!M || M->graph_safe()) && !ctx->no_graph;
    // Undeflated solves on one GPU: entry kernel, set-up, iterations and exit kernel are ONE graph replay whose per-call
    // arguments travel through a pinned block, and the host waits for the exit kernel's last store instead of the stream.
    const bool whole_on = !env_int("MI355_NO_WHOLE_GRAPH", 0);
    const bool whole = whole_on && use_graph && nvec == 0 && !eig.tag && !ctx->has_comm() && !resident_plan();
    if (whole) prepare(maxit, eps, cap_dev);
    else begin(b_in, x_io, W_in, maxit, eps, cap_dev);

    // ---- set-up tail + loop
    const int64_t ncap = std::min<int64_t>(res_cap, cap_dev);
    const bool spec_res = res_host && ncap > 0 && ncap <= RES_STAGE;
    auto enqueue_results = [&](int slot) {
      hipLaunchKernelGGL(k_solve_end, dim3(g), dim3(NT), 0, s, n, ws.st, (int)(fold || csrfold()), ws.x, x_io, ws.res_norm.p,
                         spec_res ? ws.res_stage : (double *)nullptr, (long long)ncap, &ws.flags[slot], &ws.st->x0_zero);
      MI_HIP(hipGetLastError());
    };
    bool ran_resident = false;
#ifdef MI355_EXPERIMENTAL
    if (ResidentPlan *rp = resident_plan()) {
      // One persistent launch for the whole solve. It clears `done` itself only by setting it at the end: an aborted
      // launch (bounded spin expired: the workgroups were not all resident) leaves done = 0 and the loops below take over.
      ResArgs ra{};
      ra.MS = Ad->M.p; ra.MP = Md->M.p; ra.tiles = rp->tiles.p; ra.gidx = Ad->maps.gidx.p; ra.cnt = Md->cnt.p;
      ra.tgt = Ad->maps.tgt.p; ra.jrank = Ad->maps.jrank.p; ra.conS = Ad->fold_con(); ra.conP = Md->fold_con();
      ra.recs = rp->recs.p; ra.abort = rp->abort.p; ra.st = ws.st; ra.res_norm = ws.res_norm.p;
      ra.x = ws.x; ra.b = ws.b; ra.epoch0 = rp->epoch; ra.W = Ad->maps.slot_width; ra.G = rp->G; ra.max_rows = rp->max_rows;
      DevBuf<long long> dbg;
      const bool debug = env_int("MI355_RES_DEBUG", 0) != 0;
      if (debug) { dbg.alloc(512); dbg.zero(s); ra.dbg = dbg.p; ra.dbg_wg = env_int("MI355_RES_DEBUG_WG", 0); }
      hipLaunchKernelGGL(k_pcg_resident, dim3(rp->G), dim3(RES_NTH), RES_LDS_BYTES, s, ra);
      MI_HIP(hipGetLastError());
      enqueue_results(0);
      MI_HIP(hipStreamSynchronize(s));
      if (debug) {  // wall-clock stamps (100 MHz) of one workgroup: 5 for the set-up, then 8 per iteration
        std::vector<long long> h(512);
        memcpy_sync(h.data(), dbg.p, 512 * sizeof(long long), hipMemcpyDeviceToHost);
        const ResTile &tt = rp->tiles_h[ra.dbg_wg];
        std::fprintf(stderr, "[resident] wg %d: n=%d ld=%d rows=%d U=%d regS=%d regP=%d ldsS=%d ldsP=%d resident_frac=%.3f\n", ra.dbg_wg, tt.n,
                     tt.ld, tt.nrows, tt.U, RES_WAVES * res_slots_S(tt.U), RES_WAVES * res_slots_P(tt.U), tt.ldsS, tt.ldsP, rp->resident_frac);   // (+8 rows per operator in the streaming slot)
        std::fprintf(stderr, "[resident] stamps (us since start):");
        for (int k = 0; k < 512 && h[k]; ++k) std::fprintf(stderr, "%s%.2f", (k >= 5 && (k - 5) % 8 == 0) ? "\n  " : " ", (h[k] - h[0]) * 0.01);
        std::fprintf(stderr, "\n");
      }
      if (ws.flags[0].done) {
        rp->epoch += (unsigned long long)(2 * ws.flags[0].it + 4);
        ran_resident = true;
        use_graph = false;
      } else {
        rp->reset(s);
        rp->usable = false;   // this context does not get all the CUs: stay with the graph loop from now on
        int64_t mx = maxit, cd = 0; double e = eps;
        begin(b_in, x_io, W_in, mx, e, cd);   // x0, b and the state block as they were
      }
    }
#endif
    if (use_graph) {
      const GraphKey pk{A, M, nvec, 0};
      int &predicted = ws.predicted[pk];
      // Exact prediction for short solves; long ones (hundreds of iterations, counts that drift from solve to
      // solve) round down to a multiple of 32 so that only a few graph sizes are ever instantiated.
      int64_t first = predicted > 0 ? predicted : ctx->chunk;
      if (first > 64) first -= first % 32;
      first = std::max<int64_t>(1, std::min<int64_t>(first, std::min<int64_t>(maxit, 1024)));
      if (ws.graphs.size() > 48) ws.drop_graphs();
      // folded loop: the entry for x0 == 0 when the previous solve with these operators had one (checked on the device)
      bool zero_variant = whole && fold && fused && ws.zero_x0[pk] && !env_int("MI355_NO_ZERO_ENTRY", 0);
      hipGraphExec_t g0 = nullptr;
      try {
        g0 = graph(-(int)first, whole ? (zero_variant ? 2 : 1) : 0);
      } catch (const Error &) {
        // A collective that cannot be captured (communicator attached): run this context eagerly from now on.
        // Every rank takes the same path because capture fails or succeeds identically on all of them.
        if (!ctx->comm) throw;
        ctx->no_graph = true;
        use_graph = false;
      }
      if (use_graph) {
      if (whole) {
        const unsigned long long seq = ++ws.seq;
        *ws.args = SolveArgs{b_in, x_io, x_io, spec_res ? ws.res_stage : (double *)nullptr, eps, (long long)maxit, (long long)cap_dev,
                             (long long)ncap, seq};
        ++ctx->n_replays; MI_HIP(hipGraphLaunch(g0, s));
        wait_seq(seq);
        if (ws.flags[0].respec) {   // x0 was not zero after all: the general form, and no such guess next time
          zero_variant = false;
          ws.zero_x0[pk] = false;
          g0 = graph(-(int)first, 1);
          const unsigned long long seq2 = ++ws.seq;
          ws.args->seq = seq2;
          ++ctx->n_replays; MI_HIP(hipGraphLaunch(g0, s));
          wait_seq(seq2);
        }
        if (fold) ws.zero_x0[pk] = ws.flags[0].x0z != 0;
        if (ws.stats_on) ws.stat(ws.flags[0].t_exit - ws.flags[0].t_entry);
        if (ws.fold_dbg.p && fold) {   // stamps of the solve that has just finished (100 MHz ticks -> us since the first one)
          std::vector<long long> h(512);
          MI_HIP(hipStreamSynchronize(s));
          memcpy_sync(h.data(), ws.fold_dbg.p, 512 * sizeof(long long), hipMemcpyDeviceToHost);
          const long long t0 = h[0] ? h[0] : ws.flags[0].t_entry;
          std::fprintf(stderr, "[fold stamps] wg %d, entry kernel at %.2f us\n", env_int("MI355_FOLD_DEBUG_WG", 0), (ws.flags[0].t_entry - t0) * 0.01);
          for (int r = 0; r < 64 && h[r * 8]; ++r) {
            std::fprintf(stderr, "  launch %2d:", r);
            for (int k = 0; k < 7; ++k) std::fprintf(stderr, " %8.2f", h[r * 8 + k] ? (h[r * 8 + k] - t0) * 0.01 : -1.0);
            std::fprintf(stderr, "\n");
          }
          memset_sync(ws.fold_dbg.p, 0, 512 * sizeof(long long));
        }
      } else {
        ++ctx->n_replays; MI_HIP(hipGraphLaunch(g0, s));
        enqueue_results(0);
        MI_HIP(hipStreamSynchronize(s));
      }
      if (!ws.flags[0].done) {
        hipGraphExec_t ex = graph(ctx->chunk);
        const int64_t max_launch = (maxit + ctx->chunk - 1) / ctx->chunk + 2;
        int slot = 0;
        bool stop = false;
        fetch_flags(slot);
        for (int64_t l = 0; l < max_launch && !stop; ++l) {
          ++ctx->n_replays; MI_HIP(hipGraphLaunch(ex, s));
          const int prev = slot;
          slot ^= 1;
          fetch_flags(slot);
          MI_HIP(hipEventSynchronize(ws.ev[prev]));
          stop = ws.flags[prev].done != 0;
        }
        enqueue_results(0);
        MI_HIP(hipStreamSynchronize(s));
      }
      predicted = (int)std::max<long long>(1, ws.flags[0].it - ((fold || csrfold()) ? 0 : 1));  // the folded pairs check the stop rule one launch later
      }
    }
    if (!use_graph && !ran_resident) {
      setup_tail();
      for (int64_t l = 0; l < maxit + 2; ++l) {
        fetch_flags(0);
        MI_HIP(hipEventSynchronize(ws.ev[0]));
        if (ws.flags[0].done) break;
        iteration();
      }
      enqueue_results(0);
      MI_HIP(hipStreamSynchronize(s));
    }
    const long long it = ws.flags[0].it;
    if (ctx->use_peer() && ctx->peer->take_error(s))
      raise(MI_ERR_COMM, "peer exchange: a wait for the other ranks expired (MI355_PEER_TIMEOUT_MS): %s; the iterates of this solve are not valid", ctx->peer->err_text.c_str());
    if (!ws.flags[0].done) raise(MI_ERR_HIP, "internal: Krylov loop ended without the stop flag (it=%lld)", it);
    const int64_t ncopy = std::min<int64_t>(it, ncap);
    if (res_host && ncopy > 0) {
      if (spec_res) std::memcpy(res_host, ws.res_stage, sizeof(double) * ncopy);
      else {
        MI_HIP(hipMemcpyAsync(res_host, ws.res_norm.p, sizeof(double) * ncopy, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
      }
    }
    if (it_out) *it_out = it;
    if (ws.flags[0].overflow || it > res_cap)
      return fail(MI_ERR_RES_CAPACITY, "res_norm capacity %lld < it = %lld", (long long)res_cap, it);
    return MI_OK;
  }
};

}  // namespace mi
