// Operators (`A*x`, `mul!`) and preconditioners (`M \ r`) of the hot path, device resident.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <numeric>

#include "common.hpp"
#include "kernels.hpp"
#include "assembly.hpp"
#include "exchange.hpp"

namespace mi {

inline int vec_grid(int64_t n) {
  int64_t g = (n + (int64_t)NT * 4 - 1) / ((int64_t)NT * 4);
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, MAX_PARTS));
}
inline int env_int(const char *name, int dflt) {
  const char *s = std::getenv(name);
  return s && *s ? std::atoi(s) : dflt;
}
inline int to_i32(int64_t v, int64_t lo, int64_t hi, const char *what) {
  if (v < lo || v >= hi) raise(MI_ERR_BAD_ARG, "%s: index %lld outside [%lld, %lld)", what, (long long)v,
                               (long long)lo, (long long)hi);
  return (int)v;
}

// ------------------------------------------------------------------ base class
constexpr int PART_ROWS = 4;   // fewest rows a streamed tile can have (4 waves x 1 row)
struct DenseBlockOp;
struct Operator {
  mi_ctx_s *ctx;
  int64_t n;  // operator is n x n on the Γ (or full) vector space
  // Set by a solver around the `A*x0` of its set-up: device flag "x0 is identically zero" (the result is then +0
  // without streaming A). Operators that cannot use it ignore it.
  const int *zero_hint = nullptr;
  Operator(mi_ctx_s *c, int64_t n_) : ctx(c), n(n_) {}
  virtual ~Operator() = default;
  // Enqueue y = Op(x) on ctx->stream; x, y are device pointers, x != y.
  virtual void apply(const double *x, double *y, const int *done) = 0;
  // Same, but the result may be returned as a deferred "assembled view" (kernels.hpp AsmView) that the
  // consumer kernel sums on the fly; the default materialises y and returns a plain view of it.
  virtual AsmView apply_view(const double *x, double *y, const int *done) {
    apply(x, y, done);
    return AsmView{y, 0};
  }
  // The view apply_view(x, y, ...) returns, without launching anything
  virtual AsmView view_of(double *y) { return AsmView{y, 0}; }
  // Optional fusions the solver asks for: y = Op(x) together with per-workgroup partials of w'y; and, for a
  // preconditioner, "I am diagonal" (1: identity, 2: Jacobi with `*dinv`) so that z = M \ r is folded into the r-update.
  virtual bool apply_dot(const double *x, double *y, const double *w, const double **part, int *count, const int *done) {
    return false;
  }
  // Y[:, v] = Op(X[:, v]) for v < k (column-major, leading dimensions ldx, ldy). Dense block operators stream their
  // matrices once per group of columns; the default is k single applies.
  virtual void apply_multi(const double *X, int64_t ldx, int k, double *Y, int64_t ldy) {
    for (int v = 0; v < k; ++v) apply(X + (size_t)v * ldx, Y + (size_t)v * ldy, nullptr);
  }
  virtual int diag_kind(const double **dinv) const { return 0; }
  virtual DenseBlockOp *as_dense() { return nullptr; }
  virtual struct CsrDev *as_csr() { return nullptr; }  // a plain sparse matrix (config 2): rows, row blocks, values
  virtual bool graph_safe() const { return true; }  // false: apply synchronises with the host
  // true: y is written exactly once, by the last kernel of apply(), and never read — it may then be pinned host memory
  virtual bool writes_y_once() const { return true; }
  virtual void bytes(int64_t *apply_b, int64_t *dominant_b) const = 0;
  virtual void apply_dominant(const double *x) = 0;
};

// ------------------------------------------------------------------ CSR storage on the device
struct HostCsr {
  int n_rows = 0, n_cols = 0;
  std::vector<int> rowptr, col;
  std::vector<double> val;
  int64_t nnz() const { return (int64_t)col.size(); }
};
// CSR arrays given as int64 + base -> checked int32 host CSR
inline HostCsr host_csr(int64_t n_rows, int64_t n_cols, const int64_t *rowptr, const int64_t *colidx,
                        const double *val, int base) {
  if (n_rows < 0 || n_cols < 0 || !rowptr || n_rows >= INT32_MAX || n_cols >= INT32_MAX)
    raise(MI_ERR_BAD_ARG, "csr: bad shape or NULL rowptr");
  HostCsr h;
  h.n_rows = (int)n_rows; h.n_cols = (int)n_cols;
  h.rowptr.resize(n_rows + 1);
  const int64_t nnz = rowptr[n_rows] - base;
  if (rowptr[0] != base || nnz < 0 || nnz >= INT32_MAX) raise(MI_ERR_BAD_ARG, "csr: bad rowptr[0]/nnz");
  if (nnz && (!colidx || !val)) raise(MI_ERR_BAD_ARG, "csr: NULL colidx/val");
  for (int64_t i = 0; i <= n_rows; ++i) {
    const int64_t v = rowptr[i] - base;
    if (v < 0 || v > nnz || (i && v < h.rowptr[i - 1])) raise(MI_ERR_BAD_ARG, "csr: rowptr not monotone");
    h.rowptr[i] = (int)v;
  }
  h.col.resize(nnz); h.val.assign(val, val + nnz);
  for (int64_t k = 0; k < nnz; ++k) h.col[k] = to_i32(colidx[k] - base, 0, n_cols, "csr colidx");
  return h;
}
// transpose (stable: column order inside every output row is ascending source row)
inline HostCsr transpose(const HostCsr &a) {
  HostCsr t;
  t.n_rows = a.n_cols; t.n_cols = a.n_rows;
  t.rowptr.assign(t.n_rows + 1, 0);
  for (int c : a.col) t.rowptr[c + 1]++;
  std::partial_sum(t.rowptr.begin(), t.rowptr.end(), t.rowptr.begin());
  t.col.resize(a.col.size()); t.val.resize(a.val.size());
  std::vector<int> next(t.rowptr.begin(), t.rowptr.end() - 1);
  for (int r = 0; r < a.n_rows; ++r)
    for (int k = a.rowptr[r]; k < a.rowptr[r + 1]; ++k) {
      const int q = next[a.col[k]]++;
      t.col[q] = r; t.val[q] = a.val[k];
    }
  return t;
}
// append `b` below/right of `a` (block-diagonal when col_shift > 0, vertical stack when 0)
inline void append_block(HostCsr &a, const HostCsr &b, int col_shift, int new_cols) {
  const int base = (int)a.col.size();
  if (a.rowptr.empty()) a.rowptr.push_back(0);
  for (int r = 0; r < b.n_rows; ++r) a.rowptr.push_back(base + b.rowptr[r + 1]);
  for (size_t k = 0; k < b.col.size(); ++k) { a.col.push_back(b.col[k] + col_shift); a.val.push_back(b.val[k]); }
  a.n_rows += b.n_rows; a.n_cols = new_cols;
}

struct CsrDev {
  int n_rows = 0, n_cols = 0, nblocks = 0;
  int64_t nnz = 0;
  DevBuf<int> rowptr, col;
  DevBuf<SpmvBlock> blk;
  DevBuf<int> xcd_row;  // [9] first row of the row blocks that run on XCD x (x = blockIdx & 7); [8] = n_rows
  DevBuf<double> val;
  std::vector<SpmvBlock> blocks_h;
  // `breaks` (ascending row indices): a row block never crosses one of them (per-subdomain partial sums)
  void upload(const HostCsr &h, hipStream_t s, const std::vector<int> *breaks = nullptr) {
    n_rows = h.n_rows; n_cols = h.n_cols; nnz = h.nnz();
    // Greedy row blocks of <= SPMV_TILE non-zeros (a longer row stands alone). A block prefers to start at
    // an even non-zero offset (paired loads in the kernel): if the greedy end lands on an odd offset, give
    // back up to three rows to reach an even one.
    std::vector<SpmvBlock> blocks;
    int r = 0;
    size_t nb = 0;
    while (r < n_rows) {
      while (breaks && nb < breaks->size() && (*breaks)[nb] <= r) ++nb;
      const int limit = (breaks && nb < breaks->size()) ? std::min((*breaks)[nb], n_rows) : n_rows;
      int e = r + 1;
      while (e < limit && h.rowptr[e + 1] - h.rowptr[r] <= SPMV_TILE) ++e;
      if (e < limit && (h.rowptr[e] & 1))
        for (int back = 1; back <= 3 && e - back > r; ++back)
          if ((h.rowptr[e - back] & 1) == 0) { e -= back; break; }
      blocks.push_back(SpmvBlock{r, e, h.rowptr[r], h.rowptr[e]});
      r = e;
    }
    nblocks = (int)blocks.size();
    blocks_h = blocks;
    {  // rows whose blocks run on XCD x (k_spmv_csr deals block b = (blockIdx & 7) * per + (blockIdx >> 3))
      const int per = (nblocks + 7) >> 3;
      std::vector<int> xr(9, n_rows);
      for (int x = 0; x < 8; ++x) xr[x] = x * per < nblocks ? blocks[(size_t)x * per].r0 : n_rows;
      xcd_row.upload(xr, s);
    }
    std::vector<int> rp = h.rowptr;
    if (rp.empty()) rp.push_back(0);
    rowptr.upload(rp, s); col.upload(h.col, s); val.upload(h.val, s); blk.upload(blocks, s);
  }
  // y = A x (mode 0) or y = yin - A x (mode 1); with `w`: also part[b] = Σ w[r] y[r] over the rows of block b
  void launch(int mode, const double *x, const double *yin, double *y, const int *done, hipStream_t s,
              const double *w = nullptr, double *part = nullptr) const {
    if (nblocks == 0) return;
    const int grid = ((nblocks + 7) / 8) * 8;
#define MI_SPMV(M, D) hipLaunchKernelGGL((k_spmv_csr<M, D>), dim3(grid), dim3(NT), 0, s, nblocks, blk.p, rowptr.p, col.p, val.p, x, yin, y, w, part, done)
    if (mode == 0) { if (w) MI_SPMV(0, true); else MI_SPMV(0, false); }
    else           { if (w) MI_SPMV(1, true); else MI_SPMV(1, false); }
#undef MI_SPMV
    MI_HIP(hipGetLastError());
  }
  // 12 nnz + 4 (rows+1) + 8 cols (x once) + 8 rows (y)
  int64_t bytes() const { return 12 * nnz + 4 * ((int64_t)n_rows + 1) + 8 * (int64_t)n_cols + 8 * (int64_t)n_rows; }
};

// ------------------------------------------------------------------ SparseMatrixCSC `A`
struct CsrOp : Operator {
  CsrDev A;
  DevBuf<double> sink;
  CsrOp(mi_ctx_s *c, const HostCsr &h) : Operator(c, h.n_rows) {
    if (h.n_rows != h.n_cols) raise(MI_ERR_BAD_ARG, "mi_csr_create: solver operators must be square");
    A.upload(h, c->stream);
    dot_part.alloc((size_t)A.nblocks + 1);  // not inside apply_dot: that may run under stream capture
    sink.alloc((size_t)n + 1);
  }
  DevBuf<double> dot_part;
  void apply(const double *x, double *y, const int *done) override { A.launch(0, x, nullptr, y, done, ctx->stream); }
  CsrDev *as_csr() override { return &A; }
  bool apply_dot(const double *x, double *y, const double *w, const double **part, int *count, const int *done) override {
    if (A.nblocks == 0) return false;
    A.launch(0, x, nullptr, y, done, ctx->stream, w, dot_part.p);
    *part = dot_part.p; *count = A.nblocks;
    return true;
  }
  void bytes(int64_t *a, int64_t *d) const override { *a = *d = A.bytes(); }
  void apply_dominant(const double *x) override {
    A.launch(0, x, nullptr, sink.p, nullptr, ctx->stream);
  }
};

// ------------------------------------------------------------------ diagonal / identity `M`
struct DiagOp : Operator {
  DevBuf<double> dinv;
  bool identity;
  DiagOp(mi_ctx_s *c, int64_t n_, const double *d) : Operator(c, n_), identity(d == nullptr) {
    if (d) { dinv.upload(d, (size_t)n_, c->stream); MI_HIP(hipStreamSynchronize(c->stream)); }
  }
  void apply(const double *x, double *y, const int *done) override {
    hipLaunchKernelGGL(k_diag_apply, dim3(vec_grid(n)), dim3(NT), 0, ctx->stream, (int)n,
                       identity ? (const double *)nullptr : dinv.p, x, y, done);
    MI_HIP(hipGetLastError());
  }
  int diag_kind(const double **d) const override { *d = identity ? nullptr : dinv.p; return identity ? 1 : 2; }
  void bytes(int64_t *a, int64_t *d) const override { *a = *d = (identity ? 16 : 24) * n; }
  void apply_dominant(const double *) override {}
};

// ------------------------------------------------------------------ local-to-Γ bookkeeping shared by the Schur ops
struct LocalMaps {
  int ndl = 0;   // local subdomains on this rank
  int nloc = 0;  // Σ n_Γd over local subdomains
  std::vector<int> nd, loc_off, gidx_h;
  DevBuf<int> gidx, aptr, apos, out_pos;
  bool sharded = false;  // only a slice of the subdomains lives here and a communicator exists: Γ-sums are all-reduced
  DevBuf<int> jrank, peer, tgt;  // local-order bookkeeping of the folded PCG launches (kernels.hpp PcgFold)
  std::vector<int> tgt_h;        // host copy of tgt (the peer exchange lists a rank's own table entries from it)
  int slot_width = 1;  // W: contribution slots per Γ node (max multiplicity over this rank's subdomains)
  void build(mi_ctx_s *c, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d, const int64_t *const *gather_idx,
             int base, int64_t d0, int64_t d1) {
    if (ndom <= 0 || n_gamma < 0 || n_gamma >= INT32_MAX || !n_gamma_d || !gather_idx || d0 < 0 || d1 > ndom || d0 > d1)
      raise(MI_ERR_BAD_ARG, "schur/nn create: bad ndom/n_gamma/domain slice");
    ndl = (int)(d1 - d0);
    // An operator that holds EVERY subdomain is replicated, not sharded: it never communicates, even on a context
    // with a communicator. MI355_FORCE_REDUCE=1 keeps the collective anyway (single-GPU rehearsal of the sharded path).
    sharded = c->has_comm() && (!(d0 == 0 && d1 == ndom) || env_int("MI355_FORCE_REDUCE", 0));
    int64_t tot = 0;
    for (int64_t d = d0; d < d1; ++d) {
      if (n_gamma_d[d] < 0 || n_gamma_d[d] > n_gamma) raise(MI_ERR_BAD_ARG, "n_gamma_d[%lld] out of range", (long long)d);
      if (n_gamma_d[d] && !gather_idx[d]) raise(MI_ERR_BAD_ARG, "gather_idx[%lld] is NULL", (long long)d);
      loc_off.push_back((int)tot);
      nd.push_back((int)n_gamma_d[d]);
      tot += n_gamma_d[d];
      if (tot >= INT32_MAX) raise(MI_ERR_BAD_ARG, "local interface too large");
    }
    nloc = (int)tot;
    gidx_h.resize(nloc);
    std::vector<int> cntv(n_gamma + 1, 0);
    for (int dl = 0; dl < ndl; ++dl) {
      std::vector<char> seen;  // a Dict has unique keys: each Γ node at most once per subdomain
      seen.assign((size_t)n_gamma, 0);
      for (int l = 0; l < nd[dl]; ++l) {
        const int g = to_i32(gather_idx[d0 + dl][l] - base, 0, n_gamma, "gather_idx");
        if (seen[g]) raise(MI_ERR_BAD_ARG, "gather_idx[%d] repeats Γ index %d", dl, g);
        seen[g] = 1;
        gidx_h[loc_off[dl] + l] = g;
        cntv[g + 1]++;
      }
    }
    // inverted index: contributions of every Γ node in ascending subdomain order
    std::partial_sum(cntv.begin(), cntv.end(), cntv.begin());
    std::vector<int> pos(nloc), next(cntv.begin(), cntv.end() - 1);
    for (int dl = 0; dl < ndl; ++dl)
      for (int l = 0; l < nd[dl]; ++l) pos[next[gidx_h[loc_off[dl] + l]]++] = loc_off[dl] + l;
    gidx.upload(gidx_h, c->stream); aptr.upload(cntv, c->stream); apos.upload(pos, c->stream);
    // slot form of the same index for the dense operators: local row -> g*W + j, j = rank of the
    // subdomain among the contributors of Γ node g
    for (int64_t i = 0; i < n_gamma; ++i) slot_width = std::max(slot_width, cntv[i + 1] - cntv[i]);
    std::vector<int> op(nloc);
    // A slice of the subdomains (multi-GPU): slot j of node g must be the rank of the subdomain among ALL contributors
    // of g, on every rank alike — then the ranks' slot tables are disjoint (their sum is a union, x + 0) and have one
    // width. The gather lists of the other ranks' subdomains are index arrays every rank has (set_subdomains is global);
    // if a caller leaves them NULL the local ranks are used: still a correct sum, but colliding slots are added by the
    // all-reduce (no longer the single-GPU order) and all ranks must then happen to agree on the width.
    bool global_slots = ndl < ndom;
    for (int64_t d = 0; d < ndom && global_slots; ++d) global_slots = !(n_gamma_d[d] > 0 && !gather_idx[d]);
    if (global_slots) {
      std::vector<int> seen_cnt((size_t)n_gamma, 0);
      for (int64_t d = 0; d < ndom; ++d)
        for (int64_t l = 0; l < n_gamma_d[d]; ++l) {
          const int g = to_i32(gather_idx[d][l] - base, 0, n_gamma, "gather_idx");
          const int j = seen_cnt[g]++;
          if (d >= d0 && d < d1) op[loc_off[d - d0] + l] = j;  // slot rank for now, position below
        }
      slot_width = 1;
      for (int64_t i = 0; i < n_gamma; ++i) slot_width = std::max(slot_width, seen_cnt[i]);
      if (slot_width == 3) slot_width = 4;
      for (int s = 0; s < nloc; ++s) op[s] = gidx_h[s] * slot_width + op[s];
    } else {
      if (slot_width == 3) slot_width = 4;  // 32-byte aligned slot rows -> one double4 load
      for (int64_t i = 0; i < n_gamma; ++i)
        for (int k = cntv[i]; k < cntv[i + 1]; ++k) op[pos[k]] = (int)(i * slot_width + (k - cntv[i]));
    }
    out_pos.upload(op, c->stream);
    // folded PCG: for local position `loc` of Γ node g with contributors loc_0 < loc_1 < ... (ascending subdomain):
    //   jrank[loc] = my rank among them; peer[loc*W+k] = loc_k; tgt[loc*W+k] = loc_k*W + jrank[loc]
    const int W = slot_width;
    std::vector<int> jr(nloc), pe((size_t)nloc * W + 4, -1), tg((size_t)nloc * W + 4, -1);
    for (int64_t i = 0; i < n_gamma; ++i) {
      const int m = cntv[i + 1] - cntv[i];
      for (int a = 0; a < m; ++a) {
        const int la = pos[cntv[i] + a];
        jr[la] = a;
        for (int k = 0; k < m; ++k) {
          const int lk = pos[cntv[i] + k];
          pe[(size_t)la * W + k] = lk;
          tg[(size_t)la * W + k] = lk * W + a;
        }
      }
    }
    jrank.upload(jr, c->stream); peer.upload(pe, c->stream); tgt.upload(tg, c->stream);
    tgt_h = tg;
    // sums over ranks that this operator will issue inside captured graphs need their staging before the capture
    if (sharded && c->use_peer()) c->peer->reserve_stage((size_t)n_gamma * (size_t)std::max(slot_width, 1) + 4);
  }
  void assemble(mi_ctx_s *c, int64_t n_gamma, const double *yloc, double *y, const int *done) const {
    hipLaunchKernelGGL(k_assemble, dim3(vec_grid(n_gamma)), dim3(NT), 0, c->stream, (int)n_gamma, aptr.p, apos.p, yloc,
                       y, done);
    MI_HIP(hipGetLastError());
    if (sharded) c->allreduce(y, (size_t)n_gamma);
  }
};

// ------------------------------------------------------------------ assembled Schur operator / Neumann-Neumann preconditioner
struct DenseBlockOp : Operator {
  LocalMaps maps;
  bool scale;  // true: Neumann-Neumann (gather r/cnt, result /cnt)
  int rpw, waves, ntiles = 0, max_nd = 0, max_ld = 0;  // rows per wave, waves per workgroup (4, 8 or 16)
  bool reduce_over_ranks = false;          // this rank holds only a slice of the subdomains and a communicator exists
  bool full_maps = false;                  // ... and is built on the maps of all subdomains (inactive tiles for the others)
  DevBuf<double> M, cnt, yslots;
  DevBuf<double> yslots_all;  // multi-GPU: all-reduced copy of the contribution slots (every rank's subdomains)
  // folded PCG launches: [nloc*W] local-order contributions followed by the per-tile partials of the first dot, in ONE
  // buffer (`fold_pack`; `fold_pack_all` = its sum over ranks when the launch is sharded); second partial array; [4*nloc]
  // r/p current+next copies
  DevBuf<double> fold_pack, fold_pack_all, fold_part1, fold_vec;
  size_t fold_con_n = 0, fold_pack_n = 0;
  int part_total = 0;             // slots of one partial-dot array (see the tile records)
  size_t fold_p1_off = 0;         // sharded Neumann-Neumann blocks: the second partial array (r'z) sits in the pack too
  // Peer exchange (exchange.hpp): the reduced pack is a double-buffered table at offset xt_off of every rank's arena;
  // own_idx lists the pack entries this rank's launches produce (contribution slots and per-row partials of its rows).
  bool xt_built = false;           // this operator has a table in the arenas
  bool xt_on() const { return xt_built && ctx->use_peer(); }   // ... and the exchange is switched on (mi_ctx_set_exchange)
  size_t xt_off = 0, xt_copy = 0;
  DevBuf<int> own_idx;
  int n_own = 0;
  double *fold_reduced() const { return xt_on() ? ctx->peer->local(xt_off) : fold_pack_all.p; }   // (peer: copy 0; the launches add the parity)
  double *fold_con(bool reduced = false) const { return (reduced ? fold_reduced() : fold_pack.p); }
  double *fold_part0(bool reduced = false) const { return (reduced ? fold_reduced() : fold_pack.p) + fold_con_n; }
  double *fold_part1p(bool reduced = false) const { return fold_p1_off ? (reduced ? fold_reduced() : fold_pack.p) + fold_p1_off : fold_part1.p; }
  // peer exchange, default: the folded launch stores its results into every arena itself and signals (kernels.hpp); what
  // follows the launch is the one-wave wait. MI355_XCHG_PUSH_KERNEL=1: results into the local pack, pushed by k_xchg_push.
  bool xt_direct = false;
  int n_active = 0;               // streamed tiles (the ones that count themselves in)
  void reduce_fold(const int *done, bool inwait) {
    if (xt_on() && xt_direct && inwait) return;    // the consuming launch waits itself
    if (xt_on() && xt_direct) ctx->peer->wait_advance(ctx->stream, done);
    else if (xt_on()) ctx->peer->push(xt_off, xt_copy, fold_pack.p, own_idx.p, n_own, ctx->stream, done);
    else ctx->allreduce(fold_pack.p, fold_pack_all.p, fold_pack_n);
  }
  void fold_outputs(PcgFold &f) const {   // con_out / part_out0 / part_out1 (+ the peer-store fields) of this operator's launch
    if (xt_on() && xt_direct) {
      double *t0 = ctx->peer->local(xt_off);
      f.con_out = t0; f.part_out0 = t0 + fold_con_n; f.part_out1 = fold_p1_off ? t0 + fold_p1_off : nullptr;
      f.xp = ctx->peer->peers_dev; f.xst = ctx->peer->st; f.out_stride = (long long)xt_copy; f.n_arrive = (unsigned int)n_active;
    } else {
      f.con_out = fold_con(); f.part_out0 = fold_part0(); f.part_out1 = scale ? fold_part1p() : nullptr;
    }
  }
  DevBuf<GemvTile> tiles;
  std::vector<long long> moff_h;  // per local subdomain: element offset of its block in M (row-major, ld_h[dl])
  std::vector<int> ld_h;
  std::vector<char> owned_h;      // per local subdomain: its block is stored on this rank
  int64_t alg_bytes = 0;
  DenseMeta meta{};

  DenseBlockOp(mi_ctx_s *c, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d, const int64_t *const *gather_idx,
               const double *const *blocks, const int64_t *node_cnt, int base, int64_t d0, int64_t d1)
      : Operator(c, n_gamma), scale(node_cnt != nullptr) {
    if (!blocks) raise(MI_ERR_BAD_ARG, "dense blocks pointer is NULL");
    // A slice of the subdomains (multi-GPU) is built on the maps of ALL subdomains whenever their gather lists are given
    // (index arrays every rank has): tiles of the other ranks' blocks exist but are inactive. The slot tables of the
    // ranks are then disjoint by construction, and the folded PCG launches can run sharded (solvers.hpp).
    full_maps = true;
    for (int64_t d = 0; d < ndom && full_maps; ++d) full_maps = !(n_gamma_d && n_gamma_d[d] > 0 && gather_idx && !gather_idx[d]);
    const int64_t m0 = full_maps ? 0 : d0, m1 = full_maps ? ndom : d1;
    maps.build(c, ndom, n_gamma, n_gamma_d, gather_idx, base, m0, m1);
    // Replicated operators (every subdomain present) never communicate — e.g. the Neumann-Neumann blocks copied to
    // all ranks while S is sharded, which halves the all-reduces of a multi-GPU PCG iteration.
    reduce_over_ranks = c->has_comm() && (!(d0 == 0 && d1 == ndom) || env_int("MI355_FORCE_REDUCE", 0));
    maps.sharded = reduce_over_ranks;
    rpw = env_int("MI355_GEMV_RPW", 2);
    if (rpw != 1 && rpw != 2 && rpw != 4) rpw = 2;
    waves = env_int("MI355_GEMV_WAVES", 16);
    if (waves != 4 && waves != 8 && waves != 16) waves = 16;
    if (reduce_over_ranks && full_maps && !env_int("MI355_GEMV_WAVES", 0) && !env_int("MI355_GEMV_RPW", 0)) {
      // A rank that owns 1/N of the blocks would keep 1/N of the CUs busy with 32-row tiles (a launch lasts as long as
      // one tile): cut the owned rows into about one tile per CU instead.
      int64_t owned_rows = 0;
      for (int64_t d = d0; d < d1; ++d) owned_rows += n_gamma_d[d];
      hipDeviceProp_t prop;
      MI_HIP(hipGetDeviceProperties(&prop, c->device));
      const int64_t per_tile = std::max<int64_t>(1, owned_rows / std::max(1, prop.multiProcessorCount));
      if (per_tile >= 24) { waves = 16; rpw = 2; }
      else if (per_tile >= 12) { waves = 16; rpw = 1; }
      else if (per_tile >= 6) { waves = 8; rpw = 1; }
      else { waves = 4; rpw = 1; }
    }
    std::vector<long long> moff;
    std::vector<int> ldv;
    std::vector<GemvTile> tv;
    const bool sharded_parts = reduce_over_ranks && full_maps;
    part_total = 0;
    long long tot = 0;
    auto owned = [&](int dl) { return m0 + dl >= d0 && m0 + dl < d1; };
    for (int dl = 0; dl < maps.ndl; ++dl) {
      const int n_d = maps.nd[dl];
      int l = (n_d + 15) / 16 * 16;
      // A row stride that is a multiple of 2 KiB puts every row of a tile on the same HBM channels: measured 27 % slower
      // at n_Γd = 1024 (profiles/r01_gemv_variant_sweep.txt). One extra 128-byte line per row breaks the pattern.
      if (l % 256 == 0 && l != GEMV_PANEL) l += 16;
      const bool own = owned(dl);
      if (own && n_d && !blocks[m0 + dl]) raise(MI_ERR_BAD_ARG, "dense block %d is NULL", dl);
      moff.push_back(own ? tot : 0); ldv.push_back(l);
      max_nd = std::max(max_nd, n_d);
      max_ld = std::max(max_ld, l);
      // tiles of another rank's block only do owner duties in the folded launches: as few workgroups as possible
      const int step = own ? waves * rpw : 64 * waves;
      // `active` of a streamed tile = 1 + the slot of its partial dot products. One GPU: the tile number. Sharded over ranks:
      // a layout every rank derives from the maps alone, whatever tiling each rank chose for its own blocks — subdomain
      // after subdomain, one slot per PART_ROWS rows (no tiling has fewer rows per tile), so the ranks' arrays are a
      // disjoint union of one array and the exchange adds nothing.
      for (int r = 0; r < n_d; r += step) {
        const int slot = sharded_parts ? part_total + r / PART_ROWS : (int)tv.size();
        tv.push_back(GemvTile{own ? tot : 0, n_d, l, maps.loc_off[dl], r, own ? slot + 1 : 0, std::min(step, n_d - r)});
      }
      if (sharded_parts) part_total += (n_d + PART_ROWS - 1) / PART_ROWS;
      if (own) {
        tot += (long long)n_d * l;
        alg_bytes += 8ll * n_d * n_d + 16ll * n_d + 4ll * n_d;
      }
    }
    ntiles = (int)tv.size();
    if (!sharded_parts) part_total = ntiles;
    moff_h = moff; ld_h = ldv;
    for (int dl = 0; dl < maps.ndl; ++dl) owned_h.push_back(owned(dl) ? 1 : 0);
    // (+ one zeroed panel behind the last block: the persistent kernel reads whole 128-double groups of a row without
    // clamping, so a read may run past a row's end — into the next row, or into this tail — and meets a zero operand there)
    M.alloc((size_t)tot + GEMV_PANEL);
    memset_sync(M.p + tot, 0, sizeof(double) * GEMV_PANEL);
    // column-major (Julia) -> padded row-major, one block at a time
    for (int dl = 0; dl < maps.ndl; ++dl) {
      if (!owned(dl)) continue;
      const int n_d = maps.nd[dl], l = ldv[dl];
      std::vector<double> rowm((size_t)n_d * l, 0.0);
      const double *src = blocks[m0 + dl];
      for (int j = 0; j < n_d; ++j)
        for (int i = 0; i < n_d; ++i) rowm[(size_t)i * l + j] = src[(size_t)i + (size_t)j * n_d];
      if (!rowm.empty())
        memcpy_sync(M.p + moff[dl], rowm.data(), rowm.size() * sizeof(double), hipMemcpyHostToDevice);
    }
    if (scale) {
      std::vector<double> cv(maps.nloc);
      for (int s = 0; s < maps.nloc; ++s) {
        const int64_t v = node_cnt[maps.gidx_h[s]];
        if (v <= 0) raise(MI_ERR_BAD_ARG, "node_gamma_cnt must be positive");
        cv[s] = (double)v;
      }
      cnt.upload(cv, c->stream);
    }
    tiles.upload(tv, c->stream);
    yslots.alloc((size_t)n_gamma * maps.slot_width + 4);
    yslots.zero(c->stream);  // unused slots (and, multi-GPU, the other ranks' slots) stay 0 for the lifetime of the operator
    yslots_all.alloc((size_t)n_gamma * maps.slot_width + 4);
    yslots_all.zero(c->stream);
    // contributions and first partial-dot array share one buffer: the sharded S launch all-reduces both in one call
    fold_con_n = (size_t)maps.nloc * maps.slot_width + 4;
    const size_t part_n = (size_t)part_total + 1;
    fold_pack_n = fold_con_n + part_n;
    if (sharded_parts && scale) { fold_p1_off = fold_pack_n; fold_pack_n += part_n; }  // partial r'z, exchanged with the rest
    fold_pack.alloc(fold_pack_n); fold_pack.zero(c->stream);
    fold_pack_all.alloc(fold_pack_n); fold_pack_all.zero(c->stream);
    fold_part1.alloc((size_t)ntiles + 1); fold_part1.zero(c->stream);
    if (reduce_over_ranks && c->use_peer()) {
      // staging of the generic sums this operator issues (plain applies: the slot table), reserved outside any capture
      c->peer->reserve_stage(std::max<size_t>(fold_pack_n, (size_t)n_gamma * maps.slot_width + 4));
      if (full_maps) {
        // the folded launches' table: entries this rank produces = the contribution slots its rows write (tgt) and the
        // per-row partials of its rows
        std::vector<int> own;
        const int W = maps.slot_width;
        for (int dl = 0; dl < maps.ndl; ++dl) {
          if (!owned(dl)) continue;
          for (int l = 0; l < maps.nd[dl]; ++l) {
            const int loc = maps.loc_off[dl] + l;
            for (int k = 0; k < W; ++k) { const int tg = maps.tgt_h[(size_t)loc * W + k]; if (tg >= 0) own.push_back(tg); }
          }
        }
        for (const GemvTile &t : tv)
          if (t.active) {
            own.push_back((int)fold_con_n + t.active - 1);
            if (fold_p1_off) own.push_back((int)fold_p1_off + t.active - 1);
          }
        std::sort(own.begin(), own.end());   // neighbouring threads store neighbouring slots
        n_own = (int)own.size();
        own_idx.upload(own, c->stream);
        xt_copy = (fold_pack_n + 31) & ~(size_t)31;
        xt_off = c->peer->alloc(2 * xt_copy * sizeof(double));   // (zero since the arena was created: a bump allocator never re-uses)
        xt_built = true;
        for (const GemvTile &t : tv) n_active += t.active != 0;
        xt_direct = n_active > 0 && !env_int("MI355_XCHG_PUSH_KERNEL", 0);   // (a rank without a block has nothing to count in: push kernel)
      }
    }
    fold_vec.alloc((size_t)maps.nloc * 4 + 4); fold_vec.zero(c->stream);
    MI_HIP(hipStreamSynchronize(c->stream));
    meta = DenseMeta{M.p, tiles.p, maps.gidx.p, scale ? cnt.p : nullptr, maps.out_pos.p};
  }
  void gemv(const double *x, const int *done) {
    if (!ntiles) return;
    const int *zx = zero_hint;
#define MI_GEMV(R, S, V) hipLaunchKernelGGL((k_gemv_batched<R, S, V>), dim3(ntiles), dim3(64 * V), 0, ctx->stream, meta, x, yslots.p, done, zx)
#define MI_GEMV_R(S, V) do { if (rpw == 1) MI_GEMV(1, S, V); else if (rpw == 2) MI_GEMV(2, S, V); else MI_GEMV(4, S, V); } while (0)
    if (waves == 16)     { if (scale) MI_GEMV_R(true, 16); else MI_GEMV_R(false, 16); }
    else if (waves == 8) { if (scale) MI_GEMV_R(true, 8); else MI_GEMV_R(false, 8); }
    else                 { if (scale) MI_GEMV_R(true, 4); else MI_GEMV_R(false, 4); }
#undef MI_GEMV_R
#undef MI_GEMV
    MI_HIP(hipGetLastError());
  }
  void apply(const double *x, double *y, const int *done) override {
    gemv(x, done);
    // Multi-GPU: the slot tables of the ranks are disjoint, so their sum is the full table and the Γ-sum below runs in
    // the single-GPU order (reducing y instead would add per-rank partial sums: same value, different rounding).
    if (reduce_over_ranks) ctx->allreduce(yslots.p, yslots_all.p, (size_t)n * maps.slot_width);
    hipLaunchKernelGGL(k_assemble_slots, dim3(vec_grid(n)), dim3(NT), 0, ctx->stream, (int)n, maps.slot_width,
                       reduce_over_ranks ? yslots_all.p : yslots.p, y, done);
    MI_HIP(hipGetLastError());
  }
  DenseBlockOp *as_dense() override { return this; }
  // New blocks on the same maps: `src` holds the owned blocks back to back, column-major (device pointer).
  void set_blocks(const double *src);
  static constexpr int KV = 4;  // columns per pass of apply_multi (4 x 16 KiB of LDS for the operand panels)
  DevBuf<double> yslots_multi;
  void apply_multi(const double *X, int64_t ldx, int k, double *Y, int64_t ldy) override {
    if (reduce_over_ranks || !ntiles || env_int("MI355_NO_MULTI", 0)) { Operator::apply_multi(X, ldx, k, Y, ldy); return; }
    const long long stride = (long long)n * maps.slot_width + 4;
    if (yslots_multi.n < (size_t)(stride * KV)) { yslots_multi.alloc((size_t)(stride * KV)); yslots_multi.zero(ctx->stream); }
    for (int v0 = 0; v0 < k; v0 += KV) {
      const int kv = std::min(KV, k - v0);
      const double *Xv = X + (size_t)v0 * ldx;
#define MI_GM(S) hipLaunchKernelGGL((k_gemv_multi<2, KV, S, 16>), dim3(ntiles), dim3(1024), 0, ctx->stream, meta, Xv, (long long)ldx, kv, yslots_multi.p, stride)
      if (scale) MI_GM(true); else MI_GM(false);
#undef MI_GM
      hipLaunchKernelGGL(k_assemble_slots_multi, dim3(vec_grid(n), kv), dim3(NT), 0, ctx->stream, (int)n, maps.slot_width,
                         yslots_multi.p, stride, Y + (size_t)v0 * ldy, (long long)ldy);
    }
    MI_HIP(hipGetLastError());
  }
  // One launch of the folded PCG pair (kernels.hpp k_gemv_pcg); PHASE 1 on the ΠS operator, 0 on S.
  void gemv_pcg(int phase, const PcgFold &f) {
    if (!ntiles) return;
    const bool xchg = f.xp != nullptr || f.x_inwait != 0;   // this launch stores into the peers' arenas and / or waits for them
#define MI_PCG4(R, P, C, V) do { if (xchg) hipLaunchKernelGGL((k_gemv_pcg<R, P, C, V, true>), dim3(ntiles), dim3(64 * V), 0, ctx->stream, meta, f); \
                                 else hipLaunchKernelGGL((k_gemv_pcg<R, P, C, V, false>), dim3(ntiles), dim3(64 * V), 0, ctx->stream, meta, f); } while (0)
#define MI_PCG3(R, C, V) do { if (phase) MI_PCG4(R, 1, C, V); else MI_PCG4(R, 0, C, V); } while (0)
#define MI_PCG2(R, V) do { const int c = (max_ld + 64 * V - 1) / (64 * V); \
                           if (c <= 2) MI_PCG3(R, 2, V); else if (c == 3) MI_PCG3(R, 3, V); else if (c == 4) MI_PCG3(R, 4, V); \
                           else if (c == 5) MI_PCG3(R, 5, V); else if (c == 6) MI_PCG3(R, 6, V); else MI_PCG3(R, 8, V); } while (0)
#define MI_PCG(R) do { if (waves == 16) MI_PCG2(R, 16); else if (waves == 8) MI_PCG2(R, 8); else MI_PCG2(R, 4); } while (0)
    if (rpw == 1) MI_PCG(1); else if (rpw == 2) MI_PCG(2); else MI_PCG(4);
#undef MI_PCG
#undef MI_PCG2
#undef MI_PCG3
#undef MI_PCG4
    MI_HIP(hipGetLastError());
  }
  bool same_maps(const DenseBlockOp &o) const {
    // (the tilings may differ: a sharded S cuts its few blocks into more, smaller tiles)
    return n == o.n && maps.slot_width == o.maps.slot_width && maps.nd == o.maps.nd && maps.gidx_h == o.maps.gidx_h;
  }
  AsmView view_of(double *) override {
    return reduce_over_ranks ? AsmView{yslots_all.p, maps.slot_width} : AsmView{yslots.p, maps.slot_width};
  }
  AsmView apply_view(const double *x, double *y, const int *done) override {
    gemv(x, done);
    if (!reduce_over_ranks) return AsmView{yslots.p, maps.slot_width};
    // Multi-GPU: every rank wrote only its own subdomains' slots (the rest are zero), so the out-of-place sum
    // over ranks IS the full slot table (x + 0 is exact): the consumer then takes the Γ-sum in the same
    // ascending-subdomain order as on one GPU, and no separate assemble launch is needed.
    ctx->allreduce(yslots.p, yslots_all.p, (size_t)n * maps.slot_width);
    return AsmView{yslots_all.p, maps.slot_width};
  }
  void bytes(int64_t *a, int64_t *d) const override { *a = alg_bytes + 8 * n; *d = alg_bytes; }
  void apply_dominant(const double *x) override { gemv(x, nullptr); }
};

// M[moff + i*ld + j] = src[i + j*n] (column-major block -> padded row-major block)
__global__ __launch_bounds__(NT) void k_block_to_rowmajor(int n, int ld, const double *__restrict__ src, double *__restrict__ dstm) {
  const long long tot = (long long)n * n;
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < tot; e += (long long)gridDim.x * NT) {
    const int j = (int)(e % n), i = (int)(e / n);   // consecutive threads: consecutive j of one row -> coalesced stores
    dstm[(long long)i * ld + j] = src[i + (long long)j * n];
  }
}
inline void DenseBlockOp::set_blocks(const double *src) {
  size_t off = 0;
  for (int dl = 0; dl < maps.ndl; ++dl) {
    const int n_d = maps.nd[dl];
    if (!owned_h[dl] || n_d == 0) continue;
    hipLaunchKernelGGL(k_block_to_rowmajor, dim3((int)std::max<long long>(1, std::min<long long>(((long long)n_d * n_d + NT - 1) / NT, 4096))),
                       dim3(NT), 0, ctx->stream, n_d, ld_h[dl], src + off, M.p + moff_h[dl]);
    off += (size_t)n_d * n_d;
  }
  MI_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ host staging for the interior-solve callback
struct HostStage {
  double *rhs = nullptr, *sol = nullptr;
  size_t n = 0;
  void ensure(size_t m) {
    if (m <= n) return;
    release();
    MI_HIP(hipHostMalloc((void **)&rhs, (m ? m : 1) * sizeof(double)));
    MI_HIP(hipHostMalloc((void **)&sol, (m ? m : 1) * sizeof(double)));
    n = m;
  }
  void release() {
    if (rhs) (void)hipHostFree(rhs);
    if (sol) (void)hipHostFree(sol);
    rhs = sol = nullptr; n = 0;
  }
  ~HostStage() { release(); }
};

// ------------------------------------------------------------------ interior CG on the device (kernels.hpp, IcgMeta)
struct InteriorCg {
  mi_ctx_s *ctx = nullptr;
  CsrDev A;  // block-diagonal A_II of the local subdomains
  int ndl = 0, n = 0, chunk = 64;
  double reltol = 1e-9;
  DevBuf<double> u, c, r, p_uc, p_rr, res_cur, res_nxt, tol;
  DevBuf<int> blk_dom, dom_b0, dom_b1, done_cur, done_nxt, iters, n_i;
  int *done_host = nullptr;
  hipGraphExec_t graph = nullptr;
  double *graph_x = nullptr;
  IcgMeta meta{};
  long long total_iterations = 0;  // statistics: iterations of the slowest subdomain, summed over solves
  // 2-launch form (k_icg_spmv / k_icg_update_blk), MI355_ICG_FUSED=1: measured SLOWER than the 3-launch loop at 1 M DoF
  // (51.6 vs 37.6 us per iteration: the on-the-fly direction doubles the gathered bytes of a 7 M-non-zero SpMV), so
  // it is an opt-in kept for its tests; DESIGN.md §5
  bool folded = false;
  IcgFold fm{};
  DevBuf<IcgPiece> pieces;
  DevBuf<IcgDomState> dst;   // cur[ndl], nxt[ndl]
  DevBuf<double> ur, part_rz, dinv, rho;
  DevBuf<IcgBlkInfo> binfo;
  DevBuf<int> dom_p0, dom_p1;
  IcgDomState *dst_host = nullptr;
  std::vector<int> ioff_h;
  int npieces_grid = 0;
  bool jacobi = false;       // `Pl` = Diagonal(A_II) (mi_schur_matfree_interior_precond)

  void build(mi_ctx_s *c_, const HostCsr &a, const std::vector<int> &ioff, const std::vector<int> &ni, double reltol_) {
    ctx = c_; reltol = reltol_; ndl = (int)ni.size(); n = a.n_rows;
    std::vector<int> breaks(ioff.begin(), ioff.end());
    A.upload(a, ctx->stream, &breaks);
    std::vector<int> bd(A.nblocks), b0(ndl, 0), b1(ndl, 0);
    int d = 0;
    for (int b = 0; b < A.nblocks; ++b) {
      while (d + 1 < ndl && A.blocks_h[b].r0 >= ioff[d + 1]) ++d;
      bd[b] = d;
    }
    for (int dd = 0; dd < ndl; ++dd) { b0[dd] = A.nblocks; b1[dd] = 0; }
    for (int b = 0; b < A.nblocks; ++b) { b0[bd[b]] = std::min(b0[bd[b]], b); b1[bd[b]] = std::max(b1[bd[b]], b + 1); }
    for (int dd = 0; dd < ndl; ++dd) if (b0[dd] > b1[dd]) b0[dd] = b1[dd] = 0;  // empty interior
    blk_dom.upload(bd, ctx->stream); dom_b0.upload(b0, ctx->stream); dom_b1.upload(b1, ctx->stream);
    n_i.upload(ni, ctx->stream);
    u.alloc(n + 1); c.alloc(n + 1); r.alloc(n + 1);
    p_uc.alloc(A.nblocks + 1); p_rr.alloc(A.nblocks + 1);
    res_cur.alloc(ndl + 1); res_nxt.alloc(ndl + 1); tol.alloc(ndl + 1);
    done_cur.alloc(ndl + 1); done_nxt.alloc(ndl + 1); iters.alloc(ndl + 1);
    MI_HIP(hipHostMalloc((void **)&done_host, sizeof(int) * (ndl + 1)));
    chunk = std::max(1, env_int("MI355_ICG_CHUNK", 64));
    rho.alloc(2 * (size_t)ndl + 2); part_rz.alloc((size_t)A.nblocks + 1);
    meta = IcgMeta{A.blk.p, blk_dom.p, dom_b0.p, dom_b1.p, res_cur.p, res_nxt.p, tol.p, done_cur.p, done_nxt.p, iters.p, 0,
                   nullptr, rho.p, rho.p + ndl + 1, part_rz.p, nullptr, nullptr};
    {  // 1 / diagonal of A_II (for the optional Jacobi `Pl`)
      std::vector<double> dg((size_t)n + 1, 1.0);
      for (int r_ = 0; r_ < a.n_rows; ++r_)
        for (int k = a.rowptr[r_]; k < a.rowptr[r_ + 1]; ++k)
          if (a.col[k] == r_) dg[r_] = 1.0 / a.val[k];
      dinv.upload(dg, ctx->stream);
    }
    folded = env_int("MI355_ICG_FUSED", 0) != 0;
    ioff_h = ioff;
    build_fold(a, b0, b1);   // the pieces serve both forms
    meta.p_rz = part_rz.p; meta.dom_p0 = dom_p0.p; meta.dom_p1 = dom_p1.p;
  }
  // Pieces of the vector kernel: rows cut at subdomain boundaries and at the XCD boundaries of the SpMV's block dealing,
  // then into runs of <= `rows_per` rows; workgroup k works on XCD k & 7, so the pieces are interleaved per XCD.
  void build_fold(const HostCsr &a, const std::vector<int> &b0, const std::vector<int> &b1) {
    hipStream_t s = ctx->stream;
    const int per = (A.nblocks + 7) >> 3;
    std::vector<int> xr(9, n);
    for (int x = 0; x < 8; ++x) xr[x] = x * per < A.nblocks ? A.blocks_h[(size_t)x * per].r0 : n;
    const int target = std::max(256, env_int("MI355_ICG_PIECES", 512));   // about this many pieces in all
    const int rows_per = std::max(NT, (n + target - 1) / target);
    std::vector<std::vector<IcgPiece>> per_xcd(8);
    std::vector<int> p0(ndl, 0), p1(ndl, 0);
    int slot = 0;
    bool ok = true;
    for (int d = 0; d < ndl; ++d) {
      p0[d] = slot;
      const int lo_d = ioff_h[d], hi_d = d + 1 < (int)ioff_h.size() ? ioff_h[d + 1] : n;
      for (int x = 0; x < 8; ++x) {
        const int lo = std::max(lo_d, xr[x]), hi = std::min(hi_d, xr[x + 1]);
        if (hi <= lo) continue;
        const int cnt = (hi - lo + rows_per - 1) / rows_per;
        for (int k = 0; k < cnt; ++k) {
          const int a0 = lo + (int)((long long)(hi - lo) * k / cnt), a1 = lo + (int)((long long)(hi - lo) * (k + 1) / cnt);
          per_xcd[x].push_back(IcgPiece{a0, a1, d, slot, b0[d], b1[d], slot == p0[d] ? 1 : 0, 0, p0[d], 0});
          ++slot;
        }
      }
      p1[d] = slot;
      if (p1[d] - p0[d] > NT) ok = false;   // k_icg_spmv sums a subdomain's partials with one load per thread
    }
    if (!ok) folded = false;   // (the 2-launch form sums a subdomain's partials with one load per thread)
    for (auto &v : per_xcd)
      for (auto &pc : v) pc.p1 = p1[pc.dom];
    size_t mx = 0;
    for (auto &v : per_xcd) mx = std::max(mx, v.size());
    std::vector<IcgPiece> tab(8 * std::max<size_t>(1, mx), IcgPiece{0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
    for (int x = 0; x < 8; ++x)
      for (size_t j = 0; j < per_xcd[x].size(); ++j) tab[j * 8 + x] = per_xcd[x][j];
    npieces_grid = (int)tab.size();
    pieces.upload(tab, s);
    dom_p0.upload(p0, s); dom_p1.upload(p1, s);
    {
      std::vector<int> bdh(A.nblocks), nih(ndl);
      MI_HIP(hipStreamSynchronize(s));
      memcpy_sync(bdh.data(), blk_dom.p, sizeof(int) * A.nblocks, hipMemcpyDeviceToHost);
      memcpy_sync(nih.data(), n_i.p, sizeof(int) * ndl, hipMemcpyDeviceToHost);
      std::vector<IcgBlkInfo> bi(A.nblocks);
      for (int b = 0; b < A.nblocks; ++b) {
        const int d = bdh[b];
        bi[b] = IcgBlkInfo{d, p0[d], p1[d] - p0[d], b == b0[d] ? 1 : 0, nih[d], 0, 0, 0};
      }
      binfo.upload(bi, s);
    }
    std::vector<IcgDomState> st0(2 * (size_t)ndl);
    for (auto &q : st0) { q.rho_prev = 1.0; q.tol = 0.0; q.res = 0.0; q.it = 0; q.done = 1; }   // an empty interior stays "done"
    dst.upload(st0, s);
    if (folded) ur.alloc(4 * (size_t)n + 4);
    if (part_rz.n < (size_t)slot + 1) part_rz.alloc((size_t)slot + 1);
    if (p_rr.n < (size_t)slot + 1) p_rr.alloc((size_t)slot + 1);
    MI_HIP(hipHostMalloc((void **)&dst_host, sizeof(IcgDomState) * (ndl + 1)));
    fm = IcgFold{A.blk.p, binfo.p, dst.p, dst.p + ndl, ur.p, ur.p + 2 * (size_t)n,
                 c.p, r.p, p_uc.p, p_rr.p, part_rz.p, nullptr, reltol};
  }
  ~InteriorCg() {
    if (graph) (void)hipGraphExecDestroy(graph);
    if (done_host) (void)hipHostFree(done_host);
    if (dst_host) (void)hipHostFree(dst_host);
  }
  void set_jacobi(bool on) {
    if (on != jacobi && graph) { (void)hipGraphExecDestroy(graph); graph = nullptr; }   // kernel arguments are captured
    jacobi = on;
    fm.dinv = on ? dinv.p : nullptr;
    meta.dinv = on ? dinv.p : nullptr;
  }
  // after new values of A_II: 1 / diagonal again (device side)
  void refresh_diagonal() {
    if (n == 0) return;
    hipLaunchKernelGGL(k_csr_inv_diag, dim3((n + NT - 1) / NT), dim3(NT), 0, ctx->stream, n, A.rowptr.p, A.col.p, A.val.p, dinv.p);
    MI_HIP(hipGetLastError());
  }
  void iteration(double *x) {
    hipStream_t s = ctx->stream;
    if (folded) {
      const int grid = ((A.nblocks + 7) / 8) * 8;
      hipLaunchKernelGGL(k_icg_spmv, dim3(grid), dim3(NT), 0, s, A.nblocks, fm, A.rowptr.p, A.col.p, A.val.p);
      hipLaunchKernelGGL(k_icg_update_blk, dim3(npieces_grid), dim3(NT), 0, s, fm, pieces.p, x);
      MI_HIP(hipGetLastError());
      return;
    }
    A.launch(0, u.p, nullptr, c.p, nullptr, s, u.p, p_uc.p);  // c = A u, partial u'c
    hipLaunchKernelGGL(k_icg_update, dim3(npieces_grid), dim3(NT), 0, s, meta, pieces.p, p_uc.p, u.p, c.p, x, r.p, p_rr.p);
    hipLaunchKernelGGL(k_icg_direction, dim3(npieces_grid), dim3(NT), 0, s, meta, pieces.p, p_rr.p, r.p, u.p, n_i.p);
    MI_HIP(hipGetLastError());
  }
  // x = A_II^{-1} rhs for every local subdomain (device pointers), to the relative tolerance
  void solve(const double *rhs, double *x) {
    if (A.nblocks == 0) return;
    hipStream_t s = ctx->stream;
    if (folded) {
      hipLaunchKernelGGL(k_icg_fold_init, dim3(npieces_grid), dim3(NT), 0, s, fm, pieces.p, rhs, x);
    } else {
      hipLaunchKernelGGL(k_icg_init, dim3(npieces_grid), dim3(NT), 0, s, meta, pieces.p, rhs, x, r.p, u.p, p_rr.p);
      hipLaunchKernelGGL(k_icg_start, dim3(ndl), dim3(NT), 0, s, meta, p_rr.p, reltol);
    }
    MI_HIP(hipGetLastError());
    if (!graph || graph_x != x) {  // `chunk` iterations per replay
      if (graph) { (void)hipGraphExecDestroy(graph); graph = nullptr; }
      hipGraph_t gr = nullptr;
      MI_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      try {
        for (int k = 0; k < chunk; ++k) iteration(x);
      } catch (...) {
        (void)hipStreamEndCapture(s, &gr);
        if (gr) (void)hipGraphDestroy(gr);
        throw;
      }
      MI_HIP(hipStreamEndCapture(s, &gr));
      hipError_t e = hipGraphInstantiate(&graph, gr, nullptr, nullptr, 0);
      (void)hipGraphDestroy(gr);
      if (e != hipSuccess) raise(MI_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
      graph_x = x;
    }
    const long long max_replays = (long long)n / chunk + 2;  // maxiter = size(A, 2)
    for (long long l = 0; l <= max_replays; ++l) {
      bool all = true;
      if (folded) {
        MI_HIP(hipMemcpyAsync(dst_host, dst.p, sizeof(IcgDomState) * ndl, hipMemcpyDeviceToHost, s));   // cur[]
        MI_HIP(hipStreamSynchronize(s));
        for (int d = 0; d < ndl; ++d) all = all && dst_host[d].done;
      } else {
        MI_HIP(hipMemcpyAsync(done_host, done_nxt.p, sizeof(int) * ndl, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        for (int d = 0; d < ndl; ++d) all = all && done_host[d];
      }
      if (all) break;
      MI_HIP(hipGraphLaunch(graph, s));
      total_iterations += chunk;
    }
  }
};

// ------------------------------------------------------------------ matrix-free local Schur operator (EPDD.jl:711-747)
// All local subdomains are stacked into block-diagonal CSR matrices over the concatenated local
// spaces, so one SpMV launch serves every subdomain: rhs = A_IΓ xd ; t = A_ΓΓ xd ; yloc = t - A_ΓI v.
// v = A_II^{-1} rhs is either the host callback (north_star: interior solve on the host) or the device CG
// above (the reference's own inexact `IterativeSolvers.cg(A_IIdd, rhs; reltol)`).
struct MatfreeSchurOp : Operator {
  LocalMaps maps;
  CsrDev A_IG, A_GI, A_GG;
  std::vector<int> ni, ioff;
  int64_t d0;
  int ni_tot = 0;
  mi_interior_solve_fn solve; void *user;
  std::unique_ptr<InteriorCg> icg;
  // exact interior solve by the level inverses a set-up plan keeps (mi_schur_matfree_interior_levels; setup_gj.hpp): when set,
  // it replaces the callback / the interior CG in every interior solve of this operator
  std::function<void(const double *, double *)> level_solver;
  DevBuf<double> xcat, rhs, sol, t1, yloc, t2;
  DevBuf<int> ig_perm;  // A_IG.val[k] = (caller's concatenated A_IΓ values)[ig_perm[k]]  (set_values)
  HostStage stage;

  MatfreeSchurOp(mi_ctx_s *c, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d, const int64_t *n_i,
                 const int64_t *const *gather_idx, const int64_t *const *ig_ptr, const int64_t *const *ig_idx,
                 const double *const *ig_val, const int64_t *const *gg_ptr, const int64_t *const *gg_idx,
                 const double *const *gg_val, mi_interior_solve_fn f, void *u, int base, int64_t d0_, int64_t d1,
                 const int64_t *const *ii_ptr = nullptr, const int64_t *const *ii_idx = nullptr,
                 const double *const *ii_val = nullptr, double reltol = 1e-9)
      : Operator(c, n_gamma), d0(d0_), solve(f), user(u) {
    if ((!f && !ii_ptr) || !n_i || !ig_ptr || !ig_idx || !ig_val || !gg_ptr || !gg_idx || !gg_val)
      raise(MI_ERR_BAD_ARG, "mi_schur_matfree_create: NULL argument");
    if (ii_ptr && (!ii_idx || !ii_val || !(reltol > 0.0))) raise(MI_ERR_BAD_ARG, "device interior: bad A_II arrays / reltol");
    maps.build(c, ndom, n_gamma, n_gamma_d, gather_idx, base, d0, d1);
    HostCsr ig, gi, gg, ii;
    std::vector<int> perm_h;
    int64_t itot = 0;
    for (int dl = 0; dl < maps.ndl; ++dl) { ioff.push_back((int)itot); ni.push_back((int)n_i[d0 + dl]); itot += n_i[d0 + dl]; }
    if (itot >= INT32_MAX) raise(MI_ERR_BAD_ARG, "interior too large");
    ni_tot = (int)itot;
    for (int dl = 0; dl < maps.ndl; ++dl) {
      const int64_t d = d0 + dl;
      // CSC arrays of A_IΓdd (n_i x n_Γd) are the CSR arrays of A_ΓIdd = A_IΓdd' (n_Γd x n_i)
      HostCsr gi_d = host_csr(maps.nd[dl], ni[dl], ig_ptr[d], ig_idx[d], ig_val[d], base);
      HostCsr ig_d = transpose(gi_d);
      {  // where every entry of the transpose came from, in the caller's concatenated value order
        HostCsr tag = gi_d;
        for (size_t k = 0; k < tag.val.size(); ++k) tag.val[k] = (double)(gi.val.size() + k);
        const HostCsr t = transpose(tag);
        for (double v : t.val) perm_h.push_back((int)v);
      }
      HostCsr gg_d = host_csr(maps.nd[dl], maps.nd[dl], gg_ptr[d], gg_idx[d], gg_val[d], base);
      append_block(gi, gi_d, ioff[dl], ni_tot);
      append_block(ig, ig_d, maps.loc_off[dl], maps.nloc);
      append_block(gg, gg_d, maps.loc_off[dl], maps.nloc);
      if (ii_ptr) append_block(ii, host_csr(ni[dl], ni[dl], ii_ptr[d], ii_idx[d], ii_val[d], base), ioff[dl], ni_tot);
    }
    if (gi.rowptr.empty()) { gi.rowptr = {0}; ig.rowptr = {0}; gg.rowptr = {0}; ii.rowptr = {0}; }
    A_IG.upload(ig, c->stream); A_GI.upload(gi, c->stream); A_GG.upload(gg, c->stream);
    ig_perm.upload(perm_h, c->stream);
    xcat.alloc(maps.nloc + 1); t1.alloc(maps.nloc + 1); yloc.alloc(maps.nloc + 1);
    rhs.alloc(ni_tot + 1); sol.alloc(ni_tot + 1); t2.alloc((size_t)n_gamma + 1);
    if (ii_ptr) {
      icg.reset(new InteriorCg);
      icg->build(c, ii, ioff, ni, reltol);
    } else {
      stage.ensure(ni_tot);
    }
  }
  bool graph_safe() const override { return false; }
  void apply(const double *x, double *y, const int *) override {
    hipStream_t s = ctx->stream;
    if (maps.nloc) {
      hipLaunchKernelGGL(k_gather, dim3(vec_grid(maps.nloc)), dim3(NT), 0, s, maps.nloc, maps.gidx.p, x, xcat.p);
      MI_HIP(hipGetLastError());
      A_IG.launch(0, xcat.p, nullptr, rhs.p, nullptr, s);
      A_GG.launch(0, xcat.p, nullptr, t1.p, nullptr, s);
      if (level_solver) {
        level_solver(rhs.p, sol.p);
      } else if (icg) {
        icg->solve(rhs.p, sol.p);
      } else {
        MI_HIP(hipMemcpyAsync(stage.rhs, rhs.p, sizeof(double) * ni_tot, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        for (int dl = 0; dl < maps.ndl; ++dl)
          if (solve(user, d0 + dl, ni[dl], stage.rhs + ioff[dl], stage.sol + ioff[dl]) != 0)
            raise(MI_ERR_CALLBACK, "interior solve callback failed on subdomain %lld", (long long)(d0 + dl));
        MI_HIP(hipMemcpyAsync(sol.p, stage.sol, sizeof(double) * ni_tot, hipMemcpyHostToDevice, s));
      }
      A_GI.launch(1, sol.p, t1.p, yloc.p, nullptr, s);
    }
    maps.assemble(ctx, n, yloc.p, y, nullptr);
  }
  // New block values on the same patterns (device pointers; each array is the concatenation over this operator's
  // subdomains of what was passed at create): the per-realization update of Example07:162-171 without a host trip.
  void set_values(const double *ii_val, const double *ig_val, const double *gg_val) {
    hipStream_t s = ctx->stream;
    if (ii_val) {
      if (!icg) raise(MI_ERR_BAD_ARG, "set_values: A_II lives in the interior-solve callback of this operator");
      if (icg->A.nnz) MI_HIP(hipMemcpyAsync(icg->A.val.p, ii_val, sizeof(double) * icg->A.nnz, hipMemcpyDeviceToDevice, s));
      icg->refresh_diagonal();
    }
    if (ig_val && A_GI.nnz) {
      MI_HIP(hipMemcpyAsync(A_GI.val.p, ig_val, sizeof(double) * A_GI.nnz, hipMemcpyDeviceToDevice, s));
      hipLaunchKernelGGL(k_permute, dim3(vec_grid(A_IG.nnz)), dim3(NT), 0, s, (long long)A_IG.nnz, ig_perm.p, ig_val, A_IG.val.p);
      MI_HIP(hipGetLastError());
    }
    if (gg_val && A_GG.nnz) MI_HIP(hipMemcpyAsync(A_GG.val.p, gg_val, sizeof(double) * A_GG.nnz, hipMemcpyDeviceToDevice, s));
  }
  // b_schur = b_Γ - Σ_d R_d' A_IΓdd' (A_IIdd^{-1} b_Id)   (get_schur_rhs, EPDD.jl:835-864); b_I: concatenated b_Id
  void schur_rhs(const double *b_I, const double *b_gamma, double *out) {
    hipStream_t s = ctx->stream;
    if (maps.nloc) {
      if (level_solver) {
        level_solver(b_I, sol.p);
      } else if (icg) {
        icg->solve(b_I, sol.p);
      } else {
        MI_HIP(hipMemcpyAsync(stage.rhs, b_I, sizeof(double) * ni_tot, hipMemcpyDeviceToHost, s));
        MI_HIP(hipStreamSynchronize(s));
        for (int dl = 0; dl < maps.ndl; ++dl)
          if (solve(user, d0 + dl, ni[dl], stage.rhs + ioff[dl], stage.sol + ioff[dl]) != 0)
            raise(MI_ERR_CALLBACK, "interior solve callback failed on subdomain %lld", (long long)(d0 + dl));
        MI_HIP(hipMemcpyAsync(sol.p, stage.sol, sizeof(double) * ni_tot, hipMemcpyHostToDevice, s));
      }
      A_GI.launch(0, sol.p, nullptr, yloc.p, nullptr, s);
    }
    maps.assemble(ctx, n, yloc.p, t2.p, nullptr);
    hipLaunchKernelGGL(k_sub, dim3(vec_grid(n)), dim3(NT), 0, s, (int)n, b_gamma, t2.p, out);
    MI_HIP(hipGetLastError());
  }
  // u_Id = A_IIdd^{-1} (b_Id - A_IΓdd u_Γd) for every local subdomain (get_subdomain_solutions, EPDD.jl:1014-1025);
  // b_I / u_I: concatenated over the local subdomains
  void interior_solutions(const double *u_gamma, const double *b_I, double *u_I) {
    hipStream_t s = ctx->stream;
    if (!maps.nloc) return;
    hipLaunchKernelGGL(k_gather, dim3(vec_grid(maps.nloc)), dim3(NT), 0, s, maps.nloc, maps.gidx.p, u_gamma, xcat.p);
    MI_HIP(hipGetLastError());
    A_IG.launch(1, xcat.p, b_I, rhs.p, nullptr, s);  // rhs = b_I - A_IΓ u_Γd
    if (level_solver) {
      level_solver(rhs.p, u_I);
    } else if (icg) {
      icg->solve(rhs.p, u_I);
    } else {
      MI_HIP(hipMemcpyAsync(stage.rhs, rhs.p, sizeof(double) * ni_tot, hipMemcpyDeviceToHost, s));
      MI_HIP(hipStreamSynchronize(s));
      for (int dl = 0; dl < maps.ndl; ++dl)
        if (solve(user, d0 + dl, ni[dl], stage.rhs + ioff[dl], stage.sol + ioff[dl]) != 0)
          raise(MI_ERR_CALLBACK, "interior solve callback failed on subdomain %lld", (long long)(d0 + dl));
      MI_HIP(hipMemcpyAsync(u_I, stage.sol, sizeof(double) * ni_tot, hipMemcpyHostToDevice, s));
    }
  }
  void bytes(int64_t *a, int64_t *d) const override {
    *a = A_IG.bytes() + A_GI.bytes() + A_GG.bytes() + 16ll * ni_tot;
    *d = icg ? icg->A.bytes() : A_IG.bytes();
  }
  void apply_dominant(const double *x) override {
    if (!maps.nloc) return;
    if (icg) {  // the A_II SpMV of the interior CG (x is not used: the operand is the CG direction buffer)
      icg->A.launch(0, icg->u.p, nullptr, icg->c.p, nullptr, ctx->stream, icg->u.p, icg->p_uc.p);
      return;
    }
    hipLaunchKernelGGL(k_gather, dim3(vec_grid(maps.nloc)), dim3(NT), 0, ctx->stream, maps.nloc, maps.gidx.p, x, xcat.p);
    A_IG.launch(0, xcat.p, nullptr, rhs.p, nullptr, ctx->stream);
  }
};

// ------------------------------------------------------------------ apply_global_schur (EPDD.jl:596-625)
struct GlobalSchurOp : Operator {
  int ndom;
  std::vector<CsrDev> A_IG, A_GI;  // per subdomain: (n_i x n_Γ) and its transpose (n_Γ x n_i)
  CsrDev A_GG;
  std::vector<int> ni, ioff;
  int ni_tot = 0;
  mi_interior_solve_fn solve; void *user;
  DevBuf<double> rhs, sol;
  HostStage stage;

  std::unique_ptr<InteriorCg> icg;  // device interior solve (no callback): IterativeSolvers.cg restated, all subdomains at once
  std::function<void(const double *, double *)> level_solver;   // as in MatfreeSchurOp

  GlobalSchurOp(mi_ctx_s *c, int64_t ndom_, int64_t n_gamma, const int64_t *n_i, const int64_t *const *ig_ptr,
                const int64_t *const *ig_idx, const double *const *ig_val, const int64_t *gg_ptr, const int64_t *gg_idx,
                const double *gg_val, mi_interior_solve_fn f, void *u, int base, const int64_t *const *ii_ptr = nullptr,
                const int64_t *const *ii_idx = nullptr, const double *const *ii_val = nullptr, double reltol = 0.0)
      : Operator(c, n_gamma), ndom((int)ndom_), solve(f), user(u) {
    if ((!f && !ii_ptr) || ndom_ <= 0 || !n_i || !ig_ptr || !ig_idx || !ig_val || !gg_ptr)
      raise(MI_ERR_BAD_ARG, "mi_schur_global_create: NULL argument");
    if (ii_ptr && (!ii_idx || !ii_val || !(reltol > 0.0))) raise(MI_ERR_BAD_ARG, "device interior: bad A_II arrays / reltol");
    A_IG.resize(ndom); A_GI.resize(ndom);
    int64_t itot = 0;
    for (int d = 0; d < ndom; ++d) {
      ioff.push_back((int)itot); ni.push_back((int)n_i[d]); itot += n_i[d];
      if (itot >= INT32_MAX) raise(MI_ERR_BAD_ARG, "interior too large");
      HostCsr gi_d = host_csr(n_gamma, n_i[d], ig_ptr[d], ig_idx[d], ig_val[d], base);
      A_GI[d].upload(gi_d, c->stream);
      A_IG[d].upload(transpose(gi_d), c->stream);
    }
    ni_tot = (int)itot;
    A_GG.upload(host_csr(n_gamma, n_gamma, gg_ptr, gg_idx, gg_val, base), c->stream);
    rhs.alloc(ni_tot + 1); sol.alloc(ni_tot + 1);
    if (ii_ptr) {
      HostCsr ii;
      for (int d = 0; d < ndom; ++d) append_block(ii, host_csr(ni[d], ni[d], ii_ptr[d], ii_idx[d], ii_val[d], base), ioff[d], ni_tot);
      if (ii.rowptr.empty()) ii.rowptr = {0};
      icg.reset(new InteriorCg);
      icg->build(c, ii, ioff, ni, reltol);
    } else {
      stage.ensure(ni_tot);
    }
  }
  bool graph_safe() const override { return false; }
  bool writes_y_once() const override { return false; }  // y = A_ΓΓ x, then ndom in-place `y -= A_IΓd' v` passes
  void apply(const double *x, double *y, const int *) override {
    hipStream_t s = ctx->stream;
    A_GG.launch(0, x, nullptr, y, nullptr, s);  // Sx = A_ΓΓ * x
    for (int d = 0; d < ndom; ++d) A_IG[d].launch(0, x, nullptr, rhs.p + ioff[d], nullptr, s);
    if (level_solver) {
      level_solver(rhs.p, sol.p);
    } else if (icg) {
      icg->solve(rhs.p, sol.p);
    } else {
      MI_HIP(hipMemcpyAsync(stage.rhs, rhs.p, sizeof(double) * ni_tot, hipMemcpyDeviceToHost, s));
      MI_HIP(hipStreamSynchronize(s));
      for (int d = 0; d < ndom; ++d)
        if (solve(user, d, ni[d], stage.rhs + ioff[d], stage.sol + ioff[d]) != 0)
          raise(MI_ERR_CALLBACK, "interior solve callback failed on subdomain %d", d);
      MI_HIP(hipMemcpyAsync(sol.p, stage.sol, sizeof(double) * ni_tot, hipMemcpyHostToDevice, s));
    }
    for (int d = 0; d < ndom; ++d) A_GI[d].launch(1, sol.p + ioff[d], y, y, nullptr, s);  // Sx .-= A_IΓd' * v
  }
  // A_IId^{-1} v for every subdomain (device pointers; v, out: concatenated over the subdomains)
  void interior_solve(const double *v, double *out) {
    hipStream_t s = ctx->stream;
    if (level_solver) { level_solver(v, out); return; }
    if (icg) { icg->solve(v, out); return; }
    MI_HIP(hipMemcpyAsync(stage.rhs, v, sizeof(double) * ni_tot, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    for (int d = 0; d < ndom; ++d)
      if (solve(user, d, ni[d], stage.rhs + ioff[d], stage.sol + ioff[d]) != 0)
        raise(MI_ERR_CALLBACK, "interior solve callback failed on subdomain %d", d);
    MI_HIP(hipMemcpyAsync(out, stage.sol, sizeof(double) * ni_tot, hipMemcpyHostToDevice, s));
  }
  // b_schur = b_Γ - Σ_d A_IΓd' (A_IId \ b_Id), subdomain by subdomain (get_schur_rhs, EPDD.jl:798-821)
  void schur_rhs(const double *b_I, const double *b_gamma, double *out) {
    hipStream_t s = ctx->stream;
    interior_solve(b_I, sol.p);
    MI_HIP(hipMemcpyAsync(out, b_gamma, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    for (int d = 0; d < ndom; ++d) A_GI[d].launch(1, sol.p + ioff[d], out, out, nullptr, s);
  }
  // u_Id = A_IId \ (b_Id - A_IΓd u_Γ) (get_subdomain_solutions, EPDD.jl:1014-1025)
  void interior_solutions(const double *u_gamma, const double *b_I, double *u_I) {
    hipStream_t s = ctx->stream;
    for (int d = 0; d < ndom; ++d) A_IG[d].launch(1, u_gamma, b_I + ioff[d], rhs.p + ioff[d], nullptr, s);
    interior_solve(rhs.p, u_I);
  }
  void bytes(int64_t *a, int64_t *dd) const override {
    int64_t t = A_GG.bytes();
    for (int d = 0; d < ndom; ++d) t += A_IG[d].bytes() + A_GI[d].bytes();
    *a = t + 16ll * ni_tot; *dd = A_GG.bytes();
  }
  void apply_dominant(const double *) override {}
};

}  // namespace mi

struct mi_op_s {
  std::unique_ptr<mi::Operator> impl;
  mi::DevBuf<double> hx, hy;  // staging for host-pointer mode
};
