/* mi355schur.h — C ABI of libmi355schur: the MI355X (gfx950) Schur-complement PCG hot path.
 *
 * Drop-in boundary for the domain-decomposition hot path of venkovic/julia-phd-krylov-spdes.
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference root; "EPDD.jl" = Fem/EllipticPdeDomainDecomposition.jl). The reference is pure
 * Julia: a `ccall` shim (julia/MI355Schur.jl, shown in INTEGRATION.md) binds exactly these
 * symbols and re-exports the reference's own function names on top of them.
 *
 * Conventions
 *   - Plain C: opaque handles, pointers and sizes. No C++ / torch types cross this ABI.
 *   - Every function returns an int status: MI_OK (0) or a negative MI_ERR_* code;
 *     mi_last_error() returns a thread-local description of the last failure. No exception
 *     crosses the ABI (Julia side turns non-zero into `error(...)`; MI_ERR_SINGULAR maps to
 *     LinearAlgebra.SingularException as thrown by `WtAW \ mu`, defcg.jl:53,273).
 *   - Arithmetic is IEEE fp64 throughout; indices passed in are int64 with an explicit
 *     `index_base` (1 for Julia arrays, 0 for C/numpy); they are stored as int32 on the device.
 *   - `*_create` functions read HOST arrays and copy what they need to the device; the caller
 *     keeps ownership and may free or mutate its arrays afterwards (Julia GC owns them).
 *   - Vector arguments of apply/solve calls (`x`, `y`, `b`, `W`) are host pointers by default,
 *     or device pointers after mi_ctx_set_pointer_mode(ctx, MI_PTR_DEVICE). Scalar/history
 *     outputs (`it`, `res_norm`, dot results) are always host pointers.
 *   - A context owns one HIP stream; handles are thread-compatible, not thread-safe. In
 *     multi-GPU use there is one process (or thread) and one context per GPU and every rank
 *     enters collective calls (applies and solves on a SHARDED operator, see below) together.
 *   - There is no CPU fallback: with no gfx950 device mi_ctx_create fails with MI_ERR_NO_DEVICE.
 */
#ifndef MI355SCHUR_H
#define MI355SCHUR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_ERR_BAD_ARG (-1)      /* NULL/negative/out-of-range argument, index out of bounds          */
#define MI_ERR_HIP (-2)          /* a HIP runtime call failed                                          */
#define MI_ERR_SINGULAR (-3)     /* WtAW is singular to working precision (Julia: SingularException)   */
#define MI_ERR_RES_CAPACITY (-4) /* res_norm capacity exhausted (Julia: BoundsError on res_norm[it])   */
#define MI_ERR_COMM (-5)         /* RCCL not loadable / communicator failure                           */
#define MI_ERR_NO_DEVICE (-6)    /* no usable gfx950 device                                            */
#define MI_ERR_CALLBACK (-7)     /* the interior-solve callback reported failure                       */
#define MI_ERR_BOUNDS (-8)       /* eigCG family: an index the reference would throw BoundsError on     */

#define MI_PTR_HOST 0
#define MI_PTR_DEVICE 1

typedef struct mi_ctx_s *mi_ctx_t; /* device + stream + workspaces (+ RCCL communicator)            */
typedef struct mi_op_s *mi_op_t;   /* anything usable as `A` (A*x, mul!) or as `M` (M \ r)          */
typedef struct mi_event_s *mi_event_t;
typedef struct mi_plan_s *mi_plan_t; /* index half of prepare_local_schurs for a fixed mesh / partition          */

/* Interior solve callback: sol = A_II[idom]^{-1} rhs on HOST memory, called from the calling
 * thread only (Julia @cfunction safe). Replaces `IterativeSolvers.cg(A_IIdd, rhs; Pl, reltol)`
 * inside apply_local_schur / apply_global_schur (EPDD.jl:609-619, 648-650). Return 0 on success. */
typedef int (*mi_interior_solve_fn)(void *user, int64_t idom, int64_t n, const double *rhs, double *sol);

/* ---------------------------------------------------------------- library / context */
int mi_version(void);                 /* MAJOR*10000 + MINOR*100 + PATCH */
const char *mi_last_error(void);
int mi_device_count(int *count);
int mi_ctx_create(int device, mi_ctx_t *ctx);
int mi_ctx_destroy(mi_ctx_t ctx);
int mi_ctx_set_pointer_mode(mi_ctx_t ctx, int mode);
int mi_ctx_set_stream(mi_ctx_t ctx, void *hip_stream); /* borrow a hipStream_t; NULL = context's own */
int mi_ctx_get_stream(mi_ctx_t ctx, void **hip_stream);
int mi_ctx_synchronize(mi_ctx_t ctx);
/* Iterations per follow-up hipGraph replay between host convergence checks (default 8; the first replay of a
 * solve is sized from the previous solve with the same operators). 0 disables graphs (eager launches, host check
 * every iteration). Results do not depend on it. */
int mi_ctx_set_chunk(mi_ctx_t ctx, int iterations_per_graph);

/* ---------------------------------------------------------------- multi-GPU (RCCL over xGMI)
 * The Γ-interface sum over subdomains — the reference's (dead) `@distributed (+) for idom`,
 * Fem/EllipticPdePllDomainDecomposition.jl:10-14 — is one ncclAllReduce(sum, fp64, n_Γ).
 * Rank 0 calls mi_comm_unique_id and ships the 128 bytes to the other ranks by any means
 * (torch.distributed / MPI / Julia Distributed); every rank then calls mi_ctx_comm_init. */
#define MI_COMM_ID_BYTES 128
int mi_comm_unique_id(void *id_out);
int mi_ctx_comm_init(mi_ctx_t ctx, const void *id, int rank, int n_ranks);
int mi_ctx_comm_destroy(mi_ctx_t ctx);
/* The one-shot PEER EXCHANGE (the same `(+)` over subdomains, EllipticPdePllDomainDecomposition.jl:10-14, without a
 * collective library in the loop): every rank owns an arena of peer-visible device memory; after a sharded launch a rank
 * WRITES its entries of the contribution table into every rank's arena (xGMI peer stores), releases them, stores its
 * exchange number into one flag per arena and continues when its own flags have all arrived (bounded wait: an expired
 * wait makes the solve return MI_ERR_COMM, never hang). No arithmetic on the way: the tables of the ranks are disjoint,
 * so the bits are the single-GPU loop's. Everything is kernels on the context's stream — captured into the iteration
 * graphs. Call order, on every rank: mi_ctx_peer_init; mi_ctx_peer_export and ship the MI_PEER_HANDLE_BYTES to the other
 * ranks by any means (torch.distributed / MPI / Julia Distributed); mi_ctx_peer_import for every other rank (ranks that
 * are contexts of the SAME process pass the exporter's `base` instead of a handle); mi_ctx_peer_ready. With RCCL also
 * attached the peer exchange carries the tables of the folded PCG launches and RCCL the generic all-reduces; alone it
 * carries both. mi_ctx_set_exchange(ctx, 0) switches it off again (RCCL for everything), 1 on. Sharded operators must be
 * created in the same order with the same sizes on every rank (their tables sit at equal offsets of every arena).
 * Who produces and who waits: the folded PCG launches store their results into every arena themselves (write-through
 * stores; the last tile of a launch stores the flags) and a one-wave kernel behind the launch waits — mode 1, safe when
 * several ranks share a GPU (in-process ranks, tests). Mode 2 (mi_ctx_set_exchange(ctx, 2), fine-grained arena only —
 * otherwise it is mode 1): no kernel in between, the NEXT launch polls the flags with its first matrix loads already in
 * flight. A waiting launch keeps its compute units, so mode 2 is for one GPU per rank; bench.py selects it then. */
#define MI_PEER_HANDLE_BYTES 64
int mi_ctx_peer_init(mi_ctx_t ctx, int rank, int n_ranks, int64_t arena_bytes /* 0: 64 MiB */);
int mi_ctx_peer_export(mi_ctx_t ctx, void *handle_out /* MI_PEER_HANDLE_BYTES */, void **base_out);
int mi_ctx_peer_import(mi_ctx_t ctx, int rank, const void *handle, void *same_process_base);
int mi_ctx_peer_ready(mi_ctx_t ctx);
int mi_ctx_set_exchange(mi_ctx_t ctx, int use_peer_exchange);
/* Introspection for tests and benchmarks: which path ran. NO_GRAPH: 1 when this context launches eagerly (a collective
 * that cannot be captured); PEER_EXCHANGE: 0 off, 1 on, 2 on with a fine-grained arena, 3 on with in-launch waits; GRAPH_REPLAYS: hipGraphLaunch
 * calls of the solvers so far; EXCHANGES: exchanges this rank has signalled (synchronises the stream). */
#define MI_QUERY_NO_GRAPH 0
#define MI_QUERY_PEER_EXCHANGE 1
#define MI_QUERY_GRAPH_REPLAYS 2
#define MI_QUERY_EXCHANGES 3
#define MI_QUERY_EXPERIMENTAL 5    /* 1: built with `make EXPERIMENTAL=1` (persistent on-chip PCG, rocBLAS/rocSOLVER set-up route) */
#define MI_QUERY_SPECTRAL_PINV 4   /* blocks of mi_nn_pinv that went to the eigen-decomposition (rocSOLVER) so far, process-wide */
int mi_ctx_query(mi_ctx_t ctx, int what, int64_t *out);
/* Test facility: in-process ranks. `n_ranks` contexts of ONE process (one host thread each, typically all on the same
 * device) call mi_ctx_loopback_init with the same group and then behave like ranks of a multi-GPU job: operators built
 * from a slice of the subdomains are sharded. RCCL refuses two ranks on one device; this is how the sharded paths are
 * exercised on a single-GPU box (tests/test_gpu_multirank.py, tests/test_gpu_parity.py). The group joins its contexts by
 * the peer exchange above (same-process arenas): device-side flags, graph-captured like the production path. Kernels of
 * one rank then wait for kernels of another, which needs one hardware queue per rank and copies that do not queue up
 * behind each other: set GPU_MAX_HW_QUEUES >= n_ranks and GPU_FORCE_BLIT_COPY_SIZE=1048576 in the environment BEFORE the
 * first HIP call (the runtime's defaults are 4 queues and 16 KiB). mi_ctx_loopback_init runs trial exchanges with a short
 * bound; if one expires (or GPU_MAX_HW_QUEUES < n_ranks) the whole group falls back to host-side rendezvous with eager
 * launches, as does mode 1. mode: 0 automatic, 1 host rendezvous. One process per GPU needs neither setting. */
int mi_loopback_group_create(int n_ranks, void **group);
int mi_loopback_group_destroy(void *group);
int mi_ctx_loopback_init(mi_ctx_t ctx, void *group, int rank);
int mi_loopback_group_set_mode(void *group, int mode);
int mi_ctx_allreduce_sum(mi_ctx_t ctx, double *buf, int64_t n); /* in place, follows pointer mode */

/* ---------------------------------------------------------------- operators
 * mi_csr_create — a symmetric `SparseMatrixCSC{Float64,Int64}` used as `A` in cg/pcg/defcg/defpcg
 * (`A*x` cg.jl:28,83; `mul!(Ap,A,p)` cg.jl:36,93). Pass colptr/rowval/nzval as
 * rowptr/colidx/val: for a symmetric matrix the CSC arrays are its CSR arrays.
 * A non-symmetric CSC matrix passed this way yields `A'*x` (the stdlib's gather form). */
int mi_csr_create(mi_ctx_t ctx, int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                  const int64_t *colidx, const double *val, int index_base, mi_op_t *op);

/* mi_diag_create — `M \ r` with M = diag: z = dinv .* r, or the identity when dinv == NULL
 * (Example01's AMG preconditioner is out of scope; Jacobi / none stand in). */
int mi_diag_create(mi_ctx_t ctx, int64_t n, const double *dinv, mi_op_t *op);

/* mi_schur_assembled_create — `apply_local_schurs(Sd, ind_Γd_Γ2l, node_Γ_cnt, x)`, EPDD.jl:761-785:
 * Sx = Σ_d R_d' S_d R_d x with dense local Schur complements.
 *   Sd[d]          n_gamma_d[d]^2 doubles, column-major (Julia `Array(Sd[d])`)
 *   gather_idx[d]  n_gamma_d[d] entries: Γ index of Γ_d slot l (the flattened Dict ind_Γd_Γ2l[d]:
 *                  gather_idx[d][lΓd] = lΓ). Pass the lists of ALL subdomains on every rank, also outside
 *                  [dom_begin, dom_end): they fix each subdomain's slot in the per-node contribution table, so that
 *                  the ranks' tables are disjoint and their all-reduce reproduces the single-GPU Γ-sum bit for bit
 * The local slice [dom_begin, dom_end) of the ndom subdomains is applied by this rank. On a context with a
 * communicator a proper slice makes the operator SHARDED: its applies end with one RCCL all-reduce (of the table of
 * per-subdomain contributions, so the Γ-sum keeps the single-GPU order and bits). An operator created with
 * dom_begin = 0, dom_end = ndom is REPLICATED and never communicates, with or without a communicator (single-GPU
 * use; or e.g. the Neumann-Neumann blocks copied to every rank while S is sharded: one all-reduce per PCG
 * iteration instead of two). Arrays of subdomains outside the slice may be NULL. */
int mi_schur_assembled_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d,
                              const int64_t *const *gather_idx, const double *const *Sd, int index_base,
                              int64_t dom_begin, int64_t dom_end, mi_op_t *op);

/* mi_nn_create — `NeumannNeumannSchurPreconditioner(ΠSd, ind_Γd_Γ2l, node_Γ_cnt)` (EPDD.jl:1111-1137)
 * used through `Πnn \ r` / `ldiv!` = apply_neumann_neumann_schur (EPDD.jl:1361-1403):
 * z = Σ_d R_d' D_d ΠS_d D_d R_d r, D = diag(1 ./ node_Γ_cnt). ΠSd[d] column-major dense. */
int mi_nn_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d,
                 const int64_t *const *gather_idx, const double *const *PiSd, const int64_t *node_gamma_cnt,
                 int index_base, int64_t dom_begin, int64_t dom_end, mi_op_t *op);

/* mi_schur_matfree_create — `apply_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt, x)`,
 * EPDD.jl:711-747 (per subdomain apply_local_schur, EPDD.jl:639-654):
 * S_d x_d = A_ΓΓdd x_d − A_IΓdd' (A_IIdd^{-1} (A_IΓdd x_d)). The three sparse products run on the
 * device; A_IIdd^{-1} is the host callback (BASELINE north_star: interior solve stays on the host).
 *   A_IΓdd[d]: CSC arrays (colptr n_gamma_d[d]+1, rowval, nzval) of the n_i[d] x n_gamma_d[d] block
 *   A_ΓΓdd[d]: CSC arrays of the symmetric n_gamma_d[d] x n_gamma_d[d] block */
int mi_schur_matfree_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d,
                            const int64_t *n_i, const int64_t *const *gather_idx,
                            const int64_t *const *ig_colptr, const int64_t *const *ig_rowval,
                            const double *const *ig_nzval, const int64_t *const *gg_colptr,
                            const int64_t *const *gg_rowval, const double *const *gg_nzval,
                            mi_interior_solve_fn solve, void *user, int index_base,
                            int64_t dom_begin, int64_t dom_end, mi_op_t *op);

/* mi_schur_matfree_device_create — the same operator with the interior solve ON THE DEVICE, as the reference
 * does it: `IterativeSolvers.cg(A_IIdd, A_IΓdd*xd, reltol=reltol)` from a zero initial guess (EPDD.jl:648-650,
 * default reltol 1e-9; no preconditioner — the reference's AMG `Pl` is out of scope). All local subdomains are
 * iterated together on the block-diagonal A_II with per-subdomain scalars; a subdomain stops when its residual
 * is <= reltol * ||rhs|| or after n_i iterations.
 *   A_IIdd[d]: CSC arrays of the symmetric n_i[d] x n_i[d] interior block. */
int mi_schur_matfree_device_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_gamma_d,
                                   const int64_t *n_i, const int64_t *const *gather_idx,
                                   const int64_t *const *ii_colptr, const int64_t *const *ii_rowval,
                                   const double *const *ii_nzval, const int64_t *const *ig_colptr,
                                   const int64_t *const *ig_rowval, const double *const *ig_nzval,
                                   const int64_t *const *gg_colptr, const int64_t *const *gg_rowval,
                                   const double *const *gg_nzval, double reltol, int index_base,
                                   int64_t dom_begin, int64_t dom_end, mi_op_t *op);

/* mi_schur_global_create — `apply_global_schur(A_IId, A_IΓd, A_ΓΓ, x)`, EPDD.jl:596-625:
 * Sx = A_ΓΓ x − Σ_d A_IΓd' (A_IId^{-1} (A_IΓd x)), Γ-global column indices (no gather maps).
 *   A_IΓd[d]: CSC arrays of the n_i[d] x n_gamma block;  A_ΓΓ: CSC arrays, symmetric. */
int mi_schur_global_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_i,
                           const int64_t *const *ig_colptr, const int64_t *const *ig_rowval,
                           const double *const *ig_nzval, const int64_t *gg_colptr,
                           const int64_t *gg_rowval, const double *gg_nzval,
                           mi_interior_solve_fn solve, void *user, int index_base, mi_op_t *op);
/* The same with the interior solves on the device, as for mi_schur_matfree_device_create: `IterativeSolvers.cg(A_IId[idom],
 * A_IΓd[idom]*x)` (EPDD.jl:609-619) restated; the reference calls it with the package default reltol = sqrt(eps). */
int mi_schur_global_device_create(mi_ctx_t ctx, int64_t ndom, int64_t n_gamma, const int64_t *n_i,
                                  const int64_t *const *ii_colptr, const int64_t *const *ii_rowval,
                                  const double *const *ii_nzval, const int64_t *const *ig_colptr,
                                  const int64_t *const *ig_rowval, const double *const *ig_nzval,
                                  const int64_t *gg_colptr, const int64_t *gg_rowval, const double *gg_nzval,
                                  double reltol, int index_base, mi_op_t *op);

/* mi_schur_interior_precond — the `precond` / `preconds` keyword of apply_local_schur / apply_global_schur (EPDD.jl:648-650,
 * 609-619: `IterativeSolvers.cg(A_IIdd, rhs, Pl=precond, reltol=reltol)`) for an operator made by one of the two
 * *_device_create calls above. kind 0: none (the default, plain CG); kind 1: Pl = Diagonal(A_IIdd) (Jacobi). The reference's own
 * choice, an AMG hierarchy from Preconditioners.jl, is outside the hot path; the diagonal costs nothing per
 * iteration (it is folded into the update kernel) and about halves the iteration counts on lognormal coefficient fields. */
int mi_schur_interior_precond(mi_op_t op, int kind);
/* Diagnostic: iterations the interior CG of such an operator has run so far (slowest subdomain, rounded up to replays). */
int mi_schur_interior_iterations(mi_op_t op, int64_t *iterations);

int mi_op_size(mi_op_t op, int64_t *n);
/* y = A*x (operator) or y = M \ x (preconditioner); x and y must not alias. */
int mi_op_apply(mi_op_t op, const double *x, double *y);
/* Algorithmic bytes of one apply (SURVEY.md §8d formulas) and of its dominant kernel alone. */
int mi_op_bytes(mi_op_t op, int64_t *bytes_apply, int64_t *bytes_dominant_kernel);
/* Diagnostic: launch only the dominant kernel of the apply `reps` times (x as in mi_op_apply). */
int mi_op_apply_dominant(mi_op_t op, const double *x, int reps);
/* Diagnostic: average duration (microseconds) of the dominant kernel over `reps` launches replayed from one
 * graph, measured with HIP events on the context's stream (what bench.py reports as roofline.us_per_launch). */
int mi_op_time_dominant(mi_op_t op, const double *x, int reps, double *us_per_launch);
int mi_op_destroy(mi_op_t op);

/* ---------------------------------------------------------------- BLAS-1 on the device
 * `dot`, `norm2`, `axpy!`, `axpby!` as imported at RecyclingKrylovSolvers.jl:3 (level-1 drop-ins). */
int mi_dot(mi_ctx_t ctx, int64_t n, const double *x, const double *y, double *result);
int mi_norm2(mi_ctx_t ctx, int64_t n, const double *x, double *result);
int mi_axpy(mi_ctx_t ctx, int64_t n, double a, const double *x, double *y);            /* y += a x     */
int mi_axpby(mi_ctx_t ctx, int64_t n, double a, const double *x, double b, double *y); /* y = a x + b y */

/* ---------------------------------------------------------------- solvers (whole loop on the device)
 * Reference signatures (RecyclingKrylovSolvers/cg.jl:14-18, 67-72; defcg.jl:24-29, 242-248):
 *     cg(A,b,x;maxit=0)  pcg(A,b,x,M;maxit=0)  defcg(A,b,x,W;maxit=0)  defpcg(A,b,x,W,M;maxit=0)
 *     -> (x, it, res_norm[1:it])
 * x is read as the initial guess and overwritten with the solution (the reference mutates x).
 * maxit == 0 means n (cg.jl:25). eps is the module constant 1e-7 (RecyclingKrylovSolvers.jl:21)
 * made explicit; pass eps <= 0 for that default. Stop rule (cg.jl:34, 91): iterate while
 * it < maxit && res_norm[it] > eps*norm2(b), with res_norm[it] = sqrt(r'r) of the recurrence
 * residual; `it` counts from 1. res_norm receives it entries (capacity res_cap >= it required,
 * else MI_ERR_RES_CAPACITY after the solve completed). W is n x nvec, column-major. */
int mi_cg(mi_op_t A, const double *b, double *x, int64_t maxit, double eps, double *res_norm,
          int64_t res_cap, int64_t *it);
int mi_pcg(mi_op_t A, mi_op_t M, const double *b, double *x, int64_t maxit, double eps,
           double *res_norm, int64_t res_cap, int64_t *it);
int mi_defcg(mi_op_t A, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit,
             double eps, double *res_norm, int64_t res_cap, int64_t *it);
int mi_defpcg(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec,
              int64_t maxit, double eps, double *res_norm, int64_t res_cap, int64_t *it);

/* ---------------------------------------------------------------- on-device block assembly (per-realization update)
 * Replaces the element loop of `prepare_local_schurs(cells, points, epart, ..., coeff, f, uexact)`
 * (EPDD.jl:389-546) when only the nodal coefficient vector `a` changes between calls — Example07:162-171 redoes the
 * whole loop on the host for every realization. The caller prepares the index half once:
 *   cells  3 x nel node indices (row i contiguous: vertex i of every element), `index_base` 0 or 1
 *   G      9 x nel, G[3i+j] = Δy_i Δy_j + Δx_i Δx_j (:448-453, 470);  area  nel (:456)
 *   ue     3 x nel, uexact at vertex i (Dirichlet lifting, :498-507); be  3 x nel, Δb_i = (2f_i+f_j+f_k) Area/12 (:516-525)
 *   cptr / ccode: entry k of the output is the sum, IN ORDER, of the contributions ccode[cptr[k] .. cptr[k+1]), each
 *   coded 12*element + 3i + j (ΔK_ij = Δa G_ij/4/Area with Δa = (a_1+a_2+a_3)/3; for k >= n_matrix_entries the
 *   lifting term -(ΔK_ij ue_i)) or 12*element + 9 + i (Δb_i); 0-based. The order inside an entry is the order in which
 *   `sparse(I,J,V)` / `b[k] +=` met the terms (ascending element), so results are bit-identical to the host loop.
 * mi_assembly_run: a_nodal (n_node) -> values (n_entries); host or device pointers per the context's pointer mode
 *   (device pointers: asynchronous on the context's stream, like mi_op_apply).
 * mi_schur_matfree_set_values: new block values for a matrix-free operator, same sparsity as at create; each array is
 *   the concatenation over the operator's subdomains [dom_begin, dom_end) of the arrays passed at create (A_IIdd: its
 *   CSC/CSR values; A_IΓdd: CSC values; A_ΓΓdd); NULL leaves a block unchanged; ii_val needs the device interior solve.
 * mi_schur_matfree_rhs: `get_schur_rhs` (EPDD.jl:835-864): b_schur = b_Γ - Σ_d R_d' A_IΓdd' (A_IIdd \ b_Id) with the
 *   operator's own interior solve; b_I is the concatenation of the b_Id. Collective on a sharded operator. Also accepts
 *   a global Schur operator (mi_schur_global_*create): the Γ-global form, EPDD.jl:798-821; likewise
 *   mi_schur_matfree_interior_solutions (there it is the reference's own signature, A_IΓd with global columns). */
int mi_assembly_plan_create(mi_ctx_t ctx, int64_t nel, int64_t n_node, const int64_t *cells, int index_base,
                            const double *G, const double *area, const double *ue, const double *be, int64_t n_entries,
                            int64_t n_matrix_entries, const int64_t *cptr, const int64_t *ccode, mi_plan_t *plan);
int mi_assembly_run(mi_plan_t plan, const double *a_nodal, double *values);
int mi_assembly_plan_destroy(mi_plan_t plan);
int mi_schur_matfree_set_values(mi_op_t op, const double *ii_val, const double *ig_val, const double *gg_val);
int mi_schur_matfree_rhs(mi_op_t op, const double *b_I, const double *b_gamma, double *b_schur);
/* `get_subdomain_solutions(u_Γ, A_IId, A_IΓd, b_Id)` (EPDD.jl:1014-1025): u_Id = A_IIdd \ (b_Id - A_IΓdd u_Γd) for the
 * operator's subdomains with its own interior solve; b_I and u_I are concatenations over those subdomains. */
int mi_schur_matfree_interior_solutions(mi_op_t op, const double *u_gamma, const double *b_I, double *u_I);

/* ---------------------------------------------------------------- set-up of the assembled mode on the device
 * mi_schur_setup_* — `assemble_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, ...)` (EPDD.jl:667-695): the dense local Schur complements
 * S_d = A_ΓΓdd - A_IΓdd' A_IIdd^{-1} A_IΓdd of ALL subdomains, symmetrised from the upper triangle as the reference does
 * (`Symmetric(Array(...))`, :692), and — optionally — the condensed right-hand sides w_d = A_IΓdd' (A_IIdd \ b_Id) that
 * `get_schur_rhs` subtracts from b_Γ (EPDD.jl:853-861). The reference applies apply_local_schur (interior CG, reltol 1e-9) to
 * every unit vector; here the interior is eliminated exactly, level by level over breadth-first levels grown from Γ_d, with
 * hand-written kernels: Z_k = T_k^{-1} by block Gauss-Jordan WITHOUT pivoting on the fp64 matrix cores (csrc/setup_gj.hpp;
 * every T_k is SPD), one hipGraph replay per realization — a documented deviation that only tightens S_d (measured against
 * the reference's semantics: tests/test_gpu_setup.py::test_device_setup_against_the_reference_semantics). Break-down: an
 * interior block that is singular or not positive definite (the reference's CHOLMOD paths would throw PosDefException)
 * shows as Inf / NaN in S_d: host-pointer mode returns MI_ERR_SINGULAR from run; device-pointer mode is asynchronous and
 * the next mi_nn_pinv on such blocks returns MI_ERR_SINGULAR. (MI355_SETUP_LIB=1 in an EXPERIMENTAL build selects a
 * rocBLAS / rocSOLVER block-Cholesky route instead; there potrf's info is what host-pointer mode reports.)
 *   create: the sparsity of the blocks (CSC colptr / rowval as for mi_schur_matfree_device_create), once per mesh / partition
 *   run   : one realization. ii_val / ig_val / gg_val: the concatenations over the subdomains of the blocks' CSC nzval arrays
 *           (the layout mi_assembly_run produces and mi_schur_matfree_set_values takes); b_I: concatenated b_Id or NULL.
 *           Sd receives the concatenated column-major n_Γd x n_Γd blocks, w (may be NULL) the concatenated w_d. Host or
 *           device pointers per the context's pointer mode; with device pointers the call is asynchronous on the stream.
 * mi_nn_pinv — `prepare_neumann_neumann_schur_precond(Sd, ...)` (EPDD.jl:1201-1220): ΠS_d = pinv(S_d, rtol) of the symmetric
 * blocks; rtol <= 0 means sqrt(eps(Float64)), the reference's value. LinearAlgebra.pinv drops the singular values
 * <= rtol * the largest. Three routes, tried in this order per block, all giving that result: (1) 1/||S^{-1}||_inf >
 * rtol ||S||_inf proves that nothing is dropped: the pseudo-inverse is the inverse (the Gauss-Jordan kernels of the set-up);
 * (2) floating subdomains, ||S 1||_inf <= rtol ||S||_inf: S^+ = (S + α u u')^{-1} - u u'/α, u = 1/sqrt(n), α = ||S||_inf, the
 * same kernels and the same certificate on the shifted matrix; (3) anything else (rank deficiency > 1): eigen-decomposition
 * (rocSOLVER dsyevd, bound with dlopen; singular values = |eigenvalues|). MI_QUERY_SPECTRAL_PINV counts route (3).
 * mi_dense_set_blocks — new blocks (concatenated, column-major) for an existing mi_schur_assembled / mi_nn operator on the same
 * maps: the per-realization update of S (Example07:180-199) without re-creating the operator. */
typedef struct mi_setup_s *mi_setup_t;
int mi_schur_setup_create(mi_ctx_t ctx, int64_t ndom, const int64_t *n_gamma_d, const int64_t *n_i,
                          const int64_t *const *ii_colptr, const int64_t *const *ii_rowval,
                          const int64_t *const *ig_colptr, const int64_t *const *ig_rowval,
                          const int64_t *const *gg_colptr, const int64_t *const *gg_rowval, int index_base,
                          mi_setup_t *plan);
int mi_schur_setup_run(mi_setup_t plan, const double *ii_val, const double *ig_val, const double *gg_val,
                       const double *b_I, double *Sd, double *w);
/* Level solves — the interior solve of the matrix-free applies as the north-star words it ("the A_II^{-1} interior solve stays
 * ... sparse Cholesky"): EXACT, on the device. mi_schur_setup_keep_levels(plan, 1) makes every following run keep the
 * inverses Z_k of all breadth-first levels (block LDL' of A_IIdd; Σ n_level² doubles — 6.2 GB at config 3); then
 * mi_schur_setup_interior_solve: u = A_IIdd \ f for ALL subdomains at once (f, u: concatenated interior vectors in the order
 * of b_I), a forward and a backward sweep over the kept inverses, one hipGraph replay — in place of the reference's inexact
 * `IterativeSolvers.cg(A_IIdd, rhs; reltol)` (EPDD.jl:609-619, 648-650, 813-815, 1021) and of its thousands of A_II SpMVs.
 * mi_schur_matfree_interior_levels(op, plan): a matrix-free or global Schur operator built on the same subdomains uses
 * these solves for every interior solve (apply, get_schur_rhs, get_subdomain_solutions); plan = NULL restores its own
 * (callback / interior CG). The plan must outlive the operator's use of it; values are those of the plan's last run. */
int mi_schur_setup_keep_levels(mi_setup_t plan, int on);
int mi_schur_setup_interior_solve(mi_setup_t plan, const double *f, double *u);
int mi_schur_matfree_interior_levels(mi_op_t op, mi_setup_t plan);
int mi_schur_setup_destroy(mi_setup_t plan);
int mi_nn_pinv(mi_ctx_t ctx, int64_t ndom, const int64_t *n_gamma_d, const double *Sd, double rtol, double *PiSd);
int mi_dense_set_blocks(mi_op_t op, const double *blocks);

/* ---------------------------------------------------------------- eigCG family and Init-CG (recycling solvers)
 * Reference signatures (RecyclingKrylovSolvers/eigcg.jl:27-33, 143-150; defcg.jl:111-116, 337-343; initcg.jl:28-33,
 * 106-111; callers: Example09_DefPcgMcmcStochasticEllipticPde_Functions.jl:314, 345, 364 on the NN-preconditioned
 * Schur system, with nvec = floor(1.25 ndom), spdim = 3 ndom):
 *     eigcg(A,b,x,nvec,spdim;maxit=0)      eigpcg(A,b,x,M,nvec,spdim;maxit=0)
 *     eigdefcg(A,b,x,W,spdim;maxit=0)      eigdefpcg(A,b,x,M,W,spdim;maxit=0)     -> (x, it, res_norm[1:it], V[:,1:nvec])
 *     initcg(A,b,x,W;maxit=0)              initpcg(A,b,x,M,W;maxit=0)             -> (x, it, res_norm[1:it])
 * x, it, res_norm as for mi_cg & co. V_out receives n x nvec doubles (column-major, host or device pointer like x):
 * the approximate least-dominant eigenvectors of A (of M^{-1}A for the preconditioned kinds) to hand to
 * mi_defpcg / mi_eigdefpcg / mi_initpcg as W for the next system. For the deflated kinds nvec is the column count of W.
 * The iteration runs on the device; at each thick restart (every spdim - nev iterations) the spdim x spdim projected
 * matrix visits the host for the dense eigen/SVD step. V_out is determined up to the sign/rotation of eigenvectors.
 * spdim >= 2 nvec + 1 is required (MI_ERR_BOUNDS otherwise: the reference's `V[:, nev+1]` would be out of bounds);
 * MI_ERR_BOUNDS is also returned — after x, it, res_norm have been written — where the reference's final Ritz
 * extraction indexes out of bounds (a preconditioned solve that ends with exactly nvec search directions).
 * initpcg: the reference's body reads an unallocated `z` (initcg.jl:127, UndefVarError); implemented as documented
 * there (Init-PCG: deflated initial guess, then pcg). */
int mi_eigcg(mi_op_t A, const double *b, double *x, int64_t nvec, int64_t spdim, int64_t maxit, double eps,
             double *res_norm, int64_t res_cap, int64_t *it, double *V_out);
int mi_eigpcg(mi_op_t A, mi_op_t M, const double *b, double *x, int64_t nvec, int64_t spdim, int64_t maxit,
              double eps, double *res_norm, int64_t res_cap, int64_t *it, double *V_out);
int mi_eigdefcg(mi_op_t A, const double *b, double *x, const double *W, int64_t nvec, int64_t spdim, int64_t maxit,
                double eps, double *res_norm, int64_t res_cap, int64_t *it, double *V_out);
int mi_eigdefpcg(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec, int64_t spdim,
                 int64_t maxit, double eps, double *res_norm, int64_t res_cap, int64_t *it, double *V_out);
int mi_initcg(mi_op_t A, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit, double eps,
              double *res_norm, int64_t res_cap, int64_t *it);
int mi_initpcg(mi_op_t A, mi_op_t M, const double *b, double *x, const double *W, int64_t nvec, int64_t maxit,
               double eps, double *res_norm, int64_t res_cap, int64_t *it);

/* ---------------------------------------------------------------- timing on the context's stream */
int mi_event_create(mi_event_t *ev);
int mi_event_record(mi_ctx_t ctx, mi_event_t ev);
int mi_event_elapsed_ms(mi_event_t start, mi_event_t stop, double *ms); /* synchronises on stop */
int mi_event_destroy(mi_event_t ev);

#ifdef __cplusplus
}
#endif
#endif /* MI355SCHUR_H */
