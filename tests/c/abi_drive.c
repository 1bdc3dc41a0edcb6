/* abi_drive.c — a plain-C caller of libmi355schur (no Python, no torch, no C++): the closest executable stand-in for the
 * `ccall`s of julia/MI355Schur.jl. Everything crosses the ABI the way a Julia caller passes it: Int64 index arrays that
 * are 1-BASED (index_base = 1), column-major dense blocks, host pointers, and an interior-solve callback that is a C
 * function invoked on the calling thread (a Julia @cfunction).
 *
 *   abi_drive problem.bin result.bin
 *
 * problem.bin (written by tests/test_gpu_boundary.py::test_plain_c_driver), all little-endian int64 / float64:
 *   ndom, n_gamma; node_gamma_cnt[n_gamma]; b_schur[n_gamma]; then per subdomain:
 *   n_gd, n_i; gather_idx[n_gd] (1-based); S_d[n_gd^2]; PiS_d[n_gd^2];
 *   A_IGdd: nnz, colptr[n_gd+1], rowval[nnz], nzval[nnz] (1-based CSC); A_GGdd likewise (colptr[n_gd+1]);
 *   inv(A_IIdd)[n_i^2] column-major (what the callback multiplies with).
 * result.bin: it_pcg, it_cg (as doubles); x_pcg[n]; res_norm[it_pcg]; S*b[n]; (M\b)[n]; S_matfree*b[n]; x_cg[n].
 *
 * Reference interfaces exercised: pcg / cg (RecyclingKrylovSolvers/cg.jl:14-18, 67-72), apply_local_schurs assembled
 * and matrix-free (EPDD.jl:761-785, 711-747), NeumannNeumannSchurPreconditioner `\` (EPDD.jl:1111-1137, 1389-1392). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355schur.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != MI_OK) {                                                               \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mi_last_error());                 \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

static void *xmalloc(size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
  return p;
}
static int64_t *read_i64(FILE *f, int64_t count) {
  int64_t *p = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)count);
  if (fread(p, sizeof(int64_t), (size_t)count, f) != (size_t)count) { fprintf(stderr, "short read\n"); exit(2); }
  return p;
}
static double *read_f64(FILE *f, int64_t count) {
  double *p = (double *)xmalloc(sizeof(double) * (size_t)count);
  if (fread(p, sizeof(double), (size_t)count, f) != (size_t)count) { fprintf(stderr, "short read\n"); exit(2); }
  return p;
}

/* interior solve: sol = inv(A_II[idom]) * rhs with the dense inverse the test handed over (column-major) */
typedef struct {
  int64_t ndom;
  double **inv;
  int64_t calls;
} interior_t;
static int interior_solve(void *user, int64_t idom, int64_t n, const double *rhs, double *sol) {
  interior_t *ctx = (interior_t *)user;
  if (idom < 0 || idom >= ctx->ndom) return 1;
  const double *Ainv = ctx->inv[idom];
  for (int64_t i = 0; i < n; ++i) sol[i] = 0.0;
  for (int64_t j = 0; j < n; ++j) {
    const double r = rhs[j];
    for (int64_t i = 0; i < n; ++i) sol[i] += Ainv[i + j * n] * r;
  }
  ctx->calls++;
  return 0;
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int64_t *hdr = read_i64(f, 2);
  const int64_t ndom = hdr[0], n = hdr[1];
  int64_t *cnt = read_i64(f, n);
  double *b = read_f64(f, n);
  int64_t *n_gd = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)ndom), *n_i = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)ndom);
  const int64_t **gather = (const int64_t **)xmalloc(sizeof(void *) * (size_t)ndom);
  const double **Sd = (const double **)xmalloc(sizeof(void *) * (size_t)ndom), **Pd = (const double **)xmalloc(sizeof(void *) * (size_t)ndom);
  const int64_t **igp = (const int64_t **)xmalloc(sizeof(void *) * (size_t)ndom), **igi = (const int64_t **)xmalloc(sizeof(void *) * (size_t)ndom);
  const int64_t **ggp = (const int64_t **)xmalloc(sizeof(void *) * (size_t)ndom), **ggi = (const int64_t **)xmalloc(sizeof(void *) * (size_t)ndom);
  const double **igv = (const double **)xmalloc(sizeof(void *) * (size_t)ndom), **ggv = (const double **)xmalloc(sizeof(void *) * (size_t)ndom);
  interior_t ictx = {ndom, (double **)xmalloc(sizeof(void *) * (size_t)ndom), 0};
  for (int64_t d = 0; d < ndom; ++d) {
    int64_t *sz = read_i64(f, 2);
    n_gd[d] = sz[0]; n_i[d] = sz[1];
    free(sz);
    gather[d] = read_i64(f, n_gd[d]);
    Sd[d] = read_f64(f, n_gd[d] * n_gd[d]);
    Pd[d] = read_f64(f, n_gd[d] * n_gd[d]);
    int64_t *nz = read_i64(f, 1);
    igp[d] = read_i64(f, n_gd[d] + 1); igi[d] = read_i64(f, nz[0]); igv[d] = read_f64(f, nz[0]);
    free(nz);
    nz = read_i64(f, 1);
    ggp[d] = read_i64(f, n_gd[d] + 1); ggi[d] = read_i64(f, nz[0]); ggv[d] = read_f64(f, nz[0]);
    free(nz);
    ictx.inv[d] = read_f64(f, n_i[d] * n_i[d]);
  }
  fclose(f);

  if (mi_version() < 100) { fprintf(stderr, "bad mi_version\n"); return 1; }
  mi_ctx_t ctx = NULL;
  CHECK(mi_ctx_create(0, &ctx));                       /* host pointer mode is the default, as for Julia arrays */
  mi_op_t S = NULL, M = NULL, Smf = NULL;
  CHECK(mi_schur_assembled_create(ctx, ndom, n, n_gd, gather, Sd, /*index_base=*/1, 0, ndom, &S));
  CHECK(mi_nn_create(ctx, ndom, n, n_gd, gather, Pd, cnt, /*index_base=*/1, 0, ndom, &M));
  CHECK(mi_schur_matfree_create(ctx, ndom, n, n_gd, n_i, gather, igp, igi, igv, ggp, ggi, ggv, interior_solve, &ictx,
                                /*index_base=*/1, 0, ndom, &Smf));
  int64_t sz = 0;
  CHECK(mi_op_size(S, &sz));
  if (sz != n) { fprintf(stderr, "mi_op_size: %lld != %lld\n", (long long)sz, (long long)n); return 1; }

  double *x = (double *)calloc((size_t)n, sizeof(double)), *res = (double *)xmalloc(sizeof(double) * (size_t)n);
  double *yS = (double *)xmalloc(sizeof(double) * (size_t)n), *yM = (double *)xmalloc(sizeof(double) * (size_t)n);
  double *ymf = (double *)xmalloc(sizeof(double) * (size_t)n), *xcg = (double *)calloc((size_t)n, sizeof(double));
  double *rescg = (double *)xmalloc(sizeof(double) * (size_t)n);
  int64_t it = 0, itcg = 0;
  CHECK(mi_pcg(S, M, b, x, /*maxit=*/0, /*eps=*/0.0, res, n, &it));      /* pcg(S, b_schur, zeros(n), Πnn) */
  CHECK(mi_op_apply(S, b, yS));                                           /* S * b   (mul!)                 */
  CHECK(mi_op_apply(M, b, yM));                                           /* Πnn \ b (ldiv!)                */
  CHECK(mi_op_apply(Smf, b, ymf));                                        /* matrix-free S * b, C callback  */
  if (ictx.calls != ndom) { fprintf(stderr, "callback ran %lld times, expected %lld\n", (long long)ictx.calls, (long long)ndom); return 1; }
  CHECK(mi_cg(S, b, xcg, 0, 0.0, rescg, n, &itcg));                       /* cg(S, b_schur, zeros(n))       */

  /* error convention: a NULL handle is a status code and a message, never a crash */
  if (mi_op_apply(NULL, b, yS) != MI_ERR_BAD_ARG || !mi_last_error()[0]) { fprintf(stderr, "error convention broken\n"); return 1; }
  /* res_norm capacity smaller than it: MI_ERR_RES_CAPACITY (Julia: BoundsError), `it` still reported */
  {
    double *x2 = (double *)calloc((size_t)n, sizeof(double));
    int64_t it2 = 0;
    const int rc = mi_pcg(S, M, b, x2, 0, 0.0, res + 0, 0, &it2);
    if (rc != MI_ERR_BAD_ARG && rc != MI_ERR_RES_CAPACITY) { fprintf(stderr, "res_cap = 0: rc %d\n", rc); return 1; }
    free(x2);
  }

  FILE *g = fopen(argv[2], "wb");
  if (!g) { perror(argv[2]); return 2; }
  const double head[2] = {(double)it, (double)itcg};
  fwrite(head, sizeof(double), 2, g);
  fwrite(x, sizeof(double), (size_t)n, g);
  fwrite(res, sizeof(double), (size_t)it, g);
  fwrite(yS, sizeof(double), (size_t)n, g);
  fwrite(yM, sizeof(double), (size_t)n, g);
  fwrite(ymf, sizeof(double), (size_t)n, g);
  fwrite(xcg, sizeof(double), (size_t)n, g);
  fclose(g);

  CHECK(mi_op_destroy(Smf));
  CHECK(mi_op_destroy(M));
  CHECK(mi_op_destroy(S));
  CHECK(mi_ctx_destroy(ctx));
  printf("abi_drive ok: n_gamma=%lld ndom=%lld pcg it=%lld cg it=%lld callback calls=%lld\n", (long long)n, (long long)ndom,
         (long long)it, (long long)itcg, (long long)ictx.calls);
  return 0;
}
