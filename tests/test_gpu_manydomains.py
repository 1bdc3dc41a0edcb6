"""GPU suite at the reference's own partition sizes: the domain-decomposition drivers of the reference run with 80-500
subdomains (Examples/KarhunenLoeveDomainDecompositionHelper.jl:14-32, Example03's ndom = 400), where the interface system
is too large for the single-workgroup vector kernels (n_Γ > 8192): `pcg`, `defpcg`, `cg` and the recycling pair then run the
generic multi-workgroup loop (dense-block GEMVs + deterministic two-stage reductions). 160 subdomains (16 x 10 boxes) of a
N = 400 mesh: n_Γ = 9417, blocks of up to 130 interface nodes, every solve against the oracle on the same inputs."""
import numpy as np
import pytest

from conftest import f_m1, u0734
from test_gpu_parity import assert_history, gpu_ops, orc_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def many(fem):
    N, px, py = 400, 16, 10
    mesh = fem.get_mesh(N)
    g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))[1]
    return fem.build_schur_problem(N, px, py, np.exp(g), f_m1, u0734, mesh=mesh)


def test_many_subdomain_schur_pcg_generic_loop(pkg, ctx, orc, many, monkeypatch):
    api = pkg.api
    P = many
    n, b = P.sub.n_Γ, P.b_schur
    assert len(P.Sd) == 160 and n > 8192                     # beyond FUSED_MAX_N: multi-workgroup loop kernels
    S, M = gpu_ops(pkg, ctx, P)
    So, Mo = orc_ops(orc, P)
    prev_threads = orc.set_threads(0)    # (0: query only)
    orc.set_threads(2)                   # blocks of 130 rows: the oracle's row-parallel GEMV is all fork/join beyond a few threads
    v = np.random.default_rng(3).standard_normal(n)
    ys, yo = S * v, So * v
    assert np.abs(ys - yo).max() <= 1e-13 * np.abs(yo).max()
    zs, zo = M.ldiv(v), Mo(v)
    assert np.abs(zs - zo).max() <= 1e-12 * np.abs(zo).max()
    # pcg: the folded 2-launch loop also beyond FUSED_MAX_N (multi-workgroup start-up, then the same launches as at config 3)
    want = orc.pcg(So, b, np.zeros(n), Mo)
    got = api.pcg(S, b, np.zeros(n), M)
    assert_history(got, want, So, b)
    assert np.array_equal(api.pcg(S, b, np.zeros(n), M)[0], got[0])            # replays are deterministic
    x1 = np.random.default_rng(8).standard_normal(n)
    assert_history(api.pcg(S, b, x1.copy(), M), orc.pcg(So, b, x1.copy(), Mo), So, b)   # non-zero initial guess
    # ... and the generic multi-workgroup loop it replaces there (8 launches per iteration), same bar
    monkeypatch.setenv("MI355_NO_BIG_FOLD", "1")
    gen = api.pcg(S, b, np.zeros(n), M)
    monkeypatch.delenv("MI355_NO_BIG_FOLD")
    assert_history(gen, want, So, b)
    assert abs(gen[1] - got[1]) <= 2
    # deflation with the ndom + 10 least dominant eigenvectors is Example03's set-up (Example03:206-225); any full-rank W
    # exercises the same kernels: nvec = 24 (one-wave LU solve) and nvec = 170 (generic projection kernels)
    # (the oracle's long solves dominate this test's run time on the GPU box: the second deflated solve and the recycling
    # pair are cut off by maxit, which compares the same kernels over fewer iterations)
    Q = np.linalg.qr(np.random.default_rng(4).standard_normal((n, 170)))[0]
    W = np.asfortranarray(Q[:, :24])
    assert_history(api.defpcg(S, b, np.zeros(n), W, M), orc.defpcg(So, b, np.zeros(n), W, Mo), So, b)
    W = np.asfortranarray(Q)
    gd, od = api.defpcg(S, b, np.zeros(n), W, M, maxit=40), orc.defpcg(So, b, np.zeros(n), W, Mo, maxit=40)
    assert gd[1] == od[1] == 40 and np.allclose(gd[2][:12], od[2][:12], rtol=1e-8)
    assert (gd[2] / od[2]).max() < 2.0 and (gd[2] / od[2]).min() > 0.5
    # unpreconditioned and cut off at maxit = 60, far from convergence: `it` and the history are compared (tight over the
    # first entries, then to the factor the long-solve rule allows); the unconverged iterate itself is as sensitive to the
    # summation order as the history's tail and is held to the same kind of bar, not to the converged-solve one
    xg, itg, rg = api.cg(S, b, np.zeros(n), maxit=60)
    xo_, ito_, ro_ = orc.cg(So, b, np.zeros(n), maxit=60)
    assert itg == ito_ == 60
    assert np.allclose(rg[:12], ro_[:12], rtol=1e-8)
    assert (rg / ro_).max() < 2.0 and (rg / ro_).min() > 0.5
    assert np.linalg.norm(xg - xo_) <= 1e-2 * np.linalg.norm(xo_)
    # recycling pair (Example09_..._Functions.jl:345, 364)
    nv, spdim = 20, 60
    x1, it1, r1, W1 = api.eigpcg(S, b, np.zeros(n), M, nv, spdim, maxit=90)          # two thick restarts
    xo1, ito1, ro1, Wo1 = orc.eigpcg(So, b, np.zeros(n), Mo, nv, spdim, maxit=90)
    assert it1 == ito1 == 90 and np.allclose(r1[:12], ro1[:12], rtol=1e-8)
    assert (r1 / ro1).max() < 2.0 and (r1 / ro1).min() > 0.5
    x2, it2, r2, _ = api.eigdefpcg(S, b, np.zeros(n), M, Wo1, spdim)                  # converges: the recycled space deflates
    xo2, ito2, ro2, _ = orc.eigdefpcg(So, b, np.zeros(n), Mo, Wo1, spdim)
    assert it2 < got[1] and ito2 < got[1]
    assert_history((x2, it2, r2), (xo2, ito2, ro2), So, b)
    # torch device vectors through the same path
    import torch
    xt, itt, rest = api.pcg(S, torch.from_numpy(b).cuda(), torch.zeros(n, dtype=torch.float64, device="cuda"), M)
    assert itt == got[1] and np.array_equal(xt.cpu().numpy(), got[0])
    orc.set_threads(prev_threads)


def test_pinv_of_floating_subdomains_without_an_eigensolver(pkg, ctx, many):
    """`prepare_neumann_neumann_schur_precond` (EPDD.jl:1201-1220) at the reference's partition sizes: of the 160 subdomains
    112 are floating (no Dirichlet node: S_d 1 = 0, rank n - 1). `mi_nn_pinv` takes none of them to the eigen-decomposition
    (rocSOLVER dsyevd): boundary blocks are inverted, floating ones go through S^+ = (S + α u u')^{-1} - u u'/α. Against
    numpy's SVD pinv with the reference's rtol = sqrt(eps)."""
    api = pkg.api
    P = many
    nd = P.sub.n_Γd
    Sd = np.concatenate([np.asarray(S, order="F").ravel(order="F") for S in P.Sd])
    before = ctx.query("spectral_pinv")
    Pi = api.nn_pinv(ctx, nd, Sd)
    assert ctx.query("spectral_pinv") == before, "a block went to the eigen-decomposition"
    rtol = float(np.sqrt(np.finfo(float).eps))
    off, n_floating, worst = 0, 0, 0.0
    for d, n in enumerate(nd):
        got = Pi[off:off + n * n].reshape(n, n).T
        off += n * n
        S = np.asarray(P.Sd[d])
        floating = np.abs(S.sum(axis=1)).max() <= rtol * np.abs(S).sum(axis=1).max()
        n_floating += int(floating)
        if d % 7 and not (floating and d % 3 == 0):
            continue                                                   # a sample keeps the host SVDs short
        ref = np.linalg.pinv(S, rcond=rtol)
        worst = max(worst, np.abs(got - ref).max() / np.abs(ref).max())
        if floating:
            assert np.abs(got.sum(axis=1)).max() <= 1e-6 * np.abs(got).max()    # the constants are in the kernel of S^+ too
    assert n_floating >= 100
    assert worst <= 1e-7, worst
