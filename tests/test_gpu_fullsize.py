"""GPU suite at BASELINE.json's full size (config 3: N=1000, 996 004 free DoF, 4x2 subdomains, lognormal
coefficient): direct comparison with the oracle (which finishes a 1M-DoF Schur solve in ~0.2 s) plus
size-independent properties of the operators (symmetry, linearity, Example03:175 identity)."""
import numpy as np
import pytest

from conftest import f_m1, lognormal_coeff, u0734

pytestmark = pytest.mark.gpu


def test_full_size_pcg_matches_oracle(pkg, ctx, orc, full):
    P, api = full, pkg.api
    n, b = P.sub.n_Γ, P.b_schur
    assert n == 3989 and sorted(P.sub.n_Γd) == [747, 748, 748, 749, 1247, 1247, 1249, 1249]
    S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    So = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    Mo = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    x, it, res = api.pcg(S, b, np.zeros(n), M)
    xo, ito, reso = orc.pcg(So, b, np.zeros(n), Mo)
    assert it == ito
    assert np.allclose(res, reso, rtol=1e-8, atol=1e-12 * reso[0])
    assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)
    assert np.linalg.norm(b - So * x) <= 1.01e-7 * np.linalg.norm(b) * 1.5      # true residual, oracle operator
    # host pointers (Julia arrays) and device pointers give the same bits
    import torch
    xt = torch.zeros(n, dtype=torch.float64, device="cuda")
    _, it2, res2 = api.pcg(S, torch.from_numpy(b).cuda(), xt, M)
    assert it2 == it and np.array_equal(res2, res) and np.array_equal(xt.cpu().numpy(), x)
    # operators: agreement with the oracle, symmetry, linearity
    rng = np.random.default_rng(0)
    v, w = rng.standard_normal(n), rng.standard_normal(n)
    Sv, Sw = S * v, S * w
    assert np.allclose(Sv, So * v, rtol=0, atol=1e-13 * np.abs(Sv).max())
    assert np.allclose(M.ldiv(v), Mo * v, rtol=0, atol=1e-13 * np.abs(Mo * v).max())
    assert abs(w @ Sv - v @ Sw) <= 1e-11 * abs(w @ Sv)
    assert np.allclose(S * (2.0 * v - 3.0 * w), 2.0 * Sv - 3.0 * Sw, rtol=0, atol=1e-12 * np.abs(Sv).max())
    assert v @ Sv > 0 and v @ M.ldiv(v) > 0


def test_full_size_matrix_free_equals_assembled(pkg, ctx, full):
    """Example03:175 at 1M DoF: the matrix-free operator (sparse products on the GPU, interior solves through the
    host callback) and the assembled operator agree to the accuracy of the direct interior solve."""
    P, api = full, pkg.api
    S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    Sm = api.MatrixFreeLocalSchurs(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, P.sub.gather_idx, P.sub.node_Γ_cnt, P.solvers)
    a, m = S * P.b_schur, Sm * P.b_schur
    assert np.allclose(a, m, rtol=0, atol=1e-9 * np.abs(a).max())
