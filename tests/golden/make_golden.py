"""Regenerates tests/golden/{micro,toy}.npz from the in-repo set-up code and the CPU oracle.

The reference holds no golden vectors, stored outputs or tests for this path and cannot be run in
this container (no julia), so these fixtures are produced by this repo's own fp64 CPU restatement
(oracle/krylov_oracle.c): they pin the oracle against regressions and give the GPU tests fixed
inputs/outputs that do not depend on SuperLU's pivoting on the GPU box. PARITY UNPINNED vs Julia.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402
from conftest import f_m1, lowest_eigvecs, one, u0734  # noqa: E402
from oracle import oracle as orc  # noqa: E402

fem = graft.load_package().fem


def make(name, N):
    P = fem.build_schur_problem(N, 2, 2, one, f_m1, u0734)
    n = P.sub.n_Γ
    out = dict(n_gamma=n, N=N, node_gamma_cnt=P.sub.node_Γ_cnt, b_schur=P.b_schur)
    for d in range(P.sub.ndom):
        out[f"gather_idx_{d}"] = P.sub.gather_idx[d]
        out[f"Sd_{d}"] = P.Sd[d]
        out[f"PiSd_{d}"] = P.ΠSd[d]
    S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    M = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    W = lowest_eigvecs(S, n, P.sub.ndom + 10)          # Example03:206 nev = ndom + 10
    out["W"] = W
    x0 = np.zeros(n)
    for tag, (x, it, res) in (("cg", orc.cg(S, P.b_schur, x0)), ("pcg", orc.pcg(S, P.b_schur, x0, M)),
                              ("defpcg", orc.defpcg(S, P.b_schur, x0, W, M))):
        out[f"{tag}_x"], out[f"{tag}_it"], out[f"{tag}_res_norm"] = x, it, res
        print(name, tag, "it =", it)
    rng = np.random.default_rng(12345)
    v = rng.standard_normal(n)
    out["v"], out["S_v"], out["M_v"] = v, S(v), M(v)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)


if __name__ == "__main__":
    make("micro", 40)
    make("toy", 100)
