#!/usr/bin/env python3
"""Example03_EllipticPdeDomainDecomposition.jl (lines 45-225) on the MI355X drop-in.

Same flow and prints as the reference script; only the operator construction goes through the
device library. Mesh/partition are the structured substitutes (no Triangle / METIS here), the
interior solves of the set-up are direct, the least-dominant eigenvectors come from a dense `eigh`
on the host instead of KrylovKit (Example03:209).

    python examples/example03_domain_decomposition.py [--N 100 --px 2 --py 2]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=100)      # tentative_nnode = N*N (Example03:16: 40_000)
    ap.add_argument("--px", type=int, default=2)
    ap.add_argument("--py", type=int, default=2)       # ndom = px*py (Example03:19: 20)
    args = ap.parse_args()
    pkg = graft.load_package()
    fem, api = pkg.fem, pkg.api
    a = lambda x, y: 1.0 + 0 * x                        # Example03:63-65
    f = lambda x, y: -1.0 + 0 * x                       # Example03:67-69
    uexact = lambda x, y: 0.734 + 0 * x                 # Example03:71-73

    t = time.time()
    P = fem.build_schur_problem(args.N, args.px, args.py, a, f, uexact)
    sub, mesh = P.sub, P.mesh
    n_Γ, ndom = sub.n_Γ, sub.ndom
    print(f"nnode = {mesh.points.shape[1]}\nnel = {mesh.cells.shape[1]}\nset-up {time.time() - t:.2f}s, n_Γ = {n_Γ}")

    ctx = api.Context(0)
    A, b = fem.do_isotropic_elliptic_assembly(mesh.cells, mesh.points, P.dinds, mesh.point_marker, a, f, uexact)
    A_IId, A_IΓd, A_ΓΓ, b_Id, b_Γ = fem.prepare_global_schur(mesh.cells, mesh.points, P.epart, sub, a, f, uexact)
    S_global = api.GlobalSchur(ctx, A_IId, A_IΓd, A_ΓΓ, P.solvers)                                     # Example03:101
    S_local_mat = api.LocalSchurs(ctx, P.Sd, sub.gather_idx, sub.node_Γ_cnt)                          # Example03:131-135
    ΠSnn_local_mat = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, sub.gather_idx, sub.node_Γ_cnt)  # :139-141
    S_local = api.MatrixFreeLocalSchurs(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, sub.gather_idx, sub.node_Γ_cnt, P.solvers)  # :143-150
    b_schur = P.b_schur

    d = S_global * b_schur - S_local_mat * b_schur                                                    # Example03:175
    print(f"extrema(S_global * b_schur - S_local_mat * b_schur) = ({d.min():.3e}, {d.max():.3e})")
    for name, op in (("S * b_schur (matrix-free)", S_global), ("S_local * b_schur (matrix-free)", S_local),
                     ("S_local_mat * b_schur", S_local_mat)):
        t = time.time(); op * b_schur; print(f"{name} ... {time.time() - t:.6f} seconds")            # :178-183

    u_Γ, it, _ = api.pcg(S_local_mat, b_schur, np.zeros(n_Γ), ΠSnn_local_mat)                         # Example03:193
    print(f"neumann-neumann-pcg: n = {S_local_mat.N}, iter = {it}")
    u_Id = fem.get_subdomain_solutions(u_Γ, A_IId, A_IΓd, b_Id, P.solvers)                            # :197
    u_with_dd = fem.merge_subdomain_solutions(u_Γ, u_Id, sub, P.dinds, uexact, mesh.points)           # :200
    u_no_dd = fem.append_bc(P.dinds, spla.spsolve(sp.csc_matrix(A), b), mesh.points, uexact)          # :111-117 (direct, no AMG)
    e = u_with_dd - u_no_dd
    print(f"extrema(u_with_dd - u_no_dd) = ({e.min():.3e}, {e.max():.3e})")                            # :204

    nev = ndom + 10                                                                                    # :206
    Sdense = np.column_stack([S_local_mat * col for col in np.eye(n_Γ)])
    lam, V = np.linalg.eigh((Sdense + Sdense.T) / 2)
    for tag, ϕ in (("ld", V[:, :nev]), ("md", V[:, -nev:])):
        u_Γ, it, _ = api.defpcg(S_local_mat, b_schur, np.zeros(n_Γ), np.asfortranarray(ϕ), ΠSnn_local_mat)  # :214, :224
        print(f"{tag}-def-neumann-neumann-pcg: n = {S_local_mat.N}, ndom = {ndom}, nev = {nev} ({tag}), iter = {it}")


if __name__ == "__main__":
    main()
