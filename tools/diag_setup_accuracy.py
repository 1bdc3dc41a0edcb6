"""How accurate are the device's S_d (mi_schur_setup_run) at config 3's size?  For a few subdomains: asymmetry, and the
deviation from S_d = A_ΓΓ - A_IΓ' (A_II \\ A_IΓ) computed with scipy's sparse LU (no oracle involved)."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package(); api, fem = pkg.api, pkg.fem
import torch
N = int(os.environ.get("DIAG_N", "1000")); px, py = 4, 2
seed = 481456
mesh = fem.get_mesh(N)
kl = fem.synthetic_kl(mesh.points)
_, gfield = fem.draw(kl, np.random.default_rng(seed))
P = fem.build_schur_problem(N, px, py, np.exp(gfield), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x, assemble=False, precond=False)
ctx = api.Context(0)
setup = api.SchurSetup(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd)
vals = [torch.from_numpy(v).cuda() for v in setup._vals]
Sd, _ = setup.run(*vals, None); ctx.synchronize()
blocks = [b.cpu().numpy() for b in setup.blocks(Sd)]
for d in (0, 1):
    S = blocks[d]
    asym = np.abs(S - S.T).max() / np.abs(S).max()
    t0 = time.time()
    lu = spla.splu(sp.csc_matrix(P.A_IIdd[d]))
    X = lu.solve(P.A_IΓdd[d].toarray())
    Sref = P.A_ΓΓdd[d].toarray() - P.A_IΓdd[d].T @ X
    err = np.abs(S - Sref).max() / np.abs(Sref).max()
    ones = np.ones(S.shape[0])
    print(f"subdomain {d}: n_Γd={S.shape[0]} n_I={P.A_IIdd[d].shape[0]} asym={asym:.2e} |S-Sref|max/|Sref|max={err:.2e} "
          f"|S 1|/|S|={np.abs(S @ ones).max() / np.abs(S).max():.2e} ref: {np.abs(Sref @ ones).max() / np.abs(Sref).max():.2e} "
          f"min eig dev {np.linalg.eigvalsh((S + S.T) / 2)[:2]} ref {np.linalg.eigvalsh((Sref + Sref.T) / 2)[:2]}  ({time.time() - t0:.0f}s)", flush=True)
