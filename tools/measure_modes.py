#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the headline): config 3 with host pointers (PCIe-inclusive rate),
config 3 in matrix-free mode (sparse products on the GPU, interior solves through the host callback), and
deflated PCG on the assembled operator."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
fem, api = pkg.fem, pkg.api
import torch  # noqa: E402

N = int(os.environ.get("MEASURE_N", "1000"))
mesh = fem.get_mesh(N)
_, g = fem.draw(fem.synthetic_kl(mesh.points), np.random.default_rng(481456))
P = fem.build_schur_problem(N, 4, 2, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
ctx = api.Context(0)
n, b = P.sub.n_Γ, P.b_schur
S = api.LocalSchurs(ctx, P.Sd, P.sub.gather_idx, P.sub.node_Γ_cnt)
M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)


def rate(fn, reps):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        it = fn()
    dt = (time.perf_counter() - t0) / reps
    return it, dt


bd = torch.from_numpy(b).cuda()
it, dt = rate(lambda: api.pcg(S, bd, torch.zeros_like(bd), M)[1], 200)
print(f"pcg device pointers : it={it} {dt * 1e6:8.1f} us/solve {(it - 1) / dt:9.0f} it/s")
it, dt = rate(lambda: api.pcg(S, b, np.zeros(n), M)[1], 200)
print(f"pcg host pointers   : it={it} {dt * 1e6:8.1f} us/solve {(it - 1) / dt:9.0f} it/s   (H2D b,x + D2H x per solve)")
t0 = time.perf_counter()
for _ in range(300):
    yh = S * b
t1 = time.perf_counter()
for _ in range(300):
    yd = S * bd
ctx.synchronize(); t2 = time.perf_counter()
print(f"S-apply, host pointers (mul! from Julia arrays): {(t1 - t0) / 300 * 1e6:6.1f} us   device pointers: {(t2 - t1) / 300 * 1e6:6.1f} us")
it, dt = rate(lambda: api.cg(S, bd, torch.zeros_like(bd))[1], 20)
print(f"cg (no precond)     : it={it} {dt * 1e6:8.1f} us/solve {(it - 1) / dt:9.0f} it/s")
Sd = np.zeros((n, n))
for d in range(P.sub.ndom):
    gi = P.sub.gather_idx[d]
    Sd[np.ix_(gi, gi)] += P.Sd[d]
W = np.asfortranarray(np.linalg.eigh(Sd)[1][:, :P.sub.ndom + 10])
Wd = torch.from_numpy(np.ascontiguousarray(W.T)).cuda().T
it, dt = rate(lambda: api.defpcg(S, bd, torch.zeros_like(bd), Wd, M)[1], 50)
print(f"defpcg nvec={W.shape[1]}      : it={it} {dt * 1e6:8.1f} us/solve {(it - 1) / dt:9.0f} it/s")
nvec, spdim = int(1.25 * P.sub.ndom), 3 * P.sub.ndom            # Example09:39-40
it, dt = rate(lambda: api.eigpcg(S, bd, torch.zeros_like(bd), M, nvec, spdim)[1], 50)
print(f"eigpcg nvec={nvec} spdim={spdim}: it={it} {dt * 1e6:8.1f} us/solve {(it - 1) / dt:9.0f} it/s")
We = api.eigpcg(S, bd, torch.zeros_like(bd), M, nvec, spdim)[3]
it, dt = rate(lambda: api.eigdefpcg(S, bd, torch.zeros_like(bd), M, We, spdim)[1], 50)
print(f"eigdefpcg nvec={nvec}       : it={it} {dt * 1e6:8.1f} us/solve {(it - 1) / dt:9.0f} it/s")
it, dt = rate(lambda: api.eigcg(S, bd, torch.zeros_like(bd), nvec, spdim)[1], 10)
print(f"eigcg (no precond)  : it={it} {dt * 1e6:8.1f} us/solve {(it - 1) / dt:9.0f} it/s")
# f3: numeric assembly of all blocks and right-hand sides on the device vs the host element loop
plan = fem.make_assembly_plan(P.mesh.cells, P.mesh.points, P.epart, P.sub, lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x)
dev = api.AssemblyPlan(ctx, plan)
ad = torch.from_numpy(np.exp(g)).cuda()
dev.run(ad); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    v = dev.run(ad)
ctx.synchronize(); t1 = time.perf_counter()
nb = 4 * plan.ccode.size + 8 * plan.n_entries * 2 + 8 * plan.ccode.size
print(f"device assembly     : {plan.n_entries} entries, {plan.ccode.size} contributions: {(t1 - t0) / 20 * 1e3:7.3f} ms/realization "
      f"(~{nb / ((t1 - t0) / 20) / 1e9:6.0f} GB/s of code+G+out streams)")
t0 = time.perf_counter(); fem.prepare_local_schurs(P.mesh.cells, P.mesh.points, P.epart, P.sub, np.exp(g), lambda x, y: -1.0 + 0 * x, lambda x, y: 0.734 + 0 * x); t1 = time.perf_counter()
print(f"host element loop   : {(t1 - t0) * 1e3:7.1f} ms/realization (numpy)")
if os.environ.get("MEASURE_SKIP_MATFREE"):
    sys.exit(0)
Sm = api.MatrixFreeLocalSchurs(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, P.sub.gather_idx, P.sub.node_Γ_cnt, P.solvers)
Sm * b
t0 = time.perf_counter(); Sm * b; t1 = time.perf_counter()
print(f"matrix-free S-apply : {(t1 - t0) * 1e3:8.1f} ms (8 host interior solves of ~124k unknowns + PCIe + 3 SpMV launches)")
t0 = time.perf_counter(); x, it, res = api.pcg(Sm, b, np.zeros(n), M); t1 = time.perf_counter()
print(f"matrix-free pcg     : it={it} {(t1 - t0):8.2f} s/solve {(it - 1) / (t1 - t0):9.2f} it/s")
Sg = api.MatrixFreeLocalSchurs(ctx, P.A_IIdd, P.A_IΓdd, P.A_ΓΓdd, P.sub.gather_idx, P.sub.node_Γ_cnt, None, reltol=1e-9)
Sg * b
t0 = time.perf_counter(); yg = Sg * b; t1 = time.perf_counter()
ya = S * b
print(f"matrix-free S-apply, interior CG on the device (reltol 1e-9): {(t1 - t0) * 1e3:8.1f} ms; "
      f"|Δ| vs assembled = {np.abs(yg - ya).max() / np.abs(ya).max():.2e}")
_, nb = Sg.bytes()
us = Sg.time_dominant(bd, 100)
print(f"A_II SpMV (block-diagonal, {nb / 1e6:.1f} MB algorithmic): {us:7.2f} us/launch {nb / us / 1e3:8.1f} GB/s")
t0 = time.perf_counter(); x, it, res = api.pcg(Sg, b, np.zeros(n), M); t1 = time.perf_counter()
print(f"matrix-free pcg, device interior CG: it={it} {(t1 - t0):8.2f} s/solve {(it - 1) / (t1 - t0):9.2f} it/s")
