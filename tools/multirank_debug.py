#!/usr/bin/env python3
"""In-process ranks, small problem: prints what every rank sees (debug aid for the peer exchange)."""
import os
import sys
import threading
import time
import traceback

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "1048576")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
api, fem = pkg.api, pkg.fem
from conftest import f_m1, lognormal_coeff, u0734  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 90
cases = [(int(a.split(",")[0]), int(a.split(",")[1])) for a in sys.argv[2:]] or [(2, 1)]
mesh = fem.get_mesh(N)
P = fem.build_schur_problem(N, 4, 2, lognormal_coeff(fem, mesh.points, 5), f_m1, u0734)
ndom, n, b = P.sub.ndom, P.sub.n_Γ, P.b_schur
gi, cnt = P.sub.gather_idx, P.sub.node_Γ_cnt
world, shard_nn, group, keep, out = 0, 0, None, None, None


def main(r):
    try:
        ctx = api.Context(0)
        ctx.loopback_init(group, r)
        print(r, "peer", ctx.query("peer_exchange"), "no_graph", ctx.query("no_graph"), flush=True)
        lo, hi = api.shard_domains(ndom, r, world)
        S = api.LocalSchurs(ctx, [P.Sd[d] if lo <= d < hi else None for d in range(ndom)], gi, cnt, dom_slice=(lo, hi))
        if shard_nn:
            M = api.NeumannNeumannSchurPreconditioner(ctx, [P.ΠSd[d] if lo <= d < hi else None for d in range(ndom)], gi, cnt, dom_slice=(lo, hi))
        else:
            M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, gi, cnt, dom_slice=(0, ndom))
        keep[r] = (ctx, S, M)
        print(r, "operators built t=%.2f" % (time.time() - T0), flush=True)
        BAR.wait(timeout=300)
        y = S * b
        print(r, "apply done", float(np.abs(y).max()), "t=%.2f" % (time.time() - T0), flush=True)
        res = api.pcg(S, b, np.zeros(n), M)
        print(r, "t=%.2f" % (time.time() - T0), "pcg it", res[1], "exchanges", ctx.query("exchanges"), "replays", ctx.query("graph_replays"), flush=True)
        out[r] = (y, res)
    except Exception:
        print(r, "FAILED t=%.2f" % (time.time() - T0), flush=True)
        traceback.print_exc()


T0 = time.time()
for world, shard_nn in cases:
    print("==== world", world, "shard_nn", shard_nn, flush=True)
    group = api.LoopbackGroup(world)
    BAR = threading.Barrier(world)
    keep, out = [None] * world, [None] * world
    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    keep = None
ctx1 = api.Context(0)
S1 = api.LocalSchurs(ctx1, P.Sd, gi, cnt)
M1 = api.NeumannNeumannSchurPreconditioner(ctx1, P.ΠSd, gi, cnt)
ref = api.pcg(S1, b, np.zeros(n), M1)
print("single it", ref[1])
if all(o is not None for o in out):
    for r in range(world):
        print(r, "bit-identical to rank 0:", np.array_equal(out[r][1][0], out[0][1][0]), "it", out[r][1][1],
              "res rel diff vs single", float(np.max(np.abs(out[r][1][2] - ref[2][:len(out[r][1][2])]) / ref[2][:len(out[r][1][2])])) if out[r][1][1] == ref[1] else "it differs")
