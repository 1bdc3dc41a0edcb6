"""One rank of the cross-PROCESS peer-exchange test (tests/test_gpu_multirank.py): every rank is a process of its own on
the same GPU, the arenas are mapped through `hipIpcGetMemHandle` / `hipIpcOpenMemHandle`, the handles travel over
torch.distributed (gloo on the loopback interface) — the hand-shake bench.py uses across the GPUs of a node.
    python peer_ipc_worker.py RANK WORLD PORT OUT.npz"""
import os
import sys

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
from conftest import f_m1, lognormal_coeff, u0734  # noqa: E402

dist.init_process_group("gloo")
pkg = graft.load_package()
api, fem = pkg.api, pkg.fem
N, px, py = 90, 4, 2
mesh = fem.get_mesh(N)
P = fem.build_schur_problem(N, px, py, lognormal_coeff(fem, mesh.points, 5), f_m1, u0734)
ndom, n, b = P.sub.ndom, P.sub.n_Γ, P.b_schur
gi, cnt = P.sub.gather_idx, P.sub.node_Γ_cnt


def all_gather(obj):
    lst = [None] * world
    dist.all_gather_object(lst, obj)
    return lst


ctx = api.Context(0)
ctx.peer_connect(rank, world, all_gather)
lo, hi = api.shard_domains(ndom, rank, world)
S = api.LocalSchurs(ctx, [P.Sd[d] if lo <= d < hi else None for d in range(ndom)], gi, cnt, dom_slice=(lo, hi))
M = api.NeumannNeumannSchurPreconditioner(ctx, [P.ΠSd[d] if lo <= d < hi else None for d in range(ndom)], gi, cnt, dom_slice=(lo, hi))
dist.barrier()
y = S * b
x, it, res = api.pcg(S, b, np.zeros(n), M)
v = ctx.allreduce_sum(np.full(7, float(rank + 1)))
np.savez(out, y=y, x=x, it=it, res=res, v=v, peer=ctx.query("peer_exchange"), replays=ctx.query("graph_replays"),
         exchanges=ctx.query("exchanges"))
dist.barrier()
dist.destroy_process_group()
