"""CPU suite: the N>1 path with world_size 2 over gloo.

What shards: subdomains -> ranks (`api.shard_domains`), Γ-vectors replicated, the two
Γ-sums of every PCG iteration all-reduced. On a GPU box the per-rank partial applies are the HIP
kernels and the all-reduce is RCCL inside the iteration graph (tests/test_gpu_parity.py covers the
captured collective at world_size 1 and — with 2, 3, 4 and 8 in-process ranks joined by the loopback
communicator on one GPU — the sharded operators and loops themselves). Here, without a GPU, the per-rank
partial applies are played by the oracle (as the checker) and the all-reduce by gloo, which pins the
host-side logic: the shard plan, the per-rank problem set-up (`dom_slice`), the b_schur reduction and
the fact that the replicated-vector algorithm reproduces the single-rank iterates.
"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, f_m1, one, u0734


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _allreduce(v):
    t = torch.from_numpy(v.copy())
    dist.all_reduce(t)
    return t.numpy()


def _worker(rank, world, port, N, px, py, out, replicate_precond):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    from oracle import oracle as orc
    pkg = graft.load_package()
    fem, shard = pkg.fem, pkg.api.shard_domains
    lo, hi = shard(px * py, rank, world)
    P = fem.build_schur_problem(N, px, py, one, f_m1, u0734, dom_slice=(lo, hi))
    assert all((P.Sd[d] is not None) == (lo <= d < hi) for d in range(px * py))
    n = P.sub.n_Γ
    b = _allreduce(P.b_schur)                                   # b_schur = Σ_ranks
    loc = list(range(lo, hi))
    gi = [P.sub.gather_idx[d] for d in loc]
    S_part = orc.apply_local_schurs_operator([P.Sd[d] for d in loc], gi, n)
    S = lambda v: _allreduce(S_part(v))                         # partial apply + all-reduce
    if replicate_precond:
        # bench.py's default at N>1: the NN blocks are broadcast to every rank, the NN-apply is local (one
        # all-reduce per iteration instead of two)
        Pi_all = []
        for d in range(px * py):
            owner = next(r for r in range(world) if shard(px * py, r, world)[0] <= d < shard(px * py, r, world)[1])
            nd = P.sub.n_Γd[d]
            t = torch.from_numpy(np.ascontiguousarray(P.ΠSd[d])) if rank == owner else torch.empty((nd, nd), dtype=torch.float64)
            dist.broadcast(t, owner)
            Pi_all.append(np.asfortranarray(t.numpy()))
        M_full = orc.neumann_neumann_operator(Pi_all, P.sub.gather_idx, P.sub.node_Γ_cnt)
        M = lambda v: M_full(v)
    else:
        M_part = orc.neumann_neumann_operator([P.ΠSd[d] for d in loc], gi, P.sub.node_Γ_cnt)
        M = lambda v: _allreduce(M_part(v))
    # pcg (cg.jl:67-109) on replicated vectors; every rank takes the same branches
    x = np.zeros(n)
    r = b - S(x)
    rTr = r @ r
    z = M(r)
    rTz = r @ z
    p = z.copy()
    res = [np.sqrt(rTr)]
    tol = 1e-7 * np.sqrt(b @ b)
    it = 1
    while it < n and res[-1] > tol:
        Ap = S(p)
        alpha = rTz / (p @ Ap)
        beta = 1.0 / rTz
        x += alpha * p
        r -= alpha * Ap
        rTr = r @ r
        z = M(r)
        rTz = r @ z
        beta *= rTz
        p = beta * p + z
        it += 1
        res.append(np.sqrt(rTr))
    np.savez(out.format(rank=rank), x=x, it=it, res=np.array(res), b=b)
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("replicate_precond", [False, True])
def test_two_rank_sharded_pcg_matches_single_rank(tmp_path, fem, orc, replicate_precond):
    N, px, py, world = 40, 2, 2, 2
    out = str(tmp_path / "rank{rank}.npz")
    mp.spawn(_worker, args=(world, _free_port(), N, px, py, out, replicate_precond), nprocs=world, join=True)
    P = fem.build_schur_problem(N, px, py, one, f_m1, u0734)
    n = P.sub.n_Γ
    S = orc.apply_local_schurs_operator(P.Sd, P.sub.gather_idx, n)
    M = orc.neumann_neumann_operator(P.ΠSd, P.sub.gather_idx, P.sub.node_Γ_cnt)
    x, it, res = orc.pcg(S, P.b_schur, np.zeros(n), M)
    r0, r1 = np.load(out.format(rank=0)), np.load(out.format(rank=1))
    assert np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["res"], r1["res"])   # ranks stay bit-identical
    assert np.allclose(r0["b"], P.b_schur, rtol=1e-12, atol=1e-14)
    assert int(r0["it"]) == it
    assert np.allclose(r0["res"], res, rtol=1e-8, atol=1e-12 * res[0])
    assert np.linalg.norm(r0["x"] - x) <= 1e-6 * np.linalg.norm(x)


def test_bench_launcher_spawns_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts the ranks itself (fresh children, before
    any GPU call, rendezvous on 127.0.0.1). Checked on the CPU with --launcher-selftest: the children do a gloo
    all-reduce instead of the benchmark and rank 0 prints the one JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for n in (2, 3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launcher-selftest"],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out == {"launcher": "ok", "n_gpus": n, "sum": n * (n + 1) / 2, "master": "127.0.0.1"}
    # under a launcher (WORLD_SIZE set) a mismatch with --gpus is still refused
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
