"""BASELINE.json configs[3] (1 M DoF, 8 subdomains = one per GPU, Γ-sum across the ranks) at its own size on ONE GPU:
`world` contexts of this process ("in-process ranks", one host thread and one stream each) joined by the peer exchange
(csrc/exchange.hpp: device-side flags, no host in the loop) — the reference's `@distributed (+) for idom`
(Fem/EllipticPdePllDomainDecomposition.jl:10-14) as the production path runs it: graph-captured iterations, S sharded with
the Neumann-Neumann blocks replicated (one exchange per iteration) or sharded too (two). Checked against the C oracle
(`it` equal, histories to the bar of DESIGN §3) and rank against rank (bit-identical).
What this cannot show: timing and ordering of the peer stores on real xGMI (every arena is local memory here)."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import f_m1, lognormal_coeff, u0734

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from test_gpu_parity import assert_history, orc_ops

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def run_ranks(api, world, rank_main, mode=0, timeout=600):
    group = api.LoopbackGroup(world)
    group.set_mode(mode)
    out, errs = [None] * world, []
    group.host_barrier = threading.Barrier(world)   # ranks line up before their first exchange (set-up times differ by seconds)

    def main(r):
        try:
            ctx = api.Context(0)
            ctx.loopback_init(group, r)
            ctx.host_barrier = group.host_barrier
            out[r] = rank_main(ctx, r)
        except Exception as e:                                   # noqa: BLE001
            errs.append((r, repr(e)))
            group.host_barrier.abort()                           # the others must not wait for a rank that is gone

    threads = [threading.Thread(target=main, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=timeout)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"
    return out


def sharded_ops(api, ctx, P, r, world, shard_nn):
    ndom = P.sub.ndom
    gi, cnt = P.sub.gather_idx, P.sub.node_Γ_cnt
    lo, hi = api.shard_domains(ndom, r, world)
    S = api.LocalSchurs(ctx, [P.Sd[d] if lo <= d < hi else None for d in range(ndom)], gi, cnt, dom_slice=(lo, hi))
    if shard_nn:
        M = api.NeumannNeumannSchurPreconditioner(ctx, [P.ΠSd[d] if lo <= d < hi else None for d in range(ndom)], gi, cnt,
                                                  dom_slice=(lo, hi))
    else:
        M = api.NeumannNeumannSchurPreconditioner(ctx, P.ΠSd, gi, cnt, dom_slice=(0, ndom))
    return S, M


@pytest.mark.parametrize("world,shard_nn", [(8, True), (8, False), (2, True), (2, False), (4, True), (4, False)])
def test_config4_full_size_in_process_ranks(pkg, orc, full, world, shard_nn):
    """N = 1000, 4x2 subdomains; world = 8 is configs[3] itself (one subdomain per rank). pcg on every rank through the
    folded loop across the ranks, iterations replayed from hipGraphs that contain the exchanges."""
    api, P = pkg.api, full
    n, b = P.sub.n_Γ, P.b_schur
    assert int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= world, "conftest sets GPU_MAX_HW_QUEUES before the first HIP call"

    def rank_main(ctx, r):
        S, M = sharded_ops(api, ctx, P, r, world, shard_nn)
        assert ctx.query("peer_exchange") >= 1 and ctx.query("no_graph") == 0
        ctx.host_barrier.wait(timeout=300)
        y = S * b                                                # a sharded plain apply
        e0, g0 = ctx.query("exchanges"), ctx.query("graph_replays")
        res = api.pcg(S, b, np.zeros(n), M)
        e1, g1 = ctx.query("exchanges"), ctx.query("graph_replays")
        res2 = api.pcg(S, b, np.zeros(n), M)                     # replays the instantiated graphs
        return y, res, res2, (e1 - e0, g1 - g0)

    out = run_ranks(api, world, rank_main)
    for r in range(1, world):                                    # every rank holds the same bits
        assert np.array_equal(out[r][0], out[0][0])
        for k in (1, 2):
            assert out[r][k][1] == out[0][k][1] and np.array_equal(out[r][k][2], out[0][k][2]) and np.array_equal(out[r][k][0], out[0][k][0])
        assert out[r][3] == out[0][3]
    y, res, res2, (n_xchg, n_replay) = out[0]
    assert res2[1] == res[1] and np.array_equal(res2[2], res[2]) and np.array_equal(res2[0], res[0])
    So, Mo = orc_ops(orc, P)
    assert np.allclose(y, So * b, rtol=0, atol=1e-13 * np.abs(y).max())
    want = orc.pcg(So, b, np.zeros(n), Mo)
    assert_history(res, want)                                    # it equal, res_norm entry by entry, x
    # the captured loop ran, and it contained the exchanges: one per iteration with S sharded, two with both sharded
    assert n_replay >= 1
    per_it = 2 if shard_nn else 1
    assert n_xchg >= per_it * (res[1] - 1), (n_xchg, res[1])


def test_push_kernel_exchange_gives_the_same_bits(pkg, full, monkeypatch):
    """The exchange has two producers: the folded launches storing into every arena themselves (default), and the local
    pack pushed by `k_xchg_push` (MI355_XCHG_PUSH_KERNEL=1; also what a rank without a block falls back to). Same tables,
    same bits."""
    api, P = pkg.api, full
    n, b, world = P.sub.n_Γ, P.b_schur, 4

    def rank_main(ctx, r):
        S, M = sharded_ops(api, ctx, P, r, world, True)
        ctx.host_barrier.wait(timeout=300)
        return api.pcg(S, b, np.zeros(n), M)

    direct = run_ranks(api, world, rank_main)
    monkeypatch.setenv("MI355_XCHG_PUSH_KERNEL", "1")
    pushed = run_ranks(api, world, rank_main)
    for r in range(world):
        assert pushed[r][1] == direct[0][1] and np.array_equal(pushed[r][2], direct[0][2]) and np.array_equal(pushed[r][0], direct[0][0])


@pytest.mark.parametrize("world,shard_nn", [(2, True), (4, True), (2, False)])
def test_launches_that_wait_for_their_peers_themselves(pkg, toy, world, shard_nn):
    """`mi_ctx_set_exchange(ctx, 2)`: no kernel between the folded launches — the consuming launch polls the flags, the
    exchange number travels like it / it_nxt. Meant for one GPU per rank (a waiting launch keeps its compute units); on one
    GPU it can only run where the launches of all ranks fit on the chip together: the toy problem (a few dozen small
    workgroups per rank). Same bits as the wait-kernel mode, and the exchange counter ends where it should."""
    api, P = pkg.api, toy
    n, b = P.sub.n_Γ, P.b_schur

    def rank_main(mode):
        def f(ctx, r):
            S, M = sharded_ops(api, ctx, P, r, world, shard_nn)
            ctx.set_exchange(mode)
            ctx.host_barrier.wait(timeout=300)
            e0 = ctx.query("exchanges")
            res = api.pcg(S, b, np.zeros(n), M)
            e1 = ctx.query("exchanges")
            res2 = api.pcg(S, b, np.zeros(n), M, maxit=5)          # stopped by maxit: the counter must still be exact
            y = S * b                                              # a generic all-reduce afterwards uses the same counter
            return res, res2, y, e1 - e0
        return f

    kernel_wait = run_ranks(api, world, rank_main(1))
    in_launch = run_ranks(api, world, rank_main(2))
    for r in range(world):
        for k in (0, 1):
            a, w = in_launch[r][k], kernel_wait[0][k]
            assert a[1] == w[1] and np.array_equal(a[2], w[2]) and np.array_equal(a[0], w[0])
        assert np.array_equal(in_launch[r][2], kernel_wait[0][2])
        assert in_launch[r][3] == kernel_wait[0][3]


def test_config4_deflated_and_host_rendezvous(pkg, orc, full):
    """defpcg across 8 in-process ranks at full size (S sharded, NN replicated: the deflated loop all-reduces the slot table
    after every S-apply, also for `WtA = (A W)'`), and the same pcg through the group's host-rendezvous mode (eager
    launches) — both against the oracle, ranks bit-identical."""
    api, P = pkg.api, full
    n, b, ndom, world = P.sub.n_Γ, P.b_schur, P.sub.ndom, 8
    So, Mo = orc_ops(orc, P)
    nvec, spdim = 8, 20                                           # (the solve takes 16 iterations: at most that many Ritz vectors exist)
    W = orc.eigpcg(So, b, np.zeros(n), Mo, nvec, spdim)[3]
    b2 = So(np.random.default_rng(4).standard_normal(n))

    def rank_main(ctx, r):
        S, M = sharded_ops(api, ctx, P, r, world, False)
        ctx.host_barrier.wait(timeout=300)
        return api.defpcg(S, b2, np.zeros(n), W, M), ctx.query("graph_replays")

    out = run_ranks(api, world, rank_main)
    for r in range(1, world):
        assert out[r][0][1] == out[0][0][1] and np.array_equal(out[r][0][2], out[0][0][2]) and np.array_equal(out[r][0][0], out[0][0][0])
    assert out[0][1] >= 1
    assert_history(out[0][0], orc.defpcg(So, b2, np.zeros(n), W, Mo))

    def rank_host(ctx, r):
        S, M = sharded_ops(api, ctx, P, r, world, True)
        assert ctx.query("no_graph") == 1 and ctx.query("peer_exchange") == 0
        ctx.host_barrier.wait(timeout=300)
        return api.pcg(S, b, np.zeros(n), M)

    outh = run_ranks(api, world, rank_host, mode=1)
    for r in range(1, world):
        assert outh[r][1] == outh[0][1] and np.array_equal(outh[r][2], outh[0][2]) and np.array_equal(outh[r][0], outh[0][0])
    assert_history(outh[0], orc.pcg(So, b, np.zeros(n), Mo))


def test_peer_exchange_handshake_and_expired_wait(pkg):
    """The explicit hand-shake of include/mi355schur.h (init / export / import / ready) between two contexts of this
    process, an all-reduce through it, and the bounded wait: a rank whose peer never arrives gets MI_ERR_COMM-style
    failure instead of a hang (short timeout through MI355_PEER_TIMEOUT_MS)."""
    api = pkg.api
    os.environ["MI355_PEER_TIMEOUT_MS"] = "300"
    try:
        c0, c1 = api.Context(0), api.Context(0)
        c0.peer_init(0, 2)
        c1.peer_init(1, 2)
        (_, b0), (_, b1) = c0.peer_export(), c1.peer_export()
        c0.peer_import(1, same_process_base=b1)
        c1.peer_import(0, same_process_base=b0)
        c0.peer_ready()
        c1.peer_ready()
    finally:
        os.environ.pop("MI355_PEER_TIMEOUT_MS", None)
    v0, v1 = np.arange(5.0), 10.0 * np.arange(5.0)
    res = [None, None]
    t = threading.Thread(target=lambda: res.__setitem__(1, c1.allreduce_sum(v1.copy())), daemon=True)
    t.start()
    res[0] = c0.allreduce_sum(v0.copy())
    t.join(timeout=60)
    assert np.array_equal(res[0], v0 + v1) and np.array_equal(res[1], v0 + v1)
    assert c0.query("exchanges") == 1 and c1.query("exchanges") == 1
    # rank 1 stays away: rank 0's wait expires after 0.3 s and the call fails instead of hanging
    import time
    t0 = time.time()
    with pytest.raises(Exception, match="expired"):
        c0.allreduce_sum(v0.copy())
    assert time.time() - t0 < 30.0


def test_peer_exchange_across_processes(pkg, orc, fem, tmp_path):
    """The production hand-shake: 2 PROCESSES (one context each, both on this GPU), arenas mapped through HIP IPC handles that
    travel over torch.distributed (gloo, 127.0.0.1) — tests/peer_ipc_worker.py. Both operators sharded, folded loop, the
    exchanges inside the iteration graphs; ranks bit-identical, oracle parity, and a plain all-reduce through the arenas."""
    import socket
    import subprocess
    import sys
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "peer_ipc_worker.py")
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, outs[r]], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = (np.load(o) for o in outs)
    assert int(r0["peer"]) >= 1 and int(r0["replays"]) >= 1 and int(r0["exchanges"]) == int(r1["exchanges"])
    assert np.array_equal(r0["y"], r1["y"]) and np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["res"], r1["res"])
    assert int(r0["it"]) == int(r1["it"])
    assert np.array_equal(r0["v"], np.full(7, 3.0)) and np.array_equal(r1["v"], np.full(7, 3.0))
    from conftest import f_m1 as f1, lognormal_coeff as lc, u0734 as u0
    mesh = fem.get_mesh(90)
    P = fem.build_schur_problem(90, 4, 2, lc(fem, mesh.points, 5), f1, u0)
    So, Mo = orc_ops(orc, P)
    n, b = P.sub.n_Γ, P.b_schur
    assert np.allclose(r0["y"], So * b, rtol=0, atol=1e-13 * np.abs(r0["y"]).max())
    assert_history((r0["x"], int(r0["it"]), r0["res"]), orc.pcg(So, b, np.zeros(n), Mo))


def test_bench_rehearses_its_multi_gpu_path_on_one_gpu():
    """`bench.py --force-dist`: the code `--gpus N` runs — rendezvous, peer hand-shake, the self-check of the exchange
    against RCCL, the timing of both layouts, the JSON contract — with ONE rank on this GPU (N = 200: seconds)."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--N", "200", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-secondary", "--kernel-reps", "20"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    d = json.loads(line[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    cfg = d["config"]
    assert cfg["exchange"] in ("peer", "rccl") and cfg["launches_per_iteration"] == 2
    lm = cfg["layouts_measured"]                                   # both layouts ran and were timed; the faster one was benchmarked
    assert lm and lm["peer_both_sharded_ms_per_solve"] > 0 and lm["rccl_nn_replicated_ms_per_solve"] > 0
    assert (cfg["exchange"] == "rccl") == (lm["rccl_nn_replicated_ms_per_solve"] < 0.97 * lm["peer_both_sharded_ms_per_solve"])
    assert "peer exchange rejected" not in r.stderr
