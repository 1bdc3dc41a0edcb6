#!/usr/bin/env python3
"""Kernel durations of the multi-GPU (sharded) folded loop on ONE GPU: `world` in-process ranks joined by the loopback
communicator run pcg on the 1M-DoF problem saved by `tools/profile_defl.py save`. The ranks share the GPU and the
collective synchronises with the host, so only the per-kernel durations (rocprofv3 --kernel-trace --stats) mean anything:
they show what one rank's S launch (owned blocks cut into small tiles + owner-duty tiles) and ΠS launch cost.
    rocprofv3 --kernel-trace --stats ... -- python3 tools/sharded_probe.py /tmp/p.npz 8"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
api = pkg.api
z = np.load(sys.argv[1])
world = int(sys.argv[2])
nd = int(z["ndom"])
gi = [z[f"g{k}"] for k in range(nd)]
cnt, b = z["cnt"], z["b"]
n = b.size
group = api.LoopbackGroup(world)
its = [None] * world


def rank_main(r):
    ctx = api.Context(0)
    ctx.loopback_init(group, r)
    lo, hi = api.shard_domains(nd, r, world)
    S = api.LocalSchurs(ctx, [z[f"S{k}"] if lo <= k < hi else None for k in range(nd)], gi, cnt, dom_slice=(lo, hi))
    M = api.NeumannNeumannSchurPreconditioner(ctx, [z[f"P{k}"] for k in range(nd)], gi, cnt, dom_slice=(0, nd))
    for _ in range(3):
        its[r] = api.pcg(S, b, np.zeros(n), M)[1]


threads = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=600)
print("world", world, "it", its)
