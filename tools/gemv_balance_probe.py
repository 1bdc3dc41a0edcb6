#!/usr/bin/env python3
"""Is the batched GEMV bound by the LARGEST tile (one workgroup per CU, per-CU stream rate) rather than by total
bytes? Synthetic disjoint subdomains of chosen sizes; us/launch and algorithmic GB/s of the plain S-apply kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
api = pkg.api
import torch  # noqa: E402

ctx = api.Context(0)
rng = np.random.default_rng(0)
cases = {
    "8 x 1024 (256 equal tiles)": [1024] * 8,
    "config 3 (4 x 748 + 4 x 1248)": [748, 1248, 1248, 748, 748, 1248, 1248, 748],
    "4 x 1248 + 4 x 748 (sorted)": [1248] * 4 + [748] * 4,
    "8 x 1248": [1248] * 8,
    "8 x 748": [748] * 8,
    "16 x 748": [748] * 16,
    "8 x 2048": [2048] * 8,
    "32 x 1024": [1024] * 32,
}
for name, sizes in cases.items():
    Sd, gi, off = [], [], 0
    for nd in sizes:
        Sd.append(np.asfortranarray(rng.standard_normal((nd, nd))))
        gi.append(np.arange(off, off + nd))
        off += nd
    S = api.LocalSchurs(ctx, Sd, gi, np.ones(off, dtype=np.int64))
    x = torch.from_numpy(rng.standard_normal(off)).cuda()
    _, nb = S.bytes()
    us = min(S.time_dominant(x, 300) for _ in range(3))
    ntiles = sum((nd + 31) // 32 for nd in sizes)
    print(f"{name:34s} tiles={ntiles:4d} bytes={nb / 1e6:7.1f} MB  {us:7.2f} us  {nb / us / 1e3:7.1f} GB/s", flush=True)
    del S
