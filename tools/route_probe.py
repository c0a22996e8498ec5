#!/usr/bin/env python
"""GPU time of dr_route_views_i32 as one rank of a world of W sees it (ids of W ranks replicated; dev tool).
    python tools/route_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deeprec_amd import kernels as K

dev = torch.device("cuda:0")
nu, ni, B = 10_000_000, 1_000_000, 65536
rng = np.random.default_rng(0)
for W in (1, 2, 4, 8):
    u = rng.integers(0, nu, (W, B))
    i = (rng.zipf(1.05, (W, B)) - 1) % ni
    j = rng.integers(0, ni, (W, B))
    ids = torch.from_numpy(np.stack((u, i, j), 1).astype(np.int32)).to(dev)
    rpo = (ni + W - 1) // W
    for me in (0, W - 1):
        K.route_views(ids, me, nu, ni, rpo)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            rv = K.route_views(ids, me, nu, ni, rpo)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        c = rv.counts.tolist()
        print("W=%d me=%d: %.3f ms per route_views; entries view0 %d view1 %d (B = %d)" % (W, me, ms, c[4 * W], c[4 * W + 1], B))
