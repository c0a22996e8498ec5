#!/usr/bin/env python
"""NeuMF full-catalog scoring: dr_tower_catalog_scores_f32 against the library-GEMM path (dev tool).
    python tools/neumf_eval_probe.py [users] [items]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
from deeprec_amd.models.item_ranking.neumf import NeuMF

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 6040
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 3706
with contextlib.redirect_stdout(io.StringIO()):
    m = NeuMF(None, nu, ni, 0.5, 0.01, 1, 256)
    m.build_network()
users = list(range(nu))
flop = 2.0 * nu * ni * (128 * 80 + 80 * 40 + 40 * 20 + 20 * 10)
for name, fused in (("fused", True), ("library", False)):
    sup = K.tower_catalog_supported
    if not fused:
        K.tower_catalog_supported = lambda dims: False
    n_eval = users if fused else users[:min(nu, 1024)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = 0.0
        for c0, c1, sc in m.eval_score_chunks(n_eval):
            s += float(sc[0, 0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    K.tower_catalog_supported = sup
    pairs = len(n_eval) * ni
    print("%-8s %d users x %d items: %.1f ms, %.2f G pairs/s, %.1f TFLOP/s (tower only, useful flops)"
          % (name, len(n_eval), ni, dt * 1e3, pairs / dt / 1e9, flop * len(n_eval) / nu / dt / 1e12))
