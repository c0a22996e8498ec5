#!/usr/bin/env python
"""Time the full-catalog scoring + top-K + rank kernel alone (dev tool).

    python tools/eval_probe.py [c4|c5|c4r|c5r|c5c|c5x] [iters]      (..r: ranks only, dr_score_rank_f32; c5c: Caser's dot + item
                                                                  bias over 256 factors, ranks only; c5x: dr_score_rank_exact_f32)
c4: 16 384 users x 1 M items x 128 (dot, K = 10, 100 masked train items and one held-out item per user) -- bench.py's eval.
c5: 65 536 users x 100 k items, two-distance score over 2 x 128 (K = 1, 50 masked train items, one held-out item).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeprec_amd import _abi
from deeprec_amd import kernels as K

if os.environ.get("DR_PROBE_LIB"):          # an ablation build (tools/build_abl.sh)
    _abi._LIB = _abi.load(os.environ["DR_PROBE_LIB"])


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "c4"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    multi = shape == "c4m"      # five held-out items per user (RankingMetrics-style test sets): the general kernel
    ranks_only = shape in ("c4r", "c5r", "c5c", "c5x")
    exact = shape == "c5x"
    caser = shape == "c5c"
    shape = shape[:2]
    if shape == "c4":
        Nu, Ni, d, ntr, Kk = 16384, 1_000_000, 128, 100, 10
    else:
        Nu, Ni, d, ntr, Kk = 65536, 100_000, 128, 50, 1
    U = torch.randn(Nu, d, device=dev, generator=g) * 0.1
    I = torch.randn(Ni, d, device=dev, generator=g) * 0.1
    tr = torch.sort(torch.randint(0, Ni, (Nu, ntr), device=dev, generator=g), 1)[0].to(torch.int32).reshape(-1).contiguous()
    tp = (torch.arange(Nu + 1, device=dev) * ntr).to(torch.int32)
    tg = torch.randint(0, Ni, (Nu,), device=dev, generator=g).to(torch.int32)
    gp = torch.arange(Nu + 1, device=dev).to(torch.int32)
    if multi:
        tg = torch.randint(0, Ni, (Nu * 5,), device=dev, generator=g).to(torch.int32)
        gp = (torch.arange(Nu + 1, device=dev) * 5).to(torch.int32)
    if shape == "c4":
        run = lambda: K.score_topk(U, I, Kk, train=(tp, tr), targets=(gp, tg))
        if ranks_only:
            run = lambda: K.score_rank(U, I, (gp, tg), train=(tp, tr))
        flops = 2.0 * Nu * Ni * d
    elif caser:
        Uc = torch.randn(Nu, 2 * d, device=dev, generator=g) * 0.1
        Wc = torch.randn(Ni, 2 * d, device=dev, generator=g) * 0.1
        bc = torch.randn(Ni, device=dev, generator=g) * 0.1
        run = lambda: K.score_rank(Uc, Wc, (gp, tg), item_bias=bc, train=(tp, tr))
        flops = 2.0 * Nu * Ni * 2 * d
    else:
        U2 = torch.randn(Nu, d, device=dev, generator=g) * 0.1
        I2 = torch.randn(Ni, d, device=dev, generator=g) * 0.1
        run = lambda: K.score_topk(U, I, Kk, score_kind=K.SCORE_TWO_DIST, U2=U2, I2=I2, alpha=0.2, train=(tp, tr),
                                   targets=(gp, tg))
        if ranks_only:
            run = lambda: K.score_rank(U, I, (gp, tg), score_kind=K.SCORE_TWO_DIST, U2=U2, I2=I2, alpha=0.2, train=(tp, tr))
        if exact:       # ranks in numpy's operation order (what evaluate1 calls)
            import numpy as np
            run = lambda: K.score_rank_exact(U, I, (gp, tg), score_kind=K.SCORE_TWO_DIST, U2=U2, I2=I2, w1=np.float32(0.2),
                                             w2=np.float32(0.8), train=(tp, tr))
        flops = 2.0 * Nu * Ni * 2 * d
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print("%s%s%s: %.2f ms, %.1f TFLOP/s (%.3f of 157.3)" % (shape, " ranks only" if ranks_only else "", " 5 held-out items per user" if multi else "", ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3))


main()
