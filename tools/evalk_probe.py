#!/usr/bin/env python
"""C5-shaped evaluation (65 536 users x 100 k items, two-distance score, d = 128) at K = 1 / 10 / 50: the target ranks do
not depend on K, the time does (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
dev = torch.device("cuda:0")
Nu, Ni, d = 65536, 100_000, 128
g = torch.Generator(device=dev); g.manual_seed(5)
U1, U2 = torch.randn(Nu, d, device=dev, generator=g) * 0.1, torch.randn(Nu, d, device=dev, generator=g) * 0.1
I1, I2 = torch.randn(Ni, d, device=dev, generator=g) * 0.1, torch.randn(Ni, d, device=dev, generator=g) * 0.1
tr = torch.sort(torch.randint(1, Ni, (Nu, 50), device=dev, generator=g), 1)[0].to(torch.int32).reshape(-1).contiguous()
tp = (torch.arange(Nu + 1, device=dev) * 50).to(torch.int32)
tg = torch.randint(1, Ni, (Nu,), device=dev, generator=g).to(torch.int32)
gp = torch.arange(Nu + 1, device=dev).to(torch.int32)
ranks = {}
for Kk in (1, 10, 50):
    f = lambda: K.score_topk(U1, I1, Kk, score_kind=K.SCORE_TWO_DIST, U2=U2, I2=I2, alpha=0.2, train=(tp, tr), targets=(gp, tg))
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    ranks[Kk] = r[2].clone()
    print("K=%d: %.1f ms, %.1f TFLOP/s" % (Kk, ms, 2.0 * Nu * Ni * 2 * d / ms / 1e9))
print("ranks equal:", torch.equal(ranks[1], ranks[10]), torch.equal(ranks[1], ranks[50]))
