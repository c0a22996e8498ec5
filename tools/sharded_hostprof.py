#!/usr/bin/env python
"""Host side of the row-sharded step at world 1 (C4 batch): loop time before / after the final synchronisation and a
cProfile of the step loop (dev tool)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
from deeprec_amd import sharding as sh
import bench
dev = torch.device("cuda:0")
if "--pg" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for a in sys.argv:
    if a.startswith("--sides="):
        sh.ShardedBpr.N_SIDE = int(a[8:])
MAIN = torch.cuda.Stream(device=dev, priority=-1 if "--hi" in sys.argv else 0) if "--own-main" in sys.argv else None
if MAIN is not None and "--early" in sys.argv:
    torch.cuda.set_stream(MAIN)
class A: users, items, batch = 10_000_000, 1_000_000, 65536
k = 128
bs = bench.make_batches(A, 0, dev, 8)
Pt = sh.ShardedTable(A.users, k, 1, 0, dev, 2); Qt = sh.ShardedTable(A.items, k, 1, 0, dev, 2)
Pt.init_normal_(K, 1, 0.01); Qt.init_normal_(K, 2, 0.01)
tr = sh.ShardedBpr(K, Pt, Qt)
tr.prefetch(*bs[0]); tr.prefetch(*bs[1])
opt = K.opt_params(K.OPT_ADAM, lr=0.05, alpha=1e-3, eps=1e-7, reg=0.1, reg_mode=K.REG_TOUCHED)
def opt_for(step):
    b1p, b2p = 0.9 ** step, 0.999 ** step
    return K.opt_params(K.OPT_ADAM, lr=0.05, alpha=1e-3 * (1 - b2p) ** 0.5 / (1 - b1p), beta1=0.9, beta2=0.999, eps=1e-7, reg=0.1, reg_mode=K.REG_TOUCHED)
VAR = "--var-opt" in sys.argv
def loop(n, t0=0):
    for t in range(n):
        tr.step(opt_for(t0 + t + 1) if VAR else opt, next_batch=bs[(t0 + t + 2) % 8])
if MAIN is not None:
    torch.cuda.synchronize(); torch.cuda.set_stream(MAIN)
loop(20); torch.cuda.synchronize()
a = time.perf_counter(); loop(400, 20); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
print("400 steps: host loop %.3f ms/step, with final sync %.3f ms/step" % ((b - a) / 400 * 1e3, (c - a) / 400 * 1e3))
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable(); loop(400, 420); pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
