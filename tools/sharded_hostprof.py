#!/usr/bin/env python
"""cProfile of the host side of ShardedBpr.step at world 1 through RCCL (dev tool)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from deeprec_amd import kernels as K
from deeprec_amd import sharding as sh

sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29515")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
batches = bench.make_batches(args, 0, dev, 8)
Pt = sh.ShardedTable(args.users, args.k, 1, 0, dev, 2)
Qt = sh.ShardedTable(args.items, args.k, 1, 0, dev, 2)
Pt.init_normal_(K, 1, 0.01)
Qt.init_normal_(K, 2, 0.01)
opt = K.opt_params(K.OPT_ADAM, lr=0.05, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, reg=0.1, reg_mode=K.REG_TOUCHED)
tr = sh.ShardedBpr(K, Pt, Qt)
tr.prefetch(*batches[0])
for s in range(20):
    tr.step(opt, batches[(s + 1) % 8])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
N = 200
for s in range(N):
    tr.step(opt, batches[(s + 1) % 8])
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats("tottime")
st.print_stats(28)
print(out.getvalue())
dist.destroy_process_group()
