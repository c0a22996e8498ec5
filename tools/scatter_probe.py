"""Time dr_scatter_add_sorted_f32 / dr_gather_f32 standalone (development probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
import bench

dev = torch.device("cuda:0")
rows, k = 10_000_000, 128
n = int(os.environ.get("N", 1 << 22))
g = torch.Generator(device=dev); g.manual_seed(13)
sid = torch.randint(0, rows, (n,), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
src = torch.empty((n, k), device=dev); K.init_normal_(src, 3, 0.01)
gt = torch.zeros((rows, k), device=dev)
plan = K.SegmentPlan(sid, rows, k_max=k)
uq = int(plan.view()[1].numel())
st = torch.cuda.current_stream()
t = bench.ev_time(lambda: K.scatter_add(gt, sid, src, plan=plan), 5, st)
b = n * (4 + 4 * k) + uq * 8 * k
print("scatter_add n=%d uq=%d: %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (n, uq, t, b / t / 1e6, b / t / 1e6 / 80))
t = bench.ev_time(lambda: K.gather(gt, sid), 5, st)
b = n * (4 + 8 * k)
print("gather n=%d: %.3f ms  %.0f GB/s (%.1f%%)" % (n, t, b / t / 1e6, b / t / 1e6 / 80))
