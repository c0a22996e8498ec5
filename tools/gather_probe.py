"""Gather throughput at the bench's three shapes (dev tool): python tools/gather_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
P = torch.empty((10_000_000, 128), device=dev); K.init_normal_(P, 1, 0.01)
def run(ids, reps):
    K.gather(P, ids); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): K.gather(P, ids)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    return ms, 2 * ids.numel() * 512 / ms / 1e6
for n, reps in ((65536, 200), (4 << 20, 20)):
    ids = torch.randint(0, P.shape[0], (n,), device=dev, generator=g).to(torch.int32)
    ms, gb = run(ids, reps)
    print("rows %8d: %.4f ms  %.0f GB/s (%.3f of 8 TB/s)" % (n, ms, gb, gb / 8000))
