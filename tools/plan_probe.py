#!/usr/bin/env python
"""dr_segment_plan alone at the bench's scatter-add shape (4 M ids over 10 M rows): run under rocprofv3 --kernel-trace --stats (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
dev = torch.device("cuda:0")
n, rows = 1 << 22, 10_000_000
ids = torch.randint(0, rows, (n,), device=dev, dtype=torch.int32)
ws = None
for _ in range(3):
    p = K.SegmentPlan(ids, rows, k_max=128)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    p = K.SegmentPlan(ids, rows, k_max=128)
torch.cuda.synchronize()
print("plan of %d ids: %.3f ms" % (n, (time.perf_counter() - t0) / 10 * 1e3))
