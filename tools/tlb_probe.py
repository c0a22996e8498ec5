#!/usr/bin/env python
"""Does the random-row access rate depend on the table's footprint (TLB reach)?  Uniform gather of 4 M rows of 512 B
from tables of growing size (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
import bench
dev = torch.device("cuda:0")
n, k = 1 << 22, 128
for rows in (1 << 20, 1 << 22, 10_000_000, 40_000_000, 100_000_000):
    T = torch.empty((rows, k), dtype=torch.float32, device=dev)
    T.normal_()
    ids = torch.randint(0, rows, (n,), device=dev).to(torch.int32)
    ms = bench.ev_time(lambda: K.gather(T, ids), 5, torch.cuda.current_stream())
    # sorted ids: same rows, ascending addresses (page-local)
    sid = torch.sort(ids)[0].contiguous()
    ms2 = bench.ev_time(lambda: K.gather(T, sid), 5, torch.cuda.current_stream())
    gb = n * (4 + 2 * 4 * k) / 1e6
    print("table %6.1f GB: random %.3f ms %.0f GB/s | sorted ids %.3f ms %.0f GB/s" % (rows * k * 4 / 1e9, ms, gb / ms, ms2, gb / ms2), flush=True)
    del T, ids, sid
    torch.cuda.empty_cache()
