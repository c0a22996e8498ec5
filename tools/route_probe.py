#!/usr/bin/env python
"""GPU time of dr_shard_bpr_route as one rank of a world of W sees it (ids of W ranks replicated; dev tool).
Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
import bench
dev = torch.device("cuda:0")
class A: users, items, batch = 10_000_000, 1_000_000, 65536
B, k = A.batch, 128
st = torch.cuda.current_stream()
for W in (1, 2, 4, 8):
    ids = torch.stack([torch.stack(bench.make_batches(A, r, dev, 1)[0]) for r in range(W)]).contiguous()
    ws = K.ShardWorkspace(W, 0, B, k, A.users, A.items, dev)
    for _ in range(3):
        K.shard_route(ws, ids)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(st)
    for _ in range(20):
        K.shard_route(ws, ids)
    e.record(st); e.synchronize()
    c = ws.counts.tolist()
    print("W=%d: %.3f ms per route; processes %d samples, owns %d entries, sends %d / receives %d gradient rows, fetches %d rows; "
          "workspace %.1f MB" % (W, s.elapsed_time(e) / 20, c[0], c[6], c[2], c[7], sum(c[8:8 + W]), ws.buf.numel() / 1e6))
