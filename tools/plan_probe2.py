#!/usr/bin/env python
"""The C4 batch's plans alone (dr_bpr_plan_f32: 196 608 keys; dr_segment_plan: 131 072 keys): per-call time with HIP
events; run under rocprofv3 --kernel-trace --stats for the per-kernel split (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
import bench
dev = torch.device("cuda:0")
class A: users, items, batch = 10_000_000, 1_000_000, 65536
bs = bench.make_batches(A, 0, dev, 4)
B, k = A.batch, 128
ws = K.StepWorkspace(B, k, A.users, A.items, K.REG_TOUCHED, dev)
st = torch.cuda.current_stream()
def ev(fn, it=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(st)
    for _ in range(it): fn()
    e.record(st); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
print("bpr_plan 3B=%d keys: %.1f us" % (3 * B, ev(lambda: K.bpr_plan(A.users, A.items, k, *bs[0], ws))))
idsQ = torch.cat((bs[0][1], bs[0][2])).contiguous()
pl = K.SegmentPlan(idsQ, A.items)
print("segment_plan 2B=%d keys: %.1f us" % (2 * B, ev(lambda: K.SegmentPlan(idsQ, A.items, pl.ws))))
big = torch.randint(0, A.users, (1 << 22,), device=dev, dtype=torch.int32)
pb = K.SegmentPlan(big, A.users, k_max=k)
print("segment_plan 4M keys: %.1f us" % ev(lambda: K.SegmentPlan(big, A.users, pb.ws, k_max=k), 10))
