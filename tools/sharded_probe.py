#!/usr/bin/env python
"""Where does a row-sharded step spend its time on ONE GPU (no fabric)?  Host-side enqueue time of run() and
route() against the GPU time of the same work (dev tool; bench.py --sharded gives the end-to-end number).

    python tools/sharded_probe.py [--rccl] [--steps N]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
from deeprec_amd import kernels as K
from deeprec_amd import sharding as sh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rccl", action="store_true", help="exchanges through a size-1 RCCL group instead of none")
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    sys.argv = ["bench.py"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if a.rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    batches = bench.make_batches(args, 0, dev, 8)
    Pt = sh.ShardedTable(args.users, args.k, 1, 0, dev, 2)
    Qt = sh.ShardedTable(args.items, args.k, 1, 0, dev, 2)
    Pt.init_normal_(K, 1, 0.01)
    Qt.init_normal_(K, 2, 0.01)
    opt = K.opt_params(K.OPT_ADAM, lr=0.05, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, reg=0.1, reg_mode=K.REG_TOUCHED)
    tr = sh.ShardedBpr(K, Pt, Qt, pipelined=False)
    sync = torch.cuda.synchronize

    def timed(fn, n):
        """(host enqueue ms, total ms incl. GPU drain) per call"""
        fn()
        sync()
        h = g = 0.0
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            t1 = time.perf_counter()
            sync()
            t2 = time.perf_counter()
            h += t1 - t0
            g += t2 - t0
            del r
        return h / n * 1e3, g / n * 1e3

    nb = len(batches)
    cnt = [0]

    def do_route():
        cnt[0] += 1
        return tr.route(*batches[cnt[0] % nb])

    h, g = timed(do_route, a.steps)
    print("route(): host %.3f ms, host+GPU %.3f ms" % (h, g))
    rt = tr.route(*batches[0])
    h, g = timed(lambda: tr.run(rt, opt), a.steps)
    print("run():   host %.3f ms, host+GPU %.3f ms" % (h, g))
    # pieces of route()
    u, i, j = batches[0]
    ids = torch.stack((u, i, j)).reshape(1, 3, -1)
    h, g = timed(lambda: K.route_views(ids, 0, args.users, args.items, args.items), a.steps)
    print("  route_views:   host %.3f, total %.3f" % (h, g))
    rv = K.route_views(ids, 0, args.users, args.items, args.items)
    h, g = timed(lambda: rv.counts.tolist(), a.steps)
    print("  counts sync:   host %.3f, total %.3f" % (h, g))
    if a.rccl:
        out = torch.empty_like(ids)
        h, g = timed(lambda: sh._gather_ids(out, ids[0], None), a.steps)
        print("  id all-gather through RCCL (self): host %.3f, total %.3f" % (h, g))
    h, g = timed(lambda: K.SegmentPlan(rt.ids_canon, args.items, k_max=args.k), a.steps)
    print("  plan_q:        host %.3f, total %.3f" % (h, g))
    h, g = timed(lambda: K.SegmentPlan(rt.u_loc, args.users, k_max=args.k), a.steps)
    print("  plan_p:        host %.3f, total %.3f" % (h, g))
    # pieces of run()
    h, g = timed(lambda: K.gather(Qt.w, rt.rows_fetch), a.steps)
    print("  gather Q rows (%d distinct of %d): host %.3f, total %.3f" % (rt.n_uq, 2 * rt.n_s, h, g))
    qc = K.gather(Qt.w, rt.rows_fetch)
    h, g = timed(lambda: K.bpr_forward(Pt.w, qc, rt.u_loc, rt.inv_i, rt.inv_j), a.steps)
    print("  forward:       host %.3f, total %.3f" % (h, g))
    gc, lb, stash = K.bpr_forward(Pt.w, qc, rt.u_loc, rt.inv_i, rt.inv_j)
    h, g = timed(lambda: K.bpr_item_grads(Pt.w, rt.u_loc, gc, rt.order_g), a.steps)
    print("  item grads:    host %.3f, total %.3f" % (h, g))
    qg = K.bpr_item_grads(Pt.w, rt.u_loc, gc, rt.order_g)
    h, g = timed(lambda: K.sparse_row_update(Pt.w, Pt.slot(0), Pt.slot(1), rt.u_loc, stash, opt, plan=rt.plan_p), a.steps)
    print("  update P:      host %.3f, total %.3f" % (h, g))
    h, g = timed(lambda: K.sparse_row_update(Qt.w, Qt.slot(0), Qt.slot(1), rt.ids_canon, qg, opt, src_idx=rt.perm,
                                             plan=rt.plan_q), a.steps)
    print("  update Q:      host %.3f, total %.3f" % (h, g))
    # the pipelined step with the host synchronisation taken out (split sizes cached from a first route of the same
    # batch): what is left is max(host enqueue time, GPU time of route + run on two streams)
    tp = sh.ShardedBpr(K, Pt, Qt, pipelined=True)
    cached = {}

    class Cached:
        def __init__(self, key, real):
            self.key, self.real = key, real

        def tolist(self):
            if self.key not in cached:
                cached[self.key] = self.real.tolist()
            return cached[self.key]

    real_begin = tp._route_begin
    which = [0]

    def begin(u_, i_, j_):
        ids_, rv_ = real_begin(u_, i_, j_)
        rv_.counts = Cached(which[0], rv_.counts)
        return ids_, rv_

    tp._route_begin = begin
    for rep in range(2):
        which[0] = 0
        tp.prefetch(*batches[0])
        sync()
        t0 = time.perf_counter()
        for s_ in range(a.steps):
            which[0] = (s_ + 1) % nb
            tp.step(opt, batches[(s_ + 1) % nb])
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
    print("pipelined step, no host sync: host %.3f ms / step, host+GPU %.3f ms / step" % ((t1 - t0) / a.steps * 1e3, (t2 - t0) / a.steps * 1e3))
    if a.rccl:
        x = torch.empty((rt.n_g, args.k), device=dev)
        h, g = timed(lambda: tr._x(qg, rt.n_g, rt.recv_g, rt.send_g, None), a.steps)
        print("  a2a of %d rows through RCCL (self): host %.3f, total %.3f" % (rt.n_g, h, g))
        dist.destroy_process_group()


main()
