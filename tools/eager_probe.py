#!/usr/bin/env python
"""Host enqueue time vs wall time of the eager pipelined fused step (dev tool)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeprec_amd import kernels as K

sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda:0")
batches = bench.make_batches(args, 0, dev, 8)
P = torch.empty((args.users, args.k), device=dev)
Q = torch.empty((args.items, args.k), device=dev)
K.init_normal_(P, 1, 0.01)
K.init_normal_(Q, 2, 0.01)
sP = [torch.zeros_like(P), torch.zeros_like(P)]
sQ = [torch.zeros_like(Q), torch.zeros_like(Q)]
st = K.AdamState(1e-3, 0.9, 0.999, dev)
opt = K.opt_params(K.OPT_ADAM, lr=0.05, beta1=0.9, beta2=0.999, eps=1e-7, reg=0.1, reg_mode=K.REG_TOUCHED, alpha_dev=st.alpha)
nb = len(batches)
wss = [K.StepWorkspace(args.batch, args.k, args.users, args.items, K.REG_TOUCHED, dev) for _ in range(nb)]
side = torch.cuda.Stream(device=dev)
mode = os.environ.get("MODE", "plan_first")
sides = [torch.cuda.Stream(device=dev) for _ in range(3)]
ev_end = [None] * nb
ev_plan = [None] * nb


def step(b):
    u, i, j = batches[b]
    st.tick()
    cur = torch.cuda.current_stream()
    if mode == "plan_first":
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            K.bpr_plan(args.users, args.items, args.k, *batches[(b + 1) % nb], wss[(b + 1) % nb])
        K.bpr_step(P, Q, sP, sQ, u, i, j, opt, wss[b], planned=True)
        cur.wait_stream(side)
    elif mode == "step_first":
        ev = torch.cuda.Event()
        ev.record(cur)
        K.bpr_step(P, Q, sP, sQ, u, i, j, opt, wss[b], planned=True)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            K.bpr_plan(args.users, args.items, args.k, *batches[(b + 1) % nb], wss[(b + 1) % nb])
        cur.wait_stream(side)
    elif mode == "depth2":         # plan of batch b+2 while batch b runs; nobody waits on anything recent
        if ev_end[(b - 1) % nb] is not None:
            side.wait_event(ev_end[(b - 1) % nb])
        with torch.cuda.stream(side):
            K.bpr_plan(args.users, args.items, args.k, *batches[(b + 2) % nb], wss[(b + 2) % nb])
            e = torch.cuda.Event()
            e.record(side)
            ev_plan[(b + 2) % nb] = e
        if ev_plan[b] is not None:
            cur.wait_event(ev_plan[b])
        K.bpr_step(P, Q, sP, sQ, u, i, j, opt, wss[b], planned=True)
        e = torch.cuda.Event()
        e.record(cur)
        ev_end[b] = e
    elif mode in ("depth2x2", "depth3x3"):   # plans D batches ahead, D side streams in rotation: D plans in flight
        D = 2 if mode == "depth2x2" else 3
        sd = sides[b % D]
        if ev_end[(b - 1) % nb] is not None:
            sd.wait_event(ev_end[(b - 1) % nb])
        with torch.cuda.stream(sd):
            K.bpr_plan(args.users, args.items, args.k, *batches[(b + D) % nb], wss[(b + D) % nb])
            e = torch.cuda.Event()
            e.record(sd)
            ev_plan[(b + D) % nb] = e
        if ev_plan[b] is not None:
            cur.wait_event(ev_plan[b])
        K.bpr_step(P, Q, sP, sQ, u, i, j, opt, wss[b], planned=True)
        e = torch.cuda.Event()
        e.record(cur)
        ev_end[b] = e
    elif mode == "step_only":      # plan of the next batch skipped: WRONG results, timing of the main path only
        K.bpr_step(P, Q, sP, sQ, u, i, j, opt, wss[b], planned=True)
    elif mode == "plan_only":
        K.bpr_plan(args.users, args.items, args.k, *batches[(b + 1) % nb], wss[(b + 1) % nb])


for b in range(nb):
    K.bpr_plan(args.users, args.items, args.k, *batches[b], wss[b])
torch.cuda.synchronize()
if os.environ.get("PRIO"):
    hi = torch.cuda.Stream(device=dev, priority=-1)
    torch.cuda.set_stream(hi)
    print("main stream priority", hi.priority, "side", side.priority)
for t in range(16):
    step(t % nb)
torch.cuda.synchronize()
n = 200
h = 0.0
t0 = time.perf_counter()
for t in range(n):
    a = time.perf_counter()
    step(t % nb)
    h += time.perf_counter() - a
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s: host enqueue %.1f us/step, loop wall %.1f us/step, drained %.1f us/step" %
      (mode, h / n * 1e6, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))

# ---- where the host time of one step goes ----
import ctypes as C
from deeprec_amd import _abi
lib = _abi.lib()
ws = wss[0]
u, i, j = batches[0]
argv = (K._ptr(P), K._ptr(sP[0]), K._ptr(sP[1]), P.shape[0], K._ptr(Q), K._ptr(sQ[0]), K._ptr(sQ[1]), Q.shape[0], P.shape[1],
        K._ptr(u), K._ptr(i), K._ptr(j), args.batch, C.byref(opt), K._ptr(ws.loss), K._ptr(ws.buf), ws.buf.numel(), K._stream())


def host(fn, n=200):
    fn()
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(n):
        a = time.perf_counter()
        fn()
        t += time.perf_counter() - a
        torch.cuda.synchronize()
    return t / n * 1e6


print("raw dr_bpr_step_planned_f32 call: %.1f us host" % host(lambda: lib.dr_bpr_step_planned_f32(*argv)))
print("K.bpr_step wrapper:               %.1f us host" % host(lambda: K.bpr_step(P, Q, sP, sQ, u, i, j, opt, ws, planned=True)))
print("tick:                             %.1f us host" % host(st.tick))
print("K.bpr_plan:                       %.1f us host" % host(lambda: K.bpr_plan(args.users, args.items, args.k, u, i, j, ws)))
print("dr_sum_f32:                       %.1f us host" % host(lambda: K.sum_f32(ws.loss)))
