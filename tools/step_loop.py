"""Run a few fused BPR steps of the bench workload (C4) -- the command profiled by the rocprofv3 --pmc passes
whose per-kernel HBM traffic is summarised in profiles/r03_pmc_traffic.json (tools/pmc_traffic.py)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deeprec_amd import kernels as K


def main():
    steps = int(os.environ.get("STEPS", "12"))
    sys.argv = ["bench.py"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    batches = bench.make_batches(args, 0, dev, 4)
    P = torch.empty((args.users, args.k), device=dev)
    Q = torch.empty((args.items, args.k), device=dev)
    K.init_normal_(P, 1, 0.01)
    K.init_normal_(Q, 2, 0.01)
    sP = [torch.zeros_like(P), torch.zeros_like(P)]
    sQ = [torch.zeros_like(Q), torch.zeros_like(Q)]
    st = K.AdamState(1e-3, 0.9, 0.999, dev)
    opt = K.opt_params(K.OPT_ADAM, lr=0.05, beta1=0.9, beta2=0.999, eps=1e-7, reg=0.1, reg_mode=K.REG_TOUCHED,
                       alpha_dev=st.alpha)
    ws = K.StepWorkspace(args.batch, args.k, args.users, args.items, K.REG_TOUCHED, dev)
    for t in range(steps):              # the planned step, as bench.py runs it (plan, then the fused forward + updates)
        u, i, j = batches[t % len(batches)]
        st.tick()
        K.bpr_plan(args.users, args.items, args.k, u, i, j, ws)
        K.bpr_step(P, Q, sP, sQ, u, i, j, opt, ws, planned=True)
    torch.cuda.synchronize()
    print("ok", steps)


main()
