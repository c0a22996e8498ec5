"""Run the C4 evaluator shape (16 384 users x 1 M items x 128, K=10, 100 masked train items and one held-out item
per user) a few times -- the command profiled by the rocprofv3 --pmc pass whose MFMA-utilisation counters are
kept in profiles/r01_pmc_score_kernel.csv."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K


def main():
    nu, ni, k = 16384, 1_000_000, 128
    dev = torch.device("cuda:0")
    U = torch.empty((nu, k), device=dev)
    I = torch.empty((ni, k), device=dev)
    K.init_normal_(U, 1, 0.1)
    K.init_normal_(I, 2, 0.1)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    tr = torch.sort(torch.randint(0, ni, (nu, 100), device=dev, generator=g), 1)[0].to(torch.int32).reshape(-1).contiguous()
    tp = (torch.arange(nu + 1, device=dev) * 100).to(torch.int32)
    tg = torch.randint(0, ni, (nu,), device=dev, generator=g).to(torch.int32)
    gp = torch.arange(nu + 1, device=dev).to(torch.int32)
    for _ in range(int(os.environ.get("REPS", "3"))):
        K.score_topk(U, I, 10, train=(tp, tr), targets=(gp, tg))      # score_kernel_t1: top-10 list + ranks
        K.score_rank(U, I, (gp, tg), train=(tp, tr))                  # score_rank_kernel: ranks only (the evaluators' call)
    torch.cuda.synchronize()
    print("ok")


main()
