#!/usr/bin/env python
"""One model class at the C5 shape (L = 50, T = 3, k = 128, B = 4096), a few steps: run under
`rocprofv3 --kernel-trace --stats --output-format csv` to see where a step goes (dev tool).

    python tools/c5_probe.py Caser|AttRec|PRME [steps]
"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Caser"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.int32)
    nu, ni, L, T_, B, d = 1_000_000, 100_000, 50, 3, 4096, 128
    with contextlib.redirect_stdout(io.StringIO()):
        if name == "Caser":
            from deeprec_amd.models.seq_rec.Caser import Caser
            m = Caser(None, nu, ni, 0.001, 0.01, 1, B)
            m.build_network(L, T_, num_factor=d, num_neg=1)
        elif name == "AttRec":
            from deeprec_amd.models.seq_rec.AttRec import AttRec
            m = AttRec(None, nu, ni, 0.05, 0.01, 1, B)
            m.build_network(L, T_, num_factor=d, num_neg=1)
        else:
            from deeprec_amd.models.seq_rec.PRME import PRME
            L = 1
            m = PRME(None, nu, ni, 0.001, 0.01, 1, B)
            m.build_network(1, T_, num_factor=d, num_neg=1)
    bs = [(t(rng.integers(0, nu, B)), t(rng.integers(1, ni, (B, L))), t(rng.integers(1, ni, (B, T_))),
           t(rng.integers(1, ni, (B, T_)))) for _ in range(4)]
    for x in range(3):
        m.train_op(*bs[x % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in range(steps):
        m.train_op(*bs[x % 4])
    torch.cuda.synchronize()
    print("%s C5: %.3f ms / step" % (name, (time.perf_counter() - t0) / steps * 1e3))


main()
