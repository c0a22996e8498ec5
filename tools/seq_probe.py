#!/usr/bin/env python
"""AttRec / Caser C5 training steps (U = 1e6, I = 1e5, L = 50, T = 3, k = 128, B = 4096) for rocprofv3 (dev tool).

    SEQ_MODEL=AttRec|Caser [SEQ_D=128] [SEQ_EAGER=1] rocprofv3 --kernel-trace --stats -d gpurun_out/seq -- python tools/seq_probe.py
"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deeprec_amd.models.seq_rec.AttRec import AttRec
from deeprec_amd.models.seq_rec.Caser import Caser

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
nu, ni, L, T_, B, d = 1_000_000, 100_000, 50, 3, 4096, int(os.environ.get("SEQ_D", "128"))
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.int32)
with contextlib.redirect_stdout(io.StringIO()):
    name = os.environ.get("SEQ_MODEL", "AttRec")
    cls = AttRec if name == "AttRec" else Caser
    m = cls(None, nu, ni, 0.05 if cls is AttRec else 0.001, 0.01, 1, B)
    m.build_network(L, T_, num_factor=d, num_neg=1)
bs = [(t(rng.integers(0, nu, B)), t(rng.integers(1, ni, (B, L))), t(rng.integers(1, ni, (B, T_))),
       t(rng.integers(1, ni, (B, T_)))) for _ in range(4)]
step = m.train_op if (os.environ.get("SEQ_EAGER") or not hasattr(m, "_step")) else m._step
for i in range(3):
    step(*bs[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("SEQ_STEPS", "20"))
for i in range(n):
    step(*bs[i % 4])
torch.cuda.synchronize()
print("%s d=%d B=%d: %.3f ms per step" % (name, d, B, (time.perf_counter() - t0) / n * 1e3))
