#!/usr/bin/env python
"""200 graphed NeuMF steps at the C2 shape (for rocprofv3 --kernel-trace --stats; dev tool)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deeprec_amd.models.item_ranking.neumf import NeuMF
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
t = lambda a, dt=torch.int32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
nu, ni, B = 6040, 3706, 256
with contextlib.redirect_stdout(io.StringIO()):
    m = NeuMF(None, nu, ni)
    m.build_network()
bs = [(t(rng.integers(0, nu, B)), t(rng.zipf(1.2, B) % ni), t(rng.integers(0, 2, B), torch.float32)) for _ in range(8)]
for i in range(int(os.environ.get("STEPS", "200"))):
    m._step(*bs[i % 8])
torch.cuda.synchronize()
print("ok")
