#!/usr/bin/env python
"""Wall time of dr_attn_pool_forward_f32 / _backward_f32 at the C5 shape (B = 4096, L = 50, d = 128 or ATT_D) (dev tool)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeprec_amd import kernels as K

dev = torch.device("cuda:0")
B, L, d = 4096, 50, int(os.environ.get("ATT_D", "128"))
pre, v, g = torch.randn(B, L, d, device=dev), torch.randn(B, L, d, device=dev), torch.randn(B, d, device=dev)
out, s = K.attn_pool_forward(pre, v)
for name, fn in (("fwd", lambda: K.attn_pool_forward(pre, v)), ("bwd", lambda: K.attn_pool_backward(pre, v, s, g))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print("d=%d %s %.1f us" % (d, name, (time.perf_counter() - t0) / 20 * 1e6))
