#!/usr/bin/env python
"""Timing of Caser's horizontal-convolution kernels at the C5 shape (B = 4096, L = 50, d = 128, C = 16); run it under
rocprofv3 --kernel-trace --stats for the per-kernel split (tools/build_var.sh builds A/B variants of the library,
selected with DEEPREC_AMD_LIB)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeprec_amd import kernels as K

dev = torch.device("cuda:0")
B, L, d, C = 4096, 50, 128, 16
g = torch.Generator(device=dev)
g.manual_seed(1)
e = torch.randn((B, L, d), device=dev, generator=g) * 0.01
W = torch.randn((K.hconv_rows(C, L + 1), d), device=dev, generator=g)
bias = torch.zeros(C * L, device=dev)
R = torch.randn((B, C * L), device=dev, generator=g)
out, arg = K.caser_hconv_forward(e, W, bias, C)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("active winners: %.3f" % float((arg >= 0).float().mean()))
print("backward (d e + d W + sums): %.3f ms" % timed(lambda: K.caser_hconv_backward(e, W, arg, R, C)))
print("forward: %.3f ms" % timed(lambda: K.caser_hconv_forward(e, W, bias, C)))
