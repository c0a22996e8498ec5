#!/usr/bin/env python
"""Kernel-level timing of the dense-tower kernels for a few shapes (run under rocprofv3 --kernel-trace; dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K
dev = torch.device("cuda:0")
for B, dims in ((256, [16, 16, 16, 16, 16, 16]), (256, [128, 128, 80, 40, 20, 10]), (256, [128, 128]), (16, [128, 128, 80, 40, 20, 10])):
    x = torch.randn(B, dims[0], device=dev)
    Ws = [torch.randn(dims[l], dims[l + 1], device=dev) * 0.1 for l in range(len(dims) - 1)]
    bs = [torch.zeros(dims[l + 1], device=dev) for l in range(len(dims) - 1)]
    kinds = [1] * (len(dims) - 1)
    g = torch.randn(B, dims[-1], device=dev)
    for _ in range(20):
        acts = K.mlp_forward(x, Ws, bs, kinds)
        K.mlp_backward(x, Ws, kinds, acts, g)
    torch.cuda.synchronize()
    # marker kernel between shapes
    torch.zeros(1, device=dev).add_(1)
torch.cuda.synchronize()
