#!/usr/bin/env python
"""Time dr_caser_hconv_direct_f32 alone at the C5 shape for several batch sizes (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import _abi
if os.environ.get('DR_PROBE_LIB'):
    _abi._LIB = _abi.load(os.environ['DR_PROBE_LIB'])
from deeprec_amd import kernels as K
dev = torch.device("cuda:0")
L, d, C = 50, 128, 16
W = torch.randn(K.hconv_rows(C, L + 1), d, device=dev) * 0.05
b = torch.zeros(C * L, device=dev)
for B in (4096,):
    e = torch.randn(B, L, d, device=dev) * 0.1
    K.caser_hconv_forward(e, W, b, C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        K.caser_hconv_forward(e, W, b, C)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    fl = 2.0 * B * C * d * L * (L + 1) * (L + 2) / 6
    print("B=%d: %.3f ms, %.1f TFLOP/s" % (B, ms, fl / ms / 1e9))
