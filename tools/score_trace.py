#!/usr/bin/env python
"""Phase trace of the leave-one-out scoring kernel (dev tool; needs the -DDR_ABL=9 build of tools/build_abl.sh):
per block, wave 0 stamps the cycle counter after barrier A, after barrier B, after the MFMA chain and after the
selection of its first tiles.  Prints phase durations and, for blocks that shared a CU, their relative phase."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deeprec_amd import _abi

_abi._LIB = _abi.load(os.environ["DR_PROBE_LIB"])
from deeprec_amd import kernels as K

DBG_T = 96
STRIDE = 4 + 8 * DBG_T


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    Nu, Ni, d = 16384, 1_000_000, 128
    U = torch.randn(Nu, d, device=dev, generator=g) * 0.1
    I = torch.randn(Ni, d, device=dev, generator=g) * 0.1
    tr = torch.sort(torch.randint(0, Ni, (Nu, 100), device=dev, generator=g), 1)[0].to(torch.int32).reshape(-1).contiguous()
    tp = (torch.arange(Nu + 1, device=dev) * 100).to(torch.int32)
    tg = torch.randint(0, Ni, (Nu,), device=dev, generator=g).to(torch.int32)
    gp = torch.arange(Nu + 1, device=dev).to(torch.int32)
    K.score_topk(U, I, 10, train=(tp, tr), targets=(gp, tg))
    torch.cuda.synchronize()
    buf = np.zeros(1024 * STRIDE, np.int64)
    fn = _abi._LIB.dr_debug_trace
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_int64]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    rows = buf.reshape(1024, STRIDE)
    hdr, ts = rows[:, :4], rows[:, 4:].reshape(1024, DBG_T, 8)
    np.save("gpurun_out/score_trace.npy", rows)
    # steady-state phase durations (tiles 40..90)
    sl = ts[:, 40:90]
    names = ["barrier A -> fold / mask done", "-> item tile stored (vmcnt wait + LDS writes)", "-> barrier B passed",
             "-> MFMA chain done", "-> masks patched / tau refreshed", "-> selection done", "-> next barrier A passed"]
    per = (ts[:, 41:91, 0] - sl[:, :, 0]).mean()
    print("cycles per tile: %.0f" % per)
    for x in range(6):
        print("  %-50s %7.0f" % (names[x], (sl[:, :, x + 1] - sl[:, :, x]).mean()))
    print("  %-50s %7.0f" % (names[6], (ts[:, 41:91, 0] - sl[:, :, 6]).mean()))
    # pair blocks by (xcc, hw_id cu / se / sh) among blocks of the first round (blockIdx.y < 2)
    key = {}
    for b in range(1024):
        hw, xcc, bx, by = (int(x) for x in hdr[b])
        if by >= 2:
            continue
        cu = (xcc & 0xF, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7)
        key.setdefault(cu, []).append(b)
    offs = []
    for cu, bl in key.items():
        if len(bl) == 2:
            x, y = bl
            # phase of y's barrier-B stamps relative to x's tile period, steady state
            px = ts[x, 40:90, 3]
            py = ts[y, 40:90, 3]
            period = (px[-1] - px[0]) / (len(px) - 1)
            offs.append((((py - px) % period) / period).mean())
    print("CUs with two traced blocks: %d; relative phase of their barrier B (fraction of a tile period): mean %.2f, "
          "histogram %s" % (len(offs), float(np.mean(offs)) if offs else -1,
                            np.histogram(offs, bins=10, range=(0, 1))[0].tolist() if offs else []))


main()
