"""Time dr_score_topk_f32 on the C4 evaluator shape (development probe; not part of the product path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeprec_amd import kernels as K

def main():
    nu = int(os.environ.get("NU", 16384)); ni = int(os.environ.get("NI", 1000000)); k = int(os.environ.get("KD", 128))
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    U = torch.randn(nu, k, device=dev, generator=g) * 0.1
    I = torch.randn(ni, k, device=dev, generator=g) * 0.1
    tp = torch.arange(nu + 1, device=dev, dtype=torch.int32)
    ti = torch.randint(0, ni, (nu,), device=dev, dtype=torch.int32)
    # no train mask: one held-out item per user
    def run():
        return K.score_topk(U, I, 10, targets=(tp, ti))
    run(); torch.cuda.synchronize()
    reps = int(os.environ.get("REPS", 3))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("dbg=%s nu=%d ni=%d k=%d: %.2f ms  %.1f TFLOP/s" % (os.environ.get("DR_SCORE_DBG", "0"), nu, ni, k, ms, 2.0 * nu * ni * k / ms / 1e9))

def prof():
    dev = torch.device("cuda:0")
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    os.environ["DR_SCORE_PROF"] = str(buf.data_ptr())
    main()
    torch.cuda.synchronize()
    b = buf.cpu().numpy()[:80].reshape(8, 10)
    names = ["loop", "store", "prefetch", "sel_tile", "barSel", "mfma", "sS", "barCmp", "-", "-"]
    ntile = 3907
    for w in range(8):
        print("wave", w, " ".join("%s=%d" % (n, b[w][i] // ntile) for i, n in enumerate(names) if n != "-"), "total", b[w].sum() // ntile)

if os.environ.get("PROF"):
    prof()
else:
    main()
