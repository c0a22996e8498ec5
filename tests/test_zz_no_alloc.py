"""dr_bpr_step_static_f32 (the entry point that allocates nothing; include/deeprec_amd.h) against dr_bpr_step_f32:
tables, optimizer slots and loss bit for bit, on two calls in a row, the second with a smaller batch (the static
scratch keeps the first call's contents).  Collected last on purpose (file name)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(K, rng, nu, ni, k, B, kind, dev):
    P = (rng.standard_normal((nu, k)) * 0.1).astype(np.float32)
    Q = (rng.standard_normal((ni, k)) * 0.1).astype(np.float32)
    u = rng.integers(0, nu, B).astype(np.int32)
    i = (rng.zipf(1.3, B) % ni).astype(np.int32)       # duplicate-heavy: long segments too
    j = rng.integers(0, ni, B).astype(np.int32)
    if B > 4:
        u[3] = -1                                       # an invalid sample contributes nothing
        j[1] = ni + 5
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(P), t(Q), t(u), t(i), t(j)


def _slots(K, kind, like):
    if kind == K.OPT_SGD:
        return [None, None]
    if kind == K.OPT_ADAGRAD:
        return [torch.full_like(like, 0.1), None]
    return [torch.zeros_like(like), torch.zeros_like(like)]


@pytest.mark.parametrize("kind", ["adam", "adagrad", "sgd"])
def test_static_step_equals_workspace_step(kind):
    from deeprec_amd import kernels as K
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    kk = {"adam": K.OPT_ADAM, "adagrad": K.OPT_ADAGRAD, "sgd": K.OPT_SGD}[kind]
    opt = K.opt_params(kk, lr=0.05, alpha=1e-3, eps=1e-7, reg=0.01, reg_mode=K.REG_TOUCHED)
    nu, ni = 943, 1682
    loss_s = torch.zeros(1, dtype=torch.float32, device=dev)
    # first call: the maximum batch at k = 128; second: a smaller batch and another k, same scratch
    for (k, B) in ((128, 1024), (30, 200), (128, 1), (64, 0)):
        P, Q, u, i, j = _case(K, rng, nu, ni, k, B, kk, dev)
        Pa, Qa, Pb, Qb = P.clone(), Q.clone(), P.clone(), Q.clone()
        sPa, sQa, sPb, sQb = _slots(K, kk, Pa), _slots(K, kk, Qa), _slots(K, kk, Pb), _slots(K, kk, Qb)
        for _ in range(2):                                                 # two steps each, slots carried over
            la = K.bpr_step(Pa, Qa, sPa, sQa, u, i, j, opt).clone()
            lb = K.bpr_step_static(Pb, Qb, sPb, sQb, u, i, j, opt, loss_s).clone()
            torch.cuda.synchronize()
            assert torch.equal(la, lb), (k, B, la, lb)
            assert torch.equal(Pa, Pb) and torch.equal(Qa, Qb), (k, B)
            for x, y in zip(sPa + sQa, sPb + sQb):
                assert (x is None and y is None) or torch.equal(x, y)


def test_static_step_rejects_sizes_beyond_the_maximum():
    from deeprec_amd import kernels as K
    from deeprec_amd._abi import DeepRecError
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    opt = K.opt_params(K.OPT_SGD, lr=0.1, reg_mode=K.REG_TOUCHED)
    P, Q, u, i, j = _case(K, rng, 100, 100, 16, K.STATIC_MAX_B + 1, K.OPT_SGD, dev)
    with pytest.raises(DeepRecError, match="exceeds the static maximum"):
        K.bpr_step_static(P, Q, [None, None], [None, None], u, i, j, opt, loss)
    P, Q, u, i, j = _case(K, rng, 100, 100, 256, 8, K.OPT_SGD, dev)
    with pytest.raises(DeepRecError, match="exceeds the static maximum"):
        K.bpr_step_static(P, Q, [None, None], [None, None], u, i, j, opt, loss)
    P, Q, u, i, j = _case(K, rng, 100, 100, 16, 8, K.OPT_SGD, dev)
    with pytest.raises(DeepRecError, match="DR_REG_DENSE"):
        K.bpr_step_static(P, Q, [None, None], [None, None], u, i, j,
                          K.opt_params(K.OPT_SGD, lr=0.1, reg_mode=K.REG_DENSE), loss)


_FRESH = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, %r)
from deeprec_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
nu, ni, k, B = 943, 1682, 128, 1024
t = lambda a: torch.from_numpy(a).to(dev)
P = t((rng.standard_normal((nu, k)) * 0.1).astype(np.float32)); Q = t((rng.standard_normal((ni, k)) * 0.1).astype(np.float32))
u = t(rng.integers(0, nu, B).astype(np.int32)); i = t((rng.zipf(1.3, B) %% ni).astype(np.int32)); j = t(rng.integers(0, ni, B).astype(np.int32))
sP = [torch.zeros_like(P), torch.zeros_like(P)]; sQ = [torch.zeros_like(Q), torch.zeros_like(Q)]
loss = torch.zeros(1, dtype=torch.float32, device=dev)
opt = K.opt_params(K.OPT_ADAM, lr=0.05, alpha=1e-3, eps=1e-7, reg=0.01, reg_mode=K.REG_TOUCHED)
K.init(dev)                                            # the one call that creates anything (dr_init)
for _ in range(2):                                     # the HIP runtime loads the kernels' code objects on first launch
    K.bpr_step_static(P, Q, sP, sQ, u, i, j, opt, loss)
torch.cuda.synchronize()
a0, f0 = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
for _ in range(50):
    K.bpr_step_static(P, Q, sP, sQ, u, i, j, opt, loss)
torch.cuda.synchronize()
a1, f1 = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
assert a0 == a1, (a0, a1)                              # torch's allocator: nothing requested
assert f0 == f1, (f0, f1)                              # the device (hipMemGetInfo): nothing allocated behind torch's back
# the same two steps captured into a graph and replayed == two eager steps
Pg, Qg = P.clone(), Q.clone(); sPg = [x.clone() for x in sP]; sQg = [x.clone() for x in sQ]
Pe, Qe = P.clone(), Q.clone(); sPe = [x.clone() for x in sP]; sQe = [x.clone() for x in sQ]
lg = torch.zeros(1, dtype=torch.float32, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    K.bpr_step_static(Pg, Qg, sPg, sQg, u, i, j, opt, lg)
    K.bpr_step_static(Pg, Qg, sPg, sQg, u, i, j, opt, lg)
g.replay()
for _ in range(2):
    K.bpr_step_static(Pe, Qe, sPe, sQe, u, i, j, opt, loss)
torch.cuda.synchronize()
assert torch.equal(Pg, Pe) and torch.equal(Qg, Qe) and torch.equal(lg, loss)
for x, y in zip(sPg + sQg, sPe + sQe):
    assert torch.equal(x, y)
print("static-only process ok")
"""


def test_static_step_alone_in_a_fresh_process_allocates_nothing_and_captures():
    """What the entry point exists for, checked where nothing else has run: a fresh process that calls dr_init() and then
    ONLY dr_bpr_step_static_f32 -- torch's allocator and hipMemGetInfo see no change over 50 steps, and two steps
    captured into a graph replay to the eager result."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FRESH % root], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "static-only process ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
