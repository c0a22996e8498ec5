"""Interaction + loss kernels of the sequence models (dr_seq_dist_loss_f32, dr_seq_dot_loss_f32)
against a FLOAT64 autograd evaluation of the reference's literal expressions:
  AttRec.py:190-232 + :97  (two distances, hinge with margin 0.5),
  PRME.py:104-131 + :73    (two distances, BPR on distances),
  Caser.py:79-80,184-192   (dot + bias, mean BCE with 1e-10 inside the logs).
TensorFlow is not runnable here, so "the reference" is this restatement (parity unpinned).
Tolerance: 1e-5 of each tensor's scale, stated per assertion; duplicate ids inside a sample, ids shared between
samples, every row-shape class (k a multiple of 4, of 2, odd) and batches that do not fill a workgroup are covered.
The fused and the unfused train_op of each class must also produce the same step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(name, got, ref, tol=1e-5):
    got = got.detach().cpu().double().numpy()
    ref = ref.detach().cpu().double().numpy()
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(got - ref).max() / scale
    assert err <= tol, (name, err)


def _scatter(ids, rows, n):
    """Sum per-(sample, slot) gradient rows into a dense [n, k] table gradient (float64)."""
    out = torch.zeros((n, rows.shape[-1]), dtype=torch.float64)
    out.index_add_(0, ids.reshape(-1).long().cpu(), rows.reshape(-1, rows.shape[-1]).double().cpu())
    return out


@pytest.mark.parametrize("kind", ["hinge", "bpr"])
@pytest.mark.parametrize("B,T,k,nu,ni", [(1, 1, 4, 3, 5), (37, 3, 100, 50, 40), (300, 1, 64, 1000, 2000),
                                         (65, 5, 6, 20, 9), (19, 32, 7, 11, 300), (130, 2, 256, 64, 64)])
def test_two_distance_loss_and_gradients(dev, kind, B, T, k, nu, ni):
    from deeprec_amd import kernels as K
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + k)
    sd = 0.5 if kind == "hinge" else 0.3        # distances around the margin so that both hinge branches occur
    P = torch.randn((nu, k), generator=g) * sd
    X = torch.randn((ni, k), generator=g) * sd
    Q = torch.randn((ni, k), generator=g) * sd
    o = torch.randn((B, k), generator=g) * sd
    users = torch.randint(0, nu, (B,), generator=g, dtype=torch.int32)
    tgt = torch.randint(0, ni, (B, T), generator=g, dtype=torch.int32)
    neg = torch.randint(0, ni, (B, T), generator=g, dtype=torch.int32)
    if T > 1:
        tgt[:, 1] = tgt[:, 0]                    # a duplicate inside the sample
        neg[0, 0] = tgt[0, 0]                    # target == negative: zero hinge argument + margin
    alpha = 0.2
    # float64 autograd of the literal expressions
    P64, X64, Q64, o64 = (t.double().requires_grad_(True) for t in (P, X, Q, o))

    def dist(items):
        wi, wi2 = X64[items.long()], Q64[items.long()]
        u = P64[users.long()][:, None, :]
        return alpha * ((wi - u) ** 2).sum(2) + (1 - alpha) * ((o64[:, None, :] - wi2) ** 2).sum(2)
    tp, tn = dist(tgt), dist(neg)
    if kind == "hinge":
        per = torch.clamp(tp - tn + 0.5, min=0).sum(1)
    else:
        per = -torch.log(torch.sigmoid(-tp + tn)).sum(1)
    per.sum().backward()

    d = lambda t: t.to(dev)
    loss_b, gX, gQ, gp, go = K.seq_dist_loss(d(P), d(X), d(Q), d(users), d(o), d(tgt), d(neg), alpha, 1 - alpha,
                                             K.SEQ_LOSS_HINGE if kind == "hinge" else K.SEQ_LOSS_BPR, 0.5)
    torch.cuda.synchronize()
    ids = torch.cat((tgt, neg), 1)
    _close("loss_b", loss_b, per)
    _close("gX", _scatter(ids, gX, ni), X64.grad)
    _close("gQ", _scatter(ids, gQ, ni), Q64.grad)
    _close("gp", _scatter(users, gp, nu), P64.grad)
    _close("go", go, o64.grad)


def test_two_distance_hinge_boundary_and_invalid_ids(dev):
    """tf.maximum(x, 0) passes the gradient where x >= 0 ([TF-ext], math_grad.py _MaximumGrad): an argument of exactly
    zero (target == negative with margin 0) keeps coefficient 1; a pair with an out-of-range id contributes nothing."""
    from deeprec_amd import kernels as K
    k, ni = 8, 6
    g = torch.Generator().manual_seed(5)
    P, X, Q = (torch.randn((4, k), generator=g), torch.randn((ni, k), generator=g), torch.randn((ni, k), generator=g))
    o = torch.randn((2, k), generator=g)
    users = torch.tensor([1, 2], dtype=torch.int32)
    tgt = torch.tensor([[3, 2], [1, ni]], dtype=torch.int32)         # sample 1, pair 1: target out of range
    neg = torch.tensor([[3, 4], [5, 0]], dtype=torch.int32)          # sample 0, pair 0: same item both sides
    d = lambda t: t.to(dev)
    loss_b, gX, gQ, gp, go = K.seq_dist_loss(d(P), d(X), d(Q), d(users), d(o), d(tgt), d(neg), 0.2, 0.8,
                                             K.SEQ_LOSS_HINGE, 0.0)
    torch.cuda.synchronize()
    gX = gX.cpu()
    # pair (0, 0): argument exactly 0 -> loss 0, coefficient 1: the rows are +g and -g of the same item
    assert torch.equal(gX[0, 0], -gX[0, 2]) and gX[0, 0].abs().max() > 0
    # pair (1, 1) is void: both of its rows are zero
    assert gX[1, 1].abs().max() == 0 and gX[1, 3].abs().max() == 0
    assert torch.isfinite(loss_b).all() and torch.isfinite(gp).all() and torch.isfinite(go).all()


@pytest.mark.parametrize("B,T,Tn,D,ni", [(1, 1, 1, 4, 3), (50, 3, 3, 100, 70), (257, 3, 9, 64, 1000), (33, 2, 5, 6, 12),
                                         (21, 30, 34, 10, 200), (40, 1, 2, 512, 50)])
def test_dot_bce_loss_and_gradients(dev, B, T, Tn, D, ni):
    from deeprec_amd import kernels as K
    g = torch.Generator().manual_seed(B + 7 * T + 13 * D)
    W = torch.randn((ni, D), generator=g) * 0.3
    bias = torch.randn((ni,), generator=g) * 0.1
    x = torch.randn((B, D), generator=g) * 0.3
    tgt = torch.randint(0, ni, (B, T), generator=g, dtype=torch.int32)
    neg = torch.randint(0, ni, (B, Tn), generator=g, dtype=torch.int32)
    neg[:, 0] = tgt[:, 0]                                            # the same item as a target and as a negative
    W64, b64, x64 = (t.double().requires_grad_(True) for t in (W, bias, x))

    def score(items):                                                # Caser.py:184-192: b_i + <W_i, x> via a batched matmul
        return b64[items.long()] + torch.einsum("btd,bd->bt", W64[items.long()], x64)
    sp, sn = score(tgt), score(neg)
    pos_l, neg_l = -torch.log(torch.sigmoid(sp) + 1e-10), -torch.log(1 - torch.sigmoid(sn) + 1e-10)
    (pos_l.mean() + neg_l.mean()).backward()
    per = pos_l.sum(1) / (B * T) + neg_l.sum(1) / (B * Tn)

    d = lambda t: t.to(dev)
    for scale in (1.0, 0.25):
        loss_b, gW, gb, gx = K.seq_dot_loss(d(W), d(bias), d(x), d(tgt), d(neg), scale)
        torch.cuda.synchronize()
        ids = torch.cat((tgt, neg), 1)
        _close("loss_b", loss_b, per * scale)
        _close("gW", _scatter(ids, gW, ni), W64.grad * scale)
        _close("gb", _scatter(ids, gb[..., None], ni)[:, 0], b64.grad * scale)
        _close("gx", gx, x64.grad * scale)


def test_size_limits_are_errors(dev):
    from deeprec_amd import kernels as K
    z = lambda *s: torch.zeros(s, device=dev)
    zi = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
    with pytest.raises(Exception, match="bad sizes"):
        K.seq_dist_loss(z(2, 4), z(3, 4), z(3, 4), zi(1), z(1, 4), zi(1, 33), zi(1, 33), 0.2, 0.8, 0)
    with pytest.raises(Exception, match="bad sizes"):
        K.seq_dot_loss(z(3, 4), z(3), z(1, 4), zi(1, 40), zi(1, 25))
    with pytest.raises(Exception, match="loss_kind"):
        K.seq_dist_loss(z(2, 4), z(3, 4), z(3, 4), zi(1), z(1, 4), zi(1, 1), zi(1, 1), 0.2, 0.8, 2)


def _fused_vs_unfused(make, step_args, names):
    a, b = make(), make()
    b.unfused = True
    la, lb = a.train_op(*step_args), b.train_op(*step_args)
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    for n in names:
        wa, wb = getattr(a, n), getattr(b, n)
        wa, wb = (wa.w if hasattr(wa, "w") else wa), (wb.w if hasattr(wb, "w") else wb)
        err = float((wa - wb).abs().max()) / max(float(wb.abs().max()), 1e-30)
        assert err <= 1e-5, (n, err)


def test_fused_and_unfused_train_ops_agree(dev):
    """The same step through the fused kernel and through library elementwise ops under autograd (`unfused`), from the
    same initial tables: weights after the step within 1e-5 of their scale."""
    import contextlib
    import io
    from deeprec_amd.models.seq_rec.PRME import PRME
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    rng = np.random.default_rng(3)
    nu, ni, B, L, T = 40, 60, 64, 5, 3
    users = torch.from_numpy(rng.integers(0, nu, B).astype(np.int32)).to(dev)
    seq = torch.from_numpy(rng.integers(0, ni, (B, L)).astype(np.int32)).to(dev)
    tgt = torch.from_numpy(rng.integers(0, ni, (B, T)).astype(np.int32)).to(dev)
    neg = torch.from_numpy(rng.integers(0, ni, (B, T)).astype(np.int32)).to(dev)

    def quiet(fn):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn()

    def attrec():
        torch.manual_seed(11), np.random.seed(11)
        return quiet(lambda: AttRec(None, nu, ni, device=dev).build_network(L, T, num_factor=32, num_neg=1))
    _fused_vs_unfused(attrec, (users, seq, tgt, neg), ("P", "X", "Q", "V"))

    def prme():
        torch.manual_seed(12), np.random.seed(12)
        m = quiet(lambda: PRME(None, nu, ni, device=dev).build_network(1, T, num_factor=20, num_neg=1))
        for t in (m.P, m.X, m.Q):
            t.w = torch.randn_like(t.w) * 0.3
        return m
    _fused_vs_unfused(prme, (users, seq[:, :1].contiguous(), tgt, neg), ("P", "X", "Q"))

    def caser():
        torch.manual_seed(13), np.random.seed(13)
        return quiet(lambda: Caser(None, nu, ni, device=dev).build_network(L, T, num_factor=16, num_neg=2))
    neg2 = torch.from_numpy(rng.integers(0, ni, (B, 2 * T)).astype(np.int32)).to(dev)
    _fused_vs_unfused(caser, (users, seq, tgt, neg2), ("P", "Q", "W", "b"))
