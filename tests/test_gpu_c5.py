"""BASELINE config 5 AT ITS SHAPE (testSeqRec.py:63-70 flow: model -> build_network -> train step -> evaluator):
U = 10^6 users, I = 10^5 items, L = 50, T = 3, B = 4096, k = 128 (and 150 for AttRec), tables initialised on the
device by the library's counter-based generator.

One `train_op` is compared with a float64 restatement of the reference's graph on the rows the batch touches (plus 64
untouched rows per table, which see the dense L2 term only) -- the restatement works on COMPACT tables (the touched rows,
ids renumbered), so the CPU never holds a 10^6-row table; the optimizer arithmetic is oracle.ref_ops.opt_apply.  Two
models built from the same seeds are bit-identical after the step.  The evaluators run on 4096 users x all 10^5 items:
every user's rank is bounded by a float64 scoring of the same parameters (the means must lie between the bounds), and
128 users are checked exactly against the oracle's scores in the reference's own operation order
(oracle.ref_ops.numpy_order_scores for SeqRecMetrics.py:122-123, the canonical fmaf chain for the bilinear Caser score).
TensorFlow arithmetic stays "parity unpinned" (DESIGN.md section 1): the restatement is what is matched.
"""
import contextlib
import io
import math

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro
from test_gpu_steps import _L, _compare, _t

pytestmark = pytest.mark.gpu
f32, f64 = np.float32, np.float64
NU, NI, L, T, B = 1_000_000, 100_000, 50, 3, 4096
N_EVAL, N_EXACT = 4096, 64


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def _batch(rng, n_neg):
    u = rng.integers(0, NU, B).astype(np.int32)
    seq = (rng.zipf(1.2, (B, L)) % (NI - 1) + 1).astype(np.int32)          # popular items repeat across sequences
    tgt = rng.integers(1, NI, (B, T)).astype(np.int32)
    neg = rng.integers(1, NI, (B, T * n_neg)).astype(np.int32)
    return u, seq, tgt, neg


def _rows(rng, n_rows, *id_arrays, extra=64):
    """Ascending distinct rows the batch touches + `extra` random rows (dense L2 only)."""
    return np.unique(np.concatenate([a.reshape(-1).astype(np.int64) for a in id_arrays] + [rng.integers(0, n_rows, extra)]))


def _take(w, rows, dev):
    return w[torch.from_numpy(rows).to(dev)].cpu().numpy()


def _build(cls, seed, lr, reg, **kw):
    np.random.seed(seed)
    torch.manual_seed(seed)
    m = quiet(cls, None, NU, NI, lr, reg, 1, B)
    quiet(m.build_network, L, T, **kw)
    return m


# ---------------------------------------------------------------------------------------------------------------
# evaluator protocol data: N_EVAL users with 50 train items (masked) and one held-out item each
# ---------------------------------------------------------------------------------------------------------------
def _eval_protocol(m, rng):
    from deeprec_amd.models._common import NegItems
    users = np.sort(rng.choice(NU, N_EVAL, replace=False))
    train = np.stack([rng.permutation(np.unique(rng.integers(1, NI, 80)))[:51] for _ in users])
    held = train[:, 50].copy()
    train = np.sort(train[:, :50], 1)
    csr = sp.csr_matrix((np.ones(train.size, np.int8), (np.repeat(users, 50), train.reshape(-1))), shape=(NU, NI))
    m.neg_items = NegItems(csr, NI, first_item=1)
    m.test_data = {int(u): [int(h)] for u, h in zip(users, held)}
    m.test_sequences = rng.integers(1, NI, (NU, L)).astype(np.int32)
    m.test_users = [int(u) for u in users]
    return users, train, held


def _rank_bounds(score64, eps2, train, held):
    """Per user (lo, hi): the target's 0-based rank among the candidates (items 1..NI-1 minus the train items) lies in
    [lo, hi] for any fp32 evaluation of the scores in which two scores together are off by less than eps2[user]."""
    n = score64.shape[0]
    lo, hi = np.empty(n, np.int64), np.empty(n, np.int64)
    for r in range(n):
        s = score64[r].copy()
        st = s[held[r]]
        eps = eps2[r]
        s[0] = -np.inf
        s[train[r]] = -np.inf
        lo[r] = np.count_nonzero(s > st + eps)
        hi[r] = np.count_nonzero(s >= st - eps) - 1
    return lo, hi


def _seq_means(ranks, n_cand):
    from deeprec_amd.utils.evaluation._ranks import seq_user_metrics
    hr, mrr, auc = zip(*(seq_user_metrics([int(r)], 1, n_cand, 50) for r in ranks))
    return {"HR@50": float(np.mean(hr)), "MRR": float(np.mean(mrr)), "AUC": float(np.mean(auc))}


def _exact_ranks(score32, train, held):
    """Rank of the held-out item in the reference's ordering (stable descending sort of the candidates in ascending id
    order, SeqRecMetrics.py:141-155): candidates scoring higher, plus equal-scoring candidates with a smaller id."""
    out = []
    for r in range(score32.shape[0]):
        s = score32[r].astype(f64)
        st = s[held[r]]
        ok = np.ones(NI, bool)
        ok[0] = False
        ok[train[r]] = False
        out.append(int(np.count_nonzero(ok & (s > st)) + np.count_nonzero(ok[:held[r]] & (s[:held[r]] == st))))
    return out


def _check_evaluator(m, evaluate, score64_fn, score32_fn, with_auc, rng):
    users, train, held = _eval_protocol(m, rng)
    res = quiet(evaluate, m)
    # (a) every user: float64 bounds on the rank -> bounds on the means (all three are monotone in the rank)
    lo, hi = [], []
    for c in range(0, N_EVAL, 512):
        a, b = _rank_bounds(*score64_fn(users[c:c + 512]), train[c:c + 512], held[c:c + 512])
        lo.append(a), hi.append(b)
    lo, hi = np.concatenate(lo), np.concatenate(hi)
    assert np.all(lo <= hi)
    best, worst = _seq_means(lo, NI - 51), _seq_means(hi, NI - 51)
    for k in res:
        assert worst[k] - 1e-12 <= res[k] <= best[k] + 1e-12, (k, worst[k], res[k], best[k])
    assert ("AUC" in res) == with_auc
    # (b) N_EXACT users: exactly the oracle's ranks
    sub = np.arange(0, N_EVAL, N_EVAL // N_EXACT)
    m.test_users = [int(u) for u in users[sub]]
    res_sub = quiet(evaluate, m)
    ranks = _exact_ranks(score32_fn(users[sub]), train[sub], held[sub])
    assert np.all((lo[sub] <= ranks) & (ranks <= hi[sub]))
    want = _seq_means(ranks, NI - 51)
    for k in res_sub:
        assert abs(res_sub[k] - want[k]) <= 1e-12, (k, res_sub[k], want[k])


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [128, 150])
def test_attrec_c5_step_and_evaluate1(dev, d):
    """AttRec.py:97-110,132-232 + SeqRecMetrics.py:91-177 at U = 10^6, I = 10^5, L = 50, T = 3, B = 4096."""
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.utils.evaluation.SeqRecMetrics import evaluate1
    rng = np.random.default_rng(50 + d)
    u, seq, tgt, neg = _batch(rng, 1)
    ru, ri, rv = _rows(rng, NU, u), _rows(rng, NI, seq, tgt, neg), _rows(rng, NU)
    m = _build(AttRec, 11, 0.05, 0.01, num_factor=d)
    assert getattr(m, "_step", None) is None and m.V.lazy          # the eager C5 step, V's dense pass recorded
    init = {"P": _take(m.P.w, ru, dev), "V": _take(m.V.w, rv, dev), "Q": _take(m.Q.w, ri, dev),
            "X": _take(m.X.w, ri, dev), "linear_project": m.linear_project.detach().cpu().numpy().copy()}
    uc, sc, tc, nc = np.searchsorted(ru, u), np.searchsorted(ri, seq), np.searchsorted(ri, tgt), np.searchsorted(ri, neg)
    timing = np.asarray(ro.timing_signal(L, d))

    def loss_fn(p, dt):
        v = p["Q"][_L(sc)]
        x = v + torch.from_numpy(timing).to(dt).reshape(1, L, d)
        q = torch.relu(x @ p["linear_project"])
        w = torch.softmax(q @ q.transpose(1, 2) / math.sqrt(d), -1) * (1 - torch.eye(L, dtype=dt))
        out = (w @ v).mean(1)

        def dist(items):
            return 0.2 * ((p["X"][_L(items)] - p["P"][_L(uc)][:, None, :]) ** 2).sum(2) + \
                0.8 * ((out[:, None, :] - p["Q"][_L(items)]) ** 2).sum(2)
        data = torch.clamp(dist(tc) - dist(nc) + 0.5, min=0).sum()
        return data, data
    args = (_t(u, dev), _t(seq, dev), _t(tgt, dev), _t(neg, dev))
    loss = float(m.train_op(*args).item())
    assert m.V.deferred_passes() == 1
    got = {"P": m.P.w[_t(ru, dev)], "V": m.V.w[_t(rv, dev)], "Q": m.Q.w[_t(ri, dev)], "X": m.X.w[_t(ri, dev)],
           "linear_project": m.linear_project}
    regs = {"P": 0.001, "V": 0.001, "X": 0.001, "Q": 0.001, "linear_project": 0.01}
    slots = {n: [np.full(np.shape(v), 0.05, f32)] for n, v in init.items()}
    _compare("AttRec C5 d=%d" % d, got, init, loss, loss_fn, ro.OPT_ADAGRAD, dict(lr=0.05), regs, slots)
    # two runs are identical, bit for bit
    m2 = _build(AttRec, 11, 0.05, 0.01, num_factor=d)
    loss2 = float(m2.train_op(*args).item())
    assert loss2 == loss
    for n in ("P", "V", "Q", "X"):
        assert torch.equal(getattr(m2, n).w, getattr(m, n).w), n
    assert torch.equal(m2.linear_project, m.linear_project)
    del m2
    torch.cuda.empty_cache()
    # ---- evaluate1 on N_EVAL users x all items
    al = float(m.alpha)
    X, Q = m.X.w.cpu().numpy(), m.Q.w.cpu().numpy()
    X64, Q64 = X.astype(f64), Q.astype(f64)
    x2, q2 = (X64 ** 2).sum(1), (Q64 ** 2).sum(1)

    def params(us):
        p, s, _, _ = m.getUserParam(us)
        return p.cpu().numpy(), s.cpu().numpy()

    def score64(us):
        p, s = (a.astype(f64) for a in params(us))
        sc = -al * ((p ** 2).sum(1)[:, None] + x2[None] - 2 * p @ X64.T) - (1 - al) * ((s ** 2).sum(1)[:, None] + q2[None] - 2 * s @ Q64.T)
        # every summand of the two distances has the same sign: the fp32 pairwise sums are off by a few ulp of the score
        return sc, 4e-6 * np.abs(sc).max(1)

    def score32(us):
        p, s = params(us)
        return np.concatenate([ro.numpy_order_scores(2, p[c:c + 8], X, s[c:c + 8], Q, f32(al), f32(1 - al))
                               for c in range(0, len(us), 8)])
    _check_evaluator(m, evaluate1, score64, score32, True, rng)


def test_caser_c5_step_and_evaluate_caser(dev):
    """Caser.py:79-85,98-196 + SeqRecMetrics.py:179-248 at U = 10^6, I = 10^5, L = 50, T = 3, d = 128, B = 4096,
    n_h = 16, n_v = 4, two negatives per target."""
    from deeprec_amd.models.seq_rec.Caser import Caser
    from deeprec_amd.utils.evaluation.SeqRecMetrics import evaluate_caser
    d, n_h, n_v, n_neg, reg = 128, 16, 4, 2, 1e-4
    rng = np.random.default_rng(77)
    u, seq, tgt, neg = _batch(rng, n_neg)
    ru, rq, rw, rv = _rows(rng, NU, u), _rows(rng, NI, seq), _rows(rng, NI, tgt, neg), _rows(rng, NU)
    kw = dict(n_h=n_h, n_v=n_v, num_factor=d, num_neg=n_neg)
    m = _build(Caser, 13, 0.001, reg, **kw)
    assert m.fused and m.V.lazy
    init = {"P": _take(m.P.w, ru, dev), "Q": _take(m.Q.w, rq, dev), "W": _take(m.W.w, rw, dev), "b": _take(m.b.w, rw, dev)}
    dense = {"conv_v_w": m.conv_v_w, "conv_v_b": m.conv_v_b, "full_w": m.full_w, "full_b": m.full_b,
             "conv_h_wcat": m.conv_h_wcat, "conv_h_bcat": m.conv_h_bcat}
    init.update({n: p.detach().cpu().numpy().copy() for n, p in dense.items()})
    uc, sc = np.searchsorted(ru, u), np.searchsorted(rq, seq)
    tc, nc = np.searchsorted(rw, tgt), np.searchsorted(rw, neg)
    from deeprec_amd import kernels as K

    # The max over time of a horizontal convolution (Caser.py:146-150) is not determined at fp32 precision where two
    # windows (or the best window and the ReLU's zero) are within rounding distance: 3.3 M pools per batch, a handful
    # always are, and a different winner routes the gradient through other rows.  The restatement therefore takes the
    # kernel's winners (arg: winning position, -1 = the ReLU cut it) AFTER checking, in float64, that every one of them
    # is a maximum to within 1e-6 of the activations' scale.
    with torch.no_grad():
        _, arg = K.caser_hconv_forward(m.Q.w[_t(seq, dev).long()].contiguous(), m.conv_h_wcat, m.conv_h_bcat, n_h)
    arg = arg.cpu().long()                                               # [B, n_h L], column (h - 1) n_h + c

    def windows(e, p, h):
        """Pre-activation of every window of height h: [B, L-h+1, n_h] (products + window sums, the form
        models/seq_rec/Caser.py:caser_convs uses and tests/test_golden_cpu.py checks against the literal conv2d)."""
        r0 = K.hconv_rows(n_h, h)
        G = torch.matmul(e, p["conv_h_wcat"][r0:r0 + n_h * h].t()).contiguous()                  # [B,L,n_h*h]
        s0, s1, _ = G.stride()
        return G.as_strided((B, L - h + 1, n_h, h), (s0, s1, h, s1 + 1)).sum(3) + p["conv_h_bcat"][(h - 1) * n_h:h * n_h]

    def loss_fn(p, dt):
        e = p["Q"][_L(sc)]                                                   # [B,L,d]
        out_v = torch.matmul(p["conv_v_w"][:, 0, :, 0], e) + p["conv_v_b"][None, :, None]
        outs = [out_v.permute(0, 2, 1).reshape(B, n_v * d)]
        flips = 0
        for h in range(1, L + 1):
            win = windows(e, p, h)
            a = arg[:, (h - 1) * n_h:h * n_h]
            val = win.gather(1, a.clamp(min=0)[:, None, :])[:, 0, :]
            outs.append(torch.where(a >= 0, val, torch.zeros_like(val)))
            if dt == torch.float64:
                with torch.no_grad():
                    best, where = torch.relu(win).max(1)
                    tol = 1e-6 * float(win.abs().max())
                    assert float((best - outs[-1]).abs().max()) <= tol, (h, float((best - outs[-1]).abs().max()), tol)
                    flips += int(((where != a) & (a >= 0) & (best > 0)).sum())
        if dt == torch.float64:
            print("Caser C5: %d of %d max-over-time winners differ between fp32 (kernel) and float64" % (flips, B * n_h * L))
        z = torch.relu(torch.cat(outs, 1) @ p["full_w"] + p["full_b"])
        x = torch.cat((z, p["P"][_L(uc)]), 1)

        def score(items):
            return (x[:, None, :] * p["W"][_L(items)]).sum(2) + p["b"][_L(items)]
        data = -torch.log(torch.sigmoid(score(tc)) + 1e-10).mean() - torch.log(1 - torch.sigmoid(score(nc)) + 1e-10).mean()
        return data, data
    args = (_t(u, dev), _t(seq, dev), _t(tgt, dev), _t(neg, dev))
    loss = float(m.train_op(*args).item())
    got = {"P": m.P.w[_t(ru, dev)], "Q": m.Q.w[_t(rq, dev)], "W": m.W.w[_t(rw, dev)], "b": m.b.w[_t(rw, dev)]}
    got.update(dense)
    regs = {n: reg for n in ("P", "Q", "W", "b", "conv_v_w", "full_w", "conv_h_wcat")}
    hp = dict(alpha=float(ro.adam_alpha(0.001, 0.9, 0.999, 1)), beta1=0.9, beta2=0.999, eps=1e-8)
    slots = {n: [np.zeros_like(v), np.zeros_like(v)] for n, v in init.items()}
    # Adam's first step, w -= alpha 0.1 g / (sqrt(0.001 g^2) + eps), is ill-conditioned where |g| ~ eps / 0.03 = 3e-7, and
    # among the millions of elements here some gradients are that small (the fp32 CPU restatement is 2.5e-5 / 5.5e-5 from
    # float64 on Q / W): the result must equal the float64 update of a gradient within 1e-5 of the float64 gradient's
    # scale, element by element, to 1e-5 (_compare: grad_rel)
    _compare("Caser C5", got, init, loss, loss_fn, ro.OPT_ADAM, hp, regs, slots, grad_rel=1e-5)
    # V [U, d L] (25.6 GB): the graph never reads it; its recorded dense pass is checked on 64 + rows when V is read
    v0 = None
    assert m.V.deferred_passes() == 1
    keep = {n: getattr(m, n).w.clone() for n in ("P", "Q", "W", "b")}
    keep.update({n: p.detach().clone() for n, p in dense.items()})
    # two runs are identical, bit for bit (V of the second model is never materialised beyond its initial values)
    del got
    m2 = _build(Caser, 13, 0.001, reg, **kw)
    v0 = _take(m2.V._w, rv, dev)                                             # initial rows (same seeds as m)
    assert float(m2.train_op(*args).item()) == loss
    for n in ("P", "Q", "W", "b"):
        assert torch.equal(getattr(m2, n).w, keep[n]), n
    for n in dense:
        assert torch.equal(getattr(m2, n), keep[n]), n
    del m2, keep
    torch.cuda.empty_cache()
    vg = m.V.w[_t(rv, dev)].cpu().numpy()                                    # flushes the recorded pass
    g = (f32(reg) * v0).astype(f32)
    want = ro.opt_apply(ro.OPT_ADAM, v0, g, (np.zeros_like(v0), np.zeros_like(v0)), hp)[0]
    want64 = v0.astype(f64) - hp["alpha"] * (0.1 * reg * v0.astype(f64)) / (np.sqrt(0.001 * (reg * v0.astype(f64)) ** 2) + 1e-8)
    assert np.abs(vg - want64).max() <= 1e-5 * np.abs(want64).max()
    assert np.abs(vg - want64).max() <= 4 * np.abs(want - want64).max() + 2.4e-7 * np.abs(want64).max()
    # ---- evaluate_caser on N_EVAL users x all items
    Wt, bt = m.W.w.cpu().numpy(), m.b.w.cpu().numpy()
    W64 = Wt.astype(f64)

    def score64(us):
        x = m.getUserParam(us).cpu().numpy().astype(f64)
        # a chain of k fmaf: |error| <= (k + 2) 2^-24 sum |x_c W_ic| (worst case), for the target and for the candidate
        mag = (np.abs(x) @ np.abs(W64).T + np.abs(bt.astype(f64))[None]).max(1)
        return x @ W64.T + bt.astype(f64)[None], 2 * (x.shape[1] + 2) * 2.0 ** -24 * mag

    def score32(us):
        return co.score_matrix(m.getUserParam(us).cpu().numpy(), Wt, 0, bias=bt)
    _check_evaluator(m, evaluate_caser, score64, score32, False, rng)
