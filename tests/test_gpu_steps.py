"""Step-level parity of the model classes: one `train_op` of each class on the MI355X
against a CPU restatement of the reference's graph -- torch-CPU autograd for the gradients, the oracle's TensorFlow
optimizer arithmetic (oracle.ref_ops.opt_apply) for the update -- with initial weights, ids, negatives and masks passed
in.  TensorFlow itself is not runnable here: "the reference" for this arithmetic is the restatement
(parity unpinned).

Every comparison is made against a FLOAT64 evaluation of the same step (the ground truth both fp32 paths approximate);
the fp32 CPU restatement is measured against it too, so the tolerance is not a guess: the GPU result has to be within
1e-5 (relative to the tensor's scale) of the ground truth, and no further from it than a few times the fp32 CPU path is.
"""
import contextlib
import io
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu
f32, f64 = np.float32, np.float64


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def _opt64(kind, w, g, slots, hp):
    """ro.opt_apply's formulas in float64 (the ground truth of one optimizer application)."""
    if kind == ro.OPT_SGD:
        return w - hp["lr"] * g
    if kind == ro.OPT_ADAM:
        m = slots[0] + (g - slots[0]) * (1 - hp["beta1"])
        v = slots[1] + (g * g - slots[1]) * (1 - hp["beta2"])
        return w - (m * hp["alpha"]) / (np.sqrt(v) + hp["eps"])
    if kind == ro.OPT_ADAGRAD:
        acc = slots[0] + g * g
        return w - (g * hp["lr"]) / np.sqrt(acc)
    ms = slots[0] + (g * g - slots[0]) * (1 - hp["rho"])
    mom = slots[1] * hp["momentum"] + (g * hp["lr"]) / np.sqrt(ms + hp["eps"])
    return w - mom


def _step(init, loss_fn, kind, hp, regs, slots, dtype):
    """One training step in `dtype`: autograd of loss_fn -> (+ reg * w) -> optimizer.  loss_fn(p) returns
    (data loss, total loss); the data loss is what the product's train_op reports."""
    tdt = torch.float32 if dtype == f32 else torch.float64
    leaves = {n: torch.tensor(np.asarray(v, dtype), dtype=tdt, requires_grad=True) for n, v in init.items()}
    data_loss, total = loss_fn(leaves, tdt)
    total.backward()
    out, grads = {}, {}
    for n, t in leaves.items():
        w = np.asarray(init[n], dtype)
        g = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy().astype(dtype)
        g = (g + dtype(regs.get(n, 0.0)) * w).astype(dtype)
        sl = tuple(np.asarray(s, dtype) for s in slots[n])
        if dtype == f32:
            out[n] = ro.opt_apply(kind, w, g, sl, hp)[0]
        else:
            out[n] = _opt64(kind, w, g, sl, hp)
        grads[n] = g
    return out, float(data_loss.detach()), grads


def _compare(name, got, init, loss_got, loss_fn, kind, hp, regs, slots, updated=None, grad_rel=None):
    """got: name -> tensor after the GPU step.  Against float64: <= 1e-5 of the tensor's scale, and not further from
    it than 4x the fp32 CPU restatement (plus one ulp of slack).

    grad_rel (for steps whose optimizer is ill-conditioned in places -- Adam's first step, w -= alpha 0.1 g /
    (sqrt(0.001 g^2) + eps), where |g| ~ eps / 0.03): the same 1e-5 bound, but around the float64 optimizer applied to ANY
    gradient within grad_rel * max|g| of the float64 gradient, element by element (the update is monotone in g, so the
    admissible results are the interval between the updates at g - delta and g + delta)."""
    ref64, l64, g64 = _step(init, loss_fn, kind, hp, regs, slots, f64)
    if grad_rel is not None:
        assert abs(loss_got - l64) <= 1e-5 * abs(l64), (name, "loss", loss_got, l64)
        for n in (updated or init):
            w = np.asarray(init[n], f64)
            sl = tuple(np.asarray(s_, f64) for s_ in slots[n])
            delta = grad_rel * np.abs(g64[n]).max()
            a, b = _opt64(kind, w, g64[n] - delta, sl, hp), _opt64(kind, w, g64[n] + delta, sl, hp)
            scale = max(np.abs(ref64[n]).max(), 1e-30)
            g = got[n].detach().cpu().numpy().astype(f64)
            out = np.maximum(np.minimum(a, b) - g, g - np.maximum(a, b)).max() / scale
            assert out <= 1e-5, (name, n, "outside the admissible interval by", out)
            assert np.abs(g - w).max() > 0 or np.abs(ref64[n] - w).max() == 0
        return {}
    ref32, l32, _ = _step(init, loss_fn, kind, hp, regs, slots, f32)
    assert abs(loss_got - l64) <= 1e-5 * abs(l64), (name, "loss", loss_got, l64, l32)
    report = {}
    for n in (updated or init):
        g = got[n].detach().cpu().numpy().astype(f64)
        scale = max(np.abs(ref64[n]).max(), 1e-30)
        e_gpu = np.abs(g - ref64[n]).max() / scale
        e_cpu = np.abs(ref32[n].astype(f64) - ref64[n]).max() / scale
        report[n] = (e_gpu, e_cpu)
        assert e_gpu <= 1e-5, (name, n, "gpu vs float64", e_gpu, "cpu fp32 vs float64", e_cpu)
        assert e_gpu <= 4 * e_cpu + 2.4e-7, (name, n, e_gpu, e_cpu)
        # and it really moved
        assert np.abs(g - np.asarray(init[n], f64)).max() > 0 or np.abs(ref64[n] - np.asarray(init[n], f64)).max() == 0
    for n in init:
        if updated is not None and n not in updated:           # parameters the loss does not reach stay untouched
            assert np.array_equal(got[n].detach().cpu().numpy(), np.asarray(init[n], f32)), (name, n)
    return report


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _L(a):
    return torch.from_numpy(np.asarray(a)).long()


# ---------------------------------------------------------------------------------------------------------------
def test_nnmf_step_matches_restatement_and_replays(dev):
    """nnmf.py:37-84 at the reference defaults (100 + 100 + 10 -> 210 -> 50 -> 50 -> 50 -> 1 sigmoid tower, sum of
    squares + reg * tf.norm of the four tables + kernel L2, RMSProp lr 1e-3 [TF-ext] decay 0.9, eps 1e-10, ms0 = 1)."""
    from deeprec_amd.models.rating_prediction.nnmf import NNMF
    rng = np.random.default_rng(0)
    nu, ni, B = 60, 90, 256
    dims = [210, 210, 50, 50, 50, 1]
    init = {"P": rng.standard_normal((nu, 100)) * 0.01, "Q": rng.standard_normal((ni, 100)) * 0.01,
            "U": rng.standard_normal((nu, 10)) * 0.01, "V": rng.standard_normal((ni, 10)) * 0.01}
    for l in range(5):
        init["W%d" % l] = rng.standard_normal((dims[l], dims[l + 1])) * (0.3 if l else 1.0)
        init["b%d" % l] = rng.standard_normal(dims[l + 1])
    init = {n: v.astype(f32) for n, v in init.items()}
    u = rng.integers(0, nu, B).astype(np.int32)
    i = (rng.zipf(1.5, B) % ni).astype(np.int32)
    y = rng.integers(1, 6, B).astype(f32)
    reg = 0.01

    def loss_fn(p, dt):
        x = torch.cat((p["P"][_L(u)], p["Q"][_L(i)], p["U"][_L(u)] * p["V"][_L(i)]), 1)
        for l in range(5):
            x = x @ p["W%d" % l] + p["b%d" % l]
            if l < 4:
                x = torch.sigmoid(x)
        d = torch.from_numpy(y).to(dt) - x.reshape(-1)
        data = (d * d).sum()
        tot = data + reg * (torch.linalg.norm(p["U"]) + torch.linalg.norm(p["V"]) + torch.linalg.norm(p["P"]) +
                            torch.linalg.norm(p["Q"]))
        return data, tot
    m = quiet(NNMF, None, nu, ni, 0.001, reg, 1, B)
    quiet(m.build_network, init=init)
    loss = float(m.train_op(_t(u, dev), _t(i, dev), _t(y, dev)).item())
    got = {"P": m.P.w, "Q": m.Q.w, "U": m.U.w, "V": m.V.w}
    for l in range(5):
        got["W%d" % l], got["b%d" % l] = m.kernels[l], m.biases[l]
    regs = {"W%d" % l: reg for l in range(5)}
    slots = {n: [np.ones_like(v), np.zeros_like(v)] for n, v in init.items()}
    hp = dict(lr=0.001, rho=0.9, momentum=0.0, eps=1e-10)
    _compare("NNMF", got, init, loss, loss_fn, ro.OPT_RMSPROP, hp, regs, slots)
    # the step has no host read-back any more: train() replays it as a hipGraph, and says so
    m2 = quiet(NNMF, None, nu, ni, 0.001, reg, 1, B)
    quiet(m2.build_network, init=init)
    m3 = quiet(NNMF, None, nu, ni, 0.001, reg, 1, B)
    quiet(m3.build_network, init=init)
    for s in range(4):
        m2._step(_t(u, dev), _t(i, dev), _t(y, dev))
        m3.train_op(_t(u, dev), _t(i, dev), _t(y, dev))
    assert m2._step.capture_error is None and m2._step.graph is not None and m2._step.replays == 3
    for a, b in ((m2.P.w, m3.P.w), (m2.V.w, m3.V.w), (m2.kernels[0], m3.kernels[0]), (m2.biases[4], m3.biases[4])):
        assert torch.equal(a, b)


def test_mf_class_step_matches_restatement(dev):
    """mf.py:37-55,96-103: p.q + b_u + b_i + mu, sum of squares + reg * l2_loss of the four, tf.train.AdamOptimizer."""
    from deeprec_amd.models.rating_prediction.mf import MF
    rng = np.random.default_rng(1)
    nu, ni, k, B = 70, 110, 30, 128
    init = {"P": rng.standard_normal((nu, k)) * 0.1, "Q": rng.standard_normal((ni, k)) * 0.1,
            "B_U": rng.standard_normal(nu) * 0.1, "B_I": rng.standard_normal(ni) * 0.1}
    init = {n: v.astype(f32) for n, v in init.items()}
    u = rng.integers(0, nu, B).astype(np.int32)
    i = (rng.zipf(1.4, B) % ni).astype(np.int32)
    y = rng.integers(1, 6, B).astype(f32)
    mu = 3.53

    def loss_fn(p, dt):
        pred = (p["P"][_L(u)] * p["Q"][_L(i)]).sum(1) + p["B_U"][_L(u)] + p["B_I"][_L(i)] + mu
        d = torch.from_numpy(y).to(dt) - pred
        data = (d * d).sum()
        return data, data
    m = quiet(MF, None, nu, ni, 0.001, 0.01, 1, B)
    m.build_network(k, **init)
    m.mu = mu
    loss = float(m.train_op(_t(u, dev), _t(i, dev), _t(y, dev)).item())
    hp = dict(alpha=float(ro.adam_alpha(0.001, 0.9, 0.999, 1)), beta1=0.9, beta2=0.999, eps=1e-8)
    slots = {n: [np.zeros_like(v), np.zeros_like(v)] for n, v in init.items()}
    _compare("MF", {"P": m.P, "Q": m.Q, "B_U": m.B_U, "B_I": m.B_I}, init, loss, loss_fn, ro.OPT_ADAM, hp,
             {n: 0.01 for n in init}, slots)


def _seq_batch(rng, nu, ni, B, L, T, n_neg):
    u = rng.integers(0, nu, B).astype(np.int32)
    seq = rng.integers(1, ni, (B, L)).astype(np.int32)
    seq[:, 0] = seq[0, 0]                                     # an item shared by every sequence: duplicate-heavy
    tgt = rng.integers(1, ni, (B, T)).astype(np.int32)
    neg = rng.integers(1, ni, (B, T * n_neg)).astype(np.int32)
    return u, seq, tgt, neg


def test_prme_step_matches_restatement(dev):
    """PRME.py:73-75,98,104-131 at its defaults (d = 100, L = 1, alpha 0.2, BPR on distances, 0.01 * l2 of P, X, Q,
    Adam 1e-3)."""
    from deeprec_amd.models.seq_rec.PRME import PRME
    rng = np.random.default_rng(2)
    nu, ni, d, B, T = 50, 80, 100, 128, 3
    init = {n: (rng.standard_normal((r, d)) * 0.3).astype(f32) for n, r in (("P", nu), ("V", nu), ("Q", ni), ("X", ni))}
    u, seq, tgt, neg = _seq_batch(rng, nu, ni, B, 1, T, 1)

    def loss_fn(p, dt):
        def dist(items):
            out = p["Q"][_L(seq)][:, 0, :]
            return 0.2 * ((p["X"][_L(items)] - p["P"][_L(u)][:, None, :]) ** 2).sum(2) + \
                0.8 * ((out[:, None, :] - p["Q"][_L(items)]) ** 2).sum(2)
        data = -torch.log(torch.sigmoid(-dist(tgt) + dist(neg))).sum()
        return data, data
    m = quiet(PRME, None, nu, ni, 0.001, 0.01, 1, B)
    quiet(m.build_network, 1, T, num_factor=d, init=init)
    loss = float(m.train_op(_t(u, dev), _t(seq, dev), _t(tgt, dev), _t(neg, dev)).item())
    hp = dict(alpha=float(ro.adam_alpha(0.001, 0.9, 0.999, 1)), beta1=0.9, beta2=0.999, eps=1e-8)
    slots = {n: [np.zeros_like(v), np.zeros_like(v)] for n, v in init.items()}
    _compare("PRME", {n: getattr(m, n).w for n in init}, init, loss, loss_fn, ro.OPT_ADAM, hp,
             {"P": 0.01, "X": 0.01, "Q": 0.01}, slots, updated=("P", "X", "Q"))


def test_attrec_step_matches_restatement(dev):
    """AttRec.py:97-110,132-232 at its defaults (d = 150, L = 5, T = 3, alpha 0.2, hinge margin 0.5, 0.001 * l2 of
    P, V, X, Q, kernel L2 reg_rate, Adagrad 0.05 with initial accumulator 0.05), both the hipGraph-replayed small
    step and the eager one."""
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    rng = np.random.default_rng(3)
    nu, ni, d, B, L, T = 40, 70, 150, 64, 5, 3
    init = {n: (rng.standard_normal((r, d)) * 0.2).astype(f32) for n, r in
            (("P", nu), ("V", nu), ("Q", ni), ("X", ni), ("A", ni))}
    init["W"] = (rng.standard_normal((ni, 2 * d)) * 0.01).astype(f32)
    init["b"] = (rng.standard_normal(ni) * 0.01).astype(f32)
    lim = math.sqrt(6.0 / (2 * d))
    init["linear_project"] = rng.uniform(-lim, lim, (d, d)).astype(f32)
    u, seq, tgt, neg = _seq_batch(rng, nu, ni, B, L, T, 1)
    timing = ro.timing_signal(L, d)

    def loss_fn(p, dt):
        v = p["Q"][_L(seq)]
        x = v + torch.from_numpy(np.asarray(timing)).to(dt).reshape(1, L, d)
        q = torch.relu(x @ p["linear_project"])
        w = torch.softmax(q @ q.transpose(1, 2) / math.sqrt(d), -1) * (1 - torch.eye(L, dtype=dt))
        out = (w @ v).mean(1)

        def dist(items):
            return 0.2 * ((p["X"][_L(items)] - p["P"][_L(u)][:, None, :]) ** 2).sum(2) + \
                0.8 * ((out[:, None, :] - p["Q"][_L(items)]) ** 2).sum(2)
        data = torch.clamp(dist(tgt) - dist(neg) + 0.5, min=0).sum()
        return data, data
    regs = {"P": 0.001, "V": 0.001, "X": 0.001, "Q": 0.001, "linear_project": 0.01}
    slots = {n: [np.full(np.shape(v), 0.05, f32)] for n, v in init.items()}
    names = ("P", "V", "Q", "X", "A", "W", "b")
    for graphed in (False, True):
        m = quiet(AttRec, None, nu, ni, 0.05, 0.01, 1, B)
        quiet(m.build_network, L, T, num_factor=d, init=init)
        args = (_t(u, dev), _t(seq, dev), _t(tgt, dev), _t(neg, dev))
        if not graphed:
            loss = float(m.train_op(*args).item())
            got = {n: getattr(m, n).w for n in names}
            got["linear_project"] = m.linear_project
            _compare("AttRec", got, init, loss, loss_fn, ro.OPT_ADAGRAD, dict(lr=0.05), regs, slots,
                     updated=("P", "V", "X", "Q", "linear_project"))
            m_eager = m
            for _ in range(2):
                m.train_op(*args)
        else:
            assert hasattr(m, "_step")                         # B * L * d small: train() replays the step
            for _ in range(3):
                m._step(*args)
            assert m._step.capture_error is None and m._step.replays == 2
            for n in names:
                assert torch.equal(getattr(m, n).w, getattr(m_eager, n).w), n
            assert torch.equal(m.linear_project, m_eager.linear_project)


def test_caser_step_matches_restatement(dev):
    """Caser.py:79-85,98-196 (vertical + L horizontal convolutions as conv2d, FC, concat[z, p_u] . W_i + b_i,
    mean BCE with 1e-10, reg * l2 of P, Q, V, W, b and of the conv / FC kernels, Adam 1e-3), the restatement written
    with torch's own conv2d."""
    from deeprec_amd.models.seq_rec.Caser import Caser
    rng = np.random.default_rng(4)
    nu, ni, d, B, L, T, n_h, n_v, n_neg = 40, 60, 32, 48, 5, 3, 4, 2, 2
    reg = 1e-3
    init = {"P": rng.standard_normal((nu, d)) * 0.1, "V": rng.standard_normal((nu, d * L)) * 0.1,
            "Q": rng.standard_normal((ni, d)) * 0.1, "W": rng.standard_normal((ni, 2 * d)) * 0.1,
            "b": rng.standard_normal(ni) * 0.1,
            "conv_v_w": rng.standard_normal((n_v, 1, L, 1)) * 0.5, "conv_v_b": rng.standard_normal(n_v) * 0.1,
            "full_w": rng.standard_normal((n_v * d + n_h * L, d)) * 0.2, "full_b": rng.standard_normal(d) * 0.1}
    for h in range(1, L + 1):
        init["conv_h_w%d" % h] = rng.standard_normal((n_h, 1, h, d)) * 0.3
        init["conv_h_b%d" % h] = rng.standard_normal(n_h) * 0.1
    init = {n: v.astype(f32) for n, v in init.items()}
    u, seq, tgt, neg = _seq_batch(rng, nu, ni, B, L, T, n_neg)

    def loss_fn(p, dt):
        e = p["Q"][_L(seq)][:, None]                                        # [B,1,L,d]
        out_v = F.conv2d(e, p["conv_v_w"], p["conv_v_b"])[:, :, 0, :]       # [B,n_v,d]
        outs = [out_v.permute(0, 2, 1).reshape(B, n_v * d)]                 # TF: [B,d,n_v] flattened
        for h in range(1, L + 1):
            c = torch.relu(F.conv2d(e, p["conv_h_w%d" % h], p["conv_h_b%d" % h])[:, :, :, 0])   # [B,n_h,L-h+1]
            outs.append(c.max(2)[0])
        z = torch.relu(torch.cat(outs, 1) @ p["full_w"] + p["full_b"])
        x = torch.cat((z, p["P"][_L(u)]), 1)

        def score(items):
            return (x[:, None, :] * p["W"][_L(items)]).sum(2) + p["b"][_L(items)]
        data = -torch.log(torch.sigmoid(score(tgt)) + 1e-10).mean() - torch.log(1 - torch.sigmoid(score(neg)) + 1e-10).mean()
        return data, data
    m = quiet(Caser, None, nu, ni, 0.001, reg, 1, B)
    quiet(m.build_network, L, T, n_h=n_h, n_v=n_v, num_factor=d, num_neg=n_neg, init=init)
    assert m.fused
    loss = float(m.train_op(_t(u, dev), _t(seq, dev), _t(tgt, dev), _t(neg, dev)).item())
    got = {n: getattr(m, n).w for n in ("P", "V", "Q", "W", "b")}
    got.update({"conv_v_w": m.conv_v_w, "conv_v_b": m.conv_v_b, "full_w": m.full_w, "full_b": m.full_b})
    for h in range(1, L + 1):
        got["conv_h_w%d" % h], got["conv_h_b%d" % h] = m.conv_h_w[h - 1], m.conv_h_b[h - 1]
    regs = {n: reg for n in ("P", "V", "Q", "W", "b", "conv_v_w", "full_w")}
    regs.update({"conv_h_w%d" % h: reg for h in range(1, L + 1)})
    hp = dict(alpha=float(ro.adam_alpha(0.001, 0.9, 0.999, 1)), beta1=0.9, beta2=0.999, eps=1e-8)
    slots = {n: [np.zeros_like(v), np.zeros_like(v)] for n, v in init.items()}
    _compare("Caser", got, init, loss, loss_fn, ro.OPT_ADAM, hp, regs, slots)


# ---------------------------------------------------------------------------------------------------------------
# checkpoints: train, save, load into a fresh model, train on == the uninterrupted run
# ---------------------------------------------------------------------------------------------------------------
def _ckpt_models(dev, rng):
    """(name, make() -> model, step(model, s) running training step s on fixed data) for every class."""
    from deeprec_amd.models.item_ranking.bprmf import BPRMF
    from deeprec_amd.models.item_ranking.cml import CML
    from deeprec_amd.models.item_ranking.gmf import GMF
    from deeprec_amd.models.item_ranking.neumf import NeuMF
    from deeprec_amd.models.rating_prediction.mf import MF
    from deeprec_amd.models.rating_prediction.nfm import NFM
    from deeprec_amd.models.rating_prediction.nnmf import NNMF
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    from deeprec_amd.models.seq_rec.PRME import PRME
    nu, ni, B, S = 30, 50, 32, 6
    u = [_t(rng.integers(0, nu, B).astype(np.int32), dev) for _ in range(S)]
    i = [_t(rng.integers(0, ni, B).astype(np.int32), dev) for _ in range(S)]
    j = [_t(rng.integers(0, ni, B).astype(np.int32), dev) for _ in range(S)]
    y01 = [_t(rng.integers(0, 2, B).astype(f32), dev) for _ in range(S)]
    yr = [_t(rng.integers(1, 6, B).astype(f32), dev) for _ in range(S)]
    seqb = [tuple(_t(a, dev) for a in _seq_batch(rng, nu, ni, B, 3, 2, 1)) for _ in range(S)]
    seq1 = [tuple(_t(a, dev) for a in _seq_batch(rng, nu, ni, B, 1, 2, 1)) for _ in range(S)]
    M, Fd = 40, 4
    X = [_t(rng.integers(0, M, (B, Fd)).astype(np.int32), dev) for _ in range(S)]
    Y = [_t(rng.choice([-1.0, 1.0], (B, 1)).astype(f32), dev) for _ in range(S)]
    mask = _t(np.ones((B, 8), f32), dev)
    seeds = {}

    def seeded(name, build):
        def make():
            st = np.random.get_state()
            np.random.seed(seeds.setdefault(name, int(rng.integers(1 << 30))))     # same initial weights every time
            torch.manual_seed(seeds[name])
            m = build()
            np.random.set_state(st)
            return m
        return make

    def mk(cls, args, build):
        def b():
            m = quiet(cls, *args)
            quiet(build, m)
            return m
        return b
    return [
        ("BPRMF", seeded("BPRMF", mk(BPRMF, (nu, ni, 0.01, 0.1, 1, B), lambda m: m.build_network(8))),
         lambda m, s: m.train_op(u[s], i[s], j[s])),
        ("GMF", seeded("GMF", mk(GMF, (None, nu, ni, 0.5, 0.01, 1, B), lambda m: m.build_network(8))),
         lambda m, s: m.train_op(u[s], i[s], y01[s])),
        ("CML", seeded("CML", mk(CML, (None, nu, ni, 0.1, 0.1, 1, B), lambda m: m.build_network(8))),
         lambda m, s: m.train_op(u[s], i[s], j[s], torch.ones((B, 8), dtype=torch.uint8, device=dev),
                                 torch.ones((B, 8), dtype=torch.uint8, device=dev))),
        ("NeuMF", seeded("NeuMF", mk(NeuMF, (None, nu, ni, 0.5, 0.01, 1, B), lambda m: m.build_network())),
         lambda m, s: m.train_op(u[s], i[s], y01[s])),
        ("MF", seeded("MF", mk(MF, (None, nu, ni, 0.01, 0.01, 1, B), lambda m: m.build_network(8))),
         lambda m, s: m.train_op(u[s], i[s], yr[s])),
        ("NFM", seeded("NFM", mk(NFM, (None, nu, ni, 0.05, 0.01, 1, B), lambda m: m.build_network(M, 8, 8))),
         lambda m, s: m.train_op(X[s], Y[s], mask)),
        ("NNMF", seeded("NNMF", mk(NNMF, (None, nu, ni, 0.001, 0.01, 1, B), lambda m: m.build_network(8, 4, 6))),
         lambda m, s: m.train_op(u[s], i[s], yr[s])),
        ("AttRec", seeded("AttRec", mk(AttRec, (None, nu, ni, 0.05, 0.01, 1, B), lambda m: m.build_network(3, 2, 16))),
         lambda m, s: m.train_op(*seqb[s])),
        ("Caser", seeded("Caser", mk(Caser, (None, nu, ni, 0.01, 1e-3, 1, B),
                                     lambda m: m.build_network(3, 2, n_h=4, n_v=2, num_factor=16, num_neg=1))),
         lambda m, s: m.train_op(*seqb[s])),
        ("PRME", seeded("PRME", mk(PRME, (None, nu, ni, 0.01, 0.01, 1, B), lambda m: m.build_network(1, 2, 16))),
         lambda m, s: m.train_op(*seq1[s])),
    ]


def test_every_class_resumes_from_a_checkpoint_exactly(dev, tmp_path):
    """save() after 3 steps, load() into a freshly built model (other initial weights, no optimizer state), 3 more
    steps == 6 uninterrupted steps, bit for bit: tables, dense weights, optimizer slots, Adam's step counter / beta
    powers all travel (f4)."""
    from deeprec_amd.models._common import checkpoint_tensors
    rng = np.random.default_rng(9)
    for name, make, step in _ckpt_models(dev, rng):
        a = make()
        for s in range(6):
            step(a, s)
        b = make()
        for s in range(3):
            step(b, s)
        path = str(tmp_path / (name + ".safetensors"))
        b.save(path)
        c = make()
        with torch.no_grad():
            for t in checkpoint_tensors(c).values():
                t.add_(1.0)                                        # make sure nothing survives from the fresh model
        step(c, 5)                                                 # ... including optimizer state of its own
        c.load(path)
        for s in range(3, 6):
            step(c, s)
        ta, tc = checkpoint_tensors(a), checkpoint_tensors(c)
        assert ta.keys() == tc.keys() and len(ta) >= 2, name
        for n in ta:
            assert torch.equal(ta[n], tc[n]), (name, n)
        assert a.optimizer.state() == c.optimizer.state(), name
