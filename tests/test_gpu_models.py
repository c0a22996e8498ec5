"""GPU tests of the host-side mirror: model classes and evaluate(model) protocols, through the C ABI."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


class Stub:
    pass


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def test_ranking_evaluate_matches_reference_goldens(dev):
    """RankingMetrics.evaluate(model) on stub models vs the means the REFERENCE's evaluate printed for the same
    inputs: (a) duck-typed predict path, (b) fused bilinear path (eval_operands)."""
    from deeprec_amd import kernels as K
    from deeprec_amd.utils.evaluation.RankingMetrics import evaluate
    for case in load("ranking_evaluate.json"):
        P, Q = np.array(case["P"], np.float32), np.array(case["Q"], np.float32)
        ni = Q.shape[0]
        test = {int(u): np.array(v) for u, v in case["test"].items()}
        m = Stub()
        m.device = dev
        m.num_item = ni
        m.test_users = set(u for u in test if len(test[u]) > 0)
        m.neg_items = {u: [i for i in range(ni) if i not in set(t)] for u, t in enumerate(case["train"])}
        m.test_data = test
        m.predict = lambda us, its: (P[np.array(us)] * Q[np.array(its)]).sum(1).astype(np.float32)
        res = quiet(evaluate, m)
        for k, v in case["printed"].items():
            assert abs(res[k] - v) <= 1e-12, (k, res[k], v)
        # fused path.  The stub's predict is numpy float32 arithmetic (products added by numpy's pairwise summation);
        # "order": "numpy" makes the ranks exact with respect to that order (dr_score_rank_exact_f32), so the means
        # equal the reference's on every fixture, the forced near-ties included
        Pd, Qd = torch.from_numpy(P).to(dev), torch.from_numpy(Q).to(dev)
        m.eval_operands = lambda users: {"U": Pd[torch.as_tensor(list(users), device=dev)].contiguous(), "I": Qd,
                                         "kind": K.SCORE_DOT, "order": "numpy"}
        res2 = quiet(evaluate, m)
        for k, v in case["printed"].items():
            assert abs(res2[k] - v) <= 1e-12, (k, res2[k], v)


def test_seq_evaluators_match_reference_goldens(dev):
    from deeprec_amd.utils.evaluation import SeqRecMetrics as SM
    for case in load("seq_evaluate.json"):
        P, S, X, Q = (np.array(case[k], np.float32) for k in ("P", "S", "X", "Q"))
        bias = np.array(case["bias"], np.float32)
        test = {int(u): np.array(v) for u, v in case["test"].items()}
        ni = Q.shape[0]
        exact = np.array_equal(S * 2, np.round(S * 2))
        m = Stub()
        m.device, m.num_item, m.alpha = dev, ni, case["alpha"]
        m.test_users = [u for u in range(P.shape[0])]
        m.neg_items = {u: [i for i in range(ni) if i not in set(t)] for u, t in enumerate(case["train"])}
        m.test_data = test
        users = [u for u in m.test_users if len(test[u]) > 0]
        m.test_users = users
        m.getUserParam = lambda us: (P[us], S[us], P[us], None)
        m.getItemParam = lambda items: (X, Q, bias)
        r1 = quiet(SM.evaluate1, m)
        m.getUserParam = lambda us: P[us]
        m.getItemParam = lambda items: (Q, bias)
        r2 = quiet(SM.evaluate_caser, m)
        m.predict = lambda us, its: (P[np.array(us)] * Q[np.array(its)]).sum(1).astype(np.float32)
        r3 = quiet(SM.evaluate, m)
        # evaluate1: ranks are exact in the reference's numpy operation order (MFMA pass + exact re-score of the
        # near-ties).  evaluate_caser scores with np.dot in the reference (BLAS: no fixed operation order); here a
        # k-ordered fmaf chain -- identical means unless two candidates are within rounding distance, which only the
        # forced-near-tie fixtures contain: those are checked against the oracle's canonical chain instead.
        caser_tol = 5e-2 if case.get("near_ties") else 1e-12
        for got, key, tol in ((r1, "evaluate1", 1e-12), (r2, "evaluate_caser", caser_tol), (r3, "evaluate", 1e-12)):
            for k, v in case[key].items():
                assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (key, k, got[k], v)
        if case.get("near_ties"):
            from deeprec_amd.utils.evaluation._ranks import seq_user_metrics
            sc = co.score_matrix(P, Q, 0, bias=bias)
            hr, mrr = [], []
            for u in users:
                _, _, ranks = ro.topk_and_ranks(sc[u], case["train"][u], test[u], 50)
                h, mm, _ = seq_user_metrics(sorted(int(r) for r in ranks if r >= 0), len(test[u]),
                                            ni - len(case["train"][u]), 50)
                hr.append(h), mrr.append(mm)
            assert abs(r2["HR@50"] - np.mean(hr)) <= 1e-12 and abs(r2["MRR"] - np.mean(mrr)) <= 1e-12


def _toy_data(rng, nu, ni, per_user=12):
    rows, cols = [], []
    test = {}
    for u in range(nu):
        items = rng.permutation(ni)[:per_user + 3]
        rows += [u] * per_user
        cols += items[:per_user].tolist()
        test[u] = np.sort(items[per_user:])
    train = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(nu, ni)).todok()
    return train, test


def test_bprmf_class_matches_oracle(dev):
    """BPRMF drop-in: two epochs on a toy log with the tables passed in; host sampling replayed through the
    oracle step by step (dense L2 + Keras Adam, the reference semantics)."""
    from deeprec_amd.models.item_ranking.bprmf import BPRMF
    from deeprec_amd.models import _common
    rng = np.random.default_rng(0)
    nu, ni, k, B = 40, 60, 30, 64
    train, test = _toy_data(rng, nu, ni)
    P0 = (rng.standard_normal((nu, k)) * 0.01).astype(np.float32)
    Q0 = (rng.standard_normal((ni, k)) * 0.01).astype(np.float32)
    model = quiet(BPRMF, nu, ni, 0.001, 0.1, 2, B)
    model.build_network(num_factor=k, P=P0, Q=Q0)
    quiet(model.prepare_data, train, test)
    assert model.total_batch == 480 // B
    # neg_items protocol object == the reference's list(all_items - seen)
    csr = train.tocsr()
    for u in (0, 7, 39):
        assert list(model.neg_items[u]) == sorted(set(range(ni)) - set(csr.getrow(u).nonzero()[1]))
    P, Q = P0.copy(), Q0.copy()
    sP, sQ = [np.zeros_like(P), np.zeros_like(P)], [np.zeros_like(Q), np.zeros_like(Q)]
    # the same 14 steps in float64: the ground truth the GPU (fp32) and the oracle (fp32) both approximate
    P64, Q64 = P0.astype(np.float64), Q0.astype(np.float64)
    m64 = [np.zeros_like(P64), np.zeros_like(P64), np.zeros_like(Q64), np.zeros_like(Q64)]
    step = 0
    for epoch in range(2):
        np.random.seed(100 + epoch)
        model.train()
        np.random.seed(100 + epoch)                       # replay the host-side sampling for the oracle
        idxs = np.random.permutation(model.num_training)
        ur, ir = model.user[idxs], model.item[idxs]
        jr = _common.sample_uniform_negatives(ur, model.neg_items)[:, 0]
        for b in range(model.total_batch):
            step += 1
            s = slice(b * B, (b + 1) * B)
            hp = dict(alpha=float(ro.adam_alpha(0.001, 0.9, 0.999, step)), beta1=0.9, beta2=0.999, eps=1e-7)
            P, Q, sP, sQ, _, _ = ro.pairwise_step(P, Q, sP, sQ, ur[s], ir[s], jr[s], ro.OPT_ADAM, hp, 0.1, ro.REG_DENSE)
            u_, i_, j_ = ur[s], ir[s], jr[s]
            dq = Q64[i_] - Q64[j_]
            g = -(1.0 - 1.0 / (1.0 + np.exp(-(P64[u_] * dq).sum(1))))
            gP, gQ = 0.1 * P64, 0.1 * Q64
            np.add.at(gP, u_, g[:, None] * dq)
            np.add.at(gQ, i_, g[:, None] * P64[u_])
            np.add.at(gQ, j_, -g[:, None] * P64[u_])
            a64 = 0.001 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
            for W, gW, mm, vv in ((P64, gP, m64[0], m64[1]), (Q64, gQ, m64[2], m64[3])):
                mm += (gW - mm) * (1 - 0.9)
                vv += (gW * gW - vv) * (1 - 0.999)
                W -= a64 * mm / (np.sqrt(vv) + 1e-7)
    for got, o32, t64 in ((model.P.cpu().numpy(), P, P64), (model.Q.cpu().numpy(), Q, Q64)):
        scale = np.abs(t64).max()
        e_gpu, e_cpu = np.abs(got - t64).max() / scale, np.abs(o32 - t64).max() / scale
        assert e_gpu <= 1e-5 and e_gpu <= 4 * e_cpu + 2.4e-7, (e_gpu, e_cpu)
    # predict == canonical fmaf chain; evaluate == oracle evaluator on the model's own tables
    us, its = rng.integers(0, nu, 50), rng.integers(0, ni, 50)
    Pn, Qn = model.P.cpu().numpy(), model.Q.cpu().numpy()
    assert np.array_equal(model.predict(us, its), co.pair_dot(Pn, Qn, us, its))
    res = quiet(model.test)
    ref, top10 = ro.evaluate_ranking(lambda u_, i_: co.pair_dot(Pn, Qn, u_, i_), list(model.test_users),
                                     {u: model.neg_items[u] for u in range(nu)}, test)
    for kk in ref:
        assert abs(res[kk] - ref[kk]) <= 1e-12, kk
    for u in list(model.test_users)[:10]:
        assert model.pred_ratings_10[u] == [int(x) for x in top10[u]]


def test_d1_intended_id_order(dev):
    """the reference feeds (item, user, neg) into a (user, item, neg) model (bprmf.py:117).  The build
    uses the intended order: user ids index P, item ids index Q -- ids >= num_user are valid ITEM ids."""
    from deeprec_amd.models.item_ranking.bprmf import BPRMF
    nu, ni = 5, 50
    model = quiet(BPRMF, nu, ni, 0.001, 0.1, 1, 4)
    model.build_network(num_factor=8)
    P0, Q0 = model.P.clone(), model.Q.clone()
    u = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=dev)
    i = torch.tensor([40, 41, 42, 49], dtype=torch.int32, device=dev)      # >= num_user: fine for Q
    j = torch.tensor([10, 11, 12, 13], dtype=torch.int32, device=dev)
    model.train_op(u, i, j)
    assert not torch.equal(model.Q[40], Q0[40]) and not torch.equal(model.P[0], P0[0])


def test_gmf_cml_mf_classes_smoke_and_oracle(dev):
    from deeprec_amd.models.item_ranking.cml import CML, CMLwarp
    from deeprec_amd.models.item_ranking.gmf import GMF
    from deeprec_amd.models.rating_prediction.mf import MF
    rng = np.random.default_rng(1)
    nu, ni = 30, 45
    train, test = _toy_data(rng, nu, ni, 10)
    # GMF: both call shapes (D2), one train_op vs oracle, evaluate runs
    g = quiet(GMF, nu, ni)
    g2 = quiet(GMF, None, nu, ni, 0.5, 0.01, 1, 32)
    assert (g.num_user, g.num_item) == (g2.num_user, g2.num_item) == (nu, ni)
    P0 = rng.standard_normal((nu, 10)).astype(np.float32)
    Q0 = rng.standard_normal((ni, 10)).astype(np.float32)
    g2.build_network(num_factor=10, num_neg_sample=3, P=P0, Q=Q0)
    quiet(g2.prepare_data, train, test)
    u = rng.integers(0, nu, 32).astype(np.int32)
    i = rng.integers(0, ni, 32).astype(np.int32)
    y = rng.integers(0, 2, 32).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    g2.train_op(t(u), t(i), t(y))
    ref = ro.pointwise_step(P0, Q0, [np.full_like(P0, 0.1)], [np.full_like(Q0, 0.1)], u, i, y, "bce", ro.OPT_ADAGRAD,
                            dict(lr=0.5), 0.01, ro.REG_DENSE)
    np.testing.assert_allclose(g2.P.cpu().numpy(), ref[0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g2.Q.cpu().numpy(), ref[1], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g2.predict(u, i), ro.bce_forward(ref[0], ref[1], u, i, y)[2], rtol=1e-5, atol=1e-6)
    random_state = __import__("random").getstate()
    quiet(g2.train)
    __import__("random").setstate(random_state)
    res = quiet(g2.test)
    assert set(res) >= {"ndcg@10", "map"} and 0 <= res["ndcg@10"] <= 1
    # CML / CMLwarp: one epoch + evaluate; rows stay inside the unit ball (cml.py:80-81)
    for cls in (CML, CMLwarp):
        c = quiet(cls, None, nu, ni, 0.1, 0.1, 1, 50)
        c.build_network(num_factor=20)
        quiet(c.prepare_data, train, test)
        quiet(c.train)
        assert float((c.P * c.P).sum(1).max()) <= 1 + 1e-5 and float((c.Q * c.Q).sum(1).max()) <= 1 + 1e-5
        res = quiet(c.test)
        Pn, Qn = c.P.cpu().numpy(), c.Q.cpu().numpy()
        ref, _ = ro.evaluate_ranking(lambda u_, i_: co.score_matrix(Pn, Qn, 1)[np.array(u_), np.array(i_)],
                                     list(c.test_users), {u_: c.neg_items[u_] for u_ in range(nu)}, test)
        for kk in ref:
            assert abs(res[kk] - ref[kk]) <= 1e-12, kk
        np.testing.assert_allclose(c.predict(u, i), -((Pn[u] - Qn[i]) ** 2).sum(1), rtol=1e-5, atol=1e-6)
    # MF: ratings; the fit on the training pairs improves, test() == RMSE/MAE of predict() (mf.py:80-88)
    rt = sp.random(nu, ni, 0.3, random_state=3, format="csr")
    rt.data = np.ceil(rt.data * 5)
    m = quiet(MF, None, nu, ni, 0.01, 0.01, 3, 16)
    m.build_network(num_factor=8)
    m.user, m.item, m.rating = rt.tocoo().row, rt.tocoo().col, rt.tocoo().data
    m.mu = float(np.mean(m.rating))
    before = quiet(m.test, rt.todok())
    for _ in range(20):
        quiet(m.train, rt)
    after = quiet(m.test, rt.todok())
    assert after["RMSE"] < before["RMSE"]
    d = rt.todok()
    keys = list(d.keys())
    pr = m.predict([k_[0] for k_ in keys], [k_[1] for k_ in keys]).astype(np.float64)
    yy = np.array([d[k_] for k_ in keys])
    assert abs(after["RMSE"] - np.sqrt(np.mean((yy - pr) ** 2))) < 1e-9 and abs(after["MAE"] - np.mean(np.abs(yy - pr))) < 1e-9


# ---------------------------------------------------------------------------------------------------
# autograd-bridged models: one training step against a plain torch-CPU fp32 restatement of the same graph
# (dense autograd gradients + the oracle's TF optimizer arithmetic)
# ---------------------------------------------------------------------------------------------------
def test_nfm_step_and_c3_shape(dev):
    """Config 3 shape (M=5382, F=10, k=128, B=128; nfm.py:39-82) with the dropout mask passed in."""
    from deeprec_amd.models.rating_prediction.nfm import NFM
    rng = np.random.default_rng(0)
    M, F, k, H, B = 5382, 10, 128, 128, 128
    init = {"feature_embeddings": (rng.standard_normal((M, k)) * 0.01).astype(np.float32),
            "feature_bias": (rng.standard_normal(M) * 0.01).astype(np.float32),
            "bias": np.float32(0.1), "pred_weight": (rng.standard_normal((H, 1)) * 0.09).astype(np.float32),
            "W1": rng.standard_normal((k, H)).astype(np.float32), "b1": rng.standard_normal(H).astype(np.float32)}
    m = quiet(NFM, None, 957, 4082, 0.05, 0.01, 1, B)
    m.build_network(M, k, H, init=init)
    X = rng.integers(0, M, (B, F)).astype(np.int32)
    X[:, 0] = X[0, 0]                                           # a feature shared by every row: duplicate-heavy
    Y = rng.choice([-1.0, 1.0], (B, 1)).astype(np.float32)
    mask = (rng.random((B, H)) < 0.8).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    loss = m.train_op(t(X), t(Y), t(mask))

    def loss_fn(p, dt):
        v = p["feature_embeddings"][torch.from_numpy(X).long()]
        s = v.sum(1)
        fm = 0.5 * (s * s - (v * v).sum(1))
        h = torch.relu(fm @ p["W1"] + p["b1"]) * torch.from_numpy(mask).to(dt) / 0.8
        pred = (h @ p["pred_weight"]).sum(1, keepdim=True) + p["feature_bias"][torch.from_numpy(X).long()].sum(1, keepdim=True) + p["bias"]
        d = torch.from_numpy(Y).to(dt) - pred
        return 0.5 * (d * d).sum()
    slots = {n: [np.full(np.shape(v), 0.1, np.float32)] for n, v in init.items()}
    # against the float64 ground truth of the step: within 1e-5 of each tensor's scale and no further from it than
    # 4x the fp32 CPU restatement (tests/test_gpu_steps.py:_compare)
    from test_gpu_steps import _compare
    init_np = {n: np.asarray(v, np.float32) for n, v in init.items()}
    got = {"feature_embeddings": m.feature_embeddings.w, "feature_bias": m.feature_bias.w, "W1": m.W1, "b1": m.b1,
           "pred_weight": m.pred_weight, "bias": m.bias}
    _compare("NFM", got, init_np, float(loss), lambda p, dt: (lambda v: (v, v))(loss_fn(p, dt)), ro.OPT_ADAGRAD,
             dict(lr=0.05), {"feature_embeddings": 0.01}, slots)
    # feature_bias: true sparse gradient -> untouched rows keep their accumulator and value (nfm.py:74)
    got_fb = m.feature_bias.w.cpu().numpy()
    untouched = np.setdiff1d(np.arange(M), np.unique(X))
    assert np.array_equal(got_fb[untouched], init["feature_bias"][untouched])
    # train()/test() plumbing on frappe-shaped lists
    data = {"X": X.tolist(), "Y": Y.tolist()}
    quiet(m.train, data)
    r = quiet(m.test, data, t(np.ones((B, H), np.float32)))
    assert np.isfinite(r["RMSE"])


def test_neumf_step_eval_and_c2_shape(dev):
    """NeuMF (neumf.py:61-131) at the C2 table shape (6040 users x 3706 items, k = 10 / 64, B = 256): one Adagrad step
    vs the float64 ground truth and the fp32 CPU restatement; eval_scores == predict; evaluate runs."""
    from deeprec_amd.models.item_ranking.neumf import NeuMF
    rng = np.random.default_rng(2)
    nu, ni, B = 6040, 3706, 256
    train, test = _toy_data(rng, 50, 80, 10)
    train.resize((nu, ni))
    dims = [128, 128, 80, 40, 20, 10]
    init = {"P": rng.standard_normal((nu, 10)), "Q": rng.standard_normal((ni, 10)),
            "mlp_P": rng.standard_normal((nu, 64)) * 0.1, "mlp_Q": rng.standard_normal((ni, 64)) * 0.1}
    for l in range(5):
        init["W%d" % l] = rng.standard_normal((dims[l], dims[l + 1])) * 0.2
        init["b%d" % l] = rng.standard_normal(dims[l + 1]) * 0.1
    init = {n: v.astype(np.float32) for n, v in init.items()}
    m = quiet(NeuMF, None, nu, ni, 0.5, 0.01, 1, B)
    m.build_network(init=init, num_neg_sample=4)
    quiet(m.prepare_data, train, test)
    u = rng.integers(0, nu, B).astype(np.int32)
    i = (rng.zipf(1.5, B) % ni).astype(np.int32)
    y = rng.integers(0, 2, B).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    loss = m.train_op(t(u), t(i), t(y))

    def loss_fn(p, dt):
        ul, il = torch.from_numpy(u).long(), torch.from_numpy(i).long()
        x = torch.cat((p["mlp_Q"][il], p["mlp_P"][ul]), 1)
        for l in range(5):
            x = torch.relu(x @ p["W%d" % l] + p["b%d" % l])
        z = torch.cat((p["P"][ul] * p["Q"][il], x), 1).sum(1)
        pr = torch.sigmoid(z)
        yy = torch.from_numpy(y).to(dt)
        v = -(yy * torch.log(pr + 1e-10) + (1 - yy) * torch.log(1 - pr + 1e-10)).sum()
        return v, v
    regs = {n: 0.01 for n in init if not n.startswith("b")}
    slots = {n: [np.full(v.shape, 0.1, np.float32)] for n, v in init.items()}
    got = {"P": m.P.w, "Q": m.Q.w, "mlp_P": m.mlp_P.w, "mlp_Q": m.mlp_Q.w}
    for l in range(5):
        got["W%d" % l], got["b%d" % l] = m.kernels[l], m.biases[l]
    from test_gpu_steps import _compare
    _compare("NeuMF", got, init, float(loss), loss_fn, ro.OPT_ADAGRAD, dict(lr=0.5), regs, slots)
    # evaluator fast path == predict
    users = [3, 7, 11]
    sc = m.eval_scores(users).cpu().numpy()
    for r, uu in enumerate(users):
        np.testing.assert_allclose(sc[r], m.predict([uu] * ni, list(range(ni))), rtol=1e-5, atol=1e-6)
    quiet(m.train)
    res = quiet(m.test)
    assert 0 <= res["ndcg@10"] <= 1


def _seq_dataset(tmp_path, rng, nu=25, ni=40, L=3, T=2):
    from deeprec_amd.utils.load_data.load_data_seq import DataSet
    lines = []
    for u in range(nu):
        for it in rng.permutation(ni)[:int(rng.integers(L + T + 1, L + T + 6))]:
            lines.append("%d %d 1\n" % (u + 1, it + 1))
    p = tmp_path / "seq_train.txt"
    p.write_text("".join(lines))
    train = quiet(DataSet, str(p), ['user', 'item', 'rating'], " ", L, T, True)
    tl = []
    for u in range(nu):
        tl.append("%d %d 1\n" % (u + 1, int(rng.integers(1, ni + 1))))
    pt = tmp_path / "seq_test.txt"
    pt.write_text("".join(tl))
    test = quiet(DataSet, str(pt), ['user', 'item', 'rating'], " ", 1, 1, False, train.user_map, train.item_map,
                 train.num_user, train.num_item)
    return train, test


def f32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def test_seq_models_train_and_evaluate(dev, tmp_path):
    """AttRec / Caser / PRME drop-ins (testSeqRec.py:45-70 flow): prepare, one epoch, test(); the evaluators'
    numbers equal the oracle ranking of the models' own predict() scores."""
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    from deeprec_amd.models.seq_rec.PRME import PRME
    from deeprec_amd.utils.evaluation._ranks import seq_user_metrics
    rng = np.random.default_rng(5)
    for cls, L, kw in ((AttRec, 3, dict(num_factor=16)), (Caser, 3, dict(num_factor=16, n_h=4, n_v=2)),
                       (PRME, 1, dict(num_factor=16))):
        train, test = _seq_dataset(tmp_path, rng, L=L)
        m = quiet(cls, None, train.num_user, train.num_item, 0.05 if cls is AttRec else 0.001, 0.01, 1, 20)
        quiet(m.build_network, L, 2, **kw)
        quiet(m.prepare_data, train, test)
        np.random.seed(0)
        quiet(m.train, train)
        res = quiet(m.test, test)
        assert np.isfinite(res["HR@50"]) and 0 <= res["MRR"] <= 1
        # (a) EXACT: the oracle ranking of the scores in the arithmetic each evaluator documents -- evaluate1: the
        # reference's own numpy float32 operation order (SeqRecMetrics.py:122-123) from getUserParam / getItemParam;
        # evaluate_caser and PRME's evaluate: the k-ordered fmaf chain of the scoring kernel
        ni = train.num_item
        users = list(m.test_users)
        if cls is AttRec:
            uf, sf, _, _ = m.getUserParam(m.test_users)
            i1, i2, _ = m.getItemParam(np.expand_dims(np.arange(ni), axis=1))
            a = np.float32(m.alpha)
            S = ro.numpy_order_scores(2, f32(uf), f32(i1), f32(sf), f32(i2), a, np.float32(1 - a))
        elif cls is Caser:
            uf = m.getUserParam(m.test_users)
            itf, b = m.getItemParam(np.expand_dims(np.arange(ni), axis=1))
            S = co.score_matrix(f32(uf), f32(itf), 0, bias=f32(b).reshape(-1))
        else:
            ops = m.seq_eval_operands(users)
            S = co.score_matrix(f32(ops["U"]), f32(ops["I"]), 2, f32(ops["U2"]), f32(ops["I2"]), ops["alpha"])
        hr, mrr, hr_p, mrr_p = [], [], [], []
        for r, uu in enumerate(users):
            cand = m.neg_items[uu]
            masked = np.setdiff1d(np.arange(ni), cand)
            tg = np.unique(m.test_data[uu])
            _, _, ranks = ro.topk_and_ranks(S[r], masked, tg, 50)
            h, mm, _ = seq_user_metrics(sorted(int(x) for x in ranks if x >= 0), len(m.test_data[uu]), len(cand), 50)
            hr.append(h), mrr.append(mm)
            # (b) the same from the model's own predict() (PRME.py:226-236 and friends: direct per-pair arithmetic)
            sp_ = m.predict([uu] * ni, list(range(ni)))
            _, _, ranks = ro.topk_and_ranks(sp_, masked, tg, 50)
            h, mm, _ = seq_user_metrics(sorted(int(x) for x in ranks if x >= 0), len(m.test_data[uu]), len(cand), 50)
            hr_p.append(h), mrr_p.append(mm)
        assert res["HR@50"] == float(np.mean(hr)) and res["MRR"] == float(np.mean(mrr)), cls.__name__
        # predict() rounds differently from the evaluator's form: only a near-tie may move
        assert abs(res["HR@50"] - np.mean(hr_p)) <= 0.05 and abs(res["MRR"] - np.mean(mrr_p)) <= 0.02, cls.__name__


def test_nnmf_runs(dev):
    from deeprec_amd.models.rating_prediction.nnmf import NNMF
    rt = sp.random(30, 40, 0.3, random_state=3, format="csr")
    rt.data = np.ceil(rt.data * 5)
    m = quiet(NNMF, None, 30, 40, 0.001, 0.01, 1, 32)
    quiet(m.build_network, 20, 4, 8)
    m.user, m.item, m.rating = rt.tocoo().row, rt.tocoo().col, rt.tocoo().data
    before = quiet(m.test, rt.todok())["RMSE"]
    for _ in range(30):
        quiet(m.train, rt)
    assert quiet(m.test, rt.todok())["RMSE"] < before


def test_bprmf_device_sampler_and_checkpoint(dev, tmp_path):
    """sampler="device": the epoch's negatives are dr_sample_negatives_i32 draws (== oracle restatement, never a
    train item), the step that consumes them matches the oracle; save() / load() round-trips tables and slots."""
    from deeprec_amd.models.item_ranking.bprmf import BPRMF
    rng = np.random.default_rng(4)
    nu, ni, k, B = 40, 60, 16, 96
    train, test = _toy_data(rng, nu, ni)
    P0 = (rng.standard_normal((nu, k)) * 0.01).astype(np.float32)
    Q0 = (rng.standard_normal((ni, k)) * 0.01).astype(np.float32)
    model = quiet(BPRMF, nu, ni, 0.001, 0.1, 1, B, sampler="device")
    model.build_network(num_factor=k, P=P0, Q=Q0)
    quiet(model.prepare_data, train, test)
    np.random.seed(11)
    model.train()
    np.random.seed(11)                                    # replay: permutation, then the epoch's sampler seed
    idxs = np.random.permutation(model.num_training)
    seed = int(np.random.randint(0, 2 ** 31 - 1))
    ur, ir = model.user[idxs], model.item[idxs]
    csr = train.tocsr()
    csr.sort_indices()
    jr = co.sample_negatives(csr.indptr, csr.indices, ur, ni, seed)[:, 0]
    for e in range(len(ur)):
        assert jr[e] not in csr.indices[csr.indptr[ur[e]]:csr.indptr[ur[e] + 1]]
    P, Q = P0.copy(), Q0.copy()
    sP, sQ = [np.zeros_like(P), np.zeros_like(P)], [np.zeros_like(Q), np.zeros_like(Q)]
    for b in range(model.total_batch):
        s = slice(b * B, (b + 1) * B)
        hp = dict(alpha=float(ro.adam_alpha(0.001, 0.9, 0.999, b + 1)), beta1=0.9, beta2=0.999, eps=1e-7)
        P, Q, sP, sQ, _, _ = ro.pairwise_step(P, Q, sP, sQ, ur[s], ir[s], jr[s], ro.OPT_ADAM, hp, 0.1, ro.REG_DENSE)
    np.testing.assert_allclose(model.P.cpu().numpy(), P, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(model.Q.cpu().numpy(), Q, rtol=1e-4, atol=1e-6)
    path = str(tmp_path / "bprmf.safetensors")
    model.save(path)
    other = quiet(BPRMF, nu, ni, 0.001, 0.1, 1, B)
    other.build_network(num_factor=k)
    other.load(path)
    assert torch.equal(other.P, model.P) and torch.equal(other.Q, model.Q)
    for a, b in zip(other.optimizer.slots_for(other.P), model.optimizer.slots_for(model.P)):
        assert (a is None and b is None) or torch.equal(a, b)


# ------------------------------------------------------------------ drivers (test/*.py of the reference)
@pytest.mark.parametrize("driver,argv", [
    ("test_item_ranking", ["--model", "BPRMF", "--epochs", "1", "--batch_size", "4096", "--num_factors", "16"]),
    ("test_item_ranking", ["--model", "CML", "--epochs", "1"]),
    ("test_rating_pred", ["--model", "MF", "--epochs", "1", "--batch_size", "4096"]),
    ("test_rating_pred", ["--model", "NFM", "--epochs", "1", "--batch_size", "512"]),
    ("testSeqRec", ["--model", "PRME", "--epochs", "1", "--num_factors", "32"]),
])
def test_drivers_run_on_synthetic_data(dev, driver, argv):
    """deeprec_amd/test/*.py: flags reach the model, load -> build_network -> execute runs end to end
    on ml100k- / frappe-shaped synthetic files and leaves finite parameters."""
    import importlib
    mod = importlib.import_module("deeprec_amd.test." + driver)
    model = quiet(mod.main, argv + ["--synthetic", "--device", str(dev)])
    if "--batch_size" in argv:
        assert model.batch_size == int(argv[argv.index("--batch_size") + 1])
    tabs = []
    for v in vars(model).values():                      # plain tensors (P, Q) and Table wrappers (.w)
        t = getattr(v, "w", v)
        if isinstance(t, torch.Tensor) and t.is_floating_point() and t.dim() == 2:
            tabs.append(t)
    assert tabs and all(bool(torch.isfinite(t).all()) for t in tabs)


def test_graphed_step_equals_eager_loop(dev):
    """_common.GraphedStep (one hipGraph per fixed-shape training step, used by NeuMF / NFM / NNMF / AttRec train()):
    the replayed steps leave exactly the weights the eager loop leaves."""
    from deeprec_amd.models._common import GraphedStep
    from deeprec_amd.models.item_ranking.neumf import NeuMF
    from deeprec_amd.models.rating_prediction.nfm import NFM
    rng = np.random.default_rng(11)
    t = lambda a, dt=torch.int32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)

    def neumf():
        torch.manual_seed(0)
        m = quiet(NeuMF, None, 300, 200, 0.5, 0.01, 1, 256)
        m.build_network()
        return m, lambda mm: [mm.P.w, mm.Q.w, mm.mlp_P.w, mm.mlp_Q.w] + list(mm.kernels) + list(mm.biases)
    nb = [(t(rng.integers(0, 300, 256)), t(rng.zipf(1.3, 256) % 200), t(rng.integers(0, 2, 256), torch.float32)) for _ in range(6)]

    def nfm():
        torch.manual_seed(0)
        m = quiet(NFM, None, 957, 4082, 0.05, 0.01, 1, 128)
        m.build_network(500, 64, 64)
        return m, lambda mm: [mm.feature_embeddings.w, mm.feature_bias.w, mm.W1, mm.b1, mm.pred_weight, mm.bias]
    mask = t((rng.random((128, 64)) < 0.8), torch.float32)
    fb = [(t(rng.integers(0, 500, (128, 10))), t(rng.choice([-1.0, 1.0], (128, 1)), torch.float32), mask) for _ in range(6)]

    for make, batches in ((neumf, nb), (nfm, fb)):
        a, params = make()
        start = [x.detach().clone() for x in params(a)]
        for b in batches:
            la = a.train_op(*b)
        g, _ = make()
        with torch.no_grad():
            for x, w0 in zip(params(g), start):             # same starting point (random init differs per build)
                x.copy_(w0)
        step = GraphedStep(g.train_op)
        for b in batches:
            lg = step(*b)
        assert step.graph is not None and not step.off, "the step was not captured"
        for x, y in zip(params(a), params(g)):
            assert torch.equal(x.detach(), y.detach())
        assert float(la) == float(lg)


def test_pointwise_models_device_sampler(dev):
    """GMF / NeuMF with sampler="device": an epoch's negatives come from dr_sample_negatives_i32 (never a train item of
    the user), every positive appears once with its label, and train() runs on it."""
    from deeprec_amd.models._common import device_pointwise_epoch
    from deeprec_amd.models.item_ranking.gmf import GMF
    from deeprec_amd.models.item_ranking.neumf import NeuMF
    rng = np.random.default_rng(3)
    nu, ni = 60, 90
    train, test = _toy_data(rng, nu, ni, 12)
    for cls in (GMF, NeuMF):
        m = quiet(cls, None, nu, ni, 0.5, 0.01, 1, 64, sampler="device")
        quiet(m.build_network, num_neg_sample=5)
        quiet(m.prepare_data, train, test)
        np.random.seed(1)
        u, i, y = (x.cpu().numpy() for x in device_pointwise_epoch(m))
        npos = len(m.user)
        assert len(u) == npos * 6 and int((y > 0).sum()) == npos
        csr = train.tocsr()
        seen = {uu: set(csr.indices[csr.indptr[uu]:csr.indptr[uu + 1]]) for uu in range(nu)}
        neg = y == 0
        assert all(int(it) not in seen[int(uu)] for uu, it in zip(u[neg], i[neg]))
        assert sorted(zip(u[~neg].tolist(), i[~neg].tolist())) == sorted(zip(map(int, m.user), map(int, m.item)))
        quiet(m.train)
        assert 0 <= quiet(m.test)["ndcg@10"] <= 1


def test_graphed_adam_models_equal_eager_loop(dev):
    """BPRMF and MF train with Adam: with the bias-corrected step size on the device (TFOptimizer.device_alpha) their
    steps can be replayed as hipGraphs too, and leave exactly the tables the eager loop leaves.  (Their train() keeps
    the eager loop: a handful of kernels per step, already GPU-bound; replay measured 0.13 vs 0.11 ms at C1.)"""
    from deeprec_amd.models._common import GraphedStep
    from deeprec_amd.models.item_ranking.bprmf import BPRMF
    from deeprec_amd.models.rating_prediction.mf import MF
    rng = np.random.default_rng(17)
    t = lambda a, dt=torch.int32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    nu, ni, B = 200, 150, 512
    P0 = (rng.standard_normal((nu, 30)) * 0.01).astype(np.float32)
    Q0 = (rng.standard_normal((ni, 30)) * 0.01).astype(np.float32)
    bb = [(t(rng.integers(0, nu, B)), t(rng.zipf(1.3, B) % ni), t(rng.integers(0, ni, B))) for _ in range(6)]
    mb = [(t(rng.integers(0, nu, B)), t(rng.integers(0, ni, B)), t(rng.integers(1, 6, B), torch.float32)) for _ in range(6)]

    def bprmf():
        m = quiet(BPRMF, nu, ni)
        m.build_network(P=P0, Q=Q0)
        return m, [m.P, m.Q]

    def mf():
        m = quiet(MF, None, nu, ni)
        m.build_network(P=P0, Q=Q0, B_U=np.zeros(nu, np.float32), B_I=np.zeros(ni, np.float32))
        return m, [m.P, m.Q, m.B_U, m.B_I]

    for make, batches in ((bprmf, bb), (mf, mb)):
        a, pa = make()
        for b in batches:
            la = float(a.train_op(*b))
        g, pg = make()
        g.optimizer.device_alpha(dev)
        step = GraphedStep(g.train_op)
        for b in batches:
            lg = float(step(*b))
        assert step.graph is not None and not step.off, "the step was not captured"
        for x, y in zip(pa, pg):
            assert torch.equal(x, y)
        assert la == lg


def test_attrec_fused_attention_equals_library_graph(dev):
    """AttRec.train_op with the attention block on dr_attn_pool_* (one autograd node) against the same step with the
    block written as library ops under autograd (AttRec.py:147-186 literally): tables and the projection after three
    Adagrad steps within 1e-5 of their scale; getUserParam's weights and output agree."""
    from deeprec_amd import kernels as K
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    rng = np.random.default_rng(3)
    nu, ni, L, T_, B, d = 40, 60, 6, 3, 32, 24
    init = {n: rng.normal(0, 0.05, (r, d)).astype(np.float32) for n, r in (("P", nu), ("V", nu), ("Q", ni), ("X", ni), ("A", ni))}
    init["linear_project"] = rng.uniform(-0.3, 0.3, (d, d)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.int32)
    bs = [(t(rng.integers(0, nu, B)), t(rng.integers(1, ni, (B, L))), t(rng.integers(1, ni, (B, T_))),
           t(rng.integers(1, ni, (B, T_)))) for _ in range(3)]

    def run(fused):
        old = K.ATT_MAX_L
        K.ATT_MAX_L = old if fused else 0
        try:
            m = quiet(AttRec, None, nu, ni, 0.05, 0.01, 1, B)
            quiet(m.build_network, L, T_, num_factor=d, num_neg=1, init=init)
            losses = [float(m.train_op(*b)) for b in bs]
            m.test_sequences = rng2.integers(1, ni, (nu, L))
            up = m.getUserParam(np.arange(nu))
            return m, losses, up
        finally:
            K.ATT_MAX_L = old
    rng2 = np.random.default_rng(9)
    a, la, ua = run(True)
    rng2 = np.random.default_rng(9)
    b, lb, ub = run(False)
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    for n in ("P", "Q", "X"):
        wa, wb = getattr(a, n).w.cpu().numpy(), getattr(b, n).w.cpu().numpy()
        np.testing.assert_allclose(wa, wb, rtol=1e-4, atol=1e-5 * np.abs(wb).max())
    wa, wb = a.linear_project.detach().cpu().numpy(), b.linear_project.detach().cpu().numpy()
    np.testing.assert_allclose(wa, wb, rtol=1e-4, atol=1e-5 * np.abs(wb).max())
    np.testing.assert_allclose(ua[1].cpu().numpy(), ub[1].cpu().numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ua[3].cpu().numpy(), ub[3].cpu().numpy(), rtol=1e-4, atol=1e-7)


def test_lazy_dense_pass_of_unread_tables_is_exact(dev):
    """_common.Table.lazy (Caser's / AttRec's V: in the L2 term only): the recorded dense optimizer passes, replayed when
    the table is read, leave the table and its optimizer slots bit-identical to running them inside every step; nothing
    else changes."""
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    rng = np.random.default_rng(11)
    nu, ni, L, T_, B, d = 50, 70, 5, 3, 16, 12
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.int32)
    bs = [(t(rng.integers(0, nu, B)), t(rng.integers(1, ni, (B, L))), t(rng.integers(1, ni, (B, T_))),
           t(rng.integers(1, ni, (B, T_)))) for _ in range(4)]

    def run(cls, lazy, kw, init):
        np.random.seed(123)                     # the tables / weights not in `init` draw their seeds from these
        torch.manual_seed(123)
        m = quiet(cls, None, nu, ni, 0.05 if cls is AttRec else 0.001, 0.01, 1, B)
        quiet(m.build_network, L, T_, init=init, lazy_v=lazy, **kw)
        if cls is AttRec:                       # small shapes replay a hipGraph (no host code per step): run eagerly here
            m._step = None
            m.V.lazy = lazy
        for b in bs:
            m.train_op(*b)
        n_def = m.V.deferred_passes()
        out = {n: getattr(m, n).w.clone() for n in ("P", "V", "Q")}       # reading V.w replays the recorded passes
        assert not m.V._deferred
        out["V.slots"] = [x.clone() for x in m.optimizer.slots_for(m.V.w) if x is not None]
        return out, n_def

    for cls, kw in ((Caser, dict(num_factor=d, n_h=3, n_v=2, num_neg=1)), (AttRec, dict(num_factor=d, num_neg=1))):
        init = {n: rng.normal(0, 0.05, (r, c)).astype(np.float32)
                for n, r, c in (("P", nu, d), ("Q", ni, d), ("V", nu, d * L if cls is Caser else d))}
        a, na = run(cls, True, kw, init)
        b, nb = run(cls, False, kw, init)
        assert na == len(bs) and nb == 0
        for n in ("P", "V", "Q"):
            assert torch.equal(a[n], b[n]), (cls.__name__, n)
        for x, y in zip(a["V.slots"], b["V.slots"]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("kind", ["sgd", "adam", "adagrad", "rmsprop"])
def test_dense_l2_step_without_dense_gradient_is_exact(dev, kind):
    """TFOptimizer.apply_table in DR_REG_DENSE mode for a big table with one lookup: dr_mark_rows_u8 +
    dr_dense_update_masked_f32 + dr_sparse_row_update_f32 (touched rows) against the dense-gradient form (zero fill +
    dr_scatter_add_sorted_f32 + dr_dense_update_f32): table and optimizer slots bit-identical after three steps, with
    duplicate ids and an id out of range; the flag array is all zero again afterwards."""
    from deeprec_amd import kernels as K
    from deeprec_amd.models import _common as C
    rng = np.random.default_rng(17)
    R, k, n = 5000, 32, 3000
    w0 = rng.normal(0, 0.1, (R, k)).astype(np.float32)
    steps = []
    for _ in range(3):
        ids = rng.integers(0, R // 4, n).astype(np.int32)          # heavy duplication
        ids[5] = R + 7                                             # out of range: skipped by both forms
        steps.append((torch.from_numpy(ids).to(dev), torch.from_numpy(rng.normal(0, 1, (n, k)).astype(np.float32)).to(dev)))

    def run(min_elems):
        old = C.SPARSE_DENSE_MIN
        C.SPARSE_DENSE_MIN = min_elems
        try:
            t = C.Table(torch.from_numpy(w0).to(dev))
            o = C.TFOptimizer(kind, 0.01)
            for ids, rows in steps:
                t.pending.append((ids, rows))
                o.begin_step()
                o.apply_table(t, 0.05, K.REG_DENSE)
            return t, [x.clone() for x in o.slots_for(t.w) if x is not None]
        finally:
            C.SPARSE_DENSE_MIN = old
    a, sa = run(1)
    b, sb = run(1 << 60)
    assert torch.equal(a.w, b.w)
    for x, y in zip(sa, sb):
        assert torch.equal(x, y)
    assert int(a.row_flags().sum()) == 0 and getattr(b, "_flags", None) is None


# ---------------------------------------------------------------------------------------------------
# host-side sampling / batching of the reference, replayed draw for draw
# ---------------------------------------------------------------------------------------------------
def _reference_pointwise_epoch(user, item, label, neg_items, n_neg, seed):
    """gmf.py:100-124 / neumf.py:153-180 restated literally: random.sample per positive, label 0, one permutation."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    item_temp, user_temp, labels_temp = item[:], user[:], label[:]
    user_append, item_append, values_append = [], [], []
    for u in user:
        list_of_random_items = random.sample(neg_items[u], n_neg)
        user_append += [u] * n_neg
        item_append += list_of_random_items
        values_append += [0] * n_neg
    item_temp += item_append
    user_temp += user_append
    labels_temp += values_append
    idxs = np.random.permutation(len(item_temp))
    return np.array(user_temp)[idxs], np.array(item_temp)[idxs], np.array(labels_temp)[idxs]


def test_pointwise_models_reference_sampling_stream(dev):
    """GMF.train / NeuMF.train in the default (reference) sampler mode consume Python's `random` and numpy's RNG
    exactly as gmf.py:100-124 and neumf.py:153-180 do: same negatives, same shuffle, same batches (remainder
    dropped), for the same seeds."""
    import random
    from deeprec_amd.models.item_ranking.gmf import GMF
    from deeprec_amd.models.item_ranking.neumf import NeuMF
    rng = np.random.default_rng(21)
    nu, ni, B = 30, 45, 64
    train, test = _toy_data(rng, nu, ni, 9)
    csr = train.tocsr()
    neg = {u: sorted(set(range(ni)) - set(csr.getrow(u).nonzero()[1])) for u in range(nu)}
    coo = train.tocoo()
    user, item, label = list(coo.row.reshape(-1)), list(coo.col.reshape(-1)), list(coo.data)
    for cls, n_neg in ((GMF, 5), (NeuMF, 7)):
        m = quiet(cls, None, nu, ni, 0.5, 0.01, 1, B)
        quiet(m.build_network, num_neg_sample=n_neg)
        quiet(m.prepare_data, train, test)
        ur, ir, yr = _reference_pointwise_epoch(user, item, label, neg, n_neg, 77)
        got = []
        if cls is GMF:
            m.train_op = lambda u_, i_, y_: got.append((u_.cpu().numpy(), i_.cpu().numpy(), y_.cpu().numpy())) or torch.zeros(1)
        else:
            m._step = lambda u_, i_, y_: got.append((u_.cpu().numpy(), i_.cpu().numpy(), y_.cpu().numpy())) or torch.zeros(1)
        random.seed(77)
        np.random.seed(77)
        quiet(m.train)
        assert len(got) == len(ur) // B and len(got) > 3
        for b, (gu, gi, gy) in enumerate(got):
            s = slice(b * B, (b + 1) * B)
            assert np.array_equal(gu, ur[s]) and np.array_equal(gi, ir[s]) and np.array_equal(gy, yr[s].astype(np.float32)), (cls.__name__, b)


def test_nfm_train_shuffle_and_test_clipping_match_reference(dev):
    """nfm.py:91-111: labels and feature rows are shuffled by the SAME permutation (RNG state replayed), batches are
    consecutive slices, the remainder is dropped; nfm.py:113-131: test predictions are clipped to the label range
    before the RMSE."""
    import math
    from deeprec_amd.models.rating_prediction.nfm import NFM
    rng = np.random.default_rng(8)
    M, Fd, n, B = 50, 4, 150, 32
    X = rng.integers(0, M, (n, Fd)).tolist()
    Y = [[float(v)] for v in rng.choice([-1.0, 1.0], n)]
    m = quiet(NFM, None, 10, 10, 0.05, 0.01, 1, B)
    quiet(m.build_network, M, 8, 8)
    got = []
    m._step = lambda x_, y_: got.append((x_.cpu().numpy(), y_.cpu().numpy())) or torch.zeros(1)
    data = {"X": [r[:] for r in X], "Y": [r[:] for r in Y]}
    np.random.seed(5)
    quiet(m.train, data)
    # the reference's lines on plain lists
    ref = {"X": [r[:] for r in X], "Y": [r[:] for r in Y]}
    np.random.seed(5)
    rng_state = np.random.get_state()
    np.random.shuffle(ref["Y"])
    np.random.set_state(rng_state)
    np.random.shuffle(ref["X"])
    assert len(got) == n // B
    for b, (gx, gy) in enumerate(got):
        assert np.array_equal(gx, np.array(ref["X"][b * B:(b + 1) * B])) and \
            np.array_equal(gy, np.array(ref["Y"][b * B:(b + 1) * B], np.float32))
    assert data["X"] == ref["X"] and data["Y"] == ref["Y"]                 # shuffled in place, like the reference
    # test(): a forward that returns values outside the label range
    preds = rng.standard_normal(n).astype(np.float32) * 3
    m._forward = lambda x_, mask=None: torch.from_numpy(preds).to(dev).reshape(-1, 1)
    res = quiet(m.test, {"X": X, "Y": Y})
    y_true = np.reshape(Y, (n,))
    bounded = np.minimum(np.maximum(preds.astype(np.float64), np.ones(n) * min(y_true)), np.ones(n) * max(y_true))
    assert abs(res["RMSE"] - math.sqrt(np.mean((y_true - bounded) ** 2))) <= 1e-12
    assert (np.abs(preds) > 1).any()


def test_seq_models_negative_pools_and_device_samplers(dev, tmp_path):
    """Host samplers draw for draw against the reference's loops restated (AttRec.py:448-466 pools = all items minus
    the sample's targets; Caser.py:243,302-319 pools = neg_items[u], in the reference's storage-order alignment and in
    the intended one); sampler="device" gives negatives from the same pools without host loops."""
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    from deeprec_amd.models.seq_rec.PRME import PRME
    rng = np.random.default_rng(15)
    train, test = _seq_dataset(tmp_path, rng, nu=20, ni=35, L=3, T=2)
    ni = train.num_item
    all_items = set(np.arange(ni - 1) + 1)
    csr = train.tocsr()
    neg_ref = {u: sorted(all_items - set(csr[u].indices)) for u in range(train.num_user)}

    def run(cls, build_kw, sampler):
        m = quiet(cls, None, train.num_user, ni, 0.01, 0.01, 1, 16, sampler=sampler)
        quiet(m.build_network, 3 if cls is not PRME else 1, 2, **build_kw)
        quiet(m.prepare_data, train, test)
        got = []
        rec = lambda u_, s_, t_, n_: got.append((u_.cpu().numpy(), t_.cpu().numpy(), n_.cpu().numpy())) or torch.zeros(1)
        m.train_op = rec
        m._step = rec
        np.random.seed(3)
        if cls is PRME:
            quiet(m.train, train)
        else:
            quiet(m.train, train)
        return m, got
    # AttRec (and PRME, same code): pool = all items minus the sample's own targets
    m, got = run(AttRec, dict(num_factor=8), "reference")
    np.random.seed(3)
    idxs = np.random.permutation(len(m.sequences))
    tar = m.targets[idxs]
    want = np.zeros((len(tar), 2))
    for r in range(len(tar)):
        pool = sorted(all_items - set(int(t) for t in tar[r]))
        for j_ in range(2):
            want[r, j_] = pool[np.random.randint(len(pool))]
    negs = np.concatenate([g[2] for g in got])
    assert np.array_equal(negs, want[:len(negs)].astype(np.int32)) and len(negs) >= 32
    # Caser: pool = neg_items[u]; both alignments
    for align in ("reference", "intended"):
        m, got = run(Caser, dict(num_factor=8, n_h=2, n_v=1, num_neg=1, neg_alignment=align), "reference")
        np.random.seed(3)
        idxs = np.random.permutation(len(m.sequences))
        users = m.users[:, 0] if align == "reference" else m.users[idxs][:, 0]
        want = np.zeros((len(users), 2))
        for r, u in enumerate(users):
            for j_ in range(2):
                x = neg_ref[int(u)]
                want[r, j_] = x[np.random.randint(len(x))]
        negs = np.concatenate([g[2] for g in got])
        assert np.array_equal(negs, want[:len(negs)].astype(np.int32)), align
    # device samplers: same pools, no host loop
    for cls, kw in ((AttRec, dict(num_factor=8)), (PRME, dict(num_factor=8)),
                    (Caser, dict(num_factor=8, n_h=2, n_v=1, num_neg=1, neg_alignment="intended"))):
        m, got = run(cls, kw, "device")
        assert len(got) >= 2
        for u_, t_, n_ in got:
            assert n_.shape == (16, 2) and ((n_ >= 1) & (n_ < ni)).all()
            for r in range(16):
                if cls is Caser:
                    assert not np.isin(n_[r], csr[int(u_[r])].indices).any()
                else:
                    assert not np.isin(n_[r], t_[r]).any()
