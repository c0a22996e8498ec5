"""CPU tests: the oracle and the host-side mirrors (metrics, evaluator arithmetic, loaders) against the
golden vectors generated from the reference's own numpy modules (tools/make_golden.py)."""
import io
import json
import os
import contextlib

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import ref_ops as ro

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_metric_functions_oracle_and_product():
    from deeprec_amd.utils.evaluation import RankingMetrics as RM, RatingMetrics as RT, SeqRecMetrics as SM
    from deeprec_amd.utils.evaluation._ranks import ranking_user_metrics, seq_user_metrics
    g = load("metrics_functions.json")
    assert len(g["cases"]) == 200
    for c in g["cases"]:
        k, ranked, test = c["k"], c["ranked"], c["test"]
        for mod_prn, mod_mmn, key_prn, key_mmn in ((ro.precision_recall_ndcg_at_k, ro.map_mrr_ndcg, "rank_prn", "rank_mmn"),
                                                   (RM.precision_recall_ndcg_at_k, RM.map_mrr_ndcg, "rank_prn", "rank_mmn"),
                                                   (ro.seq_precision_recall_ndcg_at_k, ro.seq_map_mrr_ndcg, "seq_prn", "seq_mmn"),
                                                   (SM.precision_recall_ndcg_at_k, SM.map_mrr_ndcg, "seq_prn", "seq_mmn")):
            np.testing.assert_allclose(mod_prn(k, ranked[:k], test), c[key_prn], rtol=1e-13, atol=0)
            np.testing.assert_allclose(mod_mmn(ranked, test), c[key_mmn], rtol=1e-13, atol=0)
        # the rank-only formulation used by the GPU evaluators
        hits = sorted(i for i, v in enumerate(ranked) if v in set(test))
        vals = ranking_user_metrics(hits, len(test))
        np.testing.assert_allclose(vals[6:], c["rank_mmn"], rtol=1e-13)
        if k in (5, 10):
            np.testing.assert_allclose(vals[0:3] if k == 5 else vals[3:6], c["rank_prn"], rtol=1e-13)
        d = ro.ranking_metrics_from_ranks(hits, len(test))
        np.testing.assert_allclose([d["map"], d["mrr"], d["ndcg"]], c["rank_mmn"], rtol=1e-13)
        hr, mrr, auc = seq_user_metrics(hits, len(test), len(ranked), k)
        assert hr == c["seq_prn"][0]
        np.testing.assert_allclose([auc, mrr], c["seq_mmn"][:2], rtol=1e-13)
    for r in g["rating"]:
        assert RT.RMSE(r["err"], r["n"]) == r["rmse"] and RT.MAE(r["err"], r["n"]) == r["mae"]
        assert ro.RMSE(r["err"], r["n"]) == r["rmse"] and ro.MAE(r["err"], r["n"]) == r["mae"]


def test_numpy_pairwise_restatement_equals_numpy():
    """oracle.ref_ops.np_pairwise_sum (the order dr_score_rank_exact_f32's re-score follows) against numpy itself, the
    arithmetic of SeqRecMetrics.py:122-123, on every length class of the algorithm."""
    rng = np.random.default_rng(3)
    for d in (1, 5, 7, 8, 9, 16, 17, 40, 127, 128, 129, 150, 200, 256, 300, 513, 1100):
        a = (rng.standard_normal((3, d)) * 10 ** rng.uniform(-3, 3)).astype(np.float32)
        b = rng.standard_normal((4, d)).astype(np.float32)
        R = np.sum((a[:, None] - b) ** 2, axis=2)
        Dt = np.sum(a[:, None] * b, axis=2)
        for u in range(3):
            for i in range(4):
                t = a[u] - b[i]
                assert ro.np_pairwise_sum(t * t) == R[u, i], d
                assert ro.np_pairwise_sum(a[u] * b[i]) == Dt[u, i], d
                assert ro.np_pairwise_sum(a[u] * b[i]) == (a[u] * b[i]).sum(), d
    # numpy_order_scores is the reference's expression itself
    P, S = rng.standard_normal((5, 40)).astype(np.float32), rng.standard_normal((5, 40)).astype(np.float32)
    X, Q = rng.standard_normal((9, 40)).astype(np.float32), rng.standard_normal((9, 40)).astype(np.float32)
    alpha = 0.2
    ref = -alpha * np.sum((P[:, None] - X) ** 2, axis=2) - (1 - alpha) * np.sum((S[:, None] - Q) ** 2, axis=2)
    got = ro.numpy_order_scores(2, P, X, S, Q, np.float32(alpha), np.float32(1 - alpha))
    assert ref.dtype == np.float32 and np.array_equal(ref, got)


def _neg(train, ni):
    return {u: [i for i in range(ni) if i not in set(t)] for u, t in enumerate(train)}


def test_oracle_ranking_evaluate_matches_reference_printout():
    for case in load("ranking_evaluate.json"):
        P, Q = np.array(case["P"], np.float32), np.array(case["Q"], np.float32)
        test = {int(u): v for u, v in case["test"].items()}
        neg = _neg(case["train"], Q.shape[0])
        users = [u for u in test if len(test[u]) > 0]
        pred = lambda us, its: (P[np.array(us)] * Q[np.array(its)]).sum(1).astype(np.float32)
        res, _ = ro.evaluate_ranking(pred, users, neg, test)
        for k, v in case["printed"].items():
            assert abs(res[k] - v) <= 1e-12, (k, res[k], v)
        # rank-of-target formulation == full-sort formulation (what the GPU evaluator relies on)
        acc = {k: [] for k in res}
        for u in users:
            s = pred([u] * Q.shape[0], list(range(Q.shape[0])))
            _, _, ranks = ro.topk_and_ranks(s, case["train"][u], test[u], 10)
            d = ro.ranking_metrics_from_ranks(ranks, len(test[u]))
            for k in acc:
                acc[k].append(d[k])
        for k, v in case["printed"].items():
            assert abs(np.mean(acc[k]) - v) <= 1e-12, k


def test_oracle_seq_evaluate_matches_reference_printout():
    from deeprec_amd.utils.evaluation._ranks import seq_user_metrics
    for case in load("seq_evaluate.json"):
        P, S, X, Q = (np.array(case[k], np.float32) for k in ("P", "S", "X", "Q"))
        bias = np.array(case["bias"], np.float32)
        test = {int(u): v for u, v in case["test"].items()}
        users = [u for u in test if len(test[u]) > 0]
        ni = Q.shape[0]
        a = case["alpha"]
        score_sets = {
            "evaluate1": -a * np.sum((P[:, None] - X) ** 2, axis=2) - (1 - a) * np.sum((S[:, None] - Q) ** 2, axis=2),
            "evaluate_caser": np.dot(P, Q.T) + bias,
            "evaluate": np.stack([(P[u] * Q).sum(1).astype(np.float32) for u in range(P.shape[0])]),
        }
        for name, scores in score_sets.items():
            hr, mrr, auc = [], [], []
            for u in users:
                _, _, ranks = ro.topk_and_ranks(scores[u], case["train"][u], test[u], 50)
                h, m, au = seq_user_metrics(sorted(int(r) for r in ranks if r >= 0), len(test[u]),
                                            ni - len(case["train"][u]), 50)
                hr.append(h), mrr.append(m), auc.append(au)
            pr = case[name]
            assert abs(np.mean(hr) - pr["HR@50"]) <= 1e-12 and abs(np.mean(mrr) - pr["MRR"]) <= 1e-12, name
            if "AUC" in pr:
                assert abs(np.mean(auc) - pr["AUC"]) <= 1e-12


def test_loaders_match_reference(tmp_path):
    from deeprec_amd.utils.load_data import load_data_content as LC, load_data_ranking as LR, load_data_rating as LT
    from deeprec_amd.utils.load_data import load_data_seq as LS
    g = load("loaders.json")
    p = tmp_path / "log.dat"
    p.write_text(g["log"])
    with contextlib.redirect_stdout(io.StringIO()):
        np.random.seed(g["neg"]["seed"])
        train, test, nu, ni = LR.load_data_neg(path=str(p), test_size=0.2, sep="\t")
        np.random.seed(g["rating"]["seed"])
        tr2, te2, _, _ = LT.load_data_rating(path=str(p), test_size=0.1, sep="\t")
    assert (nu, ni) == (g["neg"]["n_users"], g["neg"]["n_items"])
    coo = train.tocoo()
    assert sorted(zip(coo.row.tolist(), coo.col.tolist(), coo.data.tolist())) == \
        sorted(zip(g["neg"]["train_row"], g["neg"]["train_col"], g["neg"]["train_val"]))
    for u, v in g["neg"]["test"].items():
        assert np.asarray(test[int(u)]).tolist() == v
    for m, key in ((tr2, "train"), (te2, "test")):
        c = m.tocoo()
        assert sorted(zip(c.row.tolist(), c.col.tolist(), c.data.tolist())) == sorted(zip(*g["rating"][key]))
    ps = tmp_path / "seq.txt"
    ps.write_text(g["seq"]["text"])
    with contextlib.redirect_stdout(io.StringIO()):
        ds = LS.DataSet(path=str(ps), sep=" ", seq_len=g["seq"]["L"], target_len=g["seq"]["T"], isTrain=True)
    assert (ds.num_user, ds.num_item) == (g["seq"]["num_user"], g["seq"]["num_item"])
    assert ds.sequences.sequences.tolist() == g["seq"]["sequences"]
    assert ds.sequences.targets.tolist() == g["seq"]["targets"]
    assert ds.sequences.user_ids.tolist() == g["seq"]["users"]
    assert ds.test_sequences.sequences.tolist() == g["seq"]["test_sequences"]
    assert {str(k): v for k, v in ds.item_map.items()} == g["seq"]["item_map"]
    pf = tmp_path / "x.libfm"
    pf.write_text(g["libfm"]["text"])
    LC.features.clear()
    LC.count_num_feature_field(str(pf))
    d = LC.read_data(str(pf))
    assert dict(LC.features) == g["libfm"]["features"] and d["X"] == g["libfm"]["X"] and d["Y"] == g["libfm"]["Y"]


def test_oracle_internal_consistency():
    """Things no reference fixture pins (TensorFlow arithmetic, 'parity unpinned'): self-consistency of the
    restatement -- the C fmaf-chain scores agree with numpy to fp32 accuracy, the blocked sum is a sum,
    the sparse dense-mode update equals the dense formula, the scalar-C baseline step equals the numpy step."""
    rng = np.random.default_rng(0)
    U, I = rng.standard_normal((7, 30)).astype(np.float32), rng.standard_normal((11, 30)).astype(np.float32)
    np.testing.assert_allclose(co.score_matrix(U, I, 0), U.astype(np.float64) @ I.T.astype(np.float64), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(co.score_matrix(U, I, 1), -((U[:, None].astype(np.float64) - I[None]) ** 2).sum(2), rtol=1e-5)
    rows = rng.standard_normal((100, 5)).astype(np.float32)
    np.testing.assert_allclose(ro.blocked_sum(rows), rows.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    W = rng.standard_normal((20, 4)).astype(np.float32)
    ids = rng.integers(0, 20, 50)
    src = rng.standard_normal((50, 4)).astype(np.float32)
    Wd, _ = ro.sparse_row_update(ro.OPT_SGD, W, [], ids, src, None, None, 0.1, ro.REG_DENSE, {"lr": 0.5})
    G = np.zeros_like(W)
    np.add.at(G, ids, src)
    np.testing.assert_allclose(Wd, W - 0.5 * (G + 0.1 * W), rtol=1e-5, atol=1e-6)
    # C baseline step vs numpy oracle step (touched Adam)
    nu, ni, k, B = 50, 40, 8, 64
    P = (rng.standard_normal((nu, k)) * 0.1).astype(np.float32)
    Q = (rng.standard_normal((ni, k)) * 0.1).astype(np.float32)
    u, i, j = (rng.integers(0, n, B).astype(np.int32) for n in (nu, ni, ni))
    hp = dict(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-7)
    Pr, Qr, _, _, loss, _ = ro.pairwise_step(P, Q, [np.zeros_like(P)] * 2, [np.zeros_like(Q)] * 2, u, i, j,
                                             ro.OPT_ADAM, hp, 0.1, ro.REG_TOUCHED)
    Pc, Qc = P.copy(), Q.copy()
    lc = co.bpr_step(Pc, Qc, [np.zeros_like(P), np.zeros_like(P)], [np.zeros_like(Q), np.zeros_like(Q)], u, i, j, True,
                     alpha=1e-3, eps=1e-7, reg=0.1)
    assert abs(lc - loss) <= 1e-5 * abs(loss)
    np.testing.assert_allclose(Pc, Pr, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(Qc, Qr, rtol=1e-4, atol=1e-6)


def test_caser_convs_equal_conv2d():
    """deeprec_amd/models/seq_rec/Caser.py:caser_convs (GEMM form) against the literal conv2d form of
    Caser.py:118-172, forward and gradients, on CPU."""
    import torch
    import torch.nn.functional as F
    from deeprec_amd.models.seq_rec.Caser import caser_convs
    torch.manual_seed(0)
    B, L, d, n_v, n_h = 7, 6, 10, 3, 4
    e = torch.randn(B, L, d, dtype=torch.float64, requires_grad=True)
    vw = torch.randn(n_v, 1, L, 1, dtype=torch.float64, requires_grad=True)
    vb = torch.randn(n_v, dtype=torch.float64, requires_grad=True)
    hw = [torch.randn(n_h, 1, h, d, dtype=torch.float64, requires_grad=True) for h in range(1, L + 1)]
    hb = [torch.randn(n_h, dtype=torch.float64, requires_grad=True) for _ in range(L)]

    def literal():
        x = e[:, None]
        ov = F.conv2d(x, vw, vb)[:, :, 0, :].permute(0, 2, 1).reshape(B, n_v * d)
        outs = [ov]
        for w, b in zip(hw, hb):
            outs.append(torch.relu(F.conv2d(x, w, b)[:, :, :, 0]).max(2)[0])
        return torch.cat(outs, 1)

    params = [e, vw, vb] + hw + hb
    a = caser_convs(e, vw, vb, hw, hb)
    b = literal()
    assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    coef = torch.randn_like(a)
    ga = torch.autograd.grad((a * coef).sum(), params)
    gb = torch.autograd.grad((b * coef).sum(), params)
    for x, y in zip(ga, gb):
        assert torch.allclose(x, y, rtol=1e-10, atol=1e-12)


def test_attention_oracle_against_autograd():
    """oracle/ref_ops.attn_pool_forward / attn_pool_backward (AttRec.py:147-186) against the literal graph under torch
    autograd in float64, and the oracle's timing signal against the product's (AttRec.py:468-510)."""
    import torch
    from oracle import ref_ops as ro
    from deeprec_amd.models.seq_rec.AttRec import timing_signal
    rng = np.random.default_rng(5)
    for B, L, d in ((3, 5, 16), (2, 50, 128), (4, 1, 7), (2, 7, 150)):
        pre = rng.normal(0, 1, (B, L, d))
        v = rng.normal(0, 1, (B, L, d))
        gout = rng.normal(0, 1, (B, d))
        tp = torch.tensor(pre, dtype=torch.float64, requires_grad=True)
        tv = torch.tensor(v, dtype=torch.float64, requires_grad=True)
        q = torch.relu(tp)
        w = torch.softmax(q @ q.transpose(1, 2) / np.sqrt(d), -1) * (1 - torch.eye(L, dtype=torch.float64))
        out = (w @ tv).mean(1)
        out.backward(torch.tensor(gout))
        o_out, o_s, o_w = ro.attn_pool_forward(pre, v, np.float64)
        np.testing.assert_allclose(o_out, out.detach().numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(o_w, w.detach().numpy(), rtol=1e-12, atol=1e-14)
        dpre, dv = ro.attn_pool_backward(pre, v, gout)
        np.testing.assert_allclose(dpre, tp.grad.numpy(), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(dv, tv.grad.numpy(), rtol=1e-10, atol=1e-13)
        if d > 2:
            np.testing.assert_allclose(ro.timing_signal(L, d), timing_signal(L, d, "cpu")[0].numpy(), rtol=0, atol=1e-5)
