"""Evaluation Metrics for sequential recommendation -- drop-in for utils/evaluation/SeqRecMetrics.py.

evaluate1 (AttRec / PRME two-distance score, SeqRecMetrics.py:91-177), evaluate_caser (GEMM + item bias,
:179-248) and evaluate (per-user predict, :250-295) keep their protocol (getUserParam / getItemParam /
predict, test_users, neg_items, test_data, num_item, alpha) and printed lines; scoring, masking and ranking
run in one dr_score_rank_f32 / dr_score_rank_exact_f32 call (ranks only: no top-K list is kept).  Each returns the dict of means it prints.

Deviation kept explicit: the reference indexes `results[u, j]` with the user ID while `results` holds one
row per entry of test_users (SeqRecMetrics.py:107,141) -- only consistent when test_users == range(n).
Here row r of the user parameters belongs to test_users[r].
"""
import numpy as np
import torch

from ._ranks import (cutoff_metrics, full_list_metrics, hit_positions, masked_items, pack_csr, seq_user_metrics,
                     split_ranks)


def precision_recall_ndcg_at_k(k, rankedlist, test_matrix):
    """SeqRecMetrics.py:23-43 (the first value is the hit COUNT, not a precision)."""
    return cutoff_metrics(hit_positions(rankedlist, test_matrix), k, len(test_matrix))


def map_mrr_ndcg(rankedlist, test_matrix):
    """SeqRecMetrics.py:47-81 (the first value is AUC = (N - 1 - first hit) / (N - 1) over the N candidates)."""
    hits = hit_positions(rankedlist, test_matrix)
    _, mrr, ndcg = full_list_metrics(hits, len(test_matrix))
    _, _, auc = seq_user_metrics(hits, len(test_matrix), len(rankedlist))
    return auc, mrr, ndcg


def _t(a, dev):
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)


def _rank(self, users, U, I, kind, U2=None, I2=None, alpha=0.0, bias=None, scores=None, numpy_order=False):
    from ... import kernels as Kn
    dev = torch.device(getattr(self, "device", "cuda:0"))
    num_item = int(self.num_item)
    tgt = [np.unique(np.asarray(self.test_data[u], np.int64)) for u in users]
    masked = [masked_items(self.neg_items[u], num_item) for u in users]
    tp, ti = pack_csr(masked, dev)
    gp, gi = pack_csr(tgt, dev)
    if scores is not None:
        # HR@50 / MRR / AUC depend on the ranks of the held-out items only (SeqRecMetrics.py:33,59,75-80): no top-50
        # list is kept (K = 1), the rank counters are the same
        _, _, rank = Kn.topk_rank(scores, 1, train=(tp, ti), targets=(gp, gi))
    elif numpy_order:
        # ranks exact in the reference's own float32 operation order (SeqRecMetrics.py:122-123 is numpy arithmetic):
        # MFMA pass + exact re-score of the near-ties (dr_score_rank_exact_f32)
        rank = Kn.score_rank_exact(_t(U, dev), _t(I, dev), (gp, gi), score_kind=kind,
                                   U2=None if U2 is None else _t(U2, dev), I2=None if I2 is None else _t(I2, dev),
                                   w1=np.float32(alpha), w2=np.float32(1 - alpha),
                                   item_bias=None if bias is None else _t(bias, dev).reshape(-1), train=(tp, ti))
    else:
        # ranks only (dr_score_rank_f32): with one held-out item per user the streaming kernel keeps no list at all
        rank = Kn.score_rank(_t(U, dev), _t(I, dev), (gp, gi), score_kind=kind,
                             U2=None if U2 is None else _t(U2, dev), I2=None if I2 is None else _t(I2, dev),
                             alpha=alpha, item_bias=None if bias is None else _t(bias, dev).reshape(-1), train=(tp, ti))
    return split_ranks(rank, tgt), [num_item - len(m) for m in masked]


def _report(self, users, ranks, n_cand, with_auc):
    hr_at_50, mrr, auc = [], [], []
    for r, u in enumerate(users):
        h, m, a = seq_user_metrics(ranks[r], len(self.test_data[u]), n_cand[r], 50)
        hr_at_50.append(h)
        mrr.append(m)
        auc.append(a)
    print(np.sum(hr_at_50))
    print("------------------------")
    res = {"HR@50": float(np.mean(hr_at_50)), "MRR": float(np.mean(mrr))}
    print("HR@50:" + str(np.mean(hr_at_50)))
    print("MRR:" + str(np.mean(mrr)))
    if with_auc:
        res["AUC"] = float(np.mean(auc))
        print("AUC:" + str(np.mean(auc)))
    return res


def evaluate1(self):
    """SeqRecMetrics.py:91-177: results = -alpha*||p_u - X_i||^2 - (1-alpha)*||s_u - Q_i||^2."""
    from ... import kernels as Kn
    users = [u for u in self.test_users]
    user_factors, seq_factors, _, _ = self.getUserParam(self.test_users)
    item_factors_1, item_factors_2, _ = self.getItemParam(np.expand_dims(np.arange(self.num_item), axis=1))
    print(np.shape(user_factors))
    print(np.shape(item_factors_1))
    ranks, n_cand = _rank(self, users, user_factors, item_factors_1, Kn.SCORE_TWO_DIST, seq_factors, item_factors_2,
                          float(self.alpha), numpy_order=True)
    return _report(self, users, ranks, n_cand, True)


def evaluate_caser(self):
    """SeqRecMetrics.py:179-248: results = user_factors @ item_factors.T + bias_item."""
    from ... import kernels as Kn
    users = [u for u in self.test_users]
    user_factors = self.getUserParam(self.test_users)
    item_factors, bias_item = self.getItemParam(np.expand_dims(np.arange(self.num_item), axis=1))
    ranks, n_cand = _rank(self, users, user_factors, item_factors, Kn.SCORE_DOT, bias=bias_item)
    return _report(self, users, ranks, n_cand, False)


def evaluate(self):
    """SeqRecMetrics.py:250-295: per-user model.predict over the candidates."""
    users = [u for u in self.test_users]
    if hasattr(self, "seq_eval_operands"):           # models whose predict() is a two-distance / bilinear score
        ops = self.seq_eval_operands(users)
        ranks, n_cand = _rank(self, users, ops["U"], ops["I"], ops["kind"], ops.get("U2"), ops.get("I2"),
                              ops.get("alpha", 0.0), ops.get("item_bias"))
        return _report(self, users, ranks, n_cand, False)
    dev = torch.device(getattr(self, "device", "cuda:0"))
    num_item = int(self.num_item) if hasattr(self, "num_item") else 1 + max(max(self.neg_items[u]) for u in users)
    self.num_item = num_item
    scores = torch.empty((len(users), num_item), dtype=torch.float32, device=dev)
    all_items = list(range(num_item))
    for r, u in enumerate(users):
        s = self.predict([u] * num_item, all_items)
        scores[r] = torch.as_tensor(np.asarray(s, np.float32).reshape(-1), device=dev)
    ranks, n_cand = _rank(self, users, None, None, 0, scores=scores)
    return _report(self, users, ranks, n_cand, False)
