"""Evaluation Metrics for Top N Recommendation -- drop-in for the reference's
utils/evaluation/RankingMetrics.py (same function names, arguments and printed lines).

`evaluate(model)` keeps the evaluator protocol (RankingMetrics.py:68-125: reads model.test_users,
model.neg_items[u], model.test_data[u], calls model.predict) but replaces the per-user
predict + Python sort by one full-catalog scoring + masked top-K + rank-of-target launch:
  * models that expose `eval_operands()` (bilinear / distance scores) -> dr_score_topk_f32 (MFMA GEMM, scores
    never reach HBM);
  * any other model -> model.predict over each user's candidates, selection by dr_topk_rank_f32.
It prints what the reference prints and additionally RETURNS the dict of the nine means.
"""
import math

import numpy as np
import torch

from ._ranks import masked_items, pack_csr, ranking_user_metrics, split_ranks


# efficient version
def precision_recall_ndcg_at_k(k, rankedlist, test_matrix):
    """RankingMetrics.py:21-37."""
    idcg_k = 0
    dcg_k = 0
    n_k = k if len(test_matrix) > k else len(test_matrix)
    for i in range(n_k):
        idcg_k += 1 / math.log(i + 2, 2)
    s2 = set(test_matrix)
    hits = [(idx, val) for idx, val in enumerate(rankedlist) if val in s2]
    count = len(hits)
    for c in range(count):
        dcg_k += 1 / math.log(hits[c][0] + 2, 2)
    return float(count / k), float(count / len(test_matrix)), float(dcg_k / idcg_k)


def map_mrr_ndcg(rankedlist, test_matrix):
    """RankingMetrics.py:40-65."""
    ap = 0
    map = 0
    dcg = 0
    idcg = 0
    mrr = 0
    for i in range(len(test_matrix)):
        idcg += 1 / math.log(i + 2, 2)
    s2 = set(test_matrix)
    hits = [(idx, val) for idx, val in enumerate(rankedlist) if val in s2]
    count = len(hits)
    for c in range(count):
        ap += (c + 1) / (hits[c][0] + 1)
        dcg += 1 / math.log(hits[c][0] + 2, 2)
    if count != 0:
        mrr = 1 / (hits[0][0] + 1)
    if count != 0:
        map = ap / count
    return map, mrr, float(dcg / idcg)


def rank_targets(model, users, K=10):
    """Top-K lists and held-out-item ranks for `users` (list of ids).  Returns
    (topk_idx [n,K] np.int32, ranks: list of sorted hit-rank lists, n_candidates list)."""
    from ... import kernels as Kn
    dev = torch.device(getattr(model, "device", "cuda:0"))
    num_item = int(model.num_item)
    users = list(users)
    tgt = [np.unique(np.asarray(model.test_data[u], np.int64)) for u in users]
    train_csr = getattr(model, "eval_train_items", None)
    if train_csr is not None:
        masked = [train_csr(u) for u in users]
    else:
        masked = [masked_items(model.neg_items[u], num_item) for u in users]
    tp, ti = pack_csr(masked, dev)
    gp, gi = pack_csr(tgt, dev)
    ops = model.eval_operands(users) if hasattr(model, "eval_operands") else None
    if ops is not None:
        idx, _, rank = Kn.score_topk(ops["U"], ops["I"], K, score_kind=ops.get("kind", Kn.SCORE_DOT),
                                     U2=ops.get("U2"), I2=ops.get("I2"), alpha=ops.get("alpha", 0.0),
                                     item_bias=ops.get("item_bias"), train=(tp, ti), targets=(gp, gi))
    elif hasattr(model, "eval_scores"):
        idx, _, rank = Kn.topk_rank(model.eval_scores(users), K, train=(tp, ti), targets=(gp, gi))
    else:
        # duck-typed models: predict() per user over the whole catalog, selection on the GPU
        scores = torch.empty((len(users), num_item), dtype=torch.float32, device=dev)
        all_items = list(range(num_item))
        for r, u in enumerate(users):
            s = model.predict([u] * num_item, all_items)
            scores[r] = torch.as_tensor(np.asarray(s, np.float32).reshape(-1), device=dev)
        idx, _, rank = Kn.topk_rank(scores, K, train=(tp, ti), targets=(gp, gi))
    n_cand = [num_item - len(m) for m in masked]
    return idx.cpu().numpy(), split_ranks(rank, tgt), n_cand


def evaluate(self):
    """RankingMetrics.py:68-125 (prints the same nine lines; returns them as a dict as well)."""
    users = [u for u in self.test_users]
    cols = [[] for _ in range(9)]
    if users:
        topk, ranks, _ = rank_targets(self, users, 10)
        self.pred_ratings_10 = {u: [int(i) for i in topk[r] if i >= 0] for r, u in enumerate(users)}
        for r, u in enumerate(users):
            vals = ranking_user_metrics(ranks[r], len(self.test_data[u]))
            for c, v in zip(cols, vals):
                c.append(v)
    p_at_5, r_at_5, ndcg_at_5, p_at_10, r_at_10, ndcg_at_10, map, mrr, ndcg = cols
    res = {"precision@10": np.mean(p_at_10), "recall@10": np.mean(r_at_10), "precision@5": np.mean(p_at_5),
           "recall@5": np.mean(r_at_5), "map": np.mean(map), "mrr": np.mean(mrr), "ndcg": np.mean(ndcg),
           "ndcg@5": np.mean(ndcg_at_5), "ndcg@10": np.mean(ndcg_at_10)}
    print("------------------------")
    for k, v in res.items():
        print(k + ":" + str(v))
    return {k: float(v) for k, v in res.items()}
