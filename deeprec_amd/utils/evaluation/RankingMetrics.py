"""Evaluation Metrics for Top N Recommendation -- drop-in for the reference's
utils/evaluation/RankingMetrics.py (same function names, arguments and printed lines).

`evaluate(model)` keeps the evaluator protocol (RankingMetrics.py:68-125: reads model.test_users,
model.neg_items[u], model.test_data[u], calls model.predict) but replaces the per-user
predict + Python sort by one full-catalog scoring + masked top-K + rank-of-target launch:
  * models that expose `eval_operands()` (bilinear / distance scores) -> dr_score_topk_f32 (MFMA GEMM, scores
    never reach HBM);
  * models that expose `eval_score_chunks()` (NeuMF's tower) -> dr_tower_catalog_scores_f32 per user chunk,
    selection by dr_topk_rank_f32 per chunk;
  * any other model -> model.predict over each user's candidates, selection by dr_topk_rank_f32.
It prints what the reference prints and additionally RETURNS the dict of the nine means.
"""
import numpy as np
import torch

from ._ranks import (cutoff_metrics, full_list_metrics, hit_positions, masked_items, pack_csr, ranking_user_metrics,
                     split_ranks)


def precision_recall_ndcg_at_k(k, rankedlist, test_matrix):
    """RankingMetrics.py:21-37 -- `rankedlist` is the ranking already cut at k."""
    hits, recall, ndcg = cutoff_metrics(hit_positions(rankedlist, test_matrix), k, len(test_matrix))
    return float(hits / k), recall, ndcg


def map_mrr_ndcg(rankedlist, test_matrix):
    """RankingMetrics.py:40-65 over the full ranking."""
    return full_list_metrics(hit_positions(rankedlist, test_matrix), len(test_matrix))


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
        if ops.get("order") == "numpy" and gi.numel():
            # the model's predict() is numpy float32 arithmetic (pairwise-summed products / squared differences):
            # ranks are made exact with respect to THAT operation order (dr_score_rank_exact_f32)
            a = float(ops.get("alpha", 0.0))
            rank = Kn.score_rank_exact(ops["U"], ops["I"], (gp, gi), score_kind=ops.get("kind", Kn.SCORE_DOT),
                                       U2=ops.get("U2"), I2=ops.get("I2"), w1=np.float32(a), w2=np.float32(1 - a),
                                       item_bias=ops.get("item_bias"), train=(tp, ti))
    elif hasattr(model, "eval_score_chunks"):
        # scores that are no bilinear form (NeuMF's tower): the model yields [chunk, num_item] score blocks, the
        # selection runs per block -- the full [users, items] matrix never exists
        idxs, ranks = [], []
        for c0, c1, scores in model.eval_score_chunks(users):
            i_, _, r_ = Kn.topk_rank(scores, K, train=pack_csr(masked[c0:c1], dev), targets=pack_csr(tgt[c0:c1], dev))
            idxs.append(i_)
            ranks.append(r_)
        idx, rank = torch.cat(idxs), torch.cat(ranks)
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
