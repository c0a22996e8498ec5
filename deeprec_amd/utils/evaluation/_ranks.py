"""Shared machinery of the evaluators: CSR packing of the evaluator protocol's dictionaries and the
per-user metric arithmetic from target ranks.

Every ranking metric of the reference depends on the ranked list only through the 0-based positions of
the held-out items (RankingMetrics.py:31,52; SeqRecMetrics.py:33,59), so the evaluators need, per user,
the rank of each held-out item among the unmasked candidates -- which is what dr_score_topk_f32 /
dr_topk_rank_f32 return -- and never a materialised ranking.
"""
import math

import numpy as np
import torch


def pack_csr(lists, device):
    ptr = np.zeros(len(lists) + 1, np.int64)
    ptr[1:] = np.cumsum([len(l) for l in lists])
    if ptr[-1] >= 2 ** 31:
        raise ValueError("CSR too large for int32 offsets")
    idx = np.concatenate([np.asarray(l, np.int64).reshape(-1) for l in lists]) if ptr[-1] else np.zeros(0, np.int64)
    return (torch.from_numpy(ptr.astype(np.int32)).to(device), torch.from_numpy(idx.astype(np.int32)).to(device))


def masked_items(neg_items_u, num_item):
    """Items NOT in the candidate list neg_items[u] (ascending) -- the evaluator protocol hands over the
    candidates (RankingMetrics.py:84), the kernels want the complement."""
    m = np.ones(num_item, bool)
    m[np.asarray(neg_items_u, np.int64)] = False
    return np.flatnonzero(m)


def _idcg(n):
    s = 0
    for i in range(n):
        s += 1 / math.log(i + 2, 2)
    return s


def hit_positions(rankedlist, test_items):
    """Sorted 0-based positions in `rankedlist` of the entries that are held-out items."""
    wanted = set(test_items)
    return [pos for pos, item in enumerate(rankedlist) if item in wanted]


def _dcg(positions):
    s = 0
    for r in positions:
        s += 1 / math.log(r + 2, 2)
    return s


def cutoff_metrics(hit_ranks, k, n_test):
    """(hits within the list, recall, ndcg) for a list that was cut at k: every entry of hit_ranks is < k.
    precision = hits / k (RankingMetrics.py:37); the sequence evaluator reports the hit count itself
    (SeqRecMetrics.py:43)."""
    return len(hit_ranks), float(len(hit_ranks) / n_test), float(_dcg(hit_ranks) / _idcg(k if n_test > k else n_test))


def full_list_metrics(hit_ranks, n_test):
    """(average precision, reciprocal rank, ndcg) over the whole ranking (RankingMetrics.py:40-65)."""
    ap = 0
    for c, r in enumerate(hit_ranks):
        ap += (c + 1) / (r + 1)
    mp = ap / len(hit_ranks) if hit_ranks else 0
    mrr = 1 / (hit_ranks[0] + 1) if hit_ranks else 0
    return mp, mrr, float(_dcg(hit_ranks) / _idcg(n_test))


def ranking_user_metrics(hit_ranks, n_test):
    """(p5, r5, ndcg5, p10, r10, ndcg10, map, mrr, ndcg) from sorted 0-based hit ranks
    (RankingMetrics.py:21-65)."""
    out = []
    for k in (5, 10):
        hk = [r for r in hit_ranks if r < k]
        dcg = 0
        for r in hk:
            dcg += 1 / math.log(r + 2, 2)
        out += [float(len(hk) / k), float(len(hk) / n_test), float(dcg / _idcg(k if n_test > k else n_test))]
    ap = 0
    dcg = 0
    for c, r in enumerate(hit_ranks):
        ap += (c + 1) / (r + 1)
        dcg += 1 / math.log(r + 2, 2)
    mp = ap / len(hit_ranks) if hit_ranks else 0
    mrr = 1 / (hit_ranks[0] + 1) if hit_ranks else 0
    out += [mp, mrr, float(dcg / _idcg(n_test))]
    return out


def seq_user_metrics(hit_ranks, n_test, n_candidates, k=50):
    """(hit count @k, mrr, auc) (SeqRecMetrics.py:23-43, 47-81)."""
    hr = len([r for r in hit_ranks if r < k])
    mrr = 1 / (hit_ranks[0] + 1) if hit_ranks else 0
    mx = n_candidates - 1
    count_test = mx - hit_ranks[0] if hit_ranks else 0
    auc = 1.0 * count_test / mx
    return hr, mrr, auc


def split_ranks(rank, tgt_lists):
    """Per-user sorted lists of valid (>= 0) ranks from the flat kernel output."""
    rank = rank.cpu().numpy() if isinstance(rank, torch.Tensor) else np.asarray(rank)
    out, off = [], 0
    for t in tgt_lists:
        r = rank[off:off + len(t)]
        out.append(sorted(int(x) for x in r if x >= 0))
        off += len(t)
    return out
