"""Oracle-backed stand-in for the deeprec_amd.kernels namespace, CPU tensors only.

Lives under tests/ on purpose: it lets the row-sharding *exchange logic* (deeprec_amd/sharding.py)
run under gloo without a GPU.  The product never imports it.
"""
import numpy as np
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro


def _n(t):
    return None if t is None else t.detach().numpy()


def gather(table, ids, oob="zero"):
    return torch.from_numpy(ro.gather(_n(table), _n(ids), oob))


def bucket_order(keys, nkeys):
    kk = _n(keys).astype(np.int64)
    order = np.argsort(kk, kind="stable").astype(np.int32)
    return torch.from_numpy(order), torch.from_numpy(np.bincount(kk, minlength=nkeys).astype(np.int64))


class RouteViews:
    pass


def route_views(ids, me, num_user, num_item, rows_per_owner):
    """numpy restatement of dr_route_views_i32 (include/deeprec_amd.h): both roles of rank `me` from the replicated
    id lists [world, 3, B].  Unused tails of the outputs are zero here (unspecified in the library)."""
    g = _n(ids).astype(np.int64)
    W, _, B = g.shape
    WB = W * B
    u, i, j = g[:, 0, :].reshape(-1), g[:, 1, :].reshape(-1), g[:, 2, :].reshape(-1)
    ok = (u >= 0) & (u < num_user) & (i >= 0) & (i < num_item) & (j >= 0) & (j < num_item)
    item = np.concatenate((i, j))                    # list entry e: part = e // WB, sample e % WB
    uu = np.concatenate((u, u))
    ok2 = np.concatenate((ok, ok))
    r = RouteViews()
    n = 2 * WB
    r.n_list = n
    out = {name: np.zeros((2, max(n, 1)), np.int32) for name in ("rows", "inv", "order", "perm", "uniq")}
    u_loc = np.zeros(max(WB, 1), np.int32)
    counts = np.zeros(4 * W + 3, np.int64)
    for v in range(2):
        mine = ok2 & ((item % W == me) if v else (uu % W == me))
        e = np.nonzero(mine)[0]                      # ascending e = canonical order; c = index into e
        major = (uu[e] % W) if v else (item[e] % W)
        row = item[e] // W
        key = major * rows_per_owner + row
        order = np.argsort(key, kind="stable")       # y -> c
        sk = key[order]
        head = np.ones(len(e), dtype=bool)
        head[1:] = sk[1:] != sk[:-1]
        seg = np.cumsum(head) - 1
        m = len(e)
        out["rows"][v, :m] = row
        out["order"][v, :m] = order
        out["perm"][v, order] = np.arange(m)
        out["inv"][v, order] = seg
        out["uniq"][v, :int(head.sum())] = sk[head] % rows_per_owner
        counts[2 * v * W:2 * (v + 1) * W:2] = np.bincount(sk[head] // rows_per_owner, minlength=W)
        counts[2 * v * W + 1:2 * (v + 1) * W:2] = np.bincount(major, minlength=W)
        counts[4 * W + v] = m
        if v == 0:
            first = e[e < WB]
            u_loc[:len(first)] = u[first] // W
    counts[4 * W + 2] = int((~ok).sum())
    for name, a in out.items():
        setattr(r, name, torch.from_numpy(a))
    r.u_loc = torch.from_numpy(u_loc)
    r.counts = torch.from_numpy(counts)
    return r


def bpr_item_grads(P, u_loc, g, order):
    """numpy restatement of dr_bpr_item_grads_f32."""
    P, u_loc, g, order = _n(P), _n(u_loc), _n(g), _n(order)
    n = len(u_loc)
    s = order % n if n else order
    coef = np.where(order < n, g[s], -g[s]).astype(np.float32)
    return torch.from_numpy((coef[:, None] * P[u_loc[s]]).astype(np.float32).reshape(2 * n, P.shape[1]))


def bpr_forward(P, Q, u, i, j):
    lb, g, _ = ro.bpr_forward(_n(P), _n(Q), _n(u), _n(i), _n(j))
    stash = (g[:, None] * (ro.gather(_n(Q), _n(i)) - ro.gather(_n(Q), _n(j)))).astype(np.float32)
    return torch.from_numpy(g), torch.from_numpy(lb), torch.from_numpy(stash)


def scale_rows(src, coef, idx=None):
    s = _n(src) if idx is None else _n(src)[_n(idx)]
    return torch.from_numpy((_n(coef)[:, None] * s[:len(_n(coef))]).astype(np.float32))


def _hp(opt):
    return dict(lr=opt.lr, alpha=opt.alpha, beta1=opt.beta1, beta2=opt.beta2, eps=opt.eps, rho=opt.rho,
                momentum=opt.momentum)


def sparse_row_update(table, slot0, slot1, ids, src, opt, src_idx=None, coef=None, plan=None):
    slots = [s for s in (slot0, slot1) if s is not None]
    n_slots = {ro.OPT_SGD: 0, ro.OPT_ADAM: 2, ro.OPT_ADAGRAD: 1, ro.OPT_RMSPROP: 2}[opt.kind]
    W, sn = ro.sparse_row_update(opt.kind, _n(table), [_n(s) for s in slots[:n_slots]], _n(ids), _n(src),
                                 _n(src_idx), _n(coef), opt.reg, opt.reg_mode, _hp(opt))
    table.copy_(torch.from_numpy(W))
    for s, v in zip(slots, sn):
        s.copy_(torch.from_numpy(v))
    return table


def sum_f32(x):
    return torch.tensor([float(np.sum(_n(x).astype(np.float64)))], dtype=torch.float32)


def init_normal_(w, seed, stddev, offset=0):
    full = co.init_normal(offset + w.numel(), seed, stddev)[offset:]
    w.copy_(torch.from_numpy(full).reshape(w.shape))
    return w


def score_topk(U, I, K, train=None, targets=None):
    """Oracle-backed stand-in for kernels.score_topk (dot scores): oracle/c_oracle.py:score_matrix +
    oracle/ref_ops.py:topk_and_ranks per user."""
    S = co.score_matrix(_n(U), _n(I), 0)
    Nu = S.shape[0]
    tp, ti = (_n(train[0]), _n(train[1])) if train is not None else (np.zeros(Nu + 1, np.int32), np.zeros(0, np.int32))
    gp, gi = (_n(targets[0]), _n(targets[1])) if targets is not None else (np.zeros(Nu + 1, np.int32), np.zeros(0, np.int32))
    idx = np.zeros((Nu, K), np.int32)
    val = np.zeros((Nu, K), np.float32)
    ranks = np.zeros(len(gi), np.int32)
    for u in range(Nu):
        top, tops, rk = ro.topk_and_ranks(S[u], ti[tp[u]:tp[u + 1]], gi[gp[u]:gp[u + 1]], K)
        idx[u, :len(top)], val[u, :len(top)] = top, tops
        ranks[gp[u]:gp[u + 1]] = rk
    return torch.from_numpy(idx), torch.from_numpy(val), (torch.from_numpy(ranks) if targets is not None else None)
