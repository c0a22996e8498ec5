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
    return torch.from_numpy(order), torch.from_numpy(np.bincount(kk, minlength=nkeys)[:nkeys].astype(np.int64))


def step_route_keys(ids_all, me, list_start):
    """dr_step_route_keys_i32 (include/deeprec_amd.h) restated with numpy."""
    g = _n(ids_all).astype(np.int64)
    W, n_tot = g.shape
    nl = len(list_start) - 1
    lst = np.searchsorted(np.asarray(list_start[1:], np.int64), np.arange(n_tot), side="right")
    owner = g % W
    src = np.repeat(np.arange(W), n_tot).reshape(W, n_tot)
    ko = np.where(owner == me, lst[None, :] * W + src, nl * W)
    km = lst * W + owner[me]
    return (torch.from_numpy(km.astype(np.int32)), torch.from_numpy(ko.reshape(-1).astype(np.int32)),
            torch.from_numpy((g // W).reshape(-1).astype(np.int32)))


def _wire_rows(n, k, seg_start, seg_off):
    """Float offset of every entry's row in the wire buffer."""
    e = np.arange(n)
    s = np.searchsorted(np.asarray(seg_start[1:], np.int64), e, side="right")
    return np.asarray(seg_off, np.int64)[s] + (e - np.asarray(seg_start, np.int64)[s]) * k


def rows_pack(src, idx, n, seg_start, seg_off, flat, idx_bias=0):
    sn = _n(src).reshape(src.shape[0], -1)
    k = sn.shape[1]
    r = np.arange(n) if idx is None else _n(idx).astype(np.int64) - idx_bias
    rows = np.zeros((n, k), np.float32)
    ok = (r >= 0) & (r < sn.shape[0])
    rows[ok] = sn[r[ok]]
    out = _n(flat)
    for e, w in enumerate(_wire_rows(n, k, seg_start, seg_off)):
        out[w:w + k] = rows[e]


def rows_unpack(flat, idx, n, seg_start, seg_off, dst, idx_bias=0):
    dn = _n(dst).reshape(dst.shape[0], -1)
    k = dn.shape[1]
    r = np.arange(n) if idx is None else _n(idx).astype(np.int64) - idx_bias
    fl = _n(flat)
    for e, w in enumerate(_wire_rows(n, k, seg_start, seg_off)):
        if 0 <= r[e] < dn.shape[0]:
            dn[r[e]] = fl[w:w + k]


def shard_caps(world, B):
    return min(world * B, max(2 * B, B + 4096)), min(2 * world * B, max(4 * B, 2 * B + 8192))


class ShardWorkspace:
    """numpy restatement of the row-sharded step's workspace (deeprec_amd/kernels.py:ShardWorkspace, csrc/shard.hip):
    the same lists, as attributes, so a GPU test can compare them with the kernel's."""

    def __init__(self, world, me, B, k, num_user, num_item, device="cpu", cap_s=None, cap_g=None):
        d_s, d_g = shard_caps(world, B)
        self.world, self.me, self.B, self.k = world, me, B, k
        self.num_user, self.num_item = num_user, num_item
        self.cap_s, self.cap_g = int(cap_s or d_s), int(cap_g or d_g)
        self.counts = torch.zeros(8 + 6 * world, dtype=torch.int32)
        self.serve = torch.zeros(self.cap_g, dtype=torch.int32)
        self.loss = torch.zeros(1, dtype=torch.float32)


def shard_route(ws, ids=None, uij=None, counts_host=None, stream=None):
    """dr_shard_bpr_route (include/deeprec_amd.h): both roles of rank ws.me from the replicated id lists."""
    if ids is None:
        ids = torch.stack([t.reshape(-1) for t in uij]).reshape(1, 3, -1)
    g = _n(ids).astype(np.int64)
    W, me, B = ws.world, ws.me, ws.B
    WB = W * B
    nu, ni = ws.num_user, ws.num_item
    nuL, niL = (nu - me + W - 1) // W, (ni - me + W - 1) // W
    rpo = (ni + W - 1) // W
    u, i, j = g[:, 0, :].reshape(-1), g[:, 1, :].reshape(-1), g[:, 2, :].reshape(-1)
    ok = (u >= 0) & (u < nu) & (i >= 0) & (i < ni) & (j >= 0) & (j < ni)
    c = np.zeros(8 + 6 * W, np.int64)
    c[4] = int((~ok).sum())
    p = np.where(ok, u % W, -1)
    S = np.nonzero(p == me)[0]                                   # processed samples, global batch order
    n_s = len(S)
    s_of = np.full(WB, -1, np.int64)
    s_of[S] = np.arange(n_s)
    item = np.concatenate((i, j))                                # entry e: part = e // WB, sample e % WB
    gid = np.concatenate((np.arange(WB), np.arange(WB)))
    part = np.concatenate((np.zeros(WB, np.int64), np.ones(WB, np.int64)))
    ok2, pe = ok[gid], p[gid]
    own = np.where(ok2, item % W, -1)
    row = item // W
    E = np.nonzero(own == me)[0]                                 # owned entries, canonical order
    n_g = len(E)
    ws.n_s, ws.n_g = n_s, n_g
    ws.u_loc = (u[S] // W).astype(np.int32)
    ws.okey = row[E].astype(np.int32)                            # local item row of owned entry c (list position n_s + c)
    ws.osrc = np.zeros(n_g, np.int64)
    ws.ref = np.zeros((2, n_s), np.int64)
    ws.ow = np.zeros((2, n_s), np.int64)
    loc = pe[E] == me
    ws.osrc[loc] = 2 * s_of[gid[E[loc]]] + part[E[loc]]
    ws.ref[part[E[loc]], s_of[gid[E[loc]]]] = row[E[loc]]
    ws.ow[part[E[loc]], s_of[gid[E[loc]]]] = n_s + np.nonzero(loc)[0]
    # view 0: I process the sample, a peer owns the row -- sorted by (owner, row), ties in canonical order
    V0 = np.nonzero((pe == me) & (own != me) & ok2)[0]
    k0 = own[V0] * rpo + row[V0]
    o0 = np.argsort(k0, kind="stable")
    sk0 = k0[o0]
    head0 = np.ones(len(V0), bool)
    head0[1:] = sk0[1:] != sk0[:-1]
    seg0 = np.cumsum(head0) - 1
    ws.ref[part[V0[o0]], s_of[gid[V0[o0]]]] = niL + seg0
    ws.ow[part[V0[o0]], s_of[gid[V0[o0]]]] = -1 - np.arange(len(V0))
    c[8:8 + W] = np.bincount(sk0[head0] // rpo, minlength=W)
    c[8 + W:8 + 2 * W] = np.bincount(sk0 // rpo, minlength=W)
    # view 1: I own the row, a peer processes the sample -- sorted by (processing rank, row)
    m1 = ~loc
    V1 = E[m1]
    k1 = pe[V1] * rpo + row[V1]
    o1 = np.argsort(k1, kind="stable")
    sk1 = k1[o1]
    head1 = np.ones(len(V1), bool)
    head1[1:] = sk1[1:] != sk1[:-1]
    ws.osrc[np.nonzero(m1)[0][o1]] = -1 - np.arange(len(V1))
    serve = (sk1[head1] % rpo).astype(np.int32)
    ws.n_serve = len(serve)
    sv = np.zeros(ws.cap_g, np.int32)
    sv[:len(serve)] = serve[:ws.cap_g]
    ws.serve = torch.from_numpy(sv)
    c[8 + 2 * W:8 + 3 * W] = np.bincount(sk1[head1] // rpo, minlength=W)
    c[8 + 3 * W:8 + 4 * W] = np.bincount(sk1 // rpo, minlength=W)
    hp = np.bincount(p[ok], minlength=W)
    ho = np.bincount(np.concatenate((i[ok] % W, j[ok] % W)), minlength=W)
    c[8 + 4 * W:8 + 5 * W], c[8 + 5 * W:8 + 6 * W] = hp, ho
    c[0], c[1], c[2], c[3], c[6], c[7] = n_s, n_s + n_g, len(V0), len(V0) + len(V1), n_g, len(V1)
    c[5] = int((hp > ws.cap_s).any() or (ho > ws.cap_g).any())
    ws.counts = torch.from_numpy(c.astype(np.int32))
    if counts_host is not None:
        counts_host.copy_(ws.counts)
    ws.nuL, ws.niL = nuL, niL


def gather_into(table, ids, out):
    out.copy_(torch.from_numpy(ro.gather(_n(table), _n(ids))))
    return out


def shard_forward(ws, P, Q, slotsP, slotsQ, opt, fetched=None, send=None, checked=False):
    """dr_shard_bpr_forward without its in-place updates (shard_finish below applies every row: the same sums in the
    same order -- that the kernels' split between forward and tail changes nothing is what the GPU tests check)."""
    Pn, Qn = _n(P), _n(Q)
    Qcat = Qn if fetched is None else np.concatenate((Qn, _n(fetched)))
    lb, g, _ = ro.bpr_forward(Pn, Qcat, ws.u_loc, ws.ref[0], ws.ref[1])
    pu = ro.gather(Pn, ws.u_loc)
    qi, qj = ro.gather(Qcat, ws.ref[0]), ro.gather(Qcat, ws.ref[1])
    ws.g, ws.loss_b, ws.pu = g, lb, pu
    ws.stash = (g[:, None] * (qi - qj)).astype(np.float32)
    if send is not None:
        out = _n(send)
        for part, cf in ((0, g), (1, (-g).astype(np.float32))):
            rem = ws.ow[part] < 0
            out[-1 - ws.ow[part][rem]] = (cf[rem, None] * pu[rem]).astype(np.float32)


def shard_finish(ws, P, Q, slotsP, slotsQ, opt, received=None, checked=False):
    n_slots = {ro.OPT_SGD: 0, ro.OPT_ADAM: 2, ro.OPT_ADAGRAD: 1, ro.OPT_RMSPROP: 2}[opt.kind]
    s_idx, prt = ws.osrc >> 1, ws.osrc & 1
    local = ws.osrc >= 0
    src = np.zeros((ws.n_g, ws.k), np.float32)
    coef = np.ones(ws.n_g, np.float32)
    src[local] = ws.pu[s_idx[local]]
    coef[local] = np.where(prt[local] == 0, ws.g[s_idx[local]], -ws.g[s_idx[local]])
    if received is not None and (~local).any():
        src[~local] = _n(received)[-1 - ws.osrc[~local]]
    for T_, sl, ids, rows, cf in ((Q, slotsQ, ws.okey, src, coef), (P, slotsP, ws.u_loc, ws.stash, None)):
        slots = [x for x in sl if x is not None][:n_slots]
        Wn, sn = ro.sparse_row_update(opt.kind, _n(T_), [_n(x) for x in slots], ids, rows, None, cf, opt.reg, opt.reg_mode,
                                      _hp(opt))
        T_.copy_(torch.from_numpy(Wn))
        for x, v in zip(slots, sn):
            x.copy_(torch.from_numpy(v))
    ws.loss = torch.tensor([float(np.sum(ws.loss_b.astype(np.float64)))], dtype=torch.float32)
    return ws.loss


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


def score_rank(U, I, targets, train=None):
    """Oracle-backed stand-in for kernels.score_rank (dot scores): the ranks score_topk reports."""
    return score_topk(U, I, 1, train=train, targets=targets)[2]
