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
