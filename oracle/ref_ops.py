"""CPU restatement (numpy, fp32) of the DeepRec embedding/interaction hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the
reference file:line (relative to /root/reference) whose arithmetic it restates.
[TF-ext] marks TensorFlow semantics that are not in the reference tree
(TensorFlow is a third-party, unpinned dependency: README.md:26) -- those are
"parity unpinned".

Conventions shared with the HIP kernels (include/deeprec_amd.h):
  * all tables / activations fp32, ids int32 or int64;
  * duplicate-row gradient sums are taken in *blocked position order*: the
    contributions of one destination row are ordered by their position in the
    index list, summed sequentially inside blocks of SEG_CHUNK=32 and the block
    partials are then summed sequentially (deterministic; every fp32 add is
    rounded separately, no FMA contraction);
  * eval scores use the k-ordered fp32 fmaf chain (oracle/c/oracle.c), which is
    what v_mfma_f32_32x32x2_f32 computes bit for bit.
"""
import numpy as np

f32 = np.float32
SEG_CHUNK = 32

# loss / optimizer kind codes (mirrors include/deeprec_amd.h)
OPT_SGD, OPT_ADAM, OPT_ADAGRAD, OPT_RMSPROP = 0, 1, 2, 3
REG_TOUCHED, REG_DENSE, REG_NONE = 0, 1, 2


# --------------------------------------------------------------------------
# K1  embedding lookup   (bprmf.py:25-37, gmf.py:66-67, nfm.py:54, ...)
# --------------------------------------------------------------------------
def gather(table, ids, oob="zero"):
    """out[..., :] = table[ids[...], :].  [TF-ext] tf.nn.embedding_lookup: an
    out-of-range id raises on CPU and yields a zero row on GPU; both policies
    are explicit here ("error" / "zero")."""
    table = np.asarray(table, dtype=f32)
    ids = np.asarray(ids)
    flat = ids.reshape(-1).astype(np.int64)
    R = table.shape[0]
    bad = (flat < 0) | (flat >= R)
    if bad.any() and oob == "error":
        raise IndexError("gather: %d ids out of range [0,%d)" % (int(bad.sum()), R))
    safe = np.where(bad, 0, flat)
    if table.ndim == 1:
        out = table[safe].copy()
        out[bad] = 0
        return out.reshape(ids.shape)
    out = table[safe].copy()
    out[bad] = 0
    return out.reshape(ids.shape + (table.shape[1],))


# --------------------------------------------------------------------------
# K2  backward of the lookup: duplicate-summing scatter-add
#     (implicit in bprmf.py:121, gmf.py:77, cml.py:79 ...; [TF-ext]
#      IndexedSlices de-duplicated by segment-sum before apply)
# --------------------------------------------------------------------------
def segment_plan(ids):
    """Stable sort of ids -> (order, uniq_ids, seg_start[nseg+1])."""
    ids = np.asarray(ids).reshape(-1).astype(np.int64)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    if len(sid) == 0:
        return order, sid, np.zeros(1, np.int64)
    heads = np.flatnonzero(np.concatenate(([True], sid[1:] != sid[:-1])))
    return order, sid[heads], np.concatenate((heads, [len(sid)]))


def blocked_sum(rows):
    """Sum rows[m, k] in blocked position order (see module docstring)."""
    rows = np.asarray(rows, dtype=f32)
    total = np.zeros(rows.shape[1:], f32)
    for c0 in range(0, rows.shape[0], SEG_CHUNK):
        acc = np.zeros(rows.shape[1:], f32)
        for e in range(c0, min(c0 + SEG_CHUNK, rows.shape[0])):
            acc = acc + rows[e]
        total = total + acc
    return total


def segment_sums(ids, src, src_idx=None, coef=None):
    """For each unique id: blocked sum over its occurrences e (position order)
    of coef[e] * src[src_idx[e]].  Returns (uniq_ids, sums[nseg, k])."""
    src = np.asarray(src, dtype=f32)
    if src.ndim == 1:
        src = src[:, None]
    n = len(np.asarray(ids).reshape(-1))
    src_idx = np.arange(n) if src_idx is None else np.asarray(src_idx).reshape(-1)
    order, uniq, seg = segment_plan(ids)
    out = np.zeros((len(uniq), src.shape[1]), f32)
    for s in range(len(uniq)):
        e = order[seg[s]:seg[s + 1]]
        rows = src[src_idx[e]]
        if coef is not None:
            rows = (np.asarray(coef, dtype=f32)[e][:, None] * rows).astype(f32)
        out[s] = blocked_sum(rows)
    return uniq, out


def scatter_add(gtable, ids, gout):
    """gtable[ids[n], :] += gout[n, :] with duplicates summed first."""
    gtable = np.array(gtable, dtype=f32, copy=True)
    shp = gtable.shape
    g2 = gtable.reshape(shp[0], -1)
    uniq, sums = segment_sums(ids, np.asarray(gout, f32).reshape(len(np.asarray(ids).reshape(-1)), -1))
    g2[uniq] = g2[uniq] + sums
    return g2.reshape(shp)


# --------------------------------------------------------------------------
# K5  optimizer row updates  ([TF-ext] training_ops kernels; hyper-parameters
#     from bprmf.py:62, mf.py:103, gmf.py:77, cml.py:79, nfm.py:82, nnmf.py:84,
#     AttRec.py:110)
# --------------------------------------------------------------------------
def adam_alpha(lr, beta1, beta2, step):
    """[TF-ext] alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t), fp32 like the
    ApplyAdam kernel (beta powers are fp32 running products)."""
    b1p, b2p = f32(1), f32(1)
    for _ in range(int(step)):
        b1p = f32(b1p * f32(beta1))
        b2p = f32(b2p * f32(beta2))
    return f32(f32(lr) * np.sqrt(f32(1) - b2p) / (f32(1) - b1p))


def opt_apply(kind, w, g, state, hp):
    """One optimizer application on arrays (any shape), fp32, separate roundings.
    state: tuple of slot arrays (modified copies returned).  hp: dict.
      ADAM     slots (m, v)   hp: alpha, beta1, beta2, eps     (TF1 eps 1e-8, Keras 1e-7)
      ADAGRAD  slots (acc,)   hp: lr                           (acc0 = 0.1 / AttRec 0.05)
      RMSPROP  slots (ms,mom) hp: lr, rho, momentum, eps
      SGD      no slots       hp: lr
    """
    w = np.asarray(w, f32)
    g = np.asarray(g, f32)
    if kind == OPT_SGD:
        return (w - f32(hp["lr"]) * g).astype(f32), ()
    if kind == OPT_ADAM:
        m, v = state
        omb1 = f32(1) - f32(hp["beta1"])
        omb2 = f32(1) - f32(hp["beta2"])
        m = (m + (g - m) * omb1).astype(f32)
        v = (v + (g * g - v) * omb2).astype(f32)
        w = (w - (m * f32(hp["alpha"])) / (np.sqrt(v) + f32(hp["eps"]))).astype(f32)
        return w, (m, v)
    if kind == OPT_ADAGRAD:
        (acc,) = state
        acc = (acc + g * g).astype(f32)
        w = (w - (g * f32(hp["lr"])) * (f32(1) / np.sqrt(acc))).astype(f32)
        return w, (acc,)
    if kind == OPT_RMSPROP:
        ms, mom = state
        ms = (ms + (g * g - ms) * (f32(1) - f32(hp["rho"]))).astype(f32)
        mom = (mom * f32(hp["momentum"]) + (g * f32(hp["lr"])) / np.sqrt(ms + f32(hp["eps"]))).astype(f32)
        w = (w - mom).astype(f32)
        return w, (ms, mom)
    raise ValueError(kind)


def sparse_row_update(kind, W, slots, ids, src, src_idx, coef, reg, reg_mode, hp):
    """Apply one optimizer step to table W[R,k] given a sparse gradient
    expressed as (ids[e], coef[e] * src[src_idx[e]]) contributions.

    reg_mode REG_TOUCHED: only rows present in ids are updated, gradient
      = segment sum + reg * w   (documented deviation for huge tables)
    reg_mode REG_DENSE:   every row is updated with gradient
      = (segment sum or 0) + reg * w  -- reference semantics: the
      reg*tf.nn.l2_loss(<whole table>) term densifies the gradient (bprmf.py:118-120,
      gmf.py:73-75, mf.py:97-99).
    reg_mode REG_NONE:    rows present only, gradient = segment sum (cml.py:77-79).
    """
    W = np.array(W, f32, copy=True)
    one_d = W.ndim == 1
    if one_d:
        W = W[:, None]
    slots = [np.array(s, f32, copy=True).reshape(W.shape) for s in slots]
    uniq, sums = segment_sums(ids, src, src_idx, coef)
    if reg_mode == REG_DENSE:
        G = (f32(reg) * W).astype(f32)
        G[uniq] = (sums + G[uniq]).astype(f32)
        Wn, sn = opt_apply(kind, W, G, tuple(slots), hp)
        W, slots = Wn, list(sn)
    else:
        g = sums
        if reg_mode == REG_TOUCHED:
            g = (sums + f32(reg) * W[uniq]).astype(f32)
        Wn, sn = opt_apply(kind, W[uniq], g, tuple(s[uniq] for s in slots), hp)
        W[uniq] = Wn
        for s, n_ in zip(slots, sn):
            s[uniq] = n_
    if one_d:
        return W[:, 0], [s[:, 0] for s in slots]
    return W, slots


# --------------------------------------------------------------------------
# K3/K4  interactions and losses
# --------------------------------------------------------------------------
def dot_seq(a, b):
    """reduce_sum(multiply(a, b), 1) in fp32, sequential over k (bprmf.py:91-92)."""
    a = np.asarray(a, f32)
    b = np.asarray(b, f32)
    acc = np.zeros(a.shape[:-1], f32)
    for c in range(a.shape[-1]):
        acc = acc + a[..., c] * b[..., c]
    return acc


def sigmoid(x):
    x = np.asarray(x, f32)
    with np.errstate(over="ignore"):
        return (f32(1) / (f32(1) + np.exp(-x))).astype(f32)


def bpr_forward(P, Q, u, i, j):
    """bprmf.py:84-92,118-119: x = p.q_i - p.q_j; loss_b = -log(sigmoid(x))
    (log(sigmoid) computed literally); returns (loss_b[B], g[B] = dL/dx, x)."""
    pu, qi, qj = gather(P, u), gather(Q, i), gather(Q, j)
    x = (dot_seq(pu, qi) - dot_seq(pu, qj)).astype(f32)
    s = sigmoid(x)
    with np.errstate(divide="ignore"):
        loss_b = (-np.log(s)).astype(f32)
    g = (-(f32(1) - s)).astype(f32)
    return loss_b, g, x


def pairwise_step(P, Q, slotsP, slotsQ, u, i, j, opt_kind, hp, reg, reg_mode, g=None):
    """One BPR-MF training step (bprmf.py:114-123).  Returns
    (P', Q', slotsP', slotsQ', loss_sum_without_reg, loss_b).
    g: optional per-sample dL/dx to use instead of the oracle's own forward (tests feed the
    device's g -- itself checked against bpr_forward -- to verify everything downstream of the
    transcendental functions bit for bit)."""
    P = np.asarray(P, f32)
    Q = np.asarray(Q, f32)
    u = np.asarray(u).reshape(-1)
    i = np.asarray(i).reshape(-1)
    j = np.asarray(j).reshape(-1)
    B = len(u)
    loss_b, g_own, _ = bpr_forward(P, Q, u, i, j)
    g = g_own if g is None else np.asarray(g, f32)
    D = (g[:, None] * (gather(Q, i) - gather(Q, j))).astype(f32)      # dL/dp_u per sample
    Pn, sPn = sparse_row_update(opt_kind, P, slotsP, u, D, np.arange(B), None, reg, reg_mode, hp)
    idsQ = np.concatenate((i, j))
    coefQ = np.concatenate((g, -g)).astype(f32)
    srcQ = np.concatenate((u, u))
    Qn, sQn = sparse_row_update(opt_kind, Q, slotsQ, idsQ, P, srcQ, coefQ, reg, reg_mode, hp)
    return Pn, Qn, sPn, sQn, float(np.sum(loss_b.astype(np.float64))), loss_b


def l2_loss(x):
    """[TF-ext] tf.nn.l2_loss(x) = sum(x**2) / 2."""
    return float(np.sum(np.asarray(x, np.float64) ** 2) / 2)


def bce_forward(P, Q, u, i, y):
    """gmf.py:66-75: yhat = sigmoid(sum p*q); loss_b = -(y log(yhat+1e-10) +
    (1-y) log(1-yhat+1e-10)); g = dL/dlogit through the literal expression."""
    pu, qi = gather(P, u), gather(Q, i)
    z = dot_seq(pu, qi)
    p = sigmoid(z)
    y = np.asarray(y, f32)
    e = f32(1e-10)
    loss_b = (-(y * np.log(p + e) + (f32(1) - y) * np.log(f32(1) - p + e))).astype(f32)
    dp = (-(y / (p + e)) + (f32(1) - y) / (f32(1) - p + e)).astype(f32)   # dL/dyhat
    g = (dp * (p * (f32(1) - p))).astype(f32)                             # sigmoid'
    return loss_b, g, p


def sq_forward(P, Q, BU, BI, mu, u, i, y):
    """mf.py:50-55,96-97: yhat = p.q + b_u + b_i + mu; loss_b = (y - yhat)^2."""
    z = (dot_seq(gather(P, u), gather(Q, i)) + gather(BU, u) + gather(BI, i) + f32(mu)).astype(f32)
    y = np.asarray(y, f32)
    loss_b = ((y - z) * (y - z)).astype(f32)
    g = (f32(-2) * (y - z)).astype(f32)
    return loss_b, g, z


def pointwise_step(P, Q, slotsP, slotsQ, u, i, y, loss_kind, opt_kind, hp, reg, reg_mode,
                   BU=None, BI=None, slotsBU=None, slotsBI=None, mu=0.0):
    """GMF (gmf.py:57-79, loss_kind 'bce') / MF (mf.py:37-55,96-103, 'sq') step."""
    u = np.asarray(u).reshape(-1)
    i = np.asarray(i).reshape(-1)
    B = len(u)
    if loss_kind == "bce":
        loss_b, g, _ = bce_forward(P, Q, u, i, y)
    else:
        loss_b, g, _ = sq_forward(P, Q, BU, BI, mu, u, i, y)
    D = (g[:, None] * gather(Q, i)).astype(f32)
    Pn, sPn = sparse_row_update(opt_kind, P, slotsP, u, D, np.arange(B), None, reg, reg_mode, hp)
    Qn, sQn = sparse_row_update(opt_kind, Q, slotsQ, i, P, u, g, reg, reg_mode, hp)
    out = [Pn, Qn, sPn, sQn, float(np.sum(loss_b.astype(np.float64))), loss_b]
    if loss_kind == "sq":
        BUn, sBUn = sparse_row_update(opt_kind, BU, slotsBU, u, g, np.arange(B), None, reg, reg_mode, hp)
        BIn, sBIn = sparse_row_update(opt_kind, BI, slotsBI, i, g, np.arange(B), None, reg, reg_mode, hp)
        out += [BUn, BIn, sBUn, sBIn]
    return out


def hinge_forward(P, Q, u, i, j, mask_pos, mask_neg, keep, margin):
    """cml.py:68-77: d = sum(dropout((p-q)^2)); loss_b = max(d_pos - d_neg + margin, 0).
    [TF-ext] tf.nn.dropout scales kept elements by 1/keep; tf.maximum passes the
    gradient to its first argument where it is >= the second."""
    pu, qi, qj = gather(P, u), gather(Q, i), gather(Q, j)
    sc = f32(1) / f32(keep)
    mp = np.asarray(mask_pos, f32) * sc
    mn = np.asarray(mask_neg, f32) * sc
    dpos = pu - qi
    dneg = pu - qj
    acc_p = np.zeros(len(pu), f32)
    acc_n = np.zeros(len(pu), f32)
    for c in range(pu.shape[1]):
        acc_p = acc_p + (dpos[:, c] * dpos[:, c]) * mp[:, c]
        acc_n = acc_n + (dneg[:, c] * dneg[:, c]) * mn[:, c]
    a = (acc_p - acc_n + f32(margin)).astype(f32)
    loss_b = np.maximum(a, f32(0))
    act = (a >= 0).astype(f32)
    Gi = (-(act[:, None] * (f32(2) * mp * dpos))).astype(f32)     # dL/dq_i
    Gj = ((act[:, None] * (f32(2) * mn * dneg))).astype(f32)      # dL/dq_j
    return loss_b, Gi, Gj, acc_p, acc_n


def rownorm_clip(W, clip=1.0):
    """cml.py:80-81: [TF-ext] clip_by_norm(W, clip, axes=[1]):
    (row * clip) / max(||row||, clip)."""
    W = np.asarray(W, f32)
    ss = np.zeros(W.shape[0], f32)
    for c in range(W.shape[1]):
        ss = ss + W[:, c] * W[:, c]
    nrm = np.sqrt(ss)
    return ((W * f32(clip)) / np.maximum(nrm, f32(clip))[:, None]).astype(f32)


def hinge_step(P, Q, accP, accQ, u, i, j, mask_pos, mask_neg, keep, margin, lr, clip):
    """cml.py:72-81,120: sparse Adagrad on gathered rows, then whole-table clip."""
    u = np.asarray(u).reshape(-1)
    i = np.asarray(i).reshape(-1)
    j = np.asarray(j).reshape(-1)
    B = len(u)
    loss_b, Gi, Gj, _, _ = hinge_forward(P, Q, u, i, j, mask_pos, mask_neg, keep, margin)
    S = np.concatenate((Gi, Gj))
    hp = {"lr": lr}
    Pn, (aP,) = sparse_row_update(OPT_ADAGRAD, P, [accP], np.concatenate((u, u)), S, np.arange(2 * B),
                                  np.full(2 * B, -1, f32), 0.0, REG_NONE, hp)
    Qn, (aQ,) = sparse_row_update(OPT_ADAGRAD, Q, [accQ], np.concatenate((i, j)), S, np.arange(2 * B),
                                  None, 0.0, REG_NONE, hp)
    if clip is not None:
        Pn, Qn = rownorm_clip(Pn, clip), rownorm_clip(Qn, clip)
    return Pn, Qn, aP, aQ, float(np.sum(loss_b.astype(np.float64))), loss_b


# --------------------------------------------------------------------------
# NFM bi-interaction (nfm.py:54-61, 73-74)
# --------------------------------------------------------------------------
def bi_interaction_fwd(E, feats):
    """FM[b,:] = 0.5 * ((sum_f v)^2 - sum_f v^2), v = E[feats[b,f]];
    sums over f sequential in f."""
    v = gather(E, feats)                      # [B,F,k]
    s = np.zeros((v.shape[0], v.shape[2]), f32)
    q = np.zeros_like(s)
    for f in range(v.shape[1]):
        s = s + v[:, f]
        q = q + v[:, f] * v[:, f]
    return (f32(0.5) * (s * s - q)).astype(f32), s


def bi_interaction_bwd(E, feats, gFM):
    """dL/dv[b,f,:] = gFM[b,:] * (S[b,:] - v[b,f,:])."""
    v = gather(E, feats)
    _, s = bi_interaction_fwd(E, feats)
    return (np.asarray(gFM, f32)[:, None, :] * (s[:, None, :] - v)).astype(f32)


# --------------------------------------------------------------------------
# AttRec self-attention pooling (AttRec.py:142-186, 468-510)
# --------------------------------------------------------------------------
def timing_signal(length, channels, min_timescale=1.0, max_timescale=1.0e4):
    """AttRec.py:468-510 (add_timing_signal): sin / cos of position * geometric timescales, [length, channels]."""
    position = np.arange(length, dtype=f32)
    num_timescales = channels // 2
    inc = np.log(float(max_timescale) / float(min_timescale)) / (float(num_timescales) - 1)
    inv = f32(min_timescale) * np.exp(np.arange(num_timescales, dtype=f32) * f32(-inc)).astype(f32)
    scaled = position[:, None] * inv[None, :]
    signal = np.concatenate((np.sin(scaled), np.cos(scaled)), 1).astype(f32)
    if channels % 2:
        signal = np.pad(signal, ((0, 0), (0, 1)))
    return signal


def attn_pool_forward(pre, v, dtype=f32):
    """AttRec.py:150-186 after the projection GEMM.  pre [B,L,d] = (Q[item_seq] + timing) W, v [B,L,d] = Q[item_seq].
    query = key = relu(pre) (:150-160, shared "linear_project" kernel, no bias); logits = q k^T / sqrt(d) (:174);
    weights = softmax(logits) * (1 - eye) (:177-180); out = reduce_mean(weights v, 1) (:182-186).
    Returns (out [B,d], s [B,L,L] unmasked softmax, weights [B,L,L])."""
    pre, v = np.asarray(pre, dtype), np.asarray(v, dtype)
    B, L, d = pre.shape
    q = np.maximum(pre, 0)
    logits = np.matmul(q, q.transpose(0, 2, 1)) / dtype(np.sqrt(d))
    e = np.exp(logits - logits.max(-1, keepdims=True))
    s = (e / e.sum(-1, keepdims=True)).astype(dtype)
    w = s * (1 - np.eye(L, dtype=dtype))
    out = np.matmul(w, v).mean(1).astype(dtype)
    return out, s, w


def attn_pool_backward(pre, v, gout):
    """Gradient of attn_pool_forward's `out` w.r.t. pre and (directly) v, in float64 from the chain rule:
    c[m] = sum_{l != m} s[l,m]; dv[m] = c[m] gout / L; dc[m] = <gout, v[m]> / L; upstream of the softmax
    dW[l,m] = dc[m] (l != m), 0 on the diagonal; dlogits = s * (dW - rowsum(s * dW)) / sqrt(d);
    dq = (dlogits + dlogits^T) q; dpre = dq * (pre > 0).  Checked against torch autograd in tests/test_golden_cpu.py."""
    pre, v, gout = (np.asarray(a, np.float64) for a in (pre, v, gout))
    B, L, d = pre.shape
    _, s, w = attn_pool_forward(pre, v, np.float64)
    off = 1 - np.eye(L)
    c = w.sum(1)                                               # [B, L]
    dv = c[:, :, None] * gout[:, None, :] / L
    dc = np.einsum("bd,bmd->bm", gout, v) / L
    dW = dc[:, None, :] * off
    dlog = s * (dW - (s * dW).sum(-1, keepdims=True)) / np.sqrt(d)
    q = np.maximum(pre, 0)
    dq = np.matmul(dlog + dlog.transpose(0, 2, 1), q)
    return dq * (pre > 0), dv


# --------------------------------------------------------------------------
# Caser horizontal convolutions (Caser.py:132-154)
# --------------------------------------------------------------------------
def caser_hconv_forward(e, filters, biases, dtype=f32):
    """e [B,L,d]; filters[h-1] [C,h,d], biases[h-1] [C] for h = 1..L.  Per height: valid conv2d with kernel (h, d)
    (:135-144) -> relu -> max over the L-h+1 positions (:149-151); concatenated in ascending h (:153-154).
    Returns (out [B, C*L], arg [B, C*L]: winning position, -1 where the ReLU output is 0)."""
    e = np.asarray(e, dtype)
    B, L, d = e.shape
    outs, args = [], []
    for h, (w, b) in enumerate(zip(filters, biases), 1):
        w = np.asarray(w, dtype)
        win = np.lib.stride_tricks.sliding_window_view(e, (h, d), axis=(1, 2))[:, :, 0]      # [B, T, h, d]
        s = np.einsum("bthd,chd->btc", win, w) + np.asarray(b, dtype)[None, None, :]
        best = s.max(1)
        outs.append(np.maximum(best, 0))
        args.append(np.where(best > 0, s.argmax(1), -1))
    return np.concatenate(outs, 1).astype(dtype), np.concatenate(args, 1).astype(np.int32)


# --------------------------------------------------------------------------
# NeuMF's logit for every (user, item) pair   (neumf.py:73-115; evaluated over the whole catalog through predict,
# neumf.py:213-214, by RankingMetrics.py:82-101)
# --------------------------------------------------------------------------
def neumf_catalog_logits(P, Q, mlp_P, mlp_Q, Ws, bs, users, dtype=np.float64):
    """logit[r, i] = sum(P[u] * Q[i]) + sum(MLP(concat[mlp_Q[i], mlp_P[u]])) for u = users[r] and every item i; the
    reference applies tf.sigmoid to it (neumf.py:115).  ReLU after every dense layer (neumf.py:80-113)."""
    P, Q, mlp_P, mlp_Q = (np.asarray(a, dtype) for a in (P, Q, mlp_P, mlp_Q))
    Ws = [np.asarray(w, dtype) for w in Ws]
    bs = [np.asarray(b, dtype) for b in bs]
    out = np.empty((len(users), Q.shape[0]), dtype)
    for r, u in enumerate(users):
        x = np.concatenate((mlp_Q, np.broadcast_to(mlp_P[u], mlp_Q.shape)), 1)      # neumf.py:81
        for W, b in zip(Ws, bs):
            x = np.maximum(x @ W + b, 0)
        out[r] = (P[u] * Q).sum(1) + x.sum(1)                                         # neumf.py:78, 115
    return out


# --------------------------------------------------------------------------
# K6  full-catalog scoring, masked top-K, rank-of-target
#     (RankingMetrics.py:82-101, SeqRecMetrics.py:122-123,155,213,234)
# --------------------------------------------------------------------------
def rank_items(scores, cand):
    """Order of RankingMetrics.py:96-99: stable sort of (item, score) by score
    descending -> ties keep candidate order.  Returns ranked item list."""
    scores = np.asarray(scores)
    cand = np.asarray(cand)
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    return cand[order]


def topk_and_ranks(score_row, train_items, targets, K):
    """For one user: candidates = all items not in train_items, ascending id
    (== neg_items[u] order, bprmf.py:175-181).  Returns (topk ids, topk scores,
    rank of each target in the candidate ranking or -1 if not a candidate)."""
    n = len(score_row)
    mask = np.ones(n, bool)
    if len(train_items):
        mask[np.asarray(train_items, np.int64)] = False
    cand = np.flatnonzero(mask)
    ranked = rank_items(score_row[cand], cand)
    pos = {int(it): r for r, it in enumerate(ranked)}
    ranks = np.array([pos.get(int(t), -1) for t in targets], np.int64)
    top = ranked[:K]
    return top, score_row[top], ranks


def np_pairwise_sum(a):
    """numpy's float32 pairwise summation restated element by element (numpy/_core/src/umath/loops_utils.h.src,
    *_pairwise_sum) -- the order np.sum(..., axis=-1) adds the summands of SeqRecMetrics.py:122-123 in.  The HIP
    kernels of dr_score_rank_exact_f32 follow this restatement; tests/test_golden_cpu.py pins it against numpy."""
    a = np.asarray(a, f32)
    n = len(a)
    if n < 8:
        res = f32(0.0)
        for x in a:
            res = f32(res + x)
        return res
    if n <= 128:
        r = [f32(a[q]) for q in range(8)]
        c = 8
        while c < n - (n % 8):
            for q in range(8):
                r[q] = f32(r[q] + a[c + q])
            c += 8
        res = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
        while c < n:
            res = f32(res + a[c])
            c += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f32(np_pairwise_sum(a[:n2]) + np_pairwise_sum(a[n2:]))


def numpy_order_scores(kind, U, I, U2=None, I2=None, w1=0.0, w2=0.0, bias=None):
    """All-pairs scores in the reference evaluators' own numpy float32 arithmetic:
    kind 0: (p * q).sum() [+ bias]; 1: -((p - q) ** 2).sum();
    2: SeqRecMetrics.py:122-123 literally, -alpha * np.sum((p[:, None] - X) ** 2, axis=2) - (1 - alpha) * np.sum(...)
    with alpha = w1, 1 - alpha = w2 given as float32."""
    U, I = np.asarray(U, f32), np.asarray(I, f32)
    if kind == 0:
        s = np.sum(U[:, None] * I, axis=2)
    elif kind == 1:
        s = -np.sum((U[:, None] - I) ** 2, axis=2)
    else:
        U2, I2 = np.asarray(U2, f32), np.asarray(I2, f32)
        s = -f32(w1) * np.sum((U[:, None] - I) ** 2, axis=2) - f32(w2) * np.sum((U2[:, None] - I2) ** 2, axis=2)
    if bias is not None:
        s = s + np.asarray(bias, f32)
    assert s.dtype == f32
    return s


# --------------------------------------------------------------------------
# metrics (pinned against goldens generated from the reference's own numpy
# evaluator: tests/golden/metrics_*.json)
# --------------------------------------------------------------------------
import math


def _hits(ranked, test):
    s2 = set(int(t) for t in test)
    return [idx for idx, val in enumerate(ranked) if int(val) in s2]


def precision_recall_ndcg_at_k(k, rankedlist, test_matrix):
    """RankingMetrics.py:21-37."""
    n_k = k if len(test_matrix) > k else len(test_matrix)
    idcg_k = 0
    for i in range(n_k):
        idcg_k += 1 / math.log(i + 2, 2)
    hits = _hits(rankedlist, test_matrix)
    dcg_k = 0
    for h in hits:
        dcg_k += 1 / math.log(h + 2, 2)
    count = len(hits)
    return float(count / k), float(count / len(test_matrix)), float(dcg_k / idcg_k)


def map_mrr_ndcg(rankedlist, test_matrix):
    """RankingMetrics.py:40-65."""
    idcg = 0
    for i in range(len(test_matrix)):
        idcg += 1 / math.log(i + 2, 2)
    hits = _hits(rankedlist, test_matrix)
    ap = 0
    dcg = 0
    for c, h in enumerate(hits):
        ap += (c + 1) / (h + 1)
        dcg += 1 / math.log(h + 2, 2)
    mrr = 1 / (hits[0] + 1) if hits else 0
    mp = ap / len(hits) if hits else 0
    return mp, mrr, float(dcg / idcg)


def seq_precision_recall_ndcg_at_k(k, rankedlist, test_matrix):
    """SeqRecMetrics.py:23-43 (first value is the hit COUNT)."""
    p, r, n = precision_recall_ndcg_at_k(k, rankedlist, test_matrix)
    return len(_hits(rankedlist, test_matrix)), r, n


def seq_map_mrr_ndcg(rankedlist, test_matrix):
    """SeqRecMetrics.py:47-81 (first value is AUC = (N-1-first_hit)/(N-1))."""
    _, mrr, nd = map_mrr_ndcg(rankedlist, test_matrix)
    hits = _hits(rankedlist, test_matrix)
    mx = len(rankedlist) - 1
    count_test = mx - hits[0] if hits else 0
    return 1.0 * count_test / mx, mrr, nd


def RMSE(error, num):
    """RatingMetrics.py:4-6."""
    return np.sqrt(error / num)


def MAE(error_mae, num):
    """RatingMetrics.py:8-9."""
    return error_mae / num


def ranking_metrics_from_ranks(ranks, n_test):
    """The nine metrics of RankingMetrics.evaluate for one user computed from
    the 0-based ranks of the held-out items alone (ranks<0 = not a candidate):
    every metric there depends on the ranked list only through the positions of
    the test items (RankingMetrics.py:31,52)."""
    h = sorted(int(r) for r in ranks if r >= 0)
    out = {}
    for k in (5, 10):
        hk = [r for r in h if r < k]
        n_k = k if n_test > k else n_test
        idcg = sum(1 / math.log(i + 2, 2) for i in range(n_k))
        out["precision@%d" % k] = float(len(hk) / k)
        out["recall@%d" % k] = float(len(hk) / n_test)
        out["ndcg@%d" % k] = float(sum(1 / math.log(r + 2, 2) for r in hk) / idcg)
    idcg = 0
    for i in range(n_test):
        idcg += 1 / math.log(i + 2, 2)
    ap = 0
    dcg = 0
    for c, r in enumerate(h):
        ap += (c + 1) / (r + 1)
        dcg += 1 / math.log(r + 2, 2)
    out["map"] = ap / len(h) if h else 0
    out["mrr"] = 1 / (h[0] + 1) if h else 0
    out["ndcg"] = float(dcg / idcg)
    return out


def evaluate_ranking(score_fn, test_users, neg_items, test_data):
    """RankingMetrics.py:68-125 restated; returns the dict of nine means and the
    per-user top-10 lists instead of printing."""
    acc = {k: [] for k in ("precision@10", "recall@10", "precision@5", "recall@5", "map", "mrr",
                           "ndcg", "ndcg@5", "ndcg@10")}
    top10 = {}
    for u in test_users:
        item_ids = list(neg_items[u])
        scores = score_fn([u] * len(item_ids), item_ids)
        ranked = [r[0] for r in sorted(zip(item_ids, scores), key=lambda t: t[1], reverse=True)]
        top10[u] = ranked[:10]
        p5, r5, n5 = precision_recall_ndcg_at_k(5, ranked[:5], test_data[u])
        p10, r10, n10 = precision_recall_ndcg_at_k(10, ranked[:10], test_data[u])
        mp, mrr, nd = map_mrr_ndcg(ranked, test_data[u])
        for k_, v_ in (("precision@5", p5), ("recall@5", r5), ("ndcg@5", n5), ("precision@10", p10),
                       ("recall@10", r10), ("ndcg@10", n10), ("map", mp), ("mrr", mrr), ("ndcg", nd)):
            acc[k_].append(v_)
    return {k_: float(np.mean(v_)) for k_, v_ in acc.items()}, top10
