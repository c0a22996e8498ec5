"""Convolutional sequence embedding recommendation -- drop-in for models/seq_rec/Caser.py (class Caser).

Graph (Caser.py:39-196): tables P [U,d], V [U,d*L], Q [I,d], W [I,2d], b [I] ~ N(0, 0.01^2); vertical conv
(L x 1, n_v filters) and L horizontal convs (i x d, n_h filters, ReLU, max over time) on Q[item_seq];
FC -> d (ReLU); x = concat[z, p_u]; score = <x, W_i> + b_i; loss = -mean log(sigmoid(pos)+1e-10) -
mean log(1-sigmoid(neg)+1e-10) + reg*(l2 P,Q,V,W,b) + kernel reg; tf.train.AdamOptimizer(lr).
The horizontal convolutions run on the HIP kernels of csrc/caser_conv.hip (one shared filter matrix for all heights,
library GEMMs for the products, fused window-sum / bias / ReLU / max-over-time, a sparse backward through the winning
windows only); the vertical convolution and the FC layer are library GEMMs under autograd; lookups, their backward and
all optimizer applications are the HIP kernels.  test() -> SeqRecMetrics.evaluate_caser (MFMA scoring with item bias).
"""
import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.SeqRecMetrics import evaluate_caser
from .._common import (TFOptimizer, Table, active_shard, seq_dot_loss, device_of, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)
from ._seq_common import SeqBase


def caser_convs(e, conv_v_w, conv_v_b, conv_h_w, conv_h_b):
    """Caser.py:118-172 on e = Q[item_seq] [B,L,d]: the vertical convolution (n_v filters [L,1]) and the L horizontal
    ones (n_h filters [h,d], ReLU, max over positions), concatenated [B, n_v*d + n_h*L].

    Written as GEMMs (rocBLAS) instead of conv2d: every filter height is a different convolution shape, and MIOpen
    searches and compiles a solver per shape and direction on first use (48 s for L = 5 on a fresh MI355X box, minutes
    for L = 50).  Horizontal filter of height h: G = e @ W_h^T gives every (position, filter row) product, and the
    window sum out[b,t,c] = sum_r G[b,t+r,c,r] is a strided view of G summed over r."""
    B, L, d = e.shape
    n_v = conv_v_w.shape[0]
    out_v = torch.matmul(conv_v_w[:, 0, :, 0], e) + conv_v_b[None, :, None]          # [n_v,L] @ [B,L,d] -> [B,n_v,d]
    outs = [out_v.permute(0, 2, 1).reshape(B, n_v * d)]                               # TF layout [B,d,n_v] flattened
    for w, b in zip(conv_h_w, conv_h_b):
        n_h, h = w.shape[0], w.shape[2]
        G = torch.matmul(e, w[:, 0].reshape(n_h * h, d).t()).contiguous()             # [B,L,n_h*h]
        s0, s1, _ = G.stride()
        win = G.as_strided((B, L - h + 1, n_h, h), (s0, s1, h, s1 + 1))               # [b,t,c,r] -> G[b,t+r,c,r]
        outs.append(torch.relu(win.sum(3) + b).max(1)[0])                             # [B,n_h]
    return torch.cat(outs, 1)


class _HConv(torch.autograd.Function):
    """Caser.py:132-154 for all filter heights at once: dr_caser_hconv_pool_f32 over G = e W^T, and the backward through
    the winning window of every (sequence, height, filter) only (dr_caser_hconv_backward_input_f32 / _filter_f32)."""

    @staticmethod
    def forward(ctx, e, W, bias, C):
        e = e.contiguous()
        out, arg = K.caser_hconv_forward(e, W, bias, C)
        ctx.save_for_backward(e, W, arg)
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, gout):
        e, W, arg = ctx.saved_tensors
        ge, gW, gb = K.caser_hconv_backward(e, W, arg, gout.contiguous(), ctx.C)
        return ge, gW, gb, None


class _VConv(torch.autograd.Function):
    """Caser.py:118-130: dr_caser_vconv_forward_f32 / _backward_f32 (one pass over the slab each way)."""

    @staticmethod
    def forward(ctx, e, w, bias):
        e, w2 = e.contiguous(), w.reshape(w.shape[0], -1).contiguous()
        ctx.save_for_backward(e, w2)
        ctx.wshape = w.shape
        return K.caser_vconv_forward(e, w2, bias.contiguous())

    @staticmethod
    def backward(ctx, g):
        e, w2 = ctx.saved_tensors
        ge, gw, gb = K.caser_vconv_backward(e, w2, g.contiguous())
        return ge, gw.reshape(ctx.wshape), gb


def caser_convs_fused(e, conv_v_w, conv_v_b, hw_cat, hb_cat, n_h):
    """caser_convs with the horizontal block on the HIP kernels: hw_cat [n_h L (L+1) / 2, d] holds every height's
    filters (row n_h h (h-1) / 2 + c h + r), hb_cat [n_h L] their biases in output order."""
    B, L, d = e.shape
    n_v = conv_v_w.shape[0]
    if n_v <= K.VC_MAX_NV:
        out_v = _VConv.apply(e, conv_v_w, conv_v_b)
    else:
        out_v = (torch.matmul(conv_v_w[:, 0, :, 0], e) + conv_v_b[None, :, None]).permute(0, 2, 1).reshape(B, n_v * d)
    return torch.cat((out_v, _HConv.apply(e, hw_cat, hb_cat, n_h)), 1)


class Caser(SeqBase):
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.001, reg_rate=1e-6, epoch=500, batch_size=1000,
                 show_time=False, T=1, display_step=1000, verbose=False, device=None, shard=None, sampler="reference"):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        self.shard = active_shard(shard)                    # _common.Shard: data-parallel over ranks, tables row-sharded
        self.learning_rate = learning_rate
        self.epochs = epoch
        self.batch_size = batch_size
        self.reg_rate = reg_rate
        self.show_time = show_time
        self.verbose = verbose
        self.T = T
        self.display_step = display_step
        self.device = device_of(device)
        self.neg_items = dict()
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' or 'device'")
        self.sampler = sampler
        print("Caser.")

    def build_network(self, L, num_T, n_h=16, n_v=4, num_factor=150, num_neg=2, init=None, lazy_v=True,
                      neg_alignment="reference"):
        """neg_alignment: Caser.train draws its negatives with _get_neg_items(self.users.squeeze(), ...) (Caser.py:243) --
        row r of the negatives is drawn from the pool of the r-th user in STORAGE order, but is fed next to the r-th user
        of the PERMUTED epoch (Caser.py:237-254), so a sample trains against negatives of an unrelated user (which may
        be its own user's positives).  "reference" (default: a drop-in reproduces the reference's sampling stream and
        loss curve, draw for draw) keeps that; "intended" draws row r for the user it is fed with (DESIGN.md section 8,
        D9).  Either way the pool is the user's whole-train-row complement neg_items[u] (Caser.py:302-319), not the
        targets-only pool of AttRec / PRME."""
        if neg_alignment not in ("intended", "reference"):
            raise ValueError("neg_alignment must be 'intended' or 'reference'")
        self.neg_alignment = neg_alignment
        self.n_h, self.n_v, self.L, self.num_T, self.num_factor, self.num_neg = n_h, n_v, L, num_T, num_factor, num_neg
        dev, d = self.device, num_factor
        init = init or {}
        sh = self.shard

        def tab(n, shape):
            if n in init:
                w = to_dev(init[n], torch.float32, dev)
                return Table(sh.take(w) if sh else w, sh)
            if sh:
                shape = (sh.local_rows(shape[0]),) + tuple(shape[1:])
            return Table(normal_(shape, 0.01, dev), sh)
        self.P, self.V = tab("P", (self.num_user, d)), tab("V", (self.num_user, d * L))
        self.Q, self.W, self.b = tab("Q", (self.num_item, d)), tab("W", (self.num_item, 2 * d)), tab("b", (self.num_item,))
        # V is looked up by the reference (Caser.py:62) but its rows feed nothing: only the L2 term touches it
        self.V.lazy = lazy_v
        dense = lambda n, shape, std: (to_dev(init[n], torch.float32, dev) if n in init else
                                       (normal_(shape, std, dev) if std else torch.zeros(shape, device=dev))).requires_grad_(True)
        self.conv_v_w, self.conv_v_b = dense("conv_v_w", (n_v, 1, L, 1), 1.0), dense("conv_v_b", (n_v,), 0)
        # every height's filters in ONE parameter (row n_h h (h-1) / 2 + c h + r = row r of filter c of height h): one
        # optimizer application instead of 2 L, and the layout the HIP kernels read; conv_h_w[i] / conv_h_b[i] are views
        self.fused = L <= K.HC_MAX_L and d <= K.HC_MAX_D and n_h <= K.HC_MAX_C
        rows = K.hconv_rows(n_h, L + 1)
        hw, hb = torch.empty((rows, d), device=dev), torch.zeros(n_h * L, device=dev)
        for i in range(1, L + 1):
            r0 = K.hconv_rows(n_h, i)
            hw[r0:r0 + n_h * i] = (to_dev(init["conv_h_w%d" % i], torch.float32, dev) if "conv_h_w%d" % i in init
                                   else normal_((n_h, 1, i, d), 1.0, dev)).reshape(n_h * i, d)
            if "conv_h_b%d" % i in init:
                hb[(i - 1) * n_h:i * n_h] = to_dev(init["conv_h_b%d" % i], torch.float32, dev)
        self.conv_h_wcat, self.conv_h_bcat = hw.requires_grad_(True), hb.requires_grad_(True)
        self.conv_h_w = [self.conv_h_wcat[K.hconv_rows(n_h, i):K.hconv_rows(n_h, i + 1)].view(n_h, 1, i, d)
                         for i in range(1, L + 1)]
        self.conv_h_b = [self.conv_h_bcat[(i - 1) * n_h:i * n_h] for i in range(1, L + 1)]
        fin = n_v * d + n_h * L
        # tf.layers.dense default kernel init would be glorot; the reference passes random_normal_initializer (sigma 1)
        self.full_w, self.full_b = dense("full_w", (fin, d), 1.0), dense("full_b", (d,), 0)
        self.fc1_dim_v, self.fc1_dim_h = n_v * d, n_h * d
        self.optimizer = TFOptimizer("adam", self.learning_rate, eps=1e-8)
        self.unfused = False        # True: score / BCE as library elementwise ops under autograd
        self.loss_scale = 1.0       # a caller that feeds 1/n of a batch per call sets 1/n (the loss is a batch MEAN)
        return self

    def _dense_params(self):
        return [self.conv_v_w, self.conv_v_b, self.full_w, self.full_b, self.conv_h_wcat, self.conv_h_bcat]

    def _user_emb(self, item_seq, user_id, Q=None, P=None):
        """Caser.py:98-185 -> x = concat[z, p_u]  [B, 2d].  Q, P: row sources other than the model's tables (a
        data-parallel step's fetched rows; the ids are then positions in them)."""
        e = (Q or self.Q).lookup(item_seq)
        if self.fused:
            out = caser_convs_fused(e, self.conv_v_w, self.conv_v_b, self.conv_h_wcat, self.conv_h_bcat, self.n_h)
        else:                                   # beyond the kernels' limits (L > 64, d > 256): library ops
            out = caser_convs(e, self.conv_v_w, self.conv_v_b, self.conv_h_w, self.conv_h_b)
        z = torch.relu(out @ self.full_w + self.full_b)
        return torch.cat((z, (P or self.P).lookup(user_id)), 1)

    def _score(self, x, item_id, W=None, b=None):
        return (x[:, None, :] * (W or self.W).lookup(item_id)).sum(2) + (b or self.b).lookup(item_id)

    def train_op(self, user_id, item_seq, item_id, neg_item_id):
        self._zero_dense_grads()
        loss = self._forward_backward(user_id, item_seq, item_id, neg_item_id)
        self._apply_step()
        return loss

    def _zero_dense_grads(self):
        for p in self._dense_params():
            p.grad = None

    def _forward_backward(self, user_id, item_seq, item_id, neg_item_id):
        """Loss and every gradient of one batch (slice): table gradients are recorded in Table.pending, the dense ones
        accumulate in .grad.  Data-parallel: ONE route for the step's four lookups (sharding.StepRoute: the id all-gather,
        one exchange of rows, one of gradient rows) and one all-reduce of the dense gradients."""
        Qs, P, W, b, items, rt = self.Q, self.P, self.W, self.b, None, None
        scale = self.loss_scale
        if self.shard:
            T_ = item_id.shape[1]
            rt = self.shard.step_route(lambda: [item_seq, user_id, torch.cat((item_id, neg_item_id), 1)],
                                       key=(user_id, item_seq, item_id, neg_item_id))
            Qs, P, W, b = rt.fetch([(self.Q, 0), (self.P, 1), (self.W, 2), (self.b, 2)])
            item_seq, user_id, items = rt.pos(0), rt.pos(1), rt.pos(2)      # positions in the fetched rows
            item_id, neg_item_id = items[:, :T_].contiguous(), items[:, T_:].contiguous()
            scale = scale / self.shard.world        # the loss is a batch MEAN: each rank holds 1 / world of the batch
        x = self._user_emb(item_seq, user_id, Qs, P)
        if item_id.shape[1] + neg_item_id.shape[1] <= K.SEQ_MAX_SLOTS and not self.unfused:
            # lookups of W, b rows, the scores, the mean BCE and every gradient in one kernel (Caser.py:79-80,184-192)
            loss = seq_dot_loss(x, W, b, item_id, neg_item_id, scale, items=items)
        else:
            target, negative = self._score(x, item_id, W, b), self._score(x, neg_item_id, W, b)
            loss = -torch.log(torch.sigmoid(target) + 1e-10).mean() - torch.log(1 - torch.sigmoid(negative) + 1e-10).mean()
            if scale != 1.0:
                loss = loss * scale
        loss.backward()
        if rt is not None:
            rt.push()
            self.shard.all_reduce_grads(self._dense_params())
        return loss.detach()

    def prefetch(self, user_id, item_seq, item_id, neg_item_id):
        """Data-parallel only: enqueue the route of a LATER train_op(user_id, item_seq, item_id, neg_item_id) call now
        (_common.Shard.prefetch_route) -- call it before the train_op of the step in between; the later step must be
        called with these very tensors."""
        if self.shard:
            self.shard.prefetch_route((user_id, item_seq, item_id, neg_item_id),
                                      lambda: [item_seq, user_id, torch.cat((item_id, neg_item_id), 1)])

    def _apply_step(self):
        o = self.optimizer
        o.begin_step()
        with torch.no_grad():
            for t in (self.P, self.Q, self.V, self.W, self.b):
                o.apply_table(t, self.reg_rate, K.REG_DENSE)
            for p in self._dense_params():
                o.apply_dense(p.data, p.grad, self.reg_rate if p.dim() > 1 else 0.0)

    def prepare_data(self, train_data, test_data):
        self._prepare(train_data, test_data)

    def _epoch_negatives(self, usr_r, tar_r, n_neg):
        """Caser.py:243,302-319: pool = neg_items[u] (see build_network: neg_alignment)."""
        users = self.users[:, 0] if self.neg_alignment == "reference" else usr_r
        return self._sample_negatives_users(users, n_neg)

    def _epoch_negatives_device(self, u_d, t_d, n_neg, seed):
        if getattr(self, "_train_csr_dev", None) is None:
            c = self.neg_items.csr
            self._train_csr_dev = (to_dev(c.indptr, torch.int32, self.device), to_dev(c.indices, torch.int32, self.device))
        users = to_dev(self.users[:, 0], torch.int32, self.device) if self.neg_alignment == "reference" else u_d
        return K.sample_negatives(*self._train_csr_dev, users, self.num_item, seed, n_neg=n_neg, first_item=1)

    def train(self, train_data):
        self.test_sequences = train_data.test_sequences.sequences
        self._train_epoch(self.num_neg * self.num_T)

    def test(self, test_data):
        self.test_users = [i for i in range(self.num_user)]
        return evaluate_caser(self)

    def execute(self, train_data, test_data):
        self.prepare_data(train_data, test_data)
        for epoch in range(self.epochs):
            if self.verbose:
                print("Epoch: %04d;" % (epoch))
            self.train(train_data)
            if (epoch) % self.T == 0:
                print("Epoch: %04d; " % (epoch), end='')
                self.test(test_data)

    def getUserParam(self, user_id):
        """Caser.py:90-92."""
        ids = np.asarray(user_id).reshape(-1)
        u = to_dev(ids, torch.int32, self.device)
        seq = to_dev(self.test_sequences[ids, :], torch.int32, self.device)
        with torch.no_grad():
            return self._user_emb(seq, u).contiguous()

    def getItemParam(self, item_id):
        """Caser.py:94-96."""
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        return K.gather(self.W.w, i), K.gather(self.b.w, i)

    def predict(self, user_id, item_id):
        ids = np.asarray(user_id).reshape(-1)
        u = to_dev(ids, torch.int32, self.device)
        seq = to_dev(self.test_sequences[ids, :], torch.int32, self.device)
        it = to_dev(np.asarray(item_id).reshape(-1, 1), torch.int32, self.device)
        with torch.no_grad():
            return self._score(self._user_emb(seq, u), it).reshape(-1).cpu().numpy()

    def save(self, path):
        """Tables, dense weights, optimizer slots and step state as safetensors (_common.save_checkpoint); the
        reference's save() is tf.train.Saver and never called."""
        save_checkpoint(self, path)

    def load(self, path):
        """Inverse of save() after build_network(): values copied in place, optimizer state restored."""
        load_checkpoint(self, path)

