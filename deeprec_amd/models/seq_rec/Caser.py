"""Convolutional sequence embedding recommendation -- drop-in for models/seq_rec/Caser.py (class Caser).

Graph (Caser.py:39-196): tables P [U,d], V [U,d*L], Q [I,d], W [I,2d], b [I] ~ N(0, 0.01^2); vertical conv
(L x 1, n_v filters) and L horizontal convs (i x d, n_h filters, ReLU, max over time) on Q[item_seq];
FC -> d (ReLU); x = concat[z, p_u]; score = <x, W_i> + b_i; loss = -mean log(sigmoid(pos)+1e-10) -
mean log(1-sigmoid(neg)+1e-10) + reg*(l2 P,Q,V,W,b) + kernel reg; tf.train.AdamOptimizer(lr).
The convolution block is torch (MIOpen) under autograd -- SURVEY 8(f) lists it as the dense block "next" to
the hot path; lookups, their backward and all optimizer applications are the HIP kernels.  test() ->
SeqRecMetrics.evaluate_caser (MFMA scoring with item bias).
"""
import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.SeqRecMetrics import evaluate_caser
from .._common import TFOptimizer, Table, device_of, normal_, shift_sess, to_dev
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


class Caser(SeqBase):
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.001, reg_rate=1e-6, epoch=500, batch_size=1000,
                 show_time=False, T=1, display_step=1000, verbose=False, device=None, shard=None):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        self.shard = shard if (shard is not None and shard.world > 1) else None    # _common.Shard: DP over ranks
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
        print("Caser.")

    def build_network(self, L, num_T, n_h=16, n_v=4, num_factor=150, num_neg=2, init=None):
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
        dense = lambda n, shape, std: (to_dev(init[n], torch.float32, dev) if n in init else
                                       (normal_(shape, std, dev) if std else torch.zeros(shape, device=dev))).requires_grad_(True)
        self.conv_v_w, self.conv_v_b = dense("conv_v_w", (n_v, 1, L, 1), 1.0), dense("conv_v_b", (n_v,), 0)
        self.conv_h_w = [dense("conv_h_w%d" % i, (n_h, 1, i, d), 1.0) for i in range(1, L + 1)]
        self.conv_h_b = [dense("conv_h_b%d" % i, (n_h,), 0) for i in range(1, L + 1)]
        fin = n_v * d + n_h * L
        # tf.layers.dense default kernel init would be glorot; the reference passes random_normal_initializer (sigma 1)
        self.full_w, self.full_b = dense("full_w", (fin, d), 1.0), dense("full_b", (d,), 0)
        self.fc1_dim_v, self.fc1_dim_h = n_v * d, n_h * d
        self.optimizer = TFOptimizer("adam", self.learning_rate, eps=1e-8)
        return self

    def _dense_params(self):
        return [self.conv_v_w, self.conv_v_b, self.full_w, self.full_b] + self.conv_h_w + self.conv_h_b

    def _user_emb(self, item_seq, user_id):
        """Caser.py:98-185 -> x = concat[z, p_u]  [B, 2d]."""
        out = caser_convs(self.Q.lookup(item_seq), self.conv_v_w, self.conv_v_b, self.conv_h_w, self.conv_h_b)
        z = torch.relu(out @ self.full_w + self.full_b)
        return torch.cat((z, self.P.lookup(user_id)), 1)

    def _score(self, x, item_id):
        return (x[:, None, :] * self.W.lookup(item_id)).sum(2) + self.b.lookup(item_id)

    def train_op(self, user_id, item_seq, item_id, neg_item_id):
        for p in self._dense_params():
            p.grad = None
        x = self._user_emb(item_seq, user_id)
        target, negative = self._score(x, item_id), self._score(x, neg_item_id)
        loss = -torch.log(torch.sigmoid(target) + 1e-10).mean() - torch.log(1 - torch.sigmoid(negative) + 1e-10).mean()
        if self.shard:
            loss = loss / self.shard.world      # the loss is a batch MEAN: each rank holds 1/world of the batch
        loss.backward()
        if self.shard:
            self.shard.all_reduce_grads(self._dense_params())
        o = self.optimizer
        o.begin_step()
        with torch.no_grad():
            for t in (self.P, self.Q, self.V, self.W, self.b):
                o.apply_table(t, self.reg_rate, K.REG_DENSE)
            for p in self._dense_params():
                o.apply_dense(p.data, p.grad, self.reg_rate if p.dim() > 1 else 0.0)
        return loss.detach()

    def prepare_data(self, train_data, test_data):
        self._prepare(train_data, test_data)

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
        from safetensors.torch import save_file
        d = {n: getattr(self, n).w for n in ("P", "V", "Q", "W", "b")}
        d.update({"conv_v_w": self.conv_v_w, "conv_v_b": self.conv_v_b, "full_w": self.full_w, "full_b": self.full_b})
        for i, (w, b) in enumerate(zip(self.conv_h_w, self.conv_h_b), 1):
            d["conv_h_w%d" % i], d["conv_h_b%d" % i] = w, b
        save_file({k: v.detach().cpu().contiguous() for k, v in d.items()}, path)
