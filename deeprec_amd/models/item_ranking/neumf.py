"""Neural Collaborative Filtering (NeuMF = GMF + MLP) -- drop-in for models/item_ranking/neumf.py.

Graph (neumf.py:61-131): four tables P,Q [.,10], mlp_P,mlp_Q [.,64] ~ N(0,1); GMF branch p*q; MLP on
concat[mlp_q, mlp_p] 128->128->80->40->20->10, ReLU everywhere, N(0,1) kernels, zero biases, L2 on kernels;
yhat = sigmoid(sum concat[GMF, MLP]); BCE(+1e-10) + kernel reg + reg*(l2 of the four tables); Adagrad(0.5).

Lookups, their duplicate-summing backward and every optimizer application run on the HIP kernels; the MLP's
small GEMMs (<= 128x128) are torch matmuls (rocBLAS) under autograd.  Evaluation runs the tower for every (user, item)
pair of a user chunk in dr_tower_catalog_scores_f32 and selects with dr_topk_rank_f32 (the score is not a bilinear form).
"""
import random
import time

import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.RankingMetrics import evaluate
from .._common import (GraphedStep, NegItems, TFOptimizer, device_pointwise_epoch, fused_mlp, Table, device_of, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)


class NeuMF(object):
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.5, reg_rate=0.01, epoch=500, batch_size=256,
                 verbose=False, t=1, display_step=1000, device=None, sampler="reference"):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        self.learning_rate = learning_rate
        self.reg_rate = reg_rate
        self.epochs = epoch
        self.batch_size = batch_size
        self.verbose = verbose
        self.T = t
        self.display_step = display_step
        self.device = device_of(device)
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' or 'device'")
        self.sampler = sampler
        self.num_neg_sample = None
        self.test_data = None
        self.user = None
        self.item = None
        self.label = None
        self.neg_items = None
        self.test_users = None
        self.num_training = None
        self.total_batch = None
        print("You are running NeuMF.")

    def build_network(self, num_factor=10, num_factor_mlp=64, hidden_dimension=10, num_neg_sample=30, init=None):
        """init: optional dict of numpy arrays (P, Q, mlp_P, mlp_Q, W0..W4, b0..b4) for parity tests."""
        self.num_neg_sample = num_neg_sample
        dev = self.device
        init = init or {}
        tab = lambda name, rows, k: Table(to_dev(init[name], torch.float32, dev) if name in init
                                          else normal_((rows, k), 1.0, dev))
        self.P = tab("P", self.num_user, num_factor)
        self.Q = tab("Q", self.num_item, num_factor)
        self.mlp_P = tab("mlp_P", self.num_user, num_factor_mlp)
        self.mlp_Q = tab("mlp_Q", self.num_item, num_factor_mlp)
        dims = [num_factor_mlp * 2, num_factor_mlp * 2, hidden_dimension * 8, hidden_dimension * 4,
                hidden_dimension * 2, hidden_dimension]
        self.kernels, self.biases = [], []
        for l in range(5):
            W = to_dev(init["W%d" % l], torch.float32, dev) if "W%d" % l in init else normal_((dims[l], dims[l + 1]), 1.0, dev)
            b = to_dev(init["b%d" % l], torch.float32, dev) if "b%d" % l in init else torch.zeros(dims[l + 1], device=dev)
            self.kernels.append(W.requires_grad_(True))
            self.biases.append(b.requires_grad_(True))
        self.optimizer = TFOptimizer("adagrad", self.learning_rate)
        self._step = GraphedStep(self.train_op)      # B = 256 steps are launch-bound: one hipGraph per step
        return self

    def _mlp(self, x):
        return fused_mlp(x, self.kernels, self.biases, [K.ACT_RELU] * len(self.kernels))

    def _logit(self, u, i):
        gmf = self.P.lookup(u) * self.Q.lookup(i)
        mlp = self._mlp(torch.cat((self.mlp_Q.lookup(i), self.mlp_P.lookup(u)), 1))
        return torch.cat((gmf, mlp), 1).sum(1)

    def prepare_data(self, train_data, test_data):
        t = train_data.tocoo()
        self.user = list(t.row.reshape(-1))
        self.item = list(t.col.reshape(-1))
        self.label = list(t.data)
        self.test_data = test_data
        self.neg_items = self._get_neg_items(train_data.tocsr())
        self.test_users = set([u for u in self.test_data.keys() if len(self.test_data[u]) > 0])
        print("data preparation finished.")
        return self

    def train_op(self, batch_user, batch_item, batch_label):
        """One sess.run((optimizer, loss)) of neumf.py:186-187; returns the data loss (device scalar)."""
        for p in self.kernels + self.biases:
            p.grad = None
        pred = torch.sigmoid(self._logit(batch_user, batch_item))
        y = batch_label
        loss = -(y * torch.log(pred + 1e-10) + (1 - y) * torch.log(1 - pred + 1e-10)).sum()
        loss.backward()
        o = self.optimizer
        o.begin_step()
        with torch.no_grad():
            for t in (self.P, self.Q, self.mlp_P, self.mlp_Q):
                o.apply_table(t, self.reg_rate, K.REG_DENSE)
            for W, b in zip(self.kernels, self.biases):
                o.apply_dense(W.data, W.grad, self.reg_rate)
                o.apply_dense(b.data, b.grad, 0.0)
        return loss.detach()

    def train(self):
        """neumf.py:153-192."""
        if self.sampler == "device":
            u_d, i_d, y_d = device_pointwise_epoch(self)
            self.num_training = u_d.numel()
            self.total_batch = int(self.num_training / self.batch_size)
            return self._run_epoch(u_d, i_d, y_d)
        user_append, item_append = [], []
        for u in self.user:
            cand = self.neg_items[u]
            picks = random.sample(range(len(cand)), self.num_neg_sample)
            user_append += [u] * self.num_neg_sample
            item_append += [int(cand[p]) for p in picks]
        user_temp = np.array(self.user + user_append)
        item_temp = np.array(self.item + item_append)
        labels_temp = np.array(list(self.label) + [0] * len(item_append), dtype=np.float32)
        self.num_training = len(item_temp)
        self.total_batch = int(self.num_training / self.batch_size)
        idxs = np.random.permutation(self.num_training)  # shuffled ordering
        u_d = to_dev(user_temp[idxs], torch.int32, self.device)
        i_d = to_dev(item_temp[idxs], torch.int32, self.device)
        y_d = to_dev(labels_temp[idxs], torch.float32, self.device)
        return self._run_epoch(u_d, i_d, y_d)

    def _run_epoch(self, u_d, i_d, y_d):
        for i in range(self.total_batch):
            start_time = time.time()
            s = slice(i * self.batch_size, (i + 1) * self.batch_size)
            loss = self._step(u_d[s], i_d[s], y_d[s])
            if i % self.display_step == 0:
                if self.verbose:
                    print("Index: %04d; cost= %.9f" % (i + 1, float(loss.item())))
                    print("one iteration: %s seconds." % (time.time() - start_time))

    def test(self):
        return evaluate(self)

    def execute(self, train_data, test_data):
        self.prepare_data(train_data, test_data)
        for epoch in range(self.epochs):
            self.train()
            if epoch % self.T == 0:
                print("Epoch: %04d; " % epoch, end='')
                self.test()

    def save(self, path):
        """Tables, dense weights, optimizer slots and step state as safetensors (_common.save_checkpoint); the
        reference's save() is tf.train.Saver and never called."""
        save_checkpoint(self, path)

    def load(self, path):
        """Inverse of save() after build_network(): values copied in place, optimizer state restored."""
        load_checkpoint(self, path)

    def predict(self, user_id, item_id):
        """neumf.py:213-214."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        with torch.no_grad():
            return torch.sigmoid(self._logit(u, i)).cpu().numpy()

    def eval_score_chunks(self, users, chunk=None):
        """Yields (c0, c1, sigmoid scores [c1 - c0, num_item]) for consecutive chunks of `users` -- the evaluator selects
        per chunk, so neither the [users, items] matrix nor any per-pair activation exists in HBM.  The first dense layer
        is linear in concat[mlp_q, mlp_p]: its item half is computed once, its user half per chunk (two small GEMMs);
        the rest of the tower runs for every pair inside dr_tower_catalog_scores_f32 (MFMA, activations in registers)
        on top of the GMF dots from dr_score_matrix_f32.  Tower shapes the kernel is not instantiated for take the
        library-GEMM path below (128 users at a time)."""
        users = list(users)
        dims = [k.shape[1] for k in self.kernels]
        fused = len(self.kernels) == 5 and K.tower_catalog_supported(dims)
        if chunk is None:
            chunk = max(1, min(len(users), (1 << 26) // max(1, self.num_item))) if fused else 128
        with torch.no_grad():
            W0 = self.kernels[0]
            km = self.mlp_Q.w.shape[1]
            item_half = (self.mlp_Q.w @ W0[:km]).contiguous()                    # [Ni, 128]
            for c0 in range(0, len(users), chunk):
                u = to_dev(np.asarray(users[c0:c0 + chunk]), torch.int32, self.device)
                out = K.score_matrix(K.gather(self.P.w, u), self.Q.w)            # sum_k p*q for all pairs (MFMA)
                user_half = K.gather(self.mlp_P.w, u) @ W0[km:] + self.biases[0]
                if fused:
                    K.tower_catalog_scores(user_half.contiguous(), item_half, [w.detach() for w in self.kernels[1:]],
                                           [b.detach() for b in self.biases[1:]], out)
                else:
                    h = torch.relu(item_half[None] + user_half[:, None])
                    for W, b in zip(self.kernels[1:], self.biases[1:]):
                        h = torch.relu(h @ W + b)
                    out += h.sum(2)
                yield c0, c0 + len(u), torch.sigmoid_(out)

    def eval_scores(self, users, chunk=None):
        """sigmoid scores [len(users), num_item] in one tensor (tests, small catalogs)."""
        users = list(users)
        out = torch.empty((len(users), self.num_item), dtype=torch.float32, device=self.device)
        for c0, c1, sc in self.eval_score_chunks(users, chunk):
            out[c0:c1] = sc
        return out

    def eval_train_items(self, u):
        return self.neg_items.train_items(u)

    def _get_neg_items(self, data):
        return NegItems(data, self.num_item)
