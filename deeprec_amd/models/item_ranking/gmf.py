"""Generalized Matrix Factorization -- drop-in for models/item_ranking/gmf.py (class GMF).

Graph (gmf.py:57-79): yhat = sigmoid(sum_k p*q), BCE(+1e-10) + reg*(l2(P)+l2(Q)), Adagrad(0.5) -> one
dr_pointwise_step_f32 per batch.  predict (gmf.py:160-161) = dr_pointwise_forward_f32.
"""
import random
import time

import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.RankingMetrics import evaluate
from .._common import (NegItems, TFOptimizer, device_of, device_pointwise_epoch, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)


class GMF(object):
    CKPT = ("P", "Q")        # parameter tensors written by save() / read by load()
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.5, reg_rate=0.01, epoch=500, batch_size=256,
                 verbose=False, t=1, display_step=1000, device=None, reg_mode="dense", sampler="reference"):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        self.learning_rate = learning_rate
        self.reg_rate = reg_rate
        self.epochs = epoch
        self.batch_size = batch_size
        self.verbose = verbose
        self.T = t
        self.display_step = display_step
        self.device = device_of(device)
        self.reg_mode = {"dense": K.REG_DENSE, "touched": K.REG_TOUCHED}[reg_mode]
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' or 'device'")
        self.sampler = sampler
        self.num_neg_sample = None
        self.P = None
        self.Q = None
        self.optimizer = None
        self.test_data = None
        self.user = None
        self.item = None
        self.label = None
        self.neg_items = None
        self.test_users = None
        self.num_training = None
        self.total_batch = None
        self._ws = None
        print("You are running GMF.")

    def build_network(self, num_factor=10, num_neg_sample=20, P=None, Q=None):
        """gmf.py:57-79: P, Q ~ N(0, 1); tf.train.AdagradOptimizer(learning_rate)."""
        self.num_neg_sample = num_neg_sample
        dev = self.device
        self.P = normal_((self.num_user, num_factor), 1.0, dev) if P is None else to_dev(P, torch.float32, dev)
        self.Q = normal_((self.num_item, num_factor), 1.0, dev) if Q is None else to_dev(Q, torch.float32, dev)
        self.num_factor = self.P.shape[1]
        self.optimizer = TFOptimizer("adagrad", self.learning_rate)
        return self

    def prepare_data(self, train_data, test_data):
        """gmf.py:81-98."""
        t = train_data.tocoo()
        self.user = list(t.row.reshape(-1))
        self.item = list(t.col.reshape(-1))
        self.label = list(t.data)
        self.test_data = test_data
        self.neg_items = self._get_neg_items(train_data.tocsr())
        self.test_users = set([u for u in self.test_data.keys() if len(self.test_data[u]) > 0])
        print("data preparation finished.")
        return self

    def _sample_epoch(self):
        """gmf.py:100-124: num_neg_sample negatives per positive via random.sample, labels 0, then shuffle."""
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
        return user_temp[idxs], item_temp[idxs], labels_temp[idxs]

    def train_op(self, batch_user, batch_item, batch_label):
        B = batch_user.numel()
        if self._ws is None or self._ws.key[0] != B:
            self._ws = K.StepWorkspace(B, self.num_factor, self.num_user, self.num_item, self.reg_mode, self.device)
        self.optimizer.begin_step()
        return K.pointwise_step(self.P, self.Q, self.optimizer.slots_for(self.P), self.optimizer.slots_for(self.Q),
                                batch_user, batch_item, batch_label, K.LOSS_BCE,
                                self.optimizer.params(self.reg_rate, self.reg_mode), ws=self._ws)

    def train(self):
        if self.sampler == "device":
            u_d, i_d, y_d = device_pointwise_epoch(self)
            self.num_training = u_d.numel()
            self.total_batch = int(self.num_training / self.batch_size)
        else:
            user_random, item_random, labels_random = self._sample_epoch()
            u_d = to_dev(user_random, torch.int32, self.device)
            i_d = to_dev(item_random, torch.int32, self.device)
            y_d = to_dev(labels_random, torch.float32, self.device)
        for i in range(self.total_batch):
            start_time = time.time()
            s = slice(i * self.batch_size, (i + 1) * self.batch_size)
            loss = self.train_op(u_d[s], i_d[s], y_d[s])
            if i % self.display_step == 0:
                if self.verbose:
                    reg = self.reg_rate * 0.5 * float((self.P * self.P).sum() + (self.Q * self.Q).sum())
                    print("Index: %04d; cost= %.9f" % (i + 1, float(loss.item()) + reg))
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
        """gmf.py:160-161."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        return K.pointwise_forward(self.P, self.Q, u, i, K.LOSS_BCE).cpu().numpy()

    def eval_score_chunks(self, users, chunk=None):
        """Yields (c0, c1, sigmoid scores [c1 - c0, num_item]) for consecutive chunks of `users` (MFMA GEMM); the
        evaluator selects per chunk, so the [users, items] matrix never exists.  The reference ranks the SIGMOID
        (gmf.py:70, RankingMetrics.py:82-101), which saturates in fp32 and then ties: the scores, not the dots, are
        what the selection sees."""
        users = list(users)
        if chunk is None:
            chunk = max(1, min(len(users), (1 << 26) // max(1, self.num_item)))
        for c0 in range(0, len(users), chunk):
            u = to_dev(np.asarray(users[c0:c0 + chunk]), torch.int32, self.device)
            yield c0, c0 + len(u), K.score_matrix(K.gather(self.P, u), self.Q, sigmoid=True)

    def eval_scores(self, users):
        """Full-catalog sigmoid scores [len(users), num_item] in one tensor (tests, small catalogs)."""
        u = to_dev(np.asarray(list(users)), torch.int32, self.device)
        return K.score_matrix(K.gather(self.P, u), self.Q, sigmoid=True)

    def eval_train_items(self, u):
        return self.neg_items.train_items(u)

    def _get_neg_items(self, data):
        return NegItems(data, self.num_item)
