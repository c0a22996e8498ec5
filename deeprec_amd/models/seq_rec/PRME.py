"""Personalized Ranking Metric Embedding -- drop-in for models/seq_rec/PRME.py (class PRME).

Graph (PRME.py:41-131): tables P,V [U,d], Q,X [I,d] ~ N(0, 0.001^2); L must be 1:
dist = 0.2*||X_i - p_u||^2 + 0.8*||Q_prev - Q_i||^2; loss = -sum log sigmoid(-target + negative) +
0.01*(l2 P + l2 X + l2 Q); tf.train.AdamOptimizer(lr).  test() -> SeqRecMetrics.evaluate (predict protocol),
served here by the fused two-distance scoring kernel.
"""
import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation import SeqRecMetrics
from .._common import (TFOptimizer, Table, seq_dist_loss, device_of, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)
from ._seq_common import SeqBase


class PRME(SeqBase):
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.001, reg_rate=1e-2, epoch=5000, batch_size=1000,
                 show_time=False, T=1, display_step=1000, verbose=False, device=None, sampler="reference"):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
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
        self.alpha = 0.2
        print("PRME.")

    def build_network(self, L, num_T, num_factor=100, num_neg=1, init=None):
        if L != 1:
            raise ValueError("PRME squeezes the sequence axis (PRME.py:108): L must be 1")
        self.L, self.num_T, self.num_factor, self.num_neg = L, num_T, num_factor, num_neg
        dev = self.device
        init = init or {}
        tab = lambda n, rows: Table(to_dev(init[n], torch.float32, dev) if n in init else normal_((rows, num_factor), 0.001, dev))
        self.P, self.V = tab("P", self.num_user), tab("V", self.num_user)
        self.Q, self.X = tab("Q", self.num_item), tab("X", self.num_item)
        self.optimizer = TFOptimizer("adam", self.learning_rate, eps=1e-8)
        self.unfused = False        # True: the distances / loss as library elementwise ops under autograd (what the fused kernel is tested against)
        return self

    def _distance(self, item_seq, user_id, item_id):
        out = self.Q.lookup(item_seq)[:, 0, :]
        w_items, w_items_2 = self.X.lookup(item_id), self.Q.lookup(item_id)
        u = self.P.lookup(user_id)[:, None, :]
        return 0.2 * ((w_items - u) ** 2).sum(2) + 0.8 * ((out[:, None, :] - w_items_2) ** 2).sum(2)

    def train_op(self, user_id, item_seq, item_id, neg_item_id):
        if item_id.shape == neg_item_id.shape and 2 * item_id.shape[1] <= K.SEQ_MAX_SLOTS and not self.unfused:
            # lookups of X, Q rows, both distances, BPR on distances and every gradient in one kernel (PRME.py:73,104-131)
            out = self.Q.lookup(item_seq)[:, 0, :]
            loss = seq_dist_loss(out, self.P, self.X, self.Q, user_id, item_id, neg_item_id, 0.2, K.SEQ_LOSS_BPR)
        else:
            target = self._distance(item_seq, user_id, item_id)
            negative = self._distance(item_seq, user_id, neg_item_id)
            loss = -torch.log(torch.sigmoid(-target + negative)).sum()
        loss.backward()
        o = self.optimizer
        o.begin_step()
        with torch.no_grad():
            for t in (self.P, self.X, self.Q):
                o.apply_table(t, 0.01, K.REG_DENSE)
        return loss.detach()

    def prepare_data(self, train_data, test_data):
        self._prepare(train_data, test_data)

    def train(self, train_data):
        self._train_epoch(self.num_neg * self.num_T)

    def test(self, test_data):
        self.test_users = [i for i in range(self.num_user)]
        return SeqRecMetrics.evaluate(self)

    def execute(self, train_data, test_data):
        self.prepare_data(train_data, test_data)
        for epoch in range(self.epochs):
            if self.verbose:
                print("Epoch: %04d;" % (epoch))
            self.train(train_data)
            if (epoch) % self.T == 0:
                print("Epoch: %04d; " % (epoch), end='')
                self.test(test_data)

    def predict(self, user_id, item_id):
        """PRME.py:226-236."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        seq = to_dev(self.test_sequences[np.asarray(user_id).reshape(-1), :], torch.int32, self.device)
        it = to_dev(np.asarray(item_id).reshape(-1, 1), torch.int32, self.device)
        with torch.no_grad():
            return (-self._distance(seq, u, it)).reshape(-1).cpu().numpy()

    def seq_eval_operands(self, users):
        """-(0.2*||X_i - p_u||^2 + 0.8*||Q_prev(u) - Q_i||^2) for every item: the two-distance kernel."""
        u = to_dev(np.asarray(list(users)), torch.int32, self.device)
        prev = to_dev(self.test_sequences[np.asarray(list(users)), 0], torch.int32, self.device)
        return {"U": K.gather(self.P.w, u), "U2": K.gather(self.Q.w, prev), "I": self.X.w, "I2": self.Q.w,
                "kind": K.SCORE_TWO_DIST, "alpha": 0.2}

    def save(self, path):
        """Tables, dense weights, optimizer slots and step state as safetensors (_common.save_checkpoint); the
        reference's save() is tf.train.Saver and never called."""
        save_checkpoint(self, path)

    def load(self, path):
        """Inverse of save() after build_network(): values copied in place, optimizer state restored."""
        load_checkpoint(self, path)

