"""Biased matrix factorization -- drop-in for models/rating_prediction/mf.py (class MF).

Graph (mf.py:37-55,96-103): yhat = p.q + b_u + b_i + mean(rating); sum (y-yhat)^2 + reg*(l2(B_I)+l2(B_U)+
l2(P)+l2(Q)); tf.train.AdamOptimizer(lr) (eps 1e-8) -> one dr_pointwise_step_f32 per batch.  test()
(mf.py:80-88) scores every test pair in ONE launch instead of one sess.run per pair.
"""
import time

import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.RatingMetrics import MAE, RMSE
from .._common import (TFOptimizer, device_of, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)


class MF():
    CKPT = ("P", "Q", "B_U", "B_I")        # parameter tensors written by save() / read by load()
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.001, reg_rate=0.01, epoch=500, batch_size=128,
                 show_time=False, T=2, display_step=1000, device=None, reg_mode="dense"):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        self.learning_rate = learning_rate
        self.epochs = epoch
        self.batch_size = batch_size
        self.reg_rate = reg_rate
        self.show_time = show_time
        self.T = T
        self.display_step = display_step
        self.device = device_of(device)
        self.reg_mode = {"dense": K.REG_DENSE, "touched": K.REG_TOUCHED}[reg_mode]
        self.mu = 0.0
        self._ws = None
        print("MF.")

    def build_network(self, num_factor=30, P=None, Q=None, B_U=None, B_I=None):
        dev = self.device
        mk = lambda a, shape: normal_(shape, 0.01, dev) if a is None else to_dev(a, torch.float32, dev)
        self.P = mk(P, (self.num_user, num_factor))
        self.Q = mk(Q, (self.num_item, num_factor))
        self.B_U = mk(B_U, (self.num_user,))
        self.B_I = mk(B_I, (self.num_item,))
        self.num_factor = self.P.shape[1]
        self.optimizer = TFOptimizer("adam", self.learning_rate, eps=1e-8)

    def train_op(self, batch_user, batch_item, batch_rating):
        B = batch_user.numel()
        if self._ws is None or self._ws.key[0] != B:
            self._ws = K.StepWorkspace(B, self.num_factor, self.num_user, self.num_item, self.reg_mode, self.device)
        o = self.optimizer
        o.begin_step()
        return K.pointwise_step(self.P, self.Q, o.slots_for(self.P), o.slots_for(self.Q), batch_user, batch_item,
                                batch_rating, K.LOSS_SQ, o.params(self.reg_rate, self.reg_mode), self.B_U, self.B_I,
                                o.slots_for(self.B_U), o.slots_for(self.B_I), self.mu, ws=self._ws)

    def train(self, train_data):
        """mf.py:57-78."""
        self.num_training = len(self.rating)
        total_batch = int(self.num_training / self.batch_size)
        idxs = np.random.permutation(self.num_training)  # shuffled ordering
        u_d = to_dev(self.user[idxs], torch.int32, self.device)
        i_d = to_dev(self.item[idxs], torch.int32, self.device)
        y_d = to_dev(self.rating[idxs], torch.float32, self.device)
        for i in range(total_batch):
            start_time = time.time()
            s = slice(i * self.batch_size, (i + 1) * self.batch_size)
            loss = self.train_op(u_d[s], i_d[s], y_d[s])
            if i % self.display_step == 0:
                reg = self.reg_rate * 0.5 * float(sum((t * t).sum() for t in (self.B_I, self.B_U, self.P, self.Q)))
                print("Index: %04d; cost= %.9f" % (i + 1, float(loss.item()) + reg))
                if self.show_time:
                    print("one iteration: %s seconds." % (time.time() - start_time))

    def test(self, test_data):
        """mf.py:80-88 (all pairs in one launch)."""
        test_set = list(test_data.keys())
        y = np.array([float(test_data.get(k)) for k in test_set])
        pred = self.predict([k[0] for k in test_set], [k[1] for k in test_set]).astype(np.float64)
        error = np.sum((y - pred) ** 2)
        error_mae = np.sum(np.abs(y - pred))
        rmse, mae = RMSE(error, len(test_set)), MAE(error_mae, len(test_set))
        print("RMSE:" + str(rmse) + "; MAE:" + str(mae))
        return {"RMSE": float(rmse), "MAE": float(mae)}

    def execute(self, train_data, test_data):
        t = train_data.tocoo()
        self.user = t.row.reshape(-1)
        self.item = t.col.reshape(-1)
        self.rating = t.data
        self.mu = float(np.mean(list(self.rating)))       # mf.py:96
        for epoch in range(self.epochs):
            print("Epoch: %04d;" % (epoch))
            self.train(train_data)
            if (epoch) % self.T == 0:
                self.test(test_data)

    def save(self, path):
        """Tables, dense weights, optimizer slots and step state as safetensors (_common.save_checkpoint); the
        reference's save() is tf.train.Saver and never called."""
        save_checkpoint(self, path)

    def load(self, path):
        """Inverse of save() after build_network(): values copied in place, optimizer state restored."""
        load_checkpoint(self, path)

    def predict(self, user_id, item_id):
        """mf.py:117-118."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        return K.pointwise_forward(self.P, self.Q, u, i, K.LOSS_SQ, BU=self.B_U, BI=self.B_I, mu=self.mu).cpu().numpy()
