"""Neural Network Matrix Factorization -- drop-in for models/rating_prediction/nnmf.py (class NNMF).

Graph (nnmf.py:37-84): concat[P_u(100), Q_i(100), U_u*V_i(10)] -> 210 -> 50 -> 50 -> 50 (sigmoid) -> 1;
N(0,1) kernels and biases, L2 on kernels; sum (y-yhat)^2 + reg*(||U||+||V||+||P||+||Q||) (Frobenius norms,
not squared: nnmf.py:82-83); RMSProp(lr) ([TF-ext] decay 0.9, momentum 0, eps 1e-10, ms slot starts at 1).
"""
import time

import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.RatingMetrics import MAE, RMSE
from .._common import (GraphedStep, TFOptimizer, Table, fused_mlp, device_of, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)


class NNMF():
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.001, reg_rate=0.01, epoch=500, batch_size=256,
                 show_time=False, T=1, display_step=1000, device=None):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        self.learning_rate = learning_rate
        self.epochs = epoch
        self.batch_size = batch_size
        self.reg_rate = reg_rate
        self.show_time = show_time
        self.T = T
        self.display_step = display_step
        self.device = device_of(device)
        print("NNMF.")

    def build_network(self, num_factor_1=100, num_factor_2=10, hidden_dimension=50, init=None):
        print("num_factor_1=%d, num_factor_2=%d, hidden_dimension=%d" % (num_factor_1, num_factor_2, hidden_dimension))
        dev = self.device
        init = init or {}
        tab = lambda n, rows, k: Table(to_dev(init[n], torch.float32, dev) if n in init else normal_((rows, k), 0.01, dev))
        self.P = tab("P", self.num_user, num_factor_1)
        self.Q = tab("Q", self.num_item, num_factor_1)
        self.U = tab("U", self.num_user, num_factor_2)
        self.V = tab("V", self.num_item, num_factor_2)
        d0 = 2 * num_factor_1 + num_factor_2
        dims = [d0, d0, hidden_dimension, hidden_dimension, hidden_dimension, 1]
        self.kernels, self.biases = [], []
        for l in range(5):
            W = to_dev(init["W%d" % l], torch.float32, dev) if "W%d" % l in init else normal_((dims[l], dims[l + 1]), 1.0, dev)
            b = to_dev(init["b%d" % l], torch.float32, dev) if "b%d" % l in init else normal_((dims[l + 1],), 1.0, dev)
            self.kernels.append(W.requires_grad_(True))
            self.biases.append(b.requires_grad_(True))
        self.optimizer = TFOptimizer("rmsprop", self.learning_rate, eps=1e-10)
        self._step = GraphedStep(self.train_op)

    def _forward(self, u, i):
        x = torch.cat((self.P.lookup(u), self.Q.lookup(i), self.U.lookup(u) * self.V.lookup(i)), 1)
        x = fused_mlp(x, self.kernels, self.biases, [K.ACT_SIGMOID] * 4 + [K.ACT_NONE])
        return x.reshape(-1)

    def train_op(self, batch_user, batch_item, batch_rating):
        for p in self.kernels + self.biases:
            p.grad = None
        d = batch_rating - self._forward(batch_user, batch_item)
        loss = (d * d).sum()
        loss.backward()
        o = self.optimizer
        o.begin_step()
        with torch.no_grad():
            keep = []
            for t in (self.P, self.Q, self.U, self.V):
                # d/dW reg*||W||_F = (reg / ||W||_F) * W (dense).  The coefficient stays on the device
                # (dr_opt_params.reg_dev): no host read-back in the step, so it can be replayed as a hipGraph
                nrm = torch.linalg.norm(t.w)
                coef = torch.where(nrm > 0, self.reg_rate / nrm, torch.zeros_like(nrm)).reshape(1)
                keep.append(coef)
                o.apply_table(t, coef, K.REG_DENSE)
            for W, b in zip(self.kernels, self.biases):
                o.apply_dense(W.data, W.grad, self.reg_rate)
                o.apply_dense(b.data, b.grad, 0.0)
        return loss.detach()

    def train(self, train_data):
        self.num_training = len(self.rating)
        total_batch = int(self.num_training / self.batch_size)
        idxs = np.random.permutation(self.num_training)  # shuffled ordering
        u_d = to_dev(self.user[idxs], torch.int32, self.device)
        i_d = to_dev(self.item[idxs], torch.int32, self.device)
        y_d = to_dev(self.rating[idxs], torch.float32, self.device)
        for i in range(total_batch):
            start_time = time.time()
            s = slice(i * self.batch_size, (i + 1) * self.batch_size)
            loss = self._step(u_d[s], i_d[s], y_d[s])
            if i % self.display_step == 0:
                print("Index: %04d; cost= %.9f" % (i + 1, float(loss.item())))
                if self.show_time:
                    print("one iteration: %s seconds." % (time.time() - start_time))

    def test(self, test_data):
        test_set = list(test_data.keys())
        y = np.array([float(test_data.get(k)) for k in test_set])
        pred = self.predict([k[0] for k in test_set], [k[1] for k in test_set]).astype(np.float64)
        rmse, mae = RMSE(np.sum((y - pred) ** 2), len(test_set)), MAE(np.sum(np.abs(y - pred)), len(test_set))
        print("RMSE:" + str(rmse) + "; MAE:" + str(mae))
        return {"RMSE": float(rmse), "MAE": float(mae)}

    def execute(self, train_data, test_data):
        t = train_data.tocoo()
        self.user = t.row.reshape(-1)
        self.item = t.col.reshape(-1)
        self.rating = t.data
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
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        with torch.no_grad():
            return self._forward(u, i).cpu().numpy()
