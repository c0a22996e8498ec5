"""Neural Factorization Machine -- drop-in for models/rating_prediction/nfm.py (class NFM).

Graph (nfm.py:39-82): E [M,128] ~ N(0,0.01^2), feature_bias [M,1] = 0, bias = 0, pred_weight [128,1];
gather [B,F,128] -> bi-interaction 0.5*((sum v)^2 - sum v^2) (fused HIP kernel, the [B,F,k] tensor never exists)
-> dense 128 ReLU (N(0,1) kernel AND bias) -> dropout keep 0.8 (hard-coded, also at test time: SURVEY D4) ->
@ pred_weight; + sum_f feature_bias + bias; loss = l2_loss(y - yhat) + reg*l2_loss(E); Adagrad(0.05).
E gets a dense gradient (its l2 term), feature_bias a true sparse one (nfm.py:74, no l2 term).
"""
import math
import time

import numpy as np
import torch

from ... import kernels as K
from .._common import (GraphedStep, TFOptimizer, Table, fused_mlp, bi_interaction, device_of, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)


class NFM():
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.05, reg_rate=0.01, epoch=500, batch_size=128,
                 show_time=False, T=2, display_step=1000, device=None):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        self.learning_rate = learning_rate
        self.epochs = epoch
        self.batch_size = batch_size
        self.reg_rate = reg_rate
        self.show_time = show_time
        self.T = T
        self.display_step = display_step
        self.device = device_of(device)
        self.dropout_keep = 0.8
        print("NFM.")

    def build_network(self, feature_M, num_factor=128, num_hidden=128, init=None):
        dev = self.device
        init = init or {}
        get = lambda n, f: to_dev(init[n], torch.float32, dev) if n in init else f()
        self.feature_M = feature_M
        self.feature_embeddings = Table(get("feature_embeddings", lambda: normal_((feature_M, num_factor), 0.01, dev)))
        self.feature_bias = Table(get("feature_bias", lambda: torch.zeros(feature_M, device=dev)).reshape(-1).contiguous())
        self.bias = get("bias", lambda: torch.zeros((), device=dev)).requires_grad_(True)
        self.pred_weight = get("pred_weight", lambda: to_dev(
            np.random.normal(loc=0, scale=np.sqrt(2.0 / (num_factor + num_hidden)), size=(num_hidden, 1)),
            torch.float32, dev)).requires_grad_(True)
        self.W1 = get("W1", lambda: normal_((num_factor, num_hidden), 1.0, dev)).requires_grad_(True)
        self.b1 = get("b1", lambda: normal_((num_hidden,), 1.0, dev)).requires_grad_(True)
        self.optimizer = TFOptimizer("adagrad", self.learning_rate)
        self._step = GraphedStep(self.train_op)      # B = 128 steps are launch-bound: one hipGraph per step

    def _forward(self, feats, drop_mask=None):
        fm = bi_interaction(self.feature_embeddings, feats)
        h = fused_mlp(fm, [self.W1], [self.b1], [K.ACT_RELU])
        if drop_mask is None:
            drop_mask = (torch.rand_like(h) < self.dropout_keep).to(h.dtype)
        h = h * drop_mask / self.dropout_keep                       # nfm.py:69, always on
        bilinear = (h @ self.pred_weight).sum(1, keepdim=True)
        f_b = self.feature_bias.lookup(feats).sum(1, keepdim=True)
        return bilinear + f_b + self.bias

    def prepare_data(self, train_data, test_data):
        print("data preparation finished.")
        return self

    def train_op(self, batch_x, batch_y, drop_mask=None):
        """batch_x int32 [B,F], batch_y f32 [B,1] device tensors."""
        dense = [self.bias, self.pred_weight, self.W1, self.b1]
        for p in dense:
            p.grad = None
        pred = self._forward(batch_x, drop_mask)
        d = batch_y - pred
        loss = 0.5 * (d * d).sum()                                   # tf.nn.l2_loss
        loss.backward()
        o = self.optimizer
        o.begin_step()
        with torch.no_grad():
            o.apply_table(self.feature_embeddings, self.reg_rate, K.REG_DENSE)
            o.apply_table(self.feature_bias, 0.0, K.REG_NONE)
            for p in dense:
                o.apply_dense(p.data, p.grad, 0.0)
        return loss.detach()

    def train(self, train_data):
        """nfm.py:91-111: X and Y shuffled in place with the same permutation (RNG state replay)."""
        self.num_training = len(train_data['Y'])
        total_batch = int(self.num_training / self.batch_size)
        rng_state = np.random.get_state()
        np.random.shuffle(train_data['Y'])
        np.random.set_state(rng_state)
        np.random.shuffle(train_data['X'])
        x_d = to_dev(np.asarray(train_data['X']), torch.int32, self.device)
        y_d = to_dev(np.asarray(train_data['Y'], dtype=np.float32), torch.float32, self.device)
        for i in range(total_batch):
            start_time = time.time()
            s = slice(i * self.batch_size, (i + 1) * self.batch_size)
            loss = self._step(x_d[s].contiguous(), y_d[s])
            if i % self.display_step == 0:
                reg = self.reg_rate * 0.5 * float((self.feature_embeddings.w ** 2).sum())
                print("Index: %04d; cost= %.9f" % (i + 1, float(loss.item()) + reg))
                if self.show_time:
                    print("one iteration: %s seconds." % (time.time() - start_time))

    def test(self, test_data, drop_mask=None):
        """nfm.py:113-131: whole test set in one forward, predictions clipped to the label range."""
        num_example = len(test_data['Y'])
        x = to_dev(np.asarray(test_data['X']), torch.int32, self.device)
        with torch.no_grad():
            y_pred = self._forward(x, drop_mask).reshape(-1).cpu().numpy().astype(np.float64)
        y_true = np.reshape(np.asarray(test_data['Y'], dtype=np.float64), (num_example,))
        bounded = np.minimum(np.maximum(y_pred, min(y_true)), max(y_true))
        RMSE = math.sqrt(np.mean((y_true - bounded) ** 2))
        print("RMSE:" + str(RMSE))
        return {"RMSE": RMSE}

    def execute(self, train_data, test_data):
        for epoch in range(self.epochs):
            print("Epoch: %04d;" % (epoch))
            self.train(train_data)
            if (epoch) % self.T == 0 and epoch > 100:
                self.test(test_data)

    def save(self, path):
        """Tables, dense weights, optimizer slots and step state as safetensors (_common.save_checkpoint); the
        reference's save() is tf.train.Saver and never called."""
        save_checkpoint(self, path)

    def load(self, path):
        """Inverse of save() after build_network(): values copied in place, optimizer state restored."""
        load_checkpoint(self, path)

    def predict(self, features, drop_mask=None):
        """The reference's predict (nfm.py:148-149) feeds placeholders that do not exist; here it takes
        feature rows [N,F] like the training input."""
        x = to_dev(np.asarray(features), torch.int32, self.device)
        with torch.no_grad():
            return self._forward(x, drop_mask).reshape(-1).cpu().numpy()
