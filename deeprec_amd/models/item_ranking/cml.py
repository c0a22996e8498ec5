"""Collaborative Metric Learning -- drop-in for models/item_ranking/cml.py (classes CML and CMLwarp).

Graph (cml.py:56-83): d = sum_k dropout((p-q)^2, keep=.98), hinge max(d_pos - d_neg + margin, 0), Adagrad(0.1)
on the gathered rows only (true sparse gradients), then clip_by_norm(rows, 1) over both tables after every
step (cml.py:80-81,120) -> one dr_hinge_step_f32 per batch.  predict = -distance with keep 1 (cml.py:151-153).
CMLwarp (cml.py:164-300) is the same model without dropout.
"""
import time

import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.RankingMetrics import evaluate
from .._common import (NegItems, TFOptimizer, device_of, normal_, sample_uniform_negatives, shift_sess, to_dev, load_checkpoint, save_checkpoint)


class CML(object):
    CKPT = ("P", "Q")        # parameter tensors written by save() / read by load()
    _keep_rate = 0.98
    _name = "CML"

    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.1, reg_rate=0.1, epoch=500, batch_size=500,
                 verbose=False, t=5, display_step=1000, device=None, sampler="reference"):
        self.sess, self.num_user, self.num_item = shift_sess(sess, num_user, num_item)
        if sampler not in ("reference", "device"):
            raise ValueError("sampler must be 'reference' or 'device'")
        # "reference": the per-positive np.random draw of cml.py:108-111 (same RNG stream); "device": the same
        # distribution from dr_sample_negatives_i32 over the train CSR, no Python loop
        self.sampler = sampler
        self._train_csr_dev = None
        self.learning_rate = learning_rate
        self.reg_rate = reg_rate
        self.epochs = epoch
        self.batch_size = batch_size
        self.verbose = verbose
        self.T = t
        self.display_step = display_step
        self.device = device_of(device)
        self.P = None
        self.Q = None
        self.user = None
        self.item = None
        self.num_training = None
        self.test_data = None
        self.total_batch = None
        self.neg_items = None
        self.test_users = None
        self._ws = None
        print("You are running %s." % self._name)

    def build_network(self, num_factor=100, margin=0.5, norm_clip_value=1, P=None, Q=None):
        dev = self.device
        std = 1 / (num_factor ** 0.5)
        self.P = normal_((self.num_user, num_factor), std, dev) if P is None else to_dev(P, torch.float32, dev)
        self.Q = normal_((self.num_item, num_factor), std, dev) if Q is None else to_dev(Q, torch.float32, dev)
        self.num_factor, self.margin, self.norm_clip_value = self.P.shape[1], margin, norm_clip_value
        self.optimizer = TFOptimizer("adagrad", self.learning_rate)
        return self

    def prepare_data(self, train_data, test_data):
        t = train_data.tocoo()
        self.user = t.row.reshape(-1)
        self.item = t.col.reshape(-1)
        self.num_training = len(self.item)
        self.test_data = test_data
        self.total_batch = int(self.num_training / self.batch_size)
        self.neg_items = self._get_neg_items(train_data.tocsr())
        self.test_users = set([u for u in self.test_data.keys() if len(self.test_data[u]) > 0])
        print(self.total_batch)
        print("data preparation finished.")

    def train_op(self, batch_user, batch_item, batch_item_neg, mask_pos=None, mask_neg=None):
        """One optimizer + clip step (cml.py:120).  Dropout keep-masks [B,k] uint8 may be passed in (parity);
        otherwise they are drawn with torch's generator."""
        B = batch_user.numel()
        if self._ws is None or self._ws.key[0] != B:
            self._ws = K.StepWorkspace(B, self.num_factor, self.num_user, self.num_item, K.REG_NONE, self.device)
        keep = self._keep_rate
        if keep < 1.0 and mask_pos is None:
            mask_pos = (torch.rand((B, self.num_factor), device=self.device) < keep).to(torch.uint8)
            mask_neg = (torch.rand((B, self.num_factor), device=self.device) < keep).to(torch.uint8)
        self.optimizer.begin_step()
        return K.hinge_step(self.P, self.Q, self.optimizer.slots_for(self.P)[0], self.optimizer.slots_for(self.Q)[0],
                            batch_user, batch_item, batch_item_neg, self.optimizer.params(0.0, K.REG_NONE),
                            keep=keep, margin=self.margin, clip=self.norm_clip_value, mask_pos=mask_pos,
                            mask_neg=mask_neg, ws=self._ws)

    def train(self):
        idxs = np.random.permutation(self.num_training)  # shuffled ordering
        user_random = self.user[idxs]
        item_random = self.item[idxs]
        u_d = to_dev(user_random, torch.int32, self.device)
        i_d = to_dev(item_random, torch.int32, self.device)
        if self.sampler == "device":
            if self._train_csr_dev is None:
                c = self.neg_items.csr
                self._train_csr_dev = (to_dev(c.indptr, torch.int32, self.device), to_dev(c.indices, torch.int32, self.device))
            seed = int(np.random.randint(0, 2 ** 31 - 1))       # one draw per epoch keeps np.random.seed() reproducible
            j_d = K.sample_negatives(*self._train_csr_dev, u_d, self.num_item, seed)[:, 0].contiguous()
        else:
            item_random_neg = sample_uniform_negatives(user_random, self.neg_items)[:, 0]
            j_d = to_dev(item_random_neg, torch.int32, self.device)
        for i in range(self.total_batch):
            start_time = time.time()
            s = slice(i * self.batch_size, (i + 1) * self.batch_size)
            loss = self.train_op(u_d[s], i_d[s], j_d[s])
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
        """cml.py:151-153: minus the squared distance (keep_rate 1)."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        d = K.gather(self.P, u) - K.gather(self.Q, i)
        return (-(d * d).sum(1)).cpu().numpy()

    def eval_operands(self, users):
        u = to_dev(np.asarray(list(users)), torch.int32, self.device)
        return {"U": K.gather(self.P, u), "I": self.Q, "kind": K.SCORE_NEG_SQDIST}

    def eval_train_items(self, u):
        return self.neg_items.train_items(u)

    def _get_neg_items(self, data):
        return NegItems(data, self.num_item)


class CMLwarp(CML):
    """cml.py:164-300: CML without dropout."""
    _keep_rate = 1.0
    _name = "CML warp loss"
