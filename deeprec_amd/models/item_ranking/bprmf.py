"""Bayesian Personalized Ranking MF -- drop-in for models/item_ranking/bprmf.py (class BPRMF, same
constructor, build_network / prepare_data / train / test / execute / predict / save surface).

The Keras model + tf.function train_op (bprmf.py:74-94,114-123) is one dr_bpr_step_f32 call; predict
(bprmf.py:168-173) is dr_pair_dot_f32; test() runs the full-catalog evaluator on dr_score_topk_f32.

Deviations from the reference, on purpose:
  D1  train_op is fed (user, item, neg) in the intended order; the reference swaps user and item ids
      (bprmf.py:117), which indexes P with item ids.
  reg_mode: "dense" (default; reference semantics: the l2_loss terms make every row's gradient non-zero,
      bprmf.py:118-120) or "touched" (L2 and Adam moments only on gathered rows; scalable).
  sampler: "reference" (default: the reference's per-positive np.random stream, bprmf.py:130-133) or "device"
      (dr_sample_negatives_i32 on the train CSR: same distribution, counter-based stream seeded from np.random,
      no per-positive Python loop -- SURVEY 8f rank 1).
"""
import time

import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.RankingMetrics import evaluate
from .._common import (NegItems, TFOptimizer, device_of, normal_, sample_uniform_negatives, to_dev, load_checkpoint, save_checkpoint)


class BPRMF(object):
    CKPT = ("P", "Q")        # parameter tensors written by save() / read by load()
    def __init__(self, num_user, num_item, learning_rate=0.001, reg_rate=0.1, epoch=500, batch_size=1024,
                 verbose=False, t=5, display_step=1000, device=None, reg_mode="dense", sampler="reference"):
        self.num_user = num_user
        self.num_item = num_item
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
        self._train_csr_dev = None
        self.P = None
        self.Q = None
        # tf.keras.optimizers.Adam(learning_rate) (bprmf.py:62): epsilon 1e-7
        self.optimizer = TFOptimizer("adam", learning_rate, eps=1e-7)
        self.loss_sum, self.loss_count = 0.0, 0            # tf.keras.metrics.Mean (bprmf.py:61)
        self.test_data = None
        self.user = None
        self.item = None
        self.neg_items = None
        self.test_users = None
        self.num_training = None
        self.total_batch = None
        self._ws = None
        print("You are running BPRMF.")

    def build_network(self, num_factor=30, P=None, Q=None):
        """bprmf.py:74-94: P, Q ~ N(0, 0.01^2).  Initial tables may be passed in (parity tests)."""
        dev = self.device
        self.P = normal_((self.num_user, num_factor), 0.01, dev) if P is None else to_dev(P, torch.float32, dev)
        self.Q = normal_((self.num_item, num_factor), 0.01, dev) if Q is None else to_dev(Q, torch.float32, dev)
        self.num_factor = self.P.shape[1]

    def prepare_data(self, train_data, test_data):
        """bprmf.py:97-112."""
        t = train_data.tocoo()
        self.user = t.row.reshape(-1)
        self.item = t.col.reshape(-1)
        self.num_training = len(self.item)
        self.test_data = test_data
        self.total_batch = int(self.num_training / self.batch_size)
        self.neg_items = self._get_neg_items(train_data.tocsr())
        self.test_users = set([u for u in self.test_data.keys() if len(self.test_data[u]) > 0])
        print("data preparation finished.")

    def train_op(self, batch_user, batch_item, batch_item_neg):
        """bprmf.py:114-123 (with D1 fixed): one fused step on device id tensors [B] int32."""
        B = batch_user.numel()
        if self._ws is None or self._ws.key[0] != B:
            self._ws = K.StepWorkspace(B, self.num_factor, self.num_user, self.num_item, self.reg_mode, self.device)
        self.optimizer.begin_step()
        opt = self.optimizer.params(self.reg_rate, self.reg_mode)
        sP = self.optimizer.slots_for(self.P)
        sQ = self.optimizer.slots_for(self.Q)
        return K.bpr_step(self.P, self.Q, sP, sQ, batch_user, batch_item, batch_item_neg, opt, self._ws)

    def train(self):
        """bprmf.py:125-150: shuffle, one uniform negative per positive, drop the remainder batch."""
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
                    reg = self.reg_rate * 0.5 * float((self.P * self.P).sum() + (self.Q * self.Q).sum())
                    self.loss_sum += float(loss.item()) + reg
                    self.loss_count += 1
                    print("Index: %04d; cost= %.9f" % (i + 1, self.loss_sum / self.loss_count))
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
        """bprmf.py:168-173 -> np.float32 [N]."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        return K.pair_dot(self.P, self.Q, u, i).cpu().numpy()

    # evaluator fast path (utils/evaluation/RankingMetrics.py): bilinear score, train rows as the mask
    def eval_operands(self, users):
        u = to_dev(np.asarray(list(users)), torch.int32, self.device)
        return {"U": K.gather(self.P, u), "I": self.Q, "kind": K.SCORE_DOT}

    def eval_train_items(self, u):
        return self.neg_items.train_items(u)

    def _get_neg_items(self, data):
        """bprmf.py:175-181."""
        return NegItems(data, self.num_item)
