"""Self-attentive sequential recommendation -- drop-in for models/seq_rec/AttRec.py (class AttRec).

Graph (AttRec.py:41-112,132-232): tables P,V [U,d], Q,X,A [I,d] (truncated N, 0.001), W [I,2d], b [I];
Q[item_seq] + sinusoidal timing -> shared ReLU projection for query and key (no bias, Xavier, L2) ->
softmax(q k^T / sqrt(d)) with the diagonal zeroed AFTER the softmax -> weights @ Q[item_seq] -> mean over L;
score = 0.2*||X_i - p_u||^2 + 0.8*||out - Q_i||^2; hinge(target - negative + 0.5) + 0.001*(l2 P,V,X,Q) + kernel
reg; Adagrad(0.05, initial accumulator 0.05).  test() -> SeqRecMetrics.evaluate1 (fused two-distance scoring).
"""
import math

import numpy as np
import torch

from ... import kernels as K
from ...utils.evaluation.SeqRecMetrics import evaluate1
from .._common import (GraphedStep, TFOptimizer, Table, active_shard, attn_seq, seq_dist_loss, device_of, normal_, shift_sess, to_dev, load_checkpoint, save_checkpoint)
from ._seq_common import SeqBase


def timing_signal(length, channels, device, min_timescale=1.0, max_timescale=1.0e4):
    """AttRec.py:468-510 (tensor2tensor's add_timing_signal_1d), returned as [1, length, channels]."""
    position = torch.arange(length, dtype=torch.float32, device=device)
    num_timescales = channels // 2
    inc = math.log(float(max_timescale) / float(min_timescale)) / (float(num_timescales) - 1)
    inv = min_timescale * torch.exp(torch.arange(num_timescales, dtype=torch.float32, device=device) * -inc)
    scaled = position[:, None] * inv[None]
    signal = torch.cat((torch.sin(scaled), torch.cos(scaled)), 1)
    if channels % 2:
        signal = torch.nn.functional.pad(signal, (0, 1))
    return signal.reshape(1, length, channels)


class AttRec(SeqBase):
    def __init__(self, sess, num_user=None, num_item=None, learning_rate=0.05, reg_rate=1e-2, epoch=5000, batch_size=1000,
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
        self.alpha = 0.2
        print("AttSeqRec.")

    def build_network(self, L, num_T, num_factor=150, num_neg=1, init=None, lazy_v=True):
        self.L, self.num_T, self.num_factor, self.num_neg = L, num_T, num_factor, num_neg
        dev, d = self.device, num_factor
        init = init or {}

        def tab(n, rows, k, std):
            sh = self.shard
            if n in init:
                w = to_dev(init[n], torch.float32, dev)
                return Table(sh.take(w) if sh else w, sh)
            rows = sh.local_rows(rows) if sh else rows
            w = normal_((rows, k) if k else (rows,), std, dev)
            return Table(w.clamp_(-2 * std, 2 * std) if n in "PVQXA" else w, sh)    # truncated normal: |x| <= 2 sigma
        self.P, self.V = tab("P", self.num_user, d, 0.001), tab("V", self.num_user, d, 0.001)
        self.Q, self.X, self.A = tab("Q", self.num_item, d, 0.001), tab("X", self.num_item, d, 0.001), tab("A", self.num_item, d, 0.001)
        self.W, self.b = tab("W", self.num_item, 2 * d, 0.01), tab("b", self.num_item, 0, 0.01)
        if "linear_project" in init:
            Wp = to_dev(init["linear_project"], torch.float32, dev)
        else:                                                           # xavier_initializer: U(+-sqrt(6/(fan_in+fan_out)))
            lim = math.sqrt(6.0 / (d + d))
            Wp = (torch.rand((d, d), device=dev) * 2 - 1) * lim
        self.linear_project = Wp.requires_grad_(True)
        self._timing = None
        self.unfused = False        # True: distances / hinge as library elementwise ops under autograd
        self.optimizer = TFOptimizer("adagrad", self.learning_rate, initial_accumulator_value=0.05)
        # hipGraph replay pays off while the step is launch-bound (the reference's B = 1000, L = 5: ~80 short launches);
        # at C5 (B = 4096, L = 50) the kernels fill the GPU and the eager loop is faster (2.52 vs 2.69 ms).  The sharded
        # lookups synchronise with the host (split sizes), so they are never captured.
        if self.shard is None and self.batch_size * L * num_factor <= (1 << 22):
            self._step = GraphedStep(self.train_op)
        else:
            # V (AttRec.py:65) only appears in the L2 term and in getUserParam's return value: its dense pass is recorded
            # and replayed when V is read (_common.Table.lazy); never under graph replay, which runs no host code
            self.V.lazy = lazy_v
        return self

    def _seq_out(self, item_seq, want_weights=False, Q=None):
        """AttRec.py:142-186 -> (out [B,d], weights [B,L,L] or None).  Q: the table the sequence rows come from (default
        self.Q; the fetched rows of a StepRoute in a data-parallel step, item_seq then holds positions)."""
        v = (Q or self.Q).lookup(item_seq)                              # [B,L,d]
        if self._timing is None:
            self._timing = timing_signal(self.L, self.num_factor, self.device)
        if self.L <= K.ATT_MAX_L and self.num_factor <= K.ATT_MAX_D:
            out, s = attn_seq(v, self._timing, self.linear_project)     # dr_attn_pool_forward_f32 / _backward_f32
            return out, (s * (1 - torch.eye(self.L, device=self.device)) if want_weights else None)
        eye = torch.eye(self.L, device=self.device)
        x = v + self._timing                                            # longer sequences / wider factors: library ops
        q = torch.relu(x @ self.linear_project)
        logits = q @ q.transpose(1, 2) / np.sqrt(self.num_factor)
        w = torch.softmax(logits, -1) * (1 - eye)
        return (w @ v).mean(1), w

    def _distance(self, out, user_id, item_id, u=None, P=None, X=None, Q=None):
        """AttRec.py:190-232 -> [B, n].  u: the users' P rows when the caller has looked them up already (the train graph
        uses them for the target and the negative distance: one lookup, one gradient list).  P, X, Q: row sources other
        than the model's tables (a data-parallel step's fetched rows)."""
        w_items, w_items_2 = (X or self.X).lookup(item_id), (Q or self.Q).lookup(item_id)
        u = ((P or self.P).lookup(user_id) if u is None else u)[:, None, :]
        return self.alpha * ((w_items - u) ** 2).sum(2) + (1 - self.alpha) * ((out[:, None, :] - w_items_2) ** 2).sum(2)

    def train_op(self, user_id, item_seq, item_id, neg_item_id):
        self._zero_dense_grads()
        loss = self._forward_backward(user_id, item_seq, item_id, neg_item_id)
        self._apply_step()
        return loss

    def _zero_dense_grads(self):
        self.linear_project.grad = None

    def _forward_backward(self, user_id, item_seq, item_id, neg_item_id):
        """Loss and every gradient of one batch (slice): table gradients are recorded in Table.pending, the projection's
        accumulates in .grad.  Data-parallel: ONE route for the step's four lookups (sharding.StepRoute: the id
        all-gather, one exchange of rows, one of gradient rows) and one all-reduce of the dense gradient."""
        P, X, Q, Qs, items, rt = self.P, self.X, self.Q, self.Q, None, None
        if self.shard:
            T_ = item_id.shape[1]
            rt = self.shard.step_route(lambda: [item_seq, user_id, torch.cat((item_id, neg_item_id), 1)],
                                       key=(user_id, item_seq, item_id, neg_item_id))
            Qs, P, X, Q = rt.fetch([(self.Q, 0), (self.P, 1), (self.X, 2), (self.Q, 2)])
            item_seq, user_id, items = rt.pos(0), rt.pos(1), rt.pos(2)      # positions in the fetched rows
            item_id, neg_item_id = items[:, :T_].contiguous(), items[:, T_:].contiguous()
        out, _ = self._seq_out(item_seq, Q=Qs)
        if item_id.shape == neg_item_id.shape and 2 * item_id.shape[1] <= K.SEQ_MAX_SLOTS and not self.unfused:
            # lookups of P, X, Q rows, both distances, the hinge and every gradient in one kernel (AttRec.py:97,190-232)
            loss = seq_dist_loss(out, P, X, Q, user_id, item_id, neg_item_id, self.alpha, K.SEQ_LOSS_HINGE, 0.5, items=items)
        else:
            u = P.lookup(user_id)
            target = self._distance(out, user_id, item_id, u, P, X, Q)
            negative = self._distance(out, user_id, neg_item_id, u, P, X, Q)
            loss = torch.clamp(target - negative + 0.5, min=0).sum()
        loss.backward()
        if rt is not None:
            rt.push()
            self.shard.all_reduce_grads([self.linear_project])
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
            for t in (self.P, self.V, self.X, self.Q):
                o.apply_table(t, 0.001, K.REG_DENSE)
            o.apply_dense(self.linear_project.data, self.linear_project.grad, self.reg_rate)

    def prepare_data(self, train_data, test_data):
        self._prepare(train_data, test_data)

    def train(self, train_data):
        self.test_sequences = train_data.test_sequences.sequences
        self._train_epoch(self.num_neg * self.num_T)

    def test(self, test_data):
        self.test_users = [i for i in range(self.num_user)]
        return evaluate1(self)

    def execute(self, train_data, test_data):
        self.prepare_data(train_data, test_data)
        for epoch in range(self.epochs):
            if self.verbose:
                print("Epoch: %04d;" % (epoch))
            self.train(train_data)
            if (epoch) % self.T == 0 and epoch >= 5:
                print("Epoch: %04d; " % (epoch), end='')
                self.test(test_data)

    def getUserParam(self, user_id):
        """AttRec.py:116-119 -> (P_u, attention output of the user's test sequence, V_u, attention weights)."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        seq = to_dev(self.test_sequences[np.asarray(user_id).reshape(-1), :], torch.int32, self.device)
        with torch.no_grad():
            out, w = self._seq_out(seq, want_weights=True)
            return K.gather(self.P.w, u), out.contiguous(), K.gather(self.V.w, u), w

    def getItemParam(self, item_id):
        """AttRec.py:121-123 -> (X rows, Q rows, b[item_id])."""
        i = to_dev(np.asarray(item_id).reshape(-1), torch.int32, self.device)
        return K.gather(self.X.w, i), K.gather(self.Q.w, i), K.gather(self.b.w, i)

    def predict(self, user_id, item_id):
        """AttRec.py:404-415: minus the test-graph distance."""
        u = to_dev(np.asarray(user_id).reshape(-1), torch.int32, self.device)
        seq = to_dev(self.test_sequences[np.asarray(user_id).reshape(-1), :], torch.int32, self.device)
        it = to_dev(np.asarray(item_id).reshape(-1, 1), torch.int32, self.device)
        with torch.no_grad():
            out, _ = self._seq_out(seq)
            return (-self._distance(out, u, it)).reshape(-1).cpu().numpy()

    def save(self, path):
        """Tables, dense weights, optimizer slots and step state as safetensors (_common.save_checkpoint); the
        reference's save() is tf.train.Saver and never called."""
        save_checkpoint(self, path)

    def load(self, path):
        """Inverse of save() after build_network(): values copied in place, optimizer state restored."""
        load_checkpoint(self, path)

