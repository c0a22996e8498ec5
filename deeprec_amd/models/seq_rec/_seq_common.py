"""Host logic shared by the sequential recommenders (AttRec.py:319-354,356-387,448-466; PRME.py:133-160;
Caser.py:198-260): data preparation, negative sampling and the training loop skeleton."""
import time

import numpy as np
import torch

from .._common import NegItems, to_dev


class SeqBase:
    def _prepare(self, train_data, test_data):
        """AttRec.py:319-354: sequences / targets / users from the DataSet, per-sample candidate pools
        x[i] = all items (1..num_item-1; 0 is the pad) minus the sample's targets, test_data and neg_items."""
        self.sequences = train_data.sequences.sequences
        self.targets = train_data.sequences.targets
        self.users = train_data.sequences.user_ids.reshape(-1, 1)
        self.test_sequences = train_data.test_sequences.sequences
        self.all_items = set(np.arange(self.num_item - 1) + 1)
        self.test_data = dict()
        test = test_data.tocsr()
        for user in range(test.shape[0]):
            self.test_data[user] = list(set(test.indices[test.indptr[user]:test.indptr[user + 1]]))
        if not self.neg_items:
            self.neg_items = NegItems(train_data.tocsr(), self.num_item, first_item=1)
        print("Data Preparation Finish.")

    def _sample_negatives(self, targets_random, n):
        """AttRec.py:448-466 (_get_neg_items_sbpr): n draws per sample, uniformly from all_items - targets(sample),
        via np.random.randint(len(pool)) -- same RNG stream, the pool is addressed without being built."""
        num = self.num_item - 1
        out = np.zeros((len(targets_random), n))
        for r, tar in enumerate(targets_random):
            t = np.unique(np.asarray(tar, np.int64))
            t = t[(t >= 1) & (t <= num)] - 1                     # 0-based positions inside 1..num
            m = num - len(t)
            for j in range(n):
                s = np.random.randint(m)
                lo = 0
                while lo < len(t) and t[lo] - lo <= s:
                    lo += 1
                out[r, j] = s + lo + 1
        return out

    def _sample_negatives_users(self, users, n):
        """Caser.py:302-319 (_get_neg_items): n draws per row from neg_items[u] = all items 1..num_item-1 minus the
        user's whole train row, one np.random.randint(len(pool)) per draw -- same RNG stream, pools addressed through
        the train CSR without being built."""
        out = np.zeros((len(users), n))
        for r, u in enumerate(users):
            c = self.neg_items.count(int(u))
            for j in range(n):
                out[r, j] = self.neg_items.nth(int(u), np.random.randint(c))
        return out

    def _epoch_negatives(self, usr_r, tar_r, n_neg):
        """Host sampling of one epoch (AttRec / PRME: pool = all items minus the sample's own targets)."""
        return self._sample_negatives(tar_r, n_neg)

    def _epoch_negatives_device(self, u_d, t_d, n_neg, seed):
        """The same pools sampled on the device: dr_sample_negatives_excl_i32."""
        from ... import kernels as K
        return K.sample_negatives_excl(t_d, self.num_item, seed, n_neg=n_neg, first_item=1)

    def _train_epoch(self, n_neg):
        """AttRec.py:356-387 / Caser / PRME: permutation, per-sample negatives, batches (remainder dropped).
        sampler = "reference" (default): the reference's nested Python loops, draw for draw; "device": the same
        distribution from the HIP sampler, no per-sample host work."""
        self.num_training = len(self.sequences)
        self.total_batch = int(self.num_training / self.batch_size)
        idxs = np.random.permutation(self.num_training)
        seq_r, tar_r, usr_r = self.sequences[idxs], self.targets[idxs], self.users[idxs][:, 0]
        dev = self.device
        u_d = to_dev(usr_r, torch.int32, dev)
        s_d = to_dev(seq_r, torch.int32, dev)
        t_d = to_dev(tar_r, torch.int32, dev)
        if getattr(self, "sampler", "reference") == "device":
            seed = int(np.random.randint(0, 2 ** 31 - 1))     # one host draw per epoch keeps np.random.seed() reproducible
            n_d = self._epoch_negatives_device(u_d, t_d, n_neg, seed).contiguous()
        else:
            n_d = to_dev(self._epoch_negatives(usr_r, tar_r, n_neg), torch.int32, dev)
        step = getattr(self, "_step", None) or self.train_op
        sh = getattr(self, "shard", None)

        def batch(i):
            sl = slice(i * self.batch_size, (i + 1) * self.batch_size)
            if sh:          # data-parallel: this rank's contiguous part of the batch (same permutation on every rank)
                per = self.batch_size // sh.world
                sl = slice(i * self.batch_size + sh.rank * per, i * self.batch_size + (sh.rank + 1) * per)
            return u_d[sl].contiguous(), s_d[sl].contiguous(), t_d[sl].contiguous(), n_d[sl].contiguous()   # views
        ahead = sh and hasattr(self, "prefetch")
        nxt = batch(0) if self.total_batch > 0 else None
        for i in range(self.total_batch):
            start_time = time.time()
            cur, nxt = nxt, (batch(i + 1) if i + 1 < self.total_batch else None)
            if ahead and nxt is not None:                    # the next step's route is enqueued ahead of this step
                self.prefetch(*nxt)
            loss = step(*cur)
            if i % self.display_step == 0:
                if self.verbose:
                    print("Index: %04d; cost= %.9f" % (i + 1, float(loss.item())))
