"""Sequence loader -- drop-in for utils/load_data/load_data_seq.py (class DataSet, same attributes)."""
import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix


class DataSet():
    class SeqData():
        def __init__(self, user_ids, sequences, targets=None):
            self.user_ids = user_ids
            self.sequences = sequences
            self.targets = targets
            self.L = sequences.shape[1]
            self.T = None
            if np.any(targets):
                self.T = targets.shape[1]

    def __init__(self, path="../../Data/ml1m/seq/train.txt", header=['user', 'item', 'rating'], sep=" ", seq_len=1,
                 target_len=1, isTrain=False, user_map=None, item_map=None, num_users=None, num_items=None):
        """load_data_seq.py:21-92: ids are re-indexed in first-seen order; with isTrain the sliding-window
        (seq_len + target_len) sequences are built and item ids shift by +1 so that 0 is the pad item."""
        self.path, self.header, self.sep = path, header, sep
        self.seq_len, self.target_len, self.isTrain = seq_len, target_len, isTrain
        if not user_map and not item_map:
            user_map, item_map = dict(), dict()
            self.num_user = self.num_item = 0
        else:
            self.num_user, self.num_item = len(user_map), len(item_map)
        df = pd.read_csv(self.path, sep=self.sep, names=self.header)
        raw_u = df.iloc[:, 0].tolist()
        raw_i = df.iloc[:, 1].tolist()
        for u in raw_u:
            if u not in user_map:
                user_map[u] = self.num_user
                self.num_user += 1
        for i in raw_i:
            if i not in item_map:
                item_map[i] = self.num_item
                self.num_item += 1
        if num_users and num_items:
            self.num_user, self.num_item = num_users, num_items
        print("....Load data finished. Number of users:", self.num_user, "Number of items:", self.num_item)
        self.user_map, self.item_map = user_map, item_map
        self.user_ids = np.array([user_map[u] for u in raw_u])
        self.item_ids = np.array([item_map[i] for i in raw_i])
        print(len(self.item_ids))
        if isTrain:
            self.load_data_seq()

    def load_data_seq(self):
        """load_data_seq.py:94-135."""
        for k in self.item_map:
            self.item_map[k] += 1
        self.item_ids = self.item_ids + 1
        self.num_item += 1
        W = self.seq_len + self.target_len
        order = np.lexsort((self.user_ids,))            # stable by user, file order inside a user
        u_sorted, i_sorted = self.user_ids[order], self.item_ids[order]
        uids, starts, counts = np.unique(u_sorted, return_index=True, return_counts=True)
        n_sub = int(sum(c - W + 1 if c >= W else 1 for c in counts))
        sequences = np.zeros((n_sub, self.seq_len))
        targets = np.zeros((n_sub, self.target_len))
        seq_users = np.empty(n_sub)
        test_sequences = np.zeros((self.num_user, self.seq_len))
        test_users = np.empty(self.num_user)
        row = 0
        for uid, s, c in zip(uids, starts, counts):
            items = i_sorted[s:s + c]
            windows = [items[e - W:e] for e in range(c, W - 1, -1)] if c >= W else [items]   # newest window first
            for n, w in enumerate(windows):
                if n == 0:
                    test_sequences[uid][:] = w[-self.seq_len:]
                    test_users[uid] = uid
                targets[row][:] = w[-self.target_len:]
                sequences[row][:] = w[:self.seq_len]
                seq_users[row] = uid
                row += 1
        self.sequences = self.SeqData(seq_users, sequences, targets)
        self.test_sequences = self.SeqData(test_users, test_sequences)

    def tocsr(self):
        """load_data_seq.py:196-201."""
        return csr_matrix((np.ones(len(self.user_ids)), (self.user_ids, self.item_ids)),
                          shape=(self.num_user, self.num_item))
