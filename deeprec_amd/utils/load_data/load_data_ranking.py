"""Implicit-feedback loaders -- drop-in for utils/load_data/load_data_ranking.py (same names, arguments and
return values), vectorised.  Row splits come from sklearn.model_selection.train_test_split exactly as in the
reference (load_data_ranking.py:71), so a seeded np.random state reproduces the reference's split."""
import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix
from sklearn.model_selection import train_test_split


def _split(path, header, test_size, sep):
    df = pd.read_csv(path, sep=sep, names=header, engine="python")
    n_users = df[header[0]].unique().shape[0]
    n_items = df[header[1]].unique().shape[0]
    train_data, test_data = train_test_split(df, test_size=test_size)
    return pd.DataFrame(train_data), pd.DataFrame(test_data), n_users, n_items


def _test_dict(test_data, n_users, n_items):
    r = test_data.iloc[:, 0].to_numpy() - 1
    c = test_data.iloc[:, 1].to_numpy() - 1
    m = csr_matrix((np.ones(len(r), np.int64), (r, c)), shape=(n_users, n_items))
    return {u: m.getrow(u).nonzero()[1] for u in range(n_users)}


def load_data_neg(path="../data/ml100k/movielens_100k.dat", header=['user_id', 'item_id', 'rating', 'category'],
                  test_size=0.2, sep="\t"):
    """load_data_ranking.py:64-106 -> (dok train matrix of ones, {u: test items}, n_users, n_items); ids - 1."""
    train_data, test_data, n_users, n_items = _split(path, header, test_size, sep)
    r = train_data.iloc[:, 0].to_numpy() - 1
    c = train_data.iloc[:, 1].to_numpy() - 1
    train_matrix = csr_matrix((np.ones(len(r), np.int64), (r, c)), shape=(n_users, n_items))
    test_dict = _test_dict(test_data, n_users, n_items)
    print("Load data finished. Number of users:", n_users, "Number of items:", n_items)
    return train_matrix.todok(), test_dict, n_users, n_items


def load_data_all(path="../data/ml100k/movielens_100k.dat", header=['user_id', 'item_id', 'rating', 'time'],
                  test_size=0.2, sep="\t"):
    """load_data_ranking.py:8-61 -> (dense 0/1 interaction rows, {u: test items}, n_users, n_items)."""
    train_data, test_data, n_users, n_items = _split(path, header, test_size, sep)
    dense = np.zeros((n_users, n_items), np.int64)
    dense[train_data.iloc[:, 0].to_numpy() - 1, train_data.iloc[:, 1].to_numpy() - 1] = 1
    test_dict = _test_dict(test_data, n_users, n_items)
    print("Load data finished. Number of users:", n_users, "Number of items:", n_items)
    return [list(row) for row in dense], test_dict, n_users, n_items


def load_data_separately(path_train=None, path_test=None, path_val=None, header=['user_id', 'item_id', 'rating'],
                         sep=" ", n_users=0, n_items=0):
    """load_data_ranking.py:109-159: pre-split files, ids used as they are."""
    def mat(p):
        df = pd.read_csv(p, sep=sep, names=header, engine="python")
        return csr_matrix((np.ones(len(df), np.int64), (df.iloc[:, 0].to_numpy(), df.iloc[:, 1].to_numpy())),
                          shape=(n_users, n_items))
    train_matrix = mat(path_train)
    test_matrix = mat(path_test)
    test_dict = {u: test_matrix.getrow(u).nonzero()[1] for u in range(n_users)}
    print("Load data finished. Number of users:", n_users, "Number of items:", n_items)
    return train_matrix.todok(), test_dict, n_users, n_items
