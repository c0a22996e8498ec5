"""Explicit-rating loader -- drop-in for utils/load_data/load_data_rating.py."""
import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix
from sklearn.model_selection import train_test_split


def load_data_rating(path="../data/ml100k/movielens_100k.dat", header=['user_id', 'item_id', 'rating', 'category'],
                     test_size=0.1, sep="\t"):
    """load_data_rating.py:8-49 -> (dok train ratings, dok test ratings, n_users, n_items); ids - 1."""
    df = pd.read_csv(path, sep=sep, names=header, engine="python")
    n_users = df[header[0]].unique().shape[0]
    n_items = df[header[1]].unique().shape[0]
    train_data, test_data = train_test_split(df, test_size=test_size)

    def mat(d):
        d = pd.DataFrame(d)
        return csr_matrix((d.iloc[:, 2].to_numpy(), (d.iloc[:, 0].to_numpy() - 1, d.iloc[:, 1].to_numpy() - 1)),
                          shape=(n_users, n_items))
    train_matrix, test_matrix = mat(train_data), mat(test_data)
    print("Load data finished. Number of users:", n_users, "Number of items:", n_items)
    return train_matrix.todok(), test_matrix.todok(), n_users, n_items
