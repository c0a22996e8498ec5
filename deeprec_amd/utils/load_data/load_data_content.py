"""libfm feature loader -- drop-in for utils/load_data/load_data_content.py, with explicit paths (the
reference hard-codes '../Data/frappe/...' which does not exist)."""
features = {}


def load_data_fm(path="", train_file=None, test_file=None):
    """load_data_content.py:9-22 -> ({'X','Y'} train, {'X','Y'} test, features_M).
    `path` is the directory holding frappe.train.libfm / frappe.test.libfm unless the files are given."""
    import os
    train_file = train_file or os.path.join(path, "frappe.train.libfm")
    test_file = test_file or os.path.join(path, "frappe.test.libfm")
    count_num_feature_field(train_file)
    count_num_feature_field(test_file)
    features_M = len(features)
    return read_data(train_file), read_data(test_file), features_M


def count_num_feature_field(file):
    """load_data_content.py:25-36: dense feature ids in first-seen order over the 'idx:val' tokens."""
    with open(file) as f:
        for line in f:
            for e in line.strip().split(' ')[1:]:
                if e not in features:
                    features[e] = len(features)


def read_data(file):
    """load_data_content.py:38-53."""
    X, Y = [], []
    with open(file) as f:
        for line in f:
            elements = line.strip().split(' ')
            Y.append([float(elements[0])])
            X.append([features[e] for e in elements[1:]])
    return {'Y': Y, 'X': X}
