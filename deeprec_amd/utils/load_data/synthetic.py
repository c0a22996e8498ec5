"""Synthetic stand-ins for the reference's bundled data files (data/ml100k/*.dat, data/frappe/*.libfm), which
do not travel with this repository: same on-disk formats, same shapes, seeded.

  write_interaction_log : `user \t item \t rating \t time` lines, ids 1-based (load_data_ranking.py:64-106,
                          load_data_rating.py, load_data_seq.py read this format)
  write_libfm           : `label idx:1 idx:1 ...` lines with one active feature per field (load_data_content.py:25-53)
"""
import numpy as np


def write_interaction_log(path, n_user=943, n_item=1682, n_inter=100_000, min_per_user=20, seed=0, sep="\t"):
    """ml100k-shaped log: every user has at least `min_per_user` distinct items, item popularity ~ Zipf(1)."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_item + 1)
    w /= w.sum()
    extra = rng.multinomial(max(n_inter - n_user * min_per_user, 0), np.full(n_user, 1.0 / n_user))
    rows = []
    t = 880_000_000
    for u in range(n_user):
        deg = min(n_item, min_per_user + int(extra[u]))
        items = rng.choice(n_item, size=deg, replace=False, p=w)
        for it in items:
            t += int(rng.integers(1, 1000))
            rows.append((u + 1, int(it) + 1, int(rng.integers(1, 6)), t))
    order = rng.permutation(len(rows))
    with open(path, "w") as f:
        for r in order:
            f.write(sep.join(str(x) for x in rows[r]) + "\n")
    return len(rows)


def write_libfm(path, n_rows, fields=(957, 4082, 7, 7, 2, 4, 9, 3, 80, 233), seed=0):
    """frappe-shaped libfm file: 10 fields, one active feature per field, labels in {-1, 1}."""
    rng = np.random.default_rng(seed)
    offs = np.concatenate(([0], np.cumsum(fields)[:-1]))
    with open(path, "w") as f:
        for _ in range(n_rows):
            feats = [int(offs[j] + rng.integers(0, fields[j])) for j in range(len(fields))]
            f.write("%d %s\n" % (int(rng.choice((-1, 1))), " ".join("%d:1" % x for x in feats)))
    return int(sum(fields))
