#!/usr/bin/env python
"""Generates tests/golden/*.json by IMPORTING the numpy parts of the reference (utils/evaluation,
utils/load_data) from /root/reference and recording their outputs on seeded inputs.

Run here only (the reference never travels to the GPU box):  python -B tools/make_golden.py
Only inputs and outputs are stored -- no reference source text.
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
from utils.evaluation import RankingMetrics as RM      # noqa: E402
from utils.evaluation import RatingMetrics as RT       # noqa: E402
from utils.evaluation import SeqRecMetrics as SM       # noqa: E402
from utils.load_data import load_data_content as LC    # noqa: E402
from utils.load_data import load_data_ranking as LR    # noqa: E402
from utils.load_data import load_data_rating as LT     # noqa: E402
from utils.load_data import load_data_seq as LS        # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f)
    print("wrote", name)


# ---- G1: metric functions ---------------------------------------------------------------------
def g1():
    rng = np.random.default_rng(0)
    cases = []
    for c in range(200):
        n_items = int(rng.integers(6, 60))
        ranked = rng.permutation(n_items).tolist()
        n_test = int(rng.integers(1, 15))
        if c % 9 == 0:
            test = (n_items + rng.permutation(20)[:n_test]).tolist()        # zero hits
        elif c % 9 == 1:
            test = [ranked[0]] + (n_items + rng.permutation(20)[:n_test - 1]).tolist()   # hit at rank 0
        else:
            test = rng.permutation(n_items + 5)[:n_test].tolist()
        k = [5, 10, 50][c % 3]
        cases.append({
            "k": k, "ranked": ranked, "test": test,
            "rank_prn": list(RM.precision_recall_ndcg_at_k(k, ranked[:k], test)),
            "rank_mmn": list(RM.map_mrr_ndcg(ranked, test)),
            "seq_prn": list(SM.precision_recall_ndcg_at_k(k, ranked[:k], test)),
            "seq_mmn": list(SM.map_mrr_ndcg(ranked, test)),
        })
    rating = [{"err": e, "n": n, "rmse": float(RT.RMSE(e, n)), "mae": float(RT.MAE(e, n))}
              for e, n in [(8, 2), (3, 2), (123.5, 77), (0.0, 5)]]
    dump("metrics_functions.json", {"cases": cases, "rating": rating})


# ---- G2/G3: evaluator loops with stub models ----------------------------------------------------
class Stub:
    pass


def capture(fn, model):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(model)
    out = {}
    for line in buf.getvalue().splitlines():
        if ":" in line:
            k, v = line.split(":", 1)
            try:
                out[k.strip()] = float(v)
            except ValueError:
                pass
    return out


def stub_case(seed, nu, ni, k, quantise):
    rng = np.random.default_rng(seed)
    P = rng.standard_normal((nu, k)).astype(np.float32)
    Q = rng.standard_normal((ni, k)).astype(np.float32)
    if quantise:                     # coarse values -> many exact score ties -> tie-break is exercised
        P = np.round(P * 2) / 2
        Q = np.round(Q * 2) / 2
    train = [sorted(rng.permutation(ni)[:int(rng.integers(0, ni // 3))].tolist()) for _ in range(nu)]
    test = {}
    for u in range(nu):
        rest = [i for i in range(ni) if i not in set(train[u])]
        n = int(rng.integers(0, 6)) if u % 5 else 0          # some users without test items
        test[u] = rng.permutation(rest)[:n].tolist()
    return P.astype(np.float32), Q.astype(np.float32), train, test


def near_ties(*tables):
    """Forced NEAR-ties: every odd row becomes its even neighbour with one coordinate moved by one ulp, so that for every
    user the two items' scores differ in the last bits only -- which of the two ranks first then depends on the
    exact float32 operation order of the scoring."""
    rng = np.random.default_rng(99)
    for T in tables:
        for r in range(0, T.shape[0] - 1, 2):
            T[r + 1] = T[r]
            c = int(rng.integers(0, T.shape[1]))
            T[r + 1, c] = np.nextafter(T[r, c], np.float32(np.inf if rng.integers(0, 2) else -np.inf), dtype=np.float32)


def g2():
    out = []
    for seed, quant, k in [(1, False, 6), (2, True, 6), (3, True, 6), (7, "near", 24)]:
        nu, ni = 23, 97
        P, Q, train, test = stub_case(seed, nu, ni, k, quant is True)
        if quant == "near":
            near_ties(Q)
        m = Stub()
        m.test_users = set(u for u in test if len(test[u]) > 0)
        m.neg_items = {u: [i for i in range(ni) if i not in set(train[u])] for u in range(nu)}
        m.test_data = {u: np.array(test[u]) for u in test}
        m.predict = lambda us, its: (P[np.array(us)] * Q[np.array(its)]).sum(1).astype(np.float32)
        res = capture(RM.evaluate, m)
        out.append({"P": P.tolist(), "Q": Q.tolist(), "train": train, "test": {str(u): test[u] for u in test},
                    "printed": res, "near_ties": quant == "near"})
    dump("ranking_evaluate.json", out)


def g3():
    out = []
    for seed, quant, k in [(4, False, 5), (5, True, 5), (6, "near", 40), (8, "near", 150)]:
        nu, ni = 19, 83
        P, Q, train, test = stub_case(seed, nu, ni, k, quant is True)
        rng = np.random.default_rng(seed + 100)
        S = rng.standard_normal((nu, k)).astype(np.float32)
        X = rng.standard_normal((ni, k)).astype(np.float32)
        bias = rng.standard_normal(ni).astype(np.float32)
        if quant is True:
            S, X = np.round(S * 2) / 2, np.round(X * 2) / 2
        if quant == "near":
            near_ties(X, Q)
            bias[1::2] = bias[0:-1:2]
        test1 = {u: (test[u][:1] if test[u] else []) for u in test}
        m = Stub()
        m.test_users = [u for u in range(nu)]                # evaluate1 indexes results[u] by user id
        m.test_users = [u for u in m.test_users if len(test1[u]) > 0]
        m.neg_items = {u: [i for i in range(ni) if i not in set(train[u])] for u in range(nu)}
        m.test_data = {u: np.array(test1[u]) for u in test1}
        m.num_item = ni
        m.alpha = 0.2
        # evaluate1 indexes results[u, j] with results built from getUserParam(test_users): give all users
        m.getUserParam = lambda users: (P, S, P, None)
        m.getItemParam = lambda items: (X, Q, bias)
        full_users = list(range(nu))
        m_all = m
        m_all.test_users = [u for u in full_users if len(test1[u]) > 0]
        # the reference uses u both as an index into results and as a user id: valid only if results has all users
        res1 = capture(SM.evaluate1, m_all)
        m2 = Stub()
        m2.test_users = m_all.test_users
        m2.neg_items, m2.test_data, m2.num_item = m.neg_items, m.test_data, ni
        m2.getUserParam = lambda users: P
        m2.getItemParam = lambda items: (Q, bias)
        res2 = capture(SM.evaluate_caser, m2)
        m3 = Stub()
        m3.test_users, m3.neg_items, m3.test_data = m_all.test_users, m.neg_items, m.test_data
        m3.predict = lambda us, its: (P[np.array(us)] * Q[np.array(its)]).sum(1).astype(np.float32)
        res3 = capture(SM.evaluate, m3)
        out.append({"P": P.tolist(), "S": S.tolist(), "X": X.tolist(), "Q": Q.tolist(), "bias": bias.tolist(),
                    "train": train, "test": {str(u): test1[u] for u in test1}, "alpha": 0.2,
                    "near_ties": quant == "near",
                    "evaluate1": res1, "evaluate_caser": res2, "evaluate": res3})
    dump("seq_evaluate.json", out)


# ---- G4: loaders on a tiny synthetic log ---------------------------------------------------------
def g4():
    rng = np.random.default_rng(7)
    nu, ni = 12, 30
    rows = []
    for u in range(1, nu + 1):
        for i in rng.permutation(ni)[:int(rng.integers(6, 14))] + 1:
            rows.append((u, int(i), int(rng.integers(1, 6)), int(rng.integers(10**8, 10**9))))
    # make sure every item id occurs so n_items == ni
    seen = set(r[1] for r in rows)
    for i in range(1, ni + 1):
        if i not in seen:
            rows.append((1, i, 3, 1))
    text = "".join("%d\t%d\t%d\t%d\n" % r for r in rows)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "log.dat")
        open(p, "w").write(text)
        with contextlib.redirect_stdout(io.StringIO()):
            np.random.seed(11)
            train, test, n_u, n_i = LR.load_data_neg(path=p, test_size=0.2, sep="\t")
            np.random.seed(12)
            tr2, te2, _, _ = LT.load_data_rating(path=p, test_size=0.1, sep="\t")
        coo = train.tocoo()
        out = {"log": text, "neg": {"seed": 11, "n_users": int(n_u), "n_items": int(n_i),
                                    "train_row": coo.row.tolist(), "train_col": coo.col.tolist(),
                                    "train_val": coo.data.tolist(),
                                    "test": {str(u): np.asarray(v).tolist() for u, v in test.items()}}}
        c2, t2 = tr2.tocoo(), te2.tocoo()
        out["rating"] = {"seed": 12, "train": [c2.row.tolist(), c2.col.tolist(), c2.data.tolist()],
                         "test": [t2.row.tolist(), t2.col.tolist(), t2.data.tolist()]}
        # sequence loader: space separated user item rating
        seq_text = "".join("%d %d %d\n" % (r[0], r[1], r[2]) for r in rows)
        ps = os.path.join(d, "seq.txt")
        open(ps, "w").write(seq_text)
        with contextlib.redirect_stdout(io.StringIO()):
            ds = LS.DataSet(path=ps, sep=" ", seq_len=3, target_len=2, isTrain=True)
        out["seq"] = {"text": seq_text, "L": 3, "T": 2, "num_user": int(ds.num_user), "num_item": int(ds.num_item),
                      "sequences": ds.sequences.sequences.tolist(), "targets": ds.sequences.targets.tolist(),
                      "users": ds.sequences.user_ids.tolist(), "test_sequences": ds.test_sequences.sequences.tolist(),
                      "user_map": {str(k): int(v) for k, v in ds.user_map.items()},
                      "item_map": {str(k): int(v) for k, v in ds.item_map.items()}}
        # libfm reader (load_data_content.py:25-53) on a tiny file
        fm_text = "1 a:1 b:1 c:1\n-1 a:1 d:1 e:1\n1 f:1 b:1 e:1\n"
        pf = os.path.join(d, "x.libfm")
        open(pf, "w").write(fm_text)
        LC.features.clear()
        LC.count_num_feature_field(pf)
        data = LC.read_data(pf)
        out["libfm"] = {"text": fm_text, "features": dict(LC.features), "X": data["X"], "Y": data["Y"]}
    dump("loaders.json", out)


if __name__ == "__main__":
    g1()
    g2()
    g3()
    g4()
