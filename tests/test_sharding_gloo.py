"""world_size-2 (and 3) gloo runs of the row-sharded BPR step on CPU: the exchange logic of
deeprec_amd/sharding.py must reproduce, bit for bit, the single-process oracle step on the
concatenated global batch (identical results at every GPU count)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_ops as ro


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(seed, nu, ni, k, Bg):
    rng = np.random.default_rng(seed)
    P = (rng.standard_normal((nu, k)) * 0.1).astype(np.float32)
    Q = (rng.standard_normal((ni, k)) * 0.1).astype(np.float32)
    steps = []
    for _ in range(3):
        u = rng.integers(0, nu, Bg).astype(np.int32)
        i = (rng.zipf(1.3, Bg) % ni).astype(np.int32)       # duplicate-heavy
        j = rng.integers(0, ni, Bg).astype(np.int32)
        steps.append((u, i, j))
    return P, Q, steps


def _worker(rank, world, port, seed, kind, out, api):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import _oracle_ops as ops
        from deeprec_amd import kernels as K
        from deeprec_amd import sharding as sh
        nu, ni, k, Bg = 101, 57, 8, 96
        P, Q, steps = _case(seed, nu, ni, k, Bg)
        n_slots = 2 if kind == ro.OPT_ADAM else 0
        Pt = sh.ShardedTable(nu, k, world, rank, "cpu", n_slots)
        Qt = sh.ShardedTable(ni, k, world, rank, "cpu", n_slots)
        Pt.load_global(torch.from_numpy(P))
        Qt.load_global(torch.from_numpy(Q))
        Bl = Bg // world
        losses = []
        sl = slice(rank * Bl, (rank + 1) * Bl)
        mine = [tuple(torch.from_numpy(x[sl]) for x in b) for b in steps]
        tr = sh.ShardedBpr(ops, Pt, Qt)
        if api == "trainer":
            tr.prefetch(*mine[0])
        for t in range(1, len(mine) + 1):
            opt = K.opt_params(kind, lr=0.05, alpha=float(ro.adam_alpha(1e-3, 0.9, 0.999, t)), eps=1e-7, reg=0.01,
                               reg_mode=K.REG_TOUCHED)
            if api == "trainer":        # the prefetch / step protocol bench.py uses
                l = tr.step(opt, next_batch=mine[t] if t < len(mine) else None)
            else:
                l = sh.sharded_bpr_step(ops, Pt, Qt, *mine[t - 1], opt)
            dist.all_reduce(l)
            losses.append(float(l))
        Pfull = sh.all_gather_table(Pt).numpy()
        Qfull = sh.all_gather_table(Qt).numpy()
        if rank == 0:
            np.savez(out, P=Pfull, Q=Qfull, losses=np.array(losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,api", [(2, "wrapper"), (2, "trainer"), (3, "trainer")])
@pytest.mark.parametrize("kind", [ro.OPT_SGD, ro.OPT_ADAM])
def test_sharded_step_equals_single_process(tmp_path, world, kind, api):
    seed = 5
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), seed, kind, out, api), nprocs=world, join=True)
    res = np.load(out)
    P, Q, steps = _case(seed, 101, 57, 8, 96)
    sP = [np.zeros_like(P), np.zeros_like(P)] if kind == ro.OPT_ADAM else []
    sQ = [np.zeros_like(Q), np.zeros_like(Q)] if kind == ro.OPT_ADAM else []
    losses = []
    for t, (u, i, j) in enumerate(steps, 1):
        hp = dict(lr=0.05, alpha=float(ro.adam_alpha(1e-3, 0.9, 0.999, t)), beta1=0.9, beta2=0.999, eps=1e-7)
        P, Q, sP, sQ, loss, _ = ro.pairwise_step(P, Q, sP, sQ, u, i, j, kind, hp, 0.01, ro.REG_TOUCHED)
        losses.append(loss)
    assert np.array_equal(res["P"].view(np.uint32), P.view(np.uint32))
    assert np.array_equal(res["Q"].view(np.uint32), Q.view(np.uint32))
    np.testing.assert_allclose(res["losses"], losses, rtol=1e-6)


def test_shard_helpers():
    from deeprec_amd import sharding as sh
    assert [sh.shard_users(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    t = sh.ShardedTable(10, 2, 4, 1, "cpu")
    assert t.local_rows == 3          # rows 1, 5, 9
    t3 = sh.ShardedTable(10, 2, 4, 3, "cpu")
    assert t3.local_rows == 2         # rows 3, 7


@pytest.mark.parametrize("w_save,w_load", [(2, 3), (3, 1), (1, 4), (4, 4)])
def test_sharded_checkpoint_reshards(tmp_path, w_save, w_load):
    """save at one world size, load at another: every global row (table and slots) comes back where
    owner = row % world says it lives."""
    from deeprec_amd import sharding as sh
    rng = np.random.default_rng(w_save * 10 + w_load)
    rows = {"P": 103, "Q": 41}
    k = 6
    full = {n: torch.from_numpy(rng.standard_normal((r, k)).astype(np.float32)) for n, r in rows.items()}
    slot = {n: [torch.from_numpy(rng.standard_normal((r, k)).astype(np.float32)) for _ in range(2)] for n, r in rows.items()}
    prefix = str(tmp_path / "ckpt")
    for r in range(w_save):
        tabs = {}
        for n in rows:
            t = sh.ShardedTable(rows[n], k, w_save, r, "cpu", 2)
            t.load_global(full[n])
            for j in range(2):
                t.slots[j].copy_(slot[n][j][r::w_save])
            tabs[n] = t
        sh.save_sharded(prefix, tabs)
    for r in range(w_load):
        got = sh.load_sharded(prefix, ["P", "Q"], w_load, r, "cpu")
        for n in rows:
            assert torch.equal(got[n].w, full[n][r::w_load])
            for j in range(2):
                assert torch.equal(got[n].slots[j], slot[n][j][r::w_load])


def _eval_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import _oracle_ops as ops
        from deeprec_amd import sharding as sh
        nu, ni, k, K_ = 53, 40, 8, 5
        P, Q, users, train, tgt = _eval_case(nu, ni, k)
        Pt = sh.ShardedTable(nu, k, world, rank, "cpu")
        Qt = sh.ShardedTable(ni, k, world, rank, "cpu")
        Pt.load_global(torch.from_numpy(P))
        Qt.load_global(torch.from_numpy(Q))
        lo, hi = sh.shard_users(len(users), world, rank)
        tp = np.concatenate(([0], np.cumsum([len(t) for t in train[lo:hi]]))).astype(np.int32)
        ti = np.concatenate(train[lo:hi] + [np.zeros(0, np.int32)]).astype(np.int32)
        gp = np.arange(hi - lo + 1, dtype=np.int32)
        idx, val, rk, _ = sh.sharded_eval_topk(ops, Pt, Qt, torch.from_numpy(users[lo:hi]), K_,
                                               train=(torch.from_numpy(tp), torch.from_numpy(ti)),
                                               targets=(torch.from_numpy(gp), torch.from_numpy(tgt[lo:hi])))
        rk2, _ = sh.sharded_eval_ranks(ops, Pt, Qt, torch.from_numpy(users[lo:hi]),
                                       train=(torch.from_numpy(tp), torch.from_numpy(ti)),
                                       targets=(torch.from_numpy(gp), torch.from_numpy(tgt[lo:hi])))
        assert torch.equal(rk2, rk)                 # the rank-only form returns the same ranks
        m = sh.reduce_rank_metrics(rk, K_)
        parts = [None] * world
        dist.all_gather_object(parts, (idx.numpy(), rk.numpy()))
        if rank == 0:
            np.savez(out, idx=np.concatenate([p[0] for p in parts]), rk=np.concatenate([p[1] for p in parts]),
                     m=np.array(m, dtype=np.float64))
    finally:
        dist.destroy_process_group()


def _eval_case(nu, ni, k):
    rng = np.random.default_rng(9)
    P = rng.standard_normal((nu, k)).astype(np.float32)
    Q = rng.standard_normal((ni, k)).astype(np.float32)
    users = rng.permutation(nu)[:37].astype(np.int32)
    train = [np.sort(rng.choice(ni, rng.integers(0, 9), replace=False)).astype(np.int32) for _ in users]
    tgt = rng.integers(0, ni, len(users)).astype(np.int32)
    return P, Q, users, train, tgt


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_eval_equals_single_process(tmp_path, world):
    """users sharded data-parallel, item table all-gathered, user rows fetched from their owners: the same top-K
    lists, target ranks and metric sums as one process over the whole tables."""
    out = str(tmp_path / "ev.npz")
    mp.spawn(_eval_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = np.load(out)
    P, Q, users, train, tgt = _eval_case(53, 40, 8)
    K_ = 5
    S = (P[users].astype(np.float64) @ Q.T.astype(np.float64))
    from oracle import c_oracle as co
    S = co.score_matrix(P[users], Q, 0)
    hits = ndcg = mrr = n = 0
    for x in range(len(users)):
        top, _, rk = ro.topk_and_ranks(S[x], train[x], [tgt[x]], K_)
        assert np.array_equal(res["idx"][x][:len(top)], top)
        assert res["rk"][x] == rk[0]
        if rk[0] >= 0:
            n += 1
            mrr += 1.0 / (rk[0] + 1)
            if rk[0] < K_:
                hits += 1
                ndcg += 1.0 / np.log2(rk[0] + 2)
    np.testing.assert_allclose(res["m"], [n, hits / n, ndcg / n, mrr / n], rtol=1e-12)


def _edge_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import _oracle_ops as ops
        from deeprec_amd import kernels as K
        from deeprec_amd import sharding as sh
        nu, ni, k = 30, 20, 4
        P, Q, u, i, j = _edge_case(nu, ni, k, world)
        Pt = sh.ShardedTable(nu, k, world, rank, "cpu")
        Qt = sh.ShardedTable(ni, k, world, rank, "cpu")
        Pt.load_global(torch.from_numpy(P))
        Qt.load_global(torch.from_numpy(Q))
        opt = K.opt_params(K.OPT_SGD, lr=0.05, reg=0.01, reg_mode=K.REG_TOUCHED)
        B = len(u) // world
        sl = slice(rank * B, (rank + 1) * B)
        # every user row lives on rank 0 and every item row on rank 1: the other ranks process nothing / own nothing
        l = sh.sharded_bpr_step(ops, Pt, Qt, torch.from_numpy(u[sl]), torch.from_numpy(i[sl]), torch.from_numpy(j[sl]), opt)
        dist.all_reduce(l)
        # an id out of range on ONE rank is refused by EVERY rank at the count exchange (nobody is left waiting)
        bad = torch.from_numpy(u[sl]).clone()
        raised = False
        if rank == world - 1:
            bad[0] = nu + 5
        try:
            sh.ShardedBpr(ops, Pt, Qt, pipelined=False).route(bad, torch.from_numpy(i[sl]), torch.from_numpy(j[sl]))
        except IndexError:
            raised = True
        flags = [None] * world
        dist.all_gather_object(flags, raised)
        Pf, Qf = sh.all_gather_table(Pt).numpy(), sh.all_gather_table(Qt).numpy()
        if rank == 0:
            np.savez(out, P=Pf, Q=Qf, loss=float(l), flags=np.array(flags))
    finally:
        dist.destroy_process_group()


def _edge_case(nu, ni, k, world):
    rng = np.random.default_rng(77)
    P = rng.standard_normal((nu, k)).astype(np.float32)
    Q = rng.standard_normal((ni, k)).astype(np.float32)
    n = 4 * world
    u = (rng.integers(0, nu // world, n) * world).astype(np.int32)            # u % world == 0
    i = (rng.integers(0, ni // world, n) * world + 1).astype(np.int32)        # i % world == 1
    j = (rng.integers(0, ni // world, n) * world + 1).astype(np.int32)
    return P, Q, u, i, j


def test_sharded_step_ranks_without_work_and_bad_ids(tmp_path):
    world = 3
    out = str(tmp_path / "edge.npz")
    mp.spawn(_edge_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = np.load(out)
    P, Q, u, i, j = _edge_case(30, 20, 4, world)
    hp = dict(lr=0.05, alpha=0.0, beta1=0.9, beta2=0.999, eps=1e-7)
    P1, Q1, _, _, loss, _ = ro.pairwise_step(P, Q, [], [], u, i, j, ro.OPT_SGD, hp, 0.01, ro.REG_TOUCHED)
    assert np.array_equal(res["P"].view(np.uint32), P1.view(np.uint32))
    assert np.array_equal(res["Q"].view(np.uint32), Q1.view(np.uint32))
    np.testing.assert_allclose(float(res["loss"]), loss, rtol=1e-6)
    assert list(res["flags"]) == [True, True, True]


# ---------------------------------------------------------------------------------------------------------------
# one route per training step of the data-parallel sequence models (sharding.StepRoute)
# ---------------------------------------------------------------------------------------------------------------
def _route_case(seed, world):
    rng = np.random.default_rng(seed)
    R1, R2, k1, k2 = 53, 31, 6, 10
    T1 = rng.standard_normal((R1, k1)).astype(np.float32)
    T2 = rng.standard_normal((R2, k2)).astype(np.float32)
    bv = rng.standard_normal(R2).astype(np.float32)
    lists = [((rng.zipf(1.3, (7, 3)) % R1).astype(np.int32), rng.integers(0, R2, 9).astype(np.int32)) for _ in range(world)]
    lists[0][1][2] = R2 + 5                                     # a row nobody has: fetched as zeros, its gradient dropped
    grads = [[rng.standard_normal((21, k1)).astype(np.float32), rng.standard_normal((9, k2)).astype(np.float32),
              rng.standard_normal(9).astype(np.float32)] for _ in range(world)]
    return T1, T2, bv, lists, grads


def _route_worker(rank, world, port, seed, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import _oracle_ops as ops
        from deeprec_amd import sharding as sh
        from deeprec_amd.models._common import Shard, Table
        T1, T2, bv, lists, grads = _route_case(seed, world)
        shd = Shard(world, rank)
        tabs = [Table(torch.from_numpy(np.ascontiguousarray(a[rank::world])), shd) for a in (T1, T2, bv)]
        # worlds 2 and 3 build the route in its deferred form (the id-only part first, finish() when the step needs it:
        # what Shard.prefetch_route does one step ahead); world 1 in one go
        rt = sh.StepRoute(ops, [torch.from_numpy(x) for x in lists[rank]], world, rank, defer=world > 1)
        if world > 1:
            assert not hasattr(rt, "send_cnt")
            rt.finish()
            rt.finish()         # idempotent
        fr = rt.fetch([(tabs[0], 0), (tabs[1], 1), (tabs[2], 1)])
        # gradients arrive in the order a backward pass would record them: pair 2 first, then 0, then 1
        for f in (2, 0, 1):
            fr[f].pending.append((rt.pos(fr[f].l), torch.from_numpy(grads[rank][f])))
        rt.push()
        res = {"rows%d" % f: fr[f].w.numpy() for f in range(3)}
        for t, name in zip(tabs, ("T1", "T2", "b")):
            assert len(t.pending) == 1
            res["ids_" + name], res["g_" + name] = t.pending[0][0].numpy(), t.pending[0][1].numpy()
        np.savez(out % rank, **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_step_route_fetches_and_pushes_in_global_batch_order(tmp_path, world):
    """sharding.StepRoute over gloo: every (table, list) pair's rows in one exchange == a lookup in the full table;
    the pushed gradient rows arrive at the owner of each row list by list, source rank by source rank, each source's in
    list order -- what a single process records for the concatenated global batch."""
    seed = 21
    out = str(tmp_path / "route%d.npz")
    if world == 1:
        _route_worker(0, 1, _free_port(), seed, out)
    else:
        mp.spawn(_route_worker, args=(world, _free_port(), seed, out), nprocs=world, join=True)
    T1, T2, bv, lists, grads = _route_case(seed, world)
    full = (T1, T2, bv)
    for r in range(world):
        res = np.load(out % r)
        for f, (tab, l) in enumerate(((T1, 0), (T2, 1), (bv, 1))):
            ids = lists[r][l].reshape(-1)
            ok = ids < tab.shape[0]
            want = np.zeros((len(ids),) + tab.shape[1:], np.float32)
            want[ok] = tab[ids[ok]]
            assert np.array_equal(res["rows%d" % f], want), (r, f)
        for name, f, l in (("T1", 0, 0), ("T2", 1, 1), ("b", 2, 1)):
            ids = np.concatenate([lists[s][l].reshape(-1) for s in range(world)])
            g = np.concatenate([grads[s][f] for s in range(world)])
            mine = ids % world == r
            assert np.array_equal(res["ids_" + name], ids[mine] // world), (r, name)
            assert np.array_equal(res["g_" + name], g[mine]), (r, name)
