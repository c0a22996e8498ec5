"""GPU tests of the row-sharded BPR step (deeprec_amd/sharding.py) on the real HIP kernels:

* the routing entry point (dr_shard_bpr_route) against its numpy restatement in tests/_oracle_ops.py, list by list
  and bit for bit;
* one rank through RCCL (process group of size 1, pipelined trainer with its second communicator and stream);
* two ranks sharing the one GPU of the box (gloo, device tensors staged through the host for the exchanges --
  RCCL refuses two ranks on one device), pipelined: tables bit-identical to the fused single-GPU step on the
  concatenated global batch.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from oracle import ref_ops as ro

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------ routing kernels vs numpy restatement
@pytest.mark.parametrize("B,W,ni,k", [(1, 1, 10, 8), (1000, 1, 701, 64), (4096, 2, 700, 64), (20001, 8, 100000, 128),
                                      (5000, 3, 17, 32), (300, 5, 3, 4), (70000, 4, 1000003, 16)])
def test_shard_route(dev, B, W, ni, k):
    """dr_shard_bpr_route for every rank of the world against the numpy restatement (tests/_oracle_ops.py), list by
    list and bit for bit; and the property the exchanges rest on: the rows rank r fetches from owner o are, entry for
    entry, the rows o serves to r, and every gradient row r sends to o has a place in o's receive order."""
    from deeprec_amd import kernels as K
    import _oracle_ops as O
    rng = np.random.default_rng(B + W)
    nu = 100003
    ids = np.stack((rng.integers(0, nu, (W, B)), rng.zipf(1.2, (W, B)) % ni, rng.integers(0, ni, (W, B))), 1).astype(np.int32)
    views = []
    for me in range(W):
        ws = K.ShardWorkspace(W, me, B, k, nu, ni, dev)
        K.shard_route(ws, T(ids, dev))
        wo = O.ShardWorkspace(W, me, B, k, nu, ni)
        O.shard_route(wo, torch.from_numpy(ids))
        c = wo.counts.numpy()
        assert np.array_equal(N(ws.counts), c), me
        assert c[4] == 0 and c[5] == 0
        n_s, n_g = wo.n_s, wo.n_g
        nuL = (nu - me + W - 1) // W
        ckey = N(ws.list(2)).view(np.uint32)
        assert np.array_equal(ckey[:n_s], wo.u_loc), me
        assert np.array_equal(ckey[n_s:n_s + n_g], nuL + wo.okey), me
        assert np.array_equal(N(ws.list(3))[n_s:n_s + n_g], wo.osrc), me
        assert np.array_equal(N(ws.list(4))[:, :n_s], wo.ref), me
        assert np.array_equal(N(ws.list(5))[:, :n_s], wo.ow), me
        assert np.array_equal(N(ws.serve)[:wo.n_serve], wo.serve.numpy()[:wo.n_serve]), me
        assert np.all(N(ws.list(7))[n_s:n_s + n_g][wo.osrc < 0] == 1.0)
        # solo flags: the key occurs once in the combined list and its one contribution is local
        keys = np.concatenate((wo.u_loc.astype(np.int64), nuL + wo.okey.astype(np.int64)))
        _, inv, cnt = np.unique(keys, return_inverse=True, return_counts=True)
        single = cnt[inv] == 1
        solo = N(ws.list(6))
        assert np.array_equal(solo[0, :n_s], single[:n_s].astype(np.uint8)), me
        want = np.zeros((2, n_s), np.uint8)
        loc = wo.osrc >= 0
        want[wo.osrc[loc] & 1, wo.osrc[loc] >> 1] = single[n_s:][loc]
        assert np.array_equal(solo[1:, :n_s], want), me
        views.append((wo, c))
    # sender / receiver agreement, from the restatement's lists
    niLs = [(ni - r + W - 1) // W for r in range(W)]
    for r in range(W):
        wr, cr = views[r]
        fo = np.concatenate(([0], np.cumsum(cr[8:8 + W])))              # fetch-buffer offsets per owner
        go = np.concatenate(([0], np.cumsum(cr[8 + W:8 + 2 * W])))      # send-buffer offsets per owner
        rem = wr.ow < 0
        item = np.stack((ids[:, 1, :].reshape(-1), ids[:, 2, :].reshape(-1)))      # [part, global sample]
        S = np.nonzero(ids[:, 0, :].reshape(-1) % W == r)[0]
        for o in range(W):
            wo_, co = views[o]
            so = np.concatenate(([0], np.cumsum(co[8 + 2 * W:8 + 3 * W])))         # serve-list offsets per processing rank
            ro_ = np.concatenate(([0], np.cumsum(co[8 + 3 * W:8 + 4 * W])))        # receive-buffer offsets
            assert cr[8 + o] == co[8 + 2 * W + r] and cr[8 + W + o] == co[8 + 3 * W + r], (r, o)
            if o == r:
                assert cr[8 + o] == 0 and cr[8 + W + o] == 0
                continue
            for part in range(2):
                sel = rem[part] & (item[part][S] % W == o)
                seg = wr.ref[part][sel] - niLs[r] - fo[o]                # index into the rows o serves to r
                assert np.all((seg >= 0) & (seg < cr[8 + o]))
                served = wo_.serve.numpy()[so[r]:so[r + 1]]
                assert np.array_equal(served[seg], item[part][S][sel] // W), (r, o, part)
                wire = -1 - wr.ow[part][sel] - go[o]                     # position among the rows r sends to o
                assert np.all((wire >= 0) & (wire < cr[8 + W + o]))
                # o's entry that takes receive row ro_[r] + wire is the same (part, sample) entry
                recv_of = {int(-1 - x): c_ for c_, x in enumerate(wo_.osrc) if x < 0}
                E = np.nonzero(np.concatenate((item[0], item[1])) % W == o)[0]     # o's owned entries, canonical order
                for w_, s_ in zip(wire[:50], np.nonzero(sel)[0][:50]):
                    e = E[recv_of[int(ro_[r] + w_)]]
                    assert e == part * W * B + S[s_], (r, o, part)


def test_shard_route_refuses_bad_ids_and_overload(dev):
    """An id out of range anywhere in the global batch, or a rank over its capacity, is reported to EVERY rank (counts [4] /
    [5]) and the trainer refuses the batch; an overloaded batch writes no list at all (every length is reported as zero)."""
    from deeprec_amd import kernels as K
    from deeprec_amd import sharding as sh
    rng = np.random.default_rng(3)
    W, B, nu, ni, k = 4, 512, 4001, 900, 16
    ids = np.stack((rng.integers(0, nu, (W, B)), rng.integers(0, ni, (W, B)), rng.integers(0, ni, (W, B))), 1).astype(np.int32)
    bad = ids.copy()
    bad[0, 0, 3], bad[W - 1, 1, 5], bad[1, 2, 7] = -1, ni, -5
    for me in range(W):
        ws = K.ShardWorkspace(W, me, B, k, nu, ni, dev)
        K.shard_route(ws, T(bad, dev))
        assert int(ws.counts[4]) == 3
    over = ids.copy()
    over[:, 0, :] = (over[:, 0, :] // W) * W                        # every user row lives on rank 0
    for me in range(W):
        ws = K.ShardWorkspace(W, me, B, k, nu, ni, dev, cap_s=B + 100, cap_g=4 * B)
        K.shard_route(ws, T(over, dev))
        c = N(ws.counts)
        assert c[5] == 1 and c[0] == 0 and c[1] == 0 and c[3] == 0 and c[8 + 4 * W] == W * B
    Pt = sh.ShardedTable(nu, k, 1, 0, dev, 2)
    Qt = sh.ShardedTable(ni, k, 1, 0, dev, 2)
    tr = sh.ShardedBpr(K, Pt, Qt, pipelined=False)
    with pytest.raises(IndexError):
        tr.route(T(bad[0, 0], dev), T(bad[0, 1], dev), T(bad[0, 2], dev))


# ------------------------------------------------------------------ whole steps
def _case(seed, nu, ni, k, Bg, steps):
    rng = np.random.default_rng(seed)
    P = (rng.standard_normal((nu, k)) * 0.1).astype(np.float32)
    Q = (rng.standard_normal((ni, k)) * 0.1).astype(np.float32)
    out = []
    for _ in range(steps):
        out.append((rng.integers(0, nu, Bg).astype(np.int32), (rng.zipf(1.2, Bg) % ni).astype(np.int32),
                    rng.integers(0, ni, Bg).astype(np.int32)))
    return P, Q, out


def _opt(K, t):
    return K.opt_params(K.OPT_ADAM, lr=0.05, alpha=float(ro.adam_alpha(1e-3, 0.9, 0.999, t)), eps=1e-7, reg=0.01,
                        reg_mode=K.REG_TOUCHED)


def _fused_reference(dev, P0, Q0, batches):
    from deeprec_amd import kernels as K
    P, Q = T(P0, dev), T(Q0, dev)
    sP = [torch.zeros_like(P), torch.zeros_like(P)]
    sQ = [torch.zeros_like(Q), torch.zeros_like(Q)]
    losses = []
    for t, (u, i, j) in enumerate(batches, 1):
        ws = K.StepWorkspace(len(u), P.shape[1], P.shape[0], Q.shape[0], K.REG_TOUCHED, dev)
        losses.append(float(K.bpr_step(P, Q, sP, sQ, T(u, dev), T(i, dev), T(j, dev), _opt(K, t), ws)))
    return N(P), N(Q), N(sP[1]), N(sQ[1]), losses


def _train_sharded(rank, world, dev, P0, Q0, batches, pipelined):
    """Run the trainer on this rank's slice of every global batch; returns (full P, full Q, full v_P, full v_Q, losses)."""
    import torch.distributed as dist
    from deeprec_amd import kernels as K
    from deeprec_amd import sharding as sh
    k = P0.shape[1]
    Pt = sh.ShardedTable(P0.shape[0], k, world, rank, dev, 2)
    Qt = sh.ShardedTable(Q0.shape[0], k, world, rank, dev, 2)
    Pt.load_global(T(P0, dev))
    Qt.load_global(T(Q0, dev))
    tr = sh.ShardedBpr(K, Pt, Qt, pipelined=pipelined)
    Bl = len(batches[0][0]) // world
    mine = [tuple(T(x[rank * Bl:(rank + 1) * Bl], dev) for x in b) for b in batches]
    losses = []
    tr.prefetch(*mine[0])
    for t in range(len(mine)):
        l = tr.step(_opt(K, t + 1), next_batch=mine[t + 1] if t + 1 < len(mine) else None)
        if world > 1:
            l = l.cpu()
            dist.all_reduce(l)
        losses.append(float(l))
    torch.cuda.synchronize()

    def full(T_, x):
        if world == 1:
            return N(x)
        per = (T_.rows + world - 1) // world
        pad = torch.zeros((per, k), dtype=torch.float32)
        pad[:x.shape[0]] = x.cpu()
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        return torch.stack(parts, 1).reshape(per * world, k)[:T_.rows].numpy()

    return full(Pt, Pt.w), full(Qt, Qt.w), full(Pt, Pt.slots[1]), full(Qt, Qt.slots[1]), losses


def _same(got, want, what):
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what


def test_world1_through_rccl_pipelined_equals_fused_step(dev):
    """One rank, but every exchange goes through RCCL (process group of size 1) and the trainer is pipelined: the
    route of batch t+1 runs on the side stream / second communicator while batch t's kernels execute."""
    import torch.distributed as dist
    P0, Q0, batches = _case(31, 6000, 900, 64, 4096, 5)
    want = _fused_reference(dev, P0, Q0, batches)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        got = _train_sharded(0, 1, dev, P0, Q0, batches, pipelined=True)
    finally:
        dist.destroy_process_group()
    for x, y, nm in zip(got[:4], want[:4], ("P", "Q", "v_P", "v_Q")):
        _same(x, y, nm)
    np.testing.assert_allclose(got[4], want[4], rtol=1e-6)


def _stage_through_host(sh, dist):
    def staged(o, inp, out_splits, in_splits, group, async_op=False):
        oc = torch.empty(o.shape, dtype=o.dtype)              # gloo has no device all-to-all: stage through the host
        dist.all_to_all_single(oc, inp.cpu(), out_splits, in_splits, group=group)
        o.copy_(oc)

        class Done:
            def wait(self):
                return True
        return Done() if async_op else None

    def gather_staged(o, inp, group):
        oc = torch.empty(o.shape, dtype=o.dtype)
        dist.all_gather_into_tensor(oc.view(-1), inp.cpu().view(-1), group=group)
        o.copy_(oc)

    def reduce_staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)

    sh._a2a = staged
    sh._gather_ids = gather_staged
    sh._all_reduce = reduce_staged


def _worker2(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deeprec_amd import sharding as sh
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)

        _stage_through_host(sh, dist)
        P0, Q0, batches = _case(33, 5001, 801, 64, 4096, 4)
        res = _train_sharded(rank, world, dev, P0, Q0, batches, pipelined=True)
        if rank == 0:
            np.savez(out, P=res[0], Q=res[1], vP=res[2], vQ=res[3], losses=np.array(res[4]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_on_one_gpu_equal_fused_step(dev, tmp_path, world):
    """world ranks x real kernels x pipelined trainer on the box's single GPU: bit-identical to the fused
    single-GPU step over the concatenated global batch (identical results at every GPU count)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "res.npz")
    P0, Q0, batches = _case(33, 5001, 801, 64, 4096, 4)
    Bl = 4096 // world
    batches = [tuple(x[:Bl * world] for x in b) for b in batches]
    want = _fused_reference(dev, P0, Q0, batches)
    mp.spawn(_worker2, args=(world, _free_port(), out), nprocs=world, join=True)
    res = np.load(out)
    for nm, y in zip(("P", "Q", "vP", "vQ"), want[:4]):
        _same(res[nm], y, nm)
    np.testing.assert_allclose(res["losses"], want[4], rtol=1e-6)


def test_c4_size_sharded_step_world1_rccl_equals_fused_step(dev):
    """BASELINE config 4 at full size on the sharded path: 10 M x 1 M x 128, B = 65 536, one rank with the process group
    up (RCCL, world 1), pipelined trainer, 2 steps: tables and every optimizer slot equal the fused single-GPU step's
    (dr_bpr_plan_f32 + dr_bpr_step_planned_f32) bit for bit, the loss sums too."""
    import torch.distributed as dist
    from deeprec_amd import kernels as K
    from deeprec_amd import sharding as sh
    import bench
    nu, ni, k, B = 10_000_000, 1_000_000, 128, 65536

    class A:
        users, items, batch = nu, ni, B
    batches = bench.make_batches(A, 0, dev, 3)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        Pt = sh.ShardedTable(nu, k, 1, 0, dev, 2)
        Qt = sh.ShardedTable(ni, k, 1, 0, dev, 2)
        Pt.init_normal_(K, 1, 0.01)
        Qt.init_normal_(K, 2, 0.01)
        tr = sh.ShardedBpr(K, Pt, Qt, pipelined=True)
        tr.prefetch(*batches[0])
        losses = [float(tr.step(_opt(K, t + 1), next_batch=batches[t + 1])) for t in range(2)]
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    P = torch.empty((nu, k), dtype=torch.float32, device=dev)
    Q = torch.empty((ni, k), dtype=torch.float32, device=dev)
    K.init_normal_(P, 1, 0.01)
    K.init_normal_(Q, 2, 0.01)
    sP = [torch.zeros_like(P), torch.zeros_like(P)]
    sQ = [torch.zeros_like(Q), torch.zeros_like(Q)]
    ws = K.StepWorkspace(B, k, nu, ni, K.REG_TOUCHED, dev)
    want = []
    for t in range(2):
        K.bpr_plan(nu, ni, k, *batches[t], ws)
        want.append(float(K.bpr_step(P, Q, sP, sQ, *batches[t], _opt(K, t + 1), ws, planned=True)))
    assert losses == want
    for got, ref, nm in ((Pt.w, P, "P"), (Qt.w, Q, "Q"), (Pt.slots[0], sP[0], "m_P"), (Pt.slots[1], sP[1], "v_P"),
                         (Qt.slots[0], sQ[0], "m_Q"), (Qt.slots[1], sQ[1], "v_Q")):
        assert torch.equal(got, ref), nm


# ------------------------------------------------------------------ data-parallel evaluation
def _eval_case():
    rng = np.random.default_rng(41)
    nu, ni, k = 3001, 2500, 64
    P = rng.standard_normal((nu, k)).astype(np.float32)
    Q = rng.standard_normal((ni, k)).astype(np.float32)
    users = rng.permutation(nu)[:300].astype(np.int32)
    train = [np.sort(rng.choice(ni, 50, replace=False)).astype(np.int32) for _ in users]
    tgt = rng.integers(0, ni, len(users)).astype(np.int32)
    return P, Q, users, train, tgt


def _csr_of(lists):
    ptr = np.concatenate(([0], np.cumsum([len(t) for t in lists]))).astype(np.int32)
    idx = np.concatenate(list(lists) + [np.zeros(0, np.int32)]).astype(np.int32)
    return ptr, idx


def _eval_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deeprec_amd import kernels as K
        from deeprec_amd import sharding as sh
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        _stage_through_host(sh, dist)
        sh.all_gather_table = lambda T_, group=None: _gather_table_via_host(T_, dist)
        P, Q, users, train, tgt = _eval_case()
        k = P.shape[1]
        Pt = sh.ShardedTable(P.shape[0], k, world, rank, dev)
        Qt = sh.ShardedTable(Q.shape[0], k, world, rank, dev)
        Pt.load_global(T(P, dev))
        Qt.load_global(T(Q, dev))
        lo, hi = sh.shard_users(len(users), world, rank)
        tp, ti = _csr_of(train[lo:hi])
        gp = np.arange(hi - lo + 1, dtype=np.int32)
        idx, val, rk, _ = sh.sharded_eval_topk(K, Pt, Qt, T(users[lo:hi], dev), 10, train=(T(tp, dev), T(ti, dev)),
                                               targets=(T(gp, dev), T(tgt[lo:hi], dev)))
        rk2, _ = sh.sharded_eval_ranks(K, Pt, Qt, T(users[lo:hi], dev), train=(T(tp, dev), T(ti, dev)),
                                       targets=(T(gp, dev), T(tgt[lo:hi], dev)))
        assert torch.equal(rk2, rk)                 # the rank-only form (dr_score_rank_f32) returns the same ranks
        m = sh.reduce_rank_metrics(rk.cpu(), 10)
        parts = [None] * world
        dist.all_gather_object(parts, (N(idx), N(val), N(rk)))
        if rank == 0:
            np.savez(out, idx=np.concatenate([p[0] for p in parts]), val=np.concatenate([p[1] for p in parts]),
                     rk=np.concatenate([p[2] for p in parts]), m=np.array(m, dtype=np.float64))
    finally:
        dist.destroy_process_group()


def _gather_table_via_host(T_, dist):
    per = (T_.rows + T_.world - 1) // T_.world
    pad = torch.zeros((per, T_.k), dtype=torch.float32)
    pad[:T_.local_rows] = T_.w.cpu()
    parts = [torch.empty_like(pad) for _ in range(T_.world)]
    dist.all_gather(parts, pad)
    return torch.stack(parts, 1).reshape(per * T_.world, T_.k)[:T_.rows].contiguous().to(T_.w.device)


def test_sharded_eval_two_ranks_equals_single_gpu(dev, tmp_path):
    """evaluation users sharded over 2 ranks (sharing the box's GPU), item table replicated, user rows fetched from
    their owners: top-K ids, scores and target ranks are those of one dr_score_topk_f32 call over everything."""
    import torch.multiprocessing as mp
    from deeprec_amd import kernels as K
    out = str(tmp_path / "ev.npz")
    mp.spawn(_eval_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = np.load(out)
    P, Q, users, train, tgt = _eval_case()
    tp, ti = _csr_of(train)
    gp = np.arange(len(users) + 1, dtype=np.int32)
    U = K.gather(T(P, dev), T(users, dev))
    idx, val, rk = K.score_topk(U, T(Q, dev), 10, train=(T(tp, dev), T(ti, dev)), targets=(T(gp, dev), T(tgt, dev)))
    assert np.array_equal(res["idx"], N(idx)) and np.array_equal(res["rk"], N(rk))
    assert np.array_equal(res["val"].view(np.uint32), N(val).view(np.uint32))
    r = N(rk).astype(np.float64)
    ok, hit = r >= 0, (r >= 0) & (r < 10)
    want = [ok.sum(), hit.sum() / ok.sum(), (1 / np.log2(r[hit] + 2)).sum() / ok.sum(), (1 / (r[ok] + 1)).sum() / ok.sum()]
    np.testing.assert_allclose(res["m"], want, rtol=1e-12)


# ------------------------------------------------------------------ config 5: data-parallel AttRec / Caser
def _seq_case(cls_name):
    rng = np.random.default_rng(51)
    nu, ni, L, T_, d, B = 400, 300, 6, 2, 32, 64
    init = {}
    if cls_name == "AttRec":
        for n, r in (("P", nu), ("V", nu), ("Q", ni), ("X", ni), ("A", ni)):
            init[n] = (rng.standard_normal((r, d)) * 0.05).astype(np.float32)
        init["W"] = (rng.standard_normal((ni, 2 * d)) * 0.01).astype(np.float32)
        init["b"] = (rng.standard_normal(ni) * 0.01).astype(np.float32)
        init["linear_project"] = (rng.standard_normal((d, d)) * 0.2).astype(np.float32)
        kw = dict(num_factor=d, num_neg=1)
    else:
        n_h, n_v = 4, 2
        init["P"] = (rng.standard_normal((nu, d)) * 0.05).astype(np.float32)
        init["V"] = (rng.standard_normal((nu, d * L)) * 0.05).astype(np.float32)
        init["Q"] = (rng.standard_normal((ni, d)) * 0.05).astype(np.float32)
        init["W"] = (rng.standard_normal((ni, 2 * d)) * 0.05).astype(np.float32)
        init["b"] = (rng.standard_normal(ni) * 0.05).astype(np.float32)
        init["conv_v_w"] = (rng.standard_normal((n_v, 1, L, 1)) * 0.3).astype(np.float32)
        init["conv_v_b"] = np.zeros(n_v, np.float32)
        for h in range(1, L + 1):
            init["conv_h_w%d" % h] = (rng.standard_normal((n_h, 1, h, d)) * 0.1).astype(np.float32)
            init["conv_h_b%d" % h] = np.zeros(n_h, np.float32)
        init["full_w"] = (rng.standard_normal((n_v * d + n_h * L, d)) * 0.1).astype(np.float32)
        init["full_b"] = np.zeros(d, np.float32)
        kw = dict(num_factor=d, num_neg=1, n_h=n_h, n_v=n_v)
    steps = []
    for _ in range(3):
        steps.append((rng.integers(0, nu, B).astype(np.int32), (rng.zipf(1.3, (B, L)) % (ni - 1) + 1).astype(np.int32),
                      rng.integers(1, ni, (B, T_)).astype(np.int32), rng.integers(1, ni, (B, T_)).astype(np.int32)))
    return nu, ni, L, T_, B, init, kw, steps


def _seq_model(cls_name, dev, shard=None):
    import contextlib
    import io
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    nu, ni, L, T_, B, init, kw, steps = _seq_case(cls_name)
    cls = AttRec if cls_name == "AttRec" else Caser
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls(None, nu, ni, 0.05 if cls is AttRec else 0.001, 0.01, 1, B, device=dev, shard=shard)
        m.build_network(L, T_, init=init, **kw)
    return m, steps


def _seq_worker(rank, world, port, cls_name, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deeprec_amd import sharding as sh
        from deeprec_amd.models._common import Shard
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        _stage_through_host(sh, dist)
        # count what one step costs: collectives (the staged exchange functions) and host synchronisations of the route
        calls = {"collectives": 0}
        for name in ("_a2a", "_gather_ids", "_all_reduce"):
            def counted(*a, _f=getattr(sh, name), **kw):
                calls["collectives"] += 1
                return _f(*a, **kw)
            setattr(sh, name, counted)
        m, steps = _seq_model(cls_name, dev, Shard(world, rank))
        per = len(steps[0][0]) // world
        losses = []
        mines = [tuple(T(x[rank * per:(rank + 1) * per], dev) for x in b) for b in steps]
        total = 0
        for i, mine in enumerate(mines):
            calls["collectives"] = 0
            if i + 1 < len(mines):
                m.prefetch(*mines[i + 1])         # the next step's route (ids only) is enqueued ahead of this step
                assert calls["collectives"] == 1
            calls["collectives"] = 0
            l = m.train_op(*mine).cpu()
            # id all-gather + rows in + gradient rows out + ONE all-reduce of the dense gradients, for 4 lookups in 4 tables;
            # from the second step on the id all-gather has happened in the previous step's prefetch
            assert calls["collectives"] == (4 if i == 0 else 3), calls
            total += calls["collectives"] + (1 if i + 1 < len(mines) else 0)
            dist.all_reduce(l)
            losses.append(float(l))
        assert total == 4 * len(mines)
        res = {}
        names = [n for n in ("P", "V", "Q", "X", "W", "b") if hasattr(m, n)]
        for n in names:
            w = getattr(m, n).w.cpu()
            rows = _seq_case(cls_name)[5][n].shape[0]
            per_r = (rows + world - 1) // world
            pad = torch.zeros((per_r,) + tuple(w.shape[1:]))
            pad[:w.shape[0]] = w
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            res[n] = torch.stack(parts, 1).reshape((per_r * world,) + tuple(w.shape[1:]))[:rows].numpy()
        for n, p in _seq_dense(cls_name, m).items():
            res[n] = p.detach().cpu().numpy()
        if rank == 0:
            np.savez(out, losses=np.array(losses), **res)
    finally:
        dist.destroy_process_group()


def _seq_dense(cls_name, m):
    if cls_name == "AttRec":
        return {"linear_project": m.linear_project}
    return {"full_w": m.full_w, "full_b": m.full_b, "conv_v_w": m.conv_v_w, "conv_v_b": m.conv_v_b,
            "conv_h_wcat": m.conv_h_wcat, "conv_h_bcat": m.conv_h_bcat}


def _sliced_step(m, batch, world, dev):
    """What `world` data-parallel ranks compute, run by ONE process on unsharded tables: forward + backward slice by
    slice (the dense layers see the per-rank shapes, the dense gradients accumulate slice 0 + slice 1 like the all-reduce
    of two ranks), the recorded table gradients regrouped lookup by lookup with the slices in rank order -- the order a
    single process sees for the concatenated batch -- and one optimizer application."""
    from deeprec_amd.models._common import Table
    per = len(batch[0]) // world
    m._zero_dense_grads()
    m.loss_scale = 1.0 / world                                  # Caser's loss is a batch mean (AttRec's a sum: unused)
    loss = 0.0
    for r in range(world):
        loss += float(m._forward_backward(*(T(x[r * per:(r + 1) * per], dev) for x in batch)))
    for t in vars(m).values():
        if isinstance(t, Table) and t.pending:
            nl = len(t.pending) // world
            t.pending = [(torch.cat([t.pending[r * nl + j][0] for r in range(world)]),
                          torch.cat([t.pending[r * nl + j][1] for r in range(world)])) for j in range(nl)]
    m._apply_step()
    return loss


@pytest.mark.parametrize("cls_name", ["AttRec", "Caser"])
def test_data_parallel_seq_models_equal_single_gpu(dev, tmp_path, cls_name):
    """BASELINE config 5: the batch split over 2 ranks (sharing the box's GPU), every table row-sharded, ONE route per
    step (sharding.StepRoute: 3 collectives + the dense all-reduce, asserted in the workers), gradient rows applied in
    global batch order.  After 3 steps
    (a) tables AND dense weights are BIT-IDENTICAL to one process that runs the same two batch slices through the same
        kernels (_sliced_step) -- the exchange adds, drops and reorders nothing;
    (b) against the single-GPU model fed the whole batch at once they agree to fp32 tolerance only (2e-4 relative):
        the dense layers (projection / convolutions / FC, library GEMMs and batch reductions) see other shapes per
        rank and sum the batch in another order, which no exchange can undo."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "seq.npz")
    mp.spawn(_seq_worker, args=(2, _free_port(), cls_name, out), nprocs=2, join=True)
    res = np.load(out)
    tables = [n for n in ("P", "V", "Q", "X", "W", "b")]
    # (a) bit for bit
    m, steps = _seq_model(cls_name, dev)
    losses = [_sliced_step(m, b, 2, dev) for b in steps]
    np.testing.assert_allclose(res["losses"], losses, rtol=1e-6)
    for n in tables:
        if hasattr(m, n):
            _same(res[n], N(getattr(m, n).w), cls_name + "." + n)
    for n, p in _seq_dense(cls_name, m).items():
        _same(res[n], N(p), cls_name + "." + n)
    # (b) the whole batch at once
    m, steps = _seq_model(cls_name, dev)
    losses = [float(m.train_op(*(T(x, dev) for x in b))) for b in steps]
    np.testing.assert_allclose(res["losses"], losses, rtol=2e-5)
    for n in tables:
        if hasattr(m, n):
            np.testing.assert_allclose(res[n], N(getattr(m, n).w), rtol=2e-4, atol=2e-6, err_msg=n)
    for n, p in _seq_dense(cls_name, m).items():
        np.testing.assert_allclose(res[n], N(p), rtol=2e-4, atol=2e-6, err_msg=n)


@pytest.mark.parametrize("cls_name", ["AttRec", "Caser"])
def test_seq_models_world1_through_rccl_equal_unsharded(dev, cls_name):
    """Shard(world 1, force=True): the one-route step with every exchange through RCCL (size-1 communicator) is
    bit-identical to the plain single-GPU step (same kernels on the same rows in the same order)."""
    import torch.distributed as dist
    from deeprec_amd.models._common import Shard
    m0, steps = _seq_model(cls_name, dev)
    want = [float(m0.train_op(*(T(x, dev) for x in b))) for b in steps]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        m1, _ = _seq_model(cls_name, dev, Shard(1, 0, force=True))
        assert m1.shard is not None and m1.Q.shard is not None
        got = [float(m1.train_op(*(T(x, dev) for x in b))) for b in steps]
        # the same with every route after the first enqueued one step ahead (Shard.prefetch_route): the route depends on
        # the ids only, so nothing may change
        m2, _ = _seq_model(cls_name, dev, Shard(1, 0, force=True))
        bs = [tuple(T(x, dev) for x in b) for b in steps]
        got2 = []
        for i, b in enumerate(bs):
            if i + 1 < len(bs):
                m2.prefetch(*bs[i + 1])
                assert len(m2.shard._pre) == 1 + (i > 0)
            got2.append(float(m2.train_op(*b)))
        assert m2.shard._pre == []              # every prefetched route was taken over by its step
    finally:
        dist.destroy_process_group()
    assert got == want and got2 == want
    for n in ("P", "V", "Q", "X", "W", "b"):
        if hasattr(m0, n):
            _same(N(getattr(m2, n).w), N(getattr(m0, n).w), cls_name + ".prefetch." + n)
    for n in ("P", "V", "Q", "X", "W", "b"):
        if hasattr(m0, n):
            _same(N(getattr(m1, n).w), N(getattr(m0, n).w), cls_name + "." + n)
    for (n, a), b in zip(_seq_dense(cls_name, m1).items(), _seq_dense(cls_name, m0).values()):
        _same(N(a), N(b), cls_name + "." + n)


@pytest.mark.parametrize("cls_name", ["AttRec", "Caser"])
def test_seq_epoch_world1_through_rccl_prefetches_routes_and_equals_unsharded(dev, tmp_path, cls_name):
    """model.train() (one epoch of _seq_common._train_epoch) with Shard(1, 0, force=True): every route after the first is
    enqueued one step ahead of its step and finished by it; tables after the epoch are
    bit-identical to the unsharded model's."""
    import contextlib
    import io
    import torch.distributed as dist
    from deeprec_amd import sharding as sh
    from deeprec_amd.models._common import Shard
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    from deeprec_amd.utils.load_data.load_data_seq import DataSet
    rng = np.random.default_rng(9)
    nu, ni, L, T_ = 60, 50, 3, 2
    lines = []
    for u in range(nu):
        for it in rng.permutation(ni)[:int(rng.integers(L + T_ + 2, L + T_ + 8))]:
            lines.append("%d %d 1\n" % (u + 1, it + 1))
    ptr = tmp_path / "train.txt"
    ptr.write_text("".join(lines))
    pte = tmp_path / "test.txt"
    pte.write_text("".join("%d %d 1\n" % (u + 1, int(rng.integers(1, ni + 1))) for u in range(nu)))
    with contextlib.redirect_stdout(io.StringIO()):
        train = DataSet(str(ptr), ['user', 'item', 'rating'], " ", L, T_, True)
        test = DataSet(str(pte), ['user', 'item', 'rating'], " ", 1, 1, False, train.user_map, train.item_map,
                       train.num_user, train.num_item)
    cls = AttRec if cls_name == "AttRec" else Caser
    kw = dict(num_factor=16) if cls is AttRec else dict(num_factor=16, n_h=4, n_v=2)

    def epoch(shard):
        with contextlib.redirect_stdout(io.StringIO()):
            torch.manual_seed(1)
            np.random.seed(1)
            m = cls(None, train.num_user, train.num_item, 0.05 if cls is AttRec else 0.001, 0.01, 1, 16, shard=shard)
            m.build_network(L, T_, **kw)
            m.prepare_data(train, test)
            np.random.seed(2)
            m.train(train)
        return m
    m0 = epoch(None)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        built = {"ahead": 0, "all": 0}
        init = sh.StepRoute.__init__

        def counting(self, *a, defer=False, **k):
            built["all"] += 1
            built["ahead"] += bool(defer)
            return init(self, *a, defer=defer, **k)
        sh.StepRoute.__init__ = counting
        try:
            m1 = epoch(Shard(1, 0, force=True))
        finally:
            sh.StepRoute.__init__ = init
        assert m1.total_batch >= 3 and built["all"] == m1.total_batch and built["ahead"] == m1.total_batch - 1, built
        assert m1.shard._pre == []
    finally:
        dist.destroy_process_group()
    for n in ("P", "V", "Q", "X", "W", "b"):
        if hasattr(m0, n):
            _same(N(getattr(m1, n).w), N(getattr(m0, n).w), cls_name + ".epoch." + n)


# ------------------------------------------------------------------ the route's kernels vs their numpy restatement
@pytest.mark.parametrize("W,sizes", [(1, [5]), (2, [64 * 6, 64, 128]), (3, [1, 0, 7]), (8, [4096 * 50, 4096, 4096 * 6]),
                                     (16, [1000] * 16)])
def test_step_route_keys_and_wire_kernels(dev, W, sizes):
    """dr_step_route_keys_i32, dr_rows_pack_f32 and dr_rows_unpack_f32 against tests/_oracle_ops.py, bit for bit, for
    every rank of the world; rows of width 1 (a bias vector), 6, 128 and 150 share one wire buffer."""
    from deeprec_amd import kernels as K
    import _oracle_ops as O
    rng = np.random.default_rng(W + len(sizes))
    start = np.concatenate(([0], np.cumsum(sizes))).tolist()
    n_tot = start[-1]
    ids = rng.integers(-3, 100000, (W, n_tot)).astype(np.int32)
    for me in sorted({0, W // 2, W - 1}):
        got = K.step_route_keys(T(ids, dev), me, start)
        want = O.step_route_keys(torch.from_numpy(ids), me, start)
        for g, w, nm in zip(got, want, ("keys_mine", "keys_owned", "local_rows")):
            assert np.array_equal(N(g), w.numpy()), (me, nm)
    # wire: segments of different row widths interleaved per peer, padded to 4 floats
    n = min(n_tot, 3000)
    if n == 0:
        return
    for k in (1, 6, 128, 150):
        cuts = np.sort(rng.integers(0, n + 1, W - 1)) if W > 1 else np.zeros(0, np.int64)
        seg_start = [0] + [int(c) for c in cuts] + [n]
        seg_off, off = [], 8
        for s_ in range(W):
            seg_off.append(off)
            off += ((seg_start[s_ + 1] - seg_start[s_]) * k + 3) // 4 * 4 + 4 * int(rng.integers(0, 3))
        rows = 2 * n + 5
        src = rng.standard_normal((rows, k) if k > 1 else rows).astype(np.float32)
        idx = (rng.integers(-2, rows + 2, n) + 7).astype(np.int32)
        flat_g = torch.full((off,), -1.0, device=dev)
        flat_o = torch.full((off,), -1.0)
        K.rows_pack(T(src, dev), T(idx, dev), n, seg_start, seg_off, flat_g, idx_bias=7)
        O.rows_pack(torch.from_numpy(src), torch.from_numpy(idx), n, seg_start, seg_off, flat_o, idx_bias=7)
        _same(N(flat_g), flat_o.numpy(), ("pack", k))
        K.rows_pack(T(src, dev), None, n, seg_start, seg_off, flat_g)
        O.rows_pack(torch.from_numpy(src), None, n, seg_start, seg_off, flat_o)
        _same(N(flat_g), flat_o.numpy(), ("pack, no index", k))
        perm = (rng.permutation(rows)[:n] + 7).astype(np.int32)
        perm[0] = 3                                                 # below the bias: skipped
        dst_g = torch.full(src.shape, -2.0, device=dev)
        dst_o = torch.full(src.shape, -2.0)
        K.rows_unpack(flat_g, T(perm, dev), n, seg_start, seg_off, dst_g, idx_bias=7)
        O.rows_unpack(flat_o, torch.from_numpy(perm), n, seg_start, seg_off, dst_o, idx_bias=7)
        _same(N(dst_g), dst_o.numpy(), ("unpack", k))
        K.rows_unpack(flat_g, None, n, seg_start, seg_off, dst_g)
        O.rows_unpack(flat_o, None, n, seg_start, seg_off, dst_o)
        _same(N(dst_g), dst_o.numpy(), ("unpack, no index", k))
    with pytest.raises(Exception):
        K.rows_pack(T(src, dev), None, n, seg_start, [o + 1 for o in seg_off], flat_g)      # misaligned segment
    with pytest.raises(Exception):
        K.rows_pack(T(src, dev), None, n, seg_start, [off] * W, flat_g)                     # past the buffer
