"""Row-sharded embedding tables over the GPUs of one node (no reference
counterpart -- the reference is single-process, single-device).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  A table of R rows is
dealt round-robin: global row r lives on rank r % W at local row r // W (balances popular items
better than contiguous blocks).

A BPR step moves the *sample* to the data where that is cheaper than moving the data to the sample:

    id lists (u, i, j)           -> every rank             12 B per sample, ONE fixed-size all-gather
    distinct item rows           <- the owners of Q rows   4k B per DISTINCT item a rank's samples touch
    item gradient rows           -> the owners of Q rows   4k B per (sample, item) contribution

User rows never travel: a sample is processed by the rank that owns P[u].  Because the ids are replicated, every
rank derives BOTH of its roles locally (`dr_route_views_i32`: what it processes and what it owns) -- which rows each
peer will request from it, in which order the gradient rows of each peer arrive, and where each of them stands in
the canonical order -- so no requests, counts or metadata are exchanged and a route needs one collective and one
host synchronisation (the variable split sizes of the two row exchanges).

The processing rank runs the lookup / dot / loss kernel on its own P shard and the fetched Q rows, updates its P
rows locally, and ships one gradient row per item occurrence.  Duplicate item gradients are NOT pre-summed on the
sender: the owner sums them in the order a single process sees for the concatenated global batch (i-part before
j-part, then source rank, then batch position), so the tables are bit-identical at every world size.

Everything that depends on the ids only (`route`) is separated from what depends on table values (`run`), so a
trainer routes batch t+1 on a second stream / second communicator while batch t runs
(`ShardedBpr.step(..., next_batch=...)`); the host synchronisation of a route then waits on short id kernels only.

`ops` is the kernel namespace (default: deeprec_amd.kernels = the HIP library).  The package ships no
other implementation; tests inject an oracle-backed namespace to exercise the exchange logic on CPU
with gloo.
"""
import torch
import torch.distributed as dist


def _a2a(out, inp, out_splits, in_splits, group, async_op=False):
    """The row exchange of this module (module-level so that a test can stage device tensors through the host
    when it runs several ranks on one GPU under gloo).  Returns the work handle when async_op, else None."""
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def _gather_ids(out, inp, group):
    """The id replication: out [W, 3, B] <- every rank's [3, B] (module-level for the same reason as _a2a)."""
    dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=group)


class ShardedTable:
    """The local shard of a [rows, k] table plus its optimizer slots."""

    def __init__(self, rows, k, world, rank, device, n_slots=0, slot_init=0.0):
        self.rows, self.k, self.world, self.rank = int(rows), int(k), world, rank
        self.local_rows = (self.rows - rank + world - 1) // world
        self.w = torch.zeros((self.local_rows, k), dtype=torch.float32, device=device)
        self.slots = [torch.full((self.local_rows, k), float(slot_init), dtype=torch.float32, device=device)
                      for _ in range(n_slots)]

    def slot(self, i):
        return self.slots[i] if i < len(self.slots) else None

    def load_global(self, full):
        """Fill the shard from a replicated [rows, k] tensor (tests / small tables)."""
        self.w.copy_(full[self.rank::self.world])

    def init_normal_(self, ops, seed, stddev, chunk_rows=1 << 20):
        """w[r, c] = stddev * normal(hash(seed, r*k + c)) for the global rows this rank owns, generated
        in chunks of global rows so the full table never exists on one device."""
        for lo in range(0, self.rows, chunk_rows * self.world):
            hi = min(self.rows, lo + chunk_rows * self.world)
            tmp = torch.empty((hi - lo, self.k), dtype=torch.float32, device=self.w.device)
            ops.init_normal_(tmp, seed, stddev, offset=lo * self.k)
            first = (self.rank - lo) % self.world
            part = tmp[first::self.world]
            l0 = (lo + first) // self.world
            self.w[l0:l0 + part.shape[0]].copy_(part)


_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class BprRoute:
    """Everything about one step that depends on the ids only (see ShardedBpr.route)."""
    __slots__ = ("n_s", "n_uq", "n_f", "n_g", "send_f", "recv_f", "send_g", "recv_g", "u_loc", "inv_i", "inv_j",
                 "order_g", "rows_fetch", "ids_canon", "perm", "plan_p", "plan_q", "ready", "keep")


class ShardedBpr:
    """BPR-MF training steps (bprmf.py:114-123) over row-sharded P and Q.

    route(u, i, j) -> BprRoute      ids only: 1 collective (id all-gather), 1 host sync, the owners' segment plans
    run(route, opt) -> loss         table values: 2 collectives (rows in, gradient rows out), no host sync
    step(opt, next_batch)           run the prefetched route, then route next_batch on the side stream

    Every rank passes a batch of the SAME length (the all-gather is fixed-size).
    group: communicator of the row / gradient exchanges; route_group: communicator of the id all-gather (created
    here when the step is pipelined on GPUs, because the two run concurrently on different streams)."""

    def __init__(self, ops, P, Q, group=None, route_group=None, pipelined=None):
        if P.world != Q.world or P.rank != Q.rank or P.k != Q.k:
            raise ValueError("P and Q must be sharded over the same ranks and share k")
        self.ops, self.P, self.Q = ops, P, Q
        self.world, self.rank, self.group = P.world, P.rank, group
        if self.world > 64:
            raise ValueError("at most 64 ranks")
        self.cuda = P.w.is_cuda
        self.pipelined = self.cuda if pipelined is None else bool(pipelined)
        self.collective = self.world > 1 or (dist.is_available() and dist.is_initialized())
        if route_group is None and self.pipelined and self.cuda and self.collective:
            route_group = dist.new_group()          # collective call: every rank constructs its trainer
        self.route_group = route_group if route_group is not None else group
        self.rpo = (Q.rows + self.world - 1) // self.world     # local rows per owner (upper bound)
        self.side = _side_stream(P.w.device) if (self.cuda and self.pipelined) else None
        self._pending = None
        self._retired = None

    # ---- exchanges -------------------------------------------------------------------------------------
    def _x(self, send, n_out, out_splits, in_splits, group, async_op=False):
        """All-to-all-v along dim 0.  async_op: returns (out, work); the caller waits before it reads `out`."""
        if not self.collective:
            return (send, None) if async_op else send              # one rank, no process group: nothing travels
        out = torch.empty((n_out,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        work = _a2a(out, send, out_splits, in_splits, group, async_op)
        return (out, work) if async_op else out

    # ---- ids only --------------------------------------------------------------------------------------
    def route(self, u, i, j):
        return self._route_finish(self._route_begin(u, i, j))

    def _route_begin(self, u, i, j):
        """The part of a route that needs no host synchronisation: the id all-gather and dr_route_views_i32."""
        ops, W = self.ops, self.world
        P, Q = self.P, self.Q
        B = u.numel()
        if i.numel() != B or j.numel() != B:
            raise ValueError("u, i, j must have the same length")
        if 4 * W * B >= (1 << 31) - 2:
            raise ValueError("batch of %d samples per rank on %d ranks exceeds the 31-bit list index" % (B, W))
        mine = torch.stack((u.reshape(-1), i.reshape(-1), j.reshape(-1))).to(torch.int32)
        if self.collective:
            ids = torch.empty((W, 3, B), dtype=torch.int32, device=mine.device)
            _gather_ids(ids, mine, self.route_group)               # the one collective of a route
        else:
            ids = mine.reshape(1, 3, B)
        return ids, ops.route_views(ids, self.rank, P.rows, Q.rows, self.rpo)

    def _route_finish(self, begun):
        """The host synchronisation of a route (split sizes) and the owners' segment plans."""
        ops, W = self.ops, self.world
        P, Q = self.P, self.Q
        ids, rv = begun
        c = rv.counts.tolist()                                     # the one host synchronisation of a route
        # the ids are replicated, so every rank sees the same count: a bad batch on ONE rank is refused by ALL ranks
        # here, before any of them enters a row exchange
        if c[4 * W + 2]:
            raise IndexError("sharded BPR step: %d samples of the global batch have a user / item id out of range"
                             % c[4 * W + 2])
        rt = BprRoute()
        rt.send_f, rt.send_g = c[0:2 * W:2], c[1:2 * W:2]          # view 0, per owner: distinct rows, contributions
        rt.recv_f, rt.recv_g = c[2 * W:4 * W:2], c[2 * W + 1:4 * W:2]    # view 1, per processing rank
        n0, n1 = c[4 * W], c[4 * W + 1]
        rt.n_s, rt.n_g = n0 // 2, n1
        rt.n_uq, rt.n_f = sum(rt.send_f), sum(rt.recv_f)
        n_s = rt.n_s
        rt.u_loc = rv.u_loc[:n_s]
        rt.inv_i, rt.inv_j = rv.inv[0, :n_s], rv.inv[0, n_s:2 * n_s]
        rt.order_g = rv.order[0, :2 * n_s]
        rt.rows_fetch = rv.uniq[1, :rt.n_f]
        rt.ids_canon, rt.perm = rv.rows[1, :n1], rv.perm[1, :n1]
        # the owners' segment plans (namespaces without SegmentPlan -- the CPU oracle in tests -- plan in the update)
        rt.plan_p = rt.plan_q = None
        if hasattr(ops, "SegmentPlan"):
            if n_s:
                rt.plan_p = ops.SegmentPlan(rt.u_loc, P.w.shape[0], k_max=P.k)
            if rt.n_g:
                rt.plan_q = ops.SegmentPlan(rt.ids_canon, Q.w.shape[0], k_max=Q.k)
        rt.ready = None
        rt.keep = (ids, rv)                     # alive until the step that uses the route is enqueued
        return rt

    # ---- table values ----------------------------------------------------------------------------------
    def run(self, rt, opt):
        ops, P, Q, g = self.ops, self.P, self.Q, self.group
        dev = P.w.device
        # distinct item rows <- owners (the owners run K1 on their shard)
        served = ops.gather(Q.w, rt.rows_fetch) if rt.n_f else torch.empty((0, Q.k), dtype=torch.float32, device=dev)
        qc = self._x(served, rt.n_uq, rt.send_f, rt.recv_f, g)
        if rt.n_s:
            # lookup + dots + loss on the local P shard and the fetched rows; stash = g * (q_i - q_j)
            gc, loss_b, stash = ops.bpr_forward(P.w, qc, rt.u_loc, rt.inv_i, rt.inv_j)
            q_send = ops.bpr_item_grads(P.w, rt.u_loc, gc, rt.order_g)         # reads pre-step P rows
        else:
            q_send = torch.empty((0, Q.k), dtype=torch.float32, device=dev)
        q_grad, travelling = self._x(q_send, rt.n_g, rt.recv_g, rt.send_g, g, async_op=True)
        # deterministic segment updates; contributions in ascending global batch position.  The user side is local
        # and runs while the item gradients travel (q_send was computed from the pre-step P rows above).
        if rt.n_s:
            ops.sparse_row_update(P.w, P.slot(0), P.slot(1), rt.u_loc, stash, opt, plan=rt.plan_p)
        if travelling is not None:
            travelling.wait()
        if rt.n_g:
            ops.sparse_row_update(Q.w, Q.slot(0), Q.slot(1), rt.ids_canon, q_grad, opt, src_idx=rt.perm, plan=rt.plan_q)
        if rt.n_s:
            return ops.sum_f32(loss_b)
        return torch.zeros(1, dtype=torch.float32, device=dev)

    # ---- one batch ahead -------------------------------------------------------------------------------
    def _begin_ahead(self, u, i, j):
        """Enqueue the sync-free part of a route (on the side stream when pipelined).  The side stream first waits for
        an event recorded HERE on the caller's current stream, so ids produced by earlier work on that stream are
        complete before the route reads them -- and nothing enqueued on it later delays the route."""
        if self.side is None:
            return self._route_begin(u, i, j)
        ready = torch.cuda.Event()
        ready.record()                              # on the caller's current stream (no Stream object built: ~8 us each)
        self.side.wait_event(ready)
        for t in (u, i, j):
            t.record_stream(self.side)              # the caller may drop the batch right after step()
        with torch.cuda.stream(self.side):
            return self._route_begin(u, i, j)

    def _finish_ahead(self, begun):
        if self.side is None:
            return self._route_finish(begun)
        with torch.cuda.stream(self.side):
            rt = self._route_finish(begun)
            rt.ready = torch.cuda.Event()
            rt.ready.record(self.side)
        return rt

    def prefetch(self, u, i, j):
        """Route a batch now (on the side stream when pipelined); step() consumes it."""
        self._pending = self._finish_ahead(self._begin_ahead(u, i, j))

    def step(self, opt, next_batch=None):
        """Run the prefetched batch; `next_batch` = (u, i, j) is routed meanwhile.  Order on the host: (1) the sync-free
        part of next_batch's route goes to the side stream, where it waits only for what the caller's stream held at
        that moment (the previous step); (2) this batch's kernels and exchanges are enqueued; (3) only then the host
        waits for the route's split sizes -- the id kernels have run beside (2) -- and enqueues the owners' plans.
        With (3) before (2), as in the first version, the GPU's main stream sat idle through every route's host wait."""
        rt = self._pending
        if rt is None:
            raise RuntimeError("ShardedBpr.step: no routed batch (call prefetch first)")
        self._pending = None
        begun = self._begin_ahead(*next_batch) if next_batch is not None else None
        if rt.ready is not None:
            rt.ready.wait()                         # the caller's current stream waits for the route's side-stream work
        loss = self.run(rt, opt)
        if begun is not None:
            self._pending = self._finish_ahead(begun)
        # side-stream memory of this route may be reused only by a route whose kernels wait for this run(): the next
        # route to be enqueued does (its event is recorded after this point), the one enqueued above does not -- so
        # the route stays referenced until the step after this one
        self._retired = rt
        return loss


def sharded_bpr_step(ops, P, Q, u, i, j, opt, group=None):
    """One BPR-MF step (bprmf.py:114-123) over row-sharded P, Q for this rank's batch (u, i, j) of GLOBAL ids
    (int32 [B]), unpipelined.  Returns the loss sum of the samples this rank PROCESSED (those whose user row it
    owns; all-reduce it for the global loss) as a device scalar tensor."""
    tr = ShardedBpr(ops, P, Q, group=group, route_group=group, pipelined=False)
    return tr.run(tr.route(u, i, j), opt)


def shard_users(n_users, world, rank):
    """Evaluation users are split data-parallel in contiguous blocks."""
    per = (n_users + world - 1) // world
    return rank * per, min(n_users, (rank + 1) * per)


def all_gather_table(T, group=None):
    """Replicate a sharded table (item table before evaluation: one all-gather)."""
    if T.world == 1:
        return T.w.clone()
    per = (T.rows + T.world - 1) // T.world
    pad = torch.zeros((per, T.k), dtype=torch.float32, device=T.w.device)
    pad[:T.local_rows].copy_(T.w)
    parts = [torch.empty_like(pad) for _ in range(T.world)]
    dist.all_gather(parts, pad, group=group)
    full = torch.stack(parts, 1).reshape(per * T.world, T.k)   # row r = parts[r % W][r // W]
    return full[:T.rows].contiguous()


class RowRoute:
    """Routing of one id list (GLOBAL row numbers, int32 [n]) to the ranks that own the rows and back: a stable
    bucket by owner, one count exchange + host synchronisation, the local row numbers to the owners (`want`).
    fetch() brings rows back in the order of the id list; push() takes per-id rows (gradients) to the owners, where
    they arrive source rank by source rank, each source's in the order of its id list -- i.e. in the order a single
    process sees for the concatenation of every rank's list."""

    def __init__(self, ops, ids, world, group=None):
        self.world, self.group, self.n = world, group, ids.numel()
        order, cnt = ops.bucket_order((ids % world).to(torch.int32), world)
        recv = torch.empty_like(cnt)
        _a2a(recv, cnt, None, None, group)
        self.sc, self.rc = torch.stack((cnt, recv)).tolist()             # host sync: variable split sizes
        self.order = order.long()
        local = torch.div(ids, world, rounding_mode="floor").to(torch.int32)[self.order].contiguous()
        self.want = torch.empty(int(sum(self.rc)), dtype=torch.int32, device=ids.device)
        _a2a(self.want, local, self.rc, self.sc, group)

    def fetch(self, served):
        """served = owner-side rows for `want` [n_recv, ...] -> rows of the id list [n, ...]."""
        back = torch.empty((self.n,) + tuple(served.shape[1:]), dtype=served.dtype, device=served.device)
        _a2a(back, served.contiguous(), self.sc, self.rc, self.group)
        out = torch.empty_like(back)
        out[self.order] = back                                           # undo the bucketing
        return out

    def push(self, rows):
        """rows per id [n, ...] -> owner-side rows [n_recv, ...] aligned with `want`."""
        out = torch.empty((self.want.numel(),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        _a2a(out, rows[self.order].contiguous(), self.rc, self.sc, self.group)
        return out


def fetch_rows(ops, T, ids, group=None):
    """Rows `ids` (GLOBAL row numbers, int32 [n]) of a ShardedTable, in the order given: ids -> owners, the owners
    run the gather kernel on their shard, rows come back (one count exchange + host sync, two all-to-alls)."""
    if T.world == 1:
        return ops.gather(T.w, ids)
    rt = RowRoute(ops, ids, T.world, group)
    served = ops.gather(T.w, rt.want) if rt.want.numel() else torch.empty((0, T.k), dtype=torch.float32, device=ids.device)
    return rt.fetch(served)


def sharded_eval_topk(ops, P, Q, users, K, train=None, targets=None, Q_full=None, group=None):
    """Full-catalog top-K evaluation (RankingMetrics.py:82-101) with the USERS sharded data-parallel:
    `users` = this rank's evaluation users (global ids; split them with shard_users), train / targets = their CSR
    lists.  The item table is replicated with one all-gather (pass Q_full to reuse it across calls), the user rows
    come from their owners, the MFMA scoring + top-K kernel runs locally.  Returns what ops.score_topk returns plus
    the replicated item table."""
    if Q_full is None:
        Q_full = all_gather_table(Q, group)
    U = fetch_rows(ops, P, users, group)
    idx, val, rank = ops.score_topk(U, Q_full, K, train=train, targets=targets)
    return idx, val, rank, Q_full


def reduce_rank_metrics(ranks, K, group=None):
    """Leave-one-out metrics from this rank's target ranks (one held-out item per user; -1 = not a candidate),
    summed over all ranks: (users, hit-rate@K, NDCG@K, MRR).  One 4-float all-reduce."""
    r = ranks.to(torch.float64)
    ok = ranks >= 0
    hit = ok & (ranks < K)
    sums = torch.stack((ok.sum().to(torch.float64), hit.sum().to(torch.float64),
                        torch.where(hit, 1.0 / torch.log2(r + 2.0), torch.zeros_like(r)).sum(),
                        torch.where(ok, 1.0 / (r + 1.0), torch.zeros_like(r)).sum()))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(sums, group=group)
    n = max(float(sums[0]), 1.0)
    return int(sums[0]), float(sums[1]) / n, float(sums[2]) / n, float(sums[3]) / n


# ----------------------------------------------------------------------------------------------
# sharded checkpoints (the reference's save() is dead code, gmf.py:156-158, bprmf.py:164-166)
# ----------------------------------------------------------------------------------------------
def _shard_path(prefix, rank, world):
    return "%s.rank%05d-of-%05d.safetensors" % (prefix, rank, world)


def save_sharded(prefix, tables):
    """Write this rank's shard of every table in `tables` (name -> ShardedTable) with its optimizer slots to
    `<prefix>.rank<r>-of-<W>.safetensors`.  Every rank calls it; no communication."""
    from safetensors.torch import save_file
    any_t = next(iter(tables.values()))
    out, meta = {}, {"world": str(any_t.world), "rank": str(any_t.rank)}
    for name, T in tables.items():
        out[name + ".w"] = T.w.detach().cpu().contiguous()
        for j, sl in enumerate(T.slots):
            out["%s.slot%d" % (name, j)] = sl.detach().cpu().contiguous()
        meta[name + ".rows"] = str(T.rows)
        meta[name + ".k"] = str(T.k)
        meta[name + ".slots"] = str(len(T.slots))
    path = _shard_path(prefix, any_t.rank, any_t.world)
    save_file(out, path, metadata=meta)
    return path


def load_sharded(prefix, names, world, rank, device, saved_world=None):
    """Rebuild this rank's ShardedTables from a checkpoint written at ANY world size (`saved_world`; probed
    from the file names when None): global row r lived on saved rank r % Ws at local row r // Ws and now lives on
    rank r % world at local row r // world.  Only safetensors is read (no code is executed from the files)."""
    import glob
    import re
    from safetensors import safe_open
    if saved_world is None:
        found = sorted(glob.glob("%s.rank*-of-*.safetensors" % glob.escape(prefix)))
        if not found:
            raise FileNotFoundError("no checkpoint shards match %s.rank*-of-*.safetensors" % prefix)
        saved_world = int(re.search(r"-of-(\d+)\.safetensors$", found[0]).group(1))
    tables = {}
    for o in range(saved_world):
        with safe_open(_shard_path(prefix, o, saved_world), framework="pt", device="cpu") as f:
            meta = f.metadata()
            for name in names:
                rows, k, n_slots = int(meta[name + ".rows"]), int(meta[name + ".k"]), int(meta[name + ".slots"])
                if name not in tables:
                    tables[name] = ShardedTable(rows, k, world, rank, device, n_slots)
                T = tables[name]
                local_o = (rows - o + saved_world - 1) // saved_world
                g = o + saved_world * torch.arange(local_o)             # global rows held by saved rank o
                mine = (g % world) == rank
                if not bool(mine.any()):
                    continue
                dst = torch.div(g[mine], world, rounding_mode="floor").to(T.w.device)
                T.w[dst] = f.get_tensor(name + ".w")[mine].to(T.w.device)
                for j in range(n_slots):
                    T.slots[j][dst] = f.get_tensor("%s.slot%d" % (name, j))[mine].to(T.w.device)
    return tables
