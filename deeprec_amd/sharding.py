"""Row-sharded embedding tables over the GPUs of one node (no reference
counterpart -- the reference is single-process, single-device).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  A table of R rows is
dealt round-robin: global row r lives on rank r % W at local row r // W (balances popular items
better than contiguous blocks).  A training step exchanges, per table,

    ids   -> owners        all-to-all-v   (4 B per lookup)
    rows  <- owners        all-to-all-v   (4k B per lookup, owners run the gather kernel)
    grads -> owners        all-to-all-v   (4k B per lookup)

(the u list in one exchange, the i and j lists together in a second one: StepRoutes)

and the owners run the deterministic sorted-segment row update on what they received.  The received
contributions are laid out in ascending *global batch position* (rank-major, then batch order; for Q
all positive-item contributions precede all negative-item ones), i.e. exactly the order a single
process sees for the concatenated global batch, so tables are bit-identical at every world size.

`ops` is the kernel namespace (default: deeprec_amd.kernels = the HIP library).  The package ships no
other implementation; tests inject an oracle-backed namespace to exercise the exchange logic on CPU
with gloo.
"""
import torch
import torch.distributed as dist


def _a2a(out, inp, out_splits, in_splits, group):
    dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)
    return out


class Exchange:
    """Routing of one id list to the owning ranks and back."""

    def __init__(self, ids, world, rank, group=None):
        self.world, self.rank, self.group = world, rank, group
        ids = ids.reshape(-1)
        self.n = ids.numel()
        owner = (ids % world).to(torch.int64)
        # stable bucket by owner keeps batch order inside each destination
        self.order = torch.sort(owner, stable=True)[1]
        send_counts = torch.bincount(owner, minlength=world)
        recv_counts = torch.empty_like(send_counts)
        if world > 1:
            dist.all_to_all_single(recv_counts, send_counts, group=group)
        else:
            recv_counts.copy_(send_counts)
        self.send_splits = send_counts.tolist()          # host sync: sizes of the variable all-to-all
        self.recv_splits = recv_counts.tolist()
        self.n_recv = int(sum(self.recv_splits))
        local_rows = torch.div(ids, world, rounding_mode="floor").to(torch.int32)
        send_rows = local_rows[self.order].contiguous()
        self.owner_rows = torch.empty(self.n_recv, dtype=torch.int32, device=ids.device)
        if world > 1:
            _a2a(self.owner_rows, send_rows, self.recv_splits, self.send_splits, group)
        else:
            self.owner_rows.copy_(send_rows)
        self.inv = torch.empty_like(self.order)
        self.inv[self.order] = torch.arange(self.n, device=ids.device)

    def fetch(self, rows_at_owner):
        """Owner-side rows [n_recv, k] -> requester-side rows [n, k] in original id order."""
        k = rows_at_owner.shape[1]
        back = torch.empty((self.n, k), dtype=rows_at_owner.dtype, device=rows_at_owner.device)
        if self.world > 1:
            _a2a(back, rows_at_owner.contiguous(), self.send_splits, self.recv_splits, self.group)
        else:
            back.copy_(rows_at_owner)
        return back[self.inv].contiguous()

    def push(self, rows):
        """Requester-side rows [n, k] (original id order) -> owner-side rows [n_recv, k], rank-major."""
        k = rows.shape[1]
        out = torch.empty((self.n_recv, k), dtype=rows.dtype, device=rows.device)
        send = rows[self.order].contiguous()
        if self.world > 1:
            _a2a(out, send, self.recv_splits, self.send_splits, self.group)
        else:
            out.copy_(send)
        return out


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


_HELPER = {}


def _helper_stream(device):
    key = (device.type, device.index)
    if key not in _HELPER:
        _HELPER[key] = torch.cuda.Stream(device=device)
    return _HELPER[key]


class StepRoutes:
    """Routing of one BPR batch: u -> owners of P rows, (i, j) -> owners of Q rows.

    One count exchange and ONE host synchronisation serve both tables.  Wire order towards owner r is, per
    table, the requester's batch order (stable bucket by owner); for Q each requester sends [its i's | its j's].
    The owner-side canonical contribution order -- all positive items of the global batch (rank-major, batch
    order), then all negative items -- is the index list `q_perm` into what was received, so no payload is
    ever re-ordered in memory: kernels take it as `src_idx`."""

    def __init__(self, ops, u, i, j, world, rank, group=None):
        self.world, self.rank, self.group = world, rank, group
        dev = u.device
        B = u.numel()
        self.B = B
        W = world
        ij = torch.cat((i, j))
        oq = (ij % W) * 2
        oq[B:] += 1                                              # key = owner * 2 + part (0: i, 1: j)
        self.order_u, cu = ops.bucket_order((u % W).to(torch.int32), W)
        self.order_q, cq = ops.bucket_order(oq.to(torch.int32), 2 * W)
        self.order_u, self.order_q = self.order_u.long(), self.order_q.long()
        counts = torch.cat((cu[:, None], cq.view(W, 2)), 1)
        recv = torch.empty_like(counts)
        if W > 1:
            dist.all_to_all_single(recv, counts, group=group)    # row s of recv = what rank s sends here
        else:
            recv.copy_(counts)
        both = torch.stack((counts, recv)).tolist()              # the step's one host sync (variable split sizes)
        sc, rc = both
        self.send_u = [c[0] for c in sc]
        self.recv_u = [c[0] for c in rc]
        self.send_q = [c[1] + c[2] for c in sc]
        self.recv_q = [c[1] + c[2] for c in rc]
        self.n_recv_u, self.n_recv_q = sum(self.recv_u), sum(self.recv_q)
        # local row ids at the owners
        self.rows_u = self._a2a(torch.div(u, W, rounding_mode="floor").to(torch.int32)[self.order_u].contiguous(),
                                self.n_recv_u, self.recv_u, self.send_u)
        self.rows_q = self._a2a(torch.div(ij, W, rounding_mode="floor").to(torch.int32)[self.order_q].contiguous(),
                                self.n_recv_q, self.recv_q, self.send_q)
        # canonical owner-side order of the Q contributions: every source's i-part, then every source's j-part
        off, ip, jp = 0, [], []
        for s in range(W):
            ci, cj = rc[s][1], rc[s][2]
            ip.append((off, off + ci))
            jp.append((off + ci, off + ci + cj))
            off += ci + cj
        self.q_perm = torch.cat([torch.arange(lo, hi, dtype=torch.int32, device=dev) for lo, hi in ip + jp]) \
            if off else torch.zeros(0, dtype=torch.int32, device=dev)
        # where each sample's rows sit in the fetched (owner-sorted) buffers
        ar = torch.arange(B, dtype=torch.int32, device=dev)
        self.inv_u = torch.empty(B, dtype=torch.int32, device=dev)
        self.inv_u[self.order_u] = ar
        self.inv_q = torch.empty(2 * B, dtype=torch.int32, device=dev)
        self.inv_q[self.order_q] = torch.arange(2 * B, dtype=torch.int32, device=dev)

    def _a2a(self, send, n_out, out_splits, in_splits):
        out = torch.empty((n_out,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if self.world > 1:
            dist.all_to_all_single(out, send, out_splits, in_splits, group=self.group)
        else:
            out.copy_(send)
        return out

    def fetch_u(self, rows_at_owner):   # [n_recv_u, k] -> [B, k], requester's owner-sorted order
        return self._a2a(rows_at_owner, self.B, self.send_u, self.recv_u)

    def fetch_q(self, rows_at_owner):   # [n_recv_q, k] -> [2B, k]
        return self._a2a(rows_at_owner, 2 * self.B, self.send_q, self.recv_q)

    def push_u(self, rows_sorted):      # [B, k] in owner-sorted order -> [n_recv_u, k], rank-major
        return self._a2a(rows_sorted, self.n_recv_u, self.recv_u, self.send_u)

    def push_q(self, rows_sorted):      # [2B, k] -> [n_recv_q, k], wire order (see q_perm)
        return self._a2a(rows_sorted, self.n_recv_q, self.recv_q, self.send_q)


def sharded_bpr_step(ops, P, Q, u, i, j, opt, group=None):
    """One BPR-MF step (bprmf.py:114-123) over row-sharded P, Q for this rank's batch (u, i, j) of
    GLOBAL ids (int32 [B]).  Returns this rank's loss sum (device scalar tensor).

    7 collectives (1 count, 2 id, 2 row, 2 gradient all-to-alls) and one host synchronisation per step; no
    payload row is copied except by the kernels that produce or consume it (lookups and gradient rows are
    addressed through index lists)."""
    world, rank = P.world, P.rank
    B = u.numel()
    rt = StepRoutes(ops, u, i, j, world, rank, group)
    q_ids = rt.rows_q[rt.q_perm.long()].contiguous()
    # The owners' segment plans depend on the routed ids only: build them on a helper stream while the rows and
    # gradients travel.  (Namespaces without SegmentPlan -- the CPU oracle in tests -- plan inside the update.)
    plan_q = plan_p = side = None
    if world > 1 and hasattr(ops, "SegmentPlan") and u.is_cuda:   # one rank: nothing travels, nothing to hide behind
        main = torch.cuda.current_stream()
        side = _helper_stream(u.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            plan_q = ops.SegmentPlan(q_ids, Q.w.shape[0], k_max=Q.k)
            plan_p = ops.SegmentPlan(rt.rows_u, P.w.shape[0], k_max=P.k)
    # rows <- owners (the owners run K1 on their shard); they arrive in owner-sorted order and stay there
    pu = rt.fetch_u(ops.gather(P.w, rt.rows_u))
    qc = rt.fetch_q(ops.gather(Q.w, rt.rows_q))
    # local interaction + loss: sample b reads pu[inv_u[b]], qc[inv_q[b]], qc[inv_q[B + b]]
    g, loss_b, stash = ops.bpr_forward(pu, qc, rt.inv_u, rt.inv_q[:B].contiguous(), rt.inv_q[B:].contiguous())
    # grads -> owners, produced directly in wire order.  Q: +g * p_u for the i-entry, -g * p_u for the j-entry
    # of each sample (pu is a private pre-step copy, so the Q-before-P ordering of the fused step is moot here)
    coef = torch.cat((g, -g))[rt.order_q].contiguous()
    src_row = rt.inv_u[(rt.order_q % B)].contiguous()
    q_grad = rt.push_q(ops.scale_rows(pu, coef, src_row))
    p_grad = rt.push_u(ops.gather(stash, rt.order_u.to(torch.int32)))
    # the owners' deterministic segment updates; contributions in ascending global batch position
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
    ops.sparse_row_update(Q.w, Q.slot(0), Q.slot(1), q_ids, q_grad, opt, src_idx=rt.q_perm, plan=plan_q)
    ops.sparse_row_update(P.w, P.slot(0), P.slot(1), rt.rows_u, p_grad, opt, plan=plan_p)
    return ops.sum_f32(loss_b)


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
