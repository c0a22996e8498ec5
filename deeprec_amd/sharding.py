"""Row-sharded embedding tables over the GPUs of one node (no reference
counterpart -- the reference is single-process, single-device).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  A table of R rows is
dealt round-robin: global row r lives on rank r % W at local row r // W (balances popular items
better than contiguous blocks).  A training step exchanges, per table,

    ids   -> owners        all-to-all-v   (4 B per lookup)
    rows  <- owners        all-to-all-v   (4k B per lookup, owners run the gather kernel)
    grads -> owners        all-to-all-v   (4k B per lookup)

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


def sharded_bpr_step(ops, P, Q, u, i, j, opt, group=None):
    """One BPR-MF step (bprmf.py:114-123) over row-sharded P, Q for this rank's batch (u, i, j) of
    GLOBAL ids (int32 [B]).  Returns this rank's loss sum (device scalar tensor)."""
    world, rank = P.world, P.rank
    dev = u.device
    B = u.numel()
    ex_u = Exchange(u, world, rank, group)
    ex_i = Exchange(i, world, rank, group)
    ex_j = Exchange(j, world, rank, group)
    # rows <- owners (the owners run K1 on their shard)
    pu = ex_u.fetch(ops.gather(P.w, ex_u.owner_rows))
    qi = ex_i.fetch(ops.gather(Q.w, ex_i.owner_rows))
    qj = ex_j.fetch(ops.gather(Q.w, ex_j.owner_rows))
    # local interaction + loss on compact tables: row b of pu / [qi ; qj]
    ar = torch.arange(B, dtype=torch.int32, device=dev)
    qc = torch.cat((qi, qj))
    g, loss_b, stash = ops.bpr_forward(pu, qc, ar, ar, ar + B)
    # grads -> owners, then the owners' deterministic segment update.  Q first is not needed here
    # (pu is a private pre-step copy), but the contribution ORDER is: [all i ; all j], rank-major.
    gi = ops.scale_rows(pu, g)
    gj = ops.scale_rows(pu, -g)
    q_src = torch.cat((ex_i.push(gi), ex_j.push(gj)))
    q_ids = torch.cat((ex_i.owner_rows, ex_j.owner_rows))
    p_src = ex_u.push(stash)
    ops.sparse_row_update(Q.w, Q.slot(0), Q.slot(1), q_ids, q_src, opt)
    ops.sparse_row_update(P.w, P.slot(0), P.slot(1), ex_u.owner_rows, p_src, opt)
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
