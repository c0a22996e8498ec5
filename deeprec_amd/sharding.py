"""Row-sharded embedding tables over the GPUs of one node (no reference
counterpart -- the reference is single-process, single-device).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  A table of R rows is
dealt round-robin: global row r lives on rank r % W at local row r // W (balances popular items
better than contiguous blocks).

A BPR step (bprmf.py:114-123) moves the *sample* to the data where that is cheaper than moving the data to the sample:

    id lists (u, i, j)           -> every rank             12 B per sample, ONE fixed-size all-gather
    distinct item rows           <- the owners of Q rows   4k B per DISTINCT remote item a rank's samples touch
    item gradient rows           -> the owners of Q rows   4k B per (sample, remote item) contribution

User rows never travel: a sample is processed by the rank that owns P[u].  Because the ids are replicated, every
rank derives BOTH of its roles locally (dr_shard_bpr_route: what it processes and what it owns) -- which rows each
peer will request from it, in which order the gradient rows of each peer arrive, and where each of them stands in
the canonical order -- so no requests, counts or metadata are exchanged and a route needs one collective and one
host synchronisation (the variable split sizes of the two row exchanges).

Round 3: the processing rank runs the FUSED step (csrc/shard.hip).  Its combined key list [user rows of its samples ;
item rows it owns, in canonical order] is planned exactly like the single-GPU step's -- at W = 1 it IS that list and
the step is the fused step: no exchange, no copy.  Rows with one local contribution are updated by the forward
kernel; item rows of its samples that live elsewhere are read from the fetch buffer and their gradient rows are
written straight into the send buffer; one tail pass applies every other row of both tables, taking contributions
that arrived from peers from the receive buffer.  Duplicate item gradients are NOT pre-summed on the sender: the
owner sums them in the order a single process sees for the concatenated global batch (i-part before j-part, then
source rank, then batch position), so the tables are bit-identical at every world size.

Everything that depends on the ids only (`route`) is separated from what depends on table values (`run`), so a
trainer routes batch t+1 on a second stream / second communicator while batch t runs
(`ShardedBpr.step(..., next_batch=...)`); the host synchronisation of a route then waits on short id kernels only.
Every per-step buffer (three route workspaces, id lists, exchange buffers, pinned split sizes) is allocated once.

Models built from lookups under autograd (AttRec, Caser) are data-parallel over their batch with the same row-sharded
tables: `StepRoute` makes ONE route for all lookups of a step (id all-gather; route keys + two bucket orders; one host
synchronisation; one all-to-all with the rows of every (table, list) pair, one with every gradient row going back),
`RowRoute` / `fetch_rows` remain for single ad-hoc lookups (evaluation).

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
    """The id replication: out [W, ...] <- every rank's [...] (module-level for the same reason as _a2a)."""
    dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=group)


def _all_reduce(t, group):
    """Sum of the data-parallel dense gradients (module-level for the same reason as _a2a)."""
    dist.all_reduce(t, group=group)


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


def _side_streams(device, n):
    return [torch.cuda.Stream(device=device) for _ in range(n)]


class BprRoute:
    """One routed batch: its workspace slot and, after finish(), the split sizes of the two row exchanges."""
    __slots__ = ("slot", "ws", "n_fetch", "n_serve", "n_send", "n_recv", "send_f", "recv_f", "send_g", "recv_g", "ready")


class ShardedBpr:
    """BPR-MF training steps (bprmf.py:114-123) over row-sharded P and Q.

    route(u, i, j) -> BprRoute      ids only: 1 collective (id all-gather), 1 host sync, the rank's plan
    run(route, opt) -> loss         table values: 2 collectives (rows in, gradient rows out), no host sync
    step(opt, next_batch)           run the prefetched route, then route next_batch on the side stream

    Every rank passes a batch of the SAME length (the all-gather is fixed-size), the same on every call.
    group: communicator of the row / gradient exchanges; route_group: communicator of the id all-gather (created
    here when the step is pipelined on GPUs, because the two run concurrently on different streams).
    `sides` (the streams the routes alternate on) may be replaced between steps after a device synchronisation: which
    streams share a hardware queue is decided by the HIP runtime, and the step time depends on it (bench.py picks them
    by trial).
    caps = (cap_s, cap_g): the most samples a rank may process / item entries it may own per step (default
    ops.shard_caps); a batch beyond them is refused by every rank together."""

    SLOTS = 3       # route workspaces: the batch that runs + two routed ahead
    N_SIDE = 2      # side streams the routes alternate on

    def __init__(self, ops, P, Q, group=None, route_group=None, pipelined=None, caps=None):
        if P.world != Q.world or P.rank != Q.rank or P.k != Q.k:
            raise ValueError("P and Q must be sharded over the same ranks and share k")
        self.ops, self.P, self.Q = ops, P, Q
        self.world, self.rank, self.group = P.world, P.rank, group
        if self.world > 64:
            raise ValueError("at most 64 ranks")
        self.cuda = P.w.is_cuda
        self.pipelined = self.cuda if pipelined is None else bool(pipelined)
        self.collective = self.world > 1 or (dist.is_available() and dist.is_initialized())
        if route_group is None and self.pipelined and self.cuda and self.collective and self.world > 1:
            route_group = dist.new_group()          # collective call: every rank constructs its trainer
        self.route_group = route_group if route_group is not None else group
        # two side streams: the routes of batches t+1 and t+2 run beside each other and beside step t (a route is a chain
        # of short launches that takes about two step times next to the bandwidth-saturating forward kernel)
        self.sides = _side_streams(P.w.device, self.N_SIDE) if (self.cuda and self.pipelined) else None
        self._routes = 0
        self.caps = caps
        self.B = None
        self._pending = []          # routed batches, oldest first (at most SLOTS - 1)
        self._unchecked = None
        self._next_slot = 0

    # ---- buffers, once -----------------------------------------------------------------------------------
    def _allocate(self, B):
        ops, P, Q, W = self.ops, self.P, self.Q, self.world
        dev = P.w.device
        cs, cg = self.caps if self.caps is not None else ops.shard_caps(W, B)
        self.B = B
        n = self.SLOTS
        self.ws = [ops.ShardWorkspace(W, self.rank, B, P.k, P.rows, Q.rows, dev, cs, cg) for _ in range(n)]
        if W > 1:
            self.ids = [torch.empty((W, 3, B), dtype=torch.int32, device=dev) for _ in range(n)]
            self.mine = [torch.empty((3, B), dtype=torch.int32, device=dev) for _ in range(n)]
        self.counts_host = [torch.zeros(8 + 6 * W, dtype=torch.int32, pin_memory=self.cuda) for _ in range(n)]
        if self.cuda:
            self.ev_routed = [torch.cuda.Event() for _ in range(n)]     # the route's kernels and its counts copy are done
            self.ev_free = [torch.cuda.Event() for _ in range(n)]       # the caller's stream no longer reads the slot
        if W > 1:
            f32 = dict(dtype=torch.float32, device=dev)
            self.serve_buf = torch.empty((cg, P.k), **f32)        # rows the peers fetch from this rank
            self.fetch_buf = torch.empty((2 * cs, P.k), **f32)    # rows this rank fetches
            self.send_buf = torch.empty((2 * cs, P.k), **f32)     # gradient rows out
            self.recv_buf = torch.empty((cg, P.k), **f32)         # gradient rows in
        self._checked = False

    # ---- ids only --------------------------------------------------------------------------------------
    def route(self, u, i, j):
        return self._route_finish(self._route_begin(u, i, j))

    def _route_begin(self, u, i, j, stream=None):
        """The part of a route that needs no host synchronisation: the id all-gather and dr_shard_bpr_route, which ends
        with the asynchronous copy of its counts to pinned host memory.  stream: raw handle to enqueue on (world 1 only;
        default and world > 1: torch's current stream)."""
        ops, W = self.ops, self.world
        B = u.numel()
        if i.numel() != B or j.numel() != B or B < 1:
            raise ValueError("u, i, j must have the same (non-zero) length")
        if self.B is None:
            if 4 * W * B >= (1 << 31) - 8192:
                raise ValueError("batch of %d samples per rank on %d ranks exceeds the 31-bit list index" % (B, W))
            self._allocate(B)
        if B != self.B:
            raise ValueError("ShardedBpr was sized for batches of %d samples per rank, got %d" % (self.B, B))
        rt = BprRoute()
        rt.slot = self._next_slot
        self._next_slot = (self._next_slot + 1) % self.SLOTS
        rt.ws = self.ws[rt.slot]
        rt.ready = None
        if W > 1:
            ids, mine = self.ids[rt.slot], self.mine[rt.slot]
            torch.stack((u.reshape(-1), i.reshape(-1), j.reshape(-1)), out=mine)
            _gather_ids(ids, mine, self.route_group)               # the one collective of a route
            ops.shard_route(rt.ws, ids, counts_host=self.counts_host[rt.slot])
        else:
            # one rank: the batch's own three lists are the global batch (no stacked copy)
            ops.shard_route(rt.ws, uij=(u, i, j), counts_host=self.counts_host[rt.slot], stream=stream)
        return rt

    def _route_finish(self, rt):
        """The host synchronisation of a route: its counts (split sizes of the two row exchanges, refused batches)."""
        W = self.world
        if self.cuda:
            if rt.ready is None:                                   # routed on the caller's stream
                rt.ready = self.ev_routed[rt.slot]
                rt.ready.record()
            rt.ready.synchronize()                                 # the one host synchronisation of a route
        c = self.counts_host[rt.slot].tolist()
        # the ids are replicated, so every rank sees the same numbers: a bad batch on ONE rank is refused by ALL ranks
        # here, before any of them enters a row exchange
        if c[4]:
            raise IndexError("sharded BPR step: %d samples of the global batch have a user / item id out of range" % c[4])
        if c[5]:
            raise RuntimeError("sharded BPR step: a rank would process more than %d samples or own more than %d item "
                               "entries of this batch (per-rank loads: %s samples, %s entries); pass larger caps"
                               % (rt.ws.cap_s, rt.ws.cap_g, c[8 + 4 * W:8 + 5 * W], c[8 + 5 * W:8 + 6 * W]))
        if W > 1:
            rt.send_f, rt.send_g = c[8:8 + W], c[8 + W:8 + 2 * W]  # per owner: distinct rows to fetch, gradient rows out
            rt.recv_f, rt.recv_g = c[8 + 2 * W:8 + 3 * W], c[8 + 3 * W:8 + 4 * W]   # per processing rank
            rt.n_fetch, rt.n_send = sum(rt.send_f), sum(rt.send_g)
            rt.n_serve, rt.n_recv = sum(rt.recv_f), sum(rt.recv_g)
        return rt

    # ---- table values ----------------------------------------------------------------------------------
    def run(self, rt, opt):
        ops, P, Q, g = self.ops, self.P, self.Q, self.group
        ws = rt.ws
        sP, sQ = (P.slot(0), P.slot(1)), (Q.slot(0), Q.slot(1))
        ok = self._checked                          # tensors validated on the first step; they never change
        self._checked = True
        if self.world == 1:
            # the fused single-GPU step: nothing to fetch, nothing to send
            ops.shard_forward(ws, P.w, Q.w, sP, sQ, opt, checked=ok)
            return ops.shard_finish(ws, P.w, Q.w, sP, sQ, opt, checked=ok)
        # distinct remote item rows <- owners (the owners run K1 on their shard, before any of this step's updates)
        served = self.serve_buf[:rt.n_serve]
        ops.gather_into(Q.w, ws.serve[:rt.n_serve], served)
        _a2a(self.fetch_buf[:rt.n_fetch], served, rt.send_f, rt.recv_f, g)
        # lookups, dots, loss; rows with one local contribution updated in place; gradient rows of remote items -> send_buf
        ops.shard_forward(ws, P.w, Q.w, sP, sQ, opt, self.fetch_buf, self.send_buf, checked=ok)
        _a2a(self.recv_buf[:rt.n_recv], self.send_buf[:rt.n_send], rt.recv_g, rt.send_g, g)
        # every other row of both tables; contributions in ascending global batch position, local or received
        return ops.shard_finish(ws, P.w, Q.w, sP, sQ, opt, self.recv_buf, checked=ok)

    # ---- batches ahead ---------------------------------------------------------------------------------
    def _begin_ahead(self, u, i, j):
        """Enqueue the sync-free part of a route (on the side stream when pipelined).  The side stream first waits for
        an event recorded HERE on the caller's current stream, so ids produced by earlier work on that stream are
        complete before the route reads them, and the workspace slot it takes has been read by its last step -- and
        nothing enqueued on the caller's stream later delays the route."""
        if self.sides is None:
            return self._route_begin(u, i, j)
        side = self.sides[self._routes % len(self.sides)]
        self._routes += 1
        slot = self._next_slot if self.B is not None else 0
        if self.B is None:
            self._allocate(u.numel())
        free = self.ev_free[slot]
        free.record()                               # on the caller's current stream
        side.wait_event(free)
        for t in (u, i, j):
            t.record_stream(side)                   # the caller may drop the batch right after step()
        if self.world == 1:
            rt = self._route_begin(u, i, j, stream=side.cuda_stream)         # no stream context: ~30 us of host time
        else:
            with torch.cuda.stream(side):
                rt = self._route_begin(u, i, j)
        rt.ready = self.ev_routed[rt.slot]
        rt.ready.record(side)
        return rt

    def prefetch(self, u, i, j):
        """Route a batch now (on the side stream when pipelined); step() consumes the routed batches in order.  Up to
        SLOTS - 1 batches may be routed ahead."""
        if len(self._pending) >= self.SLOTS - 1:
            raise RuntimeError("ShardedBpr.prefetch: %d batches are routed ahead already" % len(self._pending))
        self._pending.append(self._begin_ahead(u, i, j))

    def step(self, opt, next_batch=None):
        """Run the oldest routed batch; `next_batch` = (u, i, j) is routed meanwhile (behind the ones already waiting).
        Order on the host: (1) at world > 1 the host reads the counts of the batch to run (split sizes of its exchanges)
        -- routed one or two steps ago; (2) the sync-free route of next_batch goes to the side stream, where it waits only
        for what the caller's stream held at that moment (the previous step: the last reader of the workspace slot it
        takes); (3) this batch's kernels and exchanges are enqueued."""
        if not self._pending:
            raise RuntimeError("ShardedBpr.step: no routed batch (call prefetch first)")
        if self._unchecked is not None:             # world 1: the counts of the batch that ran in the previous call
            done, self._unchecked = self._unchecked, None
            self._route_finish(done)
        rt = self._pending.pop(0)
        if self.world > 1 or not self.cuda:
            self._route_finish(rt)                  # the split sizes of the exchanges: the host has to see them
        if next_batch is not None:
            self._pending.append(self._begin_ahead(*next_batch))
        if rt.ready is not None:
            rt.ready.wait()                         # the caller's current stream waits for the route's side-stream work
        loss = self.run(rt, opt)
        if self.world == 1 and self.cuda:
            # One rank exchanges nothing, so the host needs no number from the route before it enqueues the step: the
            # step waits for the route on the device, and the counts are read at the next call (flush() reads them now).
            # A batch with ids out of range is therefore reported one call late at world 1; its valid samples have been
            # applied, the invalid ones contributed nothing.
            self._unchecked = rt
        return loss

    def flush(self):
        """Check the counts of the last batch that ran (world 1 defers that check by one call, see step())."""
        if self._unchecked is not None:
            done, self._unchecked = self._unchecked, None
            self._route_finish(done)


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


class StepRoute:
    """ONE route for all lookups of a data-parallel training step (AttRec.py:142,190-193,205,215; Caser.py:61-64,165-166;
    SURVEY 8e config 5): the step's id lists (item sequence, users, targets | negatives ...; GLOBAL row numbers, the same
    shapes on every rank) are replicated with one all-gather, both roles of this rank are derived locally
    (dr_step_route_keys_i32 + two stable bucketings: what it asks of each owner, in which order; what each peer asks of
    it), the counts reach the host in ONE synchronisation, and then

        fetch([(table, list), ...])   ONE all-to-all brings the rows of every (table, list) pair
        push()                        ONE all-to-all takes every gradient row back to the owner of its row,

    where they are recorded for TFOptimizer.apply_table like a local lookup's, in the order a single process sees for
    the concatenated global batch (list by list, source rank by source rank, each source's in the order of its list):
    the tables of a world-W run are bit-identical to those of one process that runs the same W batch slices.
    Three collectives and one host synchronisation per step, whatever the number of tables (the per-lookup RowRoute
    below: a count exchange, a host synchronisation and three all-to-alls per lookup).  Rows of different widths share
    the wire: a peer's block holds one 16-byte-aligned segment per (table, list) pair (csrc/wire.hip)."""

    def __init__(self, ops, lists, world, rank, group=None, defer=False):
        """defer: enqueue the id-only part (all-gather, keys, bucketings, counts -> pinned host memory) and return; finish()
        -- called by the step that uses the route -- waits for an event recorded behind those kernels.  A route built
        one step AHEAD (models/_common.py: Shard.prefetch_route, enqueued before the previous step's kernels) is complete
        by then, so the step's one host synchronisation no longer drains the work in flight."""
        self.ops, self.world, self.rank, self.group = ops, int(world), int(rank), group
        W, nl = self.world, len(lists)
        flat = [t.reshape(-1) if t.dtype == torch.int32 else t.reshape(-1).to(torch.int32) for t in lists]
        self.shapes = [tuple(t.shape) for t in lists]
        self.n = [t.numel() for t in flat]
        self.start = [0]
        for c in self.n:
            self.start.append(self.start[-1] + c)
        n_tot = self.start[-1]
        mine = (torch.cat(flat) if nl > 1 else flat[0]).contiguous()
        dev = mine.device
        ids_all = torch.empty((W, n_tot), dtype=torch.int32, device=dev)
        _gather_ids(ids_all, mine, group)                                   # collective 1 of 3
        km, ko, self._loc = ops.step_route_keys(ids_all, self.rank, self.start)
        self._order_m, cnt_m = ops.bucket_order(km, nl * W)                 # my entries by (list, owner)
        self._order_o, cnt_o = ops.bucket_order(ko, nl * W)                 # entries I own by (list, source rank)
        cnt = torch.cat((cnt_m[:nl * W], cnt_o[:nl * W]))
        self._ev = None
        if cnt.is_cuda:
            self._cnt_host = torch.empty(cnt.numel(), dtype=cnt.dtype, pin_memory=True)
            self._cnt_host.copy_(cnt, non_blocking=True)
            self._ev = torch.cuda.Event()
            self._ev.record()
        else:
            self._cnt_host = cnt
        self._pos = [None] * nl
        self._specs, self._fetched, self._order = None, [], []
        self._done = False
        if not defer:
            self.finish()

    def finish(self):
        """The ONE host synchronisation of a step: the per-(list, peer) counts, then the host-side slicing that needs them."""
        if self._done:
            return
        self._done = True
        W, nl = self.world, len(self.n)
        if self._ev is not None:
            self._ev.synchronize()
        c = self._cnt_host.tolist()
        self.send_cnt = [c[l * W:(l + 1) * W] for l in range(nl)]           # [list][owner]: rows I ask of each owner
        self.recv_cnt = [c[(nl + l) * W:(nl + l + 1) * W] for l in range(nl)]   # [list][source]: rows each peer asks of me
        self.n_own = [sum(r) for r in self.recv_cnt]
        # my entries of list l in wire order (by owner, then list position): positions in `mine`, bias start[l]
        self.send_idx = [self._order_m[self.start[l]:self.start[l + 1]] for l in range(nl)]
        # local rows the peers ask of me, by source rank, each source's in its list order
        self.want, off = [], 0
        for l in range(nl):
            self.want.append(torch.index_select(self._loc, 0, self._order_o[off:off + self.n_own[l]]))
            off += self.n_own[l]

    def pos(self, l):
        """Positions 0 .. n_l - 1 in the shape of list l: what a lookup in the fetched rows is addressed with."""
        if self._pos[l] is None:
            self._pos[l] = torch.arange(self.n[l], dtype=torch.int32, device=self.want[l].device).reshape(self.shapes[l])
        return self._pos[l]

    @staticmethod
    def _prefix(counts):
        out = [0]
        for c in counts:
            out.append(out[-1] + c)
        return out

    def _layout(self, specs):
        """Wire layout of the pairs `specs` = [(k, list)]: per peer one block, inside it one segment per pair, padded to
        4 floats.  `out`: blocks I send as an owner (rows the peers asked for); `in`: blocks I receive."""
        W = self.world
        pad = lambda x: (x + 3) // 4 * 4
        ob, ib = [0] * W, [0] * W
        ow, iw = [], []
        for k, l in specs:
            ow.append(list(ob))
            iw.append(list(ib))
            for p in range(W):
                ob[p] += pad(self.recv_cnt[l][p] * k)
                ib[p] += pad(self.send_cnt[l][p] * k)
        obase, ibase = self._prefix(ob), self._prefix(ib)
        self.out_split, self.in_split = ob, ib
        self.out_off = [[obase[p] + w[p] for p in range(W)] for w in ow]
        self.in_off = [[ibase[p] + w[p] for p in range(W)] for w in iw]

    def fetch(self, pairs):
        """pairs = [(Table, list index)]: ONE exchange; returns a FetchedRows per pair (rows in the order of the list)."""
        from .models._common import FetchedRows
        ops = self.ops
        if self._specs is not None:
            raise RuntimeError("StepRoute.fetch: one fetch per route (list every (table, list) pair in it)")
        self._specs = [(t.k, l) for t, l in pairs]
        self._layout(self._specs)
        dev = self.want[0].device
        send = torch.empty(sum(self.out_split), dtype=torch.float32, device=dev)
        recv = torch.empty(sum(self.in_split), dtype=torch.float32, device=dev)
        for f, (t, l) in enumerate(pairs):                                  # the owner's gather, straight into the wire
            ops.rows_pack(t.w, self.want[l], self.n_own[l], self._prefix(self.recv_cnt[l]), self.out_off[f], send)
        _a2a(recv, send, self.in_split, self.out_split, self.group)         # collective 2 of 3
        for f, (t, l) in enumerate(pairs):
            rows = torch.empty((self.n[l],) + tuple(t.w.shape[1:]), dtype=torch.float32, device=dev)
            ops.rows_unpack(recv, self.send_idx[l], self.n[l], self._prefix(self.send_cnt[l]), self.in_off[f], rows,
                            idx_bias=self.start[l])
            self._fetched.append(FetchedRows(t, self, l, rows))
        return list(self._fetched)

    def note(self, fetched):
        """A FetchedRows received its first gradient: tables record the pushed rows in this order (the order a single
        process's autograd appends them in)."""
        if fetched not in self._order:
            self._order.append(fetched)

    def push(self):
        """ONE exchange takes the gradient rows every fetched pair collected to the owners, where they are appended to
        Table.pending (ids = local rows, in global batch order)."""
        ops = self.ops
        if not self._order:
            return
        dev = self.want[0].device
        send = torch.empty(sum(self.in_split), dtype=torch.float32, device=dev)
        recv = torch.empty(sum(self.out_split), dtype=torch.float32, device=dev)
        for f, fr in enumerate(self._fetched):
            l = fr.l
            ops.rows_pack(fr.gradient(), self.send_idx[l], self.n[l], self._prefix(self.send_cnt[l]), self.in_off[f],
                          send, idx_bias=self.start[l])
        _a2a(recv, send, self.out_split, self.in_split, self.group)         # collective 3 of 3
        got = {}
        for f, fr in enumerate(self._fetched):
            l = fr.l
            rows = torch.empty((self.n_own[l],) + tuple(fr.w.shape[1:]), dtype=torch.float32, device=dev)
            ops.rows_unpack(recv, None, self.n_own[l], self._prefix(self.recv_cnt[l]), self.out_off[f], rows)
            got[id(fr)] = rows
        for fr in self._order:
            fr.table.pending.append((self.want[fr.l], got[id(fr)]))
        self._order = []


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


def sharded_eval_ranks(ops, P, Q, users, train=None, targets=None, Q_full=None, group=None):
    """sharded_eval_topk without the top-K list: the ranks of the held-out items only (ops.score_rank: the streaming
    kernel), which is all that reduce_rank_metrics and the reference's evaluators consume.  Returns (ranks, Q_full)."""
    if Q_full is None:
        Q_full = all_gather_table(Q, group)
    U = fetch_rows(ops, P, users, group)
    return ops.score_rank(U, Q_full, targets, train=train), Q_full


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
