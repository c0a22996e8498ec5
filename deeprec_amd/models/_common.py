"""Host-side plumbing shared by the model classes: device placement, the lazy `neg_items` mapping of the
evaluator protocol, reference-compatible negative sampling, TF-semantics optimizer state, and a minimal
autograd bridge (embedding lookups whose backward is the deterministic HIP scatter-add path).

PyTorch is used for device memory, autograd bookkeeping and the small dense layers (rocBLAS / MIOpen
GEMMs and convolutions); every embedding lookup, duplicate-summing backward, optimizer application, fused
step and full-catalog evaluation goes through the C ABI (deeprec_amd.kernels).
"""
import os

import numpy as np
import torch

from .. import kernels as K


def device_of(device=None):
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("deeprec_amd models need an MI355X: torch.cuda.is_available() is False and there is "
                               "no CPU fallback")
        device = "cuda:0"
    return torch.device(device)


def shift_sess(sess, num_user, num_item):
    """D2: the reference drivers call Model(n_user, n_item) for classes whose first parameter is `sess`
    (test_item_ranking.py:53-69).  Accept both call shapes."""
    if num_item is None:
        return None, sess, num_user
    return sess, num_user, num_item


class NegItems:
    """model.neg_items: u -> ascending list of the items the user has NOT interacted with
    (bprmf.py:175-181).  Materialised per request from the train CSR instead of as U x I Python sets."""

    def __init__(self, csr, num_item, first_item=0):
        self.csr, self.num_item, self.first = csr.tocsr(), int(num_item), int(first_item)
        self.csr.sort_indices()

    def train_items(self, u):
        return self.csr.indices[self.csr.indptr[u]:self.csr.indptr[u + 1]]

    def count(self, u):
        t = self.train_items(u)
        return self.num_item - self.first - int(np.count_nonzero(t >= self.first))

    def __getitem__(self, u):
        m = np.ones(self.num_item, bool)
        m[:self.first] = False
        m[self.train_items(u)] = False
        return np.flatnonzero(m)

    def __len__(self):
        return self.csr.shape[0]

    def __bool__(self):
        return True

    def keys(self):
        return range(self.csr.shape[0])

    def nth(self, u, s):
        """neg_items[u][s] without materialising the list: the s-th item of the complement."""
        t = self.train_items(u)
        t = t[t >= self.first] - self.first
        # complement rank: item x has rank x - (#train items < x); find smallest x with rank == s, x not in t
        lo, hi = 0, len(t)
        while lo < hi:                       # largest prefix of t lying before the answer
            mid = (lo + hi) // 2
            if t[mid] - mid <= s:
                lo = mid + 1
            else:
                hi = mid
        return int(s + lo + self.first)


def sample_uniform_negatives(users, neg, n_per_user=1):
    """The reference's per-positive draw (bprmf.py:130-133, cml.py:108-111): for every user in order,
    s = np.random.randint(len(neg_items[u])); item = neg_items[u][s].  Same RNG stream, no list building."""
    out = np.empty((len(users), n_per_user), np.int32)
    for r, u in enumerate(users):
        c = neg.count(u)
        for j in range(n_per_user):
            out[r, j] = neg.nth(u, np.random.randint(c))
    return out


def device_pointwise_epoch(model):
    """One epoch of (user, item, label) training triples for the pointwise models (gmf.py:100-124, neumf.py:153-180)
    assembled on the device: num_neg_sample negatives per positive from dr_sample_negatives_i32 (uniform over the
    user's unseen items, independent draws -- the reference's random.sample draws without replacement inside one
    positive's group; with ~30 draws from thousands of candidates the two differ by a few repeated negatives),
    label 0, then one device permutation.  Replaces the per-positive Python loop.
    Returns int32 users, int32 items, float32 labels (device tensors, shuffled)."""
    dev = model.device
    if getattr(model, "_train_csr_dev", None) is None:
        c = model.neg_items.csr
        model._train_csr_dev = (to_dev(c.indptr, torch.int32, dev), to_dev(c.indices, torch.int32, dev))
        model._pos_dev = (to_dev(np.asarray(model.user), torch.int32, dev), to_dev(np.asarray(model.item), torch.int32, dev),
                          to_dev(np.asarray(model.label, dtype=np.float32), torch.float32, dev))
    u, i, y = model._pos_dev
    n_neg = int(model.num_neg_sample)
    seed = int(np.random.randint(0, 2 ** 31 - 1))        # one host draw per epoch keeps np.random.seed() reproducible
    neg = K.sample_negatives(*model._train_csr_dev, u, model.num_item, seed, n_neg=n_neg)      # [n_pos, n_neg]
    users = torch.cat((u, u[:, None].expand(-1, n_neg).reshape(-1)))
    items = torch.cat((i, neg.reshape(-1)))
    labels = torch.cat((y, torch.zeros(neg.numel(), dtype=torch.float32, device=dev)))
    keep = items >= 0                                      # a user who has seen every item has no negative (-1)
    if not bool(keep.all()):
        users, items, labels = users[keep], items[keep], labels[keep]
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    perm = torch.randperm(users.numel(), device=dev, generator=gen)
    return users[perm].contiguous(), items[perm].contiguous(), labels[perm].contiguous()


def to_dev(a, dtype, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device)


# ----------------------------------------------------------------------------------------------
# launch-bound small-batch steps: replay them as one hipGraph
# ----------------------------------------------------------------------------------------------
class GraphedStep:
    """A fixed-shape training step (the reference's B = 128 .. 1000 steps are 50-100 short launches: lookups, a few
    small GEMMs under autograd, segment plans, optimizer applications) replayed as ONE hipGraph.

    Call 1 runs `fn` eagerly (lazily created state -- optimizer slots, dense-gradient buffers, BLAS handles --
    comes into being); call 2 copies its arguments into static buffers, captures `fn` on them (capture executes
    nothing) and replays; later calls copy and replay.  Same kernels on the same data in the same order, so results
    are bit-identical to the eager loop.  Only for steps whose host side is stateless from step to step (Adagrad /
    RMSProp / SGD; Adam with TFOptimizer.device_alpha(), which moves the bias correction to the device).

    A capture that fails is not hidden: the exception is kept in `capture_error` and reported with a warning, the host
    state the aborted capture touched (the optimizer's step counter and beta powers, the tables' recorded lookup
    gradients, which point at capture-pool memory that never executed) is put back, and the step is run eagerly from
    then on.  `owner` is the model whose TFOptimizer / Table attributes are guarded that way.  `replays` counts graph
    launches (tests assert that the models meant to replay really do).  DR_NO_GRAPH=1 keeps everything eager."""

    def __init__(self, fn, owner=None):
        import os
        self.fn, self.graph, self.calls, self.replays = fn, None, 0, 0
        self.owner = owner if owner is not None else getattr(fn, "__self__", None)
        self.off = os.environ.get("DR_NO_GRAPH") is not None
        self.capture_error = None
        self.static, self.out = None, None

    def _host_state(self):
        opts, tables = [], []
        for v in (vars(self.owner).values() if self.owner is not None else ()):
            if isinstance(v, TFOptimizer):
                opts.append((v, (v.step_count, v._b1p, v._b2p)))
            elif isinstance(v, Table):
                tables.append((v, list(v.pending), list(v._deferred)))
        return opts, tables

    @staticmethod
    def _restore(state):
        for o, (n, b1, b2) in state[0]:
            o.step_count, o._b1p, o._b2p = n, b1, b2
        for t, pend, deferred in state[1]:
            t.pending, t._deferred = pend, deferred

    def __call__(self, *tensors):
        if self.off:
            return self.fn(*tensors)
        self.calls += 1
        if self.calls == 1:
            return self.fn(*tensors)
        if self.graph is None:
            state = self._host_state()
            try:
                self.static = [t.clone() for t in tensors]
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.out = self.fn(*self.static)
                self.graph = g
            except Exception as e:                      # noqa: BLE001 -- reported, not swallowed
                import warnings
                self.capture_error = e
                self.off = True
                self._restore(state)
                torch.cuda.synchronize()
                warnings.warn("GraphedStep: hipGraph capture of %s failed (%s: %s); running eagerly"
                              % (getattr(self.fn, "__qualname__", self.fn), type(e).__name__, e), RuntimeWarning)
                return self.fn(*tensors)
        elif any(s.shape != t.shape or s.dtype != t.dtype for s, t in zip(self.static, tensors)):
            self.off = True
            return self.fn(*tensors)
        else:
            for s, t in zip(self.static, tensors):
                s.copy_(t)
        self.graph.replay()
        self.replays += 1
        return self.out


# ----------------------------------------------------------------------------------------------
# optimizer state with TensorFlow semantics ([TF-ext]; hyper-parameters cited in the model files)
# ----------------------------------------------------------------------------------------------
SPARSE_DENSE_MIN = 1 << 22     # elements: below this a table's dense-L2 step keeps the dense-gradient form


class TFOptimizer:
    """kind: 'adam' (TF1 eps 1e-8 / Keras 1e-7), 'adagrad' (initial accumulator 0.1), 'rmsprop', 'sgd'."""
    KIND = {"sgd": K.OPT_SGD, "adam": K.OPT_ADAM, "adagrad": K.OPT_ADAGRAD, "rmsprop": K.OPT_RMSPROP}
    NSLOTS = {"sgd": 0, "adam": 2, "adagrad": 1, "rmsprop": 2}

    def __init__(self, kind, learning_rate, eps=None, beta1=0.9, beta2=0.999, initial_accumulator_value=0.1,
                 rho=0.9, momentum=0.0):
        self.kind, self.lr = kind, float(learning_rate)
        self.eps = float(eps if eps is not None else (1e-8 if kind == "adam" else 1e-10))
        self.beta1, self.beta2, self.acc0, self.rho, self.momentum = beta1, beta2, initial_accumulator_value, rho, momentum
        self.step_count = 0
        self._b1p = np.float32(1)
        self._b2p = np.float32(1)
        self.slots = {}
        self._dev = None            # K.AdamState: the bias-corrected step size lives on the device (device_alpha())

    def slots_for(self, t):
        """Optimizer slots of parameter tensor `t`.  Keyed by the tensor's address, and the entry keeps `t` alive: its
        memory cannot be freed and handed to another tensor while the entry exists, so an address never names two
        parameters.  A parameter that is REPLACED (not updated in place) must be released with drop(old) -- Table.w's
        setter and load_checkpoint() copy in place instead, which keeps the slots attached."""
        key = t.data_ptr()
        ent = self.slots.get(key)
        if ent is not None and ent[0].shape != t.shape:
            raise RuntimeError("TFOptimizer: a parameter of shape %s now lives where one of shape %s was registered; "
                               "call drop(old) when replacing a parameter tensor" % (tuple(t.shape), tuple(ent[0].shape)))
        if ent is None:
            n = self.NSLOTS[self.kind]
            if self.kind == "adagrad":
                s = [torch.full_like(t, self.acc0)]
            elif self.kind == "rmsprop":
                s = [torch.ones_like(t), torch.zeros_like(t)]      # [TF-ext] RMSProp ms starts at 1, mom at 0
            else:
                s = [torch.zeros_like(t) for _ in range(n)]
            ent = (t, (s + [None, None])[:2])
            self.slots[key] = ent
        return ent[1]

    def drop(self, t):
        """Forget the slots of a parameter tensor that is being replaced (frees 1-2x its size)."""
        self.slots.pop(t.data_ptr(), None)

    def state(self):
        """Step counter and beta powers (what TensorFlow keeps in beta1_power / beta2_power) for checkpoints."""
        return [float(self.step_count), float(self._b1p), float(self._b2p)]

    def set_state(self, st):
        self.step_count = int(st[0])
        self._b1p, self._b2p = np.float32(st[1]), np.float32(st[2])
        if self._dev is not None:
            self._dev.state[0] = float(self._b1p)
            self._dev.state[1] = float(self._b2p)

    def device_alpha(self, device):
        """Adam only: keep beta1^t, beta2^t and alpha_t on the device (dr_adam_tick_f32: the same fp32 running products
        and the same expression as below), so that a training step has no per-step host scalar and can be replayed
        as a hipGraph (GraphedStep).  Call before the first step."""
        if self.kind == "adam" and self._dev is None:
            self._dev = K.AdamState(self.lr, self.beta1, self.beta2, device)
            if self.step_count:          # steps already taken with the host-side products: carry them over
                self._dev.state[0] = float(self._b1p)
                self._dev.state[1] = float(self._b2p)
        return self

    def begin_step(self):
        self.step_count += 1
        self._b1p = np.float32(self._b1p * np.float32(self.beta1))
        self._b2p = np.float32(self._b2p * np.float32(self.beta2))
        if self._dev is not None:
            self._dev.tick()

    def params(self, reg, reg_mode):
        """reg: a number, or a 1-element float32 DEVICE tensor (a coefficient computed on the device, e.g. NNMF's
        reg_rate / ||W||: dr_opt_params.reg_dev) -- the caller keeps that tensor alive until the kernels have run."""
        alpha = 0.0
        if self.kind == "adam":
            alpha = float(np.float32(self.lr) * np.sqrt(np.float32(1) - self._b2p) / (np.float32(1) - self._b1p))
        reg_dev = reg if isinstance(reg, torch.Tensor) else None
        return K.opt_params(self.KIND[self.kind], lr=self.lr, alpha=alpha, beta1=self.beta1, beta2=self.beta2,
                            eps=self.eps, rho=self.rho, momentum=self.momentum,
                            reg=0.0 if reg_dev is not None else float(reg), reg_mode=reg_mode,
                            alpha_dev=self._dev.alpha if self._dev is not None else None, reg_dev=reg_dev)

    def apply_dense(self, w, grad, reg=0.0):
        """Whole-tensor application; reg > 0 adds reg*w (tf l2_regularizer / l2_loss terms)."""
        s0, s1 = self.slots_for(w)
        has_reg = isinstance(reg, torch.Tensor) or bool(reg)
        K.dense_update(w, s0, s1, grad, self.params(reg, K.REG_DENSE if has_reg else K.REG_NONE))

    def apply_table(self, table, reg, reg_mode):
        """Apply the recorded lookups' gradients of one embedding table."""
        pend, table.pending = table.pending, []
        if (reg_mode == K.REG_DENSE and not pend and table.lazy and self._dev is None
                and not isinstance(reg, torch.Tensor)
                and not (table._w.is_cuda and torch.cuda.is_current_stream_capturing())):
            table.defer(self, self.params(reg, K.REG_DENSE))      # replayed when someone reads table.w
            return
        s0, s1 = self.slots_for(table.w)
        if reg_mode == K.REG_DENSE:
            if not pend:
                # no lookup this step (AttRec's V): the gradient is reg * w alone -- no zero-filled dense gradient to
                # write and read back (the kernel computes 0 + reg * w either way, so the result is bit-identical)
                K.dense_update(table.w, s0, s1, None, self.params(reg, K.REG_DENSE))
                return
            if len(pend) == 1 and table.w.numel() >= SPARSE_DENSE_MIN:
                # one lookup of a big table: the rows it touched get sum + reg*w from the sparse kernel, every other
                # row reg*w from a dense pass that skips them -- no zero-filled dense gradient to write, scatter into
                # and read back (bit-identical: the dense path computes (0 + sum) + reg*w for the same rows)
                ids, rows = pend[0]
                flags = table.row_flags()
                K.mark_rows(flags, ids, 1)
                K.dense_update_masked(table.w, s0, s1, flags, self.params(reg, K.REG_DENSE))
                K.sparse_row_update(table.w, s0, s1, ids, rows, self.params(reg, K.REG_TOUCHED))
                K.mark_rows(flags, ids, 0)
                return
            G = table.dense_grad()
            for ids, rows in pend:
                K.scatter_add(G, ids, rows)
            K.dense_update(table.w, s0, s1, G, self.params(reg, K.REG_DENSE))
            return
        if not pend:
            return
        ids = torch.cat([p[0] for p in pend]) if len(pend) > 1 else pend[0][0]
        rows = torch.cat([p[1] for p in pend]) if len(pend) > 1 else pend[0][1]
        K.sparse_row_update(table.w, s0, s1, ids.contiguous(), rows.contiguous(), self.params(reg, reg_mode))


class Shard:
    """Placement of row-sharded tables in a data-parallel model: global row r of every table
    lives on rank r % world at local row r // world; each rank feeds its own slice of the batch.
    force: take the sharded path at world 1 as well (every exchange through a size-1 communicator; measurements)."""

    def __init__(self, world=1, rank=0, group=None, force=False):
        self.world, self.rank, self.group, self.force = int(world), int(rank), group, bool(force)
        self._pre = []                                      # routes enqueued ahead of their steps: [(key tensors, StepRoute)]

    @property
    def active(self):
        return self.world > 1 or self.force

    def local_rows(self, rows):
        return (rows - self.rank + self.world - 1) // self.world

    def take(self, full):
        """This rank's rows of a replicated array / tensor."""
        return full[self.rank::self.world]

    def step_route(self, lists, key=None):
        """ONE route for all lookups of a training step (deeprec_amd/sharding.py:StepRoute): 3 collectives, 1 host sync.
        lists: the id lists or a callable returning them; key: the step's input tensors -- a route that prefetch_route
        enqueued for the same tensors is finished here instead of built."""
        from .. import sharding as sh
        if key is not None:
            for n, (k2, rt) in enumerate(self._pre):
                if len(k2) == len(key) and all(a is b for a, b in zip(k2, key)):
                    del self._pre[n]
                    rt.finish()
                    return rt
        return sh.StepRoute(K, lists() if callable(lists) else lists, self.world, self.rank, self.group)

    def prefetch_route(self, key, lists):
        """Enqueue the id-only part of a LATER step's route now (its id all-gather, bucketings and the copy of the counts to
        pinned host memory).  Called BEFORE the current step is
        enqueued, the route's kernels run ahead of that step's, so when the later step asks for the route its event
        has long been reached and the step's one host synchronisation does not drain the work in flight.  key: the
        tensors the later step will be called with.  Every rank calls it at the same point of its step sequence."""
        from .. import sharding as sh
        # same stream, same communicator as the steps: a stream / communicator of its own for this part measured 6 % faster
        # (AttRec 2.15 against 2.28 ms at world 1) but HUNG in a process with many streams (tools/bench_configs.py: the
        # side stream's event never completed; that stream had priority -1, and a high-priority main stream in bench.py hung the
        # same way later: no stream here has a priority), so it is not offered
        rt = sh.StepRoute(K, lists() if callable(lists) else lists, self.world, self.rank, self.group, defer=True)
        self._pre.append((tuple(key), rt))
        if len(self._pre) > 2:                              # routes nobody came for
            del self._pre[0]

    def all_reduce_grads(self, params):
        """Data-parallel dense weights: the per-rank gradients summed in ONE all-reduce of their concatenation."""
        if not self.active:
            return
        from .. import sharding as sh
        gs = [p.grad for p in params if p.grad is not None]
        if not gs:
            return
        flat = torch.cat([g.reshape(-1) for g in gs])
        sh._all_reduce(flat, self.group)
        off = 0
        for g in gs:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


def active_shard(shard):
    return shard if (shard is not None and shard.active) else None


class Table:
    """An embedding table [rows, k] (or a [rows] vector) living in HBM.  With `shard` (world > 1), `w` is this
    rank's shard: lookups fetch rows from their owners and the backward sends gradient rows to them, where they are
    recorded like local ones (deeprec_amd/sharding.py:RowRoute), so the optimizer code is the same."""

    def __init__(self, w, shard=None):
        self._w = w.contiguous()
        self.pending = []
        # lazy = True (set by a model for a table its graph never reads, e.g. Caser's / AttRec's V, which only appears
        # in the L2 term): a step in which the table has no lookup does not run its dense optimizer pass but records it;
        # the recorded passes are replayed, in order and with each step's own parameters, the moment anything reads
        # `w` (save, getUserParam, a lookup).  Same operations on the same values in the same order, only later -- or
        # never, if nobody ever looks (Caser at C5: a 25.6 GB table, 33 of 55 ms per step).
        self.lazy = False
        self._deferred = []
        self._g = None
        self._hook = torch.zeros((), device=w.device, requires_grad=True)
        self.shard = active_shard(shard)

    @property
    def w(self):
        if self._deferred:
            self.flush()
        return self._w

    @w.setter
    def w(self, value):
        """New VALUES for the table: copied in place (same shape), so optimizer slots -- keyed by the tensor -- stay
        attached.  Recorded dense passes of the old values are dropped with them."""
        self._deferred = []
        if value is self._w:
            return
        if value.shape != self._w.shape:
            raise ValueError("Table.w: shape %s does not match the table's %s" % (tuple(value.shape), tuple(self._w.shape)))
        self._w.copy_(value)

    def flush(self):
        """Run the recorded dense passes of a lazy table (see __init__)."""
        todo, self._deferred = self._deferred, []
        for opt, prm, _, count in todo:
            s0, s1 = opt.slots_for(self._w)
            for _ in range(count):
                K.dense_update(self._w, s0, s1, None, prm)

    def defer(self, opt, prm):
        """Record one dense pass; consecutive passes with identical parameters (Adagrad always; Adam once its bias
        correction has converged in fp32) share one entry, so the record stays small over long runs."""
        raw = bytes(prm)
        if self._deferred and self._deferred[-1][0] is opt and self._deferred[-1][2] == raw:
            self._deferred[-1][3] += 1
        else:
            self._deferred.append([opt, prm, raw, 1])

    def deferred_passes(self):
        return sum(e[3] for e in self._deferred)

    @property
    def k(self):
        return 1 if self._w.dim() == 1 else self._w.shape[1]

    def row_flags(self):
        """uint8 [rows], all zero between steps (TFOptimizer.apply_table marks the looked-up rows while it runs)."""
        if getattr(self, "_flags", None) is None:
            self._flags = torch.zeros(self._w.shape[0], dtype=torch.uint8, device=self._w.device)
        return self._flags

    def dense_grad(self):
        if self._g is None:
            self._g = torch.empty_like(self.w)
        self._g.zero_()
        return self._g

    def lookup(self, ids):
        """tf.nn.embedding_lookup with the HIP gather; differentiable (backward is recorded for apply_table)."""
        if self.shard is not None:
            return _ShardedLookup.apply(self._hook, self, ids)
        return _Lookup.apply(self._hook, self, ids)


class _ShardedLookup(torch.autograd.Function):
    """Lookup of GLOBAL ids in a row-sharded table: ids -> owners, owners gather (K1), rows back; the backward
    sends one gradient row per id to the owner, which records it for apply_table in canonical order."""

    @staticmethod
    def forward(ctx, hook, table, ids):
        from .. import sharding as sh
        flat = ids.reshape(-1)
        if flat.dtype != torch.int32:
            flat = flat.to(torch.int32)
        rt = sh.RowRoute(K, flat.contiguous(), table.shard.world, table.shard.group)
        vec = table.w.dim() == 1
        k = 1 if vec else table.w.shape[1]
        served = K.gather(table.w, rt.want) if rt.want.numel() else torch.empty((0,) + (() if vec else (k,)),
                                                                                dtype=torch.float32, device=ids.device)
        ctx.table, ctx.rt = table, rt
        return rt.fetch(served).reshape(tuple(ids.shape) + (() if vec else (k,)))

    @staticmethod
    def backward(ctx, gout):
        t, rt = ctx.table, ctx.rt
        g = gout.reshape(rt.n, -1) if t.w.dim() == 2 else gout.reshape(-1)
        t.pending.append((rt.want, rt.push(g.contiguous())))
        return None, None, None


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hook, table, ids):
        ctx.table, ctx.ids = table, ids
        return K.gather(table.w, ids)

    @staticmethod
    def backward(ctx, gout):
        t = ctx.table
        ids = ctx.ids.reshape(-1)
        if ids.dtype != torch.int32:
            ids = ids.to(torch.int32)
        t.pending.append((ids.contiguous(), gout.reshape(ids.numel(), -1).contiguous() if t.w.dim() == 2
                          else gout.reshape(-1).contiguous()))
        return None, None, None


class _Recorder(list):
    """pending list of a FetchedRows: tells the route when the first gradient arrives (the order tables record in)."""

    def __init__(self, owner):
        super().__init__()
        self.owner = owner

    def append(self, entry):
        self.owner.route.note(self.owner)
        super().append(entry)


class FetchedRows:
    """The rows one id list of a StepRoute fetched from a row-sharded Table, presented as a small table that is addressed
    by POSITION in the list (route.pos(l)): `.w` [n_l, k] (or [n_l]) holds the rows in list order, lookup() and the fused
    loss kernels work on it as on a Table, and the gradients they record are collected per position and pushed to the
    owners by route.push()."""

    def __init__(self, table, route, l, rows):
        self.table, self.route, self.l = table, route, l
        self._w = self.w = rows
        self.pending = _Recorder(self)
        self._hook = table._hook
        self.shard = None

    @property
    def k(self):
        return 1 if self.w.dim() == 1 else self.w.shape[1]

    def _is_identity(self, pos):
        ident = self.route.pos(self.l)
        return pos.numel() == ident.numel() and pos.data_ptr() == ident.data_ptr() and pos.is_contiguous()

    def lookup(self, pos):
        if self._is_identity(pos):
            return _FetchedAll.apply(self._hook, self, tuple(pos.shape))     # every position in order: no copy
        return _Lookup.apply(self._hook, self, pos)

    def gradient(self):
        """d loss / d row per position [n_l, k] (zeros where nothing was recorded)."""
        ent, n = list(self.pending), self.w.shape[0]
        del self.pending[:]
        if len(ent) == 1 and self._is_identity(ent[0][0]):
            return ent[0][1].reshape(self.w.shape).contiguous()
        g = torch.zeros_like(self.w)
        for pos, rows in ent:
            K.scatter_add(g, pos.reshape(-1).contiguous(), rows.contiguous())
        return g


class _FetchedAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hook, fr, shape):
        ctx.fr = fr
        return fr.w.view(shape + tuple(fr.w.shape[1:]))

    @staticmethod
    def backward(ctx, gout):
        fr = ctx.fr
        fr.pending.append((fr.route.pos(fr.l), gout.reshape(fr.w.shape).contiguous()))
        return None, None, None


class _BiInteraction(torch.autograd.Function):
    """NFM bi-interaction pooling fused with its lookup (nfm.py:54-61)."""

    @staticmethod
    def forward(ctx, hook, table, feats):
        fm, s = K.bi_interaction_fwd(table.w, feats)
        ctx.table, ctx.feats, ctx.s = table, feats, s
        return fm

    @staticmethod
    def backward(ctx, gfm):
        t = ctx.table
        gv = K.bi_interaction_bwd(t.w, ctx.feats, gfm.contiguous(), ctx.s)
        t.pending.append((ctx.feats.reshape(-1), gv.reshape(-1, gv.shape[-1])))
        return None, None, None


class _FusedMLP(torch.autograd.Function):
    """h_{l+1} = act_l(h_l W_l + b_l) for a whole tower: dr_mlp_forward_f32 / dr_mlp_backward_f32 (three launches for
    forward + backward instead of ~9 per layer under autograd)."""

    @staticmethod
    def forward(ctx, x, kinds, *params):
        L = len(kinds)
        Ws, bs = params[:L], params[L:]
        xc = x.contiguous()
        acts = K.mlp_forward(xc, list(Ws), list(bs), kinds)
        ctx.kinds = kinds
        ctx.save_for_backward(xc, acts, *Ws)
        return acts[:, acts.shape[1] - Ws[-1].shape[1]:].contiguous()

    @staticmethod
    def backward(ctx, gout):
        x, acts, *Ws = ctx.saved_tensors
        gx, gW, gb = K.mlp_backward(x, list(Ws), ctx.kinds, acts, gout.contiguous(), want_gx=ctx.needs_input_grad[0])
        return (gx, None) + tuple(gW) + tuple(gb)


def fused_mlp(x, Ws, bs, kinds):
    """Dense tower on the HIP kernels (2..8 layers, widths <= 256, B <= 1024: where a chain of library GEMM + bias +
    activation launches is pure latency).  A single layer, a large batch, anything wider, or DR_NO_FUSED_MLP=1 uses
    torch matmuls under autograd (measured: NFM's one hidden layer and B = 4096 are faster there)."""
    import os
    if (os.environ.get("DR_NO_FUSED_MLP") is not None or not (2 <= len(Ws) <= K.MLP_MAX_LAYERS) or x.shape[0] > 1024
            or max([x.shape[1]] + [w.shape[1] for w in Ws]) > K.MLP_MAX_DIM):   # large batches: library GEMMs win
        for W, b, kind in zip(Ws, bs, kinds):
            x = x @ W + b
            x = torch.relu(x) if kind == K.ACT_RELU else (torch.sigmoid(x) if kind == K.ACT_SIGMOID else x)
        return x
    return _FusedMLP.apply(x, tuple(int(a) for a in kinds), *Ws, *bs)


class _AttnSeq(torch.autograd.Function):
    """AttRec.py:142-186 between the lookup and the distances: x = v + timing; pre = x W (library GEMM);
    out = dr_attn_pool_forward_f32(pre, v).  Three launches forward, four backward, instead of ~25 under autograd."""

    @staticmethod
    def forward(ctx, v, timing, W):
        B, L, d = v.shape
        v = v.contiguous()
        x = (v + timing).reshape(B * L, d)
        pre = (x @ W).reshape(B, L, d)
        out, s = K.attn_pool_forward(pre, v)
        ctx.save_for_backward(v, x, pre, s, W)
        ctx.mark_non_differentiable(s)
        return out, s

    @staticmethod
    def backward(ctx, gout, _gs):
        v, x, pre, s, W = ctx.saved_tensors
        B, L, d = v.shape
        dpre, dv = K.attn_pool_backward(pre, v, s, gout.contiguous())
        dpre2 = dpre.reshape(B * L, d)
        # dW = x^T dpre is a [d, d] result over B*L rows: as ONE GEMM the library runs it on a handful of workgroups
        # (0.48 ms at B*L = 204 800, d = 128); split over S row blocks it is a batched GEMM plus a [S, d, d] sum
        n = B * L
        S = int(np.gcd(n, 64))
        if S > 1 and n // S >= 256:
            dW = torch.bmm(x.reshape(S, n // S, d).transpose(1, 2), dpre2.reshape(S, n // S, d)).sum(0)
        else:
            dW = x.t() @ dpre2
        # the path through the projection, added onto the direct part in the GEMM's epilogue
        dv2 = dv.reshape(n, d)
        dv2.addmm_(dpre2, W.t())
        return dv, None, dW


def attn_seq(v, timing, W):
    """(out [B, d], s [B, L, L] unmasked softmax) of AttRec's self-attention block on the HIP kernels."""
    return _AttnSeq.apply(v, timing, W)


def _check_unit_grad(gl):
    import os
    if os.environ.get("DR_CHECK_GL") and float(gl) != 1.0:
        raise RuntimeError("fused loss: backward() was called with gradient %r, but the recorded table rows are unscaled; "
                           "pass the scale into the loss kernel instead" % float(gl))


class _SeqDistLoss(torch.autograd.Function):
    """Two-distance interaction + pairwise loss fused with the item / user lookups (dr_seq_dist_loss_f32): the
    forward launches ONE kernel that also produces every gradient; the backward records the gradient rows of the
    tables X, Q, P for the segment update and hands d loss / d o back to autograd.

    The result must be the tensor backward() is called on (as the reference's `minimize(self.loss)` does,
    AttRec.py:110, PRME.py:98): the recorded TABLE rows are d loss / d row as they stand -- the incoming gradient (1) is
    not read back from the device, so that the step stays free of host synchronisation and replays as a hipGraph.  The
    gradient handed back to autograd IS scaled by the incoming one (on the device), and DR_CHECK_GL=1 makes backward()
    verify that the incoming gradient is 1 (a host read-back: debugging only) -- a caller that wants a scaled loss
    passes the scale INTO the kernel (seq_dot_loss's `scale`) instead of multiplying the result."""

    @staticmethod
    def forward(ctx, o, tP, tX, tQ, users, tgt, neg, alpha, kind, margin, items):
        loss_b, gX, gQ, gp, go = K.seq_dist_loss(tP.w, tX.w, tQ.w, users, o.contiguous(), tgt, neg, alpha, 1.0 - alpha,
                                                 kind, margin)
        ctx.tabs = (tP, tX, tQ)
        ctx.ids = (users, items.reshape(-1) if items is not None else torch.cat((tgt, neg), 1).reshape(-1))
        ctx.save_for_backward(gX, gQ, gp, go)
        return K.sum_f32(loss_b).reshape(())

    @staticmethod
    def backward(ctx, gl):
        gX, gQ, gp, go = ctx.saved_tensors
        tP, tX, tQ = ctx.tabs
        users, items = ctx.ids
        k = gp.shape[1]
        tX.pending.append((items, gX.reshape(-1, k)))
        tQ.pending.append((items, gQ.reshape(-1, k)))
        tP.pending.append((users, gp))
        _check_unit_grad(gl)
        return (go * gl,) + (None,) * 10


def seq_dist_loss(o, tP, tX, tQ, users, tgt, neg, alpha, kind, margin=0.5, items=None):
    """Scalar loss of the two-distance models; `o` [B, k] is an autograd tensor (attention output / previous-item rows),
    tP / tX / tQ are Tables (or FetchedRows) whose gradients are recorded like lookup backwards.  items: the tensor
    cat((tgt, neg), 1) when the caller holds it already (the ids the item gradients are recorded under).  Call
    backward() on the result."""
    return _SeqDistLoss.apply(o, tP, tX, tQ, users, tgt, neg, float(alpha), int(kind), float(margin), items)


class _SeqDotLoss(torch.autograd.Function):
    """Caser's score + mean BCE fused with the W / b lookups (dr_seq_dot_loss_f32); same contract as _SeqDistLoss."""

    @staticmethod
    def forward(ctx, x, tW, tb, tgt, neg, scale, items):
        loss_b, gW, gb, gx = K.seq_dot_loss(tW.w, tb.w, x.contiguous(), tgt, neg, scale)
        ctx.tabs, ctx.items = (tW, tb), (items.reshape(-1) if items is not None else torch.cat((tgt, neg), 1).reshape(-1))
        ctx.save_for_backward(gW, gb, gx)
        return K.sum_f32(loss_b).reshape(())

    @staticmethod
    def backward(ctx, gl):
        gW, gb, gx = ctx.saved_tensors
        tW, tb = ctx.tabs
        tW.pending.append((ctx.items, gW.reshape(-1, gW.shape[2])))
        tb.pending.append((ctx.items, gb.reshape(-1)))
        _check_unit_grad(gl)
        return (gx * gl, None, None, None, None, None, None)


def seq_dot_loss(x, tW, tb, tgt, neg, scale=1.0, items=None):
    return _SeqDotLoss.apply(x, tW, tb, tgt, neg, float(scale), items)


def bi_interaction(table, feats):
    return _BiInteraction.apply(table._hook, table, feats)


def normal_(shape, std, device, seed=None):
    """Initial values ~ N(0, std^2).  TensorFlow's RNG stream cannot be reproduced: parity tests
    pass tables in explicitly; here the counter-based generator of the library is used."""
    w = torch.empty(shape, dtype=torch.float32, device=device)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    K.init_normal_(w, seed, std)
    return w


# ----------------------------------------------------------------------------------------------
# checkpoints: tables, dense weights, optimizer slots, step counter and beta powers
# ----------------------------------------------------------------------------------------------
def checkpoint_tensors(model):
    """name -> live parameter tensor of a model: its Tables, its dense weights (tensors that require grad, and lists of
    them), and the plain tensors a class names in CKPT (BPRMF / GMF / MF / CML keep their tables as tensors)."""
    out = {}
    names = getattr(model, "CKPT", None)
    for attr, v in vars(model).items():
        if names is not None and attr in names and isinstance(v, torch.Tensor):
            out[attr] = v
        elif isinstance(v, Table):
            out[attr] = v.w                                    # flushes a lazy table's recorded passes first
        elif isinstance(v, torch.Tensor) and v.requires_grad and v.is_leaf:
            out[attr] = v.data
        elif isinstance(v, (list, tuple)) and v and all(isinstance(x, torch.Tensor) and x.requires_grad and x.is_leaf
                                                        for x in v):
            for i, x in enumerate(v):
                out["%s.%d" % (attr, i)] = x.data
    return out


def save_checkpoint(model, path):
    """safetensors file with every parameter, the optimizer slots that exist for it (`<name>.slot<j>`) and the
    optimizer's step counter / beta powers (`__optimizer__`).  The reference's save() (gmf.py:156-158 and 22 more
    sites) is tf.train.Saver and never called; nothing here is pickled."""
    from safetensors.torch import save_file
    opt = getattr(model, "optimizer", None)
    d = {}
    for n, t in checkpoint_tensors(model).items():
        d[n] = t
        ent = opt.slots.get(t.data_ptr()) if opt is not None else None
        if ent is not None:
            for j, sl in enumerate(ent[1]):
                if sl is not None:
                    d["%s.slot%d" % (n, j)] = sl
    if opt is not None:
        d["__optimizer__"] = torch.tensor(opt.state(), dtype=torch.float64)
    save_file({k: v.detach().cpu().contiguous() for k, v in d.items()}, path)


def load_checkpoint(model, path):
    """Inverse of save_checkpoint on a model whose build_network() has run with the same shapes: values are copied IN
    PLACE into the existing tables / weights / slots (so the optimizer's slot registry stays valid), the step counter
    and beta powers are restored -- training continues exactly where the saved model stopped."""
    from safetensors.torch import load_file
    sl = load_file(path)
    opt = getattr(model, "optimizer", None)
    live = checkpoint_tensors(model)
    missing = [n for n in live if n not in sl]
    if missing:
        raise KeyError("checkpoint %s lacks %s" % (path, missing))
    with torch.no_grad():
        for n, t in live.items():
            if tuple(sl[n].shape) != tuple(t.shape):
                raise ValueError("checkpoint tensor %s has shape %s, the model's %s" % (n, tuple(sl[n].shape), tuple(t.shape)))
            t.copy_(sl[n])
            if opt is not None and ("%s.slot0" % n) in sl:
                for j, s_ in enumerate(opt.slots_for(t)):
                    if s_ is not None:
                        s_.copy_(sl["%s.slot%d" % (n, j)])
    for v in vars(model).values():
        if isinstance(v, Table):
            v._deferred, v.pending = [], []
    if opt is not None and "__optimizer__" in sl:
        opt.set_state(sl["__optimizer__"].tolist())
