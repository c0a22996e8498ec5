"""Torch-tensor front end of the C ABI (device memory and streams only; no torch math here).

Every function takes CUDA(HIP) tensors, validates shapes / dtypes / contiguity on the host
(the kernels assume them), and enqueues on torch's current stream.
"""
import ctypes as C

import torch

from . import _abi
from ._abi import OptParams, check

OPT_SGD, OPT_ADAM, OPT_ADAGRAD, OPT_RMSPROP = 0, 1, 2, 3
REG_TOUCHED, REG_DENSE, REG_NONE = 0, 1, 2
SCORE_DOT, SCORE_NEG_SQDIST, SCORE_TWO_DIST = 0, 1, 2
LOSS_BCE, LOSS_SQ = 0, 1
OOB_ZERO, OOB_ERROR = 0, 1


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    """Raw handle of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object
    through several Python layers (~8 us, measured: 14 such calls were 0.12 ms of a 0.8 ms sharded step); the two C
    calls below return the same handle in well under a microsecond."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


_INITED = set()


def init(device=None):
    """dr_init for the current (or given) device: the library's helper streams / events and the static scratch address.
    Called by the workspace constructors; idempotent."""
    dev = torch._C._cuda_getDevice() if device is None else torch.device(device).index
    if dev not in _INITED:
        check(_abi.lib().dr_init(int(dev)), "dr_init")
        _INITED.add(dev)


def _on_current_device(t, name):
    """Launches go to the CURRENT device's current stream: a tensor of another GPU would be read through that GPU's
    pointer on the wrong stream (or fault).  One process drives one GPU here; anything else is refused."""
    cur = torch._C._cuda_getDevice()
    if t.device.index != cur:
        raise ValueError("%s lives on %s but the current device is cuda:%d (torch.cuda.set_device first)"
                         % (name, t.device, cur))


def _req(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP tensor (deeprec_amd has no CPU path)" % name)
    _on_current_device(t, name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dims, got shape %s" % (name, ndim, tuple(t.shape)))
    return t


def _ids(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP tensor" % name)
    _on_current_device(t, name)
    if t.dtype not in (torch.int32, torch.int64):
        raise TypeError("%s must be int32 or int64, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t, (32 if t.dtype == torch.int32 else 64)


def _i32(t, name):
    t = _req(t, torch.int32, name)
    return t


def opt_params(kind, lr=0.0, alpha=0.0, beta1=0.9, beta2=0.999, eps=1e-8, rho=0.9, momentum=0.0, reg=0.0,
               reg_mode=REG_TOUCHED, alpha_dev=None, reg_dev=None):
    """alpha_dev / reg_dev: optional device float tensors (1 element) overriding alpha / reg at kernel run time (graph
    replay; coefficients computed on the device).  The caller keeps them alive until the kernels have run."""
    return OptParams(kind, lr, alpha, beta1, beta2, eps, rho, momentum, reg, reg_mode,
                     alpha_dev.data_ptr() if alpha_dev is not None else None,
                     reg_dev.data_ptr() if reg_dev is not None else None)


class AdamState:
    """Device-side beta-power state (dr_adam_tick_f32): tick() once per step, pass .alpha to opt_params.

    ahead = 1: .alpha is one fixed device scalar (what a captured graph needs).  ahead = n > 1 (eager step loops): one
    launch computes the next n step sizes (dr_adam_tick_many_f32, the same bits) and .alpha moves through them, so
    build the opt_params (or set its alpha_dev) after every tick()."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, device="cuda:0", ahead=1):
        self.lr, self.beta1, self.beta2 = float(lr), float(beta1), float(beta2)
        self.state = torch.tensor([1.0, 1.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.alpha = self.state[2:3]
        self.ahead = int(ahead)
        if self.ahead > 1:
            # two halves: the kernels of the last steps of one half may still be reading it while the next is written
            self._ring = torch.zeros(2 * self.ahead, dtype=torch.float32, device=device)
            self._half, self._i = 1, self.ahead

    def tick(self):
        if self.ahead > 1:
            if self._i == self.ahead:
                self._half ^= 1
                self._i = 0
                check(_abi.lib().dr_adam_tick_many_f32(_ptr(self.state), self.lr, self.beta1, self.beta2,
                                                       self._ring.data_ptr() + 4 * self._half * self.ahead, self.ahead,
                                                       _stream()), "dr_adam_tick_many_f32")
            o = self._half * self.ahead + self._i
            self.alpha = self._ring[o:o + 1]
            self._i += 1
            return
        check(_abi.lib().dr_adam_tick_f32(_ptr(self.state), self.lr, self.beta1, self.beta2, _stream()),
              "dr_adam_tick_f32")


# ------------------------------------------------------------------ K1
def gather(table, ids, oob="zero"):
    """tf.nn.embedding_lookup(table, ids) (bprmf.py:37).  table [R,k] or [R]; ids any shape."""
    table = _req(table, torch.float32, "table")
    ids, bits = _ids(ids, "ids")
    vec = table.dim() == 1
    k = 1 if vec else table.shape[1]
    if table.dim() not in (1, 2):
        raise ValueError("table must be [rows] or [rows, k]")
    n = ids.numel()
    out = torch.empty(tuple(ids.shape) + (() if vec else (k,)), dtype=torch.float32, device=table.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=table.device) if oob == "error" else None
    check(_abi.lib().dr_gather_f32(_ptr(table), table.shape[0], k, _ptr(ids), n, bits, _ptr(out),
                                   OOB_ERROR if oob == "error" else OOB_ZERO, _ptr(cnt), _stream()), "dr_gather_f32")
    if oob == "error":
        bad = int(cnt.item())
        if bad:
            raise IndexError("gather: %d ids out of range [0,%d)" % (bad, table.shape[0]))
    return out


def scale_rows(src, coef, idx=None):
    """out[e] = coef[e] * src[idx[e]] (idx None = e)."""
    src = _req(src, torch.float32, "src", 2)
    coef = _req(coef, torch.float32, "coef", 1)
    n = coef.numel()
    if idx is not None and _i32(idx, "idx").numel() != n:
        raise ValueError("idx must have one entry per coef")
    if idx is None and src.shape[0] < n:
        raise ValueError("src has fewer rows than coef entries")
    out = torch.empty((n, src.shape[1]), dtype=torch.float32, device=src.device)
    check(_abi.lib().dr_scale_rows_f32(_ptr(src), src.shape[0], src.shape[1], _ptr(idx), _ptr(coef), n, _ptr(out),
                                       _stream()), "dr_scale_rows_f32")
    return out


# ------------------------------------------------------------------ K2
class SegmentPlan:
    """Stable sort of an id list + run boundaries, held in a device workspace."""

    def __init__(self, ids, rows, workspace=None, k_max=128):
        ids, bits = _ids(ids, "ids")
        init()
        self.n = ids.numel()
        self.rows = int(rows)
        self.k_max = int(k_max)
        need = check(_abi.lib().dr_segment_plan_bytes(self.n, self.k_max), "dr_segment_plan_bytes")
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=ids.device)
        if workspace.numel() < need or workspace.data_ptr() % 256:
            raise ValueError("plan workspace too small / misaligned (%d < %d)" % (workspace.numel(), need))
        self.ws = workspace
        check(_abi.lib().dr_segment_plan(_ptr(ids), self.n, bits, self.rows, _ptr(self.ws), self.ws.numel(),
                                         _stream()), "dr_segment_plan")

    def view(self):
        """(sorted_pos[n], uniq_ids[nseg], seg_start[nseg+1]) as host-synchronised torch tensors (debug/tests)."""
        ptrs = [C.c_void_p() for _ in range(4)]
        check(_abi.lib().dr_segment_plan_view(_ptr(self.ws), self.n, *[C.byref(p) for p in ptrs]),
              "dr_segment_plan_view")
        base = self.ws.data_ptr()
        as_i32 = self.ws.view(torch.int32)

        def sl(p, count):
            o = (p.value - base) // 4
            return as_i32[o:o + count]

        nseg = int(sl(ptrs[3], 1).item())
        return sl(ptrs[0], self.n).clone(), sl(ptrs[1], nseg).clone(), sl(ptrs[2], nseg + 1).clone()


def sample_negatives(train_ptr, train_idx, users, num_item, seed, offset=0, n_neg=1, first_item=0):
    """n_neg uniform draws per entry of `users` from the items the user has not interacted with
    (bprmf.py:126-133,175-181), from the train CSR (int32 ptr[U+1], idx ascending per user).  int32 [n, n_neg];
    -1 where a user has no candidate."""
    train_ptr = _i32(train_ptr, "train_ptr")
    train_idx = _i32(train_idx, "train_idx")
    users = _i32(users, "users")
    n = users.numel()
    out = torch.empty((n, n_neg), dtype=torch.int32, device=users.device)
    check(_abi.lib().dr_sample_negatives_i32(_ptr(train_ptr), _ptr(train_idx), _ptr(users), n, int(n_neg),
                                             train_ptr.numel() - 1, int(num_item), int(first_item), int(seed),
                                             int(offset), _ptr(out), _stream()), "dr_sample_negatives_i32")
    return out


def sample_negatives_excl(targets, num_item, seed, offset=0, n_neg=1, first_item=1):
    """n_neg uniform draws per row of `targets` (int32 [n, T]) from the items first_item .. num_item-1 that are not
    among the row's own targets (AttRec.py:448-466).  int32 [n, n_neg]."""
    targets = _req(targets, torch.int32, "targets", 2)
    n, T = targets.shape
    out = torch.empty((n, n_neg), dtype=torch.int32, device=targets.device)
    check(_abi.lib().dr_sample_negatives_excl_i32(_ptr(targets), n, T, int(n_neg), int(num_item), int(first_item),
                                                  int(seed), int(offset), _ptr(out), _stream()),
          "dr_sample_negatives_excl_i32")
    return out


def bucket_order(keys, nkeys):
    """Stable bucketing of small integer keys in [0, nkeys) (routing by owner rank, deeprec_amd/sharding.py):
    returns (order int32 [n]: positions in ascending key order, ties in input order; counts int64 [nkeys]).
    No host synchronisation."""
    keys = _i32(keys, "keys")
    n = keys.numel()
    order = torch.empty(n, dtype=torch.int32, device=keys.device)
    counts = torch.empty(nkeys, dtype=torch.int64, device=keys.device)
    need = check(_abi.lib().dr_bucket_order_bytes(n), "dr_bucket_order_bytes")
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=keys.device)
    check(_abi.lib().dr_bucket_order_i32(_ptr(keys), n, int(nkeys), _ptr(order), _ptr(counts), _ptr(ws), ws.numel(),
                                         _stream()), "dr_bucket_order_i32")
    return order, counts


def gather_into(table, ids, out):
    """gather() into a caller-provided buffer (out [n, k] float32, contiguous): no allocation."""
    table = _req(table, torch.float32, "table", 2)
    ids = _i32(ids, "ids")
    n = ids.numel()
    if _req(out, torch.float32, "out", 2).shape != (n, table.shape[1]):
        raise ValueError("out must be [%d, %d]" % (n, table.shape[1]))
    check(_abi.lib().dr_gather_f32(_ptr(table), table.shape[0], table.shape[1], _ptr(ids), n, 32, _ptr(out), OOB_ZERO,
                                   None, _stream()), "dr_gather_f32")
    return out


# ------------------------------------------------------------------ one route per step (csrc/wire.hip)
ROUTE_MAX_LISTS, WIRE_MAX_SEG = 16, 64


def _i64_array(values):
    return (C.c_int64 * len(values))(*[int(v) for v in values])


def step_route_keys(ids_all, me, list_start):
    """dr_step_route_keys_i32: ids_all int32 [world, n_tot] (every rank's concatenated id lists), list_start = host
    offsets of the lists inside one rank's part (len n_lists + 1).  Returns (keys_mine [n_tot], keys_owned
    [world * n_tot], local_rows [world * n_tot]) for bucket_order(..., n_lists * world)."""
    ids_all = _req(ids_all, torch.int32, "ids_all", 2)
    W, n_tot = ids_all.shape
    nl = len(list_start) - 1
    dev = ids_all.device
    km = torch.empty(n_tot, dtype=torch.int32, device=dev)
    ko = torch.empty(W * n_tot, dtype=torch.int32, device=dev)
    loc = torch.empty(W * n_tot, dtype=torch.int32, device=dev)
    check(_abi.lib().dr_step_route_keys_i32(_ptr(ids_all), W, int(me), n_tot, _i64_array(list_start), nl, _ptr(km),
                                            _ptr(ko), _ptr(loc), _stream()), "dr_step_route_keys_i32")
    return km, ko, loc


def _wire_args(src_or_dst, idx, n, seg_start, seg_off, flat, name):
    t = _req(src_or_dst, torch.float32, name)
    if t.dim() not in (1, 2):
        raise ValueError("%s must be [rows] or [rows, k]" % name)
    flat = _req(flat, torch.float32, "flat", 1)
    if idx is not None and _i32(idx, "idx").numel() != n:
        raise ValueError("idx must have one entry per row moved")
    if len(seg_start) != len(seg_off) + 1:
        raise ValueError("seg_start needs one more entry than seg_off")
    return t, (1 if t.dim() == 1 else t.shape[1]), flat


def rows_pack(src, idx, n, seg_start, seg_off, flat, idx_bias=0):
    """dr_rows_pack_f32: rows src[idx[e] - idx_bias] (idx None: src[e]), e < n, into the segments of the wire buffer."""
    src, k, flat = _wire_args(src, idx, n, seg_start, seg_off, flat, "src")
    check(_abi.lib().dr_rows_pack_f32(_ptr(src), src.shape[0], k, _ptr(idx), int(idx_bias), int(n), _i64_array(seg_start),
                                      _i64_array(seg_off), len(seg_off), _ptr(flat), flat.numel(), _stream()),
          "dr_rows_pack_f32")


def rows_unpack(flat, idx, n, seg_start, seg_off, dst, idx_bias=0):
    """dr_rows_unpack_f32: the wire buffer's segments into dst[idx[e] - idx_bias] (idx None: dst[e])."""
    dst, k, flat = _wire_args(dst, idx, n, seg_start, seg_off, flat, "dst")
    check(_abi.lib().dr_rows_unpack_f32(_ptr(flat), flat.numel(), k, _ptr(idx), int(idx_bias), int(n),
                                        _i64_array(seg_start), _i64_array(seg_off), len(seg_off), _ptr(dst), dst.shape[0],
                                        _stream()), "dr_rows_unpack_f32")


# ------------------------------------------------------------------ row-sharded BPR step (csrc/shard.hip)
def shard_caps(world, B):
    """Default capacities of a rank: samples it may process / item entries it may own per step.  The mean loads are B
    and 2 B; Zipf-popular items make the owner of item 0 carry ~1.5 x the mean at world 8."""
    return min(world * B, max(2 * B, B + 4096)), min(2 * world * B, max(4 * B, 2 * B + 8192))


class ShardWorkspace:
    """Device workspace of ONE routed batch of the row-sharded step (route lists, plan, stash, losses); `counts` and
    `serve` are views into it (include/deeprec_amd.h: dr_shard_bpr_list)."""

    def __init__(self, world, me, B, k, num_user, num_item, device, cap_s=None, cap_g=None):
        d_s, d_g = shard_caps(world, B)
        self.world, self.me, self.B, self.k = int(world), int(me), int(B), int(k)
        self.num_user, self.num_item = int(num_user), int(num_item)
        self.cap_s, self.cap_g = int(cap_s or d_s), int(cap_g or d_g)
        need = check(_abi.lib().dr_shard_bpr_bytes(self.world, self.B, self.k, self.cap_s, self.cap_g), "dr_shard_bpr_bytes")
        self.buf = torch.empty(need, dtype=torch.uint8, device=device)
        self.counts = self.list(0)
        self.serve = self.list(1)
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)

    def _sizes(self):
        return (self.world, self.me, self.B, self.num_user, self.num_item, self.k, self.cap_s, self.cap_g)

    def list(self, which):
        """View of one list inside the workspace (dr_shard_bpr_list): 0 counts, 1 serve rows, 2 combined keys, 3 entry
        sources, 4 item row references [2, cap_s], 5 list / wire positions [2, cap_s], 6 solo flags [3, cap_s] (uint8),
        7 coefficients (float32)."""
        p = C.c_void_p()
        check(_abi.lib().dr_shard_bpr_list(self.world, self.B, self.k, self.cap_s, self.cap_g, _ptr(self.buf), int(which),
                                           C.byref(p)), "dr_shard_bpr_list")
        off = p.value - self.buf.data_ptr()
        ck = self.cap_s + self.cap_g
        if which == 6:
            return self.buf[off:off + 3 * self.cap_s].view(3, self.cap_s)
        n = {0: 8 + 6 * self.world, 1: self.cap_g, 2: ck, 3: ck, 4: 2 * self.cap_s, 5: 2 * self.cap_s, 7: ck}[which]
        v = self.buf[off:off + 4 * n].view(torch.float32 if which == 7 else torch.int32)
        return v.view(2, self.cap_s) if which in (4, 5) else v


def shard_route(ws, ids=None, uij=None, counts_host=None, stream=None):
    """dr_shard_bpr_route: ids = [world, 3, B] int32 (every rank's u, i, j), or at world 1 uij = (u, i, j) as they are.
    counts_host: pinned int32 tensor [8 + 6 world] that receives the counts asynchronously.  stream: raw handle of the
    stream to enqueue on (default: torch's current one).  No host synchronisation."""
    pu = pi = pj = None
    if ids is not None:
        ids = _req(ids, torch.int32, "ids", 3)
        if tuple(ids.shape) != (ws.world, 3, ws.B):
            raise ValueError("ids must be [%d, 3, %d]" % (ws.world, ws.B))
    else:
        if ws.world != 1 or uij is None:
            raise ValueError("shard_route: ids [world, 3, B] is required at world > 1")
        pu, pi, pj = (_i32(t, n) for t, n in zip(uij, "uij"))
        if pu.numel() != ws.B or pi.numel() != ws.B or pj.numel() != ws.B:
            raise ValueError("u, i, j must have %d entries" % ws.B)
    if counts_host is not None and (counts_host.dtype != torch.int32 or counts_host.numel() < 8 + 6 * ws.world
                                    or counts_host.is_cuda or not counts_host.is_pinned()):
        raise ValueError("counts_host must be a pinned int32 tensor of %d entries" % (8 + 6 * ws.world))
    check(_abi.lib().dr_shard_bpr_route(_ptr(ids), _ptr(pu), _ptr(pi), _ptr(pj), ws.world, ws.me, ws.B, ws.num_user,
                                        ws.num_item, ws.k, ws.cap_s, ws.cap_g, _ptr(ws.buf), ws.buf.numel(),
                                        C.c_void_p(counts_host.data_ptr()) if counts_host is not None else None,
                                        C.c_void_p(stream) if stream is not None else _stream()), "dr_shard_bpr_route")


def _shard_tables(ws, P, Q, slotsP, slotsQ, opt, who):
    P = _req(P, torch.float32, "P", 2)
    Q = _req(Q, torch.float32, "Q", 2)
    nuL = (ws.num_user - ws.me + ws.world - 1) // ws.world
    niL = (ws.num_item - ws.me + ws.world - 1) // ws.world
    if tuple(P.shape) != (nuL, ws.k) or tuple(Q.shape) != (niL, ws.k):
        raise ValueError("%s: P / Q must be this rank's shards [%d, %d] / [%d, %d]" % (who, nuL, ws.k, niL, ws.k))
    _slots(opt, slotsP[0], slotsP[1], P, who + "(P)")
    _slots(opt, slotsQ[0], slotsQ[1], Q, who + "(Q)")


def shard_forward(ws, P, Q, slotsP, slotsQ, opt, fetched=None, send=None, checked=False):
    """dr_shard_bpr_forward on the batch routed into ws.  fetched [>= rows fetched, k] / send [>= rows sent, k]: the
    exchange buffers (None at world 1).  checked: the caller validated these very tensors on an earlier call."""
    if not checked:
        _shard_tables(ws, P, Q, slotsP, slotsQ, opt, "shard_forward")
        for name, t in (("fetched", fetched), ("send", send)):
            if t is not None and _req(t, torch.float32, name, 2).shape[1] != ws.k:
                raise ValueError("%s must have %d columns" % (name, ws.k))
        if ws.world > 1 and (fetched is None or send is None or send.shape[0] < 2 * ws.cap_s
                             or fetched.shape[0] < 2 * ws.cap_s):
            raise ValueError("shard_forward: fetched / send need 2 * cap_s = %d rows" % (2 * ws.cap_s))
    check(_abi.lib().dr_shard_bpr_forward(_ptr(P), _ptr(slotsP[0]), _ptr(slotsP[1]), _ptr(Q), _ptr(slotsQ[0]),
                                          _ptr(slotsQ[1]), *ws._sizes(), _ptr(fetched), _ptr(send), C.byref(opt),
                                          _ptr(ws.buf), ws.buf.numel(), _stream()), "dr_shard_bpr_forward")


def shard_finish(ws, P, Q, slotsP, slotsQ, opt, received=None, checked=False):
    """dr_shard_bpr_finish; returns the device scalar loss sum of the samples this rank processed."""
    if not checked:
        _shard_tables(ws, P, Q, slotsP, slotsQ, opt, "shard_finish")
        if ws.world > 1 and (received is None or _req(received, torch.float32, "received", 2).shape[1] != ws.k
                             or received.shape[0] < ws.cap_g):
            raise ValueError("shard_finish: received needs cap_g = %d rows of %d columns" % (ws.cap_g, ws.k))
    check(_abi.lib().dr_shard_bpr_finish(_ptr(P), _ptr(slotsP[0]), _ptr(slotsP[1]), _ptr(Q), _ptr(slotsQ[0]),
                                         _ptr(slotsQ[1]), *ws._sizes(), _ptr(received), C.byref(opt), _ptr(ws.loss),
                                         _ptr(ws.buf), ws.buf.numel(), _stream()), "dr_shard_bpr_finish")
    return ws.loss


def scatter_add(gtable, ids, src, src_idx=None, coef=None, plan=None):
    """gtable[ids[e]] += coef[e] * src[src_idx[e]] in place, duplicates summed deterministically."""
    gtable = _req(gtable, torch.float32, "gtable")
    src = _req(src, torch.float32, "src")
    k = 1 if gtable.dim() == 1 else gtable.shape[1]
    n = ids.numel()
    if src.numel() % k:
        raise ValueError("src does not hold rows of %d floats" % k)
    src_rows = src.numel() // k
    if src_idx is None and src_rows < n:
        raise ValueError("src holds %d rows < %d ids" % (src_rows, n))
    if src_idx is not None and _i32(src_idx, "src_idx").numel() != n:
        raise ValueError("src_idx must have one entry per id")
    if coef is not None and _req(coef, torch.float32, "coef").numel() != n:
        raise ValueError("coef must have one entry per id")
    if plan is None:
        plan = SegmentPlan(ids, gtable.shape[0], k_max=k)
    if plan.n != n or plan.rows != gtable.shape[0]:
        raise ValueError("plan was built for a different id list / table")
    check(_abi.lib().dr_scatter_add_sorted_f32(_ptr(gtable), gtable.shape[0], k, _ptr(src), src_rows, _ptr(src_idx),
                                               _ptr(coef), n, _ptr(plan.ws), plan.ws.numel(), _stream()),
          "dr_scatter_add_sorted_f32")
    return gtable


def _slots(opt, slot0, slot1, like, name):
    if opt.kind != OPT_SGD:
        if slot0 is None or _req(slot0, torch.float32, name + ".slot0").shape != like.shape:
            raise ValueError("%s: optimizer kind %d needs slot0 shaped like the table" % (name, opt.kind))
    if opt.kind in (OPT_ADAM, OPT_RMSPROP):
        if slot1 is None or _req(slot1, torch.float32, name + ".slot1").shape != like.shape:
            raise ValueError("%s: optimizer kind %d needs slot1 shaped like the table" % (name, opt.kind))


def sparse_row_update(table, slot0, slot1, ids, src, opt, src_idx=None, coef=None, plan=None):
    table = _req(table, torch.float32, "table")
    src = _req(src, torch.float32, "src")
    _slots(opt, slot0, slot1, table, "sparse_row_update")
    k = 1 if table.dim() == 1 else table.shape[1]
    n = ids.numel()
    src_rows = src.numel() // k
    if src_idx is None and src_rows < n:
        raise ValueError("src holds %d rows < %d ids" % (src_rows, n))
    if src_idx is not None:
        _i32(src_idx, "src_idx")
    if coef is not None:
        _req(coef, torch.float32, "coef")
    if plan is None:
        plan = SegmentPlan(ids, table.shape[0], k_max=k)
    check(_abi.lib().dr_sparse_row_update_f32(_ptr(table), _ptr(slot0), _ptr(slot1), table.shape[0], k, _ptr(src),
                                              src_rows, _ptr(src_idx), _ptr(coef), n, _ptr(plan.ws), plan.ws.numel(),
                                              C.byref(opt), _stream()), "dr_sparse_row_update_f32")
    return table


def dense_update(w, slot0, slot1, grad, opt):
    w = _req(w, torch.float32, "w")
    _slots(opt, slot0, slot1, w, "dense_update")
    if grad is not None and _req(grad, torch.float32, "grad").shape != w.shape:
        raise ValueError("grad must be shaped like w")
    check(_abi.lib().dr_dense_update_f32(_ptr(w), _ptr(slot0), _ptr(slot1), _ptr(grad), w.numel(), C.byref(opt),
                                         _stream()), "dr_dense_update_f32")
    return w


def dense_update_masked(table, slot0, slot1, skip, opt):
    """dr_dense_update_masked_f32: the dense-L2 optimizer pass over the rows with skip[row] == 0 (no gradient but reg*w)."""
    table = _req(table, torch.float32, "table")
    _slots(opt, slot0, slot1, table, "dense_update_masked")
    rows = table.shape[0]
    k = 1 if table.dim() == 1 else table.shape[1]
    if _req(skip, torch.uint8, "skip", 1).numel() != rows:
        raise ValueError("skip must hold one byte per table row")
    check(_abi.lib().dr_dense_update_masked_f32(_ptr(table), _ptr(slot0), _ptr(slot1), rows, k, _ptr(skip), C.byref(opt),
                                                _stream()), "dr_dense_update_masked_f32")
    return table


def mark_rows(flags, ids, value):
    """flags[ids] = value (uint8 [rows]; ids int32, out-of-range ids ignored)."""
    flags = _req(flags, torch.uint8, "flags", 1)
    ids = _i32(ids, "ids")
    check(_abi.lib().dr_mark_rows_u8(_ptr(flags), flags.numel(), _ptr(ids), ids.numel(), int(value), _stream()),
          "dr_mark_rows_u8")
    return flags


def rownorm_clip(table, clip=1.0):
    table = _req(table, torch.float32, "table", 2)
    check(_abi.lib().dr_rownorm_clip_f32(_ptr(table), table.shape[0], table.shape[1], float(clip), _stream()),
          "dr_rownorm_clip_f32")
    return table


# ------------------------------------------------------------------ fused steps
class StepWorkspace:
    """Scratch of the fused training steps, sized once for (B, k, tables, reg_mode)."""

    def __init__(self, B, k, num_user, num_item, reg_mode, device):
        self.key = (int(B), int(k), int(num_user), int(num_item), int(reg_mode))
        init(device)
        need = check(_abi.lib().dr_pairwise_step_bytes(B, k, num_user, num_item, reg_mode), "dr_pairwise_step_bytes")
        self.buf = torch.empty(need, dtype=torch.uint8, device=device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)


def _step_ws(ws, B, k, nu, ni, opt, device):
    if ws is None:
        ws = StepWorkspace(B, k, nu, ni, opt.reg_mode, device)
    if ws.key != (int(B), int(k), int(nu), int(ni), int(opt.reg_mode)):
        raise ValueError("StepWorkspace was sized for %s" % (ws.key,))
    return ws


def _check_pq(P, Q, u, ids):
    P = _req(P, torch.float32, "P", 2)
    Q = _req(Q, torch.float32, "Q", 2)
    if P.shape[1] != Q.shape[1]:
        raise ValueError("P and Q must share k")
    B = u.numel()
    for name, t in ids:
        if _i32(t, name).numel() != B:
            raise ValueError("%s must have %d entries" % (name, B))
    return B


def bpr_plan(num_user, num_item, k, u, i, j, ws, stream=None):
    """Build the id lists and segment plans of batch (u, i, j) into ws (for bpr_step(..., planned=True)).
    stream: raw handle of the stream to enqueue on (default: torch's current one)."""
    B = u.numel()
    for name, t in (("u", u), ("i", i), ("j", j)):
        if _i32(t, name).numel() != B:
            raise ValueError("%s must have %d entries" % (name, B))
    if ws.key[:4] != (B, int(k), int(num_user), int(num_item)):
        raise ValueError("StepWorkspace was sized for %s" % (ws.key,))
    check(_abi.lib().dr_bpr_plan_f32(num_user, num_item, k, _ptr(u), _ptr(i), _ptr(j), B, ws.key[4], _ptr(ws.buf),
                                     ws.buf.numel(), C.c_void_p(stream) if stream is not None else _stream()),
          "dr_bpr_plan_f32")


class PlannedBprStep:
    """One (tables, batch, workspace) combination of the planned fused step with its arguments validated ONCE: a step
    loop over resident batches then pays one ctypes call per plan and per step (bench.py; the wrappers above re-check
    shapes and dtypes on every call, ~15 us of a 150 us step)."""

    def __init__(self, P, Q, slotsP, slotsQ, u, i, j, ws):
        B = _check_pq(P, Q, u, (("u", u), ("i", i), ("j", j)))
        for sl, t, nm in ((slotsP, P, "P"), (slotsQ, Q, "Q")):
            for x in sl:
                if x is not None and _req(x, torch.float32, "slot of " + nm).shape != t.shape:
                    raise ValueError("optimizer slots must be shaped like their table")
        if ws.key[:4] != (B, P.shape[1], P.shape[0], Q.shape[0]):
            raise ValueError("StepWorkspace was sized for %s" % (ws.key,))
        self.keep = (P, Q, slotsP, slotsQ, u, i, j, ws)
        self.ws = ws
        lib = _abi.lib()
        self._plan, self._step = lib.dr_bpr_plan_f32, lib.dr_bpr_step_planned_f32
        self._pa = (P.shape[0], Q.shape[0], P.shape[1], _ptr(u), _ptr(i), _ptr(j), B, ws.key[4], _ptr(ws.buf), ws.buf.numel())
        self._sa = (_ptr(P), _ptr(slotsP[0]), _ptr(slotsP[1]), P.shape[0], _ptr(Q), _ptr(slotsQ[0]), _ptr(slotsQ[1]),
                    Q.shape[0], P.shape[1], _ptr(u), _ptr(i), _ptr(j), B)
        self._tail = (_ptr(ws.loss), _ptr(ws.buf), ws.buf.numel())

    def plan(self, stream):
        """dr_bpr_plan_f32 on the raw stream handle `stream`."""
        check(self._plan(*self._pa, C.c_void_p(stream)), "dr_bpr_plan_f32")

    def step(self, opt, stream):
        """dr_bpr_step_planned_f32 on the raw stream handle `stream`; returns the device scalar loss."""
        if opt.reg_mode != self.ws.key[4]:
            raise ValueError("StepWorkspace was sized for reg_mode %d" % self.ws.key[4])
        check(self._step(*self._sa, C.byref(opt), *self._tail, C.c_void_p(stream)), "dr_bpr_step_planned_f32")
        return self.ws.loss


def bpr_step(P, Q, slotsP, slotsQ, u, i, j, opt, ws=None, planned=False):
    """One BPRMF.train_op (bprmf.py:114-123) in place; returns the device scalar sum_b -log sigmoid(x_b).
    planned=True: ws already holds bpr_plan() of the same batch."""
    B = _check_pq(P, Q, u, (("u", u), ("i", i), ("j", j)))
    _slots(opt, slotsP[0], slotsP[1], P, "bpr_step(P)")
    _slots(opt, slotsQ[0], slotsQ[1], Q, "bpr_step(Q)")
    ws = _step_ws(ws, B, P.shape[1], P.shape[0], Q.shape[0], opt, P.device)
    fn = _abi.lib().dr_bpr_step_planned_f32 if planned else _abi.lib().dr_bpr_step_f32
    check(fn(_ptr(P), _ptr(slotsP[0]), _ptr(slotsP[1]), P.shape[0], _ptr(Q), _ptr(slotsQ[0]),
                                     _ptr(slotsQ[1]), Q.shape[0], P.shape[1], _ptr(u), _ptr(i), _ptr(j), B,
                                     C.byref(opt), _ptr(ws.loss), _ptr(ws.buf), ws.buf.numel(), _stream()),
          "dr_bpr_step_f32")
    return ws.loss


STATIC_MAX_B, STATIC_MAX_K = 1024, 128


def bpr_step_static(P, Q, slotsP, slotsQ, u, i, j, opt, loss_out):
    """dr_bpr_step_static_f32: bpr_step() whose workspace is the library's own file-scope device array -- the call
    allocates nothing (loss_out: caller's float32 [1] tensor).  B <= 1024, k <= 128, no DR_REG_DENSE; calls that
    overlap on one GPU race (include/deeprec_amd.h)."""
    B = _check_pq(P, Q, u, (("u", u), ("i", i), ("j", j)))
    init(P.device)
    _slots(opt, slotsP[0], slotsP[1], P, "bpr_step_static(P)")
    _slots(opt, slotsQ[0], slotsQ[1], Q, "bpr_step_static(Q)")
    if _req(loss_out, torch.float32, "loss_out").numel() < 1:
        raise ValueError("loss_out must hold one float")
    check(_abi.lib().dr_bpr_step_static_f32(_ptr(P), _ptr(slotsP[0]), _ptr(slotsP[1]), P.shape[0], _ptr(Q),
                                            _ptr(slotsQ[0]), _ptr(slotsQ[1]), Q.shape[0], P.shape[1], _ptr(u), _ptr(i),
                                            _ptr(j), B, C.byref(opt), _ptr(loss_out), _stream()),
          "dr_bpr_step_static_f32")
    return loss_out


def bpr_forward(P, Q, u, i, j):
    B = _check_pq(P, Q, u, (("u", u), ("i", i), ("j", j)))
    g = torch.empty(B, dtype=torch.float32, device=P.device)
    loss_b = torch.empty_like(g)
    stash = torch.empty((B, P.shape[1]), dtype=torch.float32, device=P.device)
    check(_abi.lib().dr_bpr_forward_f32(_ptr(P), _ptr(Q), P.shape[0], Q.shape[0], P.shape[1], _ptr(u), _ptr(i),
                                        _ptr(j), B, _ptr(g), _ptr(loss_b), _ptr(stash), _stream()),
          "dr_bpr_forward_f32")
    return g, loss_b, stash


def pointwise_forward(P, Q, u, i, loss_kind, y=None, BU=None, BI=None, mu=0.0, want_grad=False):
    """GMF / MF forward: returns pred[B] (and g, loss_b, stash when want_grad)."""
    B = _check_pq(P, Q, u, (("u", u), ("i", i)))
    if loss_kind == LOSS_SQ:
        if _req(BU, torch.float32, "BU", 1).shape[0] != P.shape[0] or _req(BI, torch.float32, "BI", 1).shape[0] != Q.shape[0]:
            raise ValueError("bias vectors must match the tables")
    pred = torch.empty(B, dtype=torch.float32, device=P.device)
    g = loss_b = stash = None
    if want_grad:
        if y is None or _req(y, torch.float32, "y").numel() != B:
            raise ValueError("labels y[B] are required for gradients")
        g = torch.empty_like(pred)
        loss_b = torch.empty_like(pred)
        stash = torch.empty((B, P.shape[1]), dtype=torch.float32, device=P.device)
    check(_abi.lib().dr_pointwise_forward_f32(_ptr(P), _ptr(Q), _ptr(BU), _ptr(BI), float(mu), P.shape[0], Q.shape[0],
                                              P.shape[1], loss_kind, _ptr(u), _ptr(i), _ptr(y), B, _ptr(g),
                                              _ptr(loss_b), _ptr(pred), _ptr(stash), _stream()),
          "dr_pointwise_forward_f32")
    return (pred, g, loss_b, stash) if want_grad else pred


def pointwise_step(P, Q, slotsP, slotsQ, u, i, y, loss_kind, opt, BU=None, BI=None, slotsBU=(None, None),
                   slotsBI=(None, None), mu=0.0, ws=None):
    B = _check_pq(P, Q, u, (("u", u), ("i", i)))
    if _req(y, torch.float32, "y").numel() != B:
        raise ValueError("y must have %d entries" % B)
    _slots(opt, slotsP[0], slotsP[1], P, "pointwise_step(P)")
    _slots(opt, slotsQ[0], slotsQ[1], Q, "pointwise_step(Q)")
    if loss_kind == LOSS_SQ:
        if _req(BU, torch.float32, "BU", 1).shape[0] != P.shape[0] or _req(BI, torch.float32, "BI", 1).shape[0] != Q.shape[0]:
            raise ValueError("bias vectors must match the tables")
        _slots(opt, slotsBU[0], slotsBU[1], BU, "pointwise_step(BU)")
        _slots(opt, slotsBI[0], slotsBI[1], BI, "pointwise_step(BI)")
    ws = _step_ws(ws, B, P.shape[1], P.shape[0], Q.shape[0], opt, P.device)
    check(_abi.lib().dr_pointwise_step_f32(
        _ptr(P), _ptr(slotsP[0]), _ptr(slotsP[1]), P.shape[0], _ptr(Q), _ptr(slotsQ[0]), _ptr(slotsQ[1]), Q.shape[0],
        P.shape[1], _ptr(BU), _ptr(slotsBU[0]), _ptr(slotsBU[1]), _ptr(BI), _ptr(slotsBI[0]), _ptr(slotsBI[1]),
        float(mu), loss_kind, _ptr(u), _ptr(i), _ptr(y), B, C.byref(opt), _ptr(ws.loss), _ptr(ws.buf),
        ws.buf.numel(), _stream()), "dr_pointwise_step_f32")
    return ws.loss


def hinge_step(P, Q, accP, accQ, u, i, j, opt, keep=1.0, margin=0.5, clip=1.0, mask_pos=None, mask_neg=None, ws=None):
    """One CML step (cml.py:72-81,120) in place."""
    B = _check_pq(P, Q, u, (("u", u), ("i", i), ("j", j)))
    _slots(opt, accP, None, P, "hinge_step(P)")
    _slots(opt, accQ, None, Q, "hinge_step(Q)")
    for name, m in (("mask_pos", mask_pos), ("mask_neg", mask_neg)):
        if m is not None and tuple(_req(m, torch.uint8, name).shape) != (B, P.shape[1]):
            raise ValueError("%s must be uint8 [B,k]" % name)
    ws = _step_ws(ws, B, P.shape[1], P.shape[0], Q.shape[0], opt, P.device)
    check(_abi.lib().dr_hinge_step_f32(_ptr(P), _ptr(accP), P.shape[0], _ptr(Q), _ptr(accQ), Q.shape[0], P.shape[1],
                                       _ptr(u), _ptr(i), _ptr(j), _ptr(mask_pos), _ptr(mask_neg), float(keep),
                                       float(margin), float(clip if clip is not None else 0.0), B, C.byref(opt),
                                       _ptr(ws.loss), _ptr(ws.buf), ws.buf.numel(), _stream()), "dr_hinge_step_f32")
    return ws.loss


def pair_dot(A, B_, a_idx, b_idx):
    """k-ordered fmaf chain A[a_idx[n]] . B[b_idx[n]]  (BPRMF.predict, bprmf.py:168-173)."""
    A = _req(A, torch.float32, "A", 2)
    B_ = _req(B_, torch.float32, "B", 2)
    if A.shape[1] != B_.shape[1]:
        raise ValueError("A and B must share k")
    n = a_idx.numel()
    if _i32(a_idx, "a_idx").numel() != _i32(b_idx, "b_idx").numel():
        raise ValueError("index lists differ in length")
    out = torch.empty(n, dtype=torch.float32, device=A.device)
    check(_abi.lib().dr_pair_dot_f32(_ptr(A), A.shape[0], _ptr(B_), B_.shape[0], A.shape[1], _ptr(a_idx), _ptr(b_idx),
                                     n, _ptr(out), _stream()), "dr_pair_dot_f32")
    return out


# ------------------------------------------------------------------ NFM
def bi_interaction_fwd(E, feats):
    E = _req(E, torch.float32, "E", 2)
    feats = _req(feats, torch.int32, "feats", 2)
    B, F = feats.shape
    fm = torch.empty((B, E.shape[1]), dtype=torch.float32, device=E.device)
    s = torch.empty_like(fm)
    check(_abi.lib().dr_bi_interaction_fwd_f32(_ptr(E), E.shape[0], E.shape[1], _ptr(feats), B, F, _ptr(fm), _ptr(s),
                                               _stream()), "dr_bi_interaction_fwd_f32")
    return fm, s


def bi_interaction_bwd(E, feats, gfm, s):
    E = _req(E, torch.float32, "E", 2)
    feats = _req(feats, torch.int32, "feats", 2)
    B, F = feats.shape
    if tuple(_req(gfm, torch.float32, "gfm").shape) != (B, E.shape[1]) or tuple(_req(s, torch.float32, "sum").shape) != (B, E.shape[1]):
        raise ValueError("gfm / sum must be [B,k]")
    gv = torch.empty((B, F, E.shape[1]), dtype=torch.float32, device=E.device)
    check(_abi.lib().dr_bi_interaction_bwd_f32(_ptr(E), E.shape[0], E.shape[1], _ptr(feats), B, F, _ptr(gfm), _ptr(s),
                                               _ptr(gv), _stream()), "dr_bi_interaction_bwd_f32")
    return gv


# ------------------------------------------------------------------ dense towers
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
MLP_MAX_LAYERS, MLP_MAX_DIM = 8, 256


def _mlp_lists(x, Ws, bs, acts_kinds):
    x = _req(x, torch.float32, "x", 2)
    L = len(Ws)
    if not (1 <= L <= MLP_MAX_LAYERS) or len(acts_kinds) != L or (bs is not None and len(bs) != L):
        raise ValueError("mlp: 1..%d layers with one activation (and bias) each" % MLP_MAX_LAYERS)
    dims = [x.shape[1]]
    for l, W in enumerate(Ws):
        W = _req(W, torch.float32, "W[%d]" % l, 2)
        if W.shape[0] != dims[-1]:
            raise ValueError("mlp: W[%d] has %d rows, the layer input has %d columns" % (l, W.shape[0], dims[-1]))
        if bs is not None and bs[l] is not None and _req(bs[l], torch.float32, "b[%d]" % l, 1).numel() != W.shape[1]:
            raise ValueError("mlp: b[%d] does not match W[%d]" % (l, l))
        dims.append(W.shape[1])
    if max(dims) > MLP_MAX_DIM:
        raise ValueError("mlp: layer widths above %d are not supported" % MLP_MAX_DIM)
    ptrs = lambda ts: (C.c_void_p * L)(*[t.data_ptr() if t is not None else None for t in ts])
    return x, L, dims, (C.c_int32 * (L + 1))(*dims), (C.c_int32 * L)(*[int(a) for a in acts_kinds]), ptrs


def mlp_forward(x, Ws, bs, acts_kinds):
    """Whole tower in one kernel: returns acts [B, sum(dims[1:])] (every layer's output; the result is the last
    Ws[-1].shape[1] columns)."""
    x, L, dims, cdims, cact, ptrs = _mlp_lists(x, Ws, bs, acts_kinds)
    acts = torch.empty((x.shape[0], sum(dims[1:])), dtype=torch.float32, device=x.device)
    check(_abi.lib().dr_mlp_forward_f32(_ptr(x), x.shape[0], L, cdims, cact, ptrs(Ws), ptrs(bs) if bs is not None else None,
                                        _ptr(acts), _stream()), "dr_mlp_forward_f32")
    return acts


def mlp_backward(x, Ws, acts_kinds, acts, gout, want_gx=True, want_gb=True):
    """Returns (gx or None, [gW_l], [gb_l] or None) for dLoss/dh_L = gout."""
    x, L, dims, cdims, cact, ptrs = _mlp_lists(x, Ws, None, acts_kinds)
    gout = _req(gout, torch.float32, "gout", 2)
    if tuple(gout.shape) != (x.shape[0], dims[-1]) or tuple(_req(acts, torch.float32, "acts", 2).shape) != (x.shape[0], sum(dims[1:])):
        raise ValueError("mlp_backward: gout / acts do not match the tower")
    deltas = torch.empty_like(acts)
    gx = torch.empty_like(x) if want_gx else None
    gW = [torch.empty_like(W) for W in Ws]
    gb = [torch.empty(W.shape[1], dtype=torch.float32, device=x.device) for W in Ws] if want_gb else None
    check(_abi.lib().dr_mlp_backward_f32(_ptr(x), x.shape[0], L, cdims, cact, ptrs(Ws), _ptr(acts), _ptr(gout),
                                         _ptr(deltas), _ptr(gx), ptrs(gW), ptrs(gb) if gb is not None else None,
                                         _stream()), "dr_mlp_backward_f32")
    return gx, gW, gb


# 16-wide tile counts (first width, then the four layers) dr_tower_catalog_scores_f32 is instantiated for
TOWER_SHAPES = {(8, 5, 3, 2, 1), (4, 5, 3, 2, 1), (2, 3, 2, 1, 1)}


def tower_catalog_supported(dims):
    """dims = [h1, n2, n3, n4, n5]: does dr_tower_catalog_scores_f32 have this tower?"""
    return (len(dims) == 5 and dims[0] % 16 == 0 and all(1 <= d <= 128 for d in dims)
            and tuple((d + 15) // 16 for d in dims) in TOWER_SHAPES)


def tower_catalog_scores(uh, ih, Ws, bs, out):
    """NeuMF's tower for every (user, item) pair (neumf.py:80-115): out[u, i] += sum relu-tower(relu(uh[u] + ih[i])).
    uh [nu, h1], ih [ni, h1]: the two halves of the first dense layer (bias in uh); Ws / bs: the remaining four
    layers; out [nu, ni] float32 (row stride >= ni) arrives holding the GMF dot and is updated in place."""
    uh = _req(uh, torch.float32, "uh", 2)
    ih = _req(ih, torch.float32, "ih", 2)
    if len(Ws) != 4 or len(bs) != 4:
        raise ValueError("tower_catalog_scores: four layers after the split first one")
    Ws = [_req(W, torch.float32, "W", 2) for W in Ws]
    bs = [_req(b, torch.float32, "b", 1) for b in bs]
    dims = [uh.shape[1]] + [W.shape[1] for W in Ws]
    if ih.shape[1] != dims[0] or any(W.shape[0] != dims[l] or bs[l].shape[0] != dims[l + 1] for l, W in enumerate(Ws)):
        raise ValueError("tower_catalog_scores: layer shapes do not chain")
    if out.dtype != torch.float32 or out.dim() != 2 or tuple(out.shape) != (uh.shape[0], ih.shape[0]) or out.stride(1) != 1:
        raise ValueError("tower_catalog_scores: out must be float32 [nu, ni] with unit column stride")
    _on_current_device(out, "out")
    ptrs = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    check(_abi.lib().dr_tower_catalog_scores_f32(_ptr(uh), uh.shape[0], _ptr(ih), ih.shape[0], (C.c_int32 * 5)(*dims),
                                                 ptrs(Ws), ptrs(bs), _ptr(out), out.stride(0), _stream()),
          "dr_tower_catalog_scores_f32")
    return out


# ------------------------------------------------------------------ AttRec self-attention pooling
ATT_MAX_L, ATT_MAX_D = 64, 256


def attn_pool_forward(pre, v):
    """AttRec.py:147-186 after the projection: pre, v [B, L, d] -> (out [B, d], s [B, L, L] = the unmasked softmax)."""
    pre = _req(pre, torch.float32, "pre", 3)
    v = _req(v, torch.float32, "v", 3)
    if v.shape != pre.shape:
        raise ValueError("pre and v must have the same [B, L, d] shape")
    B, L, d = pre.shape
    out = torch.empty((B, d), dtype=torch.float32, device=pre.device)
    s = torch.empty((B, L, L), dtype=torch.float32, device=pre.device)
    check(_abi.lib().dr_attn_pool_forward_f32(_ptr(pre), _ptr(v), B, L, d, _ptr(out), _ptr(s), _stream()),
          "dr_attn_pool_forward_f32")
    return out, s


def attn_pool_backward(pre, v, s, gout):
    """-> (dpre [B, L, d], dv [B, L, d]: the direct path into the values only)."""
    pre = _req(pre, torch.float32, "pre", 3)
    B, L, d = pre.shape
    if _req(v, torch.float32, "v", 3).shape != pre.shape or _req(s, torch.float32, "s", 3).shape != (B, L, L) \
            or _req(gout, torch.float32, "gout", 2).shape != (B, d):
        raise ValueError("attn_pool_backward: v [B,L,d], s [B,L,L], gout [B,d] expected")
    dpre = torch.empty_like(pre)
    dv = torch.empty_like(pre)
    check(_abi.lib().dr_attn_pool_backward_f32(_ptr(pre), _ptr(v), _ptr(s), _ptr(gout), B, L, d, _ptr(dpre), _ptr(dv),
                                               _stream()), "dr_attn_pool_backward_f32")
    return dpre, dv


# ------------------------------------------------------------------ Caser horizontal convolutions
HC_MAX_L, HC_MAX_D, HC_MAX_C = 64, 256, 64
HCONV_CHUNK_COLS = int(__import__("os").environ.get("DR_HCONV_CHUNK", "4096"))   # columns of G per library GEMM


def hconv_rows(C, h):
    """First row of height h's filters in the shared filter matrix (= rows of all shorter heights)."""
    return C * h * (h - 1) // 2


def caser_hconv_forward(e, W, bias, C, chunk_cols=None, direct=None):
    """Caser.py:132-154: e [B, L, d], W [C L (L+1) / 2, d], bias [C L] -> (out [B, C L], arg int32 [B, C L]).
    direct (default: whenever the shape allows, C <= 16 and d % 16 == 0): dr_caser_hconv_direct_f32, the windowed products
    on MFMA straight from e and W.  Otherwise the products G = e W^T are library GEMMs over column chunks of whole
    heights (<= chunk_cols columns each) followed by dr_caser_hconv_pool_f32."""
    e = _req(e, torch.float32, "e", 3)
    W = _req(W, torch.float32, "W", 2)
    bias = _req(bias, torch.float32, "bias", 1)
    B, L, d = e.shape
    if W.shape != (hconv_rows(C, L + 1), d) or bias.numel() != C * L:
        raise ValueError("caser_hconv_forward: W must be [C L (L+1) / 2, d] and bias [C L]")
    out = torch.empty((B, C * L), dtype=torch.float32, device=e.device)
    arg = torch.empty((B, C * L), dtype=torch.int32, device=e.device)
    can = C <= 16 and d % 16 == 0 and L <= HC_MAX_L and d <= HC_MAX_D
    if direct is None:
        direct = can and chunk_cols is None
    if chunk_cols is None:
        chunk_cols = HCONV_CHUNK_COLS
    if direct:
        check(_abi.lib().dr_caser_hconv_direct_f32(_ptr(e), _ptr(W), _ptr(bias), B, L, C, d, _ptr(out), _ptr(arg),
                                                   _stream()), "dr_caser_hconv_direct_f32")
        return out, arg
    e2 = e.reshape(B * L, d)
    h0 = 1
    while h0 <= L:
        h1 = h0
        while h1 < L and hconv_rows(C, h1 + 2) - hconv_rows(C, h0) <= chunk_cols:
            h1 += 1
        r0, r1 = hconv_rows(C, h0), hconv_rows(C, h1 + 1)
        G = e2 @ W[r0:r1].t()
        check(_abi.lib().dr_caser_hconv_pool_f32(_ptr(G), r1 - r0, _ptr(bias), B, L, C, h0, h1, _ptr(out), _ptr(arg),
                                                 _stream()), "dr_caser_hconv_pool_f32")
        h0 = h1 + 1
    return out, arg


def caser_hconv_backward(e, W, arg, gout, C):
    """-> (ge [B, L, d], gW like W, gbias [C L]) from gout [B, C L] and the forward's arg."""
    e = _req(e, torch.float32, "e", 3)
    W = _req(W, torch.float32, "W", 2)
    arg = _i32(arg, "arg")
    gout = _req(gout, torch.float32, "gout", 2)
    B, L, d = e.shape
    if gout.shape != (B, C * L) or arg.shape != (B, C * L) or W.shape != (hconv_rows(C, L + 1), d):
        raise ValueError("caser_hconv_backward: gout / arg must be [B, C L], W [C L (L+1) / 2, d]")
    ge = torch.empty_like(e)
    check(_abi.lib().dr_caser_hconv_backward_input_f32(_ptr(gout), _ptr(arg), _ptr(W), B, L, C, d, _ptr(ge), _stream()),
          "dr_caser_hconv_backward_input_f32")
    S = max(1, min(32, B // 128))      # batch chunks of the filter gradient (each one's rows stay in L2)
    partial = torch.empty((S,) + tuple(W.shape), dtype=torch.float32, device=e.device)
    check(_abi.lib().dr_caser_hconv_backward_filter_f32(_ptr(gout), _ptr(arg), _ptr(e), B, L, C, d, S, _ptr(partial),
                                                        _stream()), "dr_caser_hconv_backward_filter_f32")
    gW = partial[0] if S == 1 else partial.sum(0)
    gbias = (gout * (arg >= 0)).sum(0)
    return ge, gW, gbias


VC_MAX_NV = 16


def caser_vconv_forward(e, w, bias):
    """Caser.py:118-130: e [B, L, d], w [n_v, L], bias [n_v] -> out [B, d * n_v] (column col * n_v + v)."""
    e = _req(e, torch.float32, "e", 3)
    w = _req(w, torch.float32, "w", 2)
    bias = _req(bias, torch.float32, "bias", 1)
    B, L, d = e.shape
    nv = w.shape[0]
    if w.shape[1] != L or bias.numel() != nv:
        raise ValueError("caser_vconv_forward: w must be [n_v, L], bias [n_v]")
    out = torch.empty((B, d * nv), dtype=torch.float32, device=e.device)
    check(_abi.lib().dr_caser_vconv_forward_f32(_ptr(e), _ptr(w), _ptr(bias), B, L, d, nv, _ptr(out), _stream()),
          "dr_caser_vconv_forward_f32")
    return out


def caser_vconv_backward(e, w, g):
    """-> (ge [B, L, d], gw [n_v, L], gbias [n_v]) from g [B, d * n_v]."""
    e = _req(e, torch.float32, "e", 3)
    w = _req(w, torch.float32, "w", 2)
    g = _req(g, torch.float32, "g", 2)
    B, L, d = e.shape
    nv = w.shape[0]
    if g.shape != (B, d * nv) or w.shape[1] != L:
        raise ValueError("caser_vconv_backward: g must be [B, d * n_v], w [n_v, L]")
    S = max(1, min(1024, (B * d + 255) // 256))
    ge = torch.empty_like(e)
    partial = torch.empty((S, nv, L + 1), dtype=torch.float32, device=e.device)
    check(_abi.lib().dr_caser_vconv_backward_f32(_ptr(e), _ptr(w), _ptr(g), B, L, d, nv, S, _ptr(ge), _ptr(partial),
                                                 _stream()), "dr_caser_vconv_backward_f32")
    tot = partial.sum(0)
    return ge, tot[:, :L].contiguous(), tot[:, L].contiguous()


# ------------------------------------------------------------------ sequence models: interaction + loss
SEQ_LOSS_HINGE, SEQ_LOSS_BPR = 0, 1
SEQ_MAX_SLOTS = 64


def seq_dist_loss(P, X, Q, users, o, tgt, neg, alpha, beta, loss_kind, margin=0.5):
    """dr_seq_dist_loss_f32 -> (loss_b [B], gX [B,2T,k], gQ [B,2T,k], gp [B,k], go [B,k])."""
    P, X, Q = _req(P, torch.float32, "P", 2), _req(X, torch.float32, "X", 2), _req(Q, torch.float32, "Q", 2)
    o = _req(o, torch.float32, "o", 2)
    users = _i32(users, "users")
    tgt, neg = _req(tgt, torch.int32, "tgt", 2), _req(neg, torch.int32, "neg", 2)
    B, k = o.shape
    T = tgt.shape[1]
    if X.shape != Q.shape or X.shape[1] != k or P.shape[1] != k or users.numel() != B or tgt.shape != (B, T) \
            or neg.shape != (B, T):
        raise ValueError("seq_dist_loss: P [U,k], X / Q [I,k], users [B], o [B,k], tgt / neg [B,T] expected")
    dev = o.device
    loss_b = torch.empty(B, dtype=torch.float32, device=dev)
    gX = torch.empty((B, 2 * T, k), dtype=torch.float32, device=dev)
    gQ = torch.empty_like(gX)
    gp = torch.empty((B, k), dtype=torch.float32, device=dev)
    go = torch.empty_like(gp)
    check(_abi.lib().dr_seq_dist_loss_f32(_ptr(P), P.shape[0], _ptr(X), _ptr(Q), X.shape[0], k, _ptr(users), _ptr(o),
                                          _ptr(tgt), _ptr(neg), B, T, float(alpha), float(beta), int(loss_kind),
                                          float(margin), _ptr(loss_b), _ptr(gX), _ptr(gQ), _ptr(gp), _ptr(go), _stream()),
          "dr_seq_dist_loss_f32")
    return loss_b, gX, gQ, gp, go


def seq_dot_loss(W, bias, x, tgt, neg, scale=1.0):
    """dr_seq_dot_loss_f32 -> (loss_b [B], gW [B,T+Tn,D], gb [B,T+Tn], gx [B,D])."""
    W, bias, x = _req(W, torch.float32, "W", 2), _req(bias, torch.float32, "bias", 1), _req(x, torch.float32, "x", 2)
    tgt, neg = _req(tgt, torch.int32, "tgt", 2), _req(neg, torch.int32, "neg", 2)
    B, D = x.shape
    T, Tn = tgt.shape[1], neg.shape[1]
    if W.shape[1] != D or bias.numel() != W.shape[0] or tgt.shape[0] != B or neg.shape[0] != B:
        raise ValueError("seq_dot_loss: W [I,D], bias [I], x [B,D], tgt [B,T], neg [B,Tn] expected")
    dev = x.device
    loss_b = torch.empty(B, dtype=torch.float32, device=dev)
    gW = torch.empty((B, T + Tn, D), dtype=torch.float32, device=dev)
    gb = torch.empty((B, T + Tn), dtype=torch.float32, device=dev)
    gx = torch.empty((B, D), dtype=torch.float32, device=dev)
    check(_abi.lib().dr_seq_dot_loss_f32(_ptr(W), _ptr(bias), W.shape[0], D, _ptr(x), _ptr(tgt), _ptr(neg), B, T, Tn,
                                         float(scale), _ptr(loss_b), _ptr(gW), _ptr(gb), _ptr(gx), _stream()),
          "dr_seq_dot_loss_f32")
    return loss_b, gW, gb, gx


# ------------------------------------------------------------------ K6
def _csr(ptr, idx, Nu, name):
    if ptr is None:
        return None, None
    ptr = _i32(ptr, name + "_ptr")
    idx = _i32(idx, name + "_idx")
    if ptr.numel() != Nu + 1:
        raise ValueError("%s_ptr must have Nu+1 entries" % name)
    return ptr, idx


def score_topk(U, I, K, score_kind=SCORE_DOT, U2=None, I2=None, alpha=0.0, item_bias=None, train=None, targets=None):
    """All-items scoring + masked top-K + rank-of-target (RankingMetrics.py:82-101).
    train / targets: (ptr[Nu+1], idx[nnz]) int32 CSR (idx ascending per user) or None.
    Returns (topk_idx[Nu,K] int32, topk_score[Nu,K] f32, tgt_rank[nnz] int32 or None)."""
    U = _req(U, torch.float32, "U", 2)
    I = _req(I, torch.float32, "I", 2)
    Nu, k = U.shape
    Ni = I.shape[0]
    if I.shape[1] != k:
        raise ValueError("U and I must share k")
    if score_kind == SCORE_TWO_DIST:
        if tuple(_req(U2, torch.float32, "U2", 2).shape) != (Nu, k) or tuple(_req(I2, torch.float32, "I2", 2).shape) != (Ni, k):
            raise ValueError("U2 / I2 must match U / I")
    if item_bias is not None and _req(item_bias, torch.float32, "item_bias", 1).numel() != Ni:
        raise ValueError("item_bias must be [Ni]")
    tp, ti = _csr(*(train if train is not None else (None, None)), Nu, "train")
    gp, gi = _csr(*(targets if targets is not None else (None, None)), Nu, "tgt")
    dev = U.device
    topk_idx = torch.empty((Nu, K), dtype=torch.int32, device=dev)
    topk_score = torch.empty((Nu, K), dtype=torch.float32, device=dev)
    nt = gi.numel() if gi is not None else 0
    rank = torch.empty(nt, dtype=torch.int32, device=dev) if gi is not None else None
    need = check(_abi.lib().dr_score_topk_bytes(Nu, Ni, k, K, nt, score_kind), "dr_score_topk_bytes")
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    check(_abi.lib().dr_score_topk_f32(_ptr(U), _ptr(U2), Nu, _ptr(I), _ptr(I2), _ptr(item_bias), Ni, k, score_kind,
                                       float(alpha), _ptr(tp), _ptr(ti), _ptr(gp), _ptr(gi), nt, K, _ptr(topk_idx),
                                       _ptr(topk_score), _ptr(rank), _ptr(ws), ws.numel(), _stream()),
          "dr_score_topk_f32")
    return topk_idx, topk_score, rank


def score_rank(U, I, targets, score_kind=SCORE_DOT, U2=None, I2=None, alpha=0.0, item_bias=None, train=None):
    """dr_score_rank_f32: the ranks score_topk returns, without a top-K list (what the evaluators consume).
    targets / train: (ptr[Nu+1], idx[nnz]) int32 CSR.  Returns tgt_rank[nnz] int32 (-1: the held-out item is masked)."""
    U = _req(U, torch.float32, "U", 2)
    I = _req(I, torch.float32, "I", 2)
    Nu, k = U.shape
    Ni = I.shape[0]
    if I.shape[1] != k:
        raise ValueError("U and I must share k")
    if score_kind == SCORE_TWO_DIST:
        if tuple(_req(U2, torch.float32, "U2", 2).shape) != (Nu, k) or tuple(_req(I2, torch.float32, "I2", 2).shape) != (Ni, k):
            raise ValueError("U2 / I2 must match U / I")
    if item_bias is not None and _req(item_bias, torch.float32, "item_bias", 1).numel() != Ni:
        raise ValueError("item_bias must be [Ni]")
    tp, ti = _csr(*(train if train is not None else (None, None)), Nu, "train")
    gp, gi = _csr(*targets, Nu, "tgt")
    nt = gi.numel()
    rank = torch.zeros(nt, dtype=torch.int32, device=U.device)
    need = check(_abi.lib().dr_score_topk_bytes(Nu, Ni, k, 1, nt, score_kind), "dr_score_topk_bytes")
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=U.device)
    check(_abi.lib().dr_score_rank_f32(_ptr(U), _ptr(U2), Nu, _ptr(I), _ptr(I2), _ptr(item_bias), Ni, k, score_kind,
                                       float(alpha), _ptr(tp), _ptr(ti), _ptr(gp), _ptr(gi), nt, _ptr(rank), _ptr(ws),
                                       ws.numel(), _stream()), "dr_score_rank_f32")
    return rank


def score_rank_exact(U, I, targets, score_kind=SCORE_DOT, U2=None, I2=None, w1=0.0, w2=0.0, item_bias=None, train=None):
    """Ranks of the held-out items in the reference's numpy float32 operation order (dr_score_rank_exact_f32):
    MFMA pass + exact re-score of the near-ties.  targets / train: (ptr, idx) int32 CSR.  Returns int32 [nnz]."""
    U = _req(U, torch.float32, "U", 2)
    I = _req(I, torch.float32, "I", 2)
    Nu, k = U.shape
    Ni = I.shape[0]
    if I.shape[1] != k:
        raise ValueError("U and I must share k")
    if score_kind == SCORE_TWO_DIST:
        if tuple(_req(U2, torch.float32, "U2", 2).shape) != (Nu, k) or tuple(_req(I2, torch.float32, "I2", 2).shape) != (Ni, k):
            raise ValueError("U2 / I2 must match U / I")
    if item_bias is not None and _req(item_bias, torch.float32, "item_bias", 1).numel() != Ni:
        raise ValueError("item_bias must be [Ni]")
    tp, ti = _csr(*(train if train is not None else (None, None)), Nu, "train")
    gp, gi = _csr(*targets, Nu, "tgt")
    nt = gi.numel()
    rank = torch.empty(nt, dtype=torch.int32, device=U.device)
    need = check(_abi.lib().dr_score_rank_exact_bytes(Nu, Ni, k, nt, score_kind), "dr_score_rank_exact_bytes")
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=U.device)
    check(_abi.lib().dr_score_rank_exact_f32(_ptr(U), _ptr(U2), Nu, _ptr(I), _ptr(I2), _ptr(item_bias), Ni, k, score_kind,
                                             float(w1), float(w2), _ptr(tp), _ptr(ti), _ptr(gp), _ptr(gi), nt, _ptr(rank),
                                             _ptr(ws), ws.numel(), _stream()), "dr_score_rank_exact_f32")
    return rank


def topk_rank(scores, K, train=None, targets=None):
    """Same selection on a given score matrix [Nu,Ni] (NeuMF and other non-bilinear scores)."""
    scores = _req(scores, torch.float32, "scores", 2)
    Nu, Ni = scores.shape
    tp, ti = _csr(*(train if train is not None else (None, None)), Nu, "train")
    gp, gi = _csr(*(targets if targets is not None else (None, None)), Nu, "tgt")
    dev = scores.device
    topk_idx = torch.empty((Nu, K), dtype=torch.int32, device=dev)
    topk_score = torch.empty((Nu, K), dtype=torch.float32, device=dev)
    nt = gi.numel() if gi is not None else 0
    rank = torch.empty(nt, dtype=torch.int32, device=dev) if gi is not None else None
    ws = torch.empty(2 * ((nt * 4 + 255) // 256 * 256) + 256, dtype=torch.uint8, device=dev)
    check(_abi.lib().dr_topk_rank_f32(_ptr(scores), Nu, Ni, _ptr(tp), _ptr(ti), _ptr(gp), _ptr(gi), nt, K,
                                      _ptr(topk_idx), _ptr(topk_score), _ptr(rank), _ptr(ws), ws.numel(), _stream()),
          "dr_topk_rank_f32")
    return topk_idx, topk_score, rank


def score_matrix(U, I, score_kind=SCORE_DOT, U2=None, I2=None, alpha=0.0, item_bias=None, sigmoid=False):
    """Dense [Nu,Ni] score matrix through the same MFMA GEMM (GMF.predict over a catalog, small catalogs)."""
    U = _req(U, torch.float32, "U", 2)
    I = _req(I, torch.float32, "I", 2)
    if U.shape[1] != I.shape[1]:
        raise ValueError("U and I must share k")
    if score_kind == SCORE_TWO_DIST:
        if tuple(_req(U2, torch.float32, "U2", 2).shape) != tuple(U.shape) or tuple(_req(I2, torch.float32, "I2", 2).shape) != tuple(I.shape):
            raise ValueError("U2 / I2 must match U / I")
    if item_bias is not None and _req(item_bias, torch.float32, "item_bias", 1).numel() != I.shape[0]:
        raise ValueError("item_bias must be [Ni]")
    out = torch.empty((U.shape[0], I.shape[0]), dtype=torch.float32, device=U.device)
    need = check(_abi.lib().dr_score_topk_bytes(U.shape[0], I.shape[0], U.shape[1], 1, 0, score_kind),
                 "dr_score_topk_bytes")
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=U.device)
    check(_abi.lib().dr_score_matrix_f32(_ptr(U), _ptr(U2), U.shape[0], _ptr(I), _ptr(I2), _ptr(item_bias), I.shape[0],
                                         U.shape[1], score_kind, float(alpha), 1 if sigmoid else 0, _ptr(out),
                                         _ptr(ws), ws.numel(), _stream()), "dr_score_matrix_f32")
    return out


# ------------------------------------------------------------------ helpers
def init_normal_(w, seed, stddev, offset=0):
    w = _req(w, torch.float32, "w")
    check(_abi.lib().dr_init_normal_f32(_ptr(w), w.numel(), int(seed), float(stddev), int(offset), _stream()),
          "dr_init_normal_f32")
    return w


def sum_f32(x):
    x = _req(x, torch.float32, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(_abi.lib().dr_sum_f32(_ptr(x), x.numel(), _ptr(out), _stream()), "dr_sum_f32")
    return out
