#!/usr/bin/env python
"""bench.py -- headline measurement of the DeepRec embedding/interaction hot path on MI355X.

Workload (BASELINE.json configs[3], SURVEY 8d "C4"): synthetic BPR-MF, 10M users x 1M items, d=128,
B=65536 (u, i, j) triples per GPU and step, users ~ Uniform, positive items ~ Zipf(1.05), negatives
~ Uniform; Adam with touched-rows L2 (the reference's optimizer, bprmf.py:62).  It fits one
GPU, so it is also the N=1 workload; at N>1 the tables are row-sharded (deeprec_amd/sharding.py) and
every rank processes its own B triples (weak scaling).

One "step" = one pass of the hot path over one batch: 3 lookups, 2 dot products, BPR loss,
gradient, deterministic duplicate sums, Adam row update (dr_bpr_step_f32).  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=25,
                    help="timed windows of --steps steps each, back to back; ms_per_step is the median window (the first "
                         "window is the contract's 'exactly K steps' and is reported too)")
    ap.add_argument("--users", type=int, default=10_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--opt", default="adam", choices=["adam", "sgd"])
    ap.add_argument("--eval-users", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel / eval / gather extras")
    ap.add_argument("--graph", action="store_true",
                    help="replay one hipGraph per resident batch instead of launching eagerly (the eager pipelined step is "
                         "GPU-bound: ~130 us of host time per 0.21 ms step, and graph replay measures ~8 %% slower)")
    ap.add_argument("--no-graph", action="store_true", help="(default; kept for compatibility)")
    ap.add_argument("--sharded", action="store_true",
                    help="run the row-sharded (multi-GPU) step even at --gpus 1, to price its exchange logic without a fabric")
    ap.add_argument("--no-stream-calibration", action="store_true",
                    help="skip the untimed trial that picks the main / side streams (see calibrate_streams)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="build each batch's segment plans inside its own step (dr_bpr_step_f32) instead of one batch ahead")
    return ap.parse_args()


def zipf_items(n, n_items, a, gen, device):
    """Zipf(a) over [0, n_items) by inverse-CDF on a float64 table (synthetic, seeded)."""
    w = torch.arange(1, n_items + 1, dtype=torch.float64, device=device).pow_(-a)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    r = torch.rand(n, dtype=torch.float64, device=device, generator=gen)
    return torch.searchsorted(cdf, r).clamp_(max=n_items - 1).to(torch.int32)


def make_batches(args, rank, device, n_batches=8):
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    out = []
    for _ in range(n_batches):
        u = torch.randint(0, args.users, (args.batch,), device=device, generator=gen, dtype=torch.int64).to(torch.int32)
        i = zipf_items(args.batch, args.items, 1.05, gen, device)
        j = torch.randint(0, args.items, (args.batch,), device=device, generator=gen, dtype=torch.int64).to(torch.int32)
        out.append((u.contiguous(), i.contiguous(), j.contiguous()))
    return out


def ev_time(fn, iters, stream):
    """Average milliseconds of fn() measured with HIP events on the stream the kernels run on."""
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(stream)
    for _ in range(iters):
        fn()
    e.record(stream)
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the product kernels")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.sharded:
        # --sharded on one GPU: the row-sharded trainer at world 1 (it exchanges nothing there: the fused step) with a
        # process group of size 1 up, as in a multi-GPU job
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from deeprec_amd import kernels as K
    from deeprec_amd import sharding as sh

    k, B = args.k, args.batch
    adam = args.opt == "adam"
    n_slots = 2 if adam else 0
    stream = torch.cuda.current_stream()

    # Adam's bias-corrected step size lives on the device (dr_adam_tick_f32, TF's beta-power variables), so a
    # captured step keeps the exact per-step arithmetic when it is replayed
    # (--graph: one fixed scalar, ticked by a launch inside the captured step; eager: the next 16 step sizes per launch)
    adam_state = K.AdamState(1e-3, 0.9, 0.999, device, ahead=1 if args.graph else 16) if adam else None

    def opt_for(step):
        b1p, b2p = 0.9 ** step, 0.999 ** step
        return K.opt_params(K.OPT_ADAM if adam else K.OPT_SGD, lr=0.05, alpha=1e-3 * (1 - b2p) ** 0.5 / (1 - b1p),
                            beta1=0.9, beta2=0.999, eps=1e-7, reg=0.1, reg_mode=K.REG_TOUCHED)

    opt_dev = K.opt_params(K.OPT_ADAM if adam else K.OPT_SGD, lr=0.05, beta1=0.9, beta2=0.999, eps=1e-7, reg=0.1,
                           reg_mode=K.REG_TOUCHED, alpha_dev=adam_state.alpha if adam else None)
    launch = "eager"

    batches = make_batches(args, rank, device)
    if world == 1 and not args.sharded:
        P = torch.empty((args.users, k), dtype=torch.float32, device=device)
        Q = torch.empty((args.items, k), dtype=torch.float32, device=device)
        K.init_normal_(P, 1, 0.01)
        K.init_normal_(Q, 2, 0.01)
        sP = [torch.zeros_like(P) for _ in range(n_slots)] + [None, None]
        sQ = [torch.zeros_like(Q) for _ in range(n_slots)] + [None, None]
        nb = len(batches)
        wss = [K.StepWorkspace(B, k, args.users, args.items, K.REG_TOUCHED, device) for _ in range(nb)]
        ws = wss[0]
        pipelined = not args.no_pipeline
        # The segment plan of a batch depends on its ids only and is a chain of ~17 short dependent launches: beside the
        # bandwidth-saturating forward kernel that chain takes ~2x its stand-alone 100 us, longer than the step itself
        # (0.15 ms), so ONE plan in flight made the step plan-bound (tools/eager_probe.py: 0.222 ms).  DEPTH plans are
        # kept in flight instead, each on its own stream: batch b + DEPTH is planned while batch b runs.
        DEPTH = 1 if args.graph else 2
        sides = [torch.cuda.Stream(device=device) for _ in range(max(DEPTH, 1))]
        side = sides[0]

        def set_sides(new):
            sides[:] = new[:len(sides)]
        ev_end = [torch.cuda.Event() for _ in range(nb)]
        ev_plan = [torch.cuda.Event() for _ in range(nb)]
        started = {"end": set(), "plan": set()}

        calls = [K.PlannedBprStep(P, Q, sP, sQ, *batches[b], wss[b]) for b in range(nb)]   # arguments validated once
        cur_stream = {}

        def eager_step(b):
            """One step on resident batch b.  Pipelined: the segment plan of batch b + DEPTH (it depends on its ids only)
            is built on another stream while this step's kernels run; every step still does all of its own work -- one
            plan, one step -- inside the timed region."""
            u, i, j = batches[b]
            if adam:
                adam_state.tick()
                opt_dev.alpha_dev = adam_state.alpha.data_ptr()
            if not pipelined:
                return K.bpr_step(P, Q, sP, sQ, u, i, j, opt_dev, wss[b])
            if DEPTH == 1:
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    K.bpr_plan(args.users, args.items, k, *batches[(b + 1) % nb], wss[(b + 1) % nb])
                loss = K.bpr_step(P, Q, sP, sQ, u, i, j, opt_dev, wss[b], planned=True)
                cur.wait_stream(side)
                return loss
            # host path kept lean (a 0.15 ms step leaves no room for stream context managers: raw stream handles, events
            # recorded on explicit streams, arguments validated once)
            cur = cur_stream.get("s")
            if cur is None or cur_stream["id"] != torch._C._cuda_getCurrentRawStream(device.index):
                cur = cur_stream["s"] = torch.cuda.current_stream()
                cur_stream["id"] = cur.cuda_stream
            nxt = (b + DEPTH) % nb
            sd = sides[b % DEPTH]
            if (b - 1) % nb in started["end"]:
                sd.wait_event(ev_end[(b - 1) % nb])     # the workspace of batch `nxt` was last read by an earlier step
            calls[nxt].plan(sd.cuda_stream)
            ev_plan[nxt].record(sd)
            started["plan"].add(nxt)
            if b in started["plan"]:
                cur.wait_event(ev_plan[b])
            loss = calls[b].step(opt_dev, cur_stream["id"])
            ev_end[b].record(cur)
            started["end"].add(b)
            return loss

        if pipelined:
            for b in range(DEPTH):
                K.bpr_plan(args.users, args.items, k, *batches[b], wss[b])
            launch = ("eager, plans %d batches ahead on %d side streams" % (DEPTH, DEPTH)) if DEPTH > 1 else \
                "eager, plans one batch ahead on a second stream"
            launch += "; launches per step: plan 7 (3 x [histogram, ranked scatter] + finish) + step 3 (fused forward, tail, combine)"

        def step(t):
            return eager_step(t % nb)

        if args.graph and not args.no_graph:
            # the step is ~40 short launches: replay it as one hipGraph per resident batch (launch-bound otherwise)
            try:
                eager_step(0)                       # creates helper streams / events outside capture
                if pipelined:
                    K.bpr_plan(args.users, args.items, k, *batches[0], wss[0])
                torch.cuda.synchronize()
                graphs = []
                for b in range(nb):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        eager_step(b)
                    graphs.append(g)
                launch = "hipGraph replay, one graph per resident batch" + (
                    "; plans one batch ahead on a parallel branch" if pipelined else "")

                def step(t):                        # noqa: F811
                    graphs[t % nb].replay()
                    return wss[t % nb].loss
            except Exception as e:                  # capture not available: measure the eager path and say so
                launch = "eager (graph capture failed: %s)" % str(e).splitlines()[0][:120]
                torch.cuda.synchronize()
    else:
        Pt = sh.ShardedTable(args.users, k, world, rank, device, n_slots)
        Qt = sh.ShardedTable(args.items, k, world, rank, device, n_slots)
        Pt.init_normal_(K, 1, 0.01)
        Qt.init_normal_(K, 2, 0.01)

        # samples go to the owner of their user row, distinct item rows come in, item gradient rows go out; the
        # id-only routing of batch t+1 runs on a second stream / communicator while batch t's kernels execute
        trainer = sh.ShardedBpr(K, Pt, Qt)
        trainer.prefetch(*batches[0])
        trainer.prefetch(*batches[1])
        launch = ("eager; routes (id all-gather, the rank's lists and plan) two batches ahead on two side streams; launches per "
                  "step: route %d + step %d, collectives %d" % ((9, 3, 0) if world == 1 else (17, 4, 3)))

        def step(t):
            return trainer.step(opt_for(t + 1), next_batch=batches[(t + 2) % len(batches)])

        def set_sides(new):
            trainer.sides = list(new[:trainer.N_SIDE])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    done = 0
    stream_note = "default stream"
    if not args.no_stream_calibration and not (args.graph and not args.no_graph):
        done, stream_note = calibrate_streams(step, set_sides, device)
        stream = torch.cuda.current_stream()
    for t in range(args.warmup):
        step(done + t)
    done += args.warmup
    # R windows of exactly --steps steps, each bracketed by barrier + synchronize on both sides and reduced with MAX over
    # the ranks; the driver's --steps 20 is 3 ms of GPU time, one sample of a distribution -- the median window is the figure
    wins = []
    for _ in range(max(1, args.windows)):
        barrier()
        t0 = time.perf_counter()
        for t in range(args.steps):
            loss = step(done + t)
        barrier()
        dt = time.perf_counter() - t0
        done += args.steps
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        wins.append(dt)
    last_loss = float(loss.item())
    srt = sorted(wins)
    dt = srt[len(srt) // 2]
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    timed_s = sum(wins)

    result = {
        "metric": "train samples/s + top-10 eval users/s; embedding gather GB/s vs HBM peak",
        "value": value, "unit": "train samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C4 synthetic BPR-MF %dM users x %dM items d=%d, B=%d/GPU, %s touched-rows L2, "
                               "users~U, items~Zipf(1.05)" % (args.users // 10**6, args.items // 10**6, k, B,
                                                              "Adam" if adam else "SGD"),
                   "users": args.users, "items": args.items, "k": k, "batch_per_gpu": B,
                   "optimizer": args.opt, "reg_mode": "touched", "launch": launch,
                   "parallelism": "single" if world == 1 and not args.sharded else "row-sharded x%d (all-to-all)" % world},
        "last_loss_sum": last_loss,
        "windows": len(wins), "ms_per_step_first_window": wins[0] / args.steps * 1e3,
        "ms_per_step_min_max": [srt[0] / args.steps * 1e3, srt[-1] / args.steps * 1e3],
        "timed_seconds_total": timed_s,
        "timing": "median of %d back-to-back windows of %d steps, each between barrier + synchronize, MAX over ranks"
                  % (len(wins), args.steps),
        "streams": stream_note,
    }

    if (world > 1 or args.sharded) and not args.no_extras:
        ev = sharded_eval(args, K, sh, Pt, Qt, world, rank, device, barrier)
        if rank == 0:
            result["eval"] = ev
    if rank == 0 and world == 1 and not args.no_extras and not args.sharded:
        result.update(extras(args, K, P, Q, sP, sQ, batches, opt_for, stream, device, ms_per_step, eager_step))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def calibrate_streams(step, set_sides, device, pool_n=6, trial_steps=40):
    """Untimed trial before the warm-up: which streams the step and its side work (plans / routes of the next batches) run
    on.  HIP multiplexes a process's streams onto 4 hardware queues and places them when they are first used; which ones
    end up sharing a queue depends on how many streams torch, RCCL and this library created before, and the pipelined step
    is sensitive to it -- measured on one box with one tree: 0.152 / 0.169 / 0.180 ms per fused step and 0.157 / 0.200 ms
    per sharded step for different, equally legitimate choices (a direct test for queue sharing between two streams -- a
    long copy on one, a tiny kernel on the other -- did not predict which).  So a few (main, side, side) triples from a small pool are each run
    for `trial_steps` real steps and the fastest is kept.  Every rank tries the same triples in the same order (the steps
    contain collectives at world > 1).  Returns (steps consumed, description)."""
    pool = [torch.cuda.Stream(device=device) for _ in range(pool_n)]
    triples = [(0, 1, 2), (1, 2, 3), (2, 3, 4), (3, 4, 5), (0, 2, 4), (1, 3, 5), (4, 5, 0), (5, 0, 1)]
    t, best, log = 0, None, []
    for tr in triples:
        torch.cuda.synchronize()
        torch.cuda.set_stream(pool[tr[0]])
        set_sides([pool[tr[1]], pool[tr[2]]])
        for _ in range(8):
            step(t)
            t += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(trial_steps):
            step(t)
            t += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / trial_steps * 1e3
        log.append(round(ms, 4))
        if best is None or ms < best[0]:
            best = (ms, tr)
    pick = best[1]          # per process: every rank keeps its own fastest triple
    torch.cuda.synchronize()
    torch.cuda.set_stream(pool[pick[0]])
    set_sides([pool[pick[1]], pool[pick[2]]])
    return t, "main / side streams picked by an untimed trial of %d triples x %d steps (ms per step: %s; kept %s)" % (
        len(triples), trial_steps, log, list(pick))


def sharded_eval(args, K, sh, Pt, Qt, world, rank, device, barrier):
    """Top-10 evaluation with the users sharded data-parallel: every rank scores --eval-users of its own
    users against the whole catalog (item table replicated by one all-gather, user rows fetched from their owners)."""
    import torch.distributed as dist
    Nu, iters = args.eval_users, 3
    gen = torch.Generator(device=device)
    gen.manual_seed(7 + rank)
    users = torch.randint(0, args.users, (Nu,), device=device, generator=gen).to(torch.int32)
    tr = torch.sort(torch.randint(0, args.items, (Nu, 100), device=device, generator=gen), 1)[0].to(torch.int32)
    tp = (torch.arange(Nu + 1, device=device) * 100).to(torch.int32)
    tg = torch.randint(0, args.items, (Nu,), device=device, generator=gen).to(torch.int32)
    gp = torch.arange(Nu + 1, device=device).to(torch.int32)
    barrier()
    t0 = time.perf_counter()
    Qf = sh.all_gather_table(Qt)
    barrier()
    t_rep = time.perf_counter() - t0
    run = lambda: sh.sharded_eval_ranks(K, Pt, Qt, users, train=(tp, tr.reshape(-1).contiguous()), targets=(gp, tg), Q_full=Qf)
    run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        rk, _ = run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n, hr, ndcg, mrr = sh.reduce_rank_metrics(rk, 10)
    flops = 2.0 * world * Nu * iters * args.items * args.k
    return {"users_per_rank": Nu, "items": args.items, "K": 10, "ms_per_pass": dt / iters * 1e3,
            "users_per_s": world * Nu * iters / dt, "TFLOPs": flops / dt / 1e12,
            "frac_mfma_f32_peak": flops / dt / 1e12 / (MFMA_F32_PEAK_TFLOPS * world),
            "item_table_all_gather_ms": t_rep * 1e3, "users_ranked": n, "hr@10": hr, "ndcg@10": ndcg, "mrr": mrr,
            "path": "dr_score_rank_f32 (ranks of the held-out items; every @10 metric follows from them)",
            "parallelism": "users sharded x%d, item table replicated, user rows fetched from their owners" % world}


ROCPROF_NAMES = {   # live name -> kernel name in `rocprofv3 --kernel-trace --stats` output (k=128: V=4, 32 lanes/row)
    "bpr_forward_kernel": "dr::bpr_forward_kernel<4, 32>",
    "bpr_forward_update_kernel": "dr::bpr_forward_update_1col_kernel<4, 32, 1>",
    "bpr_tail_kernel": "dr::bpr_tail_kernel<4, 32, 1>",
    "bpr_tail_combine_kernel": "dr::bpr_tail_combine_kernel<4, 32, 1>",
    "lsd_hist_kernel": "dr::lsd_hist_kernel<...>", "lsd_scatter_kernel": "dr::lsd_scatter_kernel<...>",
    "bpr_finish_kernel": "dr::bpr_finish_kernel<dr::FinBpr>",
    "seg_update_kernel<item side>": "dr::seg_update_kernel<4, 32, 2, 1>",
    "seg_update_kernel<user side>": "dr::seg_update_kernel<4, 32, 1, 1>",
    "seg_long_partial_kernel": "dr::seg_long_partial_kernel<4, 32>",
    "seg_long_combine_kernel": "dr::seg_long_combine_kernel<4, 32>",
}
TRAFFIC_FILE = "r03_pmc_traffic.json"       # this round's PMC summary (tools/pmc_traffic.py)
LONG_SEG = 4        # deeprec_amd/csrc/segment.hip: segments with more contributions take the two-pass path
SEG_CHUNK = 32      # DR_SEG_CHUNK


def segment_census(ids):
    """(short rows, short contributions, long rows, long contributions, partial rows, solo rows) of one id list;
    short = 2..LONG_SEG contributions (the segment kernel's), solo = exactly one (the fused forward kernel's)."""
    cnt = torch.unique(ids, return_counts=True)[1]
    solo = cnt == 1
    short = (cnt <= LONG_SEG) & ~solo
    lc = cnt[cnt > LONG_SEG]
    return (int(short.sum()), int(cnt[short].sum()), int(lc.numel()), int(lc.sum()),
            int(((lc + SEG_CHUNK - 1) // SEG_CHUNK).sum()), int(solo.sum()))


def live_kernel_profile(K, run_step, n_steps):
    """Per-kernel durations of the step AS THE BENCH RUNS IT, from HIP events recorded by the library around each
    launch on the stream it is launched on (dr_profile_enable / dr_profile_collect)."""
    import ctypes as C
    from deeprec_amd import _abi
    lib = _abi.lib()
    for t in range(4):
        run_step(t)
    torch.cuda.synchronize()
    lib.dr_profile_enable(1)
    for t in range(n_steps):
        run_step(4 + t)
    torch.cuda.synchronize()
    lib.dr_profile_enable(0)
    cap = 4096
    names = C.create_string_buffer(1 << 18)
    ms = (C.c_float * cap)()
    tl = os.environ.get("DR_TIMELINE")          # file to write "start_us duration_us kernel" lines to (diagnostics)
    if tl:
        st_ms = (C.c_float * cap)()
        n = lib.dr_profile_collect_timeline(names, len(names), C.cast(ms, C.c_void_p), C.cast(st_ms, C.c_void_p), cap)
        if n > 0:
            with open(tl, "a") as f:
                for nm, a, d in zip(names.value.decode().split("\n")[:n], list(st_ms)[:n], list(ms)[:n]):
                    f.write("%10.1f %8.1f %s\n" % (a * 1e3, d * 1e3, nm))
    else:
        n = lib.dr_profile_collect(names, len(names), ms, cap)
    if n < 0:
        raise RuntimeError("dr_profile_collect failed")
    acc = {}
    for nm, t in zip(names.value.decode().split("\n")[:n], list(ms)[:n]):
        acc.setdefault(nm, []).append(t)
    return acc


def extras(args, K, P, Q, sP, sQ, batches, opt_for, stream, device, ms_per_step, eager_step):
    """Per-kernel rooflines (HIP events on the launch stream), standalone gather GB/s, eval users/s."""
    k, B = args.k, args.batch
    adam = args.opt == "adam"
    u, i, j = batches[0]
    rows_rw = 6 if adam else 2            # (w[,m,v]) read + written per updated row
    row = 4 * k
    out = {}
    # --- the kernels of the step, timed inside the running (pipelined, eager) step -------------------
    nb = len(batches)
    PROF_STEPS = 32 * nb                  # 256 launches of every kernel of the step
    prof = live_kernel_profile(K, lambda t: eager_step(t % nb), PROF_STEPS)
    cQ = [segment_census(torch.cat((bi, bj))) for _, bi, bj in batches]
    cP = [segment_census(bu) for bu, _, _ in batches]
    mean = lambda xs: sum(xs) / len(xs)
    # algorithmic bytes per launch (DESIGN.md "Kernels"): a destination row costs its (w, m, v) read + written and
    # 8 B of segment metadata; a contribution costs its source row and 12 B (source index, position, coefficient)
    upd = lambda c, short: (mean([x[0 if short else 2] for x in c]) * (rows_rw * row + 8)
                            + mean([x[1 if short else 3] for x in c]) * (row + 12))
    solo_rows = mean([x[5] for x in cQ]) + mean([x[5] for x in cP])
    long_part = lambda c: mean([x[3] for x in c]) * (row + 12) + mean([x[4] for x in c]) * row
    long_comb = lambda c: mean([x[4] for x in c]) * row + mean([x[2] for x in c]) * (rows_rw * row + 8)
    alg = {
        "bpr_forward_kernel": B * (12 + 3 * row + row + 16),            # ids, 3 rows; stash row, g, loss, 2 coefficients
        # fused forward, SURVEY 8(d) credit only: ids, every looked-up row read once, and for every single-contribution
        # row its (w, m, v) read + written (the first read of w is the lookup already counted).  The stash rows, g /
        # loss / coefficient writes and the solo flags are overhead, not credited.
        "bpr_forward_update_kernel": B * 12 + 3 * B * row + solo_rows * (rows_rw - 1) * row,
        "seg_update_kernel<item side>": upd(cQ, True),
        "seg_update_kernel<user side>": upd(cP, True),
        # long segments: pass 1 reads the contributions and writes one partial row per 32; pass 2 reads the partial
        # rows and reads + writes (w, m, v)
        "seg_long_partial_kernel": long_part(cQ),
        "seg_long_combine_kernel": long_comb(cQ),
        # the fused step's tail: rows with 2 .. 4 contributions and the long segments' block partials, both tables
        "bpr_tail_kernel": upd(cQ, True) + upd(cP, True) + long_part(cQ) + long_part(cP),
        "bpr_tail_combine_kernel": long_comb(cQ) + long_comb(cP),
    }
    # HBM-side bytes per launch: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, one pass each, corrected as
    # MI355X_MICROARCH.md prescribes) over tools/step_loop.py -- the same kernels on the same batches, collected in a
    # SEPARATE profiler run and committed under profiles/; not measured by this process
    # the CURRENT round's file or nothing: a stale profile must not dress up as this tree's traffic
    tp_path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not os.path.exists(tp_path):
        raise SystemExit("bench.py: profiles/%s is missing -- collect it with tools/run_round_checks.sh (two rocprofv3 --pmc "
                         "passes over tools/step_loop.py + tools/pmc_traffic.py) before the extras can report `traffic`"
                         % TRAFFIC_FILE)
    with open(tp_path) as f:
        traffic = json.load(f).get("kernels", {})
    traffic_src = "profiles/" + TRAFFIC_FILE
    kern = {}
    for nm, ts in prof.items():
        calls, t = len(ts), sum(ts) / len(ts)
        srt = sorted(ts)
        e = {"avg_ms": t, "min_ms": srt[0], "median_ms": srt[len(srt) // 2], "max_ms": srt[-1], "launches": calls,
             "calls_per_step": calls / PROF_STEPS, "rocprof_name": ROCPROF_NAMES.get(nm, nm)}
        if nm in alg and t > 0:
            e["algorithmic_bytes_per_launch"] = int(alg[nm])
            e["algorithmic_GBps"] = alg[nm] / (t * 1e-3) / 1e9
            e["frac_hbm_peak"] = e["algorithmic_GBps"] / HBM_PEAK_GBPS
            e["frac_hbm_peak_min_median_max"] = [alg[nm] / (x * 1e-3) / 1e9 / HBM_PEAK_GBPS
                                                 for x in (srt[-1], srt[len(srt) // 2], srt[0])]
        if nm in traffic:
            e["traffic_bytes_per_launch_from_profiles"] = traffic[nm]
        kern[nm] = e
    out["kernels"] = kern
    # dominant kernel = the one that moves the most bytes per step (the long-segment passes run beside it on a
    # helper stream and their durations are mostly dispatch waits behind its workgroups)
    name = max((n for n in kern if n in alg), key=lambda n: alg[n] * kern[n]["calls_per_step"])
    d = kern[name]
    out["roofline"] = {"bound": "hbm", "kernel": d["rocprof_name"], "achieved": d["algorithmic_GBps"],
                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": d["frac_hbm_peak"],
                       "frac_min_median_max": d["frac_hbm_peak_min_median_max"], "launches": d["launches"],
                       "traffic": (traffic.get(name) or {}).get("hbm_bytes"),
                       "traffic_source": (traffic_src + " (separate rocprofv3 --pmc passes, not this run)") if traffic_src else None,
                       "avg_ms": d["avg_ms"], "median_ms": d["median_ms"],
                       "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                       "algorithmic_bytes": "SURVEY 8(d): 12 B ids + 3 looked-up rows per sample + (w, m, v) read and written "
                                            "for every single-contribution row; stash / g / loss writes not credited",
                       "timing": "start/stop HIP events attached to each launch on its own stream (hipExtLaunchKernelGGL"
                                 " via dr_profile_enable), inside the pipelined eager step"}
    # standalone plan cost (generic API; the step itself uses one combined 3B-key plan built one batch ahead)
    idsQ = torch.cat((i, j)).contiguous()
    planQ = K.SegmentPlan(idsQ, args.items)
    out["kernels"]["segment_plan[2B keys, standalone API]"] = {
        "avg_ms": ev_time(lambda: K.SegmentPlan(idsQ, args.items, planQ.ws), 20, stream)}
    step_bytes = B * (12 + 3 * rows_rw * row)                          # 9228 B / 3084 B per sample at k=128
    out["step_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_sample": step_bytes // B,
                            "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    # --- K1 standalone: 3B lookups (u from P, i and j from Q) ---------------------------------
    # avg_ms: HIP events around back-to-back calls (launch gaps included); kernel_ms: the dispatch's own begin / end
    # (dr_profile_enable), which is what rocprofv3 --kernel-trace reports and what GBps / frac_hbm_peak are priced on
    gb = lambda n, t: n * (4 + 2 * row) / (t * 1e-3) / 1e9             # n*(4 + 2*4k) bytes

    def gather_entry(table, ids, iters):
        n = ids.numel()
        t_call = ev_time(lambda: K.gather(table, ids), iters, stream)
        ts = live_kernel_profile(K, lambda t: K.gather(table, ids), iters).get("gather_kernel", [t_call])
        t_k = sum(ts) / len(ts)
        return {"rows": n, "avg_ms": t_call, "kernel_ms": t_k, "GBps": gb(n, t_k), "frac_hbm_peak": gb(n, t_k) / HBM_PEAK_GBPS,
                "GBps_incl_launch_gaps": gb(n, t_call)}

    out["gather"] = {"users_uniform_10M": gather_entry(P, u, 20), "items_zipf+uniform_1M": gather_entry(Q, idsQ, 20)}
    big = torch.randint(0, args.users, (1 << 22,), device=device, dtype=torch.int64).to(torch.int32)
    out["gather"]["users_uniform_10M_4Mrows"] = gather_entry(P, big, 5)
    del big
    # --- K2 standalone: sorted-segment scatter-add of 4 M gradient rows into a 10 M-row table (uniform ids) ------
    if sP[1] is not None:
        n_sc = 1 << 22
        gen = torch.Generator(device=device)
        gen.manual_seed(13)
        sid = torch.randint(0, args.users, (n_sc,), device=device, generator=gen, dtype=torch.int64).to(torch.int32)
        src = torch.empty((n_sc, k), dtype=torch.float32, device=device)
        K.init_normal_(src, 3, 0.01)
        gt = sP[1]                                   # Adam's v of the user table: free to clobber after the timed region
        plan = K.SegmentPlan(sid, args.users, k_max=k)
        uq = int(plan.view()[1].numel())
        t_sc = ev_time(lambda: K.scatter_add(gt, sid, src, plan=plan), 5, stream)
        t_scp = ev_time(lambda: K.scatter_add(gt, sid, src, plan=K.SegmentPlan(sid, args.users, plan.ws, k_max=k)), 5, stream)
        b_sc = n_sc * (4 + row) + uq * 2 * row       # ids + grads read, unique rows read + written
        out["scatter_add"] = {"rows": n_sc, "unique_rows": uq, "avg_ms_planned": t_sc, "GBps_planned": b_sc / (t_sc * 1e-3) / 1e9,
                              "frac_hbm_peak_planned": b_sc / (t_sc * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "avg_ms_with_plan": t_scp, "GBps_with_plan": b_sc / (t_scp * 1e-3) / 1e9,
                              "frac_hbm_peak_with_plan": b_sc / (t_scp * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        del src, sid, plan
    # --- negative sampler: one draw per positive from all items minus 100 seen items / user
    ncsr = min(args.users, 1_000_000)
    gen = torch.Generator(device=device)
    gen.manual_seed(11)
    seen = torch.sort(torch.randint(0, args.items, (ncsr, 100), device=device, generator=gen), 1)[0].to(torch.int32)
    sptr = (torch.arange(ncsr + 1, device=device) * 100).to(torch.int32)
    su = torch.randint(0, ncsr, (B,), device=device, generator=gen).to(torch.int32)
    t_ns = ev_time(lambda: K.sample_negatives(sptr, seen.reshape(-1), su, args.items, 5), 20, stream)
    out["neg_sampler"] = {"draws": B, "seen_per_user": 100, "users_in_csr": ncsr, "avg_ms": t_ns,
                          "draws_per_s": B / (t_ns * 1e-3)}
    del seen, sptr, su
    # --- K6: top-10 eval over the full catalog, 100 masked train items / user, 1 held-out -----
    Nu = args.eval_users
    gen = torch.Generator(device=device)
    gen.manual_seed(7)
    users = torch.randint(0, args.users, (Nu,), device=device, generator=gen)
    Uemb = K.gather(P, users.to(torch.int32))
    tr = torch.sort(torch.randint(0, args.items, (Nu, 100), device=device, generator=gen), 1)[0].to(torch.int32)
    tp = (torch.arange(Nu + 1, device=device) * 100).to(torch.int32)
    tg = torch.randint(0, args.items, (Nu,), device=device, generator=gen).to(torch.int32)
    gp = torch.arange(Nu + 1, device=device).to(torch.int32)
    flops = 2.0 * Nu * args.items * k
    trf = tr.reshape(-1).contiguous()
    # the evaluation as RankingMetrics.evaluate consumes it: the ranks of the held-out items (dr_score_rank_f32; every
    # @5 / @10 / full-list metric of RankingMetrics.py:21-65 is a function of those ranks) ...
    t_rk = ev_time(lambda: K.score_rank(Uemb, Q, (gp, tg), train=(tp, trf)), 3, stream)
    out["eval"] = {"users": Nu, "items": args.items, "K": 10, "avg_ms": t_rk, "users_per_s": Nu / (t_rk * 1e-3),
                   "TFLOPs": flops / (t_rk * 1e-3) / 1e12, "frac_mfma_f32_peak": flops / (t_rk * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                   "path": "dr_score_rank_f32: ranks of the held-out items over the whole catalog, train items masked"}
    # ... and with the explicit top-10 list beside the ranks (dr_score_topk_f32: a service the reference does not have)
    t_ev = ev_time(lambda: K.score_topk(Uemb, Q, 10, train=(tp, trf), targets=(gp, tg)), 3, stream)
    out["eval"]["with_top10_list"] = {"avg_ms": t_ev, "users_per_s": Nu / (t_ev * 1e-3), "TFLOPs": flops / (t_ev * 1e-3) / 1e12,
                                      "frac_mfma_f32_peak": flops / (t_ev * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS}
    return out


def cpu_baseline(args):
    """The CPU port of the same step (oracle/c/oracle.c:or_bpr_step, scalar C, 1 thread) on a bounded sample."""
    from oracle import c_oracle as co
    k, B = args.k, args.batch
    Uc = min(args.users, 1_000_000)            # user table scaled down so the sample initialises in seconds
    Ic = min(args.items, 1_000_000)
    rng = np.random.default_rng(0)
    P = co.init_normal(Uc * k, 1, 0.01).reshape(Uc, k)
    Q = co.init_normal(Ic * k, 2, 0.01).reshape(Ic, k)
    adam = args.opt == "adam"
    sP = [np.zeros_like(P), np.zeros_like(P)] if adam else []
    sQ = [np.zeros_like(Q), np.zeros_like(Q)] if adam else []
    w = np.arange(1, Ic + 1, dtype=np.float64) ** -1.05
    cdf = np.cumsum(w) / w.sum()
    steps, t_total = 0, 0.0
    while steps < 3 or (t_total < 10.0 and steps < 200):
        u = rng.integers(0, Uc, B).astype(np.int32)
        i = np.minimum(np.searchsorted(cdf, rng.random(B)), Ic - 1).astype(np.int32)
        j = rng.integers(0, Ic, B).astype(np.int32)
        t0 = time.perf_counter()
        co.bpr_step(P, Q, sP, sQ, u, i, j, adam, lr=0.05, alpha=1e-3, eps=1e-7, reg=0.1)
        t_total += time.perf_counter() - t0
        steps += 1
    others = cpu_other_flavours(args, P, Q, cdf, rng)
    scalar = {"value": steps * B / t_total, "unit": "train samples/s", "cores": 1,
              "sample": "%d steps of B=%d, scalar C port oracle/c/oracle.c:or_bpr_step, ids resident" % (steps, B)}
    ref = others.pop("torch_cpu_reference_shaped")
    others["scalar_c_port"] = scalar
    # headline = the flavour shaped like the reference's own control flow: per-positive Python negative
    # draw and per-batch host feed (bprmf.py:126-150) around fp32 tensor math on the host cores (torch-CPU standing in
    # for TensorFlow, which is not installable here).  The vectorised and the scalar-C figures sit beside it.
    return {"value": ref["value"], "unit": "train samples/s", "cores": ref["cores"], "kind": "port",
            "other_flavours": others,
            "sample": ref["sample"] + "; %dM users x %dM items d=%d with the user table scaled to %d rows, %s touched-rows L2"
                      % (args.users // 10**6, args.items // 10**6, k, Uc, args.opt),
            "host_cpus": os.cpu_count()}


def cpu_other_flavours(args, P, Q, cdf, rng):
    """the same step as (i) vectorised torch-CPU math on all host threads and (ii) in the reference's
    control flow -- per-positive Python negative sampling (bprmf.py:130-133) and a per-batch host feed -- with
    torch-CPU tensor math standing in for TensorFlow's.  Bounded to a few seconds each; reported beside the scalar
    port, never as `value`."""
    k, B = args.k, args.batch
    Uc, Ic = P.shape[0], Q.shape[0]
    adam = args.opt == "adam"
    tP, tQ = torch.from_numpy(P), torch.from_numpy(Q)
    slots = {id(t): (torch.zeros_like(t), torch.zeros_like(t)) for t in (tP, tQ)} if adam else {}

    def apply(table, ids, grads, t):
        uq, inv = torch.unique(ids, return_inverse=True)
        gsum = torch.zeros((uq.numel(), k), dtype=torch.float32).index_add_(0, inv, grads)
        w = table[uq]
        gsum += 0.1 * w
        if adam:
            m, v = slots[id(table)]
            mu = m[uq].mul_(0.9).add_(gsum, alpha=0.1)
            vu = v[uq].mul_(0.999).addcmul_(gsum, gsum, value=0.001)
            m[uq], v[uq] = mu, vu
            alpha = 1e-3 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
            table[uq] = w - alpha * mu / (vu.sqrt() + 1e-7)
        else:
            table[uq] = w - 0.05 * gsum

    def torch_step(u, i, j, t):
        pu, qi, qj = tP[u], tQ[i], tQ[j]
        d = qi - qj
        g = -(1 - torch.sigmoid((pu * d).sum(1)))
        apply(tQ, torch.cat((i, j)), torch.cat((g[:, None] * pu, -g[:, None] * pu)), t)
        apply(tP, u, g[:, None] * d, t)

    def timed(make_batch, budget):
        steps, tt = 0, 0.0
        while steps < 2 or (tt < budget and steps < 100):
            t0 = time.perf_counter()
            u, i, j = make_batch()
            torch_step(u, i, j, steps + 1)
            tt += time.perf_counter() - t0
            steps += 1
        return steps, tt

    def resident_batch():       # inputs ready, as for the GPU number
        return (torch.from_numpy(rng.integers(0, Uc, B)), torch.from_numpy(np.minimum(np.searchsorted(cdf, rng.random(B)), Ic - 1)),
                torch.from_numpy(rng.integers(0, Ic, B)))

    pre = [resident_batch() for _ in range(4)]
    it = iter(range(10 ** 9))
    s1, t1 = timed(lambda: pre[next(it) % 4], 5.0)

    import random
    seen = {}                    # the reference's neg_items[u] is all items minus the seen ones; sampled by rejection here

    def reference_shaped_batch():
        u = rng.integers(0, Uc, B)
        i = np.minimum(np.searchsorted(cdf, rng.random(B)), Ic - 1)
        j = []
        for uu, ii in zip(u.tolist(), i.tolist()):       # one Python-level draw per positive (bprmf.py:130-133)
            s = seen.setdefault(uu, set())
            s.add(ii)
            x = random.randrange(Ic)
            while x in s:
                x = random.randrange(Ic)
            j.append(x)
        return torch.from_numpy(u), torch.from_numpy(i), torch.tensor(j)

    s2, t2 = timed(reference_shaped_batch, 5.0)
    return {"torch_cpu_vectorised": {"value": s1 * B / t1, "unit": "train samples/s", "cores": torch.get_num_threads(),
                                     "sample": "%d steps, index_select / index_add_ / Adam on the touched rows, ids resident" % s1},
            "torch_cpu_reference_shaped": {"value": s2 * B / t2, "unit": "train samples/s", "cores": torch.get_num_threads(),
                                           "sample": "%d steps, per-positive Python negative draw + per-batch host feed "
                                                     "(bprmf.py:126-150), torch-CPU math in TensorFlow's place" % s2}}


if __name__ == "__main__":
    main()
