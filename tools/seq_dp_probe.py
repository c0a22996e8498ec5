"""Where the data-parallel AttRec / Caser step spends its time at world 1 through RCCL (dev tool): wall per step, and
host wall inside route construction / fetch / push (monkeypatched timers).   python tools/seq_dp_probe.py [AttRec|Caser] [steps]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "AttRec"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    prefetch = os.environ.get("PREFETCH", "0") == "1"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from deeprec_amd import sharding as sh
    from deeprec_amd.models._common import Shard
    from deeprec_amd.models.seq_rec.AttRec import AttRec
    from deeprec_amd.models.seq_rec.Caser import Caser
    rng = np.random.default_rng(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.int32)
    nu, ni, L, T_, B, d = 1_000_000, 100_000, 50, 3, 4096, 128
    cls = AttRec if name == "AttRec" else Caser
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls(None, nu, ni, 0.05 if cls is AttRec else 0.001, 0.01, 1, B, shard=Shard(1, 0, force=True))
        m.build_network(L, T_, num_factor=d, num_neg=1)
    bs = [(t(rng.integers(0, nu, B)), t(rng.integers(1, ni, (B, L))), t(rng.integers(1, ni, (B, T_))),
           t(rng.integers(1, ni, (B, T_)))) for _ in range(4)]
    torch.cuda.synchronize()
    acc = {}

    def wrap(obj, attr, key):
        f = getattr(obj, attr)

        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, attr, g)
    wrap(sh.StepRoute, "__init__", "route")
    wrap(sh.StepRoute, "finish", "route_finish")
    wrap(sh.StepRoute, "fetch", "fetch")
    wrap(sh.StepRoute, "push", "push")
    wrap(m, "_apply_step", "apply")

    def run(n):
        for i in range(n):
            if prefetch:
                m.prefetch(*bs[(i + 1) % 4])
            m.train_op(*bs[i % 4])
        m.shard._pre = []
    run(5)
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    run(steps)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s world-1 RCCL%s: %.3f ms per step (host loop %.3f); host wall inside: %s" % (
        name, " prefetch" if prefetch else "", t_all / steps * 1e3, t_host / steps * 1e3,
        {k: round(v / steps * 1e3, 3) for k, v in acc.items()}))
    dist.destroy_process_group()


main()
