"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one pass each) of tools/step_loop.py into
profiles/r03_pmc_traffic.json (or the name given as third argument): mean HBM-side bytes per launch for each library kernel.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $REPO/gpurun_out/pmc_fetch -o f --output-format csv -- python $REPO/tools/step_loop.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $REPO/gpurun_out/pmc_write -o w --output-format csv -- python $REPO/tools/step_loop.py
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB... the L2's memory-
side request counters; on gfx950 FETCH_SIZE tallies 128-byte requests of 16-byte-per-lane reads at 64 bytes, so it
is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Infinity-Cache hits are included in both."""
import csv
import glob
import json
import os
import sys

SHORT = {
    "bpr_forward_update_1col_kernel": "bpr_forward_update_kernel",
    "bpr_forward_kernel": "bpr_forward_kernel",
    "seg_update_kernel<4, 32, 2, 1>": "seg_update_kernel<item side>",
    "seg_update_kernel<4, 32, 1, 1>": "seg_update_kernel<user side>",
    "seg_long_partial_kernel": "seg_long_partial_kernel",
    "seg_long_combine_kernel": "seg_long_combine_kernel",
    "bpr_tail_kernel": "bpr_tail_kernel",
    "bpr_tail_combine_kernel": "bpr_tail_combine_kernel",
}


def read(d, counter):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"]
                for pat, short in SHORT.items():
                    if pat in name:
                        rows.setdefault(short, []).append(float(r["Counter_Value"]))
    return rows


def main():
    fetch = read(sys.argv[1], "FETCH_SIZE")
    write = read(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/step_loop.py, C4 workload",
           "corrections": "KiB -> bytes (x1024); FETCH_SIZE x2 on gfx950 (16 B/lane reads tallied at half)",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        # the first launches touch cold tables; the mean skips them when there are enough samples
        fs, wsz = fetch.get(k, []), write.get(k, [])
        fs, wsz = (fs[len(fs) // 3:] or fs), (wsz[len(wsz) // 3:] or wsz)
        fb = 2 * 1024 * sum(fs) / len(fs) if fs else None
        wb = 1024 * sum(wsz) / len(wsz) if wsz else None
        out["kernels"][k] = {"fetch_bytes": fb, "write_bytes": wb,
                             "hbm_bytes": (fb or 0) + (wb or 0) if (fb is not None and wb is not None) else None,
                             "launches": [len(fs), len(wsz)]}
    name = sys.argv[3] if len(sys.argv) > 3 else "r03_pmc_traffic.json"
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


main()
