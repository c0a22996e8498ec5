"""Summarise the two rocprofv3 --pmc passes of tools/pmc_score.sh (SQ counters over tools/eval_loop.py) into
profiles/r03_pmc_score_kernel.json: per scoring kernel the counter means per launch and the instruction counts per wave
and 64-item tile (a wave-tile = 64 MFMAs of 32x32x2 at k = 128).

    bash tools/pmc_score.sh && python tools/pmc_score_summary.py gpurun_out/pmc_score1 gpurun_out/pmc_score2"""
import csv
import glob
import json
import os
import sys

KERNELS = {"score_kernel_t1": "dr::score_kernel_t1<64, true, false> (top-10 list + ranks, 64-user panels, 2 workgroups per CU)",
           "score_rank_kernel": "dr::score_rank_kernel<64, true, false, 4, false, false> (ranks only, 128-user panels, 8 waves per CU)"}
WAVE_TILES = (16384 // 32) * 2 * ((1_000_000 + 63) // 64)      # (user groups of 32) x (2 item halves) x tiles, either kernel


def main():
    out = {"source": "rocprofv3 --pmc (two passes, 8 SQ counters each; tools/pmc_score.sh) over tools/eval_loop.py: 16384 users x "
                     "1M items x 128, K = 10, 100 masked train items + 1 held-out item per user",
           "wave_tiles_per_launch": WAVE_TILES, "kernels": {}}
    for d in sys.argv[1:3]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    for pat in KERNELS:
                        if pat in r["Kernel_Name"]:
                            k = out["kernels"].setdefault(pat, {"kernel": KERNELS[pat], "_v": {}})
                            k["_v"].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for pat, k in out["kernels"].items():
        c = {n: sum(v) / len(v) for n, v in sorted(k.pop("_v").items())}
        k["counters_per_launch"] = c
        wt = WAVE_TILES
        d = {}
        if "SQ_INSTS_VALU" in c:
            d["valu_instructions_per_wave_tile_incl_64_mfma"] = c["SQ_INSTS_VALU"] / wt
            d["non_mfma_valu_per_wave_tile"] = c["SQ_INSTS_VALU"] / wt - 64
        for n, key in (("SQ_INSTS_SALU", "salu_per_wave_tile"), ("SQ_INSTS_LDS", "lds_per_wave_tile"),
                       ("SQ_INSTS_VMEM", "vmem_per_wave_tile"), ("SQ_INSTS_BRANCH", "branch_per_wave_tile")):
            if n in c:
                d[key] = c[n] / wt
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_ACTIVE_INST_LDS"):
            d["lds_bank_conflict_share_of_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"]
        if c.get("SQ_WAVE_CYCLES"):
            d["wave_cycles_per_wave_tile"] = c["SQ_WAVE_CYCLES"] / wt
            if "SQ_WAIT_ANY" in c:
                d["wait_any_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        k["derived"] = d
    name = sys.argv[3] if len(sys.argv) > 3 else "r03_pmc_score_kernel.json"
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


main()
