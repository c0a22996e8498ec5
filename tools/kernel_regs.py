"""Register / scratch / occupancy table of the kernels in one HIP source (compile remarks of
-Rpass-analysis=kernel-resource-usage).  usage: python tools/kernel_regs.py deeprec_amd/csrc/score_topk.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Iinclude",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if flt in r["name"]:
        print("%-90s V %3d A %3d scratch %4d occ %d spillV %3d spillS %3d LDS %d" % (
            r["name"].replace("dr::", "")[:90], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1),
            r.get("Occupancy", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1), r.get("LDS Size", -1)))
