"""Heuristic scan of device assembly for serialized memory round trips: a global load whose very next memory-relevant
instruction is `s_waitcnt vmcnt(0)` (nothing else put in flight beside it).  Prints, per kernel, how many loads are
waited for alone -- the shape `x = 0; if (ok) x = load(...)` compiles to.   python tools/isa_serial_loads.py a.s b.s ..."""
import re
import subprocess
import sys

for path in sys.argv[1:]:
    s = open(path).read()
    funcs = list(re.finditer(r"^(_Z\w+):", s, re.M))
    for fi, m in enumerate(funcs):
        a = m.start()
        b = s.find(".Lfunc_end", a)
        if b < 0:
            b = len(s)
        ins = [re.sub(r"\s+", " ", l.split(";")[0].strip()) for l in s[a:b].splitlines()]
        ins = [l for l in ins if l and not l.startswith(".") and not l.endswith(":")]
        alone = 0
        loads = 0
        for i, l in enumerate(ins):
            if l.startswith("global_load"):
                loads += 1
                # next instruction that is a load, a store or a wait
                for n in ins[i + 1:i + 12]:
                    if n.startswith(("global_load", "global_store", "ds_", "v_mfma")):
                        break
                    if n.startswith("s_waitcnt") and "vmcnt(0)" in n:
                        alone += 1
                        break
        if alone >= 2:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            print("%-28s %3d of %3d loads waited for alone   %s" % (path.split("/")[-1], alone, loads, name[:110]))
