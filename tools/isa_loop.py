"""Compact view of a kernel's hottest loop (the one with the most MFMAs) from a device-only assembly file:
runs of instructions of one class are collapsed to `class x n`.
    hipcc ... -S --cuda-device-only src.hip -o /tmp/x.s ; python tools/isa_loop.py /tmp/x.s <mangled-name-substring> [max lines]"""
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 120
m = [x for x in re.finditer(r"^(_Z\w+):", s, re.M) if sub in x.group(1)]
name = m[0].group(1)
a = m[0].start()
b = s.index(".Lfunc_end", a)
body = [re.sub(r"\s+", " ", l.split(";")[0].strip()) for l in s[a:b].splitlines()]
raw = s[a:b].splitlines()
# loops: header label line index -> the last backward branch to it
labels = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":")}
loops = []
for i, l in enumerate(body):
    t = l.split()
    if t and t[0].startswith(("s_cbranch", "s_branch")) and t[-1] in labels and labels[t[-1]] < i \
            and i - labels[t[-1]] < 0.9 * len(body):      # (a branch back to an exit block at the top is no loop)
        loops.append((labels[t[-1]], i))
best = max(loops, key=lambda ab: sum("v_mfma" in x for x in body[ab[0]:ab[1]]))


def cls(l):
    op = l.split()[0]
    if op.endswith(":"):
        return "label"
    if "mfma" in op:
        return "MFMA"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return l
    if op.startswith("ds_"):
        return op
    if op.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_branch")):
        return l
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return op


print(name, "loop lines", best, "mfma", sum("v_mfma" in x for x in body[best[0]:best[1]]))
out, prev, n = [], None, 0
for l in body[best[0]:best[1] + 1]:
    if not l:
        continue
    c = cls(l)
    if c == "label":
        c = l
    if c == prev:
        n += 1
    else:
        if prev is not None:
            out.append("%s x %d" % (prev, n) if n > 1 else prev)
        prev, n = c, 1
out.append("%s x %d" % (prev, n) if n > 1 else prev)
for l in out[:limit]:
    print("  ", l)
