#!/usr/bin/env python
"""ISA check behind `make check` (deeprec_amd/csrc/Makefile): no workgroup barrier with an LDS store or LDS atomic of the
same wave still in flight.

Round 2 found `s_barrier` on a loop back edge of score_kernel_t1 with the wave's last candidate store + ds_or still
outstanding (no s_waitcnt in the latch): the row's owner -- another wave -- read the word without the bit and users
lost top-K entries, different ones per run.  The source now places `s_waitcnt lgkmcnt(0)` in front of those barriers;
this script makes sure that stays true in the generated code of EVERY instantiation, whatever the compiler does later.

Input: device assembly (`hipcc -S --cuda-device-only ...`).  For every `s_barrier` of every kernel whose demangled name
matches --kernels, walk the control-flow graph backwards from the barrier along every path until either
  * an `s_waitcnt` that waits for lgkmcnt(0)      -> that path is safe, or
  * an LDS store / LDS atomic (`ds_write*`, `ds_store*`, `ds_or*`, `ds_add*`, ...) -> VIOLATION, or
  * the kernel's entry                             -> safe.
LDS loads in flight are harmless (they return pre-barrier data, which is what the program asked for).
Exit status 1 if any violation is found.
"""
import argparse
import re
import subprocess
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_Z\w+):")
BRANCH = re.compile(r"^s_(c?branch\w*)\s+(\.LBB\d+_\d+)")
LDS_WRITE = re.compile(r"^ds_(write|store|or|and|xor|add|sub|rsub|inc|dec|min|max|mskor|cmpst|cmpswap|wrxchg|wrap|append|consume|"
                       r"pk_add|condxchg)")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.splitlines()))
    except Exception:
        return {n: n for n in names}


def waits_lgkm0(ins):
    if not ins.startswith("s_waitcnt"):
        return False
    m = re.search(r"lgkmcnt\((\d+)\)", ins)
    if m:
        return int(m.group(1)) == 0
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", ins)      # raw immediate: lgkmcnt = bits 8..11 (gfx9)
    if m:
        return ((int(m.group(1), 0) >> 8) & 0xF) == 0
    return False


def parse(path):
    funcs, cur, name = {}, None, None
    for raw in open(path):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        m = FUNC.match(line)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur, name = None, None
            continue
        if line.startswith(".") and not LABEL.match(line):
            continue                                     # directives
        cur.append(line)
    return funcs


def check_function(ins):
    labels = {}
    for p, l in enumerate(ins):
        m = LABEL.match(l)
        if m:
            labels[m.group(1)] = p
    preds = {}
    for p, l in enumerate(ins):
        m = BRANCH.match(l)
        if m:
            preds.setdefault(m.group(2), []).append(p)
    bad = []
    for b, l in enumerate(ins):
        if not l.startswith("s_barrier"):
            continue
        seen, stack, hit = set(), [b - 1], None
        while stack and hit is None:
            p = stack.pop()
            while p >= 0:
                if p in seen:
                    break
                seen.add(p)
                cur = ins[p]
                m = LABEL.match(cur)
                if m:
                    for q in preds.get(m.group(1), []):
                        stack.append(q - 1)
                    # fall-through predecessor unless the previous instruction never falls through
                    if p > 0 and (ins[p - 1].startswith("s_branch") or ins[p - 1].startswith("s_endpgm")
                                  or ins[p - 1].startswith("s_setpc")):
                        break
                    p -= 1
                    continue
                if waits_lgkm0(cur):
                    break
                if LDS_WRITE.match(cur):
                    hit = (p, cur)
                    break
                p -= 1
        if hit is not None:
            bad.append((b, hit))
    return sum(1 for l in ins if l.startswith("s_barrier")), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm", nargs="+")
    ap.add_argument("--kernels", default="score_kernel", help="regex on the demangled kernel name")
    ap.add_argument("--expect-violations", action="store_true",
                    help="invert the exit status (used to show that the check catches the round-2 bug in a build without the waits)")
    a = ap.parse_args()
    pat = re.compile(a.kernels)
    total_k = total_b = 0
    failures = []
    for path in a.asm:
        funcs = parse(path)
        dm = demangle(list(funcs))
        for name, ins in funcs.items():
            if not pat.search(dm[name]):
                continue
            nb, bad = check_function(ins)
            if nb == 0:
                continue
            total_k += 1
            total_b += nb
            for b, (p, cur) in bad:
                failures.append("%s\n    s_barrier #%d reachable from `%s` (instruction %d) without s_waitcnt lgkmcnt(0)"
                                % (dm[name][:160], b, cur, p))
    print("check_barriers: %d kernels matching /%s/, %d barriers, %d violations" % (total_k, a.kernels, total_b, len(failures)))
    for f in failures:
        print("  VIOLATION " + f)
    if total_k == 0:
        print("check_barriers: no kernel matched -- refusing to pass vacuously")
        return 1
    if a.expect_violations:
        return 0 if failures else 1
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
