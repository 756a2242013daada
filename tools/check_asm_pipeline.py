#!/usr/bin/env python3
"""Invariant of the hand-pipelined kernels (elprep_amd/csrc/gload.hpp): between an asm global load and the hand-placed wait that covers
it nothing may read or overwrite the load's destination registers (the compiler does not know they are still in flight).  A wait
`s_waitcnt vmcnt(N)` covers a load when at least N hand-written vector-memory instructions (asm statements, i.e. unconditional ones)
follow the load in front of the wait: vector-memory instructions complete in issue order, compiler-generated ones in between can
only push the load further from the youngest N.  Compiles the
translation unit to ISA and walks every kernel linearly (loads issued under exec masks sit in straight-line code, so program order in
the listing is what matters; a branch target between load and wait is treated conservatively: the pending set survives labels).
usage: check_asm_pipeline.py <file.hip> [...]   exit code 1 on a violation"""
import re
import subprocess
import sys
import tempfile

VREG = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1):
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(path, hipcc="/opt/rocm/bin/hipcc"):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=off", "--cuda-device-only", "-S",
                               path, "-o", tmp.name], stderr=subprocess.DEVNULL)
        lines = open(tmp.name).read().split("\n")
    bad, kernel, pending, in_asm, loads = [], None, set(), False, []
    n_loads = 0
    for ln, raw in enumerate(lines, 1):
        l = raw.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel, pending, loads = m.group(1), set(), []
            continue
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m:
            if in_asm:  # only the hand-placed wait counts: a compiler-inserted one may sit in a branch that is not taken
                n = int(m.group(1))
                loads = [ld for ld in loads if ld[1] < n]  # loads with fewer than n hand-written instructions behind them stay pending
                pending = set().union(*[ld[0] for ld in loads]) if loads else set()
            continue
        if in_asm and (l.startswith("global_load") or l.startswith("global_store")):
            for ld in loads:
                ld[1] += 1
            if l.startswith("global_load"):
                dst = regs(l.split()[1].rstrip(","))
                loads.append([dst, 0])
                pending |= dst
                n_loads += 1
            # the address (and store data) operands are read at issue - but they must not be in flight themselves
            srcs = regs(l.split(None, 2)[2]) if l.startswith("global_load") else regs(l)
            hit = srcs & (pending - (dst if l.startswith("global_load") else set()))
            if hit:
                bad.append((kernel, ln, l, sorted(hit)))
            continue
        if pending:
            hit = regs(l) & pending
            if hit:
                bad.append((kernel, ln, l, sorted(hit)))
    return n_loads, bad


if __name__ == "__main__":
    rc = 0
    for f in sys.argv[1:]:
        n, bad = check(f)
        print(f"{f}: {n} asm loads, {len(bad)} violation(s)")
        for k, ln, l, hit in bad[:20]:
            print(f"  {k}: line {ln}: `{l}` touches in-flight v{hit}")
        rc |= 1 if bad else 0
    sys.exit(rc)
