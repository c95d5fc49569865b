#!/usr/bin/env python3
"""Static check of the HAZARD RULE of csrc/query.hip: hipcc's hazard recogniser inserts the wait states between a matrix
instruction and the first VALU / memory instruction that touches its destination registers only for instructions it KNOWS; the operands of an inline-asm
statement are not covered.  A first consumer written in inline assembly therefore reads stale registers unless enough independent instructions happen to
sit in between -- correctness by scheduling luck, and a new compiler may shuffle the luck away.

Input: the output of `hipcc -S --cuda-device-only` for a source file.  For every v_mfma_* the instructions behind it are walked in program order
(fall-through across labels; the walk of one MFMA ends at a branch, at the end of the kernel, at the first compiler-known instruction that reads or
overwrites its destination -- the recogniser has covered that one, everything later sees a finished result -- or after WINDOW wait states, whichever comes
first; s_nop N counts N + 1).  An instruction between ;;#ASMSTART / ;;#ASMEND that reads or writes a register of the destination inside the window is a
violation.  WINDOW = 20 wait states: the longest matrix instruction of the file (16 passes) + its 2..3 extra states on gfx940 / gfx950
(llvm GCNHazardRecognizer: passes + 2), rounded up.

usage: check_isa.py file.s [file.s ...]      exit code 1 on a violation; prints one summary line per kernel that has MFMAs"""
import re
import sys

WINDOW = 20
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(2) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(1), i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


STORES = ("ds_write", "ds_store", "buffer_store", "global_store", "scratch_store", "flat_store", "buffer_atomic", "global_atomic", "ds_add", "ds_min", "ds_max")


def split_ops(line):
    line = line.split(";")[0].strip()
    if not line:
        return None, []
    parts = line.split(None, 1)
    return parts[0], ([o.strip() for o in parts[1].split(",")] if len(parts) > 1 else [])


def check(path):
    lines = open(path).read().split("\n")
    prog = []          # (kernel, opcode, dst regs, src regs, from_asm, line number, text)
    kernel = None; in_asm = False
    for n, l in enumerate(lines, 1):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if s.startswith(";;#ASMEND"):
            in_asm = False; continue
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".") and not l.startswith("\t"):
            kernel = m.group(1)
        if not l.startswith("\t") or s.startswith((".", ";")):
            continue
        op, ops = split_ops(s)
        if op is None:
            continue
        if op.startswith(STORES):
            dst, src = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        else:
            dst = regs(ops[0]) if ops else set()
            src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if in_asm:
                src |= dst                 # ("+v" operands: the asm text does not say which of its operands are read-modify-write)
        prog.append((kernel, op, dst, src, in_asm, n, s))
    bad = []; per_kernel = {}
    for i, (k, op, dst, src, asm, n, text) in enumerate(prog):
        if not op.startswith("v_mfma"):
            continue
        st = per_kernel.setdefault(k, [0, 0]); st[0] += 1
        ws = 0
        for j in range(i + 1, len(prog)):
            k2, op2, d2, s2, asm2, n2, t2 = prog[j]
            if k2 != k or op2 in ("s_endpgm", "s_branch", "s_setpc_b64"):
                break
            touches = bool((s2 | d2) & dst)
            if touches and asm2 and ws < WINDOW:
                bad.append((path, k, n, text, n2, t2, ws)); st[1] += 1
                break
            if touches and not op2.startswith("v_mfma"):
                break                      # a compiler-known first consumer / overwriter: covered by the hazard recogniser
            if touches and op2.startswith("v_mfma"):
                break                      # the accumulation chain goes on: that MFMA is walked on its own
            ws += (int(t2.split()[1], 0) + 1) if op2 == "s_nop" else 1
            if ws >= WINDOW:
                break
    return bad, per_kernel


def main():
    rc = 0
    for p in sys.argv[1:]:
        bad, per_kernel = check(p)
        tot = sum(v[0] for v in per_kernel.values())
        print(f"{p}: {len(per_kernel)} kernels with matrix instructions, {tot} v_mfma walked, {len(bad)} asm statements inside the hazard window of an MFMA result")
        for path, k, n, text, n2, t2, ws in bad[:20]:
            print(f"  VIOLATION {k}: line {n}: {text}\n      -> line {n2} (asm, {ws} wait states later): {t2}")
        rc |= bool(bad)
    sys.exit(rc)


if __name__ == "__main__":
    main()
