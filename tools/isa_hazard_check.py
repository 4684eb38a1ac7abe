#!/usr/bin/env python3
"""Scan `hipcc -S` output for VALU reads of an MFMA result with fewer than 7 wait states, following
branches.  ROCm 7.2's hipcc was seen to pad this hazard on one side of a branch only (conv1x1.hip,
`if (relu)` right behind the last MFMA of an accumulator): the fall-through path read the accumulator
registers one cycle after issue.  Usage: isa_hazard_check.py file.s [...]; exit status 1 if anything is found."""
import re
import sys


def _regs(tok):
    tok = tok.split()[0] if tok else tok
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path, need=7):
    """-> list of (mfma instruction index, wait states seen, offending instruction)"""
    lines, labels = [], {}
    for raw in open(path):
        s = raw.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(lines)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        lines.append(s)
    found = []

    def walk(start, dst, waits, depth, origin):
        j = start
        while j < len(lines) and waits < need and depth < 4:
            t = lines[j]
            if t.startswith("s_nop"):
                waits += int(t.split()[1]) + 1
            elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                tgt = t.split()[1]
                if tgt in labels:
                    walk(labels[tgt], set(dst), waits + 1, depth + 1, origin)
                if t.startswith("s_branch"):
                    return
                waits += 1
            elif t.startswith("v_mfma"):
                return
            elif t.startswith(("s_", "ds_", "global_", "buffer_", "scratch_", "flat_")):
                waits += 1
            else:
                parts = t.split(None, 1)
                ops = [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else []
                src = set()
                for o in ops[1:]:
                    src |= _regs(o)
                if src & dst:
                    found.append((origin, waits, t))
                    return
                if ops:
                    dst = dst - _regs(ops[0])
                if not dst:
                    return
                waits += 1
            j += 1

    for i, l in enumerate(lines):
        if l.startswith("v_mfma"):
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
            walk(i + 1, _regs(ops[0]), 0, 0, i)
    return found


def scan_store_data(path, need=2):
    """VALU writes to a data register of a 12- or 16-byte store fewer than `need` wait states behind it.  gfx950 reads the data of such a store late for lanes 12..15 of every row of 16
    (tools/ubench/store_data_war.hip, store_war2.hip: buffer stores need 1 wait state, global stores 2); hipcc 7.2 pads
    global / flat stores and buffer stores with an IMMEDIATE soffset, but not buffer stores whose soffset is an SGPR.
    -> list of (store index, wait states, offending instruction)"""
    lines, labels = [], {}
    for raw in open(path):
        s = raw.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(lines)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        lines.append(s)
    found = []

    def walk(start, data, ws, depth, origin):
        j = start
        while j < len(lines) and ws < need and depth < 4:
            t = lines[j]
            if t.startswith("s_nop"):
                ws += int(t.split()[1]) + 1
            elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                tgt = t.split()[1]
                if tgt in labels:
                    walk(labels[tgt], data, ws + 1, depth + 1, origin)
                if t.startswith("s_branch"):
                    return
                ws += 1
            elif t.startswith("s_"):
                ws += 1
            else:
                parts = t.split(None, 1)
                ops = [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else []
                # memory instructions write their destinations a latency later, not at issue: they only count as time
                is_mem = t.startswith(("ds_", "buffer_", "global_", "flat_", "scratch_"))
                dst = set() if is_mem or not ops else _regs(ops[0])
                if t.startswith("v_permlane") and len(ops) > 1:
                    dst |= _regs(ops[1])
                if dst & data:
                    found.append((origin, ws, t))
                    return
                ws += 1
            j += 1

    for i, l in enumerate(lines):
        m = re.match(r"(buffer|global|flat|scratch)_store_dwordx[34]\s+(.*)", l)
        if not m:
            continue
        ops = [t.strip() for t in m.group(2).split(",")]
        data = _regs(ops[0]) if m.group(1) == "buffer" else _regs(ops[1])      # buffer: vdata first; global/flat: vaddr, vdata
        walk(i + 1, data, 0, 0, i)
    return found


def scan_src_overwrite(path, window=16):
    """VALU writes to a register that an MFMA issued fewer than `window` cycles earlier reads as SrcA/SrcB.
    Written while chasing position-dependent garbage in tile columns 12..15 (round 1); it turned out that hipcc
    reuses MFMA source registers 4-5 cycles after issue in every kernel of this library and the results are
    right, i.e. the operands are latched at issue — kept as a diagnostic.  Cycle model: SALU / branch 1,
    VALU / DS / VMEM issue 4, s_nop N -> N + 1.
    -> list of (mfma index, cycles, offending instruction)"""
    lines, labels = [], {}
    for raw in open(path):
        s = raw.strip()
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(lines)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        lines.append(s)
    found = []

    def walk(start, src, cyc, depth, origin):
        j = start
        while j < len(lines) and cyc < window and depth < 4:
            t = lines[j]
            if t.startswith("s_nop"):
                cyc += int(t.split()[1]) + 1
            elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                tgt = t.split()[1]
                if tgt in labels:
                    walk(labels[tgt], src, cyc + 1, depth + 1, origin)
                if t.startswith("s_branch"):
                    return
                cyc += 1
            elif t.startswith("v_mfma"):
                return                      # the pipe is busy for its 4 passes: nothing lands earlier
            elif t.startswith("s_"):
                cyc += 1
            elif t.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")):
                cyc += 4                    # results arrive a memory latency later
            else:
                parts = t.split(None, 1)
                ops = [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else []
                dst = _regs(ops[0]) if ops else set()
                if t.startswith("v_permlane") and len(ops) > 1:
                    dst |= _regs(ops[1])
                if dst & src:
                    found.append((origin, cyc, t))
                    return
                cyc += 4
            j += 1

    for i, l in enumerate(lines):
        if l.startswith("v_mfma"):
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
            walk(i + 1, _regs(ops[1]) | _regs(ops[2]), 4, 0, i)
    return found


if __name__ == "__main__":
    bad = 0
    for f in [a for a in sys.argv[1:] if not a.startswith("--")]:
        r = scan(f)
        for origin, waits, t in r:
            print(f"{f}: MFMA #{origin} result read after {waits} wait states: {t[:80]}")
        r3 = scan_store_data(f)
        for origin, ws, t in r3:
            print(f"{f}: store #{origin}: data register overwritten {ws} wait state(s) later: {t[:80]}")
        r = r + r3
        if "--src-overwrite" in sys.argv:      # informational: hipcc does this everywhere and the hardware copes
            for origin, cyc, t in scan_src_overwrite(f):
                print(f"{f}: MFMA #{origin} source operand overwritten {cyc} cycles after issue: {t[:80]}")
        bad += len(r)
        print(f"{f}: {len(r)} hazard(s)")
    sys.exit(1 if bad else 0)
