#!/usr/bin/env python3
"""Static check of the generated gfx950 assembly for one miscompile pattern.

The register allocator's live-range splitting may place its copies (v_mov / v_accvgpr_write of a
long-lived per-lane value) at the END of a block that runs under a narrowed exec mask, right in front of
the `s_or_b64 exec, exec, s[..]` that widens the mask again.  Lanes that were masked off then lose the
value although they need it later (k_chain: lanes whose landmark does not exist yet wake up when a New
landmark is appended).  Seen with hipcc 7.2 / clang 22 at -O2/-O3 with the greedy VGPR allocator; the
build therefore uses -mllvm -vgpr-regalloc=basic (no live-range splitting), and this script fails when a
run of >= MIN_COPIES register-to-register copies directly precedes an exec-widening instruction.

usage: check_exec_split.py file.s [MIN_COPIES]
"""
import re
import sys

COPY = re.compile(r"^\s*(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32|v_accvgpr_mov_b32)\s+[va]\[?\d+(:\d+)?\]?, [va]\[?\d+(:\d+)?\]?\s*$")
WIDEN = re.compile(r"^\s*s_or_b64 exec, exec, s\[\d+:\d+\]")
SAVEEXEC = re.compile(r"^\s*s_(and|andn2|or)_saveexec_b64 ")


def main():
    path = sys.argv[1]
    min_copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8  # a few copies are ordinary phi moves of the region itself
    # inline-asm blocks (between ";;#ASMSTART" and ";;#ASMEND") are hand-written and checked by eye: both checks look at compiler output only
    lines, in_asm = [], False
    for l in open(path):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif not in_asm and not re.match(r"^\s*(\.loc|;|\.Ltmp|\.cfi)", l):
            lines.append(l)
    func, bad, run, body_only = None, [], 0, False
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func, run = m.group(1), 0
            continue
        if COPY.match(l):
            if run == 0:  # copies that are the WHOLE body of a narrowed region are a predicated select, not a split
                body_only = i > 0 and SAVEEXEC.match(lines[i - 1]) is not None
            run += 1
            continue
        if WIDEN.match(l) and run >= min_copies and not body_only:
            bad.append((func, i + 1, run))
        run = 0
    # second check (round 4): k_solo<true> keeps the first half of a long window in accumulation registers a128..a255 by explicit
    # v_accvgpr instructions the register allocator does not see (csrc/solo_agpr.h).  That is only sound while the compiler's own use of
    # accumulation registers (spill slots, from a0 upwards) stays below a128: no instruction outside an inline-asm block may name one.
    agpr_hi, agpr_bad = -1, []
    func = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
        if func is None or "k_solo" not in func or re.match(r"^\s*\.", l):  # (both instantiations: the tile of the in-kernel pass lives there too)
            continue
        for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", l):
            hi = int(m.group(2) or m.group(1))
            agpr_hi = max(agpr_hi, hi)
            if hi >= 128:
                agpr_bad.append((func, i + 1, l.strip()))
    for f, ln, text in agpr_bad[:10]:
        print("%s: compiler-generated use of an accumulation register >= a128 (reserved for solo_agpr.h), line %d: %s" % (f, ln, text))
    print("k_solo: highest accumulation register the compiler itself uses: a%d (a128..a255 are solo_agpr.h's)" % agpr_hi)
    if agpr_bad:
        bad.append(("k_solo", agpr_bad[0][1], len(agpr_bad)))
    for f, ln, n in bad:
        print("%s: %d register copies directly before an exec-widening s_or_b64 (line %d of the filtered listing)" % (f, n, ln))
    print("%s: %d suspicious site(s)" % (path, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
