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
    # third check (round 5): the hand-written part of k_solo -- solo_agpr.h / solo_pass_agpr.h name a128..a255 literally, which the
    # register allocator knows nothing about.  (a) the kernel descriptors of both instantiations must grant all 512 registers with the
    # accumulation half starting at 256 (.amdhsa_next_free_vgpr 512, .amdhsa_accum_offset 256), or a255 does not exist;
    # (b) a memory instruction must not read an MFMA result for 18 cycles (solo_pass_agpr.h: pt_settle; the compiler inserts such
    # gaps for its own code, not for inline asm): walking k_solo's listing INCLUDING the asm blocks, every `global_store ... a[` must
    # be at least 18 wait states behind the last v_mfma in straight-line order (s_nop N = N + 1, any other instruction 1; a label
    # or branch resets nothing -- the count is conservative: it only ever under-estimates the distance).
    desc_bad, settle_bad = [], []
    raw = open(path).read().splitlines()
    kern = None
    seen_desc = {}
    for l in raw:
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", l)
        if m:
            kern = m.group(1)
        m = re.match(r"^\s*\.amdhsa_(next_free_vgpr|accum_offset)\s+(\d+)", l)
        if m and kern and "k_solo" in kern:
            seen_desc.setdefault(kern, {})[m.group(1)] = int(m.group(2))
    for k, d in sorted(seen_desc.items()):
        if d.get("next_free_vgpr") != 512 or d.get("accum_offset") != 256:
            desc_bad.append((k, d))
    if len(seen_desc) < 2:
        desc_bad.append(("k_solo", "expected the kernel descriptors of two instantiations, found %d" % len(seen_desc)))
    func, since_mfma, min_gap, n_stores = None, None, None, 0
    for i, l in enumerate(raw):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func, since_mfma = m.group(1), None
            continue
        if func is None or "k_solo" not in func:
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            since_mfma = 0
            continue
        if op.startswith("global_store") and re.search(r"\ba\[", t):
            n_stores += 1
            if since_mfma is not None:
                min_gap = since_mfma if min_gap is None else min(min_gap, since_mfma)
                if since_mfma < 18:
                    settle_bad.append((func, i + 1, since_mfma))
        if since_mfma is not None:
            mm = re.match(r"s_nop\s+(\d+)", t)
            since_mfma += (int(mm.group(1)) + 1) if mm else 1
    for k, d in desc_bad:
        print("%s: kernel descriptor does not grant a128..a255 (want next_free_vgpr 512, accum_offset 256): %s" % (k, d))
    for f, ln, gap in settle_bad[:10]:
        print("%s: global_store of an accumulation register only %d wait states behind a v_mfma (line %d; 18 needed: pt_settle)" % (f, gap, ln))
    print("k_solo: %d stores from accumulation registers, the closest %s wait states behind an MFMA; descriptors: %s" % (
        n_stores, min_gap, {"".join(re.findall(r"Lb([01])E", k)) or k: v for k, v in seen_desc.items()}))
    if n_stores == 0:
        desc_bad.append(("k_solo", "no accumulation-register stores found: the listing is not what this check was written for"))
        print("k_solo: no `global_store ... a[` found -- has the in-kernel pass moved?  (check 3 needs an update)")
    if desc_bad or settle_bad:
        bad.append(("k_solo", 0, len(desc_bad) + len(settle_bad)))
    for f, ln, text in agpr_bad[:10]:
        print("%s: compiler-generated use of an accumulation register >= a128 (reserved for solo_agpr.h), line %d: %s" % (f, ln, text))
    print("k_solo: highest accumulation register the compiler itself uses: a%d (a128..a255 are solo_agpr.h's)" % agpr_hi)
    if agpr_bad:
        bad.append(("k_solo", agpr_bad[0][1], len(agpr_bad)))
    for f, ln, n in bad:
        if ln == 0:
            print("%s: %d finding(s) of the accumulation-register checks above" % (f, n))
        else:
            print("%s: %d register copies directly before an exec-widening s_or_b64 (line %d of the filtered listing)" % (f, n, ln))
    print("%s: %d suspicious site(s)" % (path, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
