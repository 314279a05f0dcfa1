#!/usr/bin/env python3
"""Writes 2d-ekf-slam_amd/csrc/flush_pipe_agpr.h: one 64 x 64 tile of the 16-pair dense pass as ONE software-pipelined asm statement (round 5).

Used by k_flush_rows (ekf_flush_rows.hip): the dense pass of a full window of 32 measurements for a single large filter, one wave per SIMD, a wave
owning a run of consecutive tiles of one tile row.  At one wave per SIMD nothing but the wave itself hides a memory trip, so registers and waits are
owned by hand:
  * the tile lives in a128..a255 (16 chains of 8 registers), loaded, contracted (v_mfma_f64_16x16x4_f64, the chain as C and D) and stored from there;
  * the B operands travel two pairs (one "sub-sweep": 2 pairs x 4 column blocks = 8 loads of 8 bytes per lane) at a time through a ring of three
    buffers in v208..v255, requested two sub-sweeps (64 MFMAs) ahead of their use;
  * in the LAST sub-sweep a row-block's chains are stored as soon as its MFMAs are done (18 wait states) and the NEXT tile's chains are requested
    into the same registers at once: the next tile's trip to HBM runs under this tile's remaining MFMAs (its first two sub-sweeps of B operands
    are requested here too).  Loads come from `next`, stores go to `tile`: a pass from buffer to buffer needs nothing else;
  * every wait is an `s_waitcnt vmcnt(N)` computed by this generator from the issue order (vector memory operations complete in order): N = the
    operations issued after the youngest one that is needed.  A block may be entered with everything it assumes in flight already complete (after
    the prologue, after a diagonal tile: the caller drains) -- waits then pass at once; it must never be entered with FEWER younger operations in
    flight than assumed while the needed ones are pending, which is why the non-diagonal block's entry assumption is exactly its own exit state
    (asserted as a fixed point) and the caller drains everywhere else.
The A operands come from the wave's LDS stage (one ds_read_b64 per pair and row-block, one element ahead); pairs are applied to every chain in
ascending order, as k_flush_rb applies them.  First built for k_solo's own pass (scripts/dropped/r05_in_kernel_pass_software_pipelined.patch), where
it was bitwise equal at the first run and no faster -- that pass is MFMA-bound; the pass of a single N = 4096 filter beside its chain kernel is not.
CPU; usage: python3 scripts/r05_gen_flush_pipe.py [--check]."""
import os
import sys

TILE_BASE = 128
B_BASE = 208
NP = 16
NSS = NP // 2  # sub-sweeps of two pairs


def chain_reg(ch):
    return TILE_BASE + 8 * ch


def b_reg(buf, pl, cc):
    return B_BASE + 16 * buf + 8 * pl + 2 * cc


class Emit:
    def __init__(self, inflight):
        self.lines = []
        self.inflight = list(inflight)  # oldest first

    def ins(self, text):
        self.lines.append(text)

    def vmem(self, text, tag):
        self.lines.append(text)
        self.inflight.append(tag)

    def need(self, tags):
        """wait until every operation whose tag is in `tags` has completed (in-order completion)"""
        idx = [i for i, t in enumerate(self.inflight) if t in tags]
        if not idx:
            return
        last = max(idx)
        younger = len(self.inflight) - 1 - last
        self.lines.append("s_waitcnt vmcnt(%d)" % min(younger, 63))
        if younger <= 63:
            self.inflight = self.inflight[last + 1:]
        else:  # the counter saturates: only what lies beyond the 63 youngest is known to be complete -- and that must cover `last`
            keep = self.inflight[len(self.inflight) - 63:]
            assert last < len(self.inflight) - 63, "a needed operation lies within the 63 youngest: vmcnt cannot express the wait"
            self.inflight = keep


def live(diag, rc, cc):
    return (not diag) or cc >= rc


def gen_block(diag, entry):
    """one tile.  entry: the in-flight operations assumed at entry (oldest first)."""
    e = Emit(entry)
    # offset walkers: vb (this tile's B operands from pair 4 on), vbn (next tile's B operands), vt (stores), vn (next tile's loads)
    e.ins("v_lshl_add_u32 %[vb], %[ss], 2, %[lob]")  # pairs 0..3 were requested by the block before (or the prologue)
    e.ins("v_mov_b32 %[vbn], %[lob]")
    e.ins("v_mov_b32 %[vt], %[voff]")
    e.ins("v_mov_b32 %[vn], %[voff]")
    e.ins("ds_read_b64 %[a0], %[as] offset:0")  # A element of (pair 0, row-block 0)
    k = 0  # A elements consumed so far; element k lives in a[k & 1]

    def issue_b(buf, tag, nxt):
        walker, base = ("%[vbn]", "%[fbn]") if nxt else ("%[vb]", "%[fb]")
        for pl in range(2):
            for cc in range(4):
                e.vmem("global_load_dwordx2 v[%d:%d], %s, %s offset:%d" % (b_reg(buf, pl, cc), b_reg(buf, pl, cc) + 1, walker, base, cc * 512), tag)
            e.ins("v_add_u32 %s, %s, %s" % (walker, "%[ss]", walker))

    for ss in range(NSS):
        last = ss == NSS - 1
        # requests two sub-sweeps ahead (the buffer they go to was read by sub-sweep ss - 1, whose MFMAs have been issued)
        if ss + 2 < NSS:
            issue_b((ss + 2) % 3, "B%d" % (ss + 2), False)
        elif ss == NSS - 1:  # the next tile's first sub-sweep -> buffer 0 (read last by sub-sweep NSS - 2: its MFMAs have been issued)
            assert (NSS - 2) % 3 == 0 and (NSS - 1) % 3 == 1
            issue_b(0, "Bn0", True)
        e.need({"B%d" % ss})
        for rc in range(4):
            if ss == 0:
                e.need({"L%d" % rc})
            for pl in range(2):
                q = 2 * ss + pl
                # element order: ss-major, then rc, then pl  ->  element index k = (ss * 4 + rc) * 2 + pl
                if k + 1 < NP * 4:
                    k1 = k + 1
                    ss1, rem = divmod(k1, 8)
                    rc1, pl1 = divmod(rem, 2)
                    e.ins("ds_read_b64 %%[a%d], %%[as] offset:%d" % (k1 & 1, ((2 * ss1 + pl1) * 256 + rc1 * 64) * 8))
                    e.ins("s_waitcnt lgkmcnt(1)")
                else:
                    e.ins("s_waitcnt lgkmcnt(0)")
                for cc in range(4):
                    if not live(diag, rc, cc):
                        continue
                    c = chain_reg(rc * 4 + cc)
                    e.ins("v_mfma_f64_16x16x4_f64 a[%d:%d], %%[a%d], v[%d:%d], a[%d:%d]" % (c, c + 7, k & 1, b_reg(ss % 3, pl, cc), b_reg(ss % 3, pl, cc) + 1, c, c + 7))
                k += 1
            if last:
                # the row-block is final: 18 wait states behind its last MFMA, then its chains go out and the next tile's come in
                e.ins("s_nop 7")
                e.ins("s_nop 7")
                e.ins("s_nop 1")
                for cc in range(4):
                    ch = rc * 4 + cc
                    for h in range(2):
                        off = ((ch * 2 + h) % 4) * 1024
                        if live(diag, rc, cc):
                            r = chain_reg(ch) + 4 * h
                            e.vmem("global_store_dwordx4 %%[vt], a[%d:%d], %%[tile] offset:%d nt" % (r, r + 3, off), "S%d" % rc)
                        if (ch * 2 + h) % 4 == 3:
                            e.ins("v_add_u32 %[vt], 4096, %[vt]")
                for cc in range(4):
                    ch = rc * 4 + cc
                    for h in range(2):
                        off = ((ch * 2 + h) % 4) * 1024
                        r = chain_reg(ch) + 4 * h
                        e.vmem("global_load_dwordx4 a[%d:%d], %%[vn], %%[next] offset:%d" % (r, r + 3, off), "Ln%d" % rc)
                        if (ch * 2 + h) % 4 == 3:
                            e.ins("v_add_u32 %[vn], 4096, %[vn]")
    issue_b(1, "Bn1", True)  # the next tile's second sub-sweep -> buffer 1 (read by the last sub-sweep, whose MFMAs have been issued)
    return e


def rename_exit(tags):
    """the exit state of a block as the next block sees it at entry"""
    out = []
    for t in tags:
        if t.startswith("Bn"):
            out.append("B" + t[2:])
        elif t.startswith("Ln"):
            out.append("L" + t[2:])
        else:
            out.append("S_prev")
    return out


OPERANDS = (': [a0] "=&v"(a0_), [a1] "=&v"(a1_), [vb] "=&v"(vb_), [vbn] "=&v"(vbn_), [vt] "=&v"(vt_), [vn] "=&v"(vn_)\n'
            '        : [tile] "s"(tile), [next] "s"(next), [fb] "s"(fb), [fbn] "s"(fbn), [ss] "s"(ss), [voff] "v"(voff), [lob] "v"(lob), [as] "v"(as)\n'
            '        : "memory", ' + ", ".join('"v%d"' % r for r in range(B_BASE, 256)))


def c_block(name, doc, e):
    body = "\\n\\t\"\n        \"".join(e.lines)
    return ("// %s\n__device__ __forceinline__ void %s(const double *tile, const double *next, const double *fb, const double *fbn, unsigned ss, unsigned voff, unsigned lob, unsigned as) {\n"
            "    double a0_, a1_;\n    unsigned vb_, vbn_, vt_, vn_;\n    asm volatile(\n        \"%s\"\n        %s);\n    (void)a0_, (void)a1_, (void)vb_, (void)vbn_, (void)vt_, (void)vn_;\n}\n\n" % (doc, name, body, OPERANDS))


def main():
    # the non-diagonal block's entry assumption is its own exit state (fixed point after one pass: the exit does not depend on the entry)
    probe = gen_block(False, [])
    entry = rename_exit(probe.inflight)
    nd = gen_block(False, entry)
    assert rename_exit(nd.inflight) == entry, "exit state is not a fixed point"
    dg = gen_block(True, [])  # a diagonal tile is entered behind a drain (first tile of its row): nothing assumed in flight
    # prologue: the first tile of a wave's walk: all chains, the first two sub-sweeps of B operands, then a drain
    p = Emit([])
    p.ins("v_mov_b32 %[vn], %[voff]")
    p.ins("v_mov_b32 %[vbn], %[lob]")
    for ch in range(16):
        for h in range(2):
            r = chain_reg(ch) + 4 * h
            p.vmem("global_load_dwordx4 a[%d:%d], %%[vn], %%[next] offset:%d" % (r, r + 3, ((ch * 2 + h) % 4) * 1024), "L")
            if (ch * 2 + h) % 4 == 3:
                p.ins("v_add_u32 %[vn], 4096, %[vn]")
    for s_ in range(2):
        for pl in range(2):
            for cc in range(4):
                p.vmem("global_load_dwordx2 v[%d:%d], %%[vbn], %%[fbn] offset:%d" % (b_reg(s_, pl, cc), b_reg(s_, pl, cc) + 1, cc * 512), "B")
            p.ins("v_add_u32 %[vbn], %[ss], %[vbn]")
    p.ins("s_waitcnt vmcnt(0)")
    head = ("// flush_pipe_agpr.h -- generated by scripts/r05_gen_flush_pipe.py: one tile of the 16-pair dense pass (k_flush_rows) as ONE software-pipelined asm statement\n"
            "// (tile in a128..a255; B operands two pairs at a time through a ring of three buffers in v208..v255, requested two sub-sweeps ahead; the\n"
            "// next tile's chains requested as this tile's row-blocks are stored; every s_waitcnt vmcnt computed from the issue order).  %d pairs.\n"
            "// pp_tile_nd: a non-diagonal tile; its entry assumption is its own exit state.  pp_tile_dg: a diagonal tile (chains below the diagonal\n"
            "// neither multiplied nor stored), entered and left behind a drain (the caller's s_waitcnt vmcnt(0)).  pp_prologue: the requests a first\n"
            "// tile needs (`next` / `fbn` name that tile), drained.  tile / next / fb / fbn are wave-uniform; voff = lane * 16, lob = lo * 8 (lo = the\n"
            "// lane's element of a 64-row operand block), as = LDS byte address of the wave's A stage + lo * 8, ss = bytes between two pairs.\n#pragma once\n\n" % NP)
    text = head
    text += c_block("pp_tile_nd", "%d instructions" % len(nd.lines), nd)
    text += c_block("pp_tile_dg", "%d instructions" % len(dg.lines), dg)
    text += c_block("pp_prologue", "%d instructions" % len(p.lines), p)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "2d-ekf-slam_amd", "csrc", "flush_pipe_agpr.h")
    if "--check" in sys.argv:
        same = os.path.exists(path) and open(path).read() == text
        print("flush_pipe_agpr.h %s the generator's output" % ("is" if same else "DIFFERS from"))
        return 0 if same else 1
    open(path, "w").write(text)
    if "--dump" in sys.argv:
        for l in nd.lines:
            print(l)
    return 0


if __name__ == "__main__":
    sys.exit(main())
