#!/usr/bin/env python3
"""Generates tonal_wino63_kloop.h: the K loop of wino63v_nt_kernel (tonal_wino63.hip) as inline-asm text with PINNED registers.

Why (round 5): written with builtins, the steady-state loop compiled to exactly the hand-specified order, but the peeled K-steps at
the two ends of a tile did not - the register allocator gave the 128 accumulator registers other homes there (MFMAs whose
destination is not their third source, v_mov_b64 chains between the pieces, fragments read from LDS straight into scratch memory:
62 - 107 spilled VGPRs per instantiation) and the in-kernel stamps priced the first two K-steps of a tile at 7 200 - 12 800 cycles
against 6 144 of matrix work, the last two at up to 17 500 against 10 240.  Here every K-step of a tile is ONE asm statement whose
vector operands are tied to fixed registers, so the allocator has nothing to decide:

    acc[i]   v[16 i : 16 i + 15]          i < 8     (128 accumulator registers, v0 .. v127)
    faL[i]   v[128 + 4 i : 131 + 4 i]     i < 4     A fragments of transforms 0-3 (four k each)
    fbL[i]   v[144 + 4 i : 147 + 4 i]               B fragments of transforms 0-3
    faH[i]   v[160 + 4 i : 163 + 4 i]               A fragments of transforms 4-7 (carried across the closing barrier)
    fbH[i]   v[176 + 4 i : 179 + 4 i]               B fragments of transforms 4-7

The issue order is the one the tuned loop had (r04_kernel_notes.md 9.1, 9.9): the 8 reads of half-set L behind the barrier that
opened the step, the 16 carried MFMAs of half-set H with one LDS-DMA piece per two of them, the 16 MFMAs of L with one read of H
each (or per two), the closing wait + barrier in the middle of L's MFMAs so that eight MFMAs are left to cover the first reads of
the next step.  `s_waitcnt lgkmcnt(N)` in front of an MFMA is computed here from the reads still allowed to be in flight (LDS
operations of a wave return in order).

    python gen_wino63_kloop.py > tonal_wino63_kloop.h      (the Makefile does; the generated header is committed)
"""
A_PLANE, B_PLANE, B_BASE = 4096, 2048, 32768          # bytes per transform plane of a stage (128 hexes / 64 columns x 32 B)
H_A, H_B = 4 * A_PLANE, 4 * B_PLANE                   # half-set H = transforms 4-7


def acc(i):
    return f"v[{16 * i}:{16 * i + 15}]"


def frag(kind, i):          # kind: aL bL aH bH
    base = {"aL": 128, "bL": 144, "aH": 160, "bH": 176}[kind] + 4 * i
    return base


class Seq:
    def __init__(self):
        self.lines = []
        self.reads = []            # destination base registers of the ds_reads issued so far, in order

    def emit(self, s):
        self.lines.append(s)

    def read(self, kind, i):
        base = frag(kind, i)
        if kind[0] == "a":
            off = i * A_PLANE + (H_A if kind[1] == "H" else 0)
            self.emit(f"ds_read_b128 v[{base}:{base + 3}], %[a]" + (f" offset:{off}" if off else ""))
        else:
            off = B_BASE + i * B_PLANE + (H_B if kind[1] == "H" else 0)
            self.emit(f"ds_read_b128 v[{base}:{base + 3}], %[b] offset:{off}")
        self.reads.append(base)

    def need(self, bases):
        """wait until the reads into these registers have returned (no-op if they were waited for already)"""
        idx = [max(k for k, b in enumerate(self.reads) if b == base) for base in bases if base in self.reads]
        if not idx:
            return
        self.emit(f"s_waitcnt lgkmcnt({len(self.reads) - 1 - max(idx)})")
        # everything up to max(idx) has returned: forget it (so later needs of older reads emit nothing)
        self.reads = self.reads[max(idx) + 1:]

    def mfma(self, half, i, q, zero=False):
        a, b = frag("a" + half, i), frag("b" + half, i)
        self.need([a, b])
        d = acc(i + (4 if half == "H" else 0))
        self.emit(f"v_mfma_f32_32x32x2_f32 {d}, v{a + q}, v{b + q}, {'0' if zero else d}")

    def dma(self, n):
        """LDS-DMA piece n of a step: A pieces 0-3 (M0 = m0a + 1024 n), B pieces 4, 5 (M0 = m0b + 1024 (n - 4))"""
        if n < 4:
            self.emit(f"buffer_load_dwordx4 %[va{n}], %[ra], %[sa] offen lds")
        else:
            self.emit(f"buffer_load_dwordx4 %[vb{n - 4}], %[rb], %[sb] offen lds")
        # M0 for the next piece right behind this one (>= 1 instruction between an M0 write and the piece that uses it)
        if n in (0, 1, 2):
            self.emit("s_add_u32 m0, m0, 0x400")
        elif n == 3:
            self.emit("s_mov_b32 m0, %[m0b]")
        elif n == 4:
            self.emit("s_add_u32 m0, m0, 0x400")

    def text(self, name):
        body = " \\\n".join(f'  "{ln}\\n"' for ln in self.lines)
        return f"#define {name} \\\n{body}\n"


def kstep(kind, early=True, nh=4):
    """kind: first | step1 | norm | prelast | prelast_z | last;  nh: transforms in half-set H (4; 2 for the six-batch form of
    tl_conv1_wino63v_dgrad_nt, whose transforms 6 and 7 are zero: 24 MFMAs per K-step instead of 32)"""
    s = Seq()
    dma = kind in ("first", "step1", "norm")
    zero_h = kind in ("step1", "prelast_z")
    if kind == "first":
        # no carried half-set: L of step 0 starts the accumulators 0-3 from the zero constant; H of step 0 is read here and
        # runs in step 1.  The six pieces of step 2 ride between the MFMAs.
        for i in range(4):
            s.read("aL", i)
            s.read("bL", i)
        s.emit("s_mov_b32 m0, %[m0a]")
        order = [(q, i) for q in range(4) for i in range(4)]
        hreads = [(k, i) for i in range(nh) for k in ("aH", "bH")]
        piece = 0
        for n, (q, i) in enumerate(order):
            s.mfma("L", i, q, zero=(q == 0))
            if n % 2 == 1:
                if hreads:
                    s.read(*hreads.pop(0))
                if piece < 6:
                    s.dma(piece)
                    piece += 1
        while hreads:
            s.read(*hreads.pop(0))
        s.emit("s_waitcnt %[w]")                       # vmcnt(stores of the epilogue in front + 6) lgkmcnt(0)
        s.emit("s_barrier")
        return s
    for i in range(4):
        s.read("aL", i)
        s.read("bL", i)
    # ---- the carried half-set H(s - 1): 16 MFMAs, one LDS-DMA piece per two of them from the second pair on
    if dma:
        s.emit("s_mov_b32 m0, %[m0a]")
    piece = 0
    order = [(q, i) for q in range(4) for i in range(4)]
    order_h = [(q, i) for q in range(4) for i in range(nh)]
    # (the reads of L are in flight: H's fragments were waited for before the closing barrier of the step in front)
    for n, (q, i) in enumerate(order_h):
        a, b = frag("aH", i), frag("bH", i)
        d = acc(i + 4)
        s.emit(f"v_mfma_f32_32x32x2_f32 {d}, v{a + q}, v{b + q}, {'0' if (zero_h and q == 0) else d}")
        # (nh = 2: eight carried MFMAs - a piece behind every one from the second on)
        if dma and (n % 2 == 1 or (nh < 4 and n >= 1)) and piece < 6:
            s.dma(piece)
            piece += 1
    assert not dma or piece == 6
    # ---- L(s): 16 MFMAs, the reads of H(s) between them; the closing wait + barrier after the eighth
    hreads = [(k, i) for i in range(nh) for k in ("aH", "bH")]
    for n, (q, i) in enumerate(order):
        s.mfma("L", i, q)
        if early or n % 2 == 1:
            if hreads:
                s.read(*hreads.pop(0))
        if n == (7 if early else 15) and kind != "last":
            while hreads:                              # (all eight are out by now)
                s.read(*hreads.pop(0))
            s.emit("s_waitcnt vmcnt(6) lgkmcnt(0)" if kind in ("step1", "norm") else "s_waitcnt vmcnt(0) lgkmcnt(0)")
            s.reads = []
            s.emit("s_barrier")
    if kind == "last":
        while hreads:
            s.read(*hreads.pop(0))
        # the tile's last half-set: nothing carries it
        for q, i in order_h:
            s.mfma("H", i, q)
        # the epilogue's first vector instruction may read an accumulator: XDL write -> VALU read hazard (the compiler's
        # hazard recogniser does not look into asm statements)
        s.emit("s_nop 15")
        s.emit("s_nop 3")
    return s


def issue_only():
    s = Seq()
    s.emit("s_mov_b32 m0, %[m0a]")
    for n in range(6):
        s.dma(n)
    return s


def main():
    out = ["// GENERATED by gen_wino63_kloop.py - do not edit (the generator's docstring has the register map and the reasons)",
           "#pragma once", ""]

    def emit_set(prefix, nh):
        na = 4 + nh

        def accs(lo, hi, mode):
            return [f'"{mode}{{{acc(i)}}}"((A)[{i}])' for i in range(lo, hi)]

        def frags(mode):
            r = []
            for kind, var in (("aL", "FAL"), ("bL", "FBL"), ("aH", "FAH"), ("bH", "FBH")):
                for i in range(4 if kind[1] == "L" else nh):
                    b = frag(kind, i)
                    r.append(f'"{mode}{{v[{b}:{b + 3}]}}"(({var})[{i}])')
            return r

        def define(name, regs, comment):
            out.append("// " + comment)
            out.append(f"#define {name}(A, FAL, FBL, FAH, FBH) \\\n  " + ", \\\n  ".join(regs) + "\n")

        # What a statement declares decides what the allocator must keep alive ACROSS THE EPILOGUE: an operand that is read
        # ("+") by the first statement of a tile would hold its register through the whole epilogue of the tile in front.
        define(f"{prefix}_REGS", accs(0, na, "+") + frags("+"), "steady state: every accumulator and fragment is read and written")
        define(f"{prefix}_REGS_FIRST", accs(0, 4, "=&") + frags("=&"),
               "first K-step of a tile: accumulators 0-3 start from zero, all fragments are (re)read - nothing is live on entry")
        define(f"{prefix}_REGS_STEP1", accs(0, 4, "+") + accs(4, na, "=&") + frags("+"),
               "second K-step: the accumulators of half-set H start from zero")
        for early in (True, False):
            sfx = "_E" if early else "_L"
            for kind in ("step1", "norm", "prelast", "prelast_z", "last"):
                out.append(kstep(kind, early, nh).text(f"{prefix}_{kind.upper()}{sfx}"))
        out.append(kstep("first", True, nh).text(f"{prefix}_FIRST"))

    emit_set("V6K", 4)
    out.append("// ---- the six-batch form (transforms 6, 7 are zero: tl_conv1_wino63v_dgrad_nt): half-set H = transforms 4, 5")
    emit_set("V6K6", 2)
    out.append(issue_only().text("V6K_ISSUE"))
    print("\n".join(out))


if __name__ == "__main__":
    main()
