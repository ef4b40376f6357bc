#!/usr/bin/env python3
"""Writes omni-recall-rag_amd/csrc/orr_screen_tile16_asm.inc: the hand-scheduled K-tile of screen_tile16_kernel
(orr_screen.hip) as assembler text -- 64 v_mfma_i32_16x16x64_i8 on FIXED accumulation registers (tile (i, j) of the wave's
8 x 8 accumulator tiles is a[4 (8 i + j) .. + 3]), the 16 fragment reads and the 8 LDS-DMA requests placed one per MFMA slot.
Fragment registers (operands of the asm statement, pinned): query fragments 0..3 v[192:207], 4..7 v[208:223], row fragments
v[224:255].  The K-tile's bookkeeping (ring stages, LDS addresses, request offsets) is part of the text as well.  Run from the repository root after changing the schedule; the output is committed."""
import os

A_LO, A_HI, FB = 192, 208, 224


def acc(i, j):
    n = 4 * (8 * i + j)
    return "a[%d:%d]" % (n, n + 3)


def vreg(base, k):
    return "v[%d:%d]" % (base + 4 * k, base + 4 * k + 3)


def mfma(i, j, first=False):
    a = vreg(A_LO, i) if i < 4 else vreg(A_HI, i - 4)
    # (the first K-tile of an output tile multiplies onto the constant 0: no zeroing of 256 registers in front of it)
    return "v_mfma_i32_16x16x64_i8 %s, %s, %s, %s" % (acc(i, j), a, vreg(FB, j), "0" if first else acc(i, j))


def request(k, nt):
    # M0 (the LDS base of the pieces) changes once, between pieces 3 and 4: the immediate offset moves both addresses;
    # pieces 4..7 take their source offset from the second offset register (4 KiB further on)
    out = []
    if k == 0:
        out = ["v_readfirstlane_b32 %[st], %[m0v]", "s_nop 3", "s_mov_b32 m0, %[st]", "s_nop 0"]   # ([st]: a scalar scratch register)
    if k == 4:
        out = ["s_add_u32 m0, %[st], 0x1000", "s_nop 0"]
    return out + ["global_load_lds_dwordx4 %%[%s], %%[src] offset:%d%s" % ("vo" if k < 4 else "vo2", (k & 3) * 1024, " nt" if nt else "")]


# The bookkeeping of a K-tile rides in the shadow of its MFMAs (an MFMA keeps the matrix core busy for 16 cycles, the wave may
# issue other instructions meanwhile); as compiler-made code BETWEEN the K-tiles' texts -- twenty dependent scalar
# instructions -- it left the matrix core idle for 230 cycles per K-tile (1,308 against the 1,076 the text takes by itself,
# tools/ktile16_rate.hip).  The state lives in VECTOR registers, operands of the statement (scalar in/out operands of an asm
# statement come back as divergent values in this compiler: "illegal VGPR to SGPR copy"):
#   [pa]           LDS address of this K-tile's query fragments 4..7;  [pan] / [pbn]: of the next K-tile's fragments;
#   [pat] / [pbt]  of the K-tile after that (made in the first half, taken over in the second); rings of 3 / 6 stages of 16 KiB
#                  from [pab] / [pbb] to [pae] / [pbe]
#   [m0v]          LDS address of the stage requested LAST, the same in every lane (advanced in the first half, M0 of the
#                  requests in the second; ring from [m0s] to [m0e])
#   [vo] / [vo2]   source offsets of the pieces 0..3 / 4..7 requested LAST (likewise)
STAGE = 16384


def ring_step(dst, cur, start, end):
    return ["v_add_u32 %%[%s], 0x%x, %%[%s]" % (dst, STAGE, cur), "v_cmp_eq_u32 vcc, %%[%s], %%[%s]" % (end, dst),
            "v_cndmask_b32 %%[%s], %%[%s], %%[%s], vcc" % (dst, dst, start)]


FIRST_HALF_EXTRA = dict(zip([(1, 0), (1, 1), (1, 2)], [[l] for l in ring_step("pbt", "pbn", "pbb", "pbe")]))
FIRST_HALF_EXTRA.update(zip([(1, 3), (1, 4), (1, 5)], [[l] for l in ring_step("pat", "pan", "pab", "pae")]))
FIRST_HALF_REQ = dict(zip([(2, 0), (2, 1), (2, 2)], [[l] for l in ring_step("m0v", "m0v", "m0s", "m0e")]))
FIRST_HALF_REQ[(2, 3)] = ["v_add_u32 %%[vo], 0x%x, %%[vo]" % STAGE]
FIRST_HALF_REQ[(2, 4)] = ["v_add_u32 %%[vo2], 0x%x, %%[vo2]" % STAGE]


def k_tile(req, nt, first=False):
    out = []
    # first half: query tiles 0..3 against all eight row tiles; the fragments of query tiles 4..7 arrive meanwhile
    for i in range(4):
        for j in range(8):
            if i == 0 and j == 6:
                # row fragments 6 and 7 were requested at the very end of the previous K-tile (counted wait there): they are
                # the two oldest of the five reads in flight here
                out.append("s_waitcnt lgkmcnt(3)")
            out.append(mfma(i, j, first))
            if i == 0 and j % 2 == 0:
                out.append("ds_read_b128 %s, %%[pa] offset:%d" % (vreg(A_HI, j // 2), (4 + j // 2) * 1024))
            out += FIRST_HALF_EXTRA.get((i, j), [])
            if req:
                out += FIRST_HALF_REQ.get((i, j), [])
    out += ["s_waitcnt vmcnt(%[wn])", "s_waitcnt lgkmcnt(0)", "s_barrier"]
    # second half, column by column: row fragment j is free after its column and is re-read (next K-tile) two columns later
    for j in range(8):
        for i in range(4, 8):
            out.append(mfma(i, j, first))
            if i == 4 and j >= 2:
                out.append("ds_read_b128 %s, %%[pbn] offset:%d" % (vreg(FB, j - 2), (j - 2) * 1024))
            if i == 5 and j < 4:
                out.append("ds_read_b128 %s, %%[pan] offset:%d" % (vreg(A_LO, j), j * 1024))
            if i == 6 and req:
                out += request(j, nt)
            if (i, j) == (7, 3):
                out.append("v_mov_b32 %[pa], %[pan]")                # (the last read through [pan] is behind MFMA (5, 3))
            if (i, j) == (5, 4):
                out.append("v_mov_b32 %[pan], %[pat]")
    out.append("ds_read_b128 %s, %%[pbn] offset:%d" % (vreg(FB, 6), 6 * 1024))
    out.append("ds_read_b128 %s, %%[pbn] offset:%d" % (vreg(FB, 7), 7 * 1024))
    out.append("v_mov_b32 %[pbn], %[pbt]")
    out.append("s_waitcnt lgkmcnt(2)")          # all but the last two reads (row fragments 6, 7: awaited where they are used)
    return out


def c_string(lines):
    return " \\\n    ".join('"%s\\n\\t"' % l for l in lines)


def render():
    out = []
    out.append("// GENERATED by tools/gen_tile16_asm.py -- do not edit; see that script for the schedule and the register plan.\n")
    out.append("#define ORR_T16_ZERO \\\n    " + c_string(["v_accvgpr_write_b32 a%d, 0" % n for n in range(256)]) + "\n")
    for name, req, nt in (("REQ", True, False), ("REQ_NT", True, True), ("NOREQ", False, False)):
        out.append("#define ORR_T16_KTILE_%s \\\n    %s\n" % (name, c_string(k_tile(req, nt))))
        out.append("#define ORR_T16_KTILE_FIRST_%s \\\n    %s\n" % (name, c_string(k_tile(req, nt, True))))
    # the same K-tile in two statements (up to and including the barrier / the rest): the epilogue's loads are requested
    # between them -- behind the K-tile's counted wait, which would otherwise wait for them as well
    lines = k_tile(False, False)
    cut = lines.index("s_barrier") + 1
    out.append("#define ORR_T16_KTILE_NOREQ_H1 \\\n    %s\n" % c_string(lines[:cut]))
    out.append("#define ORR_T16_KTILE_NOREQ_H2 \\\n    %s\n" % c_string(lines[cut:]))
    out.append("#define ORR_T16_ACC_CLOBBERS " + ", ".join('"a%d"' % n for n in range(256)) + "\n")
    return "".join(out)


def main():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.normpath(os.path.join(here, "..", "omni-recall-rag_amd", "csrc", "orr_screen_tile16_asm.inc"))
    text = render()
    if "--check" in sys.argv[1:]:
        # the Makefile's guard: the committed text must be what this script makes (the script is the source of the K-tile)
        with open(path) as f:
            have = f.read()
        if have != text:
            sys.stderr.write("%s is not what tools/gen_tile16_asm.py generates: run the script and commit its output\n" % path)
            raise SystemExit(1)
        return
    with open(path, "w") as f:
        f.write(text)
    print("wrote", path)


if __name__ == "__main__":
    main()
