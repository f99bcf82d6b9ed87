"""Generates aurora_amd/csrc/gemm_a4_loop.inc: the hand-scheduled main loop of the four-wave 256 x 256 bf16 GEMM tile
(linear_kernel_256a4, gemm_a4.hip) as ONE inline-asm block -- every instruction, register and wait placed here, nothing left
to hipcc's scheduler.

    python tools/gen_gemm_asm.py [variant ...]     writes aurora_amd/csrc/gemm_a4_loop.inc (variant 0; committed: the build does
                                                   not run this) / gemm_a4_loop_v<N>.inc (experiments, not committed)

Tile: 4 waves (one per SIMD, 512 registers each), wave tile 128 x 128 = 8 x 8 MFMA tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16);
256 accumulators in a[0:255]; K in units of 64 = two 32-wide stages, each an LDS image of gemm.hip's 256 x 256 kernels
([256 rows][64 B] per operand, 16-byte pieces XOR-swizzled), four images = two units in LDS.

What the schedule is built on (profiles/r06_gemm_a4_dev.log; cycles per MFMA of the loop from its own s_memtime stamps, 8192^3):
    MFMAs alone 16.3-16.5 | + the 32 ds_read_b128 of a unit, anywhere: 16.4 (free) | + 16 LDS-DMA pieces (buffer_load ... lds) behind
    every 8th / 4th / 2nd MFMA: 16.55 / 16.65 / 17.9 | 16 buffer_load_dwordx4 to registers behind every 4th / 2nd: 16.7 / 19.5 |
    16 ds_write_b128 behind every 4th MFMA: 19.0 (21 cycles each), consecutive 22.3; as 2 x b64 behind every 2nd: 18.4.
So the refill goes by LDS-DMA (a ds_write costs the wave that owns the matrix pipe ~21 cycles wherever it is put; the first form
of this loop, staging whole 128-byte lines through registers, ran at 20.4-21.9), spread over the MFMAs, never two memory
instructions behind one MFMA.

Per unit u (128 MFMAs per wave; registers named literally, map below) -- a ring of four stage images, three stages (96 KiB) in
flight, two barriers per unit:
  H1: 64 MFMAs on fragment set A (stage 2u); between them the 16 fragment reads of stage 2u+1 -> set B and the 8 LDS-DMA pieces
      of stage 2u+4 into the image of stage 2u (last read in H2 of the unit before).
      s_waitcnt vmcnt(16) (everything but the two youngest stages' pieces has landed: stage 2u+2) lgkmcnt(0); s_barrier.
  H2: 64 MFMAs on set B; the 16 fragment reads of stage 2u+2 -> set A, the 8 pieces of stage 2u+5 into the image of stage 2u+1.
      s_waitcnt vmcnt(16) lgkmcnt(0); s_barrier.
A piece has two halves (~2,100 cycles) to land.  (With all 16 pieces of a unit issued in H2 and one barrier per unit -- a piece
then has ONE half -- the loop ran at 21.4-22 cycles per MFMA: the wait at the barrier, not the issue.)
"""
import sys

# ---- register map (VGPRs; the compiler keeps its own values below V_FIRST) ----
V_FIRST = 48
VO = 48          # v[48:55]   buffer voffsets of a stage's 8 pieces (4 activation, 4 weight)
RX, RW = 56, 57  # fragment read bases (activation / weight tile), toggled between the two unit slots
XA, WA, XB, WB = 64, 96, 128, 160   # fragment sets: 8 fragments x 4 registers each
V_END = 192

ROW2 = 64
OPER2 = 256 * ROW2
STAGE2 = 2 * OPER2


def acc(j, q):
    """accumulator tile of weight fragment j (0..7) x activation fragment q (0..7): acc[h = j >> 2][fn = j & 3][fm = q] of the
    eight-wave kernels' epilogues"""
    return 4 * (((j >> 2) * 4 + (j & 3)) * 8 + q)


def mfma(wset, xset, j, q):
    a = acc(j, q)
    return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[{wset + 4 * j}:{wset + 4 * j + 3}], v[{xset + 4 * q}:{xset + 4 * q + 3}], a[{a}:{a + 3}]"


def read_x(dst, f, image):
    return f"ds_read_b128 v[{dst + 4 * f}:{dst + 4 * f + 3}], v{RX} offset:{image * STAGE2 + f * 16 * ROW2}"


def read_w(dst, j, image):
    return f"ds_read_b128 v[{dst + 4 * j}:{dst + 4 * j + 3}], v{RW} offset:{image * STAGE2 + (j >> 2) * 64 * ROW2 + (j & 3) * 4 * ROW2}"


def dma_piece(p):
    """piece p (0..15) of a unit: stage ks = p >> 3, operand (p >> 2) & 1 (0 activations, 1 weights), piece r = p & 3 of that
    operand's four: rows r * 64 + (tid >> 2) of the tile.  Two instructions: M0 = the wave's LDS destination, then the load."""
    ks, o, r = p >> 3, (p >> 2) & 1, p & 3
    rs = "%[rsw]" if o else "%[rsx]"
    k = "%[kb]" if ks else "%[ka]"
    return [f"s_add_u32 m0, %[ldsw], {ks * STAGE2 + o * OPER2 + r * 4096}",
            f"buffer_load_dwordx4 v{VO + 4 * o + r}, {rs}, {k} offen lds"]


def half(wset, xset, fillers):
    """64 MFMAs, activation fragment q outer; fillers[k] = instructions issued behind MFMA k"""
    out = []
    k = 0
    for q in range(8):
        for j in range(8):
            out.append(mfma(wset, xset, j, q))
            out.extend(fillers.get(k, []))
            k += 1
    return out


def build(variant):
    ab = VARIANTS[variant]
    lines = []
    emit = lines.append
    # ---------------- prologue: units 0 and 1 on their way, accumulators cleared, unit 0 published, fragments of stage 0 ----------------
    emit("s_memtime %[t0]")
    for r in range(4):
        emit(f"v_mov_b32 v{VO + r}, %[vox{r}]")
        emit(f"v_mov_b32 v{VO + 4 + r}, %[vow{r}]")
    emit(f"v_mov_b32 v{RX}, %[offx]")
    emit(f"v_mov_b32 v{RW}, %[offw]")
    emit("s_mov_b32 %[ka], 0")
    emit("s_mov_b32 %[kb], 64")
    for p in range(16):                       # unit 0 -> slot 0
        a, b = dma_piece(p)
        emit(a)
        emit("s_nop 0")
        emit(b)
    emit("s_min_u32 %[ka], 128, %[kmax]")
    emit("s_add_u32 %[kb], %[ka], 64")
    emit("s_xor_b32 %[ldsw], %[ldsw], 0x10000")
    for p in range(16):                       # unit 1 (or the last unit again) -> slot 1
        a, b = dma_piece(p)
        emit(a)
        emit("s_nop 0")
        emit(b)
    emit("s_xor_b32 %[ldsw], %[ldsw], 0x10000")
    emit("s_mov_b32 %[knext], 256")
    for a in range(256):
        emit(f"v_accvgpr_write_b32 a{a}, 0")
    emit("s_waitcnt vmcnt(16)")
    emit("s_barrier")
    for f in range(8):
        emit(read_w(WA, f, 0))
    for f in range(8):
        emit(read_x(XA, f, 0))
    emit("s_waitcnt lgkmcnt(0)")
    if ab.get("ring"):
        emit("s_barrier")          # (the first half refills stage 0's image: everybody has read it)
    emit("s_memtime %[t1]")
    # ---------------- the loop: one unit per iteration ----------------
    emit("1:")
    rs1, r01 = ab.get("rs1", 4), ab.get("r01", 0)
    rs2, r02 = ab.get("rs2", 4), ab.get("r02", 0)
    ds, d0 = ab.get("ds", 4), ab.get("d0", 2)
    rd1 = [read_w(WB, f, 1) for f in range(8)] + [read_x(XB, f, 1) for f in range(8)]
    rd2 = [read_w(WA, f, 0) for f in range(8)] + [read_x(XA, f, 0) for f in range(8)]
    fill = {}
    for k, ins in enumerate(rd1):
        fill.setdefault(r01 + rs1 * k, []).append(ins)
    # where the DMA of this iteration reads: unit u + 2, clamped to the last one
    emit("s_min_u32 %[ka], %[knext], %[kmax]")
    emit("s_add_u32 %[kb], %[ka], 64")
    emit("s_add_u32 %[knext], %[knext], 128")
    if ab.get("ring"):
        # Ring of four stages, two barriers per unit: H1 refills the image of stage 2u (read in H2 of the unit before) with stage
        # 2u + 4, H2 the image of stage 2u + 1 (read in H1) with stage 2u + 5 -- three stages (96 KiB) in flight, a piece has two
        # halves (~2,100 cycles) to land; at the end of a half everything but the two youngest stages' pieces has landed.
        for p in range(8):
            a, b = dma_piece(p)
            fill.setdefault(d0 + ds * p - 1, []).append(a)
            fill.setdefault(d0 + ds * p, []).append(b)
        assert all(0 <= k < 64 for k in fill), sorted(fill)
        lines.extend(half(WA, XA, fill))
        emit("s_waitcnt vmcnt(16) lgkmcnt(0)")
    else:
        lines.extend(half(WA, XA, fill))
        emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_barrier")
    emit(f"v_xor_b32 v{RX}, 0x10000, v{RX}")
    emit(f"v_xor_b32 v{RW}, 0x10000, v{RW}")
    fill = {}
    for k, ins in enumerate(rd2):
        fill.setdefault(r02 + rs2 * k, []).append(ins)
    for p in (range(8, 16) if ab.get("ring") else range(16)):
        a, b = dma_piece(p)
        q = p - 8 if ab.get("ring") else p
        fill.setdefault(d0 + ds * q - 1, []).append(a)      # M0 one MFMA ahead of its load
        fill.setdefault(d0 + ds * q, []).append(b)
    assert all(0 <= k < 64 for k in fill), sorted(fill)
    lines.extend(half(WB, XB, fill))
    if ab.get("ring"):
        emit("s_waitcnt vmcnt(16) lgkmcnt(0)")
        emit("s_barrier")
    emit("s_xor_b32 %[ldsw], %[ldsw], 0x10000")
    emit("s_sub_u32 %[count], %[count], 1")
    emit("s_cmp_lg_u32 %[count], 0")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_cbranch_scc1 1b")
    # ---------------- drain ----------------
    emit("s_memtime %[t2]")
    emit("s_waitcnt vmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    emit("s_nop 15")
    emit("s_nop 15")
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
    out = []
    out.append("// GENERATED by tools/gen_gemm_asm.py (variant %d: %s) -- do not edit; see that file for the schedule and the register map." % (variant, ab))
    out.append("#define A4_TEXT_%d \\" % variant)
    for l in lines:
        out.append(f'    "{l}\\n\\t" \\')
    out.append('    ""')
    if variant == 0:
        names = [f'"v{i}"' for i in range(V_FIRST, V_END)] + [f'"a{i}"' for i in range(256)]
        out.append("#define A4_CLOBBERS \\")
        for i in range(0, len(names), 16):
            out.append("    " + ", ".join(names[i:i + 16]) + (", \\" if i + 16 < len(names) else ""))
    return "\n".join(out) + "\n"


# variant 0 ships; the others are placement experiments of the same instructions (tools/gemm_a4_stamps.py)
VARIANTS = {
    0: dict(ring=1, rs1=4, r01=0, rs2=4, r02=0, ds=4, d0=2),   # four-stage ring, two barriers per unit; a DMA piece behind every 4th MFMA of the first half of each half
    1: dict(ring=1, rs1=4, r01=0, rs2=4, r02=0, ds=8, d0=2),   # ... every 8th (17.2-18.2 cycles per MFMA, wall clock within 1 % of variant 0)
    2: dict(ring=1, rs1=2, r01=1, rs2=2, r02=1, ds=8, d0=4),   # reads behind every 2nd MFMA (the same)
    3: dict(rs1=4, r01=0, rs2=4, r02=0, ds=4, d0=2),           # two units, one barrier per unit, all 16 pieces in H2: 21.4-22 cycles per MFMA (a piece has one half to land)
}

if __name__ == "__main__":
    from pathlib import Path
    dst = Path(__file__).resolve().parents[1] / "aurora_amd" / "csrc"
    which = [int(a) for a in sys.argv[1:]] or [0]
    for v in which:
        (dst / ("gemm_a4_loop.inc" if v == 0 else f"gemm_a4_loop_v{v}.inc")).write_text(build(v))
