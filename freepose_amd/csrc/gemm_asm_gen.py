"""Generator of gemm_asm_loop.inc — the hand-scheduled main loop of the 256x256 persistent bf16 GEMM (gemm_asm.hip), gfx950.

    python freepose_amd/csrc/gemm_asm_gen.py > freepose_amd/csrc/gemm_asm_loop.inc        (the output is committed)

Why assembly (round 4; profiles/r04_gemm_asm_loop.log): with ONE wave per SIMD and the whole accumulator file (a[0:255]) holding a
128x128 wave tile, the loop reads half the LDS fragment bytes per flop of the 16-wave kernel — LDS reads cost ~14 % of the power at
the socket cap — and an in-order wave needs every MFMA gap planned: hipcc's schedule of the same geometry (tools/experiments/
gemm_w1.hip, round 1) ran at a third of the rate.

Geometry: 4 waves (2 x 2), wave tile 128 tokens x 128 features = 8 x 8 fragments of `v_mfma_f32_16x16x32_bf16`; the W fragment is the
MFMA A operand with the library's PERMUTED row map (fragment (grp, j), row 4 lg + r  <->  feature 64 grp + 16 lg + 4 j + r), so a lane's
accumulators a[4 (8 i + 4 grp + j) + r] are 16 consecutive features of token 16 i + li: exactly the block layout gemm_epilogue.h consumes.
K in stages of 64 (two 64 KiB LDS buffers); operands arrive by LDS-DMA (`buffer_load_dwordx4 ... offen lds`, 8 rows x 128 B per
wave-instruction), the bank swizzle applied to the lane's SOURCE offset (X: slot ^ (row & 7); W: slot ^ key_perm(row), the library's
key) and again on the fragment read.  Rows past the end of X / W are cut off by the buffer descriptor (num_records), which is why
every piece has its own voffset register (the range check sees voffset only).

Statements (all `asm volatile`; between them the compiler owns every register except the accumulator file):
  PROLOGUE  stages 0 and 1 of the workgroup's first tile
  INIT      acc[i][jj] = W-record[jj] x X-record[i]  (one MFMA per fragment, C = 0): bias / folded-LayerNorm start values
  TILE      the K loop of one tile.  On entry stages 0, 1 of the tile are in flight or landed; on exit the accumulators hold the tile
            and stages 0, 1 of the NEXT tile are in flight (their descriptors are inputs; the last tile passes its own again).

Step g (buffer b = g & 1); at its top: set A = fragments (g, k-half 0), stage g in buffer b, stage g+1 in flight into buffer b^1
  phase 0: 64 MFMA on set A | 16 ds_read (g, k-half 1) -> set B | s_waitcnt vmcnt(0) lgkmcnt(0), s_barrier
  phase 1: 64 MFMA on set B | 16 DMA pieces of stage g+2 -> buffer b | 16 ds_read (g+1, k-half 0) <- buffer b^1 -> set A | lgkmcnt(0)
Fillers follow the MFMAs at a fixed rate (one per two MFMAs in phase 0, one per MFMA in phase 1): measured best (r04_gemm_asm_loop.log).
"""
from __future__ import annotations

# Three geometries are generated (macros FP_GASM4_*, FP_GASM8_* and FP_GASMS_*):
#   "4": 256x256 tile, 4 waves (2 x 2, one per SIMD): wave tile 128 x 128, 256 accumulator registers, fragments / addresses in v[64:215]
#   "8": 256x256 tile, 8 waves (2 x 4, two per SIMD): wave tile 128 x 64,  128 accumulator registers, everything else in v[16:127] (256
#        registers per wave in total); the loop is 1-2 % slower (1.5x the fragment reads per flop) but two waves share each SIMD's
#        vector ALU in the epilogue
#   "S": 128x128 tile, 4 waves (2 x 2), wave tile 64 x 64, 64 accumulator registers, everything else in v[64:143] — the SMALL tier
#        (round 5): launches below the 256x256 tier run one or two workgroups per CU, i.e. one or two waves per SIMD, where hipcc's
#        schedule of the same loop leaves every phase of a K step (fragment reads, MFMAs, barrier, DMA issue) serial:
#        1670 clocks per step against 512 of MFMA time (profiles/r05_ab.md §2).  Same step structure as the big geometries at a quarter
#        of the size: per wave and step 32 MFMAs, 16 fragment reads, 8 DMA pieces.
TC = TR = PCS = NACC = NW = LDS_SHIFT = STORES = XTILE = 0
RATE0 = RATE1 = 1.0
S_RSX, S_RSW = 36, 40        # buffer descriptors of the tile being LOADED
S_DK, S_CNT, S_LX0, S_LW0, S_LX1, S_LW1 = 44, 45, 46, 47, 48, 49
S_LAST = 49
V_SETA_X = V_SETA_W = V_SETB_X = V_SETB_W = V_XP = V_WP = V_LAST = 0
V_AX, V_AW = {}, {}


def set_geometry(geo):
    global TC, TR, PCS, NACC, NW, LDS_SHIFT, STORES, XTILE, RATE0, RATE1, V_SETA_X, V_SETA_W, V_SETB_X, V_SETB_W, V_XP, V_WP, V_LAST
    nw = {"4": 4, "8": 8, "S": 4}[geo]
    NW = nw
    bt = 128 if geo == "S" else 256     # tile edge
    TC = bt // 2 // 16                  # X fragments per wave (two wave rows)
    TR = bt // (nw // 2) // 16          # W fragments per wave
    PCS = bt // 8 // nw                 # DMA pieces per operand, wave and stage
    NACC = 4 * TC * TR
    LDS_SHIFT = {128: 12, 256: 13}[bt * 4 // nw]   # log2 of a wave's bytes per operand and stage (bt / nw rows of 128 B)
    XTILE = bt * 128                    # bytes of one operand tile; a stage is X tile + W tile
    STORES = 2 * TC * (TR // 4)         # output stores per wave and tile (two per 16 x 64 block): the entry wait's count
    RATE0 = 0.5 if nw == 4 else 0.4     # fillers per MFMA in phase 0 (fragment reads) ...
    RATE1 = {"4": 1.0, "8": 1.2, "S": 1.6}[geo]   # ... and in phase 1 (DMA pieces + fragment reads; "S": 25 fillers behind 16 MFMAs)
    v0 = 16 if geo == "8" else 64
    V_SETA_X = v0
    V_SETA_W = V_SETA_X + 4 * TC
    V_SETB_X = V_SETA_W + 4 * TR
    V_SETB_W = V_SETB_X + 4 * TC
    V_XP = V_SETB_W + 4 * TR
    V_WP = V_XP + PCS
    base = V_WP + PCS
    for n, key in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        V_AX[key] = base + n
        V_AW[key] = base + 4 + n
    V_LAST = base + 7


def vr(b):
    return f"v[{b}:{b + 3}]"


def sr(b):
    return f"s[{b}:{b + 3}]"


def mfma(i, jj, xs, ws, c0=False):
    f = i * TR + jj
    acc = f"a[{4 * f}:{4 * f + 3}]"
    return f"v_mfma_f32_16x16x32_bf16 {acc}, {vr(ws + 4 * jj)}, {vr(xs + 4 * i)}, {'0' if c0 else acc}"


def frag_reads(buf, kh, xs, ws):
    """W fragments first (jj = 4 grp + j at byte offset grp * 8192 + j * 512), then X0..X7 (the MFMA order i outer, jj inner)"""
    out = [f"ds_read_b128 {vr(ws + 4 * jj)}, v{V_AW[(buf, kh)]} offset:{(jj >> 2) * 8192 + (jj & 3) * 512}" for jj in range(TR)]
    out += [f"ds_read_b128 {vr(xs + 4 * i)}, v{V_AX[(buf, kh)]} offset:{i * 2048}" for i in range(TC)]
    return out


def dma_groups(buf):
    g = []
    for q in range(PCS):
        g.append([f"s_mov_b32 m0, s{S_LX0 if buf == 0 else S_LX1}" if q == 0 else "s_add_u32 m0, m0, 0x400",
                  f"buffer_load_dwordx4 v{V_XP + q}, {sr(S_RSX)}, s{S_DK} offen lds"])
    for q in range(PCS):
        g.append([f"s_mov_b32 m0, s{S_LW0 if buf == 0 else S_LW1}" if q == 0 else "s_add_u32 m0, m0, 0x400",
                  f"buffer_load_dwordx4 v{V_WP + q}, {sr(S_RSW)}, s{S_DK} offen lds"])
    return g


def interleave(mfmas, fillers, rate):
    out, fi, acc = [], 0, 0.0
    for ins in mfmas:
        out.append(ins)
        acc += rate
        while acc >= 1.0 - 1e-9 and fi < len(fillers):
            out.append(fillers[fi]); fi += 1; acc -= 1.0
    out.extend(fillers[fi:])
    return out


def step(buf, preload_next=True):
    L = []
    m0 = [mfma(i, jj, V_SETA_X, V_SETA_W) for i in range(TC) for jj in range(TR)]
    L += interleave(m0, frag_reads(buf, 1, V_SETB_X, V_SETB_W), RATE0)
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    m1 = [mfma(i, jj, V_SETB_X, V_SETB_W) for i in range(TC) for jj in range(TR)]
    groups = dma_groups(buf)
    reads = frag_reads(buf ^ 1, 0, V_SETA_X, V_SETA_W) if preload_next else []
    fill = []
    nrd = len(reads)
    for q in range(2 * PCS):
        fill += groups[q]
        fill += reads[q * nrd // (2 * PCS):(q + 1) * nrd // (2 * PCS)]
    fill.append(f"s_add_u32 s{S_DK}, s{S_DK}, 0x80")
    L += interleave(m1, fill, (1.0 if preload_next else 0.6) * RATE1)
    L += ["s_waitcnt lgkmcnt(0)"]
    return L


def setup_pieces(vx0, vw0, sx8, sw8):
    """per-piece source offsets: X piece q = vX0 + q * 8 rows; W the same, its 16-byte slot additionally ^ 2 ((q >> 1) & 3) (key_perm)"""
    L = [f"v_mov_b32 v{V_XP}, {vx0}", f"v_mov_b32 v{V_WP}, {vw0}"]
    for q in range(1, PCS):
        L.append(f"v_add_u32 v{V_XP + q}, {sx8}, v{V_XP + q - 1}")
        L.append(f"v_add_u32 v{V_WP + q}, {sw8}, v{V_WP + q - 1}")
    for q in range(PCS):
        c = 32 * ((q >> 1) & 3)
        if c:
            L.append(f"v_xor_b32 v{V_WP + q}, {c}, v{V_WP + q}")
    return L


def setup_lds(lds_base, wave):
    return [f"s_lshl_b32 s{S_LX0}, {wave}, {LDS_SHIFT}", f"s_add_u32 s{S_LX0}, s{S_LX0}, {lds_base}", f"s_add_u32 s{S_LW0}, s{S_LX0}, {hex(XTILE)}",
            f"s_add_u32 s{S_LX1}, s{S_LX0}, {hex(2 * XTILE)}", f"s_add_u32 s{S_LW1}, s{S_LW0}, {hex(2 * XTILE)}"]


def descriptors(xlo, xhi, xrec, wlo, whi, wrec):
    return [f"s_mov_b32 s{S_RSX}, {xlo}", f"s_mov_b32 s{S_RSX + 1}, {xhi}", f"s_mov_b32 s{S_RSX + 2}, {xrec}", f"s_mov_b32 s{S_RSX + 3}, 0x00020000",
            f"s_mov_b32 s{S_RSW}, {wlo}", f"s_mov_b32 s{S_RSW + 1}, {whi}", f"s_mov_b32 s{S_RSW + 2}, {wrec}", f"s_mov_b32 s{S_RSW + 3}, 0x00020000"]


def prologue_stmt():
    # operands: 0-5 X lo, hi, records, W lo, hi, records | 6 strideX8 7 strideW8 8 lds_base 9 wave | 10 vX0 11 vW0
    L = ["s_nop 4"]
    L += descriptors("%0", "%1", "%2", "%3", "%4", "%5") + setup_pieces("%10", "%11", "%6", "%7") + setup_lds("%8", "%9")
    L.append(f"s_mov_b32 s{S_DK}, 0")
    for buf in (0, 1):
        for g in dma_groups(buf):
            L += g
        L.append(f"s_add_u32 s{S_DK}, s{S_DK}, 0x80")
    return L


def init_stmt():
    # operands: 0..TR-1 the W-side records (A operand, fragment jj), TR..TR+TC-1 the X-side records (fragment i)
    L = ["s_nop 1"]
    for i in range(TC):
        for jj in range(TR):
            f = i * TR + jj
            L.append(f"v_mfma_f32_16x16x32_bf16 a[{4 * f}:{4 * f + 3}], %{jj}, %{TR + i}, 0")
    return L


def tile_stmt():
    # operands: 0-5 THIS tile's X lo, hi, records, W lo, hi, records | 6-11 the NEXT tile's | 12 strideX8 13 strideW8 14 lds_base
    #           15 wave 16 K/128 - 1 (body iterations before the last) | 17 vX0 18 vW0 19 vAX 20 vAW | 21 drain (s)
    # Entry wait: stages 0, 1 of this tile are the OLDEST vector-memory operations in flight; behind them sit the previous tile's
    # epilogue (STORES = 32 / 16 output stores per wave in the 4- / 8-wave geometry, always issued: rows / columns past the end are clipped by the store's descriptor, not
    # branched around) and the next record loads.  vmcnt retires in order, so vmcnt(STORES) = "the stages have landed" without waiting
    # for the store burst to drain.  The workgroup's first tile has no epilogue behind its stages: operand 21 != 0 -> vmcnt(0).
    L = ["s_nop 4"]
    L += descriptors("%0", "%1", "%2", "%3", "%4", "%5") + setup_pieces("%17", "%18", "%12", "%13") + setup_lds("%14", "%15")
    L += [f"v_mov_b32 v{V_AX[(0, 0)]}, %19", f"v_xor_b32 v{V_AX[(0, 1)]}, 64, v{V_AX[(0, 0)]}",
          f"v_add_u32 v{V_AX[(1, 0)]}, {hex(2 * XTILE)}, v{V_AX[(0, 0)]}", f"v_add_u32 v{V_AX[(1, 1)]}, {hex(2 * XTILE)}, v{V_AX[(0, 1)]}",
          f"v_mov_b32 v{V_AW[(0, 0)]}, %20", f"v_xor_b32 v{V_AW[(0, 1)]}, 64, v{V_AW[(0, 0)]}",
          f"v_add_u32 v{V_AW[(1, 0)]}, {hex(2 * XTILE)}, v{V_AW[(0, 0)]}", f"v_add_u32 v{V_AW[(1, 1)]}, {hex(2 * XTILE)}, v{V_AW[(0, 1)]}"]
    L += ["s_cmp_eq_u32 %21, 0", "s_cbranch_scc1 L_nodrain_%=", "s_waitcnt vmcnt(0)", "L_nodrain_%=:", f"s_waitcnt vmcnt({STORES})", "s_barrier"]
    L += frag_reads(0, 0, V_SETA_X, V_SETA_W)
    L += ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 s{S_DK}, 0x100", f"s_mov_b32 s{S_CNT}, %16"]
    L += ["L_k_%=:"] + step(0) + step(1)
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 L_k_%="]
    # the last two steps load stages 0, 1 of the next tile
    L += descriptors("%6", "%7", "%8", "%9", "%10", "%11") + [f"s_mov_b32 s{S_DK}, 0"]
    L += step(0) + step(1, preload_next=False)
    L += ["s_nop 15", "s_nop 15"]          # MFMA results -> v_accvgpr_read in the epilogue
    return L


def emit(name, lines):
    print(f"#define {name} \\")
    for ln in lines:
        print(f'    "{ln}\\n\\t" \\')
    print('    ""')


def clob(vlo, vhi, slo, shi, acc):
    c = [f"v{i}" for i in range(vlo, vhi + 1)] + [f"s{i}" for i in range(slo, shi + 1)] + ([f"a{i}" for i in range(NACC)] if acc else [])
    return ", ".join(f'"{x}"' for x in c + ["memory", "scc"])


def main():
    print("// generated by freepose_amd/csrc/gemm_asm_gen.py — do not edit (the generator documents the loop)")
    for geo in ("4", "8", "S"):
        set_geometry(geo)
        p = f"FP_GASM{geo}"
        emit(f"{p}_PROLOGUE_TEXT", prologue_stmt())
        print(f"#define {p}_PROLOGUE_CLOBBERS " + clob(V_XP, V_WP + PCS - 1, S_RSX, S_LAST, False))
        emit(f"{p}_INIT_TEXT", init_stmt())
        print(f"#define {p}_INIT_CLOBBERS " + ", ".join(f'"a{i}"' for i in range(NACC)))
        emit(f"{p}_TILE_TEXT", tile_stmt())
        print(f"#define {p}_TILE_CLOBBERS " + clob(V_SETA_X, V_LAST, S_RSX, S_LAST, True))


if __name__ == "__main__":
    main()
