"""Generator of the hand-scheduled K loop of tools/gemm_asm/gemm_asm_lab.hip (round 4, VERDICT r03 item 1, stage 1).

Emits `loop_body.inc`: ONE inline-asm string — the whole persistent tile walk of a 256x256x64 bf16 GEMM workgroup with
  * 4 waves (one per SIMD), 128x128 outputs per wave, the 64 accumulator fragments (16x16 fp32) resident in a[0:255];
  * operands by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, 8 rows x 128 B) into two 64 KiB stage
    buffers; the LDS image is lane-linear, so the bank swizzle (16-byte slot ^ (row & 7)) is applied to the per-lane SOURCE
    address and again on the fragment-read address;
  * `v_mfma_f32_16x16x32_bf16` fed from two fragment register sets (A: k-half 0, B: k-half 1 of a stage); every MFMA of a
    phase is followed by the phase's fillers (fragment reads for the next phase, DMA pieces two stages ahead) at a fixed rate;
  * ONE barrier per K step (behind phase 0), `vmcnt(0)` only there (stage g+1, requested 1.7 phases earlier).

Step g (buffer b = g & 1), invariant at its top: set A = fragments (g, k-half 0); stage g is in buffer b; stage g+1 is in flight
into buffer b^1.
  phase 0: 64 MFMA on set A | 16 ds_read (g, k-half 1) -> set B | s_waitcnt vmcnt(0) lgkmcnt(0), s_barrier
  phase 1: 64 MFMA on set B | 16 DMA pieces of stage g+2 -> buffer b (all its readers passed the barrier),
                              16 ds_read (g+1, k-half 0) from buffer b^1 -> set A | s_waitcnt lgkmcnt(0)

python tools/gemm_asm/gen_loop.py [--rate0 N] [--rate1 N] > tools/gemm_asm/loop_body.inc
"""
from __future__ import annotations

import argparse

# ---- fixed registers -------------------------------------------------------------------------------------------------------
S = dict(
    rsX=32, rsW=36,          # buffer descriptors (4 each)
    tab=40,                  # pair: this workgroup's tile table (8 bytes per tile: x offset, w offset)
    out=42,                  # pair: check-mode dump base
    K2=44, stride8=45, nkh=46, ntile=47, store=48,
    dx=49, dw=50, dk=51,     # DMA cursor: tile offsets (incl. the wave's row offset) and k byte offset
    nx=52, nw=53,            # next tile's offsets (pair, loaded from the table)
    t=54, cnt=55, seq8=56,
    ldsX0=57, ldsW0=58, ldsX1=59, ldsW1=60,
    waveoff=61, tmp=62,
)
# geometry: 2 x NWN waves; a wave owns 128 rows of X (TC = 8 fragments) and 256 / NWN rows of W (TR fragments)
NWN = 2
TC, TR, NW, PCS = 8, 8, 4, 8           # PCS = DMA pieces per operand, wave and stage (32 / NW)
V0 = 64                                # first fixed VGPR
V_SETA_X = V_SETA_W = V_SETB_X = V_SETB_W = V_DMA = V_OUT = 0
V_AX, V_AW = {}, {}


def set_geometry(nwn):
    global NWN, TR, NW, PCS, V0, V_SETA_X, V_SETA_W, V_SETB_X, V_SETB_W, V_DMA, V_OUT
    NWN = nwn
    TR = 16 // nwn
    NW = 2 * nwn
    PCS = 32 // NW
    V0 = 64 if nwn == 2 else 16        # two waves per SIMD: everything (accumulators included) must fit 256 registers
    V_SETA_X = V0
    V_SETA_W = V_SETA_X + 4 * TC
    V_SETB_X = V_SETA_W + 4 * TR
    V_SETB_W = V_SETB_X + 4 * TC
    V_DMA = V_SETB_W + 4 * TR
    base = V_DMA + 1
    for n, key in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        V_AX[key] = base + n            # (buffer, k-half) -> LDS read base of the X fragments
        V_AW[key] = base + 4 + n
    V_OUT = base + 8


def s(name, off=0):
    return f"s{S[name] + off}"


def srange(name, n):
    return f"s[{S[name]}:{S[name] + n - 1}]"


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def mfma(i, j, xs, ws, first):
    f = i * TR + j
    acc = f"a[{4 * f}:{4 * f + 3}]"
    c = "0" if first else acc
    return f"v_mfma_f32_16x16x32_bf16 {acc}, {vr(xs + 4 * i)}, {vr(ws + 4 * j)}, {c}"


def frag_reads(buf, kh, xs, ws):
    """W fragments first, then X0..X7: the next phase's MFMA order (i outer, j inner) needs W0-7 + X0 first, X7 last"""
    out = []
    for j in range(TR):
        out.append(f"ds_read_b128 {vr(ws + 4 * j)}, v{V_AW[(buf, kh)]} offset:{j * 2048}")
    for i in range(TC):
        out.append(f"ds_read_b128 {vr(xs + 4 * i)}, v{V_AX[(buf, kh)]} offset:{i * 2048}")
    return out


def dma_pieces(buf):
    """16 pieces of the stage at the DMA cursor into buffer `buf`, then the cursor advance; each piece = [m0, soffset, load]"""
    groups = []
    for q in range(PCS):
        g = []
        g.append(f"s_mov_b32 m0, {s('ldsX0' if buf == 0 else 'ldsX1')}" if q == 0 else "s_add_u32 m0, m0, 0x400")
        g.append(f"s_add_u32 {s('t')}, {s('dx')}, {s('dk')}" if q == 0 else f"s_add_u32 {s('t')}, {s('t')}, {s('stride8')}")
        g.append(f"buffer_load_dwordx4 v{V_DMA}, {srange('rsX', 4)}, {s('t')} offen lds")
        groups.append(g)
    for q in range(PCS):
        g = []
        g.append(f"s_mov_b32 m0, {s('ldsW0' if buf == 0 else 'ldsW1')}" if q == 0 else "s_add_u32 m0, m0, 0x400")
        g.append(f"s_add_u32 {s('t')}, {s('dw')}, {s('dk')}" if q == 0 else f"s_add_u32 {s('t')}, {s('t')}, {s('stride8')}")
        g.append(f"buffer_load_dwordx4 v{V_DMA}, {srange('rsW', 4)}, {s('t')} offen lds")
        groups.append(g)
    adv = [
        f"s_add_u32 {s('dk')}, {s('dk')}, 0x80",
        f"s_cmp_eq_u32 {s('dk')}, {s('K2')}",
        f"s_cselect_b32 {s('dk')}, 0, {s('dk')}",
        f"s_cselect_b32 {s('dx')}, {s('nx')}, {s('dx')}",
        f"s_cselect_b32 {s('dw')}, {s('nw')}, {s('dw')}",
    ]
    return groups, adv


def interleave(mfmas, fillers, rate, start=0):
    """after MFMA number m >= start, `rate` fillers (a float rate r < 1 means one filler every 1/r MFMAs)"""
    out, fi, acc = [], 0, 0.0
    for m, ins in enumerate(mfmas):
        out.append(ins)
        if m >= start:
            acc += rate
            while acc >= 1.0 - 1e-9 and fi < len(fillers):
                out.append(fillers[fi]); fi += 1; acc -= 1.0
    out.extend(fillers[fi:])
    return out


ABLATE = set()   # measurement only (wrong numerics): "dma" drops the steady-state DMA loads, "read" the steady-state fragment reads


STAGGER = "none"   # "comb": wave w issues its DMA piece q behind MFMA 4q + w of phase 1; "block": behind MFMAs 16w .. 16w + 15


def place(mfmas, slots):
    """slots: {mfma index: [instructions issued right behind it]}"""
    out = []
    for m, ins in enumerate(mfmas):
        out.append(ins)
        out.extend(slots.get(m, []))
    return out


def step(buf, first, rate0, rate1, fold_next_tile, wave=0):
    lines = []
    # ---- phase 0: set A, reads of (g, k-half 1) -> set B ------------------------------------------------------------------
    m0 = [mfma(i, j, V_SETA_X, V_SETA_W, first) for i in range(TC) for j in range(TR)]
    lines += interleave(m0, [] if "read" in ABLATE else frag_reads(buf, 1, V_SETB_X, V_SETB_W), rate0)
    lines += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    if fold_next_tile:   # the table entry requested at the top of the tile has landed (lgkmcnt(0) above): add the wave's row offset
        lines += [f"s_add_u32 {s('nx')}, {s('nx')}, {s('waveoff')}", f"s_add_u32 {s('nw')}, {s('nw')}, {s('waveoff')}"]
    # ---- phase 1: set B, DMA of stage g+2 -> this buffer, reads of (g+1, k-half 0) from the other buffer -> set A -----------
    m1 = [mfma(i, j, V_SETB_X, V_SETB_W, False) for i in range(TC) for j in range(TR)]
    groups, adv = dma_pieces(buf)
    reads = frag_reads(buf ^ 1, 0, V_SETA_X, V_SETA_W)
    fill = []
    nrd = len(reads)
    for q in range(2 * PCS):
        fill += [g for g in groups[q] if not ("dma" in ABLATE and g.startswith("buffer_load"))]
        if "read" not in ABLATE:
            fill += reads[q * nrd // (2 * PCS):(q + 1) * nrd // (2 * PCS)]
    fill += adv
    if STAGGER == "none":
        lines += interleave(m1, fill, rate1)
    else:
        slots = {}
        for q in range(16):
            m = (4 * q + wave) if STAGGER == "comb" else (16 * wave + q)
            slots.setdefault(m, []).extend(g for g in groups[q] if not ("dma" in ABLATE and g.startswith("buffer_load")))
            if "read" not in ABLATE:
                mr = (4 * q + (wave + 2) % 4) if STAGGER == "comb" else ((16 * wave + 32 + q) % 64 if q < 8 or True else 0)
                slots.setdefault(mr % 64, []).append(reads[q])
        slots.setdefault(63, []).extend(adv)
        if STAGGER == "block":   # reads in ascending issue order: re-place them at 4 q + 2 (they must issue W0..7, X0..7 in order)
            for m in list(slots):
                slots[m] = [x for x in slots[m] if not x.startswith("ds_read")]
            if "read" not in ABLATE:
                for q in range(16):
                    slots.setdefault(4 * q + 2, []).append(reads[q])
        lines += place(m1, slots)
    lines += ["s_waitcnt lgkmcnt(0)"]
    return lines


def body(first, rate0, rate1, wave=0):
    return step(0, first, rate0, rate1, first, wave) + step(1, False, rate0, rate1, False, wave)


def prologue():
    L = []
    a = L.append
    # inputs -> fixed registers (operand numbering: see gemm_asm_lab.hip)
    a(f"s_mov_b32 {s('rsX', 0)}, %0"); a(f"s_mov_b32 {s('rsX', 1)}, %1"); a(f"s_mov_b32 {s('rsX', 2)}, %11"); a(f"s_mov_b32 {s('rsX', 3)}, 0x00020000")
    a(f"s_mov_b32 {s('rsW', 0)}, %2"); a(f"s_mov_b32 {s('rsW', 1)}, %3"); a(f"s_mov_b32 {s('rsW', 2)}, %12"); a(f"s_mov_b32 {s('rsW', 3)}, 0x00020000")
    a(f"s_mov_b32 {s('tab', 0)}, %4"); a(f"s_mov_b32 {s('tab', 1)}, %5")
    a(f"s_mov_b32 {s('out', 0)}, %6"); a(f"s_mov_b32 {s('out', 1)}, %7")
    a(f"s_mov_b32 {s('K2')}, %8"); a(f"s_lshl_b32 {s('stride8')}, %8, 3")
    a(f"s_lshr_b32 {s('nkh')}, %9, 1"); a(f"s_sub_u32 {s('nkh')}, {s('nkh')}, 1")
    a(f"s_mov_b32 {s('ntile')}, %10"); a(f"s_mov_b32 {s('store')}, %13")
    # wave-dependent: row offset of the wave's pieces (wave * 64 rows), LDS destinations of its pieces
    sh_rows, sh_lds = (6, 13) if NWN == 2 else (5, 12)   # rows of X / W a wave moves per stage: 64 or 32; its LDS bytes: 8 or 4 KiB
    a(f"s_lshl_b32 {s('tmp')}, %14, {sh_rows}"); a(f"s_mul_i32 {s('waveoff')}, {s('tmp')}, %8")
    a(f"s_lshl_b32 {s('ldsX0')}, %14, {sh_lds}"); a(f"s_add_u32 {s('ldsW0')}, {s('ldsX0')}, 0x8000")
    a(f"s_add_u32 {s('ldsX1')}, {s('ldsX0')}, 0x10000"); a(f"s_add_u32 {s('ldsW1')}, {s('ldsW0')}, 0x10000")
    a(f"v_mov_b32 v{V_DMA}, %15")
    a(f"v_mov_b32 v{V_AX[(0, 0)]}, %16"); a(f"v_xor_b32 v{V_AX[(0, 1)]}, 64, v{V_AX[(0, 0)]}")
    a(f"v_add_u32 v{V_AX[(1, 0)]}, 0x10000, v{V_AX[(0, 0)]}"); a(f"v_add_u32 v{V_AX[(1, 1)]}, 0x10000, v{V_AX[(0, 1)]}")
    a(f"v_mov_b32 v{V_AW[(0, 0)]}, %17"); a(f"v_xor_b32 v{V_AW[(0, 1)]}, 64, v{V_AW[(0, 0)]}")
    a(f"v_add_u32 v{V_AW[(1, 0)]}, 0x10000, v{V_AW[(0, 0)]}"); a(f"v_add_u32 v{V_AW[(1, 1)]}, 0x10000, v{V_AW[(0, 1)]}")
    a(f"v_mov_b32 v{V_OUT}, %18")
    # first tile's offsets
    a(f"s_load_dwordx2 {srange('nx', 2)}, {srange('tab', 2)}, 0x0")
    a("s_waitcnt lgkmcnt(0)")
    a(f"s_add_u32 {s('dx')}, {s('nx')}, {s('waveoff')}"); a(f"s_add_u32 {s('dw')}, {s('nw')}, {s('waveoff')}")
    a(f"s_mov_b32 {s('dk')}, 0"); a(f"s_mov_b32 {s('seq8')}, 8")
    a(f"s_mov_b32 {s('nx')}, {s('dx')}"); a(f"s_mov_b32 {s('nw')}, {s('dw')}")
    # stages 0 and 1 of the first tile (cursor ends at stage 2)
    for buf in (0, 1):
        groups, adv = dma_pieces(buf)
        for g in groups:
            L.extend(g)
        L.extend(adv)
    a(f"s_waitcnt vmcnt({2 * PCS})"); a("s_barrier")
    L.extend(frag_reads(0, 0, V_SETA_X, V_SETA_W))
    a("s_waitcnt lgkmcnt(0)")
    return L


def epilogue():
    L = ["s_nop 15", "s_nop 15"]
    for f in range(TC * TR):
        L.append(f"global_store_dwordx4 v{V_OUT}, a[{4 * f}:{4 * f + 3}], {srange('out', 2)}")
        L.append(f"v_add_u32 v{V_OUT}, 0x400, v{V_OUT}")
    L.append(f"v_add_u32 v{V_OUT}, {hex((NW - 1) * TC * TR * 1024)}, v{V_OUT}")
    L += ["s_nop 7"]
    return L


def program(rate0, rate1):
    L = prologue()
    if STAGGER == "none":
        return L + wave_program(rate0, rate1, 0, "")
    for w in (1, 2, 3):
        L.append(f"s_cmp_eq_u32 %14, {w}")
        L.append(f"s_cbranch_scc1 L_wave{w}_%=")
    for w in (0, 1, 2, 3):
        L.append(f"L_wave{w}_%=:")
        L += wave_program(rate0, rate1, w, f"w{w}")
        L.append("s_branch L_end_%=")
    L.append("L_end_%=:")
    return L


def wave_program(rate0, rate1, wave, tag):
    L = []
    L.append(f"L_tile{tag}_%=:")
    L.append(f"s_load_dwordx2 {srange('nx', 2)}, {srange('tab', 2)}, {s('seq8')}")
    L.append(f"s_add_u32 {s('seq8')}, {s('seq8')}, 8")
    L += body(True, rate0, rate1, wave)
    L.append(f"s_mov_b32 {s('cnt')}, {s('nkh')}")
    L.append(f"s_cmp_eq_u32 {s('cnt')}, 0")
    L.append(f"s_cbranch_scc1 L_kdone{tag}_%=")
    L.append(f"L_k{tag}_%=:")
    L += body(False, rate0, rate1, wave)
    L.append(f"s_sub_u32 {s('cnt')}, {s('cnt')}, 1")
    L.append(f"s_cmp_lg_u32 {s('cnt')}, 0")
    L.append(f"s_cbranch_scc1 L_k{tag}_%=")
    L.append(f"L_kdone{tag}_%=:")
    L.append(f"s_cmp_eq_u32 {s('store')}, 0")
    L.append(f"s_cbranch_scc1 L_nostore{tag}_%=")
    L += epilogue()
    L.append(f"L_nostore{tag}_%=:")
    L.append(f"s_sub_u32 {s('ntile')}, {s('ntile')}, 1")
    L.append(f"s_cmp_lg_u32 {s('ntile')}, 0")
    L.append(f"s_cbranch_scc1 L_tile{tag}_%=")
    L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return L


def clobbers():
    c = [f"v{i}" for i in range(V0, V_OUT + 1)] + [f"a{i}" for i in range(4 * TC * TR)] + [f"s{i}" for i in range(32, 64)]
    return c + ["memory", "scc", "vcc"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rate0", type=float, default=1.0)
    ap.add_argument("--rate1", type=float, default=2.0)
    ap.add_argument("--nwn", type=int, default=2, choices=[2, 4], help="waves along N: 2 = 4 waves of 128x128, 4 = 8 waves of 128x64")
    ap.add_argument("--stagger", default="none", choices=["none", "comb", "block"])
    ap.add_argument("--ablate", default="", help="comma list of dma, read (measurement only, wrong numerics)")
    a = ap.parse_args()
    ABLATE.update(x for x in a.ablate.split(",") if x)
    global STAGGER
    STAGGER = a.stagger
    set_geometry(a.nwn)
    L = program(a.rate0, a.rate1)
    print(f"// generated by tools/gemm_asm/gen_loop.py --rate0 {a.rate0} --rate1 {a.rate1}: {len(L)} lines; do not edit")
    print(f"#define FP_ASM_NWN {NWN}")
    print("#define FP_ASM_LOOP_TEXT \\")
    for ln in L:
        print(f'    "{ln}\\n\\t" \\')
    print('    ""')
    print("#define FP_ASM_LOOP_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()))


if __name__ == "__main__":
    main()
