// bf16 MFMA GEMM with fused ViT epilogues, gfx950 only.
//
// Structure (one workgroup = BM x BN output tile, BK = 64 per pipeline stage):
//   * both operand tiles are staged HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip);
//     the LDS image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and
//     again on the ds_read address (same involution on both sides);
//   * tiles are [rows][64 bf16] = 128-B rows; the 16-B slot index is XORed with a 3-bit key that is
//     distinct for the rows one ds_read_b128 lane group touches -> conflict-free fragment reads;
//   * MFMA is v_mfma_f32_16x16x32_bf16 in the SWAPPED form: the operand whose index must end up
//     contiguous in a lane's accumulators (output features n for the normal epilogues, tokens m for
//     the transposed V store) goes into the A slot with a PERMUTED row->fragment map, so one lane owns
//     16 (or 32) consecutive outputs and the epilogue stores full 16-B vectors / 128-B lines;
//   * double-buffered LDS, one barrier per K tile, next tile's DMA issued before the MFMA block;
//   * XCD-aware tile order: consecutive tiles of one X row-panel stay on one XCD's L2.
//
// Reference op being replaced: the nn.Linear calls inside the hub DINOv2 blocks driven by
// src/pipeline/retrieval/dino.py:16-23 (patch_embed.proj, attn.qkv, attn.proj, mlp.fc1, mlp.fc2).
#include "gemm_bf16.h"
#include "gemm_epilogue.h"
#include "internal.h"

#include <mutex>

// The PRODUCT build (libfreepose_hip.so) instantiates exactly one kernel per (epilogue, tile tier) — the variants below that the
// rounds' A/B runs selected — reads no environment variable and has no measurement hooks.  The LAB build (-DFP_LAB,
// libfreepose_hip_lab.so, loaded only by tools/) additionally compiles the alternative main loops / tile shapes and selects them
// through fp_lab_set_option("gemm_variant" / "gemm_dbg") or FP_GEMM_VARIANT / FP_GEMM_DBG.
#ifdef FP_LAB
#include <stdlib.h>
#endif
#define FP_GEMM_DEFAULT_VARIANT 238   // 2|4|8|32|64|128 (profiles/r02_ab.md: +4..6 % over 110 on every ViT shape)
#define FP_GEMM_VAR_BIG (4 | 32 | 64 | 128)   // 16-wave 256x256: table GELU, persistent walk, streaming epilogue I/O, split DMA issue
#define FP_GEMM_STREAM_BYTES (128L << 20)     // big-tier outputs above this are stored non-temporally (launch_epi: streaming policy)
#define FP_GEMM_VAR_SMALL 6                   // 128x128 (4 waves): pipelined fragment reads, table GELU
#define FP_GEMM_VAR_TINY (6 | 1024)           // 64x64 (2 waves, 1 for the transposed store): the same on a K-tile ring of run-time depth

namespace {

constexpr int BK = 64;         // bf16 per K stage  (128-byte LDS rows)
constexpr long BIG_MIN_TILES = 192;   // fewer 256x256 tiles than this never run on the big tier (launch_epi, fp_gemm_fuses_ln_part)
constexpr int ROWB = BK * 2;   // bytes per LDS row

__device__ __forceinline__ int key_plain(int row) { return (row >> 1) & 7; }
template <int T>
__device__ __forceinline__ int key_perm(int row) {
    const int rl = row % (16 * T);
    const int a = rl / (4 * T);
    const int b = rl & 3;
    return ((a << 1) | (b >> 1)) & 7;
}

template <int BM, int BN, int WM, int WN, int EPI, int VAR>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(FpGemmArgs p) {
    constexpr bool TRANS = FpEpiTraits<EPI>::TRANS;
    constexpr bool LNF = FpEpiTraits<EPI>::LN;     // LayerNorm folded into this GEMM's epilogue (gemm_bf16.h)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16;  // 16-row fragments of X per wave
    constexpr int TN = BN / WN / 16;  // 16-row fragments of W per wave
    // R operand = MFMA A slot (permuted rows, lane-contiguous outputs); C operand = B slot (plain)
    constexpr int TR = TRANS ? TM : TN;
    constexpr int TC = TRANS ? TN : TM;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int IX = BM / 8 / NW;  // glds instructions per wave per stage, X tile
    constexpr int IW = BN / 8 / NW;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over waves");

    // dynamic LDS: [GELU table, 16 KiB, table-GELU kernels only][K-tile buffer 0][K-tile buffer 1][epilogue slabs, 2 KiB per wave].
    // The table-GELU kernels keep no slab region: the 16-wave 256x256 kernel has no room for table + slabs (16 + 128 + 32 KiB),
    // and the 4-wave 128x128 kernel would drop from two resident workgroups per CU to one (88 KiB instead of 80).  Their slabs
    // live in the K-tile buffer the LAST K step consumed (free during the epilogue: the next tile's first stage is prefetched
    // into the other one), behind one extra barrier per tile.
    constexpr bool LUT = FpEpiTraits<EPI>::GELU && (VAR & 4) != 0;
    constexpr bool RELOC = LUT;
    constexpr int TAB = LUT ? fp_gemm::GELU_TAB_BYTES : 0;
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    char* const smem = smem_raw + TAB;                                     // K-tile buffers start here

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef FP_LAB
    // lab build, gemm_dbg = 1024: phase timestamps (100 MHz wall clock) of every workgroup's wave 0 into the scratch buffer —
    // entry, first K tile landed, K loop done, epilogue done (tools/step_ablate.py phases)
    unsigned long long dbg_t[4] = {0, 0, 0, 0}, dbg_c0 = 0;
    if (FP_GEMM_DBG_BIT(p, 1024)) { dbg_t[0] = wall_clock64(); dbg_c0 = __builtin_amdgcn_s_memtime(); }
#endif
    // The epilogue's row-coalescing slabs are a STATIC shared array of their own: the K-tile buffers are filled by LDS-DMA, and for LDS
    // accesses that may alias a DMA destination hipcc waits until the copy has landed (vmcnt(0)) — carved out of the same dynamic
    // array, the slabs made the first ds_write of every epilogue wait for the next tile's prefetch.  Distinct LDS variables carry
    // no-alias scopes through the LDS lowering.  (The table-GELU kernels declare none: their slabs live in a K-tile buffer, and
    // their table must sit at LDS address 0.)
    char* epi_stage = nullptr;                                              // row-coalescing slab of the epilogue
    if constexpr (!RELOC) {
        __shared__ __attribute__((aligned(1024))) char slab_mem[NW * fp_gemm::EPI_STAGE_BYTES];
        epi_stage = slab_mem + wave * fp_gemm::EPI_STAGE_BYTES;
    }
    if constexpr (LUT) {
        // gelu_tab16's inline-asm gathers use table-relative byte offsets as absolute LDS addresses
        if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem_raw != 0u) __builtin_trap();
        for (int o = tid * 16; o < fp_gemm::GELU_TAB_BYTES; o += NW * 64 * 16)
            *(uint4*)(smem_raw + o) = *(const uint4*)((const char*)p.gelu_tab + o);
        __syncthreads();   // (the pipelined loop's raw s_barrier does not wait for LDS writes)
    }
    // slab of this wave when the slabs share a K-tile buffer (RELOC): `buf` = the buffer the tile's last K step read
    auto reloc_stage = [&](int buf) {
        __syncthreads();                                                    // every wave is done reading that buffer
        return smem + buf * STAGE + wave * fp_gemm::EPI_STAGE_BYTES;
    };

    // ---- XCD-aware tile order (bijective for any grid size).  PERSIST: a resident grid walks the tiles t = block,
    // block + grid, ...; the grid is a multiple of 8, so a workgroup's tiles keep its XCD under the same remap.
    constexpr bool PERSIST = (VAR & 32) != 0;
    // SK (VAR bit 2048): the balanced tier.  The grid's G workgroups share the launch's U = ntiles x (K / 64) units evenly: workgroup w
    // (in XCD-contiguous logical order) walks the units [u, u1) of the sequence "tile 0's K steps, tile 1's K steps, ..." — whole tiles
    // end in the ordinary epilogue, a tile whose K range it shares with its neighbours goes through fp32 partial tiles (below).
    constexpr bool SK = (VAR & 2048) != 0;
    static_assert(!SK || ((VAR & 2) != 0 && !PERSIST), "the balanced tier is built on the software-pipelined loop");
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int m0, n0;

    // ---- per-lane DMA source offsets ------------------------------------------------------------
    uint32_t offX[IX], offW[IW];
    auto set_tile = [&](int t, int& tm0, int& tn0) {
        int tile_m, tile_n;
        if constexpr (SK) fp_gemm_tile_of_id<4>(t, tiles_m, tiles_n, tile_m, tile_n);   // t is already a logical tile id
        else fp_gemm_tile<(VAR & 1024) ? 16 : (VAR & 512) ? 8 : 4>(t, ntiles, tiles_m, tiles_n, tile_m, tile_n);   // (lab: 512 / 1024 = strips of 8 / 16 n-tiles)
        tm0 = tile_m * BM;
        tn0 = tile_n * BN;
#pragma unroll
        for (int it = 0; it < IX; ++it) {
            const int row = (it * NW + wave) * 8 + (lane >> 3);
            const int key = TRANS ? key_perm<TM>(row) : key_plain(row);
            const int ks = (lane & 7) ^ key;
            const int rg = min(tm0 + row, p.M - 1);
            offX[it] = (uint32_t)rg * (uint32_t)p.ldx * 2u + ks * 16;
        }
#pragma unroll
        for (int it = 0; it < IW; ++it) {
            const int row = (it * NW + wave) * 8 + (lane >> 3);
            const int key = TRANS ? key_plain(row) : key_perm<TN>(row);
            const int ks = (lane & 7) ^ key;
            const int rg = min(tn0 + row, p.N - 1);
            offW[it] = (uint32_t)rg * (uint32_t)p.ldw * 2u + ks * 16;
        }
    };
    int tile = blockIdx.x;
    if constexpr ((VAR & 32) != 0) {
        // experiment (FP_GEMM_DBG=32): stagger the resident workgroups over one tile period so that the chip's 256 epilogues (each a
        // 128 KiB store burst) do not all hit the memory system at the same moment
        if (FP_GEMM_DBG_BIT(p, 32)) {
            const int phase = (blockIdx.x >> 3) & 7;
            const int n = phase * (p.K / BK) * 6;      // ~ phase/8 of a tile: a K step is ~3 000 cycles = 48 x s_sleep(1)
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
        }
    }
    int sk_w = 0, sk_base = 0, sk_rem = 0, sk_u = 0, sk_u1 = 0;
    if constexpr (SK) {
        const int G = gridDim.x, U = ntiles * (p.K / BK);
        sk_w = fp_gemm_xcd_remap(blockIdx.x, G);
        sk_base = U / G;
        sk_rem = U - sk_base * G;
        sk_u = sk_w * sk_base + min(sk_w, sk_rem);
        sk_u1 = sk_u + sk_base + (sk_w < sk_rem ? 1 : 0);
        if (sk_u >= sk_u1) return;                        // (the launcher never asks for more workgroups than units)
    } else {
        set_tile(tile, m0, n0);
    }
    const char* gX = (const char*)p.X;
    const char* gW = (const char*)p.W;

    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * STAGE;
        const size_t kb = (size_t)kt * ROWB;
#pragma unroll
        for (int it = 0; it < IX; ++it)
            glds16(gX + offX[it] + kb, sb + (it * NW + wave) * 1024);
#pragma unroll
        for (int it = 0; it < IW; ++it)
            glds16(gW + offW[it] + kb, sb + BM * ROWB + (it * NW + wave) * 1024);
    };

    auto stage_x = [&](int buf, int kt) {
        char* sb = smem + buf * STAGE;
        const size_t kb = (size_t)kt * ROWB;
#pragma unroll
        for (int it = 0; it < IX; ++it) glds16(gX + offX[it] + kb, sb + (it * NW + wave) * 1024);
    };
    auto stage_w = [&](int buf, int kt) {
        char* sb = smem + buf * STAGE;
        const size_t kb = (size_t)kt * ROWB;
#pragma unroll
        for (int it = 0; it < IW; ++it) glds16(gW + offW[it] + kb, sb + BM * ROWB + (it * NW + wave) * 1024);
    };

    // ---- per-lane fragment read offsets (bytes inside a stage, before the k-step XOR) -----------
    const int li = lane & 15, lg = lane >> 4;
    // R operand rows: permuted  rl = (li>>2)*4*TR + 4*f + (li&3)
    // C operand rows: plain     rl = 16*f + li
    int rowR0, keyR, rowC0, keyC;
    {
        const int tile_r = TRANS ? wm * (16 * TM) : wn * (16 * TN);
        const int tile_c = TRANS ? wn * (16 * TN) : wm * (16 * TM);
        rowR0 = tile_r + (li >> 2) * 4 * TR + (li & 3);  // + 4*f
        keyR = (((li >> 2) << 1) | ((li & 3) >> 1)) & 7;  // = key_perm(row) for every f
        rowC0 = tile_c + li;                              // + 16*f
        keyC = (li >> 1) & 7;                             // key_plain(16 f + li) = (li>>1)&7 | ((16f>>1)&7)=0
    }
    const int baseR = (TRANS ? 0 : BM * ROWB) + rowR0 * ROWB;
    const int baseC = (TRANS ? BM * ROWB : 0) + rowC0 * ROWB;

    // Accumulators start at the BIAS of their output feature instead of zero (normal epilogues): the 64 v_mov per wave and tile
    // are needed either way, and the epilogue loses its 64 v_add + ~40 unpack instructions per wave and tile (vector ALU work
    // cannot hide under the matrix pipe on gfx950).  Same-box A/B of two builds: qk +0.9 %, proj +2.0 %, fc2 +0.7 %, fc1 0.
    // Layout (gemm_epilogue.h): acc[i][4 grp + j][r] is feature n0 + 64 (wn + grp) + 16 lg + 4 j + r for every row block i.
    f32x4_t acc[TC][TR];
    auto init_acc = [&](int m0_of_init, int tn0) {
        // The accumulators start at bias[n] (plain epilogues) or at b'[n] sigma[m] - mean[m] cs[n] (LN-folded ones, gemm_bf16.h) — formed
        // by ONE MFMA per accumulator block from 16-byte operand records (only the k-chunk of lanes lg == 0 is non-zero) with the inline
        // constant 0 as C: no accumulator zeroing, no unpacking, no vector-ALU arithmetic at the tile boundary, where every VALU
        // instruction is paid four times (the SIMD's four waves in lock-step, matrix pipe idle).  bias * 1.0 is exact.
        const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
        const bf16x8_t zfrag = __builtin_bit_cast(bf16x8_t, make_uint4(0u, 0u, 0u, 0u));
        // feature index / token index of fragment f of the R operand (permuted rows) and of the C operand (plain rows)
        const int rperm = (li >> 2) * 4 * TR + (li & 3);           // + 4 f
        bf16x8_t fr0[TR], fc0[TC];
        if constexpr (LNF) {
            const int featw = tn0 + wn * (16 * TN), tokw = m0_of_init + wm * (16 * TM);
#pragma unroll
            for (int f = 0; f < TR; ++f) {
                const int idx = TRANS ? min(tokw + rperm + 4 * f, p.M - 1) : min(featw + rperm + 4 * f, p.N - 1);
                fr0[f] = lg == 0 ? __builtin_bit_cast(bf16x8_t, (TRANS ? p.ln_mfrag : p.ln_cfrag)[idx]) : zfrag;
            }
#pragma unroll
            for (int f = 0; f < TC; ++f) {
                const int idx = TRANS ? min(featw + 16 * f + li, p.N - 1) : min(tokw + 16 * f + li, p.M - 1);
                fc0[f] = lg == 0 ? __builtin_bit_cast(bf16x8_t, (TRANS ? p.ln_cfrag : p.ln_mfrag)[idx]) : zfrag;
            }
        } else if constexpr (!TRANS) {
            const int featw = tn0 + wn * (16 * TN);
#pragma unroll
            for (int f = 0; f < TR; ++f) {
                const int idx = min(featw + rperm + 4 * f, p.N - 1);
                fr0[f] = lg == 0 ? __builtin_bit_cast(bf16x8_t, make_uint4((uint32_t)p.bias[idx], 0u, 0u, 0u)) : zfrag;
            }
            const bf16x8_t onef = lg == 0 ? __builtin_bit_cast(bf16x8_t, make_uint4(0x3f80u, 0u, 0u, 0u)) : zfrag;
#pragma unroll
            for (int f = 0; f < TC; ++f) fc0[f] = onef;
        }
        if constexpr (LNF || !TRANS) {
            if (LNF && FP_GEMM_DBG_BIT(p, 2)) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TR; ++j) acc[i][j] = zero4;
            } else {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr0[j], fc0[i], zero4, 0, 0, 0);
            }
        } else {   // transposed V store: measured slower with the bias in the accumulators (-3.7 %: spills), it adds it in the epilogue
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TR; ++j) acc[i][j] = zero4;
        }
    };
    auto zero_acc = [&]() { init_acc(m0, n0); };   // (name kept: every loop form re-arms the accumulators through it)
    const int nkt = p.K / BK;
    constexpr bool RING = (VAR & 2) && (VAR & 1024);          // K-tile ring of run-time depth (software-pipelined loop, 64x64 tier)
    int last_slot = (nkt - 1) & 1;                            // ring slot the last K step read (RELOC slabs live there)
    const int NS = RING ? p.ring : 2;                         // ring depth of the software-pipelined loop
    // K range of the current segment: the whole tile, or (SK) the part of a tile inside this workgroup's unit range
    int kt0 = 0, nk = nkt, sk_tile = 0;
    bool sk_more = false;
    do {
    if constexpr (SK) {
        sk_tile = sk_u / nkt;
        kt0 = sk_u - sk_tile * nkt;
        nk = min(nkt - kt0, sk_u1 - sk_u);
        if (sk_more) __syncthreads();                     // every wave is done with the previous segment's K tiles and slabs
        set_tile(sk_tile, m0, n0);
    }
    if constexpr ((VAR & 2) != 0 && !PERSIST) {
        // the software-pipelined loop's first NS K tiles go out before anything else: they stream in while the row statistics are
        // finalised and the accumulators initialised below
        for (int s = 0; s < NS; ++s)
            if (s < nk) stage(s, kt0 + s);
    }
    if constexpr (LNF && !TRANS && !PERSIST) {
        // Small tiers, row statistics not finalised yet (FpGemmArgs::ln_part): this tile's rows are finalised here — the arithmetic of
        // stats_finalize_kernel, shared through fp_ln_finalize_row — and written where init_acc and the epilogue (and the V^T launch
        // that follows on the stream) read them.  Every workgroup of a row block writes the same bits.
        if (p.ln_part) {
            for (int r = tid; r < BM; r += NW * 64) {
                const int row = m0 + r;
                if (row < p.M) {
                    uint4 rec;
                    float rstd;
                    fp_ln_finalize_row(p.ln_part, (size_t)p.ln_part_ld, p.ln_part_nb, row, p.ln_inv_d, p.ln_eps, rec, rstd);
                    p.ln_mfrag[row] = rec;
                    p.ln_rstd[row] = rstd;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    if (SK && kt0 > 0) {
        // a later K slice of a shared tile: its partial sum starts from zero (the slice that holds K step 0 carries the bias / LayerNorm init)
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TR; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    } else {
        init_acc(m0, n0);
    }

    auto load_frags = [&](const char* sb, int kk, bf16x8_t (&fr)[TR], bf16x8_t (&fc)[TC]) {
        const int slotR = (((kk << 2) | lg) ^ keyR) << 4;
        const int slotC = (((kk << 2) | lg) ^ keyC) << 4;
#pragma unroll
        for (int f = 0; f < TR; ++f) fr[f] = *(const bf16x8_t*)(sb + baseR + f * 4 * ROWB + slotR);
#pragma unroll
        for (int f = 0; f < TC; ++f) fc[f] = *(const bf16x8_t*)(sb + baseC + f * 16 * ROWB + slotC);
    };
    auto mma_block = [&](const bf16x8_t (&fr)[TR], const bf16x8_t (&fc)[TC]) {
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TR; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[j], fc[i], acc[i][j], 0, 0, 0);
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(0);
    };

    if constexpr (PERSIST) {
        // ---- persistent plain loop: the LDS buffers alternate on a step counter that runs across tiles; during the
        // LAST k-step of a tile the first stage of the workgroup's next tile is already being fetched, so address
        // set-up, pipeline fill and workgroup launch hide behind that step's MFMAs and the epilogue, and the
        // epilogue's stores drain while the next tile computes.
        static_assert((VAR & 2) == 0, "persistent form uses the plain loop");
        const int tstride = gridDim.x;
        int g = 0;
        stage(0, 0);
        for (;;) {
            const bool has_next = tile + tstride < ntiles;
            int m0n = 0, n0n = 0;
            for (int kt = 0; kt < nkt; ++kt, ++g) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const bool more = kt + 1 < nkt;
                if (!more && has_next) set_tile(tile + tstride, m0n, n0n);
                const bool do_stage = more || has_next;
                const int nbuf = (g + 1) & 1, nk = more ? kt + 1 : 0;
                const char* sb = smem + (g & 1) * STAGE;
                if constexpr ((VAR & 128) == 0) {
                    if (do_stage) stage(nbuf, nk);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        bf16x8_t fr[TR], fc[TC];
                        load_frags(sb, kk, fr, fc);
                        mma_block(fr, fc);
                    }
                } else {
                    // split DMA issue: the X pieces go out behind the first fragment reads (their LDS latency covers the issue),
                    // the W pieces behind the first MFMA block; MFMA blocks run at raised priority.  Measured in tools/gemm_lab.hip:
                    // +4 % at K = 4096 / 8192, neutral at K = 1024.
                    bf16x8_t fr[TR], fc[TC];
                    load_frags(sb, 0, fr, fc);
                    if (do_stage) stage_x(nbuf, nk);
                    __builtin_amdgcn_s_setprio(1);
                    mma_block(fr, fc);
                    __builtin_amdgcn_s_setprio(0);
                    if (do_stage) stage_w(nbuf, nk);
                    load_frags(sb, 1, fr, fc);
                    __builtin_amdgcn_s_setprio(1);
                    mma_block(fr, fc);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            if (FP_GEMM_DBG_BIT(p, 8)) {   // lab build only (gemm_dbg = 8): the main loop without its epilogue — one store keeps the accumulators live
                if (acc[0][0][0] == 123456.789f) p.C[0] = 0;
            } else {
            if constexpr (RELOC) epi_stage = reloc_stage((g - 1) & 1);
            fp_gemm::epilogue<BM, BN, WM, WN, EPI, VAR, TC, TR>(p, acc, m0, n0, wm, wn, li, lg, epi_stage, smem_raw);
            }
            if (!has_next) return;
            tile += tstride;
            m0 = m0n;
            n0 = n0n;
            zero_acc();
        }
    } else if constexpr ((VAR & 2) == 0) {
        // ---- plain double-buffered loop: one barrier per K tile, fragments read right before use ----
        stage(0, 0);
        for (int kt = 0; kt < nkt; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nkt) stage((kt + 1) & 1, kt + 1);
            const char* sb = smem + (kt & 1) * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t fr[TR], fc[TC];
                load_frags(sb, kk, fr, fc);
                mma_block(fr, fc);
            }
        }
    } else {
        // ---- software-pipelined loop: fragments of k-step j+1 are read while k-step j's MFMAs run; the single
        // barrier of a tile sits at its SECOND k-step, where every wave has finished reading the tile, so the DMA
        // of tile t+2 is issued half a tile earlier and the next tile's first fragments are already in flight.
        // Ring of NS K-tile buffers, NS - 1 stages in flight: 2 for the 128x128 tier, 2 .. 8 (p.ring) for the 64x64 tier.  A launch of
        // the small tiers is a few workgroups per CU, each walking K alone: with two buffers every K step waits for one fresh memory
        // round trip (weights come from HBM: 0.7 us per step measured on a single-crop ViT-L forward), with a deeper ring the round
        // trips overlap.  The launcher takes the deepest ring that still keeps the whole grid resident (LDS per workgroup = NS x 16
        // KiB).  Same K order, same MFMA sequence: the bits do not depend on the depth.
        bf16x8_t frA[TR], fcA[TC], frB[TR], fcB[TC];
        constexpr int IPS = IX + IW;   // LDS-DMA instructions per stage and wave
        // "at most n later stages' instructions outstanding" (vmcnt counts in issue order and takes an immediate); LG: also lgkmcnt(0)
        auto wait_stages = [&](int n, auto lg_c) {
            constexpr bool LG = decltype(lg_c)::value;
#define FP_WAIT_CASE(N)                                                                                          \
    case N:                                                                                                      \
        if constexpr (LG) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((N) * IPS <= 63 ? (N) * IPS : 63) : "memory");   \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N) * IPS <= 63 ? (N) * IPS : 63) : "memory");              \
        break;
            if constexpr (!RING) {
                if constexpr (LG) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                switch (n) {
                    FP_WAIT_CASE(1) FP_WAIT_CASE(2) FP_WAIT_CASE(3) FP_WAIT_CASE(4) FP_WAIT_CASE(5) FP_WAIT_CASE(6) FP_WAIT_CASE(7)
                    default:
                        if constexpr (LG) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
#undef FP_WAIT_CASE
        };
        // (the first NS stages were issued at the top of the kernel) stage 0 has landed when at most the later stages' instructions are
        // outstanding; any vector-memory operation issued since (row statistics, init records) only makes this wait stricter
        if (RING) wait_stages(nk >= NS ? NS - 1 : 0, std::false_type{});
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef FP_LAB
        if (FP_GEMM_DBG_BIT(p, 1024)) dbg_t[1] = wall_clock64();
#endif
        // `settle`: pass a fragment set through an empty asm.  hipcc cannot count LDS reads across the loop back-edge
        // and would otherwise emit lgkmcnt(0) AFTER the prefetch reads are issued (waiting for the prefetch itself);
        // with the settle placed BEFORE the prefetch its wait covers only reads issued a whole MFMA block earlier.
        auto settle = [&](bf16x8_t (&fr)[TR], bf16x8_t (&fc)[TC]) {
#pragma unroll
            for (int f = 0; f < TR; ++f) asm volatile("" : "+v"(fr[f]));
#pragma unroll
            for (int f = 0; f < TC; ++f) asm volatile("" : "+v"(fc[f]));
        };
        constexpr bool SPREAD = (VAR & 4096) != 0;
        // second half of a K step, interleaved: MFMA block B (TC x TR) with [DMA: the IPS pieces of stage `kt_dma` into slot `dslot`] and
        // the TR + TC fragment reads of the next step's first half (from `nsb`)
        auto spread_half = [&](auto dma_c, int dslot, int kt_dma, const char* nsb) {
            constexpr bool DMA = decltype(dma_c)::value;
            constexpr int NM = TC * TR, NP = IX + IW, NR = TR + TC;
            char* db = smem + dslot * STAGE;
            const size_t kb = (size_t)kt_dma * ROWB;
            const int slotR = ((0 | lg) ^ keyR) << 4, slotC = ((0 | lg) ^ keyC) << 4;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int i = m / TR, j = m % TR;
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frB[j], fcB[i], acc[i][j], 0, 0, 0);
                if constexpr (DMA) {
                    if ((m * NP) / NM != ((m + 1) * NP) / NM) {          // piece q goes out behind MFMA m
                        const int q = (m * NP) / NM;
                        if (q < IX) glds16(gX + offX[q] + kb, db + (q * NW + wave) * 1024);
                        else glds16(gW + offW[q - IX] + kb, db + BM * ROWB + ((q - IX) * NW + wave) * 1024);
                    }
                }
                if ((m * NR) / NM != ((m + 1) * NR) / NM) {              // fragment read f behind MFMA m
                    const int f = (m * NR) / NM;
                    if (f < TR) frA[f] = *(const bf16x8_t*)(nsb + baseR + f * 4 * ROWB + slotR);
                    else fcA[f - TR] = *(const bf16x8_t*)(nsb + baseC + (f - TR) * 16 * ROWB + slotC);
                }
            }
            // pin the order: per MFMA at most one DMA piece and one fragment read behind it
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (DMA && (m * NP) / NM != ((m + 1) * NP) / NM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((m * NR) / NM != ((m + 1) * NR) / NM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        load_frags(smem, 0, frA, fcA);
        int cur = 0;   // ring slot of stage kt
        // (lab build, gemm_dbg: 64 = no DMA in the steady state, 128 = no MFMAs, 256 = no fragment reads, 512 = no barrier — wrong results,
        //  what a K step of the small tiers is made of: profiles/r05_ab.md)
        for (int kt = 0; kt < nk; ++kt) {
            const char* sb = smem + cur * STAGE;
            const int nxt = cur + 1 == NS ? 0 : cur + 1;
#ifdef FP_LAB
            // lab build, gemm_dbg = 2048: shader-clock stamps inside K step 8 (wave 0 of every workgroup) — where the wave waits
            const bool stamp = FP_GEMM_DBG_BIT(p, 2048) && kt == 8 && p.sk_ws;
            unsigned long long st[8];
#define FP_STAMP(i) if (stamp) { asm volatile("" ::: "memory"); st[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); }
#else
#define FP_STAMP(i)
#endif
            FP_STAMP(0)
            settle(frA, fcA);
            FP_STAMP(1)
            if (!FP_GEMM_DBG_BIT(p, 256)) load_frags(sb, 1, frB, fcB);
            if (!FP_GEMM_DBG_BIT(p, 128)) mma_block(frA, fcA);
            FP_STAMP(2)
            if (kt + 1 < nk) {
                // stage kt+1 must have landed; stages kt+2 .. kt+NS-1 may stay in flight (near the end of K fewer were issued: drain)
                wait_stages(kt + NS - 1 < nk ? NS - 2 : 0, std::true_type{});
                FP_STAMP(3)
                if (!FP_GEMM_DBG_BIT(p, 512)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                FP_STAMP(4)
                if constexpr (SPREAD) {
                    // SPREAD (VAR bit 4096): the second half of the step as ONE interleaved stream — an LDS-DMA piece holds its issuing wave
                    // for ~50 clocks (profiles/r05_ab.md: 8 pieces in a row = 400 of a step's 1670 clocks with the matrix pipe idle), so
                    // each piece goes out behind a pair of MFMAs of block B, the next step's first fragment reads between them
                    settle(frB, fcB);
                    if (kt + NS < nk) spread_half(std::true_type{}, cur, kt0 + kt + NS, smem + nxt * STAGE);
                    else spread_half(std::false_type{}, cur, 0, smem + nxt * STAGE);
                    cur = nxt;
                    continue;
                }
                if (kt + NS < nk && !FP_GEMM_DBG_BIT(p, 64)) stage(cur, kt0 + kt + NS);   // every wave is done reading stage kt: its slot takes stage kt + NS
                FP_STAMP(5)
                settle(frB, fcB);
                FP_STAMP(6)
                if (!FP_GEMM_DBG_BIT(p, 256)) load_frags(smem + nxt * STAGE, 0, frA, fcA);
            }
            if (!FP_GEMM_DBG_BIT(p, 128)) mma_block(frB, fcB);
            FP_STAMP(7)
#ifdef FP_LAB
            if (stamp && tid == 0) {
                unsigned long long* o = (unsigned long long*)p.sk_ws + 65536 + (size_t)blockIdx.x * 8;
                for (int i = 0; i < 8; ++i) o[i] = st[i];
            }
#endif
#undef FP_STAMP
            cur = nxt;
        }
        last_slot = (nk - 1) % NS;
#ifdef FP_LAB
        if (FP_GEMM_DBG_BIT(p, 1024)) dbg_t[2] = wall_clock64();
#endif
    }

    bool finish = true;                                   // this workgroup runs the tile's epilogue
    if constexpr (SK) {
        if (nk < nkt) {
            // ---- a K slice of a tile shared with neighbouring workgroups.  Every slice writes its fp32 partial tile (agent-coherent
            // stores: the workgroups sit on different XCDs, whose L2s are not coherent with each other) to one of its two scratch slots
            // and counts itself in; the LAST to arrive adds the slices IN K ORDER — slice 0 carries the bias / LayerNorm init — so the
            // sum does not depend on who that is, and runs the epilogue.  Nobody waits for anybody.
            typedef fp_gemm::u32x4_t u32x4;
            constexpr int TILE_F = BM * BN;                // floats per partial tile
            const int ut0 = sk_tile * nkt;
            const int cut = sk_rem * (sk_base + 1);
            auto wg_of = [&](int u) { return u < cut ? u / (sk_base + 1) : sk_rem + (u - cut) / sk_base; };
            const int w_first = wg_of(ut0), nsl = wg_of(ut0 + nkt - 1) - w_first + 1;
            const int voff = (wave * 64 + lane) * 16;      // fragment f of the lane sits at voff + f * NW * 1024
            {
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(p.sk_ws + (size_t)(2 * sk_w + (kt0 > 0 ? 0 : 1)) * TILE_F), 0, TILE_F * 4, 0x00020000);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TR; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rw, voff + (i * TR + j) * NW * 1024, 0, 16);   // sc1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* const flag = (int*)(smem + NS * STAGE);   // 16 spare bytes behind the K-tile ring (launch_cfg)
            if (tid == 0) *flag = __hip_atomic_fetch_add(p.sk_cnt + sk_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            finish = *flag == nsl - 1;
            if (finish) {
                if (tid == 0) __hip_atomic_store(p.sk_cnt + sk_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                for (int sl = 0; sl < nsl; ++sl) {
                    const int w = w_first + sl;
                    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)(p.sk_ws + (size_t)(2 * w + (sl == 0 ? 1 : 0)) * TILE_F), 0, TILE_F * 4, 0x00020000);
#pragma unroll
                    for (int i = 0; i < TC; ++i)
#pragma unroll
                        for (int j = 0; j < TR; ++j) {
                            const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rr, voff + (i * TR + j) * NW * 1024, 0, 16));
                            acc[i][j] = sl == 0 ? v : acc[i][j] + v;
                        }
                }
            }
        }
    }
    if (finish) {
        if constexpr (RELOC) epi_stage = reloc_stage(last_slot);
        fp_gemm::epilogue<BM, BN, WM, WN, EPI, VAR, TC, TR>(p, acc, m0, n0, wm, wn, li, lg, epi_stage, smem_raw);
    }
#ifdef FP_LAB
    if (FP_GEMM_DBG_BIT(p, 1024) && p.sk_ws && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_t[3] = wall_clock64();
        unsigned long long* o = (unsigned long long*)p.sk_ws + (size_t)blockIdx.x * 5;
        o[0] = dbg_t[0]; o[1] = dbg_t[1]; o[2] = dbg_t[2]; o[3] = dbg_t[3]; o[4] = __builtin_amdgcn_s_memtime() - dbg_c0;   // shader clocks
    }
#endif
    if constexpr (SK) {
        sk_u += nk;
        sk_more = sk_u < sk_u1;
    }
    } while (SK && sk_more);
}

// bf16(gelu_erf(x)) for the 8192 input patterns of the table window, evaluated with the device expression of the direct variant
__global__ void gelu_table_kernel(uint16_t* tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fp_gemm::GELU_TAB_ENTRIES) return;
    const float x = __uint_as_float(fp_gemm::gelu_tab_pattern(i) << 16);
    tab[i] = (uint16_t)(__float_as_uint(rbf(fp_gemm::gelu_erf(x))) >> 16);
}

__global__ void gelu_direct_kernel(const bf16_t* x, bf16_t* y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (bf16_t)(__float_as_uint(rbf(fp_gemm::gelu_erf(__uint_as_float((uint32_t)x[i] << 16)))) >> 16);
}

template <int BM, int BN, int WM, int WN, int EPI, int VAR>
int launch_cfg(const FpGemmArgs& a, hipStream_t stream, int sk_grid = 0) {
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr bool SK = (VAR & 2048) != 0;                   // balanced tier: sk_grid workgroups share the (tile, K step) units
    constexpr bool LUT = FpEpiTraits<EPI>::GELU && (VAR & 4) != 0;
    constexpr bool RING = (VAR & 2) && (VAR & 1024);
    constexpr int TABB = LUT ? fp_gemm::GELU_TAB_BYTES : 0;
    constexpr int SLABS = LUT ? 0 : WM * WN * fp_gemm::EPI_STAGE_BYTES;
    constexpr int RMAX = RING ? ((160 * 1024 - TABB - SLABS - 16) / STAGE < 8 ? (160 * 1024 - TABB - SLABS - 16) / STAGE : 8) : 2;   // deepest ring that fits
    constexpr int SMEM_MAX = TABB + RMAX * STAGE + (SK ? 16 : 0);   // dynamic part; the slabs are static (see the kernel)
    static_assert(SMEM_MAX + SLABS <= 160 * 1024, "LDS budget");
    auto kern = gemm_bf16_kernel<BM, BN, WM, WN, EPI, VAR>;
    FP_DYN_LDS_ONCE(kern, SMEM_MAX);
    int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    if constexpr (SK) {
        FP_REQUIRE(sk_grid > 0 && sk_grid <= tiles * (a.K / BK) && a.sk_ws && a.sk_cnt && tiles <= a.sk_cnt_n &&
                       (size_t)sk_grid * 2 * BM * BN * 4 <= a.sk_ws_bytes, "gemm: balanced tier without scratch (grid %d, %d tiles)", sk_grid, tiles);
        tiles = sk_grid;
    }
    FpGemmArgs ar = a;
    ar.ring = 2;
    if constexpr (RING) {
        // the deepest ring (4 / 3 / 2 K-tile buffers) with which the whole grid is still resident at once; no deeper than K.  Deeper
        // rings were measured and bring nothing (profiles/r04_ab.md §3: caps 8 / 6 / 4 / 3 within 1 %, 2 buffers 20-28 % slower at B = 1)
        static int ncu_r = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
        constexpr int IPS = (BM + BN) / 8 / (WM * WN);
        constexpr int FIXED = TABB + (LUT ? 0 : WM * WN * fp_gemm::EPI_STAGE_BYTES);
        const int nkt = a.K / BK;
        int cap = 4;
#ifdef FP_LAB
        cap = fp_opt_get(FP_OPT_GEMM_RING, 4);   // lab: up to 8
#endif
        for (const int r : {8, 6, 4, 3}) {
            if ((r - 1) * IPS > 63 || r > nkt || r > cap || r > RMAX) continue;
            const long resident = (long)ncu_r * ((160 * 1024) / (FIXED + r * STAGE));
            if (tiles <= resident) { ar.ring = r; break; }
        }
    }
    const int SMEM = TABB + ar.ring * STAGE + (SK ? 16 : 0);
    if constexpr ((VAR & 32) != 0) {   // one resident workgroup per CU (128 KiB of LDS each)
        static int ncu = [] { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n & ~7; }();
        tiles = tiles < ncu ? tiles : ncu;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), SMEM, stream, ar);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

template <int EPI>
int launch_epi(const FpGemmArgs& a, hipStream_t stream) {
    // big tile once the grid can fill the chip with it, else the 128x128 tile
    const long tiles_big = (long)cdiv(a.M, 256) * cdiv(a.N, 256);
    // kernel variant bits (template parameter VAR): 1 = s_setprio around MFMA blocks, 2 = software-pipelined fragment reads (8-wave
    // and smaller kernels), 4 = table GELU in the fc1 epilogue (gemm_epilogue.h; 0 = direct erff expression), 8 = 16-wave big tile,
    // 32 = persistent tile walk, 64 = streaming epilogue I/O: non-temporal output stores and residual loads (the last two with the
    // 16-wave big tile), 128 = split DMA issue (X pieces behind the first fragment reads, W pieces behind the first MFMA block) +
    // MFMA priority (persistent 16-wave kernel).  The product always runs FP_GEMM_DEFAULT_VARIANT; only the lab build can change it.
#ifdef FP_LAB
    static int env_var = [] { const char* e = getenv("FP_GEMM_VARIANT"); return e ? atoi(e) : FP_GEMM_DEFAULT_VARIANT; }();
    const int var = fp_opt_get(FP_OPT_GEMM_VARIANT, env_var);
#else
    constexpr int var = FP_GEMM_DEFAULT_VARIANT;
#endif
    // The 256x256 kernels run one resident workgroup per CU, so their time goes in whole rounds of `ncu` tiles.  When the last
    // round would be mostly empty (e.g. 300 tiles on 256 CUs: the ~20-crop batches of the video path) the 128x128 kernel —
    // ~15-20 % less efficient per flop but 8x finer grained — is faster: measured 0.053 vs 0.068 ms (proj), 0.157 vs 0.192
    // (fc2), 0.046 vs 0.063 (V) at M = 19 152, while the big tile wins whenever >= ~75 % of its rounds are filled.
    static int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 8 ? (n & ~7) : 256; }();
    const long rounds_big = (tiles_big + ncu - 1) / ncu;
    if constexpr (FpEpiTraits<EPI>::LN) {
        // row statistics handed over as partial sums: the small tiers finalise them in the kernel prologue; anything that may touch the
        // big tier (or is a transposed store) gets them from the finalisation kernel first
        if (a.ln_part && (FpEpiTraits<EPI>::TRANS || !fp_gemm_fuses_ln_part(a.M, a.N))) {
            FP_REQUIRE(a.ln_part_nb > 0 && a.ln_part_ld >= a.M, "gemm: ln_part needs ln_part_nb and ln_part_ld");
            const int rc = fp_stats_finalize(a.ln_part, a.ln_mfrag, a.ln_rstd, a.M, a.ln_part_nb * 64,
                                             a.ln_eps, stream, a.ln_part_ld);
            if (rc != FP_OK) return rc;
            FpGemmArgs b = a;
            b.ln_part = nullptr;
            return launch_epi<EPI>(b, stream);
        }
    }
    const bool filled = tiles_big * 4 >= rounds_big * ncu * 3;          // >= 75 % of the big-tile rounds are real work
    bool big = tiles_big >= BIG_MIN_TILES && filled && !(var & 256);    // bit 256 (A/B only): force the 128x128 kernel
    // (fp_gemm_fuses_ln_part shares BIG_MIN_TILES: a launch that still carries partial sums was promised the small tiers, which finalise
    //  them in their prologue — the 256x256 kernels would read stale records)
    FP_REQUIRE(!(big && a.ln_part), "gemm: partial row statistics on the big tier (M=%d N=%d)", a.M, a.N);
    // The hand-scheduled 128x128 kernel (gemm_asm.hip, geometry "S": a persistent walk over two resident workgroups per CU, so no second
    // row split below) as the small tier of the row-major epilogues: LAB BUILD ONLY — +8 ... 30 % on isolated launches with a resident X,
    // -4 ... +17 % on ViT forwards (profiles/r05_ab.md §3), not shipped.  gemm_variant bit 1048576 = on; with it bit 2097152 = only where
    // the launch gives every CU a 128x128 tile, bit 4194304 = only for K >= 2048.
    bool asm_small = false;
#ifdef FP_LAB
    asm_small = !FpEpiTraits<EPI>::TRANS && (var & 1048576) && fp_gemm_asm_small_supported(a, EPI);
    if ((var & 2097152) && (long)cdiv(a.M, 128) * cdiv(a.N, 128) < ncu) asm_small = false;
    if ((var & 4194304) && a.K < 2048) asm_small = false;
#endif
    // ROW SPLIT (round 3): whole rounds of the resident grid on 256x256 tiles, the remaining rows on the finer tiers — for the launch
    // sizes in between (the video path's ~20-crop batches: 600 qk tiles = 2.34 rounds, 300 proj / fc2 tiles = 1.17) where the big
    // tile wastes most of a round and the small tile, ~0.56 of a big round per round of 2 tiles per CU at half the work, pays its
    // lower efficiency on every row.  Rows are independent and every tier produces the same bits (tests/test_gpu_fullsize.py), so
    // the split is invisible in the results.  Cost model in units of one big round; the split must win by 4 %.  Not for the
    // transposed-V and patch-embed epilogues (their row -> output mapping is per crop).  Bit 4096 (A/B only): never split.
    if constexpr (EPI != FP_EPI_VT && EPI != FP_EPI_LN_VT && EPI != FP_EPI_PATCH) {
        const long tiles_n = cdiv(a.N, 256);
        const long full = tiles_big / ncu;                               // whole rounds
        const long rb = full * ncu / tiles_n;                            // 256-row blocks they cover
        if (!(var & (256 | 4096)) && !a.no_split && full >= 1 && rb * 256 < a.M && tiles_big >= BIG_MIN_TILES) {
            const double r_small = asm_small ? 0.45 : 0.56;   // a round of two 128x128 tiles per CU in units of a big round (half its work)
            auto small_cost = [&](long rows) { return (double)cdiv(cdiv(rows, 128) * cdiv((long)a.N, 128), 2L * ncu) * r_small; };
            const double t_big = (double)rounds_big, t_small = small_cost(a.M);
            const double t_now = big ? t_big : t_small;
            const double t_split = (double)cdiv(rb * tiles_n, (long)ncu) + small_cost(a.M - rb * 256);
            if (t_split < 0.96 * t_now) {
                const size_t m1 = (size_t)rb * 256;
                FpGemmArgs a1 = a, a2 = a;
                a1.M = (int)m1; a1.no_split = 1;
                a2.M = a.M - (int)m1; a2.no_split = 1;
                a2.X = a.X + m1 * a.ldx;
                a2.C = a.C + m1 * a.ldc;
                if (a.resid) a2.resid = a.resid + m1 * a.ldr;
                if (a.ln_mfrag) a2.ln_mfrag = a.ln_mfrag + m1;
                if (a.ln_rstd) a2.ln_rstd = a.ln_rstd + m1;
                if (a.ln_part) a2.ln_part = a.ln_part + m1;
                if (a.stat_part) a2.stat_part = a.stat_part + m1;
                const int rc = launch_epi<EPI>(a1, stream);              // whole rounds: takes the 256x256 tier below
                if (rc != FP_OK) return rc;
                return launch_epi<EPI>(a2, stream);                      // the remainder picks its own tier (128x128 or 64x64)
            }
        }
    }
    // ROW SPLIT below the big tier (round 4): the 128x128 kernel keeps two workgroups per CU resident, so its time also goes in whole
    // rounds (2 x ncu tiles) — 6 crops @518^2 give the N = 1024 layers 520 tiles = one round + 8 tiles, i.e. TWO rounds (the B = 5 -> 6
    // step of a ViT-L forward was 6.1 -> 8.7 ms).  Whole rounds run on 128x128 tiles, the remaining rows pick their own tier (a
    // remainder below one tile per CU takes the one-wave 64x64 tiles, whose round is ~half the cost).  Same bits from every tier.
    if constexpr (EPI != FP_EPI_VT && EPI != FP_EPI_LN_VT && EPI != FP_EPI_PATCH) {
        const long tn = cdiv(a.N, 128), slots = 2L * ncu, tiles_mid_all = (long)cdiv(a.M, 128) * tn;
        const long fullm = tiles_mid_all / slots;
        const long rbm = fullm * slots / tn;                              // 128-row blocks the whole rounds cover
        if (!big && !asm_small && !(var & (256 | 4096)) && !a.no_split && fullm >= 1 && rbm * 128 < a.M) {
            const long rem_rows = a.M - rbm * 128, rem_mid = cdiv(rem_rows, 128L) * tn;
            const double t_now = (double)cdiv(tiles_mid_all, slots);
            const double t_rem = rem_mid < ncu ? 0.5 * (double)cdiv(cdiv(rem_rows, 64L) * cdiv((long)a.N, 64L), 4L * ncu) : (double)cdiv(rem_mid, slots);
            const double t_split = (double)cdiv(rbm * tn, slots) + t_rem;
            if (t_split < 0.9 * t_now) {
                const size_t m1 = (size_t)rbm * 128;
                FpGemmArgs a1 = a, a2 = a;
                a1.M = (int)m1; a1.no_split = 1;
                a2.M = a.M - (int)m1; a2.no_split = 1;
                a2.X = a.X + m1 * a.ldx;
                a2.C = a.C + m1 * a.ldc;
                if (a.resid) a2.resid = a.resid + m1 * a.ldr;
                if (a.ln_mfrag) a2.ln_mfrag = a.ln_mfrag + m1;
                if (a.ln_rstd) a2.ln_rstd = a.ln_rstd + m1;
                if (a.ln_part) a2.ln_part = a.ln_part + m1;
                if (a.stat_part) a2.stat_part = a.stat_part + m1;
                const int rc = launch_epi<EPI>(a1, stream);
                if (rc != FP_OK) return rc;
                return launch_epi<EPI>(a2, stream);
            }
        }
    }
#ifdef FP_LAB
    // BALANCED TIER (round 5, LAB BUILD ONLY — measured and not shipped, profiles/r05_ab.md §1): G workgroups share the launch's (tile, K
    // step) units evenly (stream-K); a tile whose K range is spread over several workgroups is completed by the last of them to arrive,
    // which adds the fp32 partial tiles in K order.  Forms: gemm_sk = 2 / 3 / 4 -> 128x128 / 64x64 / 128x128-on-a-ring units,
    // gemm_sk_grid = G.
    if constexpr (EPI != FP_EPI_PATCH) {
        int sk_tier = 0, sk_grid = 0;
        const int f = fp_opt_get(FP_OPT_GEMM_SK, 0), g = fp_opt_get(FP_OPT_GEMM_SK_GRID, 0);
        if (f >= 2 && f <= 4 && !big && a.sk_ws && a.sk_cnt && a.sk_mode != 1) {
            sk_tier = f - 1;
            const long units = (long)cdiv(a.M, f == 3 ? 64 : 128) * cdiv(a.N, f == 3 ? 64 : 128) * (a.K / BK);
            sk_grid = (int)std::min<long>(units, g > 0 ? g : (f == 3 ? 4L : f == 4 ? 1L : 2L) * ncu);
            sk_grid = (int)std::min<long>(sk_grid, (long)(a.sk_ws_bytes / (f == 3 ? 2 * 64 * 64 * 4 : 2 * 128 * 128 * 4)));
        }
        if (sk_tier == 1) return launch_cfg<128, 128, 2, 2, EPI, FP_GEMM_VAR_SMALL | 2048>(a, stream, sk_grid);
        if (sk_tier == 2) {
            if constexpr (FpEpiTraits<EPI>::TRANS) return launch_cfg<64, 64, 1, 1, EPI, FP_GEMM_VAR_TINY | 2048>(a, stream, sk_grid);
            else return launch_cfg<64, 64, 2, 1, EPI, FP_GEMM_VAR_TINY | 2048>(a, stream, sk_grid);
        }
        if (sk_tier == 3) return launch_cfg<128, 128, 2, 2, EPI, FP_GEMM_VAR_SMALL | 1024 | 2048>(a, stream, sk_grid);
    }
#endif
#ifdef FP_LAB
    if constexpr (!FpEpiTraits<EPI>::TRANS) {
        if (!big && asm_small) return fp_gemm_asm_small(a, EPI, stream);
    }
#endif
    // A launch that cannot even give every CU one 128x128 tile (a single 518^2 crop: 88 tiles for N = 1024) runs one-wave
    // 64x64 tiles instead — 4x the workgroups, all CUs busy (bit 2048, A/B only: keep the 128x128 kernel).
    const long tiles_mid = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const bool tiny = !big && ((tiles_mid < ncu && !(var & 2048)) || (var & 32768));   // bit 32768 (lab A/B only): one-wave tiles whenever not big
    // Big tier: the hand-scheduled one-wave-per-SIMD kernel (gemm_asm.hip) for the row-major
    // epilogues where it is the faster one (fp_gemm_asm_preferred: long K), the 16-wave HIP kernel otherwise.
    if constexpr (!FpEpiTraits<EPI>::TRANS) {
        // lab A/B only: bit 8192 = never, bit 16384 = the 4-wave kernel wherever it supports the shape, bit 65536 = the 8-wave kernel
        if (big && (var & 65536) && fp_gemm_asm_supported(a, EPI)) return fp_gemm_asm(a, EPI, 8, stream);
        if (big && !(var & 8192) && ((var & 16384) ? fp_gemm_asm_supported(a, EPI) : fp_gemm_asm_preferred(a, EPI))) return fp_gemm_asm(a, EPI, 4, stream);
    }
#ifdef FP_LAB
    if constexpr (EPI < FP_EPI_LN_BIAS) {   // lab build: the alternative kernels of the plain epilogues (A/B runs)
        if (var != FP_GEMM_DEFAULT_VARIANT) {
#define FP_GEMM_CASE(V)                                                          \
    case V: return big ? launch_cfg<256, 256, 2, 4, EPI, V>(a, stream)          \
               : tiny ? launch_cfg<64, 64, 1, 1, EPI, V>(a, stream)             \
                      : launch_cfg<128, 128, 2, 2, EPI, V>(a, stream);
            if (big && (var & 8)) {   // 16-wave workgroup (4 waves/SIMD), 64x64 per wave
                switch (var & (32 | 64)) {   // 32: persistent tile walk, 64: streaming (non-temporal) output stores
                    case 32: return launch_cfg<256, 256, 4, 4, EPI, 4 | 32>(a, stream);
                    case 64: return launch_cfg<256, 256, 4, 4, EPI, 4 | 64>(a, stream);
                    case 96:
                        if ((var & 128) && (var & 512)) return launch_cfg<256, 256, 4, 4, EPI, 4 | 96 | 128 | 512>(a, stream);
                        if ((var & 128) && (var & 1024)) return launch_cfg<256, 256, 4, 4, EPI, 4 | 96 | 128 | 1024>(a, stream);   // whole-N sweep of fc1 (round 6)
                        return (var & 128) ? launch_cfg<256, 256, 4, 4, EPI, FP_GEMM_VAR_BIG>(a, stream)
                                           : launch_cfg<256, 256, 4, 4, EPI, 4 | 96>(a, stream);
                    default: return launch_cfg<256, 256, 4, 4, EPI, 4>(a, stream);
                }
            }
            switch (var & 7) {   // measured on MI355X (profiles/): 6 is the fastest; 0 is kept as the plain baseline for A/B runs
                FP_GEMM_CASE(0)
                default:
                FP_GEMM_CASE(6)
            }
#undef FP_GEMM_CASE
        }
    }
#endif
    // one kernel per (epilogue, tile tier)
    // tiny tier: 64x64 tiles.  Row-major epilogues split the tile over TWO waves (32 x 64 each): a launch that cannot fill the chip is
    // bound by one wave's walk along K, and per K step a lone wave issues 16 LDS-DMA pieces for 32 MFMAs — two waves halve both
    // (ViT-L B = 1 @518^2: see profiles/r04_ab.md §4).  The transposed V store needs 64-token wave tiles and keeps one wave.
    // STREAMING POLICY of the big tier (round 5).  Non-temporal full-line stores (and residual loads) keep a large output from dirtying
    // every XCD's L2 (round 1: qk 0.373 -> 0.312 ms at 214 crops) — but an output that fits in the 256 MiB Infinity Cache beside its
    // consumer's other operands is what the next kernel reads, and bypassing the caches sends that kernel to HBM.  Same-process A/B of
    // ViT-L forwards with the threshold at 0 / 128 / 256 / 512 MiB / never, three boxes (profiles/r05_stream_policy_ab.log): ordinary
    // stores below 128 MiB are 1.8-3.1 % faster at 5, 8 and 32 crops @420^2, -3 ... +1 % at 12 / 21 crops (box-dependent), neutral from
    // 96 crops on; a 256 MiB threshold loses up to 1.5 % at 21-32 crops on two of the boxes; never streaming loses 1-3 % from 48 crops on.
    long stream_bytes = FP_GEMM_STREAM_BYTES;
#ifdef FP_LAB
    stream_bytes = (long)fp_opt_get(FP_OPT_GEMM_STREAM_MB, (int)(FP_GEMM_STREAM_BYTES >> 20)) << 20;   // lab A/B: the threshold in MiB (0 = always stream)
    if (var & 8388608) stream_bytes = 1L << 60;                                                          // bit 8388608: never stream
#endif
    const bool stream_out = (long)a.M * a.N * 2 > stream_bytes;
    if (big && !stream_out) return launch_cfg<256, 256, 4, 4, EPI, FP_GEMM_VAR_BIG & ~64>(a, stream);
    if constexpr (FpEpiTraits<EPI>::TRANS) {
#ifdef FP_LAB
        if (!big && !tiny && (var & 524288)) return (var & 262144) ? launch_cfg<128, 128, 2, 4, EPI, FP_GEMM_VAR_SMALL | 4096>(a, stream) : launch_cfg<128, 128, 2, 2, EPI, FP_GEMM_VAR_SMALL | 4096>(a, stream);   // lab A/B: spread DMA issue
        if (!big && tiny && (var & 524288)) return launch_cfg<64, 64, 1, 1, EPI, FP_GEMM_VAR_TINY | 4096>(a, stream);
        if (!big && !tiny && (var & 262144)) return launch_cfg<128, 128, 2, 4, EPI, FP_GEMM_VAR_SMALL>(a, stream);   // lab A/B: 128x128 tier on 8 waves
#endif
        return big ? launch_cfg<256, 256, 4, 4, EPI, FP_GEMM_VAR_BIG>(a, stream)
             : tiny ? launch_cfg<64, 64, 1, 1, EPI, FP_GEMM_VAR_TINY>(a, stream)
                    : launch_cfg<128, 128, 2, 2, EPI, FP_GEMM_VAR_SMALL>(a, stream);
    } else {
#ifdef FP_LAB
        if (!big && tiny && (var & 131072)) return launch_cfg<64, 64, 2, 1, EPI, FP_GEMM_VAR_SMALL>(a, stream);   // lab A/B: two K-tile buffers
        if (!big && !tiny && (var & 524288)) return (var & 262144) ? launch_cfg<128, 128, 4, 2, EPI, FP_GEMM_VAR_SMALL | 4096>(a, stream) : launch_cfg<128, 128, 2, 2, EPI, FP_GEMM_VAR_SMALL | 4096>(a, stream);   // lab A/B: spread DMA issue
        if (!big && tiny && (var & 524288)) return launch_cfg<64, 64, 2, 1, EPI, FP_GEMM_VAR_TINY | 4096>(a, stream);
        if (!big && !tiny && (var & 262144)) return launch_cfg<128, 128, 4, 2, EPI, FP_GEMM_VAR_SMALL>(a, stream);   // lab A/B: 128x128 tier on 8 waves
#endif
        return big ? launch_cfg<256, 256, 4, 4, EPI, FP_GEMM_VAR_BIG>(a, stream)
             : tiny ? launch_cfg<64, 64, 2, 1, EPI, FP_GEMM_VAR_TINY>(a, stream)
                    : launch_cfg<128, 128, 2, 2, EPI, FP_GEMM_VAR_SMALL>(a, stream);
    }
}

}  // namespace

bool fp_gemm_fuses_ln_part(int M, int N) { return (long)cdiv(M, 256) * cdiv(N, 256) < BIG_MIN_TILES; }

// Per-device GELU table (16 KiB), built on first use; fp_ctx_create calls this so that no launch path ever allocates.
int fp_gemm_gelu_table(const uint16_t** out) {
    static std::mutex mu;
    static uint16_t* tabs[64] = {};
    int dev = 0;
    FP_HIP(hipGetDevice(&dev));
    FP_REQUIRE(dev >= 0 && dev < 64, "gemm: device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(mu);
    if (!tabs[dev]) {
        uint16_t* t = nullptr;
        FP_HIP(hipMalloc((void**)&t, fp_gemm::GELU_TAB_BYTES));
        hipLaunchKernelGGL(gelu_table_kernel, dim3(fp_gemm::GELU_TAB_ENTRIES / 256), dim3(256), 0, 0, t);
        FP_LAUNCH_CHECK();
        FP_HIP(hipDeviceSynchronize());
        tabs[dev] = t;
    }
    if (out) *out = tabs[dev];
    return FP_OK;
}

int fp_gemm_bf16(const FpGemmArgs& a_in, int epi, hipStream_t stream) {
    FpGemmArgs a = a_in;
#ifdef FP_LAB
    static const int dbg_env = [] { const char* e = getenv("FP_GEMM_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = fp_opt_get(FP_OPT_GEMM_DBG, dbg_env);
#endif
    if (a.stat_ld == 0) a.stat_ld = a.M;
    if (epi == FP_EPI_BIAS_GELU || epi == FP_EPI_LN_GELU) {
        const int rc = fp_gemm_gelu_table(&a.gelu_tab);
        if (rc != FP_OK) return rc;
    }
    FP_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    FP_REQUIRE(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    FP_REQUIRE(a.N % 16 == 0, "gemm: N=%d must be a multiple of 16", a.N);
    FP_REQUIRE(((size_t)a.M * a.ldx * 2) < 0xffffffffull && ((size_t)a.N * a.ldw * 2) < 0xffffffffull,
               "gemm: operand larger than 4 GiB (M=%d ldx=%d)", a.M, a.ldx);
    FP_REQUIRE((a.ldx % 8) == 0 && (a.ldw % 8) == 0 && (a.ldc % 8) == 0, "gemm: leading dims must be multiples of 8");
    switch (epi) {
        case FP_EPI_BIAS: return launch_epi<FP_EPI_BIAS>(a, stream);
        case FP_EPI_BIAS_GELU: return launch_epi<FP_EPI_BIAS_GELU>(a, stream);
        case FP_EPI_BIAS_LS_RES:
            FP_REQUIRE(a.gamma && a.resid, "gemm: LS_RES epilogue needs gamma and resid");
            return launch_epi<FP_EPI_BIAS_LS_RES>(a, stream);
        case FP_EPI_PATCH:
            FP_REQUIRE(a.pos && a.P > 0 && a.npad > 0, "gemm: PATCH epilogue needs pos/P/npad");
            return launch_epi<FP_EPI_PATCH>(a, stream);
        case FP_EPI_VT:
            FP_REQUIRE(a.npad % 16 == 0 && a.M % 16 == 0 && a.heads > 0 && a.N == a.heads * 64,
                       "gemm: VT epilogue needs npad%%16==0, M%%16==0, N==heads*64");
            return launch_epi<FP_EPI_VT>(a, stream);
        case FP_EPI_LN_BIAS:
        case FP_EPI_LN_GELU:
            FP_REQUIRE(a.ln_mfrag && a.ln_rstd && a.ln_cfrag && a.N % 64 == 0 && a.M >= 16,
                       "gemm: LN-folded epilogue needs ln_mfrag, ln_rstd, ln_cfrag, N %% 64 == 0 and M >= 16 (M=%d N=%d)", a.M, a.N);
            return epi == FP_EPI_LN_BIAS ? launch_epi<FP_EPI_LN_BIAS>(a, stream) : launch_epi<FP_EPI_LN_GELU>(a, stream);
        case FP_EPI_LN_VT:
            FP_REQUIRE(a.ln_mfrag && a.ln_rstd && a.ln_cfrag, "gemm: LN-folded epilogue needs ln_mfrag, ln_rstd and ln_cfrag");
            FP_REQUIRE(a.npad % 16 == 0 && a.M % 16 == 0 && a.heads > 0 && a.N == a.heads * 64,
                       "gemm: VT epilogue needs npad%%16==0, M%%16==0, N==heads*64");
            return launch_epi<FP_EPI_LN_VT>(a, stream);
        case FP_EPI_LS_RES_STATS:
            FP_REQUIRE(a.gamma && a.resid && a.stat_part && a.N % 64 == 0, "gemm: LS_RES_STATS epilogue needs gamma, resid, stat_part, N %% 64 == 0");
            return launch_epi<FP_EPI_LS_RES_STATS>(a, stream);
        default: fp_set_error("gemm: unknown epilogue %d", epi); return FP_ERR_INVALID;
    }
}

int fp_gemm_gelu_direct(const bf16_t* x, bf16_t* y, size_t n, hipStream_t stream) {
    if (n == 0) return FP_OK;
    hipLaunchKernelGGL(gelu_direct_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

const char* fp_gemm_kernel_name(int epi) {
    (void)epi;
    return "gemm_bf16_kernel";
}
