// Flash-style multi-head attention forward for the ViT blocks (K4 of SURVEY.md §2.3), gfx950 only.
//
//   O[b, t, h*64:(h+1)*64] = softmax_keys( Q[b,t,h] · K[b,:,h]^T / 8 ) · V[b,:,h]
//
// Inputs come straight from the qkv GEMMs:
//   QK : [B*npad, 2*D] bf16   (q | k, head h at columns h*64)
//   Vt : [B, H, 64, npad] bf16 (V stored transposed per head by the FP_EPI_VT epilogue)
// so both MFMA stages read K-contiguous 16-byte operands and no transpose is ever needed:
//   S^T[key,q] = mfma(A = K rows,  B = Q rows)   -> a lane owns 16 keys of ONE query column:
//                                                   softmax statistics stay (almost) lane-local
//   O^T[d,q]   = mfma(A = V^T rows, B = P^T)      -> P^T fragments are built in registers from the
//                                                   S^T accumulators; the contraction index is
//                                                   re-labelled (key order inside a 32-key step) so
//                                                   no cross-lane exchange is required
// One workgroup = 4 waves x 32 queries = 128 queries of one (crop, head); K / V^T tiles of 64 keys are
// DMA'd (global_load_lds) into a double-buffered, XOR-swizzled LDS ring shared by the 4 waves.
// Sequence length is runtime (905 tokens @420^2, 1374 @518^2, 261 for ViT-S@224^2), keys >= n_tok are
// masked; rows are padded per crop to npad (multiple of 16).
//
// Reference op replaced: the attention inside hub DINOv2 `blk(x)` (src/pipeline/retrieval/dino.py:18-19),
// xformers memory_efficient_attention in the reference environment (environment_cuda.yaml:33).
#include "internal.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

constexpr int HD = 64;      // head dim
constexpr int KVB = 64;     // keys per tile
constexpr int QW = 32;      // queries per wave
constexpr int NWAVE = 4;
constexpr int QB = QW * NWAVE;
constexpr int ROWB = 128;   // bytes per LDS row (64 bf16)
constexpr int TILE = KVB * ROWB;          // 8 KiB (K tile) == 64 d-rows * 128 B (V^T tile)
constexpr int STAGE = 2 * TILE;
constexpr float LAZY_THR = 40.0f;   // logits; 40 * log2(e)/8 = 7.2 -> probabilities stay below 2^8 between rescales

struct AttnArgs {
    const bf16_t* QK; int ldqk;  // elements
    const bf16_t* Vt;
    bf16_t* O; int ldo;
    int B, H, n_tok, npad, D;
    float scale_log2e;           // log2(e) / sqrt(hd)
    int q_base;                  // first query row (per crop) this launch covers: the 256-query kernel takes [0, q_base), this one the tail
};

// combine a value with the lane whose id differs in bit 4 (16-lane rows) / bit 5 (32-lane halves): gfx950
// v_permlane16_swap / v_permlane32_swap are plain VALU instructions (a ds_bpermute round trip costs ~100 cycles of
// latency on the softmax critical path).  swap(a=v, b=v) leaves {own, partner} in the two results on every lane.
// raw v_max_f32 (and v_max3_f32 in softmax_max): fmaxf() makes hipcc canonicalise each operand first (v_max x, x, x — 12 extra VALU instructions
// per K/V tile in the softmax, and the vector ALU is this kernel's co-bound); logits are never signalling NaNs
__device__ __forceinline__ float vmax2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float xlane_max16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xlane_max32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ int key_plain(int row) { return (row >> 1) & 7; }
// K tile rows are read in the order 32 (fk>>1) + 8 a + 4 (fk&1) + b (a = li>>2, b = li&3): key follows (a, b>>1)
__device__ __forceinline__ int key_krow(int row) { return ((((row >> 3) & 3) << 1) | ((row >> 1) & 1)) & 7; }
__device__ __forceinline__ int key_perm4(int row) {  // rows laid out 16a + 4f + b (a,b in 0..3)
    const int rl = row & 63;
    return (((rl >> 4) << 1) | ((rl & 3) >> 1)) & 7;
}

template <int NSLOT, int VAR = 0>  // LDS ring depth: NSLOT-1 K/V tiles in flight (16 KiB per slot); VAR bit 1: MFMA segments at raised priority
__global__ __launch_bounds__(NWAVE * 64, (VAR & 16) ? 4 : 3) void attn_fwd_kernel(AttnArgs p) {   // 3 waves/SIMD: <= 168 VGPRs
    // VAR bit 16 (the shipped flavour since round 4): FOUR waves per SIMD — 128 registers, no spills — by reading the V^T fragments
    // per 32-key half right before their MFMAs instead of the whole tile up front, and finishing both query halves' softmax before
    // the PV products.  Bit-identical to the 3-wave flavour; +0.8-1.4 % at 1374 tokens, +2.3-2.7 % at 905 (profiles/r04_ab.md §5).
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // XCD-aware block order (1-D grid): the q-blocks of one (crop, head) are consecutive LOGICAL ids, hence run on one
    // XCD and share its L2 copy of that head's K / V^T (rocprofv3: 4.4x over-fetch when they were spread over 8 XCDs)
    const int nqb = (p.npad - p.q_base + QB - 1) / QB;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, pos = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    const int qb = bid % nqb, bh = bid / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int q0 = p.q_base + qb * QB + wave * QW;
    // A wave none of whose 32 queries exists (the last query block of 1374 / 905 tokens holds 94 / 9 queries of 128; q0 is a multiple
    // of 32 and npad of 16, so "q0 >= npad" is exactly "no row to store") only takes part in each tile's DMA and barrier.
    // 905 tokens: 718 -> 744 TFLOP/s; 1374: +0.5 % (profiles/r03_ab.md §3).
    const bool idle = q0 >= p.npad;

    const size_t rowbase = (size_t)b * p.npad;
    const char* gQK = (const char*)p.QK;
    const char* gVt = (const char*)(p.Vt + ((size_t)b * p.H + h) * HD * p.npad);

    // ---- Q fragments (B operand): lane (li -> query, lg -> 8-wide d slot) ---------------------
    bf16x8_t qf[2][2];
#pragma unroll
    for (int fq = 0; fq < 2; ++fq) {
        const int q = min(q0 + 16 * fq + li, p.npad - 1);
        const char* src = gQK + ((rowbase + q) * p.ldqk + h * HD) * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[fq][kk] = *(const bf16x8_t*)(src + (kk * 4 + lg) * 16);
    }

    // ---- DMA source offsets: each wave moves 2 x 1 KiB of the K tile and 2 x 1 KiB of V^T ------
    // K tile row r = key (plain map), V^T tile row r = d (perm map)
    uint32_t offK[2];
    int rowK[2];
    uint32_t offV[2];
    int slotV[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (it * NWAVE + wave) * 8 + (lane >> 3);
        rowK[it] = row;
        offK[it] = (uint32_t)((p.D + h * HD) * 2 + (((lane & 7) ^ key_krow(row)) << 4));
        slotV[it] = (lane & 7) ^ key_perm4(row);
        offV[it] = (uint32_t)row * (uint32_t)p.npad * 2u;
    }
    // Unclamped tiles go through buffer descriptors: per-lane byte offset fixed for the whole kernel (VGPR), the tile's offset in
    // the scalar operand — the loop spends no vector ALU instruction on DMA addresses (this kernel is co-bound by VALU issue).
    // Only a tile that can reach past the crop's npad rows takes the clamped path (keys >= n_tok are masked anyway, the clamp
    // just keeps the reads inside the buffers).
    const char* gKb = gQK + rowbase * (size_t)p.ldqk * 2;          // this crop's rows (uniform)
    uint32_t lofK[2], lofV[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        lofK[it] = (uint32_t)rowK[it] * (uint32_t)p.ldqk * 2u + offK[it];
        lofV[it] = offV[it] + (uint32_t)(slotV[it] * 16);
    }
    const uint32_t kstep = (uint32_t)p.ldqk * 2u;                   // bytes per key row
    auto stage = [&](int buf, int kv0) {
        char* sb = smem + buf * STAGE;
        if (kv0 + KVB <= p.npad) {
#pragma unroll
            for (int it = 0; it < 2; ++it) glds16_buf(gKb, lofK[it], (uint32_t)kv0 * kstep, sb + (it * NWAVE + wave) * 1024);
#pragma unroll
            for (int it = 0; it < 2; ++it) glds16_buf(gVt, lofV[it], (uint32_t)kv0 * 2u, sb + TILE + (it * NWAVE + wave) * 1024);
        } else {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int key = min(kv0 + rowK[it], p.npad - 1);
                glds16(gQK + (rowbase + key) * (size_t)p.ldqk * 2 + offK[it], sb + (it * NWAVE + wave) * 1024);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int k8 = min(kv0 + slotV[it] * 8, p.npad - 8);
                glds16(gVt + offV[it] + (size_t)k8 * 2, sb + TILE + (it * NWAVE + wave) * 1024);
            }
        }
    };

    // ---- fragment read offsets -----------------------------------------------------------------
    // K fragment fk, A-row i = li  <->  tile key  32 (fk>>1) + 8 (li>>2) + 4 (fk&1) + (li&3).  With this row permutation a
    // lane's S^T registers of fragments (2s, 2s+1) are the 8 CONSECUTIVE keys 32 s + 8 lg + j, i.e. exactly the MFMA
    // B-operand k-order of the P^T fragment, so the matching V^T fragment is one contiguous ds_read_b128.
    const int keyK = (((li >> 2) << 1) | ((li & 3) >> 1)) & 7;     // key_krow(row) for every fk
    const int baseK = ((li >> 2) * 8 + (li & 3)) * ROWB;           // + (32 (fk>>1) + 4 (fk&1)) * ROWB
    const int keyV = (((li >> 2) << 1) | ((li & 3) >> 1)) & 7;     // key_perm4(row) for every fd
    const int baseV = TILE + ((li >> 2) * 16 + (li & 3)) * ROWB;   // + fd*4*ROWB

    f32x4_t o[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) o[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float mrow[2] = {-1e30f, -1e30f};
    float mthr[2] = {-1e30f, -1e30f};   // mrow + LAZY_THR and -mrow * scale, kept beside mrow (they change only on a rescale)
    float nmb[2] = {0.f, 0.f};
    // PRE (VAR bit 64): Q arrives multiplied by log2(e) / sqrt(hd) (folded into the q rows of the qkv weights), and the first S^T MFMA
    // starts from C = -reference, so the accumulators ARE the exponents: no multiply-add per element in the softmax.
    constexpr bool PRE = (VAR & 64) != 0;
    // VAR bit 256 (lab A/B, round 5): the row sums on the VECTOR ALU — one v_dot2_f32_bf16 per packed P word (16 per tile and wave) into a
    // per-lane partial, reduced over the query's four lanes once at the end — instead of the fifth "V^T" fragment of ones (4 of the
    // tile's 36 MFMAs).  The review's question: do the dot products find free issue slots behind the PV MFMAs?
    constexpr bool VSUM = (VAR & 256) != 0;
    float vsum[2] = {0.f, 0.f};
    f32x4_t nref[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};   // -reference (log2 units) on every register
    f32x4_t lacc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};   // [0]: running sum of P per query
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // 3-slot LDS ring, two K/V tiles in flight: each wave issues 4 DMA instructions per tile, so `vmcnt(4)` means
    // "tile t has landed, tile t+1 may still be in flight".  Raw s_barrier (a __syncthreads() would drain vmcnt to 0).
    // The barrier of iteration t also proves every wave finished reading tile t-1, whose slot tile t+2 now reuses.
    const int ntile = (p.n_tok + KVB - 1) / KVB;
    // VAR bit 4 (chosen by the host when the LAST tile's valid keys all sit in its first 32: 1374 tokens = 21 tiles + 30 keys, 905 =
    // 14 + 9): that tile goes FIRST — softmax accumulation does not care about the order — through a half-length body (NKS = 1:
    // half the S^T MFMAs, exponentials and PV MFMAs), then the full tiles follow in the uniform loop.  Being a compile-time prologue
    // it adds no second body at a join inside the loop (which cost 36 spills when tried).
    constexpr bool SHORT = (VAR & 4) != 0;
    auto tile_of = [&](int sq) { return SHORT ? (sq == 0 ? ntile - 1 : sq - 1) : sq; };   // sequence position -> K/V tile
    constexpr int PF = NSLOT - 1;   // tiles in flight
    stage(0, tile_of(0) * KVB);
#pragma unroll
    for (int i = 1; i < PF; ++i)
        if (ntile > i) stage(i, tile_of(i) * KVB);
    // The Q fragments came from ordinary global loads; while a DMA is in flight hipcc would protect their first use in
    // the loop with vmcnt(0) EVERY iteration (draining the ring).  Wait once here and pass them through an empty asm so
    // the compiler sees them as ready registers.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int fq = 0; fq < 2; ++fq)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(qf[fq][kk]));
    if constexpr (VAR & 2) {
        // de-phase the workgroups that share a CU: identical programs started together stay in lock-step (all in their MFMA
        // segment, then all in their softmax segment), so a start delay spread over one tile period lets one wave's VALU work
        // run beside another's MFMAs
        const int d = (blockIdx.x * 5) % 12;
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(2);   // 2 x 64 cycles per step: 0 .. ~1400 cycles
    }
    // one K/V tile; `slot` = its ring slot — a compile-time constant in the 2-slot kernel (the loop below is unrolled by the
    // ring), so every fragment read address is a loop-invariant lane base plus an immediate
    // `nks_c` = 32-key halves of the tile that are computed: 2, or 1 for a LAST tile whose valid keys all sit in its first half
    // (1374 tokens = 21 tiles + 30 keys, 905 = 14 tiles + 9): half the S^T MFMAs, exponentials and PV MFMAs of that tile.
    // A wave whose 32 queries all lie beyond the padded sequence (`idle`: the last query block of 1374 / 905 tokens holds 94 / 9
    // queries of 128) only takes part in the tile's DMA and barrier.
    // `first_c` (compile-time): the launch's first tile, peeled off the loop — it always moves the softmax reference (PRE)
    auto tile_body = [&](int t, auto slot_c, auto nks_c, auto first_c) {
        const int slot = slot_c;
        constexpr int NKS = decltype(nks_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        // tiles t+1 .. t+PF-1 may stay in flight (4 DMA instructions per tile per wave); near the end fewer are pending
        const int ahead = min(PF - 1, ntile - 1 - t);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + PF < ntile) stage(slot >= 1 ? slot - 1 : NSLOT - 1, tile_of(t + PF) * KVB);   // slot of step t-1 = (slot+PF) % NSLOT
        if (idle) return;
        const char* sb = smem + slot * STAGE;
        const int kv0 = tile_of(t) * KVB;

        // ---- S^T = K Q^T -------------------------------------------------------------------------
        // the first d-half starts from the inline constant 0 as the MFMA's C operand: no accumulator zeroing (32 v_mov per tile and
        // wave, a quarter of the loop's vector-ALU instructions in the r03 PMC run)
        f32x4_t s[4][2];
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = (((kk << 2) | lg) ^ keyK) << 4;
#pragma unroll
            for (int fk = 0; fk < 2 * NKS; ++fk) {
                const bf16x8_t kf = *(const bf16x8_t*)(sb + baseK + (32 * (fk >> 1) + 4 * (fk & 1)) * ROWB + slot);
#pragma unroll
                for (int fq = 0; fq < 2; ++fq)
                    s[fk][fq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[fq][kk], kk == 0 ? (PRE ? nref[fq] : f32x4_t{0.f, 0.f, 0.f, 0.f}) : s[fk][fq], 0, 0, 0);
            }
        }
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(0);
        if (kv0 + KVB > p.n_tok) {  // wave-uniform: only the last tile masks
#pragma unroll
            for (int fk = 0; fk < 2 * NKS; ++fk)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kv0 + 32 * (fk >> 1) + 8 * lg + 4 * (fk & 1) + r >= p.n_tok) {
                        s[fk][0][r] = -1e30f;
                        s[fk][1][r] = -1e30f;
                    }
        }

        // ---- V^T fragments of the whole tile: issued now, their LDS latency hides behind the first softmax -----------
        // contraction slot (lg, j) <-> key 32 ks + 8 lg + j: one 16-byte read per fragment (slot 4 ks + lg), conflict free
        // under the same XOR key as the GEMM's permuted-row operand.  (bf16-typed like the K reads: an integer-typed LDS
        // load makes hipcc protect it against the in-flight LDS DMA with a vmcnt(0), draining the ring every tile.)
        constexpr bool JITV = (VAR & 16) != 0;
        bf16x8_t vf[JITV ? 1 : 2][4];
        if constexpr (!JITV) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int sv = (((ks << 2) | lg) ^ keyV) << 4;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) vf[ks][fd] = *(const bf16x8_t*)(sb + baseV + fd * 4 * ROWB + sv);
        }
        }

        // ---- online softmax of one 16-query half (per query column; 4 lanes lg=0..3 share a query) ---------------------
        // Issue slots are the bound here (one VALU-class instruction per SIMD per 4 clocks, MFMA included), so:
        //  * the running maximum is LAZY: it is raised (and O, l rescaled) only when some query of the wave exceeds its
        //    reference by more than LAZY_THR logits; until then probabilities are taken against the old reference and may
        //    reach 2^8 — harmless in fp32 accumulators and in bf16 P (same relative precision), and the common case skips
        //    the exp / 18 multiplies of the rescale;
        //  * the row sums come from the matrix pipe: a fifth "V^T" fragment of ones accumulates sum_k P[k,q] (over all
        //    four lanes' keys at once) in lacc, replacing 16 packed adds and the cross-lane reduction at the end.
        bf16x8_t pf[2][2];  // [fq][ks]  B-operand fragments of P^T
        auto softmax_max = [&](int fq) {
            float mx;   // one statement: between separate asm statements hipcc pads every dependent pair with an s_nop
            if constexpr (NKS == 1)
                asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8"
                    : "=&v"(mx)
                    : "v"(s[0][fq][0]), "v"(s[0][fq][1]), "v"(s[0][fq][2]), "v"(s[0][fq][3]), "v"(s[1][fq][0]), "v"(s[1][fq][1]),
                      "v"(s[1][fq][2]), "v"(s[1][fq][3]));
            else
            asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"
                "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max_f32 %0, %0, %16"
                : "=&v"(mx)
                : "v"(s[0][fq][0]), "v"(s[0][fq][1]), "v"(s[0][fq][2]), "v"(s[0][fq][3]), "v"(s[1][fq][0]), "v"(s[1][fq][1]),
                  "v"(s[1][fq][2]), "v"(s[1][fq][3]), "v"(s[2][fq][0]), "v"(s[2][fq][1]), "v"(s[2][fq][2]), "v"(s[2][fq][3]),
                  "v"(s[3][fq][0]), "v"(s[3][fq][1]), "v"(s[3][fq][2]), "v"(s[3][fq][3]));
            // The lazy test needs no cross-lane work: "some query's row maximum exceeds its reference + LAZY_THR" is the same
            // predicate as "some LANE's local maximum does" (a row maximum is the maximum of its 4 lanes).  Only the rare
            // rescale reduces over the 4 lanes sharing a query (lane bits 4 and 5: VALU row/half swaps, not ds_bpermute).
            if constexpr (PRE) {
                // accumulators are already (logit - reference) in log2 units: the lazy test is a compare with a constant.  The FIRST
                // tile always moves the reference to its row maximum (from 0, which may sit far above every logit).
                constexpr float THR2 = LAZY_THR * 1.4426950408889634f / 8.0f;
                if (FIRST || __builtin_amdgcn_ballot_w64(mx > THR2) != 0) {   // wave-uniform
                    mx = xlane_max16(mx);
                    mx = xlane_max32(mx);
                    const float delta = FIRST ? mx : vmax2(mx, 0.f);   // how far the reference moves (>= 0 after the first tile)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nref[fq][r] -= delta;
#pragma unroll
                    for (int fk = 0; fk < 2 * NKS; ++fk)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[fk][fq][r] -= delta;
                    if constexpr (!FIRST) {   // (the first tile finds O and l at zero: nothing to rescale)
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        lacc[fq][0] *= alpha;
                        if constexpr (VSUM) vsum[fq] *= alpha;
#pragma unroll
                        for (int fd = 0; fd < 4; ++fd)
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[fd][fq][r] *= alpha;
                    }
                }
            } else
            if (__builtin_amdgcn_ballot_w64(mx > mthr[fq]) != 0) {   // wave-uniform
                mx = xlane_max16(mx);
                mx = xlane_max32(mx);
                const float mnew = vmax2(mrow[fq], mx);
                // raw v_exp_f32 (results below 2^-126 flush to 0, which is what a masked / negligible weight should be)
                const float alpha = __builtin_amdgcn_exp2f((mrow[fq] - mnew) * p.scale_log2e);
                mrow[fq] = mnew;
                mthr[fq] = mnew + LAZY_THR;
                nmb[fq] = -(mnew * p.scale_log2e);
                lacc[fq][0] *= alpha;
#pragma unroll
                for (int fd = 0; fd < 4; ++fd)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[fd][fq][r] *= alpha;
            }
        };
        auto softmax_exp = [&](int fq) {
            float pv[4][4];
            if constexpr (PRE) {
#pragma unroll
                for (int fk = 0; fk < 2 * NKS; ++fk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[fk][r] = __builtin_amdgcn_exp2f(s[fk][fq][r]);
            } else if constexpr (VAR & 128) {   // lab: one packed FMA per two elements (the round-3 form; 4.5 % slower, same bits)
                const f32x2_t sc2 = {p.scale_log2e, p.scale_log2e};
                const f32x2_t mb2 = {nmb[fq], nmb[fq]};
#pragma unroll
                for (int fk = 0; fk < 2 * NKS; ++fk)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2_t a = {s[fk][fq][r], s[fk][fq][r + 1]};
                        const f32x2_t e = __builtin_elementwise_fma(a, sc2, mb2);
                        pv[fk][r] = __builtin_amdgcn_exp2f(e[0]);
                        pv[fk][r + 1] = __builtin_amdgcn_exp2f(e[1]);
                    }
            } else {
                // one plain v_fma_f32 per element, as asm so that hipcc does not pair them into v_pk_fma_f32: a packed fp32 instruction
                // costs more than its two halves beside MFMAs on this part (903 -> 944 TFLOP/s at 1374 tokens, identical bits)
#pragma unroll
                for (int fk = 0; fk < 2 * NKS; ++fk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e;
                        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(e) : "v"(s[fk][fq][r]), "v"(p.scale_log2e), "v"(nmb[fq]));
                        pv[fk][r] = __builtin_amdgcn_exp2f(e);
                    }
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                uint4 w;
                w.x = pack_bf2(pv[2 * ks][0], pv[2 * ks][1]);
                w.y = pack_bf2(pv[2 * ks][2], pv[2 * ks][3]);
                w.z = pack_bf2(pv[2 * ks + 1][0], pv[2 * ks + 1][1]);
                w.w = pack_bf2(pv[2 * ks + 1][2], pv[2 * ks + 1][3]);
                pf[fq][ks] = __builtin_bit_cast(bf16x8_t, w);
            }
        };
        // ---- O^T += V^T P^T (and l += 1^T P^T) for one half ------------------------------------------------------------
        auto pv_half = [&](int fq) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int fd = 0; fd < 4; ++fd)
                    o[fd][fq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[JITV ? 0 : ks][fd], pf[fq][ks], o[fd][fq], 0, 0, 0);
                lacc[fq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[fq][ks], lacc[fq], 0, 0, 0);
            }
        };

        if constexpr (JITV) {
            softmax_max(0);
            softmax_exp(0);
            softmax_max(1);
            softmax_exp(1);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int sv = (((ks << 2) | lg) ^ keyV) << 4;
#pragma unroll
                for (int fd = 0; fd < 4; ++fd) vf[0][fd] = *(const bf16x8_t*)(sb + baseV + fd * 4 * ROWB + sv);
#pragma unroll
                for (int fq = 0; fq < 2; ++fq) {
#pragma unroll
                    for (int fd = 0; fd < 4; ++fd)
                        o[fd][fq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0][fd], pf[fq][ks], o[fd][fq], 0, 0, 0);
                    if constexpr (VSUM) {
                        typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
                        const uint4 pw = __builtin_bit_cast(uint4, pf[fq][ks]);
                        const bf16x2_hw one2 = __builtin_bit_cast(bf16x2_hw, 0x3f803f80u);
                        vsum[fq] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, pw.x), one2, vsum[fq], false);
                        vsum[fq] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, pw.y), one2, vsum[fq], false);
                        vsum[fq] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, pw.z), one2, vsum[fq], false);
                        vsum[fq] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw, pw.w), one2, vsum[fq], false);
                    } else {
                        lacc[fq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[fq][ks], lacc[fq], 0, 0, 0);
                    }
                }
            }
            return;
        }
        softmax_max(0);
        softmax_exp(0);
        softmax_max(1);
        __builtin_amdgcn_sched_barrier(0);
        // half 0's ten MFMAs are spread between slices of half 1's exp / convert work.  Measured +-2 % against issuing
        // them as a block (profiles/r01_ab.md): a SIMD's matrix and vector pipes barely overlap on this part, so the
        // kernel's bound is MFMA time + VALU time per tile and the lever was removing VALU work, not scheduling it.
        softmax_exp(1);
        pv_half(0);
#pragma unroll
        for (int g = 0; g < 5 * NKS; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(1);
        pv_half(1);
        if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(0);
    };
    using full_c = std::integral_constant<int, 2>;
    using half_c = std::integral_constant<int, 1>;
    using yes_c = std::true_type;
    using no_c = std::false_type;
    if constexpr (SHORT) {
        static_assert(NSLOT == 2, "the short-tail prologue is written for the 2-slot ring");
        tile_body(0, std::integral_constant<int, 0>{}, half_c{}, yes_c{});
        for (int t = 1; t < ntile; t += 2) {
            tile_body(t, std::integral_constant<int, 1>{}, full_c{}, no_c{});
            if (t + 1 < ntile) tile_body(t + 1, std::integral_constant<int, 0>{}, full_c{}, no_c{});
        }
    } else if constexpr (NSLOT == 2 && PRE) {
        tile_body(0, std::integral_constant<int, 0>{}, full_c{}, yes_c{});
        for (int t = 1; t < ntile; t += 2) {
            tile_body(t, std::integral_constant<int, 1>{}, full_c{}, no_c{});
            if (t + 1 < ntile) tile_body(t + 1, std::integral_constant<int, 0>{}, full_c{}, no_c{});
        }
    } else if constexpr (NSLOT == 2) {   // (unscaled q: the running maximum starts at -1e30, the first tile needs no special case)
        for (int t = 0; t < ntile; t += 2) {
            tile_body(t, std::integral_constant<int, 0>{}, full_c{}, no_c{});
            if (t + 1 < ntile) tile_body(t + 1, std::integral_constant<int, 1>{}, full_c{}, no_c{});
        }
    } else {
        tile_body(0, 0, full_c{}, yes_c{});
        int slot = NSLOT > 1 ? 1 : 0;
        for (int t = 1; t < ntile; ++t) {
            tile_body(t, slot, full_c{}, no_c{});
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
    }

    // ---- normalise and store: lane owns 16 consecutive d (= 16 lg + 4 fd + r) of query li ---------
#pragma unroll
    for (int fq = 0; fq < 2; ++fq) {
        float rowsum = lacc[fq][0];             // D-row 4 lg of the ones product: the full row sum on every lane
        if constexpr (VSUM) {                   // per-lane partials over this lane's keys: add the query's four lanes (lane bits 4, 5)
            rowsum = vsum[fq];
            rowsum += lane_xor<16>(rowsum);
            rowsum += lane_xor<32>(rowsum);
        }
        const float inv = 1.0f / rowsum;
        const int q = q0 + 16 * fq + li;
        if (q < p.npad) {
            uint4 w0, w1;
            w0.x = pack_bf2(o[0][fq][0] * inv, o[0][fq][1] * inv);
            w0.y = pack_bf2(o[0][fq][2] * inv, o[0][fq][3] * inv);
            w0.z = pack_bf2(o[1][fq][0] * inv, o[1][fq][1] * inv);
            w0.w = pack_bf2(o[1][fq][2] * inv, o[1][fq][3] * inv);
            w1.x = pack_bf2(o[2][fq][0] * inv, o[2][fq][1] * inv);
            w1.y = pack_bf2(o[2][fq][2] * inv, o[2][fq][3] * inv);
            w1.z = pack_bf2(o[3][fq][0] * inv, o[3][fq][1] * inv);
            w1.w = pack_bf2(o[3][fq][2] * inv, o[3][fq][3] * inv);
            uint4* dst = (uint4*)(p.O + (rowbase + q) * p.ldo + h * HD + lg * 16);
            dst[0] = w0;
            dst[1] = w1;
        }
    }
}

}  // namespace

// QK [B*npad, 2D] (ldqk elements), Vt [B,H,64,npad], O [B*npad, D] (ldo elements).
// q_prescaled: the q columns hold q * log2(e) / sqrt(64) (the ViT path folds that factor into the q rows of the LayerNorm-folded qkv
// weights, vit_misc.hip ln_fold_kernel), so S^T accumulates straight into base-2 exponents relative to the running reference.
int fp_attention_fwd(const bf16_t* QK, int ldqk, const bf16_t* Vt, bf16_t* O, int ldo, int B, int H,
                     int n_tok, int npad, bool q_prescaled, hipStream_t stream) {
    FP_REQUIRE(B > 0 && H > 0 && n_tok > 0 && npad >= n_tok && npad % 16 == 0,
               "attention: bad shape B=%d H=%d n_tok=%d npad=%d", B, H, n_tok, npad);
    FP_REQUIRE(npad >= 8, "attention: npad too small");
    AttnArgs a;
    a.QK = QK; a.ldqk = ldqk; a.Vt = Vt; a.O = O; a.ldo = ldo;
    a.B = B; a.H = H; a.n_tok = n_tok; a.npad = npad; a.D = H * HD;
    a.scale_log2e = FP_ATTN_QSCALE;
    a.q_base = 0;
    dim3 grid(cdiv(npad - a.q_base, QB) * H * B);
    // a last K/V tile whose valid keys all sit in its first half is processed first by a half-length body (kernel flavour 4)
    const bool short_tail = n_tok > KVB && n_tok - (cdiv(n_tok, KVB) - 1) * KVB <= KVB / 2;
#ifdef FP_LAB   // lab build: LDS ring depth 2 (default) / 3 / 4 and the archived flavours (profiles/r01_ab.md, r03_ab.md) for A/B runs
    const int nslot = fp_opt_get(FP_OPT_ATTN_SLOTS, 2);
    static int env_var = [] { const char* e = getenv("FP_ATTN_VARIANT"); return e ? atoi(e) : 0; }();
    const int avar = fp_opt_get(FP_OPT_ATTN_VARIANT, env_var);
    if (!q_prescaled) {
    if (nslot == 2 && (avar & 2)) { hipLaunchKernelGGL((attn_fwd_kernel<2, 2>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a); FP_LAUNCH_CHECK(); return FP_OK; }
    if (nslot == 2 && (avar & 1)) { hipLaunchKernelGGL((attn_fwd_kernel<2, 1>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a); FP_LAUNCH_CHECK(); return FP_OK; }
    if (nslot == 4) { hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, dim3(NWAVE * 64), 4 * STAGE, stream, a); FP_LAUNCH_CHECK(); return FP_OK; }
    if (nslot == 3) { hipLaunchKernelGGL(attn_fwd_kernel<3>, grid, dim3(NWAVE * 64), 3 * STAGE, stream, a); FP_LAUNCH_CHECK(); return FP_OK; }
    if (avar & 8) { hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, dim3(NWAVE * 64), 2 * STAGE, stream, a); FP_LAUNCH_CHECK(); return FP_OK; }   // short-tail off
    if ((avar & 128) && !q_prescaled) {   // packed multiply-add in the softmax (the form shipped until round 4)
        if (short_tail) hipLaunchKernelGGL((attn_fwd_kernel<2, 4 | 16 | 128>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<2, 16 | 128>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        FP_LAUNCH_CHECK();
        return FP_OK;
    }
    if (avar & 32) {   // the round-3 flavour: three waves per SIMD, whole-tile V^T prefetch, half 0's PV under half 1's exponentials
        if (short_tail) hipLaunchKernelGGL((attn_fwd_kernel<2, 4>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        else hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        FP_LAUNCH_CHECK();
        return FP_OK;
    }
    }
#endif
#ifdef FP_LAB
    if (q_prescaled && (fp_opt_get(FP_OPT_ATTN_VARIANT, 0) & 256)) {   // lab A/B: row sums by v_dot2_f32_bf16 instead of the ones MFMA
        if (short_tail) hipLaunchKernelGGL((attn_fwd_kernel<2, 4 | 16 | 64 | 256>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<2, 16 | 64 | 256>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        FP_LAUNCH_CHECK();
        return FP_OK;
    }
#endif
    if (q_prescaled) {
        if (short_tail) hipLaunchKernelGGL((attn_fwd_kernel<2, 4 | 16 | 64>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<2, 16 | 64>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
    } else {
        if (short_tail) hipLaunchKernelGGL((attn_fwd_kernel<2, 4 | 16>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<2, 16>), grid, dim3(NWAVE * 64), 2 * STAGE, stream, a);
    }
    FP_LAUNCH_CHECK();
    return FP_OK;
}
