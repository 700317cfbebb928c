// ARCHIVED EXPERIMENT (not built into the library; result in profiles/r01_ab.md): one-wave-per-SIMD bf16 MFMA GEMM (gfx950).
// Same math / operand layout / fused epilogues as gemm_bf16.hip.  Build check:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I freepose_amd/csrc -c tools/experiments/gemm_w1.hip -o /dev/null
//
// Why: the 8-wave 256x256 kernels are bound by operand supply through LDS (profiles/r01_ab.md: 0 bank conflicts, ~50 %
// MFMA-busy; the DMA fill alone takes ~80 % of the loop time and three different wave schedules converge).  The only
// lever left is work per LDS byte, and that is capped by accumulator registers.  With ONE wave per SIMD a wave owns the
// whole 512-entry register file, so:
//   * workgroup tile 320 x 256 (4 waves as 2 x 2, wave tile 160 x 128): 142 flop per byte of LDS fill (+11 %), and
//     18 fragment reads per 80 MFMAs instead of 12 per 32 (-40 % ds_read traffic per MFMA);
//   * no second wave competes for the SIMD's matrix pipe: the wave's own instruction stream interleaves, per row of 8
//     MFMAs, one global_load_lds (DMA for step j+3) and the ds_reads that refill, for step j+1, exactly the operand
//     registers the row has just finished with (X fragment i) or the other half of a double-buffered set (W fragments);
//   * K in 32-deep steps, 4-slot LDS ring (36 KiB per slot), one counted vmcnt + one barrier per step.
//
// Invariant at the top of step j: the fragments of step j are in registers; step j+1 has landed and is visible to all
// four waves.  During step j: DMA for step j+3 goes into the slot last read during step j-2; the reads issued are of
// step j+1.  End of step j: wait own pieces of step j+2 (vmcnt(9): only step j+3's nine pieces may remain), barrier.
#include "gemm_bf16.h"
#include "gemm_epilogue.h"

#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, WM = 2, WN = 2, NW = 4;
constexpr int TM = 8, TN = 8;           // 16-row fragments per wave: X (plain rows, MFMA B slot), W (permuted, A slot)
constexpr int KS = 32, SROW = 64;
constexpr int XB = BM * SROW;           // 20 KiB
constexpr int STG = (BM + BN) * SROW;   // 36 KiB
constexpr int NST = 4;
constexpr int PIECES = STG / 1024 / NW; // 9 DMA pieces (16 rows x 64 B) per wave per step

__device__ __forceinline__ int keyq(int q) { return (0x1230 >> (4 * (q & 3))) & 3; }  // {0,3,2,1}[q]

template <int EPI>
__global__ __launch_bounds__(NW * 64, 1) void gemm_w1_kernel(FpGemmArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int tile_m, tile_n;
    fp_gemm_tile(blockIdx.x, gridDim.x, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- DMA: 36 pieces per step (20 of X, 16 of W); wave w moves pieces w, w+4, ..., w+32 -------------------------
    uint32_t off[PIECES];
    int ldsoff[PIECES];
    bool isW[PIECES];
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int piece = it * NW + wave;                 // 0..35
        const bool w = piece >= BM / 16;
        const int row = (w ? piece - BM / 16 : piece) * 16 + (lane >> 2);
        const int key = w ? keyq(row >> 4) : keyq(row >> 2);   // W rows: a*16+4f+b inside 64-row groups; X rows: plain
        const int ks = ((lane & 3) ^ key) << 4;
        isW[it] = w;
        off[it] = w ? (uint32_t)min(n0 + row, p.N - 1) * (uint32_t)p.ldw * 2u + ks
                    : (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)p.ldx * 2u + ks;
        ldsoff[it] = piece * 1024;                        // X pieces first, W pieces follow (XB = 20 pieces)
    }
    const char* gX = (const char*)p.X;
    const char* gW = (const char*)p.W;
    auto issue_piece = [&](int step, int it) {
        char* sb = smem + (step & (NST - 1)) * STG;
        glds16((isW[it] ? gW : gX) + off[it] + (size_t)step * (KS * 2), sb + ldsoff[it]);
    };

    // ---- fragment addresses ---------------------------------------------------------------------------------------
    const int li = lane & 15, lg = lane >> 4;
    const int slot = (lg ^ keyq(li >> 2)) << 4;            // same form for plain rows and for permuted 64-row groups
    const int baseC = (wm * (16 * TM) + li) * SROW + slot;                                   // + 16 f rows
    const int baseR = XB + (wn * (16 * TN) + (li >> 2) * 16 + (li & 3)) * SROW + slot;       // + (grp*64 + 4 f') rows
    auto addrC = [&](int step, int f) { return smem + (step & (NST - 1)) * STG + baseC + f * 16 * SROW; };
    auto addrR = [&](int step, int f) { return smem + (step & (NST - 1)) * STG + baseR + ((f >> 2) * 64 + (f & 3) * 4) * SROW; };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fc[TM], frA[TN], frB[TN];

    const int nsteps = p.K / KS;
    // ---- prologue ----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < PIECES; ++it) issue_piece(0, it);
    if (nsteps > 1) {
#pragma unroll
        for (int it = 0; it < PIECES; ++it) issue_piece(1, it);
    }
    if (nsteps > 2) {
#pragma unroll
        for (int it = 0; it < PIECES; ++it) issue_piece(2, it);
    }
    if (nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < TM; ++f) fc[f] = *(const bf16x8_t*)addrC(0, f);
#pragma unroll
    for (int f = 0; f < TN; ++f) frA[f] = *(const bf16x8_t*)addrR(0, f);
    if (nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");     // step 1 landed (step 2 may fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one k-step: 10 rows of 8 MFMAs; after row i its X fragment register is refilled for step j+1, rows 0..7 also
    // refill W fragment i of the OTHER set, rows 0..8 issue one DMA piece of step j+3
    auto step_body = [&](int j, bf16x8_t (&cur)[TN], bf16x8_t (&nxt)[TN], auto dma_c, auto more_c) {
        constexpr bool dma = decltype(dma_c)::value, more = decltype(more_c)::value;   // compile-time: no branches in the rows
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[jn], fc[i], acc[i][jn], 0, 0, 0);
            if constexpr (dma) { if (i < PIECES) issue_piece(j + 3, i); }
            if constexpr (more) {
                fc[i] = *(const bf16x8_t*)addrC(j + 1, i);
                if (i < TN) nxt[i] = *(const bf16x8_t*)addrR(j + 1, i);
            }
            // pin the interleave: 8 MFMA, then the memory ops of this row
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        if constexpr (dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");   // own pieces of step j+2 landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    int j = 0;
    for (; j + 4 < nsteps; j += 2) {            // steady state: steps j and j+1 both have j+3 < nsteps
        step_body(j, frA, frB, T{}, T{});
        step_body(j + 1, frB, frA, T{}, T{});
    }
    for (; j < nsteps; ++j) {                   // tail (<= 4 steps): generic flags, register sets alternate by parity
        const bool d = j + 3 < nsteps, m = j + 1 < nsteps;
        if (!(j & 1)) { if (d) step_body(j, frA, frB, T{}, T{}); else if (m) step_body(j, frA, frB, F{}, T{}); else step_body(j, frA, frB, F{}, F{}); }
        else          { if (d) step_body(j, frB, frA, T{}, T{}); else if (m) step_body(j, frB, frA, F{}, T{}); else step_body(j, frB, frA, F{}, F{}); }
    }

    fp_gemm::epilogue<BM, BN, WM, WN, EPI, 4, TM, TN>(p, acc, m0, n0, wm, wn, li, lg,
                                                     smem + NST * STG + wave * fp_gemm::EPI_STAGE_BYTES);
}

template <int EPI>
int launch_w1(const FpGemmArgs& a, hipStream_t stream) {
    constexpr int SMEM = NST * STG + NW * fp_gemm::EPI_STAGE_BYTES;
    auto kern = gemm_w1_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        FP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NW * 64), SMEM, stream, a);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

}  // namespace

// one-wave-per-SIMD schedule; non-transposed epilogues only; caller guarantees K % 32 == 0
int fp_gemm_bf16_w1(const FpGemmArgs& a, int epi, hipStream_t stream) {
    switch (epi) {
        case FP_EPI_BIAS: return launch_w1<FP_EPI_BIAS>(a, stream);
        case FP_EPI_BIAS_GELU: return launch_w1<FP_EPI_BIAS_GELU>(a, stream);
        case FP_EPI_BIAS_LS_RES: return launch_w1<FP_EPI_BIAS_LS_RES>(a, stream);
        case FP_EPI_PATCH: return launch_w1<FP_EPI_PATCH>(a, stream);
        default: fp_set_error("gemm_w1: unsupported epilogue %d", epi); return FP_ERR_INVALID;
    }
}
