// ARCHIVED EXPERIMENT (not built into the library; results in profiles/r01_ab.md, variant 22).  Build check:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I freepose_amd/csrc -c tools/experiments/gemm_ap.hip -o /dev/null
// Anti-phase bf16 MFMA GEMM (gfx950): same math, operand layout and fused epilogues as gemm_bf16.hip, different
// schedule.  rocprofv3 on the double-buffered kernel showed 0 LDS bank conflicts but only ~50 % MFMA-busy: its 8 waves
// run in lockstep, so the two waves sharing a SIMD want the matrix pipe at the same time and the memory path at the
// same time, and 32 % of wave time is parked at the per-tile vmcnt(0)+barrier.  Here:
//
//   * K is consumed in 32-deep steps; a step's operand slabs (256 x 64 B of X, 256 x 64 B of W = 32 KiB) live in a
//     4-slot LDS ring (128 KiB), so three steps are always in flight behind the one being consumed;
//   * the workgroup's waves form two groups (wave rows wm = 0 / 1; waves w and w+4 share a SIMD) that run the SAME
//     program ONE barrier-slot apart: while group A issues its 32 MFMAs of step j ("M slot"), group B issues the DMA
//     for step j+3 and reads its 12 fragments of step j ("L slot"), then the roles swap.  Every slot ends in s_barrier,
//     which keeps the groups in anti-phase: each SIMD always has one wave feeding the matrix pipe and one wave on the
//     memory path;
//   * counted waits only: a wave issues 4 DMA pieces per step, so `vmcnt(8)` = "everything up to step j+1 has landed,
//     j+2 and j+3 may still fly".  A DMA reuses the slot of step j-1 only after the barrier that follows the last read
//     of that step (group B's L slot).
//
// Slot calendar (global slot g = 1, 2, ...):  group A runs L(j) at g = 2j+1 and M(j) at g = 2j+2; group B starts one
// barrier later: L(j) at g = 2j+2, M(j) at g = 2j+3.  At the end of every slot each wave waits until its own pieces of
// the step that is read in the NEXT slot have landed, then all 8 waves meet at the barrier.
//
// LDS image: rows of 64 B (4 slots of 16 B).  A ds_read_b128 lane group spans four rows x four slots of a 256-B bank
// row; slot index is XORed with KEY[(row>>2)&3] (plain rows) / KEY[(row>>4)&3] (permuted rows), KEY = {0,3,2,1}, which
// makes all 16 lanes of every lane group hit distinct 16-B units.  global_load_lds writes lane-linearly, so the same
// permutation is applied to each lane's SOURCE address.
#include "gemm_bf16.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
constexpr int TM = 8, TN = 4;
constexpr int KS = 32;                  // K elements per step
constexpr int SROW = 64;                // bytes per LDS row
constexpr int XB = BM * SROW;           // 16 KiB
constexpr int STG = (BM + BN) * SROW;   // 32 KiB per ring slot
constexpr int NST = 4;

__device__ __forceinline__ int keyq(int q) { return (0x1230 >> (4 * (q & 3))) & 3; }  // {0,3,2,1}[q]

template <int ALLOW>
__device__ __forceinline__ void wait_vm_steps() {
    if constexpr (ALLOW <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (ALLOW == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (ALLOW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
__device__ __forceinline__ void wait_vm_dyn(int allow) {
    if (allow <= 0) wait_vm_steps<0>();
    else if (allow == 1) wait_vm_steps<1>();
    else if (allow == 2) wait_vm_steps<2>();
}
__device__ __forceinline__ void slot_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
}

template <int EPI, int VAR>
__global__ __launch_bounds__(NW * 64) void gemm_ap_kernel(FpGemmArgs p) {
    constexpr bool TRANS = (EPI == FP_EPI_VT);
    constexpr int TR = TRANS ? TM : TN;
    constexpr int TC = TRANS ? TN : TM;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool groupB = wm != 0;
    char* epi_stage = smem + NST * STG + wave * fp_gemm::EPI_STAGE_BYTES;   // row-coalescing slab of the epilogue

    int tile_m, tile_n;
    fp_gemm_tile(blockIdx.x, gridDim.x, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN, tile_m, tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- DMA: each wave moves 2 pieces (16 rows x 64 B) of the X slab and 2 of the W slab per step --------------
    uint32_t offX[2], offW[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (it * NW + wave) * 16 + (lane >> 2);
        // permuted rows are laid out a*(4*TR) + 4f + b: the key must follow a = row / (4*TR)  (>>5 for TR=8, >>4 for TR=4)
        const int kx = TRANS ? keyq(row >> 5) : keyq(row >> 2);
        const int kw = TRANS ? keyq(row >> 2) : keyq(row >> 4);
        offX[it] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)p.ldx * 2u + (((lane & 3) ^ kx) << 4);
        offW[it] = (uint32_t)min(n0 + row, p.N - 1) * (uint32_t)p.ldw * 2u + (((lane & 3) ^ kw) << 4);
    }
    const char* gX = (const char*)p.X;
    const char* gW = (const char*)p.W;
    auto issue = [&](int step) {
        char* sb = smem + (step & (NST - 1)) * STG;
        const size_t kb = (size_t)step * (KS * 2);
#pragma unroll
        for (int it = 0; it < 2; ++it) glds16(gX + offX[it] + kb, sb + (it * NW + wave) * 1024);
#pragma unroll
        for (int it = 0; it < 2; ++it) glds16(gW + offW[it] + kb, sb + XB + (it * NW + wave) * 1024);
    };

    // ---- fragment addresses (R operand: permuted rows, A slot; C operand: plain rows, B slot) -----------------------
    const int li = lane & 15, lg = lane >> 4;
    const int tile_r = TRANS ? wm * (16 * TM) : wn * (16 * TN);
    const int tile_c = TRANS ? wn * (16 * TN) : wm * (16 * TM);
    const int rowR0 = tile_r + (li >> 2) * 4 * TR + (li & 3);              // + 4 f
    const int rowC0 = tile_c + li;                                           // + 16 f
    const int baseR = (TRANS ? 0 : XB) + rowR0 * SROW;
    const int baseC = (TRANS ? XB : 0) + rowC0 * SROW;
    const int slotR = (lg ^ keyq(li >> 2)) << 4;   // permuted: row / (4*TR) = li>>2 for every fragment f
    const int slotC = (lg ^ keyq(li >> 2)) << 4;   // plain: (row>>2)&3 = li>>2 for every fragment f

    f32x4_t acc[TC][TR];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fr[TR], fc[TC];

    auto load_frags = [&](int step) {
        const char* sb = smem + (step & (NST - 1)) * STG;
#pragma unroll
        for (int f = 0; f < TR; ++f) fr[f] = *(const bf16x8_t*)(sb + baseR + f * 4 * SROW + slotR);
#pragma unroll
        for (int f = 0; f < TC; ++f) fc[f] = *(const bf16x8_t*)(sb + baseC + f * 16 * SROW + slotC);
    };
    auto mma = [&]() {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TR; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[j], fc[i], acc[i][j], 0, 0, 0);
    };

    const int nsteps = p.K / KS;
    // ---- prologue: three steps in flight, step 0 visible to everyone, then group B drops one slot behind ----------
    issue(0);
    if (nsteps > 1) issue(1);
    if (nsteps > 2) issue(2);
    int issued = min(2, nsteps - 1);
    wait_vm_dyn(issued);                 // own pieces of step 0 landed
    slot_barrier();
    if (groupB) slot_barrier();

    for (int j = 0; j < nsteps; ++j) {
        // ---------------- L slot: DMA for step j+3, fragments of step j ----------------
        if (j + 3 < nsteps) {
            if constexpr (!(VAR & 64)) issue(j + 3);   // ablation bit 64: no in-loop DMA (timing probe, wrong results)
            issued = (VAR & 64) ? issued : j + 3;
        }
        load_frags(j);
        {   // next slot's reader: group B reads step j (A's view) / group A reads step j+1 (B's view)
            const int needed = groupB ? j + 1 : j;
            if (needed < nsteps) wait_vm_dyn(issued - needed);
        }
        slot_barrier();
        // ---------------- M slot: 32 MFMAs of step j ----------------
        if constexpr (!(VAR & 32)) mma();             // ablation bit 32: no MFMAs (timing probe, wrong results)
        else {
#pragma unroll
            for (int f = 0; f < TR; ++f) asm volatile("" ::"v"(fr[f]));
#pragma unroll
            for (int f = 0; f < TC; ++f) asm volatile("" ::"v"(fc[f]));
        }
        if (j + 1 < nsteps) {
            wait_vm_dyn(issued - (j + 1));   // step j+1 is read in the next slot (by A: L(j+1); by B in the one after)
            slot_barrier();
        } else if (!groupB) {
            slot_barrier();                  // A's last barrier pairs with B's barrier after its last L slot
        }
    }
    fp_gemm::epilogue<BM, BN, WM, WN, EPI, (VAR & 4), TC, TR>(p, acc, m0, n0, wm, wn, li, lg, epi_stage);
}

template <int EPI, int VAR = 4>
int launch_ap(const FpGemmArgs& a, hipStream_t stream) {
    constexpr int SMEM = NST * STG + NW * fp_gemm::EPI_STAGE_BYTES;
    auto kern = gemm_ap_kernel<EPI, VAR>;
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

// anti-phase schedule; caller guarantees K % 32 == 0 and the usual gemm preconditions
int fp_gemm_bf16_ap(const FpGemmArgs& a, int epi, hipStream_t stream) {
    switch (epi) {
        case FP_EPI_BIAS: return launch_ap<FP_EPI_BIAS>(a, stream);
        case FP_EPI_BIAS_GELU: return launch_ap<FP_EPI_BIAS_GELU>(a, stream);
        case FP_EPI_BIAS_LS_RES: {
            const int abl = fp_opt_get(FP_OPT_GEMM_VARIANT, 0) & (32 | 64);   // timing ablations (A/B probing only)
            if (abl == 32) return launch_ap<FP_EPI_BIAS_LS_RES, 4 | 32>(a, stream);
            if (abl == 64) return launch_ap<FP_EPI_BIAS_LS_RES, 4 | 64>(a, stream);
            if (abl == 96) return launch_ap<FP_EPI_BIAS_LS_RES, 4 | 96>(a, stream);
            return launch_ap<FP_EPI_BIAS_LS_RES>(a, stream);
        }
        case FP_EPI_PATCH: return launch_ap<FP_EPI_PATCH>(a, stream);
        case FP_EPI_VT: return launch_ap<FP_EPI_VT>(a, stream);
        default: fp_set_error("gemm_ap: unknown epilogue %d", epi); return FP_ERR_INVALID;
    }
}
