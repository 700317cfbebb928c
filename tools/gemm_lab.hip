// GEMM main-loop laboratory (development probe, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab [M N K]...
// C[M,N] = X[M,K] W[N,K]^T, bf16 in, fp32 accumulate, bf16 out (plain direct store: the lab compares MAIN LOOPS).
//   kernel 0 "base16" : the library's round-1 loop — 16 waves (4x4, 64x64 per wave), 2 K-tile buffers, vmcnt(0) + barrier per K-tile
//   kernel 1 "pp8"    : 8 waves (2x4, 128x64 per wave) in two groups that run ONE barrier apart (ping-pong): while one group issues
//                       the 16 MFMAs of a quadrant, the other issues its LDS fragment reads and 2 DMA pieces; a K-tile is 4 such
//                       phases; DMA is issued region by region as soon as a region's last reader has passed a barrier, and waited
//                       for with a COUNTED vmcnt (10 pieces may stay in flight), never vmcnt(0).
// Both are persistent (one workgroup per CU walks tiles) so that short-K shapes include tile turnaround.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    bf16x2_hw v;
    v[0] = static_cast<__bf16>(lo);
    v[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(uint32_t, v);
}

struct Args { const bf16_t* X; const bf16_t* W; bf16_t* C; int M, N, K; };

constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128;
constexpr int STAGE = (BM + BN) * ROWB;   // 64 KiB

__device__ __forceinline__ void tile_of(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    // column strips of 4 n-tiles swept m-major (the library's order, without the XCD remap: ids are already XCD-contiguous below)
    constexpr int SW = 4;
    const int full = tiles_n / SW, tail = tiles_n - full * SW, in_full = full * tiles_m * SW;
    if (t < in_full) { const int strip = t / (tiles_m * SW), rem = t - strip * (tiles_m * SW); tm = rem / SW; tn = strip * SW + (rem - tm * SW); }
    else { const int rem = t - in_full; tm = rem / tail; tn = full * SW + (rem - tm * tail); }
}
__device__ __forceinline__ int xcd_remap(int block, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = block & 7, pos = block >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}

// direct store of a wave's accumulators: lane (li, lg) owns, for token 16 i + li, the 16 features lg*16 .. +15
template <int TC>
__device__ __forceinline__ void store_acc(const Args& p, f32x4_t (&acc)[TC][4], int mrow0, int ncol0, int li, int lg) {
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int m = mrow0 + 16 * i + li;
        const int n = ncol0 + lg * 16;
        if (m < p.M && n < p.N) {
            u32x4_t o0, o1;
            o0.x = pack_bf2(acc[i][0][0], acc[i][0][1]); o0.y = pack_bf2(acc[i][0][2], acc[i][0][3]);
            o0.z = pack_bf2(acc[i][1][0], acc[i][1][1]); o0.w = pack_bf2(acc[i][1][2], acc[i][1][3]);
            o1.x = pack_bf2(acc[i][2][0], acc[i][2][1]); o1.y = pack_bf2(acc[i][2][2], acc[i][2][3]);
            o1.z = pack_bf2(acc[i][3][0], acc[i][3][1]); o1.w = pack_bf2(acc[i][3][2], acc[i][3][3]);
            u32x4_t* op = (u32x4_t*)(p.C + (size_t)m * p.N + n);
            __builtin_nontemporal_store(o0, op);
            __builtin_nontemporal_store(o1, op + 1);
        }
    }
}

// =====================================================================================================================
// kernel 0: 16 waves, plain double buffer (the round-1 library loop, persistent)
// =====================================================================================================================
template <int ORD>   // 0: DMA first (round-1 order), 1: fragment reads of kk=0 first, then DMA, 2: DMA split 2 + 2 around the kk=0 MFMAs, 3: as 2 with s_setprio
__global__ __launch_bounds__(1024) void gemm_base16(Args p) {
    constexpr int NW = 16, WN = 4, TM = 4, TN = 4, IX = BM / 8 / NW, IW = BN / 8 / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int li = lane & 15, lg = lane >> 4;
    uint32_t offX[IX], offW[IW];
    int m0, n0;
    auto set_tile = [&](int t, int& tm0, int& tn0) {
        int tm, tn;
        tile_of(t, tiles_m, tiles_n, tm, tn);
        tm0 = tm * BM; tn0 = tn * BN;
#pragma unroll
        for (int it = 0; it < IX; ++it) {
            const int row = (it * NW + wave) * 8 + (lane >> 3);
            const int key = (row >> 1) & 7;
            offX[it] = (uint32_t)min(tm0 + row, p.M - 1) * (uint32_t)p.K * 2u + (((lane & 7) ^ key) << 4);
        }
#pragma unroll
        for (int it = 0; it < IW; ++it) {
            const int row = (it * NW + wave) * 8 + (lane >> 3);
            const int rl = row & 63, key = (((rl >> 4) << 1) | ((rl & 3) >> 1)) & 7;
            offW[it] = (uint32_t)min(tn0 + row, p.N - 1) * (uint32_t)p.K * 2u + (((lane & 7) ^ key) << 4);
        }
    };
    const int first = xcd_remap(blockIdx.x, gridDim.x), tstride = gridDim.x;
    int tile = first;
    set_tile(tile, m0, n0);
    const char* gX = (const char*)p.X;
    const char* gW = (const char*)p.W;
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * STAGE;
        const size_t kb = (size_t)kt * ROWB;
#pragma unroll
        for (int it = 0; it < IX; ++it) glds16(gX + offX[it] + kb, sb + (it * NW + wave) * 1024);
#pragma unroll
        for (int it = 0; it < IW; ++it) glds16(gW + offW[it] + kb, sb + BM * ROWB + (it * NW + wave) * 1024);
    };
    const int rowR0 = wn * 64 + (li >> 2) * 16 + (li & 3), keyR = (((li >> 2) << 1) | ((li & 3) >> 1)) & 7;
    const int rowC0 = wm * 64 + li, keyC = (li >> 1) & 7;
    const int baseR = BM * ROWB + rowR0 * ROWB, baseC = rowC0 * ROWB;
    f32x4_t acc[TM][TN];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    zero();
    const int nkt = p.K / BK;
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
            const int nbuf = (g + 1) & 1, nk = more ? kt + 1 : 0;
            const bool do_stage = more || has_next;
            auto stage_part = [&](int lo, int hi) {   // DMA pieces lo..hi-1 of this wave's 4 (2 X then 2 W)
                char* sbn = smem + nbuf * STAGE;
                const size_t kb = (size_t)nk * ROWB;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (it < lo || it >= hi) continue;
                    if (it < 2) glds16(gX + offX[it] + kb, sbn + (it * NW + wave) * 1024);
                    else glds16(gW + offW[it - 2] + kb, sbn + BM * ROWB + ((it - 2) * NW + wave) * 1024);
                }
            };
            if (ORD == 0 && do_stage) stage_part(0, 4);
            const char* sb = smem + (g & 1) * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t fr[TN], fc[TM];
                const int slotR = (((kk << 2) | lg) ^ keyR) << 4, slotC = (((kk << 2) | lg) ^ keyC) << 4;
#pragma unroll
                for (int f = 0; f < TN; ++f) fr[f] = *(const bf16x8_t*)(sb + baseR + f * 4 * ROWB + slotR);
#pragma unroll
                for (int f = 0; f < TM; ++f) fc[f] = *(const bf16x8_t*)(sb + baseC + f * 16 * ROWB + slotC);
                if (kk == 0 && do_stage) {
                    if (ORD == 1) stage_part(0, 4);
                    if (ORD >= 2) stage_part(0, 2);
                }
                if (ORD == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[j], fc[i], acc[i][j], 0, 0, 0);
                if (ORD == 3) __builtin_amdgcn_s_setprio(0);
                if (kk == 0 && do_stage && ORD >= 2) stage_part(2, 4);
            }
        }
        store_acc<TM>(p, acc, m0 + wm * 64, n0 + wn * 64, li, lg);
        if (!has_next) return;
        tile += tstride; m0 = m0n; n0 = n0n;
        zero();
    }
}

// =====================================================================================================================
// kernel 1: 8 waves, ping-pong groups, region-wise DMA with counted waits
// =====================================================================================================================
// LDS: buffer b (b = g & 1 for global K-tile g) = [X rows 0..255][W rows 0..255], 128-B rows, swizzled 16-B slots.
// Regions of a buffer and who reads them (wave (wm, wn); group = wm):
//   XT(wm) = X rows wm*128 + [0,64)    read by group wm in phase 0      XB(wm) = X rows wm*128 + [64,128)  read in phase 2
//   W01    = W rows r with (r & 8) == 0 (fragments 0,1 of every wn)     read by both groups in phase 0
//   W23    = W rows r with (r & 8) != 0 (fragments 2,3)                 read by both groups in phase 1
// Interval calendar (an interval = the time between two barriers): group 0 runs mem(p) of K-tile g in interval 8g + 2p and
// comp(p) in 8g + 2p + 1; group 1 one interval later.  A region may be overwritten by the DMA of K-tile g+2 once every reader
// has waited for its reads (start of its comp phase) and passed the barrier that ends that comp phase:
//   group wm, mem(P1,g): XT(wm) of g+2      mem(P2,g): its share of W01 of g+2      mem(P3,g): its share of W23 of g+2
//   mem(P0,g+1): XB(wm) of g+2              -> 2 pieces per wave per phase, issue order = K-tile order per region
// Landing: at the end of every mem phase a wave waits until at most 10 of its pieces are outstanding, i.e. everything it issued
// 5 or more phases ago has landed; the barrier that follows publishes it.  First readers: XT and W01 of g+2 in mem(P0,g+2)
// (issued >= 6 phases earlier), W23 in mem(P1,g+2) (issued at mem(P3,g): 6 phases), XB in mem(P2,g+2) (issued at mem(P0,g+1): 6).
template <int ABL>   // ablation bits: 1 = no DMA in the loop, 2 = no MFMA, 4 = no fragment reads in the loop, 8 = no barriers in the loop,
                     // 16 = DMA sources private to the workgroup and L2-hot (timing probe), 32 = shared sources but always K-tile 0 (L2-hot)
__global__ __launch_bounds__(512) void gemm_pp8(Args p) {
    constexpr int XB_OFF = 0, WB_OFF = BM * ROWB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int nkt = p.K / BK;
    const int first = xcd_remap(blockIdx.x, gridDim.x), tstride = gridDim.x;
    const int my_tiles = first < ntiles ? (ntiles - first + tstride - 1) / tstride : 0;
    const int G = my_tiles * nkt;                       // global K-tiles this workgroup consumes
    if (G == 0) return;
    const char* gX = (const char*)p.X;
    const char* gW = (const char*)p.W;

    // ---- DMA bookkeeping: per region a (tile, kt) cursor that advances by one K-tile per issue -------------------------------
    // per-lane row offsets for the 4 regions of the CURRENT cursor tile; cursors of the four regions may sit in different tiles
    struct Cur { int tile; int kt; uint32_t off[2]; };
    Cur cXT, cXB, cW01, cW23;
    const int prow = lane >> 3, pslot = lane & 7;
    auto x_off = [&](int tile, int half, uint32_t (&off)[2]) {
        int tm, tn;
        tile_of(tile, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = wm * 128 + half * 64 + (wn * 2 + q) * 8 + prow;         // tile row of this lane's piece
            const int key = (row >> 1) & 7;
            if constexpr (ABL & 16) tm = (int)blockIdx.x % (p.M / BM / 2);
            off[q] = (uint32_t)min(tm * BM + row, p.M - 1) * (uint32_t)p.K * 2u + ((pslot ^ key) << 4);
        }
    };
    auto w_off = [&](int tile, int odd, uint32_t (&off)[2]) {
        int tm, tn;
        tile_of(tile, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = 2 * (wave * 2 + q) + odd;                               // 8-row piece of the W tile; even: W01, odd: W23
            const int row = piece * 8 + prow;
            const int rl = row & 63, key = (((rl >> 4) << 1) | ((rl & 3) >> 1)) & 7;
            off[q] = (uint32_t)min(tn * BN + row, p.N - 1) * (uint32_t)p.K * 2u + ((pslot ^ key) << 4);
            if constexpr (ABL & 16)   // private: rows of X's upper half stand in for W
                off[q] = (uint32_t)(((int)blockIdx.x % (p.M / BM / 2) + p.M / BM / 2) * BM + row) * (uint32_t)p.K * 2u + ((pslot ^ key) << 4);
        }
    };
    auto x_dst = [&](int half, int q) { return XB_OFF + (wm * 128 + half * 64 + (wn * 2 + q) * 8) * ROWB; };
    auto w_dst = [&](int odd, int q) { return WB_OFF + (2 * (wave * 2 + q) + odd) * 8 * ROWB; };
    auto advance = [&](Cur& c, int kind) {   // next K-tile of this region's stream
        if (++c.kt == nkt) {
            c.kt = 0;
            c.tile += tstride;
            if (c.tile < ntiles) {
                if (kind == 0) x_off(c.tile, 0, c.off);
                else if (kind == 1) x_off(c.tile, 1, c.off);
                else if (kind == 2) w_off(c.tile, 0, c.off);
                else w_off(c.tile, 1, c.off);
            }
        }
    };
    // issue the next K-tile of one region into the buffer its global index selects; gidx = that global K-tile index
    auto issue_x = [&](Cur& c, int half, int gidx) {
        if (gidx < G) {
            char* sb = smem + (gidx & 1) * STAGE;
            const size_t kb = (ABL & 48) ? 0 : (size_t)c.kt * ROWB;
            if constexpr (!(ABL & 1)) {
                glds16(gX + c.off[0] + kb, sb + x_dst(half, 0));
                glds16(gX + c.off[1] + kb, sb + x_dst(half, 1));
            }
            advance(c, half);
        }
    };
    auto issue_w = [&](Cur& c, int odd, int gidx) {
        if (gidx < G) {
            char* sb = smem + (gidx & 1) * STAGE;
            const size_t kb = (ABL & 48) ? 0 : (size_t)c.kt * ROWB;
            if constexpr (!(ABL & 1)) {
                glds16(((ABL & 16) ? gX : gW) + c.off[0] + kb, sb + w_dst(odd, 0));
                glds16(((ABL & 16) ? gX : gW) + c.off[1] + kb, sb + w_dst(odd, 1));
            }
            advance(c, 2 + odd);
        }
    };
    cXT.tile = cXB.tile = cW01.tile = cW23.tile = first;
    cXT.kt = cXB.kt = cW01.kt = cW23.kt = 0;
    x_off(first, 0, cXT.off); x_off(first, 1, cXB.off); w_off(first, 0, cW01.off); w_off(first, 1, cW23.off);

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    const int keyR = (((li >> 2) << 1) | ((li & 3) >> 1)) & 7, keyC = (li >> 1) & 7;
    const int baseR = WB_OFF + (wn * 64 + (li >> 2) * 16 + (li & 3)) * ROWB;   // + 4 f rows
    const int baseC = XB_OFF + (wm * 128 + li) * ROWB;                          // + 16 f rows
    const int slotR0 = ((lg ^ keyR) << 4), slotR1 = (((4 | lg) ^ keyR) << 4);
    const int slotC0 = ((lg ^ keyC) << 4), slotC1 = (((4 | lg) ^ keyC) << 4);

    f32x4_t acc[8][4];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    zero();
    bf16x8_t fx[4][2], fw[4][2];   // X fragments of the current half (kk = 0,1); W fragments 0..3 (kk = 0,1)

    auto rd_x = [&](const char* sb, int half) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            fx[f][0] = *(const bf16x8_t*)(sb + baseC + (half * 4 + f) * 16 * ROWB + slotC0);
            fx[f][1] = *(const bf16x8_t*)(sb + baseC + (half * 4 + f) * 16 * ROWB + slotC1);
        }
    };
    auto rd_w = [&](const char* sb, int f0) {
#pragma unroll
        for (int f = f0; f < f0 + 2; ++f) {
            fw[f][0] = *(const bf16x8_t*)(sb + baseR + f * 4 * ROWB + slotR0);
            fw[f][1] = *(const bf16x8_t*)(sb + baseR + f * 4 * ROWB + slotR1);
        }
    };
    auto comp = [&](int half, int j0) {   // 16 MFMAs: X fragments of `half` x W fragments j0, j0+1 x kk 0,1
        if constexpr (ABL & 2) {
#pragma unroll
            for (int f = 0; f < 4; ++f) { asm volatile("" ::"v"(fx[f][0]), "v"(fx[f][1])); asm volatile("" ::"v"(fw[f][0]), "v"(fw[f][1])); }
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = j0; j < j0 + 2; ++j)
                    acc[half * 4 + f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][kk], fx[f][kk], acc[half * 4 + f][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // counted wait: valid while every phase issues its 2 pieces; in the last two K-tiles of the stream issues are skipped, so the
    // count no longer bounds the age of what is outstanding -> drain completely there
    auto wait_dma = [&](bool tail) {
        if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    };
    auto wait_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };

    // ---- prologue: K-tiles 0 and 1 completely, in the steady-state issue order per wave, then everything lands ------------------
    issue_x(cXT, 0, 0); issue_w(cW01, 0, 0); issue_w(cW23, 1, 0); issue_x(cXB, 1, 0);
    issue_x(cXT, 0, 1); issue_w(cW01, 0, 1); issue_w(cW23, 1, 1); issue_x(cXB, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (wm == 1) bar();                                  // group 1 runs one interval behind

    int tile = first, m0, n0;
    { int tm, tn; tile_of(tile, tiles_m, tiles_n, tm, tn); m0 = tm * BM; n0 = tn * BN; }
    int kt = 0;
    for (int g = 0; g < G; ++g) {
        const char* sb = smem + (g & 1) * STAGE;
        const bool tail = g + 3 >= G;
        // -------- phase 0: mem = XT, W01 reads + DMA of XB(g+1) [its buffer half was read in phase 2 of g-1] ----------------
        if constexpr (!(ABL & 4)) { rd_w(sb, 0); rd_x(sb, 0); } else if (g == 0) { rd_w(sb, 0); rd_x(sb, 0); }
        if (g >= 1) issue_x(cXB, 1, g + 1);
        wait_dma(tail);
        bar();
        wait_lds();
        comp(0, 0);
        bar();
        // -------- phase 1: mem = W23 reads + DMA of XT(g+2) ----------------------------------------------------------------
        if constexpr (!(ABL & 4)) rd_w(sb, 2); else if (g == 0) rd_w(sb, 2);
        issue_x(cXT, 0, g + 2);
        wait_dma(tail);
        bar();
        wait_lds();
        comp(0, 2);
        bar();
        // -------- phase 2: mem = XB reads + DMA of W01(g+2) ------------------------------------------------------------------
        if constexpr (!(ABL & 4)) rd_x(sb, 1);
        issue_w(cW01, 0, g + 2);
        wait_dma(tail);
        bar();
        wait_lds();
        comp(1, 2);
        bar();
        // -------- phase 3: mem = DMA of W23(g+2) -----------------------------------------------------------------------------
        issue_w(cW23, 1, g + 2);
        wait_dma(tail);
        bar();
        comp(1, 0);
        bar();
        // -------- end of an output tile ---------------------------------------------------------------------------------------
        if (++kt == nkt) {
            kt = 0;
            if (wm == 0) bar();                          // group 0 idles one interval so that both groups store together
            store_acc<8>(p, acc, m0 + wm * 128, n0 + wn * 64, li, lg);
            zero();
            if (wm == 1) bar();
            tile += tstride;
            if (tile < ntiles) { int tm, tn; tile_of(tile, tiles_m, tiles_n, tm, tn); m0 = tm * BM; n0 = tn * BN; }
        }
    }
    if (wm == 0) bar();                                  // match group 1's extra opening barrier
}

// ---- naive reference (checks a sub-block) ---------------------------------------------------------------------------------
__global__ void ref_kernel(Args p, int m_lo, int rows, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = m_lo + blockIdx.y;
    if (n >= p.N || blockIdx.y >= rows) return;
    float s = 0.f;
    for (int k = 0; k < p.K; ++k) {
        const uint32_t a = ((uint32_t)p.X[(size_t)m * p.K + k]) << 16, b = ((uint32_t)p.W[(size_t)n * p.K + k]) << 16;
        s += __uint_as_float(a) * __uint_as_float(b);
    }
    out[(size_t)blockIdx.y * p.N + n] = s;
}
__global__ void fill_kernel(bf16_t* x, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u + seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        // sum of 4 bytes - 510 : roughly normal, full-range mantissas
        const float v = ((float)((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) - 510.f) * (1.0f / 148.f) * scale;
        x[i] = (bf16_t)(__float_as_uint(v) >> 16);
    }
}

template <class F>
static float time_ms(F&& f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

typedef void (*kern_t)(Args);
struct Variant { const char* name; kern_t k; int threads; int smem; };

int main(int argc, char** argv) {
    std::vector<std::array<int, 3>> shapes;
    for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
    if (shapes.empty()) shapes = {{88064, 2048, 1024}, {88064, 1024, 1024}, {88064, 4096, 1024}, {88064, 1024, 4096}, {8192, 8192, 8192}};
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount & ~7;
    Variant vars[] = {
        {"base16", gemm_base16<0>, 1024, 2 * STAGE},
        {"pp8", gemm_pp8<0>, 512, 2 * STAGE},
        {"pp8-noDMA", gemm_pp8<1>, 512, 2 * STAGE},
        {"pp8-noMFMA", gemm_pp8<2>, 512, 2 * STAGE},
        {"pp8-noLDSrd", gemm_pp8<4>, 512, 2 * STAGE},
        {"pp8-onlyMFMA", gemm_pp8<5>, 512, 2 * STAGE},
        {"dma+bar", gemm_pp8<6>, 512, 2 * STAGE},
        {"b16-rd1st", gemm_base16<1>, 1024, 2 * STAGE},
        {"b16-split", gemm_base16<2>, 1024, 2 * STAGE},
        {"b16-split-prio", gemm_base16<3>, 1024, 2 * STAGE},
    };
    for (auto& v : vars) CK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, v.smem));

    // ---- correctness on a small ragged problem (all variants that compute) ---------------------------------------------------
    {
        const int M = 1000, N = 768, K = 512;
        bf16_t *X, *W, *C;
        float* R;
        CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        CK(hipMalloc(&R, (size_t)M * N * 4));
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, X, (size_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, W, (size_t)N * K, 2u, 0.05f);
        Args a{X, W, C, M, N, K};
        hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, a, 0, M, R);
        std::vector<float> ref((size_t)M * N);
        CK(hipMemcpy(ref.data(), R, ref.size() * 4, hipMemcpyDeviceToHost));
        for (int vi = 0; vi < 2; ++vi) {
            CK(hipMemset(C, 0xff, (size_t)M * N * 2));
            const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
            hipLaunchKernelGGL(vars[vi].k, dim3(tiles < ncu ? tiles : ncu), dim3(vars[vi].threads), vars[vi].smem, 0, a);
            CK(hipDeviceSynchronize());
            std::vector<bf16_t> out((size_t)M * N);
            CK(hipMemcpy(out.data(), C, out.size() * 2, hipMemcpyDeviceToHost));
            double maxerr = 0;
            size_t bad = 0;
            for (size_t i = 0; i < out.size(); ++i) {
                uint32_t u = ((uint32_t)out[i]) << 16;
                float f;
                memcpy(&f, &u, 4);
                const double e = fabs((double)f - ref[i]), tol = 0.02 + 0.01 * fabs(ref[i]);
                if (!(e <= tol)) ++bad;
                if (e > maxerr) maxerr = e;
            }
            printf("check %-8s M=%d N=%d K=%d: max |err| %.4f, %zu bad of %zu\n", vars[vi].name, M, N, K, maxerr, bad, out.size());
        }
        // two tiles per workgroup (persistent walk) with a tiny grid
        {
            CK(hipMemset(C, 0xff, (size_t)M * N * 2));
            hipLaunchKernelGGL(vars[1].k, dim3(8), dim3(vars[1].threads), vars[1].smem, 0, a);
            CK(hipDeviceSynchronize());
            std::vector<bf16_t> out((size_t)M * N);
            CK(hipMemcpy(out.data(), C, out.size() * 2, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < out.size(); ++i) {
                uint32_t u = ((uint32_t)out[i]) << 16;
                float f;
                memcpy(&f, &u, 4);
                if (!(fabs((double)f - ref[i]) <= 0.02 + 0.01 * fabs(ref[i]))) ++bad;
            }
            printf("check pp8 grid=8 (persistent walk): %zu bad\n", bad);
        }
        hipFree(X); hipFree(W); hipFree(C); hipFree(R);
    }

    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        bf16_t *X, *W, *C;
        CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, X, (size_t)M * K, 11u, 1.0f);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, W, (size_t)N * K, 12u, 0.02f);
        Args a{X, W, C, M, N, K};
        const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
        const int grid = tiles < ncu ? tiles : ncu;
        const double fl = 2.0 * M * N * K;
        printf("M=%d N=%d K=%d (%d tiles):", M, N, K, tiles);
        for (int round = 0; round < 2; ++round)
            for (auto& v : vars) {
                const float ms = time_ms([&] { hipLaunchKernelGGL(v.k, dim3(grid), dim3(v.threads), v.smem, 0, a); }, 10);
                if (round == 1) printf("  %s %.3f ms %.0f TF |", v.name, ms, fl / ms / 1e9);
            }
        printf("\n");
        fflush(stdout);
        hipFree(X); hipFree(W); hipFree(C);
    }
    return 0;
}
