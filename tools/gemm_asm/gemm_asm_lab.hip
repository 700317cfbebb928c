// Hand-scheduled bf16 GEMM main loop for gfx950 (development probe, not part of the library) — VERDICT r03 "next round" item 1, stage 1:
// the K loop of the persistent 256x256 kernel as ONE inline-asm body with AGPR-resident accumulators at one wave per SIMD
// (4 waves, 128x128 outputs per wave = half the LDS fragment bytes per flop of the library's 16-wave kernel), keeping the library's
// XCD-aware strip order, LDS-DMA operand staging with the swizzle on the source address, and the persistent tile walk.  LOOP ONLY:
// there is no fused epilogue (store = 0: nothing is written; store = 1: raw fp32 accumulator dump for the correctness check).
//
//   python tools/gemm_asm/gen_loop.py > tools/gemm_asm/loop_body.inc
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_asm/gemm_asm_lab.hip -o /tmp/gemm_asm_lab
//   /tmp/gemm_asm_lab [--secs S] [M N K]...        (default shapes: the bench's qk / proj / fc1 / fc2 at M = 294 464)
//
// The loop, its register map and its hazards are documented in gen_loop.py.
#include <hip/hip_runtime.h>

#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "loop_body.inc"

typedef unsigned short bf16_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const bf16_t* X; const bf16_t* W;   // [Mpad,K], [N,K]
    const uint2* table;                 // [grid][tiles_per_wg + 1] {x byte offset, w byte offset}
    float* out;                         // check-mode dump
    const int* ntiles_wg;               // [grid]
    int K, tab_stride, store;
    uint32_t recX, recW;
};

constexpr int NWN = FP_ASM_NWN, NWAVE = 2 * NWN, TRF = 16 / NWN, NFRAG = 8 * TRF;   // waves along N, waves, W fragments per wave, accumulator fragments
constexpr size_t DUMP_WAVE = (size_t)NFRAG * 1024, DUMP_TILE = DUMP_WAVE * NWAVE;

__global__ __launch_bounds__(NWAVE * 64) void gemm_asm_kernel(Args p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // 2 stages x 64 KiB at LDS address 0 (the asm uses absolute addresses)
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = wave / NWN, wn = wave % NWN;
    const uint32_t li = lane & 15, lg = lane >> 4;
    const uint32_t K2 = (uint32_t)p.K * 2u;
    const uint32_t vdma = (uint32_t)(lane >> 3) * K2 + ((((uint32_t)lane & 7u) ^ ((uint32_t)lane >> 3)) << 4);
    const uint32_t sw = (lg ^ (li & 7u)) << 4;
    const uint32_t vax = (wm * 128u + li) * 128u + sw;
    const uint32_t vaw = 32768u + (wn * (256u / NWN) + li) * 128u + sw;
    const uint32_t vout = (uint32_t)lane * 16u + wave * (uint32_t)DUMP_WAVE;
    const uint64_t xa = (uint64_t)p.X, wa = (uint64_t)p.W;
    const uint64_t ta = (uint64_t)(p.table + (size_t)blockIdx.x * p.tab_stride);
    const uint64_t oa = (uint64_t)p.out + (size_t)blockIdx.x * (size_t)(p.tab_stride - 1) * DUMP_TILE;
    const uint32_t nt = (uint32_t)p.ntiles_wg[blockIdx.x];
    if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
    if (nt == 0) return;
    asm volatile(FP_ASM_LOOP_TEXT
                 :
                 : "s"((uint32_t)xa), "s"((uint32_t)(xa >> 32) & 0xffffu), "s"((uint32_t)wa), "s"((uint32_t)(wa >> 32) & 0xffffu),
                   "s"((uint32_t)ta), "s"((uint32_t)(ta >> 32)), "s"((uint32_t)oa), "s"((uint32_t)(oa >> 32)),
                   "s"(K2), "s"((uint32_t)p.K / 64u), "s"(nt), "s"(p.recX), "s"(p.recW), "s"((uint32_t)p.store), "s"(wave),
                   "v"(vdma), "v"(vax), "v"(vaw), "v"(vout)
                 : FP_ASM_LOOP_CLOBBERS);
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
// the library's tile order (freepose_amd/csrc/gemm_bf16.h fp_gemm_tile): XCD-contiguous ids, column strips of 4 n-tiles swept m-major
static void tile_of(int block, int nblocks, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int SW = 4;
    const int q = nblocks >> 3, r = nblocks & 7, xcd = block & 7, pos = block >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    const int full = tiles_n / SW, tail = tiles_n - full * SW, in_full = full * tiles_m * SW;
    if (id < in_full) { const int strip = id / (tiles_m * SW), rem = id - strip * (tiles_m * SW); tm = rem / SW; tn = strip * SW + (rem - tm * SW); }
    else { const int rem = id - in_full; tm = rem / tail; tn = full * SW + (rem - tm * tail); }
}

__global__ void ref_kernel(const bf16_t* X, const bf16_t* W, int M, int N, int K, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        s += __uint_as_float(((uint32_t)X[(size_t)m * K + k]) << 16) * __uint_as_float(((uint32_t)W[(size_t)n * K + k]) << 16);
    out[(size_t)m * N + n] = s;
}
__global__ void fill_kernel(bf16_t* x, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u + seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        const float v = ((float)((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) - 510.f) * (1.0f / 148.f) * scale;
        x[i] = (bf16_t)(__float_as_uint(v) >> 16);   // roughly normal, full-range mantissas (the fill of tools/gemm_lab.hip)
    }
}

struct Problem {
    int M, N, K, Mpad, tiles_m, tiles_n, ntiles, grid, tps;
    bf16_t *X = nullptr, *W = nullptr;
    uint2* table = nullptr;
    int* ntw = nullptr;
    std::vector<std::array<int, 2>> tile_mn;   // [grid * tps] (m0, n0) for the check
};

static Problem make_problem(int M, int N, int K, int ncu, int force_grid, bool zero) {
    Problem p;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + 255) / 256; p.tiles_n = N / 256; p.Mpad = p.tiles_m * 256;
    p.ntiles = p.tiles_m * p.tiles_n;
    p.grid = force_grid > 0 ? force_grid : (p.ntiles < ncu ? p.ntiles : ncu);
    if (p.grid > p.ntiles) p.grid = p.ntiles;
    p.tps = (p.ntiles + p.grid - 1) / p.grid;
    CK(hipMalloc(&p.X, (size_t)p.Mpad * K * 2)); CK(hipMalloc(&p.W, (size_t)N * K * 2));
    if (zero) { CK(hipMemset(p.X, 0, (size_t)p.Mpad * K * 2)); CK(hipMemset(p.W, 0, (size_t)N * K * 2)); }
    else {
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p.X, (size_t)p.Mpad * K, 11u, 1.0f);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p.W, (size_t)N * K, 12u, 0.02f);
    }
    std::vector<uint2> tab((size_t)p.grid * (p.tps + 1));
    std::vector<int> ntw(p.grid);
    p.tile_mn.assign((size_t)p.grid * p.tps, {-1, -1});
    for (int b = 0; b < p.grid; ++b) {
        int cnt = 0;
        uint2 last{0, 0};
        for (int t = b; t < p.ntiles; t += p.grid, ++cnt) {
            int tm, tn;
            tile_of(t, p.ntiles, p.tiles_m, p.tiles_n, tm, tn);
            last = make_uint2((uint32_t)((size_t)tm * 256 * K * 2), (uint32_t)((size_t)tn * 256 * K * 2));
            tab[(size_t)b * (p.tps + 1) + cnt] = last;
            p.tile_mn[(size_t)b * p.tps + cnt] = {tm * 256, tn * 256};
        }
        ntw[b] = cnt;
        for (int c = cnt; c <= p.tps; ++c) tab[(size_t)b * (p.tps + 1) + c] = last;   // the loop requests one entry past its last tile
    }
    CK(hipMalloc(&p.table, tab.size() * sizeof(uint2))); CK(hipMemcpy(p.table, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice));
    CK(hipMalloc(&p.ntw, ntw.size() * 4)); CK(hipMemcpy(p.ntw, ntw.data(), ntw.size() * 4, hipMemcpyHostToDevice));
    return p;
}
static void free_problem(Problem& p) { hipFree(p.X); hipFree(p.W); hipFree(p.table); hipFree(p.ntw); }

static void launch(const Problem& p, float* out, int store) {
    Args a{p.X, p.W, p.table, out, p.ntw, p.K, p.tps + 1, store, 0xffffffffu, 0xffffffffu};
    hipLaunchKernelGGL(gemm_asm_kernel, dim3(p.grid), dim3(NWAVE * 64), 131072, 0, a);
}

static bool check(int M, int N, int K, int ncu, int force_grid) {
    Problem p = make_problem(M, N, K, ncu, force_grid, false);
    float *out, *ref;
    const size_t out_bytes = (size_t)p.grid * p.tps * DUMP_TILE;
    CK(hipMalloc(&out, out_bytes)); CK(hipMemset(out, 0xff, out_bytes));
    CK(hipMalloc(&ref, (size_t)p.Mpad * N * 4));
    hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, p.Mpad), dim3(256), 0, 0, p.X, p.W, p.Mpad, N, K, ref);
    launch(p, out, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> ho(out_bytes / 4), hr((size_t)p.Mpad * N);
    CK(hipMemcpy(ho.data(), out, out_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, seen = 0;
    double maxerr = 0;
    for (int b = 0; b < p.grid; ++b)
        for (int sq = 0; sq < p.tps; ++sq) {
            const auto mn = p.tile_mn[(size_t)b * p.tps + sq];
            if (mn[0] < 0) continue;
            for (int w = 0; w < NWAVE; ++w)
                for (int f = 0; f < NFRAG; ++f)
                    for (int l = 0; l < 64; ++l)
                        for (int r = 0; r < 4; ++r) {
                            const int i = f / TRF, j = f % TRF;
                            const int m = mn[0] + (w / NWN) * 128 + 16 * i + (l >> 4) * 4 + r, n = mn[1] + (w % NWN) * (256 / NWN) + 16 * j + (l & 15);
                            const float got = ho[((((size_t)b * p.tps + sq) * NWAVE + w) * NFRAG + f) * 256 + l * 4 + r];
                            const float want = hr[(size_t)m * N + n];
                            const double e = fabs((double)got - want);
                            if (!(e <= 1e-3 + 1e-3 * fabs(want))) ++bad;
                            if (e > maxerr) maxerr = e;
                            ++seen;
                        }
        }
    printf("check M=%d N=%d K=%d grid=%d (%d tiles, %d per workgroup): %zu values, max |err| %.3g, %zu bad\n", M, N, K, p.grid, p.ntiles, p.tps, seen, maxerr, bad);
    hipFree(out); hipFree(ref);
    free_problem(p);
    return bad == 0 && seen == (size_t)p.Mpad * N;
}

int main(int argc, char** argv) {
    double secs = 0;
    int zero = 0, nocheck = 0;
    std::vector<std::array<int, 3>> shapes;
    for (int i = 1; i < argc;) {
        if (!strcmp(argv[i], "--secs") && i + 1 < argc) { secs = atof(argv[i + 1]); i += 2; }
        else if (!strcmp(argv[i], "--zero")) { zero = 1; i += 1; }
        else if (!strcmp(argv[i], "--nocheck")) { nocheck = 1; i += 1; }   // ablated loops (gen_loop.py --ablate) compute garbage
        else if (i + 2 < argc) { shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])}); i += 3; }
        else break;
    }
    const int Mb = 214 * 1376;
    if (shapes.empty()) shapes = {{Mb, 2048, 1024}, {Mb, 1024, 1024}, {Mb, 4096, 1024}, {Mb, 1024, 4096}};
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount & ~7;
    CK(hipFuncSetAttribute((const void*)gemm_asm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    bool ok = true;
    if (!nocheck) {
    ok &= check(512, 512, 1024, ncu, 0);      // one tile per workgroup
    ok &= check(1280, 768, 1024, ncu, 8);     // persistent walk: 15 tiles on 8 workgroups, ragged tail of the walk
    ok &= check(700, 1024, 4096, ncu, 8);     // ragged M (zero-padded rows), long K
    }
    if (!ok) { printf("CHECK FAILED\n"); return 1; }
    for (auto& s : shapes) {
        Problem p = make_problem(s[0], s[1], s[2], ncu, 0, zero != 0);
        const double fl = 2.0 * s[0] * (double)s[1] * s[2];
        for (int i = 0; i < 3; ++i) launch(p, nullptr, 0);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        int reps = 0;
        float ms = 0;
        const auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(e0, 0);
        do {
            for (int i = 0; i < 10; ++i) launch(p, nullptr, 0);
            reps += 10;
            CK(hipStreamSynchronize(0));
        } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("asm-loop %s M=%d N=%d K=%d: %.4f ms  %.0f TF  (%d launches)\n", zero ? "zero" : "random", s[0], s[1], s[2], ms / reps, fl / (ms / reps) / 1e9, reps);
        fflush(stdout);
        free_problem(p);
    }
    return 0;
}
