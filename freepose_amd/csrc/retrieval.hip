// Retrieval kernels (K10, K12, K13 of SURVEY.md §2.3), gfx950 only.  All HBM-bound byte/row streaming:
// coalesced 16-byte loads, one wave per row, wave-shuffle reductions, no MFMA.
//
// Canonical arithmetic ("dot64"), shared with oracle/fp_oracle.c so indices AND scores are bit-exact:
//   lane l (0..63) owns elements (c*64 + l)*8 + e  (c = 0.., e = 0..7) of the D-vector,
//   p_l = fmaf chain over (c, e) ascending; s = xor-butterfly sum over offsets 32,16,8,4,2,1;
//   score = bf16_rne(s)  (the reference rounds bank@feature to bf16 before .float()/topk:
//   scripts/extract_proposals_ground.py:137-140).
// Ordering for top-k: score descending, then index ascending (torch.topk's tie order is unspecified;
// this is the documented canonical rule).
#include "internal.h"

#include <stdlib.h>

namespace {

__device__ __forceinline__ uint32_t score_key16(float s_bf16_rounded) {
    const uint32_t b = __float_as_uint(s_bf16_rounded) >> 16;
    return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
}
__device__ __forceinline__ float key16_to_float(uint32_t k) {
    const uint32_t b = (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu);
    return __uint_as_float(b << 16);
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = f2bf(x[i]);
}

// ---------------------------------------------------------------------------------------------
// bank scan: keys[q, r] = sortable16( bf16( bank[r] . query[q] ) )
template <int NCH, int QT, bool FULL>   // FULL: D == NCH * 512, every lane's 8-element slot is in range (no per-load test)
__global__ __launch_bounds__(256) void bank_scan_kernel(const bf16_t* __restrict__ bank,
                                                        const bf16_t* __restrict__ queries,
                                                        uint16_t* __restrict__ keys, int N, int D, int q_begin,
                                                        int Q, int ldk) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    // Balanced partition: wave w owns a contiguous run of floor(N/nwave) (+1 for the first N%nwave waves) rows, streamed in
    // chunks of RU rows (RU x 2 KB in flight) with the next chunk requested before the current one is reduced.  The host
    // picks the grid (2..8 workgroups per CU) that leaves the smallest remainder, e.g. 3 per CU for N = 46 037 (0.1 %).
    constexpr int RU = 4;
    const int base_rows = N / nwave, rem = N - base_rows * nwave;
    const int r_begin = wave * base_rows + min(wave, rem);
    const int r_end = r_begin + base_rows + (wave < rem ? 1 : 0);
    uint4 bufA[RU][NCH], bufB[RU][NCH];
    float qv[QT][NCH][8];                 // the queries, unpacked (filled in after the first row requests are out)
    auto fetch = [&](uint4 (&dst)[RU][NCH], int r0) {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (r0 + u >= r_end) break;      // wave-uniform: rows past the run are neither fetched nor stored
            const int r = r0 + u;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int base = (c * 64 + lane) * 8;
                dst[u][c] = (FULL || base < D) ? *(const uint4*)(bank + (size_t)r * D + base) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto reduce_chunk = [&](const uint4 (&raw)[RU][NCH], int r0) {
        float acc[RU][QT];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[u][q] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t w[4] = {raw[u][c].x, raw[u][c].y, raw[u][c].z, raw[u][c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = lo_bf(w[e]), x1 = hi_bf(w[e]);
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        acc[u][q] = __fmaf_rn(x0, qv[q][c][2 * e], acc[u][q]);
                        acc[u][q] = __fmaf_rn(x1, qv[q][c][2 * e + 1], acc[u][q]);
                    }
                }
            }
        }
        // the chunk's four rows are reduced together; 16-lane row u of the wave ends up with row (r0 + u)'s dot product
        static_assert(RU == 4, "wave_sum4 reduces four rows");
        const int urow = lane >> 4;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const float sdot = wave_sum4(acc[0][q], acc[1][q], acc[2][q], acc[3][q]);
            if ((lane & 15) == 0 && r0 + urow < r_end && q_begin + q < Q)
                keys[(size_t)(q_begin + q) * ldk + r0 + urow] = (uint16_t)score_key16(rbf(sdot));
        }
    };
    // two chunk buffers used alternately (no register copy at the hand-over): B is requested before A is reduced, etc.
    // The first two chunks are requested BEFORE the queries: a wave owns only ~9 rows of the 46 037-row bank (5 120 waves), so the
    // kernel is a start-up transient, and queries-then-rows cost two dependent memory round trips where one suffices (r03: the
    // cold-bank pass went 19.3 -> see profiles/r03_ab.md).
    if (r_begin < r_end) fetch(bufA, r_begin);
    if (r_begin + RU < r_end) fetch(bufB, r_begin + RU);
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int base = (c * 64 + lane) * 8;
            const bool ok = (q_begin + q < Q) && (FULL || base < D);
            const uint4 qw = ok ? *(const uint4*)(queries + (size_t)(q_begin + q) * D + base) : make_uint4(0, 0, 0, 0);
            const uint32_t w[4] = {qw.x, qw.y, qw.z, qw.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { qv[q][c][2 * e] = lo_bf(w[e]); qv[q][c][2 * e + 1] = hi_bf(w[e]); }
        }
    for (int r0 = r_begin; r0 < r_end; r0 += 2 * RU) {
        if (r0 != r_begin && r0 + RU < r_end) fetch(bufB, r0 + RU);
        reduce_chunk(bufA, r0);
        if (r0 + RU < r_end) {
            if (r0 + 2 * RU < r_end) fetch(bufA, r0 + 2 * RU);
            reduce_chunk(bufB, r0 + RU);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// exact top-k of 16-bit keys with (key desc, index asc) order: two-level radix select + ordered
// compaction + bitonic sort of the k survivors.  One workgroup per query.
constexpr int SEL_T = 1024;
constexpr int KMAX = 1024;

__device__ __forceinline__ int block_excl_scan(int v, int* sh, int& total) {
    // sh: SEL_T ints.  simple Hillis-Steele over waves: wave scan + wave totals
    // all sums are carried in unsigned arithmetic: callers pack two 16-bit counts into one word, and the upper one reaches
    // 2^15 and beyond on degenerate rows (46 037 equal keys), which would overflow a signed add
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned x = (unsigned)v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned y = (unsigned)__shfl_up((int)x, off, 64);
        if (lane >= off) x += y;
    }
    __syncthreads();
    if (lane == 63) sh[w] = (int)x;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int i = 0; i < SEL_T / 64; ++i) {
        const unsigned t = (unsigned)sh[i];
        if (i < w) base += t;
        tot += t;
    }
    total = (int)tot;
    __syncthreads();
    return (int)(base + x - (unsigned)v);
}

// LDSKEYS: the query's whole key row (N x 2 B; 92 KB for the 46 037-row bank) is staged into LDS once with coalesced loads
// and every later pass (two histograms, the ordered compaction's count and write sweeps) reads LDS — the global-read form
// spent its 54 us in four latency-bound sweeps, two of them with a 90-byte stride between neighbouring threads.
template <bool LDSKEYS>
__global__ __launch_bounds__(SEL_T) void topk_select_kernel(const uint16_t* __restrict__ keys, int ldk, int N, int k,
                                                            int idx_offset, float* __restrict__ out_scores,
                                                            int* __restrict__ out_idx) {
    extern __shared__ uint16_t lkeys[];   // [ldk] when LDSKEYS (rows are padded to 8 keys: 16-byte staging loads)
    __shared__ int hist[256];
    __shared__ int sh[SEL_T / 64 + 2];
    __shared__ int s_hi, s_T, s_gt;
    __shared__ unsigned long long cand[KMAX];
    const int q = blockIdx.x, tid = threadIdx.x;
    const uint16_t* kq = keys + (size_t)q * ldk;
    if constexpr (LDSKEYS) {
        const uint4* src = (const uint4*)kq;
        uint4* dst = (uint4*)lkeys;
        for (int i = tid; i < ldk / 8; i += SEL_T) dst[i] = src[i];   // pad keys (>= N) are never read below
        __syncthreads();
        kq = lkeys;
    }
    const int per = (N + SEL_T - 1) / SEL_T;
    const int lo = tid * per, hi = min(lo + per, N);

    // pass 1: histogram of the high byte
    for (int i = tid; i < 256; i += SEL_T) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += SEL_T) atomicAdd(&hist[kq[i] >> 8], 1);
    __syncthreads();
    if (tid == 0) {
        int c = 0, b = 255;
        for (; b >= 0; --b) { if (c + hist[b] >= k) break; c += hist[b]; }
        s_hi = b; s_gt = c;  // c keys strictly above bin b
    }
    __syncthreads();
    const int hb = s_hi;
    __syncthreads();
    // pass 2: histogram of the low byte inside bin hb
    for (int i = tid; i < 256; i += SEL_T) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += SEL_T) { const int key = kq[i]; if ((key >> 8) == hb) atomicAdd(&hist[key & 255], 1); }
    __syncthreads();
    if (tid == 0) {
        int c = s_gt, b = 255;
        for (; b >= 0; --b) { if (c + hist[b] >= k) break; c += hist[b]; }
        s_T = (hb << 8) | b; s_gt = c;  // c keys strictly greater than T
    }
    __syncthreads();
    const int T = s_T, ngt = s_gt, need_eq = k - ngt;
    // ordered compaction (index order): first all > T, then the first need_eq == T
    int cg = 0, ce = 0;
    for (int i = lo; i < hi; ++i) { const int key = kq[i]; cg += key > T; ce += key == T; }
    int tot;
    int og = block_excl_scan(cg, sh, tot);
    int oe = block_excl_scan(ce, sh, tot);
    for (int i = lo; i < hi; ++i) {
        const int key = kq[i];
        if (key > T) { cand[og++] = ((unsigned long long)key << 32) | (unsigned)(0x7fffffff - i); }
        else if (key == T) { if (oe < need_eq) cand[ngt + oe] = ((unsigned long long)key << 32) | (unsigned)(0x7fffffff - i); ++oe; }
    }
    // pad to a power of two and bitonic-sort descending on (key, -index)
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    for (int i = k + tid; i < n2; i += SEL_T) cand[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n2; i += SEL_T) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = ((i & size) == 0);
                    const unsigned long long a = cand[i], b = cand[j];
                    if ((a < b) == desc) { cand[i] = b; cand[j] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += SEL_T) {
        const unsigned long long c = cand[i];
        out_scores[(size_t)q * k + i] = key16_to_float((uint32_t)(c >> 32));
        out_idx[(size_t)q * k + i] = idx_offset + (0x7fffffff - (int)(unsigned)(c & 0xffffffffu));
    }
}

// ---- register-resident select (N <= 48 * SEL_T keys per query) ---------------------------------------------------------------
// The two-level histogram above serialises on LDS atomics when the keys are skewed — and bf16 cosine scores are: with an
// anisotropic bank nearly every key shares one high byte, so each ds_add hits one address 64 ways (43 us for 46 037 keys).
// Here no atomics are used at all: a thread keeps its 2*MAXW consecutive keys packed in MAXW registers and the exact k-th
// largest key T is found by COUNTING,
//   1. per-thread maximum; L = the k-th largest of the SEL_T maxima (bit-wise binary search over 1024 values: one compare per
//      thread and iteration).  At least k keys are >= L, so T >= L; G = the block maximum bounds T from above;
//   2. binary search for T in [L, G] with exact block-wide counts of keys >= mid (packed 16-bit saturating subtract / min /
//      add: 3 VALU per two keys); [L, G] spans a handful of bf16 values, so this takes ~4-6 iterations;
//   3. ordered compaction (all keys > T, then the first k - #greater keys == T in index order) and the bitonic sort of the k
//      survivors, as before.  Same total order (key desc, index asc) -> bit-identical results.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int wave_sum_i(int v) {
    v += __float_as_int(lane_xor<32>(__int_as_float(v)));
    v += __float_as_int(lane_xor<16>(__int_as_float(v)));
    v += __float_as_int(lane_xor<8>(__int_as_float(v)));
    v += __float_as_int(lane_xor<4>(__int_as_float(v)));
    v += __float_as_int(lane_xor<2>(__int_as_float(v)));
    v += __float_as_int(lane_xor<1>(__int_as_float(v)));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, __float_as_int(lane_xor<32>(__int_as_float(v))));
    v = max(v, __float_as_int(lane_xor<16>(__int_as_float(v))));
    v = max(v, __float_as_int(lane_xor<8>(__int_as_float(v))));
    v = max(v, __float_as_int(lane_xor<4>(__int_as_float(v))));
    v = max(v, __float_as_int(lane_xor<2>(__int_as_float(v))));
    v = max(v, __float_as_int(lane_xor<1>(__int_as_float(v))));
    return v;
}
constexpr int SEL_W = 24;                 // packed words per thread: thread t owns keys [48 t, 48 t + 48)
constexpr int SEL_CAP = SEL_T * 2 * SEL_W;   // 49 152 keys

__global__ __launch_bounds__(SEL_T) void topk_select_reg_kernel(const uint16_t* __restrict__ keys, int ldk, int N, int k,
                                                                int idx_offset, float* __restrict__ out_scores,
                                                                int* __restrict__ out_idx) {
    extern __shared__ uint16_t lkeys[];   // [SEL_CAP] staged key row, zero beyond N
    __shared__ int slots[8][SEL_T / 64];  // per-iteration wave partials (one barrier per block-wide count)
    __shared__ unsigned short smax[SEL_T];
    __shared__ int sh[SEL_T / 64 + 2];
    __shared__ unsigned long long cand[KMAX];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint16_t* kq = keys + (size_t)q * ldk;
    {
        // all six 16-byte loads of a thread are issued before the first is stored: a load -> store loop pays the full memory
        // latency per iteration (the key row was just written by the scan kernel: it comes from HBM / MALL, not from this XCD's L2)
        const uint4* src = (const uint4*)kq;
        uint4* dst = (uint4*)lkeys;
        const int n16 = N >> 3;                      // whole 16-byte groups inside the row
        constexpr int NLD = SEL_CAP / 8 / SEL_T;     // 6
        uint4 v[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * SEL_T;
            v[j] = i < n16 ? src[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) dst[tid + j * SEL_T] = v[j];
        __syncthreads();
        if (tid < 8) { const int i = (n16 << 3) + tid; if (i < N) lkeys[i] = kq[i]; }   // the ragged tail of the row
    }
    __syncthreads();
    uint32_t w[SEL_W];
    {
        const uint4* lw = (const uint4*)lkeys + (size_t)tid * (SEL_W / 4);
#pragma unroll
        for (int j = 0; j < SEL_W / 4; ++j) {
            const uint4 v = lw[j];
            w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
        }
    }
    const int lo = tid * 2 * SEL_W, hi = min(lo + 2 * SEL_W, N);

    // ---- 1. maxima: every wave finds L (k-th largest thread maximum) and G (largest) redundantly from the LDS copy ----------
    uint32_t m2 = 0u;
#pragma unroll
    for (int j = 0; j < SEL_W; ++j) m2 = pk_max_u16(m2, w[j]);
    smax[tid] = (unsigned short)max(m2 & 0xffffu, m2 >> 16);
    __syncthreads();
    uint32_t mw[SEL_T / 128];                          // this lane's 16 maxima, packed
    {
        const uint4* sm = (const uint4*)smax + lane * 2;
        const uint4 a = sm[0], b = sm[1];
        mw[0] = a.x; mw[1] = a.y; mw[2] = a.z; mw[3] = a.w; mw[4] = b.x; mw[5] = b.y; mw[6] = b.z; mw[7] = b.w;
    }
    const uint32_t ones = 0x00010001u;
    auto pk_count_ge = [&](const uint32_t* v, int n, int c) -> int {   // halves of v[0..n) that are >= c (c >= 1)
        const uint32_t cc = (uint32_t)(c - 1) * 0x00010001u;
        uint32_t acc = 0u;
#pragma unroll
        for (int j = 0; j < n; ++j) acc = pk_add_u16(acc, pk_min_u16(pk_sub_sat_u16(v[j], cc), ones));
        return (int)(acc & 0xffffu) + (int)(acc >> 16);
    };
    uint32_t g2 = 0u;
#pragma unroll
    for (int j = 0; j < SEL_T / 128; ++j) g2 = pk_max_u16(g2, mw[j]);
    const int G = wave_max_i((int)max(g2 & 0xffffu, g2 >> 16));
    int L = 0;
    for (int bit = 15; bit >= 0; --bit) {
        const int c = L | (1 << bit);
        if (c > G) continue;
        if (wave_sum_i(pk_count_ge(mw, SEL_T / 128, c)) >= k) L = c;
    }
    // ---- 1b. fast path: every key >= L is a candidate (at least k of them; typically k .. 2k).  If they fit the candidate buffer
    // the answer is a rank sort of their (key, index) composites — distinct 64-bit values whose descending order IS the canonical
    // (key desc, index asc) order — and no threshold search is needed.  Tie-heavy rows (more than KMAX keys >= L) take the
    // counting search below.
    {
        unsigned long long gem = 0ull;               // bit b: this thread's key lo + b is >= max(L, 1) (pads are 0)
        const uint32_t cc = (uint32_t)(max(L, 1) - 1) * 0x00010001u;
#pragma unroll
        for (int j = 0; j < SEL_W; ++j) {
            const uint32_t d = pk_min_u16(pk_sub_sat_u16(w[j], cc), ones);
            gem |= (unsigned long long)((d & 1u) | ((d >> 15) & 2u)) << (2 * j);
        }
        int M;
        int off = block_excl_scan(__popcll(gem), sh, M);
        if (M <= KMAX) {
            while (gem) {
                const int b = __ffsll((long long)gem) - 1;
                gem &= gem - 1;
                const int i = lo + b;
                cand[off++] = ((unsigned long long)lkeys[i] << 32) | (unsigned)(0x7fffffff - i);
            }
            __syncthreads();
            if (tid < M) {
                const unsigned long long mine = cand[tid];
                int rank = 0;
                for (int j = 0; j < M; ++j) rank += cand[j] > mine;
                if (rank < k) {
                    out_scores[(size_t)q * k + rank] = key16_to_float((uint32_t)(mine >> 32));
                    out_idx[(size_t)q * k + rank] = idx_offset + (0x7fffffff - (int)(unsigned)(mine & 0xffffffffu));
                }
            }
            return;
        }
    }
    // ---- 2. exact k-th largest key T in [L, G]: block-wide counts, one barrier each ----------------------------------------
    int it = 0;
    auto block_count_ge = [&](int c) -> int {
        const int part = wave_sum_i(pk_count_ge(w, SEL_W, c));
        int* sl = slots[it & 7];
        ++it;
        if (lane == 0) sl[wv] = part;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int i = 0; i < SEL_T / 64; ++i) t += sl[i];
        return t;
    };
    int tlo = L, thi = G;                            // invariant: #keys >= tlo is >= k (L = 0 counts every real key: k <= N)
    while (tlo < thi) {
        const int mid = (tlo + thi + 1) >> 1;
        if (block_count_ge(mid) >= k) tlo = mid; else thi = mid - 1;
    }
    const int T = tlo;
    const int ngt = T >= 65535 ? 0 : block_count_ge(T + 1);
    const int need_eq = k - ngt;
    // ---- 3. ordered compaction (index order): first all > T, then the first need_eq == T -----------------------------------
    int cg = 0, ce = 0;
#pragma unroll
    for (int j = 0; j < SEL_W; ++j) {
        const int i0 = lo + 2 * j;
        const int k0 = (int)(w[j] & 0xffffu), k1 = (int)(w[j] >> 16);
        if (i0 < hi) { cg += k0 > T; ce += k0 == T; }
        if (i0 + 1 < hi) { cg += k1 > T; ce += k1 == T; }
    }
    int tot;
    // both counts stay below 65 536 (a row holds <= 49 152 keys): one unsigned scan carries the two
    const unsigned both = (unsigned)block_excl_scan((int)((unsigned)cg | ((unsigned)ce << 16)), sh, tot);
    int og = (int)(both & 0xffffu), oe = (int)(both >> 16);
#pragma unroll
    for (int j = 0; j < SEL_W; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = lo + 2 * j + h;
            if (i >= hi) continue;
            const int key = h ? (int)(w[j] >> 16) : (int)(w[j] & 0xffffu);
            if (key > T) cand[og++] = ((unsigned long long)key << 32) | (unsigned)(0x7fffffff - i);
            else if (key == T) { if (oe < need_eq) cand[ngt + oe] = ((unsigned long long)key << 32) | (unsigned)(0x7fffffff - i); ++oe; }
        }
    }
    __syncthreads();
    if (k <= 256) {
        // rank sort: candidates are distinct 64-bit values; a candidate's rank is the number of larger ones (broadcast LDS reads)
        if (tid < k) {
            const unsigned long long mine = cand[tid];
            int rank = 0;
            for (int j = 0; j < k; ++j) rank += cand[j] > mine;
            out_scores[(size_t)q * k + rank] = key16_to_float((uint32_t)(mine >> 32));
            out_idx[(size_t)q * k + rank] = idx_offset + (0x7fffffff - (int)(unsigned)(mine & 0xffffffffu));
        }
        return;
    }
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    for (int i = k + tid; i < n2; i += SEL_T) cand[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n2; i += SEL_T) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = ((i & size) == 0);
                    const unsigned long long a = cand[i], b = cand[j];
                    if ((a < b) == desc) { cand[i] = b; cand[j] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += SEL_T) {
        const unsigned long long c = cand[i];
        out_scores[(size_t)q * k + i] = key16_to_float((uint32_t)(c >> 32));
        out_idx[(size_t)q * k + i] = idx_offset + (0x7fffffff - (int)(unsigned)(c & 0xffffffffu));
    }
}

// merge C candidates (score f32 holding a bf16 value, global idx) per query down to k, same ordering
__global__ __launch_bounds__(1024) void topk_merge_kernel(const float* __restrict__ cs, const int* __restrict__ ci,
                                                           int C, int k, float* __restrict__ out_scores,
                                                           int* __restrict__ out_idx) {
    extern __shared__ unsigned long long mc[];
    const int q = blockIdx.x, tid = threadIdx.x;
    int n2 = 1;
    while (n2 < C) n2 <<= 1;
    for (int i = tid; i < n2; i += blockDim.x) {
        unsigned long long v = 0ull;
        if (i < C) {
            const float s = cs[(size_t)q * C + i];
            // full 32-bit sortable key (scores are bf16 values, but keep it generic)
            uint32_t b = __float_as_uint(s);
            b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
            v = ((unsigned long long)b << 32) | (unsigned)(0x7fffffff - ci[(size_t)q * C + i]);
        }
        mc[i] = v;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n2; i += blockDim.x) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = ((i & size) == 0);
                    const unsigned long long a = mc[i], b = mc[j];
                    if ((a < b) == desc) { mc[i] = b; mc[j] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += blockDim.x) {
        const unsigned long long c = mc[i];
        uint32_t b = (uint32_t)(c >> 32);
        b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
        out_scores[(size_t)q * k + i] = __uint_as_float(b);
        out_idx[(size_t)q * k + i] = 0x7fffffff - (int)(unsigned)(c & 0xffffffffu);
    }
}

// ---------------------------------------------------------------------------------------------
// K13 patchwise template score (src/pipeline/estimators/pose_estimator.py:85-88,
// online_pose_estimator.py:68-79) with the reference's bf16 rounding points:
//   tn = bf16(t / max(bf16(||t||), eps)) per patch row;  d[t,p] = bf16( tn . qn[p] );
//   score[t] = bf16( (sum_p d[t,p]) / P )
// qn is the already-normalised (or, frame-0 quirk, raw) query [P, D]; weights optional [T,P] f32
// (mask_scores variant: score = sum(d*w)/sum(w)).
// bf16( x / n ) with the IEEE quotient's rounding but without its ~12-instruction expansion: r = v_rcp_f32(n) (1 ulp), one
// residual correction step puts q within 1 fp32 ulp of the correctly rounded quotient, and that can only change the
// bf16 rounding when q's discarded 16 bits sit next to the midpoint 0x8000 — only those lanes (9 in 65536) take the exact
// division.  Bit-identical to rbf(__fdiv_rn(x, n)) (what the oracle computes); the kernel was VALU-bound on the divisions.
__device__ __forceinline__ float div_rbf(float x, float n, float r) {
    float q = x * r;
    const float e = __fmaf_rn(-q, n, x);
    q = __fmaf_rn(e, r, q);
    if ((__float_as_uint(q) & 0xffffu) - 0x7ffcu <= 8u) q = __fdiv_rn(x, n);
    return rbf(q);
}

template <int NCH>
__global__ __launch_bounds__(256) void template_dots_kernel(const bf16_t* __restrict__ tmpl,
                                                            const bf16_t* __restrict__ qn, float* __restrict__ dots,
                                                            int T, int P, int D) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
    const long rows = (long)T * P;
    // the row a wave will process NEXT is requested before the current one is reduced: two rows (4 KB) in flight per wave
    uint4 na[NCH], nb[NCH];
    auto fetch = [&](long r) {
        const int pidx = (int)(r % P);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int base = (c * 64 + lane) * 8;
            na[c] = make_uint4(0, 0, 0, 0);
            nb[c] = make_uint4(0, 0, 0, 0);
            if (base < D) {
                na[c] = *(const uint4*)(tmpl + (size_t)r * D + base);
                nb[c] = *(const uint4*)(qn + (size_t)pidx * D + base);
            }
        }
    };
    if (wave < rows) fetch(wave);
    for (long r = wave; r < rows; r += nwave) {
        float x[NCH][8], qq[NCH][8];
        float ss = 0.f;
        uint4 ca[NCH], cb[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { ca[c] = na[c]; cb[c] = nb[c]; }
        if (r + nwave < rows) fetch(r + nwave);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint4 a = ca[c], b = cb[c];
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[c][2 * e] = lo_bf(aw[e]); x[c][2 * e + 1] = hi_bf(aw[e]);
                qq[c][2 * e] = lo_bf(bw[e]); qq[c][2 * e + 1] = hi_bf(bw[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __fmaf_rn(x[c][e], x[c][e], ss);
        }
        ss = wave_sum(ss);
        const float nrm = fmaxf(rbf(fp_sqrt_rn(ss)), 1e-12f);
        const float rinv = __builtin_amdgcn_rcpf(nrm);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __fmaf_rn(div_rbf(x[c][e], nrm, rinv), qq[c][e], acc);
        acc = wave_sum(acc);
        if (lane == 0) dots[r] = rbf(acc);
    }
}

// Pre-normalised template store: the reference normalises the cached [T,P,D] tensor on every call
// (pose_estimator.py:85-88); F.normalize of a bf16 tensor IS a bf16 tensor, so storing tn once (fp_l2_normalize when the
// features enter the cache) is bit-identical to the reference's intermediate, and the scorer becomes a streaming dot:
//   d[t,p] = bf16( tn[t,p,:] . qn[p,:] )   in the canonical dot64 order (same chain as template_dots_kernel).
// A wave owns ONE patch index p and walks templates t = t0, t0 + TS, ...: its query row stays unpacked in registers (no
// second load stream), template rows (2 KB, full lines) are requested two ahead of the one being reduced.
template <int NCH>
__global__ __launch_bounds__(256) void template_dots_normed_kernel(const bf16_t* __restrict__ tn, const bf16_t* __restrict__ qn,
                                                                   float* __restrict__ dots, int T, int P, int D, int TS) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int pidx = wave / TS, t0 = wave % TS;
    if (pidx >= P) return;
    float qq[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int base = (c * 64 + lane) * 8;
        uint4 b = make_uint4(0, 0, 0, 0);
        if (base < D) b = *(const uint4*)(qn + (size_t)pidx * D + base);
        const uint32_t bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { qq[c][2 * e] = lo_bf(bw[e]); qq[c][2 * e + 1] = hi_bf(bw[e]); }
    }
    const size_t tstride = (size_t)P * D;
    const bf16_t* row = tn + (size_t)pidx * D;
    auto fetch = [&](int t, uint4 (&dst)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int base = (c * 64 + lane) * 8;
            dst[c] = make_uint4(0, 0, 0, 0);
            if (base < D && t < T) {
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
                const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)(row + (size_t)t * tstride + base));
                dst[c] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
    };
    auto reduce = [&](const uint4 (&src)[NCH], int t) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t aw[4] = {src[c].x, src[c].y, src[c].z, src[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = __fmaf_rn(lo_bf(aw[e]), qq[c][2 * e], acc);
                acc = __fmaf_rn(hi_bf(aw[e]), qq[c][2 * e + 1], acc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) dots[(size_t)t * P + pidx] = rbf(acc);
    };
    uint4 a0[NCH], a1[NCH], a2[NCH];
    fetch(t0, a0);
    fetch(t0 + TS, a1);
    for (int t = t0; t < T; t += 3 * TS) {          // three register buffers in rotation: two rows always in flight
        fetch(t + 2 * TS, a2);
        reduce(a0, t);
        if (t + TS >= T) break;
        fetch(t + 3 * TS, a0);
        reduce(a1, t + TS);
        if (t + 2 * TS >= T) break;
        fetch(t + 4 * TS, a1);
        reduce(a2, t + 2 * TS);
    }
}

__global__ __launch_bounds__(64) void template_mean_kernel(const float* __restrict__ dots,
                                                           const float* __restrict__ weights,
                                                           float* __restrict__ scores, int T, int P) {
    const int t = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f, wacc = 0.f;
    for (int pidx = lane; pidx < P; pidx += 64) {
        const float d = dots[(size_t)t * P + pidx];
        if (weights) {
            const float w = weights[(size_t)t * P + pidx];
            acc += d * w;  // bf16 score * fp32 mask promotes to fp32 in the reference (online :72-74)
            wacc += w;
        } else {
            acc += d;
        }
    }
    acc = wave_sum(acc);
    if (weights) {
        wacc = wave_sum(wacc);
        if (lane == 0) scores[t] = acc / wacc;
    } else if (lane == 0) {
        scores[t] = rbf(acc / (float)P);
    }
}


// ---------------------------------------------------------------------------------------------
// K11 per-view fine re-rank (scripts/extract_proposals_ground.py:147-160; video variant :160-176):
//   for each coarse candidate c of query q:  s_v = bf16( normalize_bf16(view_v) . f_q ) over the mesh's views,
//   score[q,c] = float32 numpy mean of the top-k s_v (sorted descending, numpy pairwise summation order).
// The per-view descriptors are kept raw fp32->bf16 (as np.load(...).to(bf16)) in one device-resident store
// [sum_views, D] with per-mesh row offsets; they are normalised on the fly with the reference's rounding points.
// One workgroup per (query, candidate); one wave per view row, scores into LDS, bitonic sort, ordered mean.
constexpr int RR_MAXV = 1024;
template <int NCH>
__global__ __launch_bounds__(256) void rerank_views_kernel(const bf16_t* __restrict__ views, const int* __restrict__ offsets,
                                                           const int* __restrict__ cand, const bf16_t* __restrict__ queries,
                                                           float* __restrict__ out, int C, int D, int k) {
    __shared__ float sc[RR_MAXV];
    const int q = blockIdx.y, c = blockIdx.x;
    const int mesh = cand[(size_t)q * C + c];
    const int r0 = offsets[mesh], nv = min(offsets[mesh + 1] - r0, RR_MAXV);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float qv[NCH][8];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int base = (ch * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[ch][e] = base < D ? bf2f(queries[(size_t)q * D + base + e]) : 0.f;
    }
    for (int v = wave; v < nv; v += 4) {
        const bf16_t* row = views + (size_t)(r0 + v) * D;
        float x[NCH][8];
        float ss = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int base = (ch * 64 + lane) * 8;
            uint4 a = make_uint4(0, 0, 0, 0);
            if (base < D) a = *(const uint4*)(row + base);
            const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[ch][2 * e] = lo_bf(w[e]); x[ch][2 * e + 1] = hi_bf(w[e]); }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __fmaf_rn(x[ch][e], x[ch][e], ss);
        }
        ss = wave_sum(ss);
        const float nrm = fmaxf(rbf(fp_sqrt_rn(ss)), 1e-12f);
        const float rinv = __builtin_amdgcn_rcpf(nrm);
        float acc = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __fmaf_rn(div_rbf(x[ch][e], nrm, rinv), qv[ch][e], acc);
        acc = wave_sum(acc);
        if (lane == 0) sc[v] = rbf(acc);
    }
    int n2 = 1;
    while (n2 < nv) n2 <<= 1;
    __syncthreads();
    for (int i = nv + threadIdx.x; i < n2; i += blockDim.x) sc[i] = -3.0e38f;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = ((i & size) == 0);
                    const float a = sc[i], b = sc[j];
                    if ((a < b) == desc) { sc[i] = b; sc[j] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        // numpy float32 mean of the kk top values: pairwise summation (8 running sums, tree, sequential tail; n <= 128)
        const int kk = min(k, nv);
        float res;
        if (kk < 8) {
            res = 0.f;
            for (int i = 0; i < kk; ++i) res += sc[i];
        } else {
            float r[8];
            for (int j = 0; j < 8; ++j) r[j] = sc[j];
            int i = 8;
            for (; i < kk - (kk % 8); i += 8)
                for (int j = 0; j < 8; ++j) r[j] += sc[i + j];
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; i < kk; ++i) res += sc[i];
        }
        out[(size_t)q * C + c] = kk > 0 ? __fdiv_rn(res, (float)kk) : -3.0e38f;
    }
}

}  // namespace

int fp_cast_f32_bf16(const float* x, bf16_t* y, size_t n, hipStream_t s) {
    if (n == 0) return FP_OK;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, s, x, y, n);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

// keys: workspace [Q, N] u16
int fp_bank_scan(const bf16_t* bank, const bf16_t* queries, uint16_t* keys, int ldk, int N, int D, int Q, hipStream_t s) {
    FP_REQUIRE(N > 0 && Q > 0 && D % 8 == 0 && D <= 1536, "bank_scan: bad shape N=%d D=%d Q=%d", N, D, Q);
    const int nch = cdiv(D, 512);
    // grid: 2..8 workgroups (4 waves each) per CU, whichever leaves the smallest remainder of rows per wave
    static int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    int blocks = ncu * 2;
    {
        double best = 1e30;
        for (int bpc = 2; bpc <= 8; ++bpc) {
            const long nw = (long)ncu * bpc * 4;
            const double waste = (double)(((long)N + nw - 1) / nw * nw) / N;
            if (waste <= best) { best = waste; blocks = ncu * bpc; }
        }
        blocks = std::min(blocks, cdiv(N, 4));
    }
    for (int qb = 0; qb < Q;) {
        const int left = Q - qb;
#define FP_SCAN(NCHV, QTV)                                                                                          \
    do {                                                                                                            \
        if (D == NCHV * 512)                                                                                        \
            hipLaunchKernelGGL((bank_scan_kernel<NCHV, QTV, true>), dim3(blocks), dim3(256), 0, s, bank, queries,   \
                               keys, N, D, qb, Q, ldk);                                                             \
        else                                                                                                        \
            hipLaunchKernelGGL((bank_scan_kernel<NCHV, QTV, false>), dim3(blocks), dim3(256), 0, s, bank, queries,  \
                               keys, N, D, qb, Q, ldk);                                                             \
    } while (0)
        // (8 queries per pass was measured in round 4 — profiles/r04_ab.md §3: 44.4 us against 2 x 21.9 us for two 4-query passes.
        // Beyond one query the pass is bound by vector-ALU issue, ~4 us per extra query, and at 8 x 16 unpacked query values per
        // lane the kernel drops to one wave per SIMD; a matrix-pipe form would change the canonical summation order that makes the
        // indices bit-exact.  Four per pass stays.)
        if (left >= 4) {
            if (nch == 1) FP_SCAN(1, 4); else if (nch == 2) FP_SCAN(2, 4); else FP_SCAN(3, 4);
            qb += 4;
        } else {
            if (nch == 1) FP_SCAN(1, 1); else if (nch == 2) FP_SCAN(2, 1); else FP_SCAN(3, 1);
            qb += 1;
        }
#undef FP_SCAN
        FP_LAUNCH_CHECK();
    }
    return FP_OK;
}

int fp_topk_select(const uint16_t* keys, int ldk, int N, int Q, int k, int idx_offset, float* out_scores, int* out_idx,
                   hipStream_t s) {
    FP_REQUIRE(k > 0 && k <= KMAX && k <= N, "topk: k=%d out of range (N=%d, max %d)", k, N, KMAX);
    FP_REQUIRE(ldk >= N && ldk % 8 == 0, "topk: key row stride %d must be >= N and a multiple of 8", ldk);
    const size_t key_bytes = (size_t)ldk * 2;
#ifdef FP_LAB
    static int env_sel = [] { const char* e = getenv("FP_TOPK_SELECT"); return e ? atoi(e) : 1; }();   // 0: histogram kernels (A/B)
    const bool use_sel = fp_opt_get(FP_OPT_TOPK_SELECT, env_sel) != 0;
#else
    constexpr bool use_sel = true;
#endif
    if (use_sel && N <= SEL_CAP) {   // counting select on register-resident keys (no LDS atomics)
        const size_t lds = (size_t)SEL_CAP * 2;
        FP_DYN_LDS_ONCE(topk_select_reg_kernel, 128 * 1024);
        hipLaunchKernelGGL(topk_select_reg_kernel, dim3(Q), dim3(SEL_T), lds, s, keys, ldk, N, k, idx_offset, out_scores, out_idx);
        FP_LAUNCH_CHECK();
        return FP_OK;
    }
    if (key_bytes <= 128 * 1024) {   // the key row fits beside the candidate buffer: all passes from LDS
        FP_DYN_LDS_ONCE(topk_select_kernel<true>, 128 * 1024);
        hipLaunchKernelGGL(topk_select_kernel<true>, dim3(Q), dim3(SEL_T), key_bytes, s, keys, ldk, N, k, idx_offset, out_scores, out_idx);
    } else {
        hipLaunchKernelGGL(topk_select_kernel<false>, dim3(Q), dim3(SEL_T), 0, s, keys, ldk, N, k, idx_offset, out_scores, out_idx);
    }
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_topk_merge_launch(const float* cs, const int* ci, int Q, int C, int k, float* out_scores, int* out_idx,
                  hipStream_t s) {
    FP_REQUIRE(C > 0 && k > 0 && k <= C && C <= 8192, "topk_merge: bad C=%d k=%d", C, k);
    int n2 = 1;
    while (n2 < C) n2 <<= 1;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(Q), dim3(1024), n2 * sizeof(unsigned long long), s, cs, ci, C, k,
                       out_scores, out_idx);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

// dots: workspace [T*P] f32
int fp_template_score_launch(const bf16_t* tmpl, const bf16_t* qn, const float* weights, float* dots, float* scores, int T,
                      int P, int D, int templates_normalised, hipStream_t s) {
    FP_REQUIRE(T > 0 && P > 0 && D % 8 == 0 && D <= 1536, "template_score: bad shape");
    const long rows = (long)T * P;
    if (templates_normalised) {
        // ~16 waves per CU in flight: TS template slices per patch index
        const int TS = std::max(1, std::min(T, cdiv(256 * 16, P)));
        const int nblk = cdiv(P * TS, 4);
        const int nchn = cdiv(D, 512);
        if (nchn == 1) hipLaunchKernelGGL(template_dots_normed_kernel<1>, dim3(nblk), dim3(256), 0, s, tmpl, qn, dots, T, P, D, TS);
        else if (nchn == 2) hipLaunchKernelGGL(template_dots_normed_kernel<2>, dim3(nblk), dim3(256), 0, s, tmpl, qn, dots, T, P, D, TS);
        else hipLaunchKernelGGL(template_dots_normed_kernel<3>, dim3(nblk), dim3(256), 0, s, tmpl, qn, dots, T, P, D, TS);
        FP_LAUNCH_CHECK();
        hipLaunchKernelGGL(template_mean_kernel, dim3(T), dim3(64), 0, s, dots, weights, scores, T, P);
        FP_LAUNCH_CHECK();
        return FP_OK;
    }
    const int blocks = (int)std::min<long>((rows + 3) / 4, 256 * 16);
    const int nch = cdiv(D, 512);
    if (nch == 1) hipLaunchKernelGGL(template_dots_kernel<1>, dim3(blocks), dim3(256), 0, s, tmpl, qn, dots, T, P, D);
    else if (nch == 2) hipLaunchKernelGGL(template_dots_kernel<2>, dim3(blocks), dim3(256), 0, s, tmpl, qn, dots, T, P, D);
    else hipLaunchKernelGGL(template_dots_kernel<3>, dim3(blocks), dim3(256), 0, s, tmpl, qn, dots, T, P, D);
    FP_LAUNCH_CHECK();
    hipLaunchKernelGGL(template_mean_kernel, dim3(T), dim3(64), 0, s, dots, weights, scores, T, P);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

// per-view re-rank: out [Q,C] f32.  k <= 128 (numpy pairwise block), views per mesh <= 1024.
int fp_rerank_views_launch(const bf16_t* views, const int* offsets, const int* cand, const bf16_t* queries, float* out,
                           int Q, int C, int D, int k, hipStream_t s) {
    FP_REQUIRE(Q > 0 && C > 0 && D % 8 == 0 && D <= 1536 && k > 0 && k <= 128, "rerank_views: bad shape");
    const int nch = cdiv(D, 512);
    dim3 grid(C, Q);
    if (nch == 1) hipLaunchKernelGGL(rerank_views_kernel<1>, grid, dim3(256), 0, s, views, offsets, cand, queries, out, C, D, k);
    else if (nch == 2) hipLaunchKernelGGL(rerank_views_kernel<2>, grid, dim3(256), 0, s, views, offsets, cand, queries, out, C, D, k);
    else hipLaunchKernelGGL(rerank_views_kernel<3>, grid, dim3(256), 0, s, views, offsets, cand, queries, out, C, D, k);
    FP_LAUNCH_CHECK();
    return FP_OK;
}
