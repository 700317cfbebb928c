// bf16 MFMA GEMM for the ViT linear layers (K1,K3,K5,K6,K7 of SURVEY.md §2.3).
//   C[M,N] = epilogue( X[M,K] · W[N,K]^T )      X, W, C bf16 row-major, fp32 accumulate.
// W keeps the torch nn.Linear layout ([out,in]) so a DINOv2 state-dict tensor is used as-is.
#pragma once
#include "common.h"

enum FpGemmEpi {
    FP_EPI_BIAS = 0,        // C = acc + bias                           (QK part of qkv)
    FP_EPI_BIAS_GELU = 1,   // C = gelu_erf(acc + bias)                 (fc1)
    FP_EPI_BIAS_LS_RES = 2, // C = resid + gamma * (acc + bias)         (attn.proj / fc2 + LayerScale + residual)
    FP_EPI_PATCH = 3,       // C[tokrow(m)] = bf16(acc + bias) + pos[p] (patch-embed -> token buffer)
    FP_EPI_VT = 4,          // Vt[b,h,d,t] = acc + bias                 (V part of qkv, stored transposed per head)
    // ---- LayerNorm folded into the consuming GEMM:  LN(x) W^T + b  =  rstd (x W'^T - mean colsum(W')) + b'
    // with W' = W diag(gamma_ln) (bf16), colsum over the bf16 W', b' = b + W beta_ln (fp32).  X is the RAW residual stream.
    // The correction rides in the ACCUMULATOR INIT: acc0[m][n] = b'[n] sigma[m] - mean[m] cs[n] (sigma = 1/rstd), the K loop adds
    // x W'^T on top in fp32, and the epilogue is one multiply by rstd[m] — no per-feature constants are live in the epilogue.
    // acc0 is a rank-2 outer product and is formed ON THE MATRIX PIPE: one extra MFMA per accumulator block whose operands carry,
    // in 6 of their 32 k-slots, the two-piece bf16 splits (hi + lo, 16 mantissa bits) of (b', cs) per feature and (sigma, -mean) per
    // row:  b' sigma ~ b'h sh + b'h sl + b'l sh  (relative error 2^-16), likewise mean cs.  Vector-ALU work at a tile boundary costs
    // four times its instruction count (the four waves of a SIMD do it in lock-step with the matrix pipe idle): the VALU form of this
    // init was +4-5 % on the qk / V / fc1 launches (profiles/r03_ab.md).
    FP_EPI_LN_BIAS = 5,     // C = rstd (acc - mean cs) + b'            (QK part of qkv on the un-normalised x)
    FP_EPI_LN_GELU = 6,     // C = gelu_erf(bf16(rstd (acc - mean cs) + b'))   (fc1)
    FP_EPI_LN_VT = 7,       // Vt[b,h,d,t] = rstd (acc - mean cs) + b'  (V part of qkv)
    // ---- the producer side: LS_RES that also writes, per row and 64-column block, (sum x, sum x^2) of its bf16 OUTPUT rows
    FP_EPI_LS_RES_STATS = 8 // C = resid + gamma * (acc + bias);  stat_part[n/64][m] = (sum, sum of squares) over the block
};

template <int EPI>
struct FpEpiTraits {
    static constexpr bool LN = EPI == FP_EPI_LN_BIAS || EPI == FP_EPI_LN_GELU || EPI == FP_EPI_LN_VT;
    static constexpr bool TRANS = EPI == FP_EPI_VT || EPI == FP_EPI_LN_VT;
    static constexpr bool GELU = EPI == FP_EPI_BIAS_GELU || EPI == FP_EPI_LN_GELU;
    static constexpr bool LSRES = EPI == FP_EPI_BIAS_LS_RES || EPI == FP_EPI_LS_RES_STATS;
    static constexpr bool STATS = EPI == FP_EPI_LS_RES_STATS;
};

struct FpGemmArgs {
    const bf16_t* X; int ldx;      // [M,K]
    const bf16_t* W; int ldw;      // [N,K]
    bf16_t* C; int ldc;            // [M,N] (or token buffer / Vt)
    const bf16_t* bias;            // [N] or null
    const bf16_t* gamma;           // [N] LayerScale
    const bf16_t* resid; int ldr;  // [M,N]
    int M, N, K;
    // FP_EPI_PATCH: m = b*P + p  ->  row b*npad + tok_off + p ; pos is [P,N] bf16
    const bf16_t* pos; int P; int npad; int tok_off;
    // FP_EPI_VT: rows m = b*npad + t ; n = h*64 + d ; Vt is [B,H,64,npad]
    int heads;
    // FP_EPI_BIAS_GELU: device table of fp_gemm_gelu_table() (filled in by fp_gemm_bf16; callers leave it null)
    const uint16_t* gelu_tab;
    // FP_EPI_LN_*: init-MFMA operand records, 8 bf16 (16 bytes) each — per row   {sh, sl, sh, -mh, -ml, -mh, 0, 0}  (sigma, mean),
    // per feature {b'h, b'h, b'l, ch, ch, cl, 0, 0}  (b', colsum(W')) — and rstd = 1/sigma [M] fp32 for the epilogue
    uint4* ln_mfrag;     // (written by a small-tier launch that finalises FpGemmArgs::ln_part in its prologue, read-only otherwise)
    float* ln_rstd;
    const uint4* ln_cfrag;
    // optional (row-major LN epilogues): the producing GEMM's partial row statistics [D/64][ln_part_ld] (sum, sum of squares).  When set,
    // the row records are not there yet: a launch that runs on the small tile tiers finalises the rows of each tile in the kernel's
    // prologue (and writes ln_mfrag / ln_rstd back for later consumers of the same rows — every workgroup of a row block writes the same
    // bits); a launch that takes the big tier runs fp_stats_finalize first.  Saves a 5 us dependent kernel per LayerNorm at small batch.
    const float2* ln_part; int ln_part_ld; int ln_part_nb; float ln_eps; float ln_inv_d;   // nb = D / 64 blocks, inv_d = 1 / D (host-rounded)
    // FP_EPI_LS_RES_STATS: partial row statistics [N/64][M] (sum, sum of squares), N % 64 == 0
    float2* stat_part;
    int no_split;  // 1: never split this launch by rows between the tile tiers (set on the parts of a split; callers may set it too)
    int stat_ld;   // row stride of stat_part (= the whole problem's M; a row-split launch covers only part of it).  0 = M
    int ring;      // K-tile ring depth of the 64x64 tier (2 .. 8 buffers; chosen by the launcher from the grid size, gemm_bf16.hip)
    // ---- balanced tier (stream-K, gemm_bf16.hip): a launch that would leave most of the chip idle, or fill its last round badly, is run
    // by a grid of G workgroups that share the (tile, K step) units evenly; a tile whose K range is spread over several workgroups is
    // completed by the LAST of them to arrive, which adds the fp32 partial tiles in K order (run-to-run deterministic).  The caller
    // lends the scratch: sk_ws = 2 partial tiles per workgroup (fp32), sk_cnt = one zero-initialised arrival counter per tile (the
    // kernel leaves them zero).  Null = the tier is not used.  sk_mode: 0 = the launcher decides, 1 = never.
    float* sk_ws; size_t sk_ws_bytes;
    int* sk_cnt; int sk_cnt_n;
    int sk_mode;
#ifdef FP_LAB
    int dbg;   // LAB BUILD ONLY (libfreepose_hip_lab.so, tools/): measurement bits with wrong numerics — 2 = LN-folded kernels start from
               // zero accumulators, 8 = persistent kernels skip the epilogue, 16 = epilogue without its stores, 32 = staggered start
#endif
};
// measurement hooks exist only in the lab build; in the product they are the constant 0 and the guarded code is not compiled in
#ifdef FP_LAB
#define FP_GEMM_DBG_BIT(p, bit) ((p).dbg & (bit))
#else
#define FP_GEMM_DBG_BIT(p, bit) 0
#endif

// (sigma, -mean) of one row as the init-MFMA operand record {sh, sl, sh, -mh, -ml, -mh, 0, 0} (two-piece bf16 splits), and 1 / sigma, from the
// per-64-column partial sums of the producing epilogue added in block order: the ONE definition used by stats_finalize_kernel
// (vit_misc.hip) and by the small-tier GEMM prologue (gemm_bf16.hip) — same bits wherever a row is finalised.
__device__ __forceinline__ uint4 fp_ln_row_record(float mean, float sigma) {
    const float sh = rbf(sigma), sl = rbf(sigma - sh), nm = -mean, mh = rbf(nm), ml = rbf(nm - mh);
    return make_uint4(pack_bf2(sh, sl), pack_bf2(sh, mh), pack_bf2(ml, mh), 0u);
}
__device__ __forceinline__ void fp_ln_finalize_row(const float2* __restrict__ part, size_t ld, int nb, int r, float inv_d, float eps, uint4& rec,
                                                   float& rstd) {
    float s = 0.f, q = 0.f;
    for (int b = 0; b < nb; ++b) {
        const float2 p = part[(size_t)b * ld + r];
        s += p.x;
        q += p.y;
    }
    const float mean = s * inv_d;
    const float var = fmaxf(__fmaf_rn(-mean, mean, q * inv_d), 0.f);
    const float sigma = __fsqrt_rn(var + eps);
    rec = fp_ln_row_record(mean, sigma);
    rstd = __builtin_amdgcn_rcpf(sigma);
}

// Tile order shared by the GEMM kernels.  blockIdx -> logical id (XCD-contiguous, bijective) -> (tile_m, tile_n) in
// column STRIPS of 4 n-tiles swept m-major: the 32 tiles an XCD runs concurrently then cover 8 X row-panels x 4 W
// panels, so a strip's W slabs (<= 2 MB) stay in the XCD's 4 MB L2 for the whole sweep and each X panel is fetched once
// per strip.  (rocprofv3, fc1 with N = 4096: the plain row-major order streamed the 8 MB weight matrix once per 2
// row-panels — FETCH_SIZE 9x the algorithmic bytes.)
// blockIdx -> logical id: workgroups are dealt to the 8 XCDs round-robin; the remap gives each XCD a CONTIGUOUS run of logical ids
__device__ __forceinline__ int fp_gemm_xcd_remap(int block, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = block & 7, pos = block >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}
// logical tile id -> (tile_m, tile_n) in column strips of SW n-tiles swept m-major
template <int SW = 4>
__device__ __forceinline__ void fp_gemm_tile_of_id(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int full = tiles_n / SW, tail = tiles_n - full * SW;
    const int in_full = full * tiles_m * SW;
    if (id < in_full) {
        const int strip = id / (tiles_m * SW), rem = id - strip * (tiles_m * SW);
        tm = rem / SW;
        tn = strip * SW + (rem - tm * SW);
    } else {
        const int rem = id - in_full;
        tm = rem / tail;
        tn = full * SW + (rem - tm * tail);
    }
}
template <int SW = 4>
__device__ __forceinline__ void fp_gemm_tile(int block, int nblocks, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int id = fp_gemm_xcd_remap(block, nblocks);
    const int full = tiles_n / SW, tail = tiles_n - full * SW;
    const int in_full = full * tiles_m * SW;
    if (id < in_full) {
        const int strip = id / (tiles_m * SW), rem = id - strip * (tiles_m * SW);
        tm = rem / SW;
        tn = strip * SW + (rem - tm * SW);
    } else {
        const int rem = id - in_full;
        tm = rem / tail;
        tn = full * SW + (rem - tm * tail);
    }
}

// Launch on `stream`. Returns FP_OK / error code (fp_last_error() has the text).
int fp_gemm_bf16(const FpGemmArgs& a, int epi, hipStream_t stream);
// true when a row-major LN-folded launch of this size stays on the small tile tiers, i.e. finalises FpGemmArgs::ln_part in its prologue
// (otherwise fp_gemm_bf16 runs fp_stats_finalize first; callers that account for that kernel separately call it themselves)
bool fp_gemm_fuses_ln_part(int M, int N);
// builds (once per device) and returns the bf16 -> bf16 GELU table the fc1 epilogue gathers from
int fp_gemm_gelu_table(const uint16_t** out);
// the hand-scheduled 256x256 kernel (gemm_asm.hip): the big-tile tier of the row-major epilogues
bool fp_gemm_asm_supported(const FpGemmArgs& a, int epi);
bool fp_gemm_asm_preferred(const FpGemmArgs& a, int epi);   // supported AND measured faster than the 16-wave kernel
int fp_gemm_asm(const FpGemmArgs& a, int epi, int waves, hipStream_t stream);   // waves: 4 (one per SIMD) or 8 (two per SIMD)
// the hand-scheduled 128x128 kernel (gemm_asm.hip, geometry "S"): the small tier of the row-major epilogues
bool fp_gemm_asm_small_supported(const FpGemmArgs& a, int epi);
int fp_gemm_asm_small(const FpGemmArgs& a, int epi, hipStream_t stream);
// y[i] = bf16(gelu_erf(x[i])): the direct expression, elementwise (test entry fp_op_gelu)
int fp_gemm_gelu_direct(const bf16_t* x, bf16_t* y, size_t n, hipStream_t stream);
// name of the kernel variant used for (epi) — for profiles / bench bookkeeping
const char* fp_gemm_kernel_name(int epi);
