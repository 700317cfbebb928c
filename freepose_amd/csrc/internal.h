// internal launcher prototypes shared by the C-ABI translation units
#pragma once
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "gemm_bf16.h"

struct fp_ctx {
    int device = 0;
    // named, grow-only workspaces (device memory).  One context per thread/GPU, no locking.
    struct Buf { void* p = nullptr; size_t bytes = 0; };
    std::map<std::string, Buf> bufs;
    int get(const char* name, size_t bytes, void** out);
    // optional RCCL communicator (comm.hip: fp_comm_init); null = single rank
    void* comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    // run-time options of THIS context (fp_ctx_set_option); -1 = built-in default
    int opt_ln_fused = -1;       // 1 (default): LayerNorm 1 / 2 folded into the qkv / fc1 GEMMs of fp_vit_forward; 0: separate kernel
    int opt_raster_tiled = -1;   // default: by triangle count / image size; 1 / 0 force the LDS-tiled / global-buffer strategy
    int opt_comm_timeout_s = -1; // seconds fp_comm_init waits for the rendezvous of all ranks before it fails (default 180)
    int opt_row_split = -1;      // 1 (default): GEMM launches between the tile tiers are split by rows; 0: never
    int opt_stream_k = -1;       // 1 (default): small GEMM launches may run on the balanced tier (K slices summed in K order: results depend
                                 // on the launch size in the last place); 0: never — every tier then gives the same bits for a row
    // scratch of the balanced GEMM tier (gemm_bf16.h FpGemmArgs::sk_*): fp32 partial tiles + zero-initialised arrival counters
    float* sk_ws = nullptr;
    int* sk_cnt = nullptr;
    static constexpr size_t SK_WS_BYTES = (size_t)64 << 20;
    static constexpr int SK_CNT_N = 16384;
    int sk_scratch(FpGemmArgs& g, hipStream_t s);   // lends the scratch to a launch (allocates on first use)
    int* ffa_arrive = nullptr;   // per-crop arrival counters of the fused FFA launch (zero between calls: the last arriver resets its own)
    size_t total() const;
    void release();
};

// attention.hip
// log2(e) / sqrt(head_dim = 64): what the softmax multiplies q k^T by before exp2 — or what the q rows of the folded qkv weights carry
constexpr float FP_ATTN_QSCALE = 1.4426950408889634f / 8.0f;
int fp_attention_fwd(const bf16_t* QK, int ldqk, const bf16_t* Vt, bf16_t* O, int ldo, int B, int H, int n_tok,
                     int npad, bool q_prescaled, hipStream_t stream);
// vit_misc.hip
int fp_im2col_norm(const bf16_t* img, bf16_t* A, int B, int H, int W, int ps, int KP, hipStream_t s);
int fp_token_init(bf16_t* X, const bf16_t* cls, const bf16_t* pos0, const bf16_t* reg, int nreg, int B, int n_tok,
                  int npad, int D, hipStream_t s);
int fp_layernorm(const bf16_t* X, bf16_t* Y, const bf16_t* gamma, const bf16_t* beta, int rows, int D, float eps,
                 int rows_per_b, int in_stride_b, int in_off, hipStream_t s, int l2_normalize = 0);
int fp_posembed_aa(const bf16_t* src, bf16_t* dst, int G, int gh, int gw, int D, hipStream_t s);
int fp_ffa_pool(const bf16_t* feats, const uint8_t* mask, bf16_t* out, float* out_f32, int B, int P, int D, int gh,
                int gw, int cell, uint8_t* pm_scratch, hipStream_t s, bf16_t* out_norm = nullptr, int* arrive = nullptr);
int fp_l2norm_rows(const bf16_t* X, bf16_t* Y, int rows, int D, hipStream_t s);
// LayerNorm folded into the consuming GEMM (gemm_bf16.h FP_EPI_LN_*): row statistics and the one-off weight fold
int fp_row_stats(const bf16_t* X, uint4* mfrag, float* rstd, int rows, int D, float eps, hipStream_t s);
int fp_stats_finalize(const float2* part, uint4* mfrag, float* rstd, int rows, int D, float eps, hipStream_t s, int part_ld = 0);   // part_ld: row stride of part (0 = rows)
int fp_ln_fold(const bf16_t* W, const bf16_t* gamma, const bf16_t* beta, const bf16_t* bias, bf16_t* Wf, uint4* cfrag, int N, int K,
               int n_scaled, float row_scale, hipStream_t s);
// retrieval.hip
int fp_cast_f32_bf16(const float* x, bf16_t* y, size_t n, hipStream_t s);
int fp_bank_scan(const bf16_t* bank, const bf16_t* queries, uint16_t* keys, int ldk, int N, int D, int Q, hipStream_t s);
int fp_topk_select(const uint16_t* keys, int ldk, int N, int Q, int k, int idx_offset, float* out_scores, int* out_idx,
                   hipStream_t s);
int fp_topk_merge_launch(const float* cs, const int* ci, int Q, int C, int k, float* out_scores, int* out_idx,
                         hipStream_t s);
int fp_rerank_views_launch(const bf16_t* views, const int* offsets, const int* cand, const bf16_t* queries, float* out,
                           int Q, int C, int D, int k, hipStream_t s);
int fp_template_score_launch(const bf16_t* tmpl, const bf16_t* qn, const float* weights, float* dots, float* scores,
                             int T, int P, int D, int templates_normalised, hipStream_t s);
