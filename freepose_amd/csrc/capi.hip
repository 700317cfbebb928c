// C-ABI entry points of libfreepose_hip.so (see include/freepose_hip.h) + context / ViT forward driver.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <cmath>

#include "../../include/freepose_hip.h"
#include "internal.h"

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void fp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* fp_last_error(void) { return g_err; }
extern "C" int fp_version(void) { return 100; }

#ifdef FP_LAB
// lab build only: process-global experiment toggles (tools/ A/B runs)
static int g_opts[FP_OPT_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
int fp_opt_get(int key, int dflt) { return (key >= 0 && key < FP_OPT_COUNT && g_opts[key] >= 0) ? g_opts[key] : dflt; }
extern "C" int fp_lab_set_option(const char* name, int value) {
    FP_REQUIRE(name, "lab_set_option: null name");
    if (!strcmp(name, "gemm_variant")) g_opts[FP_OPT_GEMM_VARIANT] = value;
    else if (!strcmp(name, "attn_slots")) g_opts[FP_OPT_ATTN_SLOTS] = value;
    else if (!strcmp(name, "gemm_dbg")) g_opts[FP_OPT_GEMM_DBG] = value;   // measurement hooks with wrong numerics
    else if (!strcmp(name, "attn_variant")) g_opts[FP_OPT_ATTN_VARIANT] = value;
    else if (!strcmp(name, "topk_select")) g_opts[FP_OPT_TOPK_SELECT] = value;
    else if (!strcmp(name, "gemm_ring")) g_opts[FP_OPT_GEMM_RING] = value;   // cap on the 64x64 tier's K-tile ring depth
    else if (!strcmp(name, "gemm_sk")) g_opts[FP_OPT_GEMM_SK] = value;       // balanced tier: 1 never, 2 / 3 / 4 force a form (gemm_bf16.hip)
    else if (!strcmp(name, "gemm_sk_grid")) g_opts[FP_OPT_GEMM_SK_GRID] = value;
    else if (!strcmp(name, "raster_dbg")) g_opts[FP_OPT_RASTER_DBG] = value;           // tile kernel ablation bits (raster.hip)
    else if (!strcmp(name, "gemm_stream_mb")) g_opts[FP_OPT_GEMM_STREAM_MB] = value;   // big tier: outputs above this many MiB are stored non-temporally
    else { fp_set_error("lab_set_option: unknown option '%s'", name); return FP_ERR_INVALID; }
    return FP_OK;
}
// lab build only: copy the head of the balanced tier's scratch buffer to the host (gemm_dbg = 1024 leaves per-workgroup phase timestamps there)
extern "C" int fp_lab_read_scratch(fp_ctx* ctx, void* host, size_t bytes) {
    FP_REQUIRE(ctx && host && ctx->sk_ws && bytes <= fp_ctx::SK_WS_BYTES, "lab_read_scratch: no scratch");
    FP_HIP(hipDeviceSynchronize());
    FP_HIP(hipMemcpy(host, ctx->sk_ws, bytes, hipMemcpyDeviceToHost));
    return FP_OK;
}
#endif

#ifdef FP_LAB
// lab build only: copy a named workspace of the context to the host (raster.dbg: the tile kernel's phase clocks)
extern "C" int fp_lab_read_buffer(fp_ctx* ctx, const char* name, void* host, size_t bytes) {
    FP_REQUIRE(ctx && name && host, "lab_read_buffer: null argument");
    auto it = ctx->bufs.find(name);
    FP_REQUIRE(it != ctx->bufs.end() && it->second.bytes >= bytes, "lab_read_buffer: no such buffer");
    FP_HIP(hipDeviceSynchronize());
    FP_HIP(hipMemcpy(host, it->second.p, bytes, hipMemcpyDeviceToHost));
    return FP_OK;
}
#endif

extern "C" int fp_ctx_set_option(fp_ctx* ctx, const char* name, int value) {
    FP_REQUIRE(ctx && name, "ctx_set_option: null argument");
    if (!strcmp(name, "ln_fused")) ctx->opt_ln_fused = value;
    else if (!strcmp(name, "raster_tiled")) ctx->opt_raster_tiled = value;
    else if (!strcmp(name, "gemm_row_split")) ctx->opt_row_split = value;
    else if (!strcmp(name, "comm_timeout_s")) ctx->opt_comm_timeout_s = value;
#ifdef FP_LAB
    else if (!strcmp(name, "gemm_stream_k")) ctx->opt_stream_k = value;   // lab: 0 = never lend the balanced tier its scratch
#endif
    else { fp_set_error("ctx_set_option: unknown option '%s'", name); return FP_ERR_INVALID; }
    return FP_OK;
}

int fp_ctx::get(const char* name, size_t bytes, void** out) {
    Buf& b = bufs[name];
    if (b.bytes < bytes) {
        if (b.p) FP_HIP(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
        size_t want = bytes + bytes / 8 + 4096;
        FP_HIP(hipMalloc(&b.p, want));
        b.bytes = want;
    }
    *out = b.p;
    return FP_OK;
}
int fp_ctx::sk_scratch(FpGemmArgs& g, hipStream_t s) {
    g.sk_mode = opt_stream_k == 0 ? 1 : 0;
#ifndef FP_LAB
    g.sk_mode = 1;   // the balanced tier is compiled into the lab build only (gemm_bf16.hip): the product lends no scratch
    (void)s;
#endif
    if (g.sk_mode == 1) return FP_OK;
    if (!sk_ws) {
        FP_HIP(hipMalloc((void**)&sk_ws, SK_WS_BYTES));
        FP_HIP(hipMalloc((void**)&sk_cnt, SK_CNT_N * sizeof(int)));
        FP_HIP(hipMemsetAsync(sk_cnt, 0, SK_CNT_N * sizeof(int), s));   // the kernels leave the counters at zero
        FP_HIP(hipStreamSynchronize(s));                                 // (other streams of this context may use them next)
    }
    g.sk_ws = sk_ws; g.sk_ws_bytes = SK_WS_BYTES;
    g.sk_cnt = sk_cnt; g.sk_cnt_n = SK_CNT_N;
    return FP_OK;
}
size_t fp_ctx::total() const {
    size_t t = 0;
    for (auto& kv : bufs) t += kv.second.bytes;
    return t;
}
void fp_ctx::release() {
    for (auto& kv : bufs)
        if (kv.second.p) (void)hipFree(kv.second.p);
    bufs.clear();
    if (sk_ws) (void)hipFree(sk_ws);
    if (sk_cnt) (void)hipFree(sk_cnt);
    if (ffa_arrive) (void)hipFree(ffa_arrive);
    sk_ws = nullptr;
    sk_cnt = nullptr;
    ffa_arrive = nullptr;
}

extern "C" int fp_ctx_create(int device, fp_ctx** out) {
    FP_REQUIRE(out, "ctx_create: null out");
    int n = 0;
    FP_HIP(hipGetDeviceCount(&n));
    FP_REQUIRE(device >= 0 && device < n, "ctx_create: device %d out of range (%d visible)", device, n);
    FP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FP_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fp_set_error("ctx_create: device %d is %s; this library contains gfx950 (MI355X) code only", device,
                     prop.gcnArchName);
        return FP_ERR_STATE;
    }
    // the code object is built for gfx950:sramecc+ only (the fc1 epilogue's LDS table gather relies on the d16 load behaviour of that
    // mode, gemm_epilogue.h): on a device or partition reporting sramecc- no kernel image would load — say so instead of failing on
    // the first launch with "no kernel image is available"
    if (strstr(prop.gcnArchName, "sramecc-")) {
        fp_set_error("ctx_create: device %d is %s; libfreepose_hip is built for gfx950:sramecc+ (SRAM ECC enabled) only", device, prop.gcnArchName);
        return FP_ERR_STATE;
    }
    if (int rc = fp_gemm_gelu_table(nullptr)) return rc;   // per-device constant table of the fc1 epilogue
    fp_ctx* c = new fp_ctx();
    c->device = device;
    *out = c;
    return FP_OK;
}
extern "C" int fp_comm_destroy(fp_ctx* ctx);
extern "C" int fp_ctx_destroy(fp_ctx* ctx) {
    if (!ctx) return FP_OK;
    (void)fp_comm_destroy(ctx);
    ctx->release();
    delete ctx;
    return FP_OK;
}
extern "C" size_t fp_ctx_workspace_bytes(const fp_ctx* ctx) { return ctx ? ctx->total() : 0; }

// ---------------------------------------------------------------------------------------------
// ViT
struct VitBlockW {
    const bf16_t *n1w = nullptr, *n1b = nullptr, *qkvw = nullptr, *qkvb = nullptr, *projw = nullptr,
                 *projb = nullptr, *ls1 = nullptr, *n2w = nullptr, *n2b = nullptr, *fc1w = nullptr, *fc1b = nullptr,
                 *fc2w = nullptr, *fc2b = nullptr, *ls2 = nullptr;
};
struct fp_vit {
    fp_ctx* ctx = nullptr;
    fp_vit_arch a{};
    int KP = 0;  // padded patch-embed K
    const bf16_t *cls = nullptr, *pos = nullptr, *reg = nullptr, *pe_b = nullptr, *normw = nullptr, *normb = nullptr;
    bf16_t* pe_w = nullptr;  // [dim, KP] private padded copy
    bf16_t* ones = nullptr;  // LayerScale gamma = 1 for checkpoints without ls*
    std::vector<VitBlockW> blk;
    // LayerNorm folded into the consuming GEMMs (gemm_bf16.h FP_EPI_LN_*): per block W' = W diag(gamma_ln) for qkv / fc1 and the
    // (colsum(W'), b') pairs, built on the device the first time a forward runs after the weights changed
    struct VitFold { bf16_t *qkvw = nullptr, *fc1w = nullptr; uint4 *qkv_cb = nullptr, *fc1_cb = nullptr; };
    std::vector<VitFold> fold;
    int folded_upto = 0;             // blocks [0, folded_upto) carry valid folded weights (a forward folds only the blocks it runs)
    hipEvent_t fold_ev = nullptr;    // recorded behind the last fold; forwards on OTHER streams wait for it
    hipStream_t fold_stream = nullptr;
    bool refold = false;             // a weight was replaced after a fold: forwards in flight on other streams may still read the fold buffers
    // pos-embed cache per (gh,gw)
    std::map<std::pair<int, int>, bf16_t*> pos_cache;
    // profiling
    bool prof = false;
    struct EvPair { hipEvent_t a, b; int cls; };
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    float ms_gemm = 0, ms_attn = 0, ms_other = 0;
    double gemm_flops = 0;   // algorithmic (unpadded) FLOPs of the GEMM launches issued while profiling
    long gemm_launches = 0;
};

extern "C" int fp_vit_create(fp_ctx* ctx, const fp_vit_arch* arch, fp_vit** out) {
    FP_REQUIRE(ctx && arch && out, "vit_create: null argument");
    FP_REQUIRE(arch->dim % 64 == 0 && arch->heads * 64 == arch->dim, "vit_create: dim=%d heads=%d (head dim must be 64)",
               arch->dim, arch->heads);
    FP_REQUIRE(arch->mlp_dim % 64 == 0 && arch->depth > 0 && arch->patch > 0, "vit_create: bad arch");
    fp_vit* v = new fp_vit();
    v->ctx = ctx;
    v->a = *arch;
    v->blk.resize(arch->depth);
    v->fold.resize(arch->depth);
    v->KP = cdiv(3 * arch->patch * arch->patch, 64) * 64;
    if (hipMalloc((void**)&v->pe_w, (size_t)arch->dim * v->KP * 2) != hipSuccess ||
        hipMalloc((void**)&v->ones, (size_t)arch->dim * 2) != hipSuccess) {
        fp_set_error("vit_create: hipMalloc failed");
        delete v;
        return FP_ERR_HIP;
    }
    FP_HIP(hipMemset(v->pe_w, 0, (size_t)arch->dim * v->KP * 2));
    std::vector<bf16_t> one(arch->dim, f2bf(1.0f));
    FP_HIP(hipMemcpy(v->ones, one.data(), (size_t)arch->dim * 2, hipMemcpyHostToDevice));
    *out = v;
    return FP_OK;
}
extern "C" int fp_vit_destroy(fp_vit* v) {
    if (!v) return FP_OK;
    if (v->pe_w) (void)hipFree(v->pe_w);
    if (v->ones) (void)hipFree(v->ones);
    for (auto& kv : v->pos_cache) (void)hipFree(kv.second);
    for (auto& f : v->fold) {
        if (f.qkvw) (void)hipFree(f.qkvw);
        if (f.fc1w) (void)hipFree(f.fc1w);
        if (f.qkv_cb) (void)hipFree(f.qkv_cb);
        if (f.fc1_cb) (void)hipFree(f.fc1_cb);
    }
    for (auto& p : v->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (v->fold_ev) (void)hipEventDestroy(v->fold_ev);
    delete v;
    return FP_OK;
}

extern "C" int fp_vit_set_weight(fp_vit* v, const char* name, const void* d, size_t numel, void* stream) {
    FP_REQUIRE(v && name && d, "vit_set_weight: null argument");
    const fp_vit_arch& a = v->a;
    const bf16_t* p = (const bf16_t*)d;
    const size_t D = a.dim;
    if (v->folded_upto > 0) v->refold = true;
    v->folded_upto = 0;   // any new tensor invalidates the folded LayerNorm weights (in-place updates of a registered tensor: re-register it)
    auto need = [&](size_t n) -> bool {
        if (numel != n) { fp_set_error("vit_set_weight: %s has %zu elements, expected %zu", name, numel, n); return false; }
        return true;
    };
    std::string s(name);
    if (s == "cls_token") { if (!need(D)) return FP_ERR_INVALID; v->cls = p; return FP_OK; }
    if (s == "pos_embed") {
        if (!need((size_t)(1 + a.pos_grid * a.pos_grid) * D)) return FP_ERR_INVALID;
        v->pos = p;
        for (auto& kv : v->pos_cache) (void)hipFree(kv.second);
        v->pos_cache.clear();
        return FP_OK;
    }
    if (s == "register_tokens") { if (!need((size_t)a.n_reg * D)) return FP_ERR_INVALID; v->reg = p; return FP_OK; }
    if (s == "mask_token") return FP_OK;  // unused at inference
    if (s == "patch_embed.proj.weight") {
        const int K = 3 * a.patch * a.patch;
        if (!need(D * K)) return FP_ERR_INVALID;
        FP_HIP(hipMemcpy2DAsync(v->pe_w, (size_t)v->KP * 2, p, (size_t)K * 2, (size_t)K * 2, D,
                                hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return FP_OK;
    }
    if (s == "patch_embed.proj.bias") { if (!need(D)) return FP_ERR_INVALID; v->pe_b = p; return FP_OK; }
    if (s == "norm.weight") { if (!need(D)) return FP_ERR_INVALID; v->normw = p; return FP_OK; }
    if (s == "norm.bias") { if (!need(D)) return FP_ERR_INVALID; v->normb = p; return FP_OK; }
    int bi = -1;
    char rest[64] = {0};
    if (sscanf(name, "blocks.%d.%63s", &bi, rest) == 2 && bi >= 0 && bi < a.depth) {
        VitBlockW& w = v->blk[bi];
        std::string r(rest);
        const size_t M = a.mlp_dim;
        struct Ent { const char* n; const bf16_t** slot; size_t numel; } tab[] = {
            {"norm1.weight", &w.n1w, D}, {"norm1.bias", &w.n1b, D},
            {"attn.qkv.weight", &w.qkvw, 3 * D * D}, {"attn.qkv.bias", &w.qkvb, 3 * D},
            {"attn.proj.weight", &w.projw, D * D}, {"attn.proj.bias", &w.projb, D},
            {"ls1.gamma", &w.ls1, D}, {"norm2.weight", &w.n2w, D}, {"norm2.bias", &w.n2b, D},
            {"mlp.fc1.weight", &w.fc1w, M * D}, {"mlp.fc1.bias", &w.fc1b, M},
            {"mlp.fc2.weight", &w.fc2w, D * M}, {"mlp.fc2.bias", &w.fc2b, D}, {"ls2.gamma", &w.ls2, D}};
        for (auto& e : tab)
            if (r == e.n) { if (!need(e.numel)) return FP_ERR_INVALID; *e.slot = p; return FP_OK; }
    }
    fp_set_error("vit_set_weight: unknown tensor name '%s'", name);
    return FP_ERR_INVALID;
}

static int vit_pos(fp_vit* v, int gh, int gw, hipStream_t s, const bf16_t** out) {
    const fp_vit_arch& a = v->a;
    if (gh == a.pos_grid && gw == a.pos_grid) { *out = v->pos + a.dim; return FP_OK; }  // no resize needed
    auto key = std::make_pair(gh, gw);
    auto it = v->pos_cache.find(key);
    if (it == v->pos_cache.end()) {
        bf16_t* buf = nullptr;
        FP_HIP(hipMalloc((void**)&buf, (size_t)gh * gw * a.dim * 2));
        int rc = fp_posembed_aa(v->pos + a.dim, buf, a.pos_grid, gh, gw, a.dim, s);
        if (rc) { (void)hipFree(buf); return rc; }
        it = v->pos_cache.emplace(key, buf).first;
    }
    *out = it->second;
    return FP_OK;
}

// W' = W diag(gamma_ln), (colsum, b') for the qkv and fc1 layers of the first L blocks (device kernels on `s`, once per weight load)
static int vit_fold(fp_vit* v, int first, int L, hipStream_t s) {
    const fp_vit_arch& a = v->a;
    const size_t D = a.dim, Mm = a.mlp_dim;
    for (int i = first; i < L; ++i) {
        const VitBlockW& w = v->blk[i];
        fp_vit::VitFold& f = v->fold[i];
        if (!f.qkvw) FP_HIP(hipMalloc((void**)&f.qkvw, 3 * D * D * 2));
        if (!f.qkv_cb) FP_HIP(hipMalloc((void**)&f.qkv_cb, 3 * D * sizeof(uint4)));
        if (!f.fc1w) FP_HIP(hipMalloc((void**)&f.fc1w, Mm * D * 2));
        if (!f.fc1_cb) FP_HIP(hipMalloc((void**)&f.fc1_cb, Mm * sizeof(uint4)));
        int rc;
        // q rows carry log2(e) / sqrt(head_dim): fp_attention_fwd(..., q_prescaled = true) below
        if ((rc = fp_ln_fold(w.qkvw, w.n1w, w.n1b, w.qkvb, f.qkvw, f.qkv_cb, (int)(3 * D), (int)D, (int)D, FP_ATTN_QSCALE, s))) return rc;
        if ((rc = fp_ln_fold(w.fc1w, w.n2w, w.n2b, w.fc1b, f.fc1w, f.fc1_cb, (int)Mm, (int)D, 0, 1.0f, s))) return rc;
    }
    return FP_OK;
}

extern "C" double fp_vit_flops(const fp_vit* v, int B, int H, int W, int layer) {
    const fp_vit_arch& a = v->a;
    const double P = (double)(H / a.patch) * (W / a.patch), N = P + 1 + a.n_reg, D = a.dim;
    const int L = std::min(layer, a.depth);
    const double mlp_ratio = (double)a.mlp_dim / a.dim;
    const double per_block = (8.0 + 4.0 * mlp_ratio) * N * D * D + 4.0 * N * N * D;  // 24 N D^2 + 4 N^2 D for ratio 4
    return B * (L * per_block + 2.0 * P * (3.0 * a.patch * a.patch) * D);
}

namespace {
// Per-kernel-class timing with HIP events on the launch stream.  Events are only RECORDED here (no host sync, so the
// timed region is not perturbed); fp_vit_profile_read() synchronises once and sums the elapsed times.
struct ProfScope {
    fp_vit* v; hipStream_t s; int cls; size_t slot;
    ProfScope(fp_vit* v_, hipStream_t s_, float* acc_) : v(v_), s(s_), slot((size_t)-1) {
        cls = acc_ == &v->ms_gemm ? 0 : (acc_ == &v->ms_attn ? 1 : 2);
        if (!v->prof) return;
        if (v->ev_used == v->ev_pool.size()) {
            fp_vit::EvPair p{};
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
            v->ev_pool.push_back(p);
        }
        slot = v->ev_used++;
        v->ev_pool[slot].cls = cls;
        (void)hipEventRecord(v->ev_pool[slot].a, s);
    }
    ~ProfScope() {
        if (slot != (size_t)-1) (void)hipEventRecord(v->ev_pool[slot].b, s);
    }
};
}  // namespace

extern "C" int fp_vit_forward(fp_vit* v, const void* d_images, int B, int H, int W, int layer, int feature_type,
                              void* d_out, void* stream) {
    FP_REQUIRE(v && d_images && d_out, "vit_forward: null argument");
    const fp_vit_arch& a = v->a;
    hipStream_t s = (hipStream_t)stream;
    FP_REQUIRE(B > 0 && H % a.patch == 0 && W % a.patch == 0, "vit_forward: B=%d H=%d W=%d (patch %d)", B, H, W, a.patch);
    FP_REQUIRE(feature_type >= 0 && feature_type <= 3, "vit_forward: feature_type %d", feature_type);
    FP_REQUIRE(v->cls && v->pos && v->pe_b && v->normw && v->normb && (a.n_reg == 0 || v->reg),
               "vit_forward: embedding / final-norm weights not set");
    // dino.py:18-21 breaks when blk_idx + 1 == layer: a layer outside [1, depth] (too large, zero or negative) never matches,
    // so all blocks run
    const int L = (layer >= 1 && layer <= a.depth) ? layer : a.depth;
    for (int i = 0; i < L; ++i) {
        const VitBlockW& w = v->blk[i];
        FP_REQUIRE(w.n1w && w.n1b && w.qkvw && w.qkvb && w.projw && w.projb && w.n2w && w.n2b && w.fc1w && w.fc1b &&
                       w.fc2w && w.fc2b, "vit_forward: weights of block %d not set", i);
    }
    const int D = a.dim, gh = H / a.patch, gw = W / a.patch, P = gh * gw;
    const int n_tok = P + 1 + a.n_reg;
    const int npad = cdiv(n_tok, 16) * 16;
    const size_t M = (size_t)B * npad;
    FP_REQUIRE(M * (size_t)a.mlp_dim * 2 < 0xffffffffull, "vit_forward: batch too large for 32-bit tile offsets (B=%d)", B);

    bf16_t *A0, *X, *Y, *QK, *Vt, *AO, *H1;
    int rc;
    const int nosplit = v->ctx->opt_row_split == 0;   // fp_ctx_set_option(ctx, "gemm_row_split", 0)
    // LayerNorm 1 / 2 folded into the qkv / fc1 GEMMs (default; fp_ctx_set_option(ctx, "ln_fused", 0) runs the separate kernel)
    const bool lnf = v->ctx->opt_ln_fused != 0 && D % 64 == 0;
    uint4* stat = nullptr;     // per-row init-MFMA records (sigma, -mean splits)
    float2* part = nullptr;
    float* rstd = nullptr;
    if (lnf) {
        // only the blocks this call runs (their weights were checked above): a caller that registered 22 of 24 blocks never
        // touches the other two.  The fold is enqueued on `s`; a later forward on another stream waits for it through the event.
        // Ordering across streams: whatever this call does with the fold buffers — read blocks folded earlier, fold further blocks — comes
        // behind the last fold (the event is re-recorded on `s` only after `s` has waited for it, so waiting for the newest record
        // covers every earlier fold).  A RE-fold (a weight was replaced) overwrites buffers that forwards on other streams may still be
        // reading: the device is drained first — once per weight load, never on the steady path.
        if (v->fold_ev && s != v->fold_stream) FP_HIP(hipStreamWaitEvent(s, v->fold_ev, 0));
        if (v->folded_upto < L) {
            if (v->refold) {
                FP_HIP(hipDeviceSynchronize());
                v->refold = false;
            }
            if ((rc = vit_fold(v, v->folded_upto, L, s))) return rc;
            if (!v->fold_ev) FP_HIP(hipEventCreateWithFlags(&v->fold_ev, hipEventDisableTiming));
            FP_HIP(hipEventRecord(v->fold_ev, s));
            v->fold_stream = s;
            v->folded_upto = L;
        }
        if ((rc = v->ctx->get("vit.ln_stat", M * sizeof(uint4), (void**)&stat))) return rc;
        if ((rc = v->ctx->get("vit.ln_rstd", M * sizeof(float), (void**)&rstd))) return rc;
        if ((rc = v->ctx->get("vit.ln_part", M * (size_t)(D / 64) * sizeof(float2), (void**)&part))) return rc;
    }
    if ((rc = v->ctx->get("vit.im2col", (size_t)B * P * v->KP * 2, (void**)&A0))) return rc;
    if ((rc = v->ctx->get("vit.x", M * D * 2, (void**)&X))) return rc;
    if ((rc = v->ctx->get("vit.y", M * D * 2, (void**)&Y))) return rc;
    if ((rc = v->ctx->get("vit.qk", M * 2 * D * 2, (void**)&QK))) return rc;
    if ((rc = v->ctx->get("vit.vt", M * D * 2 + 256, (void**)&Vt))) return rc;
    if ((rc = v->ctx->get("vit.ao", M * D * 2, (void**)&AO))) return rc;
    if ((rc = v->ctx->get("vit.h1", M * (size_t)a.mlp_dim * 2, (void**)&H1))) return rc;

    const bf16_t* pos_patch;
    if ((rc = vit_pos(v, gh, gw, s, &pos_patch))) return rc;

    // ---- K0-K2: normalise + unfold, patch-embed GEMM scattering into the token buffer ----------
    {
        ProfScope ps(v, s, &v->ms_other);
        if ((rc = fp_im2col_norm((const bf16_t*)d_images, A0, B, H, W, a.patch, v->KP, s))) return rc;
        if ((rc = fp_token_init(X, v->cls, v->pos, v->reg, a.n_reg, B, n_tok, npad, D, s))) return rc;
    }
    {
        ProfScope ps(v, s, &v->ms_gemm);
        FpGemmArgs g{}; g.no_split = nosplit; if ((rc = v->ctx->sk_scratch(g, s))) return rc;
        g.X = A0; g.ldx = v->KP; g.W = v->pe_w; g.ldw = v->KP; g.C = X; g.ldc = D; g.bias = v->pe_b;
        g.M = B * P; g.N = D; g.K = v->KP; g.pos = pos_patch; g.P = P; g.npad = npad; g.tok_off = 1 + a.n_reg;
        if ((rc = fp_gemm_bf16(g, FP_EPI_PATCH, s))) return rc;
        if (v->prof) { v->gemm_flops += 2.0 * g.M * (3.0 * a.patch * a.patch) * D; v->gemm_launches += 1; }
    }
    const int Mi = (int)M;
    const double Malg = (double)B * n_tok;  // algorithmic rows (pad rows are overhead, not counted as work)
    for (int i = 0; i < L; ++i) {
        const VitBlockW& w = v->blk[i];
        const fp_vit::VitFold& f = v->fold[i];
        const bool stats_out = lnf;               // proj feeds LN2 of this block; fc2 feeds LN1 of the next one (not after the last)
        {
            ProfScope ps(v, s, &v->ms_other);
            if (!lnf) { if ((rc = fp_layernorm(X, Y, w.n1w, w.n1b, Mi, D, a.ln_eps, 0, 0, 0, s))) return rc; }
            else if (i == 0) { if ((rc = fp_row_stats(X, stat, rstd, Mi, D, a.ln_eps, s))) return rc; }     // rows from patch-embed + token init
            // i > 0: the previous fc2 left partial sums.  On the small tile tiers the qk launch below finalises them in its prologue (and the
            // V^T launch reads what it wrote); launches that reach the big tier get them from the finalisation kernel
            else if (!fp_gemm_fuses_ln_part(Mi, (int)(2 * D))) { if ((rc = fp_stats_finalize(part, stat, rstd, Mi, D, a.ln_eps, s))) return rc; }
        }
        {
            ProfScope ps(v, s, &v->ms_gemm);
            FpGemmArgs g{}; g.no_split = nosplit; if ((rc = v->ctx->sk_scratch(g, s))) return rc;
            g.X = lnf ? X : Y; g.ldx = D; g.W = lnf ? f.qkvw : w.qkvw; g.ldw = D; g.C = QK; g.ldc = 2 * D; g.bias = w.qkvb;
            g.M = Mi; g.N = 2 * D; g.K = D; g.ln_mfrag = stat; g.ln_rstd = rstd; g.ln_cfrag = f.qkv_cb;
            if (lnf && i > 0 && fp_gemm_fuses_ln_part(Mi, (int)(2 * D))) { g.ln_part = part; g.ln_part_ld = Mi; g.ln_part_nb = (int)(D / 64); g.ln_eps = a.ln_eps; g.ln_inv_d = 1.0f / (float)D; }
            if ((rc = fp_gemm_bf16(g, lnf ? FP_EPI_LN_BIAS : FP_EPI_BIAS, s))) return rc;
            FpGemmArgs gv{}; gv.no_split = nosplit; if ((rc = v->ctx->sk_scratch(gv, s))) return rc;
            gv.X = lnf ? X : Y; gv.ldx = D; gv.W = (lnf ? f.qkvw : w.qkvw) + (size_t)2 * D * D; gv.ldw = D; gv.C = Vt; gv.ldc = 8;
            gv.bias = w.qkvb + 2 * D; gv.M = Mi; gv.N = D; gv.K = D; gv.npad = npad; gv.heads = a.heads;
            gv.ln_mfrag = stat; gv.ln_rstd = rstd; gv.ln_cfrag = lnf ? f.qkv_cb + 2 * D : nullptr;
            if ((rc = fp_gemm_bf16(gv, lnf ? FP_EPI_LN_VT : FP_EPI_VT, s))) return rc;
            if (v->prof) { v->gemm_flops += 2.0 * Malg * 3.0 * D * D; v->gemm_launches += 2; }
        }
        {
            ProfScope ps(v, s, &v->ms_attn);
            if ((rc = fp_attention_fwd(QK, 2 * D, Vt, AO, D, B, a.heads, n_tok, npad, /*q_prescaled=*/lnf, s))) return rc;
        }
        {
            ProfScope ps(v, s, &v->ms_gemm);
            FpGemmArgs g{}; g.no_split = nosplit; if ((rc = v->ctx->sk_scratch(g, s))) return rc;
            g.X = AO; g.ldx = D; g.W = w.projw; g.ldw = D; g.C = X; g.ldc = D; g.bias = w.projb;
            g.gamma = w.ls1 ? w.ls1 : v->ones; g.resid = X; g.ldr = D; g.M = Mi; g.N = D; g.K = D; g.stat_part = part;
            if ((rc = fp_gemm_bf16(g, stats_out ? FP_EPI_LS_RES_STATS : FP_EPI_BIAS_LS_RES, s))) return rc;
            if (v->prof) { v->gemm_flops += 2.0 * Malg * (double)D * D; v->gemm_launches += 1; }
        }
        {
            ProfScope ps(v, s, &v->ms_other);
            if (!lnf) { if ((rc = fp_layernorm(X, Y, w.n2w, w.n2b, Mi, D, a.ln_eps, 0, 0, 0, s))) return rc; }
            else if (!fp_gemm_fuses_ln_part(Mi, (int)a.mlp_dim)) { if ((rc = fp_stats_finalize(part, stat, rstd, Mi, D, a.ln_eps, s))) return rc; }
            // (small tiers: the fc1 launch below finalises proj's partial sums in its prologue)
        }
        {
            ProfScope ps(v, s, &v->ms_gemm);
            FpGemmArgs g{}; g.no_split = nosplit; if ((rc = v->ctx->sk_scratch(g, s))) return rc;
            g.X = lnf ? X : Y; g.ldx = D; g.W = lnf ? f.fc1w : w.fc1w; g.ldw = D; g.C = H1; g.ldc = a.mlp_dim; g.bias = w.fc1b;
            g.M = Mi; g.N = a.mlp_dim; g.K = D; g.ln_mfrag = stat; g.ln_rstd = rstd; g.ln_cfrag = f.fc1_cb;
            if (lnf && fp_gemm_fuses_ln_part(Mi, (int)a.mlp_dim)) { g.ln_part = part; g.ln_part_ld = Mi; g.ln_part_nb = (int)(D / 64); g.ln_eps = a.ln_eps; g.ln_inv_d = 1.0f / (float)D; }
            if ((rc = fp_gemm_bf16(g, lnf ? FP_EPI_LN_GELU : FP_EPI_BIAS_GELU, s))) return rc;
            FpGemmArgs g2{}; g2.no_split = nosplit; if ((rc = v->ctx->sk_scratch(g2, s))) return rc;
            g2.X = H1; g2.ldx = a.mlp_dim; g2.W = w.fc2w; g2.ldw = a.mlp_dim; g2.C = X; g2.ldc = D; g2.bias = w.fc2b;
            g2.gamma = w.ls2 ? w.ls2 : v->ones; g2.resid = X; g2.ldr = D; g2.M = Mi; g2.N = D; g2.K = a.mlp_dim; g2.stat_part = part;
            if ((rc = fp_gemm_bf16(g2, (stats_out && i + 1 < L) ? FP_EPI_LS_RES_STATS : FP_EPI_BIAS_LS_RES, s))) return rc;
            if (v->prof) { v->gemm_flops += 4.0 * Malg * (double)D * a.mlp_dim; v->gemm_launches += 2; }
        }
    }
    // ---- K8: final norm + token slice (dino.py:23-30) ------------------------------------------------
    {
        ProfScope ps(v, s, &v->ms_other);
        int rows_per_b, off;
        if (feature_type == 0) { rows_per_b = 1; off = 0; }
        else if (feature_type == 1) { rows_per_b = a.n_reg; off = 1; }
        else { rows_per_b = P; off = 1 + a.n_reg; }
        if (rows_per_b > 0)
            if ((rc = fp_layernorm(X, (bf16_t*)d_out, v->normw, v->normb, B * rows_per_b, D, a.ln_eps, rows_per_b, npad,
                                   off, s, feature_type == 3)))
                return rc;
    }
    return FP_OK;
}

extern "C" int fp_vit_profile(fp_vit* v, int enable) {
    FP_REQUIRE(v, "vit_profile: null");
    v->prof = enable != 0;
    v->ev_used = 0;
    v->ms_gemm = v->ms_attn = v->ms_other = 0;
    v->gemm_flops = 0;
    v->gemm_launches = 0;
    return FP_OK;
}
extern "C" int fp_vit_profile_read(fp_vit* v, float* g, float* at, float* o, double* fl) {
    FP_REQUIRE(v, "vit_profile_read: null");
    float acc[3] = {0.f, 0.f, 0.f};
    for (size_t i = 0; i < v->ev_used; ++i) {
        FP_HIP(hipEventSynchronize(v->ev_pool[i].b));
        float ms = 0.f;
        FP_HIP(hipEventElapsedTime(&ms, v->ev_pool[i].a, v->ev_pool[i].b));
        acc[v->ev_pool[i].cls] += ms;
    }
    v->ms_gemm = acc[0]; v->ms_attn = acc[1]; v->ms_other = acc[2];
    if (g) *g = v->ms_gemm;
    if (at) *at = v->ms_attn;
    if (o) *o = v->ms_other;
    if (fl) *fl = v->gemm_flops;
    return FP_OK;
}
extern "C" long fp_vit_profile_gemm_launches(const fp_vit* v) { return v ? v->gemm_launches : 0; }

// ---------------------------------------------------------------------------------------------
// FFA / retrieval / template score
extern "C" int fp_ffa(fp_ctx* ctx, const void* d_feats, const uint8_t* d_mask, int B, int gh, int gw, int D, int cell,
                      int normalize, void* d_out_bf16, float* d_out_f32, void* stream) {
    FP_REQUIRE(ctx && d_feats && d_mask && (d_out_bf16 || d_out_f32), "ffa: null argument");
    if (B == 0) return FP_OK;
    hipStream_t s = (hipStream_t)stream;
    bf16_t* tmp = (bf16_t*)d_out_bf16;
    int rc;
    if (normalize || !tmp)
        if ((rc = ctx->get("ffa.tmp", (size_t)B * D * 2, (void**)&tmp))) return rc;
    uint8_t* pm = nullptr;
    if (cell > 1 && (rc = ctx->get("ffa.pm", (size_t)B * gh * gw, (void**)&pm))) return rc;
    if (normalize) FP_REQUIRE(d_out_bf16, "ffa: normalize needs a bf16 output");
    // up to FFA_FUSED_MAX crops (the queries of an image / a frame window) the masked-mean kernel normalises the rows itself (its last
    // column-slab workgroup per crop); beyond (bank building) the row kernel does
    constexpr int FFA_FUSED_MAX = 16;
    int* arrive = nullptr;
    if (normalize && B <= FFA_FUSED_MAX && D % 8 == 0 && (((uintptr_t)tmp | (uintptr_t)d_out_bf16) & 15) == 0) {
        if (!ctx->ffa_arrive) {
            FP_HIP(hipMalloc((void**)&ctx->ffa_arrive, FFA_FUSED_MAX * sizeof(int)));
            FP_HIP(hipMemsetAsync(ctx->ffa_arrive, 0, FFA_FUSED_MAX * sizeof(int), s));
        }
        arrive = ctx->ffa_arrive;
    }
    if ((rc = fp_ffa_pool((const bf16_t*)d_feats, d_mask, tmp, normalize ? nullptr : d_out_f32, B, gh * gw, D, gh, gw,
                          cell, pm, s, arrive ? (bf16_t*)d_out_bf16 : nullptr, arrive)))
        return rc;
    if (normalize && !arrive)
        if ((rc = fp_l2norm_rows(tmp, (bf16_t*)d_out_bf16, B, D, s))) return rc;
    return FP_OK;
}

extern "C" int fp_l2_normalize(fp_ctx* ctx, const void* x, int rows, int D, void* y, void* stream) {
    FP_REQUIRE(ctx && x && y, "l2_normalize: null argument");
    return fp_l2norm_rows((const bf16_t*)x, (bf16_t*)y, rows, D, (hipStream_t)stream);
}

extern "C" int fp_bank_prepare(fp_ctx* ctx, const float* d_bank_f32, int N, int D, void* d_bank_bf16, void* stream) {
    FP_REQUIRE(ctx && d_bank_f32 && d_bank_bf16 && N > 0, "bank_prepare: null argument");
    hipStream_t s = (hipStream_t)stream;
    bf16_t* tmp;
    int rc;
    if ((rc = ctx->get("bank.cast", (size_t)N * D * 2, (void**)&tmp))) return rc;
    if ((rc = fp_cast_f32_bf16(d_bank_f32, tmp, (size_t)N * D, s))) return rc;
    return fp_l2norm_rows(tmp, (bf16_t*)d_bank_bf16, N, D, s);
}

extern "C" int fp_bank_topk(fp_ctx* ctx, const void* d_bank, int N, int D, const void* d_queries, int Q, int k,
                            int idx_offset, float* d_out_scores, int32_t* d_out_idx, void* stream) {
    FP_REQUIRE(ctx && d_bank && d_queries && d_out_scores && d_out_idx, "bank_topk: null argument");
    if (Q == 0) return FP_OK;
    hipStream_t s = (hipStream_t)stream;
    uint16_t* keys;
    int rc;
    const int ldk = (N + 7) & ~7;   // key rows padded to 16 bytes (vector staging in the select kernel)
    if ((rc = ctx->get("topk.keys", (size_t)Q * ldk * 2, (void**)&keys))) return rc;
    if ((rc = fp_bank_scan((const bf16_t*)d_bank, (const bf16_t*)d_queries, keys, ldk, N, D, Q, s))) return rc;
    return fp_topk_select(keys, ldk, N, Q, k, idx_offset, d_out_scores, d_out_idx, s);
}

extern "C" int fp_topk_merge(fp_ctx* ctx, const float* cs, const int32_t* ci, int Q, int C, int k, float* os,
                             int32_t* oi, void* stream) {
    FP_REQUIRE(ctx && cs && ci && os && oi, "topk_merge: null argument");
    if (Q == 0) return FP_OK;
    return fp_topk_merge_launch(cs, ci, Q, C, k, os, oi, (hipStream_t)stream);
}

extern "C" int fp_rerank_views(fp_ctx* ctx, const void* d_views, const int32_t* d_offsets, const int32_t* d_cand,
                               const void* d_queries, int Q, int C, int D, int k, float* d_out, void* stream) {
    FP_REQUIRE(ctx && d_views && d_offsets && d_cand && d_queries && d_out, "rerank_views: null argument");
    if (Q == 0 || C == 0) return FP_OK;
    return fp_rerank_views_launch((const bf16_t*)d_views, d_offsets, d_cand, (const bf16_t*)d_queries, d_out, Q, C, D, k,
                                  (hipStream_t)stream);
}

extern "C" int fp_template_score(fp_ctx* ctx, const void* d_tmpl, const void* d_query, const float* d_weights, int T,
                                 int P, int D, float* d_scores, void* stream) {
    FP_REQUIRE(ctx && d_tmpl && d_query && d_scores, "template_score: null argument");
    if (T == 0) return FP_OK;
    float* dots;
    int rc;
    if ((rc = ctx->get("tmpl.dots", (size_t)T * P * 4, (void**)&dots))) return rc;
    return fp_template_score_launch((const bf16_t*)d_tmpl, (const bf16_t*)d_query, d_weights, dots, d_scores, T, P, D, 0,
                                    (hipStream_t)stream);
}

extern "C" int fp_template_score_normed(fp_ctx* ctx, const void* d_tmpl_normed, const void* d_query, const float* d_weights,
                                        int T, int P, int D, float* d_scores, void* stream) {
    FP_REQUIRE(ctx && d_tmpl_normed && d_query && d_scores, "template_score_normed: null argument");
    if (T == 0) return FP_OK;
    float* dots;
    int rc;
    if ((rc = ctx->get("tmpl.dots", (size_t)T * P * 4, (void**)&dots))) return rc;
    return fp_template_score_launch((const bf16_t*)d_tmpl_normed, (const bf16_t*)d_query, d_weights, dots, d_scores, T, P, D, 1,
                                    (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// kernel-level entry points
extern "C" int fp_op_gemm(fp_ctx* ctx, const void* X, int ldx, const void* W, int ldw, void* Cc, int ldc, const void* bias,
                          const void* gamma, const void* resid, int ldr, int M, int N, int K, int epi, void* stream) {
    FP_REQUIRE(ctx && X && W && Cc && bias, "op_gemm: null argument");
    FP_REQUIRE(epi >= 0 && epi <= 2, "op_gemm: epi %d (0..2)", epi);
    FpGemmArgs g{};
    g.X = (const bf16_t*)X; g.ldx = ldx; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = (bf16_t*)Cc; g.ldc = ldc;
    g.bias = (const bf16_t*)bias; g.gamma = (const bf16_t*)gamma; g.resid = (const bf16_t*)resid; g.ldr = ldr;
    g.M = M; g.N = N; g.K = K; g.no_split = ctx->opt_row_split == 0;
    { const int rc = ctx->sk_scratch(g, (hipStream_t)stream); if (rc) return rc; }
    return fp_gemm_bf16(g, epi, (hipStream_t)stream);
}
extern "C" int fp_op_gemm_vt(fp_ctx* ctx, const void* X, int ldx, const void* W, int ldw, void* Vt, const void* bias, int M, int N,
                             int K, int npad, int heads, void* stream) {
    FP_REQUIRE(ctx && X && W && Vt, "op_gemm_vt: null argument");
    FpGemmArgs g{};
    g.X = (const bf16_t*)X; g.ldx = ldx; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = (bf16_t*)Vt; g.ldc = 8;
    g.bias = (const bf16_t*)bias; g.M = M; g.N = N; g.K = K; g.npad = npad; g.heads = heads;
    { const int rc = ctx->sk_scratch(g, (hipStream_t)stream); if (rc) return rc; }
    return fp_gemm_bf16(g, FP_EPI_VT, (hipStream_t)stream);
}
// LayerNorm folded into a linear layer, as fp_vit_forward runs LN1 -> qkv and LN2 -> fc1 (kernel-level entry for the tests):
// fold W' / (colsum, b') -> row statistics of X -> the LN-folded GEMM.  mode 0: bias, 1: bias + GELU, 2: transposed V store.
extern "C" int fp_op_ln_linear(fp_ctx* ctx, const void* X, int M, int K, const void* g_ln, const void* b_ln, float eps,
                               const void* W, int N, const void* bias, int mode, int npad, int heads, int n_scaled, float row_scale,
                               void* out, void* stream) {
    FP_REQUIRE(ctx && X && g_ln && b_ln && W && bias && out, "op_ln_linear: null argument");
    FP_REQUIRE(mode >= 0 && mode <= 2, "op_ln_linear: mode %d (0..2)", mode);
    hipStream_t s = (hipStream_t)stream;
    bf16_t* Wf;
    uint4 *cb, *stat;
    float* rstd;
    int rc;
    if ((rc = ctx->get("op.ln_wf", (size_t)N * K * 2, (void**)&Wf))) return rc;
    if ((rc = ctx->get("op.ln_cb", (size_t)N * sizeof(uint4), (void**)&cb))) return rc;
    if ((rc = ctx->get("op.ln_stat", (size_t)M * sizeof(uint4), (void**)&stat))) return rc;
    if ((rc = ctx->get("op.ln_rstd", (size_t)M * sizeof(float), (void**)&rstd))) return rc;
    FP_REQUIRE(n_scaled >= 0 && n_scaled <= N, "op_ln_linear: n_scaled %d outside [0, N]", n_scaled);
    if ((rc = fp_ln_fold((const bf16_t*)W, (const bf16_t*)g_ln, (const bf16_t*)b_ln, (const bf16_t*)bias, Wf, cb, N, K, n_scaled, row_scale, s))) return rc;
    if ((rc = fp_row_stats((const bf16_t*)X, stat, rstd, M, K, eps, s))) return rc;
    FpGemmArgs g{};
    g.X = (const bf16_t*)X; g.ldx = K; g.W = Wf; g.ldw = K; g.C = (bf16_t*)out; g.ldc = mode == 2 ? 8 : N;
    g.bias = (const bf16_t*)bias; g.M = M; g.N = N; g.K = K; g.npad = npad; g.heads = heads; g.ln_mfrag = stat; g.ln_rstd = rstd; g.ln_cfrag = cb;
    g.no_split = ctx->opt_row_split == 0;
    if ((rc = ctx->sk_scratch(g, s))) return rc;
    return fp_gemm_bf16(g, mode == 0 ? FP_EPI_LN_BIAS : (mode == 1 ? FP_EPI_LN_GELU : FP_EPI_LN_VT), s);
}
// LayerScale + residual GEMM that also emits the row statistics of its OUTPUT (what the next LN-folded GEMM consumes):
// d_stat f32 [M,2] = (mean, rstd) of the bf16 rows of C, from the epilogue's per-64-column partial sums.
extern "C" int fp_op_gemm_stats(fp_ctx* ctx, const void* X, int ldx, const void* W, int ldw, void* Cc, int ldc, const void* bias,
                                const void* gamma, const void* resid, int ldr, int M, int N, int K, float eps, float* d_stat,
                                void* stream) {
    FP_REQUIRE(ctx && X && W && Cc && bias && gamma && resid && d_stat, "op_gemm_stats: null argument");
    float2* part;
    uint4* ms;
    float* rstd;
    int rc;
    if ((rc = ctx->get("op.ln_part", (size_t)M * (N / 64) * sizeof(float2), (void**)&part))) return rc;
    if ((rc = ctx->get("op.ln_stat", (size_t)M * sizeof(uint4), (void**)&ms))) return rc;
    if ((rc = ctx->get("op.ln_rstd", (size_t)M * sizeof(float), (void**)&rstd))) return rc;
    FpGemmArgs g{};
    g.X = (const bf16_t*)X; g.ldx = ldx; g.W = (const bf16_t*)W; g.ldw = ldw; g.C = (bf16_t*)Cc; g.ldc = ldc;
    g.bias = (const bf16_t*)bias; g.gamma = (const bf16_t*)gamma; g.resid = (const bf16_t*)resid; g.ldr = ldr;
    g.M = M; g.N = N; g.K = K; g.stat_part = part; g.no_split = ctx->opt_row_split == 0;
    if ((rc = ctx->sk_scratch(g, (hipStream_t)stream))) return rc;
    if ((rc = fp_gemm_bf16(g, FP_EPI_LS_RES_STATS, (hipStream_t)stream))) return rc;
    if ((rc = fp_stats_finalize(part, ms, rstd, M, N, eps, (hipStream_t)stream))) return rc;
    // d_stat [M,6] = the row record's 4 words {sh|sl, sh|-mh, -ml|-mh, 0} reinterpreted as floats are NOT meaningful: hand back the raw
    // record words (4 x u32 as f32 bit patterns) followed by (rstd, 0); the Python wrapper decodes mean = -(mh + ml), sigma = sh + sl
    FP_HIP(hipMemcpy2DAsync(d_stat, 24, ms, 16, 16, M, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    FP_HIP(hipMemcpy2DAsync(d_stat + 4, 24, rstd, 4, 4, M, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FP_OK;
}
// y = bf16(gelu_erf(x)) elementwise with the direct fp32 expression — the DEFINITION the fc1 epilogue's table is filled from; the
// tests compare the table-GELU GEMM against it on all 65 536 bf16 inputs
extern "C" int fp_op_gelu(const void* x, void* y, size_t n, void* stream) {
    FP_REQUIRE(x && y, "op_gelu: null argument");
    return fp_gemm_gelu_direct((const bf16_t*)x, (bf16_t*)y, n, (hipStream_t)stream);
}
extern "C" int fp_op_attention(const void* QK, int ldqk, const void* Vt, void* O, int ldo, int B, int H, int n_tok,
                               int npad, int q_prescaled, void* stream) {
    FP_REQUIRE(QK && Vt && O, "op_attention: null argument");
    return fp_attention_fwd((const bf16_t*)QK, ldqk, (const bf16_t*)Vt, (bf16_t*)O, ldo, B, H, n_tok, npad, q_prescaled != 0,
                            (hipStream_t)stream);
}
extern "C" int fp_op_im2col_norm(const void* img, void* A, int B, int H, int W, int ps, int KP, void* stream) {
    FP_REQUIRE(img && A && B > 0, "op_im2col_norm: null argument / empty batch");
    return fp_im2col_norm((const bf16_t*)img, (bf16_t*)A, B, H, W, ps, KP, (hipStream_t)stream);
}
extern "C" int fp_op_layernorm(const void* X, void* Y, const void* g, const void* b, int rows, int D, float eps,
                               void* stream) {
    FP_REQUIRE(X && Y && g && b, "op_layernorm: null argument");
    return fp_layernorm((const bf16_t*)X, (bf16_t*)Y, (const bf16_t*)g, (const bf16_t*)b, rows, D, eps, 0, 0, 0,
                        (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// timers
struct FpTimer { hipEvent_t a, b; };
extern "C" int fp_timer_create(void** out) {
    FP_REQUIRE(out, "timer_create: null");
    FpTimer* t = new FpTimer();
    FP_HIP(hipEventCreate(&t->a));
    FP_HIP(hipEventCreate(&t->b));
    *out = t;
    return FP_OK;
}
extern "C" int fp_timer_start(void* t, void* s) { FP_HIP(hipEventRecord(((FpTimer*)t)->a, (hipStream_t)s)); return FP_OK; }
extern "C" int fp_timer_stop(void* t, void* s) { FP_HIP(hipEventRecord(((FpTimer*)t)->b, (hipStream_t)s)); return FP_OK; }
extern "C" int fp_timer_elapsed_ms(void* t, float* ms) {
    FP_HIP(hipEventSynchronize(((FpTimer*)t)->b));
    FP_HIP(hipEventElapsedTime(ms, ((FpTimer*)t)->a, ((FpTimer*)t)->b));
    return FP_OK;
}
extern "C" int fp_timer_destroy(void* t) {
    if (!t) return FP_OK;
    (void)hipEventDestroy(((FpTimer*)t)->a);
    (void)hipEventDestroy(((FpTimer*)t)->b);
    delete (FpTimer*)t;
    return FP_OK;
}
