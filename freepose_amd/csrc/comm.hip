// C-ABI communication entry points (fp_comm_init / fp_allgather_topk / fp_allgather_poses): RCCL over xGMI for
// hosts that do not go through torch.distributed.  One communicator per fp_ctx (= per process / GPU).  librccl is opened
// lazily with dlopen(RTLD_LOCAL) on the first fp_comm_* call: a PyTorch process — which carries its own RCCL and talks to it
// through torch.distributed (freepose_amd/parallel.py) — never loads a second copy unless it asks for these entry points.
//
// The path's only collectives are tiny all-gathers: Q*k (score, index) candidate pairs per rank for bank-row
// sharding, and fixed-size result rows (poses) for proposal / frame / object sharding.  Payloads are <= 100 KB: latency-bound,
// one call per batch.
#include <dlfcn.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/freepose_hip.h"
#include "internal.h"

namespace {

// the handful of RCCL symbols used, with their published C signatures (rccl.h): resolved with dlsym
typedef struct { char internal[128]; } fp_nccl_uid;
typedef void* fp_nccl_comm;
typedef int (*pfn_get_uid)(fp_nccl_uid*);
typedef int (*pfn_init_rank)(fp_nccl_comm*, int, fp_nccl_uid, int);
typedef int (*pfn_destroy)(fp_nccl_comm);
typedef int (*pfn_allgather)(const void*, void*, size_t, int /*ncclDataType_t*/, fp_nccl_comm, hipStream_t);
typedef const char* (*pfn_errstr)(int);

struct Rccl {
    void* lib = nullptr;
    pfn_get_uid get_uid = nullptr;
    pfn_init_rank init_rank = nullptr;
    pfn_destroy destroy = nullptr;
    pfn_allgather allgather = nullptr;
    pfn_errstr errstr = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return FP_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    FP_REQUIRE(h, "comm: cannot open librccl (%s)", dlerror());
    g_rccl.get_uid = (pfn_get_uid)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (pfn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.destroy = (pfn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.allgather = (pfn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.errstr = (pfn_errstr)dlsym(h, "ncclGetErrorString");
    FP_REQUIRE(g_rccl.get_uid && g_rccl.init_rank && g_rccl.destroy && g_rccl.allgather, "comm: librccl lacks a required symbol");
    g_rccl.lib = h;
    return FP_OK;
}

#define FP_NCCL(call)                                                                                             \
    do {                                                                                                          \
        int r__ = (call);                                                                                         \
        if (r__ != 0) {                                                                                           \
            fp_set_error("%s failed: %s", #call, g_rccl.errstr ? g_rccl.errstr(r__) : "rccl error");             \
            return FP_ERR_HIP;                                                                                    \
        }                                                                                                         \
    } while (0)

// [R][Q][k] (rank-major, what the all-gather produces) -> [Q][R*k] (what fp_topk_merge consumes)
__global__ void regroup_kernel(const float* __restrict__ s_in, const int32_t* __restrict__ i_in, float* __restrict__ s_out,
                               int32_t* __restrict__ i_out, int R, int Q, int k) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R * Q * k) return;
    const int r = t / (Q * k), rem = t - r * (Q * k), q = rem / k, j = rem - q * k;
    const size_t o = (size_t)q * R * k + (size_t)r * k + j;
    s_out[o] = s_in[t];
    i_out[o] = i_in[t];
}

}  // namespace

extern "C" int fp_comm_unique_id(void* out_id128) {
    FP_REQUIRE(out_id128, "comm_unique_id: null argument");
    int rc = rccl_load();
    if (rc) return rc;
    fp_nccl_uid id;
    FP_NCCL(g_rccl.get_uid(&id));
    memcpy(out_id128, &id, sizeof(id));
    return FP_OK;
}

extern "C" int fp_comm_init(fp_ctx* ctx, int nranks, int rank, const void* unique_id128) {
    FP_REQUIRE(ctx && unique_id128 && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument");
    FP_REQUIRE(!ctx->comm, "comm_init: context already has a communicator");
    int rc = rccl_load();
    if (rc) return rc;
    fp_nccl_uid id;
    memcpy(&id, unique_id128, sizeof(id));
    bool blank = true;
    for (size_t k = 0; k < sizeof(id); ++k) blank = blank && id.internal[k] == 0;
    FP_REQUIRE(!blank, "comm_init: the unique id is all zero (fp_comm_unique_id of rank 0 never reached this rank)");
    // ncclCommInitRank is a rendezvous of `nranks` ranks and waits for ever when they never all arrive — ranks that disagree on
    // nranks or hold different ids, a rank that died before it got here.  It runs on a helper thread; after `comm_timeout_s` seconds
    // (fp_ctx_set_option, default 180) the call returns FP_ERR_STATE with a message instead of hanging the job (the helper stays
    // blocked inside RCCL and is abandoned: the process is expected to exit on this error).
    struct Rendezvous {
        std::mutex m;
        std::condition_variable cv;
        bool done = false;
        int rc = 0, hip_rc = 0;
        fp_nccl_comm comm = nullptr;
    };
    auto st = std::make_shared<Rendezvous>();
    const int device = ctx->device;
    std::thread([st, device, nranks, rank, id]() {
        int hip_rc = (int)hipSetDevice(device), rc = 0;
        fp_nccl_comm c = nullptr;
        if (hip_rc == 0) rc = g_rccl.init_rank(&c, nranks, id, rank);
        std::lock_guard<std::mutex> lk(st->m);
        st->rc = rc; st->hip_rc = hip_rc; st->comm = c; st->done = true;
        st->cv.notify_all();
    }).detach();
    const int limit_s = ctx->opt_comm_timeout_s > 0 ? ctx->opt_comm_timeout_s : 180;
    {
        std::unique_lock<std::mutex> lk(st->m);
        if (!st->cv.wait_for(lk, std::chrono::seconds(limit_s), [&] { return st->done; })) {
            fp_set_error("comm_init: the rendezvous of %d ranks (this is rank %d) did not complete within %d s — do all ranks pass the same "
                         "nranks and the unique id of rank 0, and did every rank get here?", nranks, rank, limit_s);
            return FP_ERR_STATE;
        }
    }
    FP_REQUIRE(st->hip_rc == 0, "comm_init: hipSetDevice(%d) failed on the rendezvous thread", device);
    if (st->rc != 0) {
        fp_set_error("ncclCommInitRank failed: %s", g_rccl.errstr ? g_rccl.errstr(st->rc) : "rccl error");
        return FP_ERR_HIP;
    }
    FP_HIP(hipSetDevice(ctx->device));
    fp_nccl_comm c = st->comm;
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_size = nranks;
    return FP_OK;
}

extern "C" int fp_comm_destroy(fp_ctx* ctx) {
    if (!ctx || !ctx->comm) return FP_OK;
    FP_NCCL(g_rccl.destroy((fp_nccl_comm)ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_size = 1;
    ctx->comm_rank = 0;
    return FP_OK;
}

extern "C" int fp_comm_size(const fp_ctx* ctx) { return ctx && ctx->comm ? ctx->comm_size : 1; }
extern "C" int fp_comm_rank(const fp_ctx* ctx) { return ctx && ctx->comm ? ctx->comm_rank : 0; }

extern "C" int fp_allgather_bytes(fp_ctx* ctx, const void* d_send, size_t bytes, void* d_recv, void* stream) {
    FP_REQUIRE(ctx, "allgather_bytes: null context");
    if (bytes == 0) return FP_OK;                     // nothing to exchange (every rank passes the same size): pointers may be null
    FP_REQUIRE(d_send && d_recv, "allgather_bytes: null buffer");
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->comm) {   // single rank: the gather is a copy
        FP_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, s));
        return FP_OK;
    }
    FP_NCCL(g_rccl.allgather(d_send, d_recv, bytes, /*ncclUint8*/ 1, (fp_nccl_comm)ctx->comm, s));
    return FP_OK;
}

extern "C" int fp_allgather_topk(fp_ctx* ctx, const float* d_scores, const int32_t* d_idx, int Q, int k, int k_out,
                                 float* d_out_scores, int32_t* d_out_idx, void* stream) {
    FP_REQUIRE(ctx && d_scores && d_idx && d_out_scores && d_out_idx && Q > 0 && k > 0 && k_out > 0,
               "allgather_topk: bad argument");
    const int R = ctx->comm ? ctx->comm_size : 1;
    FP_REQUIRE(k_out <= R * k, "allgather_topk: k_out=%d exceeds the %d gathered candidates", k_out, R * k);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)Q * k;
    float *gs, *ms;
    int32_t *gi, *mi;
    int rc;
    if ((rc = ctx->get("comm.gs", (size_t)R * n * 4, (void**)&gs))) return rc;
    if ((rc = ctx->get("comm.gi", (size_t)R * n * 4, (void**)&gi))) return rc;
    if ((rc = ctx->get("comm.ms", (size_t)R * n * 4, (void**)&ms))) return rc;
    if ((rc = ctx->get("comm.mi", (size_t)R * n * 4, (void**)&mi))) return rc;
    if ((rc = fp_allgather_bytes(ctx, d_scores, n * 4, gs, stream))) return rc;
    if ((rc = fp_allgather_bytes(ctx, d_idx, n * 4, gi, stream))) return rc;
    const int total = R * Q * k;
    hipLaunchKernelGGL(regroup_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, gs, gi, ms, mi, R, Q, k);
    FP_LAUNCH_CHECK();
    return fp_topk_merge_launch(ms, mi, Q, R * k, k_out, d_out_scores, d_out_idx, s);
}

extern "C" int fp_allgather_poses(fp_ctx* ctx, const double* d_rows, int n_rows, int row_len, double* d_out, void* stream) {
    FP_REQUIRE(ctx && d_out && n_rows >= 0 && row_len > 0 && (n_rows == 0 || d_rows), "allgather_poses: bad argument");
    return fp_allgather_bytes(ctx, d_rows, (size_t)n_rows * row_len * sizeof(double), d_out, stream);
}
