/* fp_oracle.c — CPU restatement of FreePose's per-proposal hot path (scalar C, one thread).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under freepose_amd/ imports, links or executes this file; it is used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / reported CPU baseline.
 *
 * Each function names the reference lines (ponimatkin/freepose) whose behaviour it restates.  Where the
 * reference leaves floating-point evaluation order to a library (torch matmul / mean on bf16 tensors) this
 * file FIXES one order — "dot64", below — and the HIP kernels implement the same order, so indices and
 * scores compare bit for bit.  The reference's own bf16 rounding points are kept.
 *
 * Pinning status (details: DESIGN.md §5):
 *   pinned by golden vectors produced by the reference's own code here (tests/golden, oracle/gen_golden*.py): crop/resize/pad,
 *     Proposals, RLE, rotation grids, depth->cloud extents and z, geodesic distance, the estimator's scores, FFA / bank scores
 *     / top-100 as the reference's torch expressions evaluate them, the TrackingRefiner box / intrinsics / threshold arithmetic;
 *   PARITY UNPINNED (dependency not installable here, no reference fixtures): the rasteriser vs pyrender/OpenGL (conventions
 *     restated from renderer.py, held by known-answer tests) and fpo_roi_align vs torchvision.ops.roi_align (restated from
 *     the published operator, held by known-answer tests).
 *
 * Build: gcc -O2 -std=c11 -fPIC -shared -ffp-contract=off -o libfp_oracle.so fp_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint16_t bf16;

static inline float bf2f(bf16 h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static inline bf16 f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16)(u >> 16);
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

/* ---- dot64: the canonical 64-lane summation order ------------------------------------------------
 * lane l owns elements (c*64 + l)*8 + e ; per-lane fmaf chain over (c,e) ascending ; then a xor-butterfly
 * over offsets 32,16,8,4,2,1 (every lane adds its partner: all lanes end with the same value).          */
static float butterfly64(float* p) {
    float t[64];
    for (int off = 32; off > 0; off >>= 1) {
        for (int l = 0; l < 64; ++l) t[l] = p[l] + p[l ^ off];
        memcpy(p, t, sizeof(t));
    }
    return p[0];
}
static float dot64_bf(const bf16* a, const bf16* b, int D) {
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float acc = 0.f;
        for (int base = l * 8; base < D; base += 512)
            for (int e = 0; e < 8; ++e) acc = fmaf(bf2f(a[base + e]), bf2f(b[base + e]), acc);
        p[l] = acc;
    }
    return butterfly64(p);
}
static float dot64_ff(const float* a, const float* b, int D) {
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float acc = 0.f;
        for (int base = l * 8; base < D; base += 512)
            for (int e = 0; e < 8; ++e) acc = fmaf(a[base + e], b[base + e], acc);
        p[l] = acc;
    }
    return butterfly64(p);
}

/* F.normalize(x, dim=-1) on a bf16 row (scripts/extract_proposals_ground.py:41,134; pose_estimator.py:88):
 * n = bf16(sqrt(sum x^2)), y = bf16(x / max(n, 1e-12)) */
void fpo_l2norm_rows(const bf16* x, bf16* y, int rows, int D) {
    for (int r = 0; r < rows; ++r) {
        const bf16* xr = x + (size_t)r * D;
        float n = rbf(sqrtf(dot64_bf(xr, xr, D)));
        if (n < 1e-12f) n = 1e-12f;
        for (int i = 0; i < D; ++i) y[(size_t)r * D + i] = f2bf(bf2f(xr[i]) / n);
    }
}

/* bank prep: fp32 -> bf16 -> normalise (extract_proposals_ground.py:39-41) */
void fpo_bank_prepare(const float* bank, bf16* out, int N, int D) {
    bf16* tmp = (bf16*)malloc((size_t)N * D * 2);
    for (size_t i = 0; i < (size_t)N * D; ++i) tmp[i] = f2bf(bank[i]);
    fpo_l2norm_rows(tmp, out, N, D);
    free(tmp);
}

/* scores = bf16(bank @ q).float()  (extract_proposals_ground.py:137) */
void fpo_bank_scores(const bf16* bank, const bf16* q, float* scores, int N, int D) {
    for (int r = 0; r < N; ++r) scores[r] = rbf(dot64_bf(bank + (size_t)r * D, q, D));
}

/* top-k with the canonical order (score desc, index asc); extract_proposals_ground.py:140 (torch.topk,
 * whose tie order is unspecified). */
typedef struct { float s; int i; } cand_t;
static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
void fpo_topk(const float* scores, int N, int k, int idx_offset, float* out_s, int32_t* out_i) {
    cand_t* c = (cand_t*)malloc((size_t)N * sizeof(cand_t));
    for (int r = 0; r < N; ++r) { c[r].s = scores[r]; c[r].i = r; }
    qsort(c, N, sizeof(cand_t), cand_cmp);
    for (int j = 0; j < k; ++j) { out_s[j] = c[j].s; out_i[j] = c[j].i + idx_offset; }
    free(c);
}
void fpo_bank_topk(const bf16* bank, int N, int D, const bf16* queries, int Q, int k, int idx_offset, float* out_s,
                   int32_t* out_i) {
    float* sc = (float*)malloc((size_t)N * 4);
    for (int q = 0; q < Q; ++q) {
        fpo_bank_scores(bank, queries + (size_t)q * D, sc, N, D);
        fpo_topk(sc, N, k, idx_offset, out_s + (size_t)q * k, out_i + (size_t)q * k);
    }
    free(sc);
}
/* merge of per-shard candidate lists: same ordering on (score, global index) */
void fpo_topk_merge(const float* cs, const int32_t* ci, int C, int k, float* out_s, int32_t* out_i) {
    cand_t* c = (cand_t*)malloc((size_t)C * sizeof(cand_t));
    for (int r = 0; r < C; ++r) { c[r].s = cs[r]; c[r].i = ci[r]; }
    qsort(c, C, sizeof(cand_t), cand_cmp);
    for (int j = 0; j < k; ++j) { out_s[j] = c[j].s; out_i[j] = c[j].i; }
    free(c);
}

/* FFA (scripts/extract_retrieval_features.py:51-57; extract_proposals_ground.py:129-134):
 * mask -> cv2.resize(INTER_AREA) > 0 == any pixel of the cell x cell block; feat[mask].mean(0) in bf16: fp32 accumulate, divide,
 * round.  torch leaves the order of that sum unspecified; the canonical order (round 6, shared with csrc/vit_misc.hip ffa_kernel) is
 * BLOCKED: the P patches are cut into 32 blocks of BL = ceil(P / 32) consecutive patch indices, the masked rows of a block are added
 * in ascending patch order to an accumulator that starts at 0, the 32 block sums are added in ascending block order
 * (((s0 + s1) + s2) + ...).  out_bf may be NULL; out_f32 gets the bf16 value widened (the .float() of :57). */
#define FPO_FFA_NB 32
void fpo_ffa(const bf16* feats, const uint8_t* mask, int B, int gh, int gw, int D, int cell, bf16* out_bf,
             float* out_f32) {
    const int P = gh * gw, Wm = gw * cell;
    uint8_t* pm = (uint8_t*)malloc(P);
    float* acc = (float*)malloc((size_t)D * 4);
    float* blk = (float*)malloc((size_t)D * 4);
    for (int b = 0; b < B; ++b) {
        int cnt = 0;
        for (int p = 0; p < P; ++p) {
            const int py = p / gw, px = p % gw;
            int any = 0;
            for (int dy = 0; dy < cell; ++dy)
                for (int dx = 0; dx < cell; ++dx)
                    any |= mask[(size_t)b * gh * cell * Wm + (size_t)(py * cell + dy) * Wm + px * cell + dx];
            pm[p] = any ? 1 : 0;
            cnt += pm[p];
        }
        const int BL = (P + FPO_FFA_NB - 1) / FPO_FFA_NB;
        for (int j = 0; j < FPO_FFA_NB; ++j) {
            const int p_lo = j * BL, p_hi = (p_lo + BL < P) ? p_lo + BL : P;
            for (int d = 0; d < D; ++d) blk[d] = 0.f;
            for (int p = p_lo; p < p_hi; ++p)
                if (pm[p])
                    for (int d = 0; d < D; ++d) blk[d] += bf2f(feats[((size_t)b * P + p) * D + d]);
            if (j == 0) for (int d = 0; d < D; ++d) acc[d] = blk[d];
            else for (int d = 0; d < D; ++d) acc[d] += blk[d];
        }
        for (int d = 0; d < D; ++d) {
            const float m = acc[d] / (float)cnt;
            if (out_bf) out_bf[(size_t)b * D + d] = f2bf(m);
            if (out_f32) out_f32[(size_t)b * D + d] = rbf(m);
        }
    }
    free(pm); free(acc); free(blk);
}

/* patchwise template score (src/pipeline/estimators/pose_estimator.py:85-88; online_pose_estimator.py:68-79):
 * einsum(F.normalize(T), q, 'b n d, b n d -> b n').mean(-1), all tensors bf16.  q is used as given.
 * weights != NULL: (scores*masks).sum(-1)/masks.sum(-1) in fp32 (online :69-74). */
void fpo_template_score(const bf16* tmpl, const bf16* q, const float* weights, int T, int P, int D, float* scores) {
    float* tn = (float*)malloc((size_t)D * 4);
    float* qf = (float*)malloc((size_t)D * 4);
    float* dots = (float*)malloc((size_t)P * 4);
    for (int t = 0; t < T; ++t) {
        for (int p = 0; p < P; ++p) {
            const bf16* x = tmpl + ((size_t)t * P + p) * D;
            float n = rbf(sqrtf(dot64_bf(x, x, D)));
            if (n < 1e-12f) n = 1e-12f;
            for (int i = 0; i < D; ++i) { tn[i] = rbf(bf2f(x[i]) / n); qf[i] = bf2f(q[(size_t)p * D + i]); }
            dots[p] = rbf(dot64_ff(tn, qf, D));
        }
        /* mean over patches: lane l sums p = l, l+64, ... then butterfly */
        float pl[64], wl[64];
        for (int l = 0; l < 64; ++l) {
            float a = 0.f, w = 0.f;
            for (int p = l; p < P; p += 64) {
                if (weights) { a += dots[p] * weights[(size_t)t * P + p]; w += weights[(size_t)t * P + p]; }
                else a += dots[p];
            }
            pl[l] = a; wl[l] = w;
        }
        const float s = butterfly64(pl);
        if (weights) scores[t] = s / butterfly64(wl);
        else scores[t] = rbf(s / (float)P);
    }
    free(tn); free(qf); free(dots);
}

/* torch's CPU nearest-neighbour source index for F.interpolate(scale_factor=s) on the NCHW (C=3) tensors of
 * bbox_utils.py:35,52.  ATen picks one of two kernels per call: when out_h + out_w <= 128 the one built on
 * nearest_idx (identity when out == in, idx >> 1 when out == 2*in, else the scale rule); otherwise the generic
 * kernel, which always applies the scale rule  min(floorf(dst * float(1/s)), in-1).  `small` = out_h + out_w <= 128.
 * (probed on torch 2.10 CPU; the 420-px pipeline always takes the generic kernel) */
static int nearest_src(int dst, int in, int out, float inv_scale, int small) {
    if (small && out == in) return dst;
    if (small && out == 2 * in) return dst >> 1;
    int s = (int)floorf((float)dst * inv_scale);
    return s > in - 1 ? in - 1 : s;
}

/* numpy's float32 pairwise summation for n <= 128 (np.mean of the top-k scores, extract_proposals_ground.py:156) */
static float np_pairwise_sum_f32(const float* a, int n) {
    if (n < 8) { float r = 0.f; for (int i = 0; i < n; ++i) r += a[i]; return r; }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}
static int fdesc_cmp(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return (x < y) - (x > y); }

/* per-view fine re-rank (scripts/extract_proposals_ground.py:147-160): per candidate mesh, normalise its per-view
 * descriptors in bf16, dot with the query (bf16 result), top-k, numpy float32 mean.  views bf16 [sum_views, D] (raw),
 * offsets [n_mesh+1], cand [Q,C], out [Q,C]. */
void fpo_rerank_views(const bf16* views, const int32_t* offsets, const int32_t* cand, const bf16* queries, int Q, int C, int D,
                      int k, float* out) {
    float* tn = (float*)malloc((size_t)D * 4);
    float* qf = (float*)malloc((size_t)D * 4);
    float* sc = (float*)malloc(1024 * 4);
    for (int q = 0; q < Q; ++q) {
        for (int i = 0; i < D; ++i) qf[i] = bf2f(queries[(size_t)q * D + i]);
        for (int c = 0; c < C; ++c) {
            const int mesh = cand[(size_t)q * C + c];
            const int r0 = offsets[mesh];
            int nv = offsets[mesh + 1] - r0; if (nv > 1024) nv = 1024;
            for (int v = 0; v < nv; ++v) {
                const bf16* x = views + (size_t)(r0 + v) * D;
                float n = rbf(sqrtf(dot64_bf(x, x, D)));
                if (n < 1e-12f) n = 1e-12f;
                for (int i = 0; i < D; ++i) tn[i] = rbf(bf2f(x[i]) / n);
                sc[v] = rbf(dot64_ff(tn, qf, D));
            }
            qsort(sc, nv, sizeof(float), fdesc_cmp);
            const int kk = k < nv ? k : nv;
            out[(size_t)q * C + c] = kk > 0 ? np_pairwise_sum_f32(sc, kk) / (float)kk : -3.0e38f;
        }
    }
    free(tn); free(qf); free(sc);
}

/* CropResizePad.__call__ (src/utils/bbox_utils.py:20-56).  images f32 [n_img,C,H,W] (src_u8=0) or
 * u8 [n_img,H,W,C] (src_u8=1, value/255 as in renderer.py:121); out f32 [n,C,target,target].
 * mask_mode as in include/freepose_hip.h.  Returns 0, or 1+i if box i does not resize to `target`. */
int fpo_crop_resize_pad(const void* images, int src_u8, int n_img, int C, int H, int W, const int32_t* boxes, int n,
                        float ext, int target, const uint8_t* masks, int mask_mode, float* out) {
    for (int i = 0; i < n; ++i) {
        int x0 = boxes[4 * i], y0 = boxes[4 * i + 1], x1 = boxes[4 * i + 2], y1 = boxes[4 * i + 3];
        const int bw = x1 - x0, bh = y1 - y0;
        if (ext == 0.f) {                        /* :22-28 with integer bbox_extend */
            if (x0 < 0) x0 = 0; if (x1 > W) x1 = W; if (y0 < 0) y0 = 0; if (y1 > H) y1 = H;
        } else {                                 /* float32 tensor arithmetic, truncating assignment */
            const float ew = ext * (float)bw, eh = ext * (float)bh;
            const float fx0 = (float)x0 - ew, fx1 = (float)x1 + ew, fy0 = (float)y0 - eh, fy1 = (float)y1 + eh;
            x0 = fx0 > 0.f ? (int)fx0 : 0; x1 = fx1 < (float)W ? (int)fx1 : W;
            y0 = fy0 > 0.f ? (int)fy0 : 0; y1 = fy1 < (float)H ? (int)fy1 : H;
        }
        const int cw = x1 - x0, ch = y1 - y0;
        const int side = cw > ch ? cw : ch;
        /* :30  `self.target_max / torch.max(...)` is Tensor.__rtruediv__ = reciprocal(tensor) * scalar in
         * float32 (two roundings, not one division); then .item() widens to double (:34) */
        const float recip = 1.0f / (float)side;
        const double scale = (double)(recip * (float)target);
        const int h1 = (int)floor((double)ch * scale), w1 = (int)floor((double)cw * scale);
        const float inv1 = (float)(1.0 / scale);                          /* nearest: float(1/scale_factor) */
        int pad_t = 0, pad_l = 0, S_h = h1, S_w = w1;
        if ((double)w1 / (double)h1 != 1.0) {                             /* :41-48 */
            pad_t = (target - h1) / 2; if (pad_t < 0 || target - h1 < 0) pad_t = 0;
            pad_l = (target - w1) / 2; if (pad_l < 0 || target - w1 < 0) pad_l = 0;
            S_h = target; S_w = target;
        }
        const double scale2 = (double)target / (double)S_h;               /* :52-54 */
        const int outsz = (int)floor((double)S_h * scale2);
        const float inv2 = (float)(1.0 / scale2);
        /* a crop the reference cannot make: empty, or F.interpolate would have to produce a side of 0 px (torch raises, :35) */
        if (outsz != target || cw <= 0 || ch <= 0 || h1 <= 0 || w1 <= 0) return 1 + i;
        const int img = n_img == 1 ? 0 : i;
        for (int oy = 0; oy < target; ++oy)
            for (int ox = 0; ox < target; ++ox) {
                const int small1 = (h1 + w1) <= 128, small2 = (outsz + outsz) <= 128;
                const int y2 = nearest_src(oy, S_h, outsz, inv2, small2);
                const int x2 = nearest_src(ox, S_w, outsz, inv2, small2);
                const int yy = y2 - pad_t, xx = x2 - pad_l;
                const int valid = !(yy < 0 || yy >= h1 || xx < 0 || xx >= w1);
                int ys = 0, xs = 0;
                if (valid) {
                    ys = y0 + nearest_src(yy, ch, h1, inv1, small1);
                    xs = x0 + nearest_src(xx, cw, w1, inv1, small1);
                }
                float m = 1.f;
                if (valid && masks && mask_mode) m = masks[((size_t)i * H + ys) * W + xs] ? 1.f : 0.f;
                for (int c = 0; c < C; ++c) {
                    float v = 0.f;
                    if (valid) {
                        if (mask_mode == 2) v = m;
                        else {
                            if (src_u8 == 1) v = (float)((double)((const uint8_t*)images)[(((size_t)img * H + ys) * W + xs) * C + c] / 255.0);
                            else if (src_u8 == 2) v = (float)((const uint8_t*)images)[(((size_t)img * H + ys) * W + xs) * C + c] / 255.0f; /* utils.py:20 */
                            else v = ((const float*)images)[(((size_t)img * C + c) * H + ys) * W + xs];
                            v *= m;
                        }
                    }
                    out[(((size_t)i * C + c) * target + oy) * target + ox] = v;
                }
            }
    }
    return 0;
}

/* super-Fibonacci rotation grid (src/pipeline/estimators/pose_estimator.py:121-147, renderer.py:12-35):
 * scalar-last quaternion -> matrix as scipy Rotation.from_quat(q).as_matrix() */
void fpo_generate_rotations(int n, double* out) {
    const double phi = sqrt(2.0), psi = 1.533751168755204288118041, PI = 3.14159265358979323846;
    for (int i = 0; i < n; ++i) {
        const double s = i + 0.5, r = sqrt(s / n), R = sqrt(1.0 - s / n);
        const double al = 2.0 * PI * s / phi, be = 2.0 * PI * s / psi;
        double x = r * sin(al), y = r * cos(al), z = R * sin(be), w = R * cos(be);
        const double nn = sqrt(x * x + y * y + z * z + w * w);
        x /= nn; y /= nn; z /= nn; w /= nn;
        double* M = out + (size_t)i * 9;
        M[0] = x * x - y * y - z * z + w * w; M[1] = 2 * (x * y - z * w); M[2] = 2 * (x * z + y * w);
        M[3] = 2 * (x * y + z * w); M[4] = -x * x + y * y - z * z + w * w; M[5] = 2 * (y * z - x * w);
        M[6] = 2 * (x * z - y * w); M[7] = 2 * (y * z + x * w); M[8] = -x * x - y * y + z * z + w * w;
    }
}

/* geodesic neighbourhood (online_pose_estimator.py:25-34,55-56): angle of R_i R_prev^T in degrees < thresh */
int fpo_geodesic_select(const double* grid, int G, const double* Rp, double thresh_deg, int32_t* out_idx) {
    int n = 0;
    for (int i = 0; i < G; ++i) {
        const double* R = grid + (size_t)i * 9;
        double D[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) D[3 * r + c] = R[3 * r] * Rp[3 * c] + R[3 * r + 1] * Rp[3 * c + 1] + R[3 * r + 2] * Rp[3 * c + 2];
        const double cosv = 0.5 * (D[0] + D[4] + D[8] - 1.0);
        const double a = D[7] - D[5], b = D[2] - D[6], c = D[3] - D[1];
        const double ang = atan2(0.5 * sqrt(a * a + b * b + c * c), cosv) * 57.29577951308232;
        if (ang < thresh_deg) out_idx[n++] = i;
    }
    return n;
}

/* depth>0 mask bbox with the <100 px fallback (renderer.py:112-119; template.py:73-78) and the x/y extents of
 * depthmap_to_pointcloud (src/pipeline/utils.py:122-145,157-158).  out [Hn,8] as in freepose_hip.h */
void fpo_depth_extents(const float* depth, int Hn, int Hh, int W, double fx, double fy, double cx, double cy, double* out) {
    for (int v = 0; v < Hn; ++v) {
        const float* d = depth + (size_t)v * Hh * W;
        int cnt = 0, xmin = 1 << 30, ymin = 1 << 30, xmax = -1, ymax = -1, any = 0;
        double Xmin = 1e300, Xmax = -1e300, Ymin = 1e300, Ymax = -1e300;
        for (int y = 0; y < Hh; ++y)
            for (int x = 0; x < W; ++x) {
                const float z = d[y * W + x];
                if (z != 0.f) {
                    any = 1;
                    const double X = ((double)x - cx) / fx * (double)z, Y = ((double)y - cy) / fy * (double)z;
                    if (X < Xmin) Xmin = X; if (X > Xmax) Xmax = X; if (Y < Ymin) Ymin = Y; if (Y > Ymax) Ymax = Y;
                    if (z > 0.f) { ++cnt; if (x < xmin) xmin = x; if (x > xmax) xmax = x; if (y < ymin) ymin = y; if (y > ymax) ymax = y; }
                }
            }
        if (cnt < 100) {
            const int lo = 105, hx = (315 < W ? 315 : W) - 1, hy = (315 < Hh ? 315 : Hh) - 1;
            if (cnt == 0) { xmin = lo; ymin = lo; xmax = hx; ymax = hy; }
            else { if (lo < xmin) xmin = lo; if (lo < ymin) ymin = lo; if (hx > xmax) xmax = hx; if (hy > ymax) ymax = hy; }
        }
        double* o = out + (size_t)v * 8;
        o[0] = xmin; o[1] = ymin; o[2] = xmax; o[3] = ymax;
        o[4] = any ? Xmax - Xmin : 0.0; o[5] = any ? Ymax - Ymin : 0.0;
        o[6] = cnt; o[7] = 0.0;
    }
}

/* ---- rasteriser: restatement of the arithmetic contract in freepose_amd/csrc/raster.hip's header
 * (pyrender/OpenGL is not in /root/reference: renderer.py:37-41,53-55,66 fix K, flip, ambient, no-cull;
 * bop_toolkit_lib/renderer_py.py:186-231 the K->projection convention).  Scan order here is triangle-major
 * with an explicit (depth, id) min — the same total order the atomic 64-bit min realises on the GPU. */
typedef struct { int xi, yi; float iz, zc; float xh, yh; } svert_t;   /* xh, yh: homogeneous pixel coords of a vertex behind the near plane */

/* Near plane (renderer.py:62-67: znear 0.05): a triangle that STRADDLES it is rasterised in homogeneous coordinates — an exact clip at
 * z = znear without new geometry (contract in raster.hip's header).  Vertex k -> (xk, yk, wk) = (pixel x * z, pixel y * z, z): from the
 * snapped fixed-point pixel when in front (shared edges coincide with ordinary triangles), from the camera-space point otherwise. */
typedef struct { double a[3], b[3], c[3], det; } strad_t;
static void strad_vertex(const svert_t* v, double* x, double* y, double* w) {
    *w = (double)v->zc;
    if (v->zc > 0.05f) { *x = ((double)v->xi * (1.0 / 256.0)) * *w; *y = ((double)v->yi * (1.0 / 256.0)) * *w; }
    else { *x = (double)v->xh; *y = (double)v->yh; }
}
static void strad_setup(const svert_t* v0, const svert_t* v1, const svert_t* v2, strad_t* q) {
    double x[3], y[3], w[3];
    strad_vertex(v0, &x[0], &y[0], &w[0]); strad_vertex(v1, &x[1], &y[1], &w[1]); strad_vertex(v2, &x[2], &y[2], &w[2]);
    for (int k = 0; k < 3; ++k) {
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        q->a[k] = y[k1] * w[k2] - y[k2] * w[k1];
        q->b[k] = x[k2] * w[k1] - x[k1] * w[k2];
        q->c[k] = x[k1] * y[k2] - x[k2] * y[k1];
    }
    q->det = (x[0] * q->a[0] + y[0] * q->b[0]) + w[0] * q->c[0];
}
static int strad_pixel(const strad_t* q, int px, int py, float* d, float* b0, float* b1, float* b2) {
    const double X = (double)px + 0.5, Y = (double)py + 0.5;
    double e0 = (q->a[0] * X + q->b[0] * Y) + q->c[0];
    double e1 = (q->a[1] * X + q->b[1] * Y) + q->c[1];
    double e2 = (q->a[2] * X + q->b[2] * Y) + q->c[2];
    double det = q->det;
    if (det < 0.0) { e0 = -e0; e1 = -e1; e2 = -e2; det = -det; }
    if (!(det > 0.0) || e0 < 0.0 || e1 < 0.0 || e2 < 0.0) return 0;
    const double S = (e0 + e1) + e2;
    if (!(S > 0.0)) return 0;
    const double z = det / S;
    if (!(z > (double)0.05f)) return 0;
    *d = (float)z;
    *b0 = (float)(e0 / S); *b1 = (float)(e1 / S); *b2 = (float)(e2 / S);
    return 1;
}
static int topleft(int dx, int dy) { return (dy < 0) || (dy == 0 && dx > 0); }

/* Shading contract (shared with raster.hip; pyrender's shader is not in /root/reference, so this is a stated rule, not a pin):
 *   base colour of a fragment:
 *     textured mesh : perspective-correct per-corner UV (U = fma(q2,u2, fma(q1,u1, q0*u0)) * depth, q_i = b_i*iz_i).  The texture
 *                     is filtered as stored (u8 values 0..255, i.e. in the image's gamma space — what a GL_RGBA8 texture does):
 *       level k       : size max(1, tw>>k) x max(1, th>>k); level k+1 = 2x2 box of level k, (a+b+c+d+2)>>2 per channel (the
 *                       second row / column is clamped when the source size is 1)
 *       bilinear(k)   : x = U*wk - 0.5, y = (1-V)*hk - 0.5 (v = 1 is the image's first row) ; (x0,y0) = floor ; REPEAT wrap ;
 *                       top = fma(wx, t01-t00, t00), bot = fma(wx, t11-t10, t10), val = fma(wy, bot-top, top)
 *       level of detail (filter 1, default = GL_LINEAR_MIPMAP_LINEAR with analytic derivatives): with g?_i = d(b_i)/d(px|py) * iz_i
 *                       (constants of the triangle), dU/dx = fma(gx2,u2-U, fma(gx1,u1-U, gx0*(u0-U))) * depth (same for V, y);
 *                       rho2 = max((dU/dx*tw)^2 + (dV/dx*th)^2, (dU/dy*tw)^2 + (dV/dy*th)^2).  rho2 <= 1 (magnification) or one
 *                       level only: bilinear(0).  Otherwise e = exponent(rho2), m = mantissa in [1,2), lg = LOG2P(m-1) (degree-5
 *                       polynomial, |err| < 2e-5), l0 = e>>1, fr = 0.5*((e&1) + lg) [= 0.5*log2(rho2) - l0]; l0 >= last level:
 *                       bilinear(last); else val = fma(fr, bilinear(l0+1) - bilinear(l0), bilinear(l0)).
 *                       filter 0: bilinear(0) always (the rule before mip-maps).
 *       shade 1       : lin = sRGB->linear of val/255 by linear interpolation in the 256-entry table DEC (i = min(int(val), 254),
 *                       lin = fma(val-i, DEC[i+1]-DEC[i], DEC[i])): sample first, decode after — the order of a shader that calls
 *                       srgb_to_linear(texture(...)) ; c = lin*Kd
 *       shade 0       : c255 = val*Kd
 *     vertex colours: cv = fma(q2,c2, fma(q1,c1, q0*c0)) * depth  (0..255 units)
 *   output: shade 0 (linear, the round-1 rule): u8(min(255, ambient*cv + 0.5))            [textured: cv = c255]
 *           shade 1 (gamma, default): u8 = #{ k in 1..255 : THR[k] <= ambient*c }, THR[k] = ((k-0.5)/255)^2.2 — i.e.
 *           round(255 * x^(1/2.2)) evaluated by table search so host and device agree bit for bit      [vertex: c = cv*(1/255.f)] */
static float g_dec[256], g_thr[256];
static int g_tables_ready = 0;
static void shade_tables(void) {
    if (g_tables_ready) return;
    for (int i = 0; i < 256; ++i) {
        const double s = (double)i / 255.0;
        g_dec[i] = (float)(s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4));
        g_thr[i] = i == 0 ? 0.f : (float)pow(((double)i - 0.5) / 255.0, 2.2);
    }
    g_tables_ready = 1;
}
void fpo_shade_tables(float* dec256, float* thr256) { shade_tables(); memcpy(dec256, g_dec, 1024); memcpy(thr256, g_thr, 1024); }
static uint8_t encode_gamma(float x) {
    int lo = 0, hi = 255;                      /* largest k with THR[k] <= x (THR[0] = 0 <= x for x >= 0; NaN / negative -> 0) */
    if (!(x > 0.f)) return 0;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (g_thr[mid] <= x) lo = mid; else hi = mid - 1; }
    return (uint8_t)lo;
}
static int wrapi(int a, int n) { int m = a % n; return m < 0 ? m + n : m; }

/* mip chain of an RGB u8 texture (see the contract): returns malloc'd texels of all levels, level k at offset off[k] (texels) */
#define FPO_MAX_LEV 16
static uint8_t* build_mips(const uint8_t* tex, int th, int tw, int* nlev, size_t* off, int* lw, int* lh) {
    int n = 0; size_t total = 0;
    for (int w = tw, h = th;; w = w > 1 ? w >> 1 : 1, h = h > 1 ? h >> 1 : 1) {
        lw[n] = w; lh[n] = h; off[n] = total; total += (size_t)w * h; ++n;
        if ((w == 1 && h == 1) || n == FPO_MAX_LEV) break;
    }
    uint8_t* m = (uint8_t*)malloc(total * 3);
    memcpy(m, tex, (size_t)tw * th * 3);
    for (int k = 1; k < n; ++k) {
        const uint8_t* src = m + off[k - 1] * 3; uint8_t* dst = m + off[k] * 3;
        const int sw = lw[k - 1], sh = lh[k - 1];
        for (int y = 0; y < lh[k]; ++y)
            for (int x = 0; x < lw[k]; ++x) {
                const int x0 = 2 * x < sw ? 2 * x : sw - 1, x1 = 2 * x + 1 < sw ? 2 * x + 1 : sw - 1;
                const int y0 = 2 * y < sh ? 2 * y : sh - 1, y1 = 2 * y + 1 < sh ? 2 * y + 1 : sh - 1;
                for (int ch = 0; ch < 3; ++ch)
                    dst[((size_t)y * lw[k] + x) * 3 + ch] = (uint8_t)((src[((size_t)y0 * sw + x0) * 3 + ch] + src[((size_t)y0 * sw + x1) * 3 + ch] +
                                                                     src[((size_t)y1 * sw + x0) * 3 + ch] + src[((size_t)y1 * sw + x1) * 3 + ch] + 2) >> 2);
            }
    }
    *nlev = n;
    return m;
}
/* bilinear sample of one level, values in 0..255 units */
static void bilinear_level(const uint8_t* lev, int w, int h, float U, float Vv, float out[3]) {
    float x = fmaf(U, (float)w, -0.5f), y = fmaf(1.0f - Vv, (float)h, -0.5f);
    x = fminf(fmaxf(x, -1.0e6f), 1.0e6f); y = fminf(fmaxf(y, -1.0e6f), 1.0e6f);
    if (!(x == x)) x = 0.f;
    if (!(y == y)) y = 0.f;
    const float xf = floorf(x), yf = floorf(y);
    const float wx = x - xf, wy = y - yf;
    const int x0 = wrapi((int)xf, w), x1 = wrapi((int)xf + 1, w);
    const int y0 = wrapi((int)yf, h), y1 = wrapi((int)yf + 1, h);
    for (int ch = 0; ch < 3; ++ch) {
        const float t00 = (float)lev[((size_t)y0 * w + x0) * 3 + ch], t01 = (float)lev[((size_t)y0 * w + x1) * 3 + ch];
        const float t10 = (float)lev[((size_t)y1 * w + x0) * 3 + ch], t11 = (float)lev[((size_t)y1 * w + x1) * 3 + ch];
        const float top = fmaf(wx, t01 - t00, t00), bot = fmaf(wx, t11 - t10, t10);
        out[ch] = fmaf(wy, bot - top, top);
    }
}
/* LOG2P(t) ~ log2(1 + t) on [0,1): t * Horner(c1..c5), every step an fmaf */
static float log2p(float t) {
    float a = 0.045148879289627075f;
    a = fmaf(a, t, -0.19357527792453766f);
    a = fmaf(a, t, 0.41560569405555725f);
    a = fmaf(a, t, -0.7090963125228882f);
    a = fmaf(a, t, 1.441917061805725f);
    return a * t;
}

void fpo_rasterize_tex(const float* verts, int V, const int32_t* faces, int F, const uint8_t* colors /* [V,3] or NULL */,
                       const float* uv /* [F,3,2] or NULL */, const uint8_t* tex /* [th,tw,3] or NULL */, int th, int tw,
                       const float* kd3 /* or NULL = 1,1,1 */, const float* poses, int Hn, float scale, float fx, float fy,
                       float cx, float cy, int W, int Hh, uint8_t* rgb, float* depth, float ambient, int shade, int filter, int cull);
void fpo_rasterize_amb(const float* verts, int V, const int32_t* faces, int F, const uint8_t* colors /* [V,3] or NULL */,
                       const float* poses, int Hn, float scale, float fx, float fy, float cx, float cy, int W, int Hh,
                       uint8_t* rgb, float* depth, float ambient) {
    fpo_rasterize_tex(verts, V, faces, F, colors, NULL, NULL, 0, 0, NULL, poses, Hn, scale, fx, fy, cx, cy, W, Hh, rgb, depth, ambient, 0, 0, 0);
}

/* vertex stage alone: fixed-point 24.8 window coordinates (image convention: x right, y down, pixel centres at +0.5) and the
 * camera-frame depth Zc.  Pinned against the reference's own K -> OpenGL projection (bop_toolkit_lib/renderer_py.py:186-231
 * with the OpenCV->OpenGL flip of renderer.py:37-41) by tests/test_golden_cpu.py. */
void fpo_project_vertices(const float* verts, int V, const float* poses, int Hn, float scale, float fx, float fy, float cx,
                          float cy, int32_t* xy /* [Hn,V,2] */, float* zc /* [Hn,V] */) {
    const float ZNEAR = 0.05f;
    for (int h = 0; h < Hn; ++h) {
        const float* P = poses + (size_t)h * 16;
        for (int i = 0; i < V; ++i) {
            const float sx = scale * verts[3 * i], sy = scale * verts[3 * i + 1], sz = scale * verts[3 * i + 2];
            const float Xc = fmaf(P[0], sx, fmaf(P[1], sy, fmaf(P[2], sz, P[3])));
            const float Yc = fmaf(P[4], sx, fmaf(P[5], sy, fmaf(P[6], sz, P[7])));
            const float Zc = fmaf(P[8], sx, fmaf(P[9], sy, fmaf(P[10], sz, P[11])));
            int xi = 0, yi = 0;
            if (Zc > ZNEAR) {
                const float iz = 1.0f / Zc;
                float u = fmaf(fx, Xc * iz, cx), v = fmaf(fy, Yc * iz, cy);
                u = fminf(fmaxf(u, -30000.f), 30000.f); v = fminf(fmaxf(v, -30000.f), 30000.f);
                xi = (int)rintf(u * 256.0f); yi = (int)rintf(v * 256.0f);
            }
            xy[((size_t)h * V + i) * 2] = xi; xy[((size_t)h * V + i) * 2 + 1] = yi;
            zc[(size_t)h * V + i] = Zc;
        }
    }
}

void fpo_rasterize_tex(const float* verts, int V, const int32_t* faces, int F, const uint8_t* colors, const float* uv,
                       const uint8_t* tex, int th, int tw, const float* kd3, const float* poses, int Hn, float scale, float fx,
                       float fy, float cx, float cy, int W, int Hh, uint8_t* rgb, float* depth, float ambient, int shade, int filter,
                       int cull /* 1: back faces (renderer.py:63-66 without SKIP_CULL_FACES) are not drawn */) {
    const float ZNEAR = 0.05f;
    const int textured = uv && tex && th > 0 && tw > 0;
    int nlev = 0, lw[FPO_MAX_LEV], lh[FPO_MAX_LEV];
    size_t loff[FPO_MAX_LEV];
    uint8_t* mips = textured ? build_mips(tex, th, tw, &nlev, loff, lw, lh) : NULL;
    const float kd[3] = {kd3 ? kd3[0] : 1.f, kd3 ? kd3[1] : 1.f, kd3 ? kd3[2] : 1.f};
    shade_tables();
    svert_t* sv = (svert_t*)malloc((size_t)V * sizeof(svert_t));
    uint64_t* zb = (uint64_t*)malloc((size_t)W * Hh * 8);
    for (int h = 0; h < Hn; ++h) {
        const float* P = poses + (size_t)h * 16;
        for (int i = 0; i < V; ++i) {
            const float sx = scale * verts[3 * i], sy = scale * verts[3 * i + 1], sz = scale * verts[3 * i + 2];
            const float Xc = fmaf(P[0], sx, fmaf(P[1], sy, fmaf(P[2], sz, P[3])));
            const float Yc = fmaf(P[4], sx, fmaf(P[5], sy, fmaf(P[6], sz, P[7])));
            const float Zc = fmaf(P[8], sx, fmaf(P[9], sy, fmaf(P[10], sz, P[11])));
            sv[i].zc = Zc; sv[i].xi = 0; sv[i].yi = 0; sv[i].iz = 0.f; sv[i].xh = 0.f; sv[i].yh = 0.f;
            if (Zc > ZNEAR) {
                const float iz = 1.0f / Zc;
                float u = fmaf(fx, Xc * iz, cx), v = fmaf(fy, Yc * iz, cy);
                u = fminf(fmaxf(u, -30000.f), 30000.f); v = fminf(fmaxf(v, -30000.f), 30000.f);
                sv[i].xi = (int)rintf(u * 256.0f); sv[i].yi = (int)rintf(v * 256.0f); sv[i].iz = iz;
            } else {   /* behind the near plane: homogeneous pixel coordinates, used by triangles that straddle it */
                sv[i].xh = fmaf(fx, Xc, cx * Zc); sv[i].yh = fmaf(fy, Yc, cy * Zc);
            }
        }
        for (size_t i = 0; i < (size_t)W * Hh; ++i) zb[i] = ~0ull;
        for (int f = 0; f < F; ++f) {
            int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
            svert_t a = sv[i0], b = sv[i1], c = sv[i2];
            const int nfront = (a.zc > ZNEAR) + (b.zc > ZNEAR) + (c.zc > ZNEAR);
            if (nfront == 1 || nfront == 2) {   /* straddles the near plane: homogeneous rasterisation over the whole frame */
                strad_t q; strad_setup(&a, &b, &c, &q);
                if (cull && q.det > 0.0) continue;             /* back face: same sign convention as the screen-space area below */
                for (int py = 0; py < Hh; ++py)
                    for (int px = 0; px < W; ++px) {
                        float d, b0, b1, b2;
                        if (!strad_pixel(&q, px, py, &d, &b0, &b1, &b2)) continue;
                        uint32_t db; memcpy(&db, &d, 4);
                        const uint64_t key = ((uint64_t)db << 32) | (uint32_t)f;
                        if (key < zb[(size_t)py * W + px]) zb[(size_t)py * W + px] = key;
                    }
                continue;
            }
            if (nfront != 3) continue;
            int64_t area2 = (int64_t)(b.xi - a.xi) * (c.yi - a.yi) - (int64_t)(b.yi - a.yi) * (c.xi - a.xi);
            /* x right, y down, z forward: a counter-clockwise-from-outside triangle that faces the camera has NEGATIVE area here */
            if (cull && area2 > 0) continue;
            if (area2 < 0) { svert_t t = b; b = c; c = t; area2 = -area2; }
            if (area2 == 0) continue;
            int mnx = a.xi < b.xi ? a.xi : b.xi; if (c.xi < mnx) mnx = c.xi;
            int mxx = a.xi > b.xi ? a.xi : b.xi; if (c.xi > mxx) mxx = c.xi;
            int mny = a.yi < b.yi ? a.yi : b.yi; if (c.yi < mny) mny = c.yi;
            int mxy = a.yi > b.yi ? a.yi : b.yi; if (c.yi > mxy) mxy = c.yi;
            int bx0 = (int)floor((double)(mnx - 128 + 255) / 256.0), by0 = (int)floor((double)(mny - 128 + 255) / 256.0);
            int bx1 = (int)floor((double)(mxx - 128) / 256.0), by1 = (int)floor((double)(mxy - 128) / 256.0);
            if (bx0 < 0) bx0 = 0; if (by0 < 0) by0 = 0; if (bx1 > W - 1) bx1 = W - 1; if (by1 > Hh - 1) by1 = Hh - 1;
            for (int py = by0; py <= by1; ++py)
                for (int px = bx0; px <= bx1; ++px) {
                    const int64_t sx = (int64_t)px * 256 + 128, sy = (int64_t)py * 256 + 128;
                    const int64_t w0 = (int64_t)(c.xi - b.xi) * (sy - b.yi) - (int64_t)(c.yi - b.yi) * (sx - b.xi);
                    const int64_t w1 = (int64_t)(a.xi - c.xi) * (sy - c.yi) - (int64_t)(a.yi - c.yi) * (sx - c.xi);
                    const int64_t w2 = (int64_t)(b.xi - a.xi) * (sy - a.yi) - (int64_t)(b.yi - a.yi) * (sx - a.xi);
                    if (w0 < 0 || w1 < 0 || w2 < 0) continue;
                    if (w0 == 0 && !topleft(c.xi - b.xi, c.yi - b.yi)) continue;
                    if (w1 == 0 && !topleft(a.xi - c.xi, a.yi - c.yi)) continue;
                    if (w2 == 0 && !topleft(b.xi - a.xi, b.yi - a.yi)) continue;
                    const float fa = (float)area2;
                    const float b0 = (float)w0 / fa, b1 = (float)w1 / fa, b2 = (float)w2 / fa;
                    const float izp = fmaf(b2, c.iz, fmaf(b1, b.iz, b0 * a.iz));
                    const float d = 1.0f / izp;
                    if (!(d > 0.f)) continue;
                    uint32_t db; memcpy(&db, &d, 4);
                    const uint64_t key = ((uint64_t)db << 32) | (uint32_t)f;
                    if (key < zb[(size_t)py * W + px]) zb[(size_t)py * W + px] = key;
                }
        }
        for (int py = 0; py < Hh; ++py)
            for (int px = 0; px < W; ++px) {
                const uint64_t key = zb[(size_t)py * W + px];
                float d = 0.f; uint8_t col[3] = {0, 0, 0};
                if (key != ~0ull) {
                    const int f = (int)(uint32_t)(key & 0xffffffffu);
                    uint32_t db = (uint32_t)(key >> 32); memcpy(&d, &db, 4);
                    int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
                    int k0 = 0, k1 = 1, k2 = 2;                       /* corner order follows the orientation swap */
                    svert_t a = sv[i0], b = sv[i1], c = sv[i2];
                    const int nfront = (a.zc > ZNEAR) + (b.zc > ZNEAR) + (c.zc > ZNEAR);
                    const int strad = nfront != 3;                    /* a winner with a vertex behind the plane is a straddler */
                    float fa = 1.f, dd = 1.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
                    if (strad) {       /* true (3-D) barycentrics, original corner order, level-0 texture */
                        strad_t sq; strad_setup(&a, &b, &c, &sq);
                        float ds;
                        strad_pixel(&sq, px, py, &ds, &q0, &q1, &q2);
                    } else {
                    int64_t area2 = (int64_t)(b.xi - a.xi) * (c.yi - a.yi) - (int64_t)(b.yi - a.yi) * (c.xi - a.xi);
                    if (area2 < 0) { svert_t t = b; b = c; c = t; int ti = i1; i1 = i2; i2 = ti; k1 = 2; k2 = 1; area2 = -area2; }
                    const int64_t sx = (int64_t)px * 256 + 128, sy = (int64_t)py * 256 + 128;
                    const int64_t w0 = (int64_t)(c.xi - b.xi) * (sy - b.yi) - (int64_t)(c.yi - b.yi) * (sx - b.xi);
                    const int64_t w1 = (int64_t)(a.xi - c.xi) * (sy - c.yi) - (int64_t)(a.yi - c.yi) * (sx - c.xi);
                    const int64_t w2 = (int64_t)(b.xi - a.xi) * (sy - a.yi) - (int64_t)(b.yi - a.yi) * (sx - a.xi);
                    fa = (float)area2;
                    const float b0 = (float)w0 / fa, b1 = (float)w1 / fa, b2 = (float)w2 / fa;
                    const float izp = fmaf(b2, c.iz, fmaf(b1, b.iz, b0 * a.iz));
                    dd = 1.0f / izp;
                    q0 = b0 * a.iz; q1 = b1 * b.iz; q2 = b2 * c.iz;
                    }
                    if (textured) {
                        const float* t = uv + (size_t)f * 6;
                        const float u0 = t[2 * k0], u1 = t[2 * k1], u2 = t[2 * k2];
                        const float v0 = t[2 * k0 + 1], v1 = t[2 * k1 + 1], v2 = t[2 * k2 + 1];
                        const float U = fmaf(q2, u2, fmaf(q1, u1, q0 * u0)) * dd;
                        const float Vv = fmaf(q2, v2, fmaf(q1, v1, q0 * v0)) * dd;
                        float val[3];
                        int l0 = 0, two = 0; float fr = 0.f;
                        if (filter && nlev > 1 && !strad) {
                            /* d(w_i)/d(px) = -256 (y_b - y_a), d(w_i)/d(py) = 256 (x_b - x_a) of the edge opposite corner i */
                            const float gx0 = (float)(-(int64_t)(c.yi - b.yi) * 256) / fa * a.iz, gy0 = (float)((int64_t)(c.xi - b.xi) * 256) / fa * a.iz;
                            const float gx1 = (float)(-(int64_t)(a.yi - c.yi) * 256) / fa * b.iz, gy1 = (float)((int64_t)(a.xi - c.xi) * 256) / fa * b.iz;
                            const float gx2 = (float)(-(int64_t)(b.yi - a.yi) * 256) / fa * c.iz, gy2 = (float)((int64_t)(b.xi - a.xi) * 256) / fa * c.iz;
                            const float dux = fmaf(gx2, u2 - U, fmaf(gx1, u1 - U, gx0 * (u0 - U))) * dd * (float)tw;
                            const float dvx = fmaf(gx2, v2 - Vv, fmaf(gx1, v1 - Vv, gx0 * (v0 - Vv))) * dd * (float)th;
                            const float duy = fmaf(gy2, u2 - U, fmaf(gy1, u1 - U, gy0 * (u0 - U))) * dd * (float)tw;
                            const float dvy = fmaf(gy2, v2 - Vv, fmaf(gy1, v1 - Vv, gy0 * (v0 - Vv))) * dd * (float)th;
                            float r2 = fmaxf(fmaf(dux, dux, dvx * dvx), fmaf(duy, duy, dvy * dvy));
                            if (!(r2 == r2)) r2 = 0.f;
                            if (r2 > 1.0f) {
                                uint32_t rb; memcpy(&rb, &r2, 4);
                                const int e = (int)(rb >> 23) - 127;
                                uint32_t mb = (rb & 0x7fffffu) | 0x3f800000u; float m; memcpy(&m, &mb, 4);
                                const float lg = log2p(m - 1.0f);
                                l0 = e >> 1; fr = 0.5f * ((float)(e & 1) + lg);
                                if (l0 >= nlev - 1) { l0 = nlev - 1; fr = 0.f; } else two = 1;
                            }
                        }
                        bilinear_level(mips + loff[l0] * 3, lw[l0], lh[l0], U, Vv, val);
                        if (two) {
                            float v1l[3];
                            bilinear_level(mips + loff[l0 + 1] * 3, lw[l0 + 1], lh[l0 + 1], U, Vv, v1l);
                            for (int ch = 0; ch < 3; ++ch) val[ch] = fmaf(fr, v1l[ch] - val[ch], val[ch]);
                        }
                        for (int ch = 0; ch < 3; ++ch) {
                            if (shade) {
                                int ii = (int)val[ch]; if (ii > 254) ii = 254; if (ii < 0) ii = 0;
                                const float lin = fmaf(val[ch] - (float)ii, g_dec[ii + 1] - g_dec[ii], g_dec[ii]);
                                col[ch] = encode_gamma(ambient * (lin * kd[ch]));
                            } else {
                                float amb = fminf(ambient * (val[ch] * kd[ch]) + 0.5f, 255.0f); if (amb < 0.f) amb = 0.f; col[ch] = (uint8_t)amb;
                            }
                        }
                    } else {
                        for (int ch = 0; ch < 3; ++ch) {
                            float c0 = 255.f, c1 = 255.f, c2 = 255.f;
                            if (colors) { c0 = (float)colors[3 * i0 + ch]; c1 = (float)colors[3 * i1 + ch]; c2 = (float)colors[3 * i2 + ch]; }
                            const float cv = fmaf(q2, c2, fmaf(q1, c1, q0 * c0)) * dd;
                            if (shade) col[ch] = encode_gamma(ambient * (cv * (1.0f / 255.f)));
                            else { float amb = fminf(ambient * cv + 0.5f, 255.0f); if (amb < 0.f) amb = 0.f; col[ch] = (uint8_t)amb; }
                        }
                    }
                }
                depth[((size_t)h * Hh + py) * W + px] = d;
                uint8_t* o = rgb + (((size_t)h * Hh + py) * W + px) * 3;
                o[0] = col[0]; o[1] = col[1]; o[2] = col[2];
            }
    }
    free(sv); free(zb); free(mips);
}


/* ---- f-3: RoIAlign forward, aligned=False (torchvision.ops.roi_align as called by src/pipeline/refiner_utils.py:127-132).
 * Restated from the published operator (torchvision/csrc/ops/cpu/roi_align_kernel.cpp); torchvision itself is not
 * available in this image, so this piece of the oracle is "parity unpinned" and held by known-answer tests only. */
static float fpo_roi_bilinear(const float* in, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * in[y_low * W + x_low] + w2 * in[y_low * W + x_high] + w3 * in[y_high * W + x_low] + w4 * in[y_high * W + x_high];
}

int fpo_roi_align(const float* images, int n_img, int C, int H, int W, const float* rois, int n, int PH, int PW, int sampling,
                  float spatial_scale, float* out) {
    (void)n_img;
    for (int r = 0; r < n; ++r) {
        const float* roi = rois + (size_t)r * 5;
        const int b = (int)roi[0];
        const float roi_start_w = roi[1] * spatial_scale, roi_start_h = roi[2] * spatial_scale;
        const float roi_end_w = roi[3] * spatial_scale, roi_end_h = roi[4] * spatial_scale;
        const float roi_w = fmaxf(roi_end_w - roi_start_w, 1.0f), roi_h = fmaxf(roi_end_h - roi_start_h, 1.0f);
        const float bin_h = roi_h / (float)PH, bin_w = roi_w / (float)PW;
        const int grid_h = sampling > 0 ? sampling : (int)ceilf(roi_h / (float)PH);
        const int grid_w = sampling > 0 ? sampling : (int)ceilf(roi_w / (float)PW);
        const int cnt_i = grid_h * grid_w > 1 ? grid_h * grid_w : 1;
        const float count = (float)cnt_i;
        for (int c = 0; c < C; ++c) {
            const float* in = images + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < grid_h; ++iy) {
                        const float y = roi_start_h + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
                        for (int ix = 0; ix < grid_w; ++ix) {
                            const float x = roi_start_w + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
                            acc += fpo_roi_bilinear(in, H, W, y, x);
                        }
                    }
                    out[(((size_t)r * C + c) * PH + ph) * PW + pw] = acc / count;
                }
        }
    }
    return 0;
}
