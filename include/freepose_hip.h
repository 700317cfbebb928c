/* freepose_hip.h — C ABI of libfreepose_hip.so, the MI355X (gfx950) implementation of FreePose's
 * per-proposal 6D-pose hot path.  Plain pointers and sizes only; every pointer named `d_*` or documented
 * "device" is a HIP device pointer owned by the caller, every call takes the HIP stream to enqueue on
 * (void* = hipStream_t; NULL = default stream) and returns 0 on success or an FP_ERR_* code, with
 * fp_last_error() giving the text.  No exceptions cross this boundary.  Nothing here touches torch.
 *
 * Each entry names the reference interface it replaces (file:line relative to ponimatkin/freepose).
 * bf16 tensors are raw uint16 bit patterns.
 */
#ifndef FREEPOSE_HIP_H
#define FREEPOSE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP_OK 0
#define FP_ERR_INVALID 1
#define FP_ERR_HIP 2
#define FP_ERR_STATE 3

typedef struct fp_ctx fp_ctx; /* one per (thread, GPU): owns workspaces, no hidden globals */
typedef struct fp_vit fp_vit; /* ViT weights table + per-resolution pos-embed cache         */
typedef struct fp_mesh fp_mesh; /* device copy of a triangle mesh for the rasteriser         */

const char* fp_last_error(void);
int fp_version(void);
int fp_ctx_create(int device, fp_ctx** out);
int fp_ctx_destroy(fp_ctx* ctx);
/* Run-time options of ONE context (no process-global state, no environment variables: the library never calls getenv).
 * value < 0 restores the default.
 *   "ln_fused":       1 (default) LayerNorm 1 / 2 folded into the qkv / fc1 GEMMs of fp_vit_forward on a ViT of this context,
 *                     0 the separate LayerNorm kernel (same reference rounding points; used by the parity tests)
 *   "raster_tiled":   unset = LDS-tiled rasteriser for meshes up to 131 072 triangles and images up to 704 px, global
 *                     visibility-buffer strategy otherwise; 1 / 0 force one of them (both bit-identical)
 *   "comm_timeout_s": seconds fp_comm_init waits for the rendezvous of all ranks (default 180) before it returns FP_ERR_STATE
 *   "gemm_row_split": 1 (default) a GEMM launch between the tile tiers runs whole rounds of the resident grid on 256x256 tiles and
 *                     the remaining rows on the finer tiers; 0 never splits (bit-identical; tests/test_gpu_kernels.py)
 * The measurement variants of earlier rounds (alternative GEMM main loops, attention ring depths, ...) are not in this library:
 * they are compiled only into the lab build (python -m freepose_amd.build --lab -> libfreepose_hip_lab.so, used by tools/). */
int fp_ctx_set_option(fp_ctx* ctx, const char* name, int value);
/* bytes currently held in the context's workspaces (diagnostics) */
size_t fp_ctx_workspace_bytes(const fp_ctx* ctx);

/* ---- a1/a2: DINOv2FeatureExtractor (src/pipeline/retrieval/dino.py:8-32; hub dinov2 ViT) -------------- */
typedef struct {
    int dim;       /* 1024 (ViT-L), 384 (ViT-S), 768 (ViT-B)            */
    int depth;     /* 24 / 12 / 12                                     */
    int heads;     /* dim / 64                                         */
    int mlp_dim;   /* 4 * dim                                          */
    int patch;     /* 14                                               */
    int n_reg;     /* 4 register tokens (0 for non-reg checkpoints)    */
    int pos_grid;  /* 37 (pos_embed is [1 + 37*37, dim])               */
    float ln_eps;  /* 1e-6                                             */
} fp_vit_arch;

int fp_vit_create(fp_ctx* ctx, const fp_vit_arch* arch, fp_vit** out);
int fp_vit_destroy(fp_vit* vit);
/* Register one state-dict tensor by its DINOv2 name (dino.py:10 loads exactly this checkpoint layout):
 *   cls_token, pos_embed, register_tokens, patch_embed.proj.{weight,bias}, blocks.{i}.norm1.{weight,bias},
 *   blocks.{i}.attn.qkv.{weight,bias}, blocks.{i}.attn.proj.{weight,bias}, blocks.{i}.ls1.gamma,
 *   blocks.{i}.norm2.{weight,bias}, blocks.{i}.mlp.fc1.{weight,bias}, blocks.{i}.mlp.fc2.{weight,bias},
 *   blocks.{i}.ls2.gamma, norm.{weight,bias}
 * d_bf16 is a device pointer to the bf16 tensor in its native layout; the caller keeps it alive.
 * patch_embed.proj.weight is copied into a K-padded private buffer. */
int fp_vit_set_weight(fp_vit* vit, const char* name, const void* d_bf16, size_t numel, void* stream);
/* feature_type: 0 = cls [B,dim], 1 = reg [B,n_reg,dim], 2 = patch [B,P,dim]  (dino.py:25-30);
 * 3 = patch features with every row F.normalize()d — what the estimators score (pose_estimator.py:85-88,
 * online_pose_estimator.py:72-76 normalise the template / hypothesis features right after this call): the bits of
 * fp_l2_normalize(feature_type 2), written by the final-norm kernel itself.
 * d_images: bf16 [B,3,H,W] in [0,1] (ImageNet normalisation is fused, dino.py:12,16);
 * runs blocks 0..layer-1 (all blocks if layer > depth, dino.py:18-21) then the final norm. */
int fp_vit_forward(fp_vit* vit, const void* d_images, int B, int H, int W, int layer, int feature_type,
                   void* d_out_bf16, void* stream);
/* algorithmic FLOPs of one fp_vit_forward call (SURVEY §8d formula) */
double fp_vit_flops(const fp_vit* vit, int B, int H, int W, int layer);

/* ---- a3: FFA descriptor (scripts/extract_retrieval_features.py:49-57, extract_proposals_ground.py:126-134) */
/* d_feats bf16 [B,P,D]; d_mask u8 [B, gh*cell, gw*cell] (cell=14: any-pool == cv2 INTER_AREA > 0) or
 * [B,P] with cell=1.  d_out_bf16 [B,D] (may be NULL), d_out_f32 [B,D] (bf16 values widened, may be NULL);
 * normalize != 0 applies F.normalize(dim=-1) with bf16 rounding points to d_out_bf16. */
int fp_ffa(fp_ctx* ctx, const void* d_feats, const uint8_t* d_mask, int B, int gh, int gw, int D, int cell,
           int normalize, void* d_out_bf16, float* d_out_f32, void* stream);

/* ---- a4: cosine top-k retrieval (scripts/extract_proposals_ground.py:39-41,136-140) ------------------- */
/* bank prep: fp32 [N,D] -> bf16 -> row L2-normalise in bf16 (ground.py:40-41).  d_bank_bf16 out [N,D]. */
int fp_bank_prepare(fp_ctx* ctx, const float* d_bank_f32, int N, int D, void* d_bank_bf16, void* stream);
/* scores = bf16(bank @ q).float(); top-k by (score desc, index asc).  d_queries bf16 [Q,D].
 * idx_offset is added to every returned index (bank-row sharding across ranks). */
int fp_bank_topk(fp_ctx* ctx, const void* d_bank_bf16, int N, int D, const void* d_queries, int Q, int k,
                 int idx_offset, float* d_out_scores, int32_t* d_out_idx, void* stream);
/* merge per-shard candidates [Q,C] down to [Q,k] with the same ordering (after an RCCL all-gather). */
int fp_topk_merge(fp_ctx* ctx, const float* d_cand_scores, const int32_t* d_cand_idx, int Q, int C, int k,
                  float* d_out_scores, int32_t* d_out_idx, void* stream);
/* per-view fine re-rank (scripts/extract_proposals_ground.py:147-160, --topk k): d_views bf16 [sum_views, D] = the
 * per-mesh descriptor files back to back (np.load(...).to(bf16), NOT normalised), d_offsets i32 [n_mesh+1] row offsets,
 * d_cand i32 [Q,C] coarse candidates (mesh rows), d_queries bf16 [Q,D].  d_out f32 [Q,C] = numpy float32 mean of the
 * top-k per-view scores bf16(normalize_bf16(view) . f).  k <= 128, <= 1024 views per mesh. */
int fp_rerank_views(fp_ctx* ctx, const void* d_views, const int32_t* d_offsets, const int32_t* d_cand,
                    const void* d_queries, int Q, int C, int D, int k, float* d_out, void* stream);
/* F.normalize(x, dim=-1) on bf16 rows with the reference's rounding points. */
int fp_l2_normalize(fp_ctx* ctx, const void* d_x_bf16, int rows, int D, void* d_y_bf16, void* stream);

/* ---- a7/a10: patchwise template score (pose_estimator.py:85-88, online_pose_estimator.py:68-79) ------- */
/* d_tmpl bf16 [T,P,D] raw template features (normalised on the fly), d_query bf16 [P,D] used AS GIVEN
 * (callers pass F.normalize'd or raw features exactly where the reference does).  d_weights f32 [T,P] or
 * NULL (mask_scores variant).  d_scores f32 [T]. */
int fp_template_score(fp_ctx* ctx, const void* d_tmpl, const void* d_query, const float* d_weights, int T, int P,
                      int D, float* d_scores, void* stream);
/* The same score for a PRE-NORMALISED template store: d_tmpl_normed = fp_l2_normalize of the raw [T*P, D]
 * features, done once when they enter the cache.  F.normalize of a bf16 tensor is a bf16 tensor, so this is the reference's
 * own intermediate (pose_estimator.py:85) and the scores are bit-identical to fp_template_score on the raw features; the
 * kernel is a streaming dot (one pass over T*P*D*2 bytes). */
int fp_template_score_normed(fp_ctx* ctx, const void* d_tmpl_normed, const void* d_query, const float* d_weights, int T,
                             int P, int D, float* d_scores, void* stream);

/* ---- a5: CropResizePad (src/utils/bbox_utils.py:20-56) as used by Proposals (src/pipeline/utils.py:32-52)
 * and MeshRenderer.generate_proposals (renderer.py:109-130) -------------------------------------------- */
/* d_images: src_fmt 0 = f32 [n_img,C,H,W] in [0,1]; 1 = u8 [n_img,H,W,C] -> float(double(x)/255) like
 * renderer.py:121; 2 = u8 [n_img,H,W,C] -> float(x)/255.f like Proposals (utils.py:20).  n_img is 1 (all boxes crop the same image, Proposals) or n (one image per box, renders).
 * d_boxes i32 [n,4] xyxy BEFORE extension.  d_masks u8 [n,H,W] or NULL; mask_mode 0 = ignore,
 * 1 = multiply pixels by the mask (mask_rgb=True, utils.py:39-40), 2 = output the mask itself as 0/1
 * (utils.py:35-37,48-51).  d_out: out_fmt 0 = f32, 1 = bf16, shape [n,C,target,target].
 * Nearest-neighbour index rules replicate F.interpolate exactly (bbox_utils.py:29-35,52-54).
 * A box the reference cannot crop — empty after the clip, a resized side of 0 px (torch raises inside F.interpolate, :35) or an
 * exactly square crop that comes out one pixel short of `target` — gets an all-zero crop here (the boxes are device data: no
 * read-back); the Python mirror classes raise on such boxes like the reference when the boxes are host-resident
 * (freepose_amd/src/utils/bbox_utils.py unresizable_box: the same arithmetic on the host). */
int fp_crop_resize_pad(fp_ctx* ctx, const void* d_images, int src_fmt, int n_img, int C, int H, int W,
                       const int32_t* d_boxes, int n, float bbox_extend, int target, const uint8_t* d_masks,
                       int mask_mode, void* d_out, int out_fmt, void* stream);

/* ---- f-3: TrackingRefiner photo crop (src/pipeline/refiner_utils.py:92-132 crop_image -> torchvision.ops.roi_align,
 * output 518x518, sampling_ratio 2, aligned=False) ------------------------------------------------------------------ */
/* d_images f32 [n_img,C,H,W]; d_rois f32 [n,5] = (image index, x1, y1, x2, y2); d_out f32 [n,C,pooled_h,pooled_w].
 * sampling_ratio <= 0 selects the adaptive ceil(roi/pooled) grid of the original operator. */
int fp_roi_align(fp_ctx* ctx, const float* d_images, int n_img, int C, int H, int W, const float* d_rois, int n,
                 int pooled_h, int pooled_w, int sampling_ratio, float spatial_scale, float* d_out, void* stream);

/* ---- a8/a10: rotation grids and neighbourhood (pose_estimator.py:121-147, online_pose_estimator.py:25-34,55-56) */
/* super-Fibonacci rotations, fp64 [n,3,3] row-major written to HOST memory (one-off setup, n=600/20000). */
int fp_generate_rotations(int n, double* h_out);
/* indices i with geodesic(R_i, R_prev) < thresh_deg; d_grid f64 [G,3,3] (device), h_Rprev9 f64 row-major
 * (host); out idx ascending (np.where order), count in *h_n.  Synchronises the stream (the count decides
 * how many hypotheses are rendered next). */
int fp_geodesic_select(fp_ctx* ctx, const double* d_grid, int G, const double* h_Rprev9, double thresh_deg,
                       int32_t* d_out_idx, int* h_n, void* stream);

/* ---- a11: MeshRenderer (src/pipeline/retrieval/renderer.py:43-95; pyrender/OpenGL semantics) ---------- */
/* vertices f32 [V,3], faces i32 [F,3], per-vertex colours u8 [V,3] or NULL (white).  All host pointers. */
int fp_mesh_upload(fp_ctx* ctx, const float* h_verts, int V, const int32_t* h_faces, int F,
                   const uint8_t* h_colors, fp_mesh** out);
/* textured mesh, as pyrender.Mesh.from_trimesh(mesh) receives it from trimesh.load(obj, force='mesh') (renderer.py:43-45,70-72;
 * scripts/dino_inference_video.py:93-101, scripts/render_templates.py:58-66): per-corner texture coordinates uv f32 [F,3,2]
 * (OBJ `vt` convention: v up), diffuse texture u8 [th,tw,3] (image rows top to bottom), material diffuse factor kd f32 [3] or
 * NULL (= 1,1,1).  Fragments are shaded with a perspective-correct texture fetch (REPEAT wrap, trilinear over a box-filtered
 * mip chain built at upload — the sampler pyrender gives a trimesh texture; fp_mesh_set_filter(.., 0) = bilinear level 0); the
 * exact arithmetic is the contract at the top of csrc/raster.hip.  All host pointers. */
int fp_mesh_upload_textured(fp_ctx* ctx, const float* h_verts, int V, const int32_t* h_faces, int F, const float* h_uv,
                            const uint8_t* h_texture, int th, int tw, const float* h_kd3, fp_mesh** out);
int fp_mesh_destroy(fp_mesh* mesh);
/* ambient light factor of the scene the mesh is rendered in: 2 (default; renderer.py:53-55,80-82) or 5
 * (tracking_refiner.py:33) */
int fp_mesh_set_ambient(fp_mesh* mesh, float ambient);
/* output transfer: 1 (default) = gamma rule, u8 = round(255 (ambient c)^(1/2.2)) with sRGB-decoded texels (how a PBR ambient
 * term reaches the frame buffer); 0 = linear rule, u8 = min(255, 255 ambient c + .5).  pyrender's shader is third-party and
 * not in the reference tree, so neither is pinned; both are bit-exact against the oracle. */
int fp_mesh_set_shading(fp_mesh* mesh, int mode);
/* texture minification: 1 (default) = trilinear mip-maps, level of detail from the analytic UV derivatives of the fragment;
 * 0 = bilinear fetch of level 0 only.  Replaces the sampler state of pyrender's texture objects (renderer.py:43-47,70-74). */
int fp_mesh_set_filter(fp_mesh* mesh, int mode);
/* back-face culling: 0 (default, what every caller of the reference uses: renderer.py:66,93 pass SKIP_CULL_FACES) = both sides drawn;
 * 1 = `cull_faces=True` (renderer.py:63-64,90-91): triangles whose counter-clockwise-from-outside side faces away are not drawn. */
int fp_mesh_set_cull(fp_mesh* mesh, int mode);
/* vertex stage of the rasteriser alone: window coordinates in 24.8 fixed point d_xy i32 [Hn,V,2] (x right, y down, pixel
 * centres at +0.5; 0,0 for vertices at or behind the near plane) and camera-frame depth d_zc f32 [Hn,V].  Conventions pinned
 * against the reference's K -> OpenGL projection (bop_toolkit_lib/renderer_py.py:186-231, renderer.py:37-41). */
int fp_project_vertices(fp_ctx* ctx, const fp_mesh* mesh, const float* d_poses, int Hn, float scale, float fx, float fy,
                        float cx, float cy, int32_t* d_xy, float* d_zc, void* stream);
/* poses f32 [Hn,4,4] (OpenCV camera frame, object->camera), intrinsics fx,fy,cx,cy, image W x Hh.
 * scale multiplies the vertices (rendering_scale 0.25).  Outputs rgb u8 [Hn,Hh,W,3], depth f32 [Hn,Hh,W]
 * (metric eye depth, 0 = background).  Ambient-only shading, no culling (renderer.py:53-55,66). */
int fp_rasterize(fp_ctx* ctx, const fp_mesh* mesh, const float* d_poses, int Hn, float scale, float fx, float fy,
                 float cx, float cy, int W, int Hh, uint8_t* d_rgb, float* d_depth, void* stream);
/* fp_rasterize with the consumers of the depth image fused into the tile epilogue (renderer.py:98-130 takes the
 * depth>0 bounding box of every render, pose_estimator.py:104-112 the cloud extents of the winners): d_ext f64 [Hn,8] = exactly what
 * fp_depth_extents gives on the depth image, d_boxes i32 [Hn,4] = its first four columns as CropResizePad takes them.  d_depth,
 * d_ext, d_boxes may each be NULL (not d_ext and d_boxes both): without d_depth the 4 bytes per pixel are neither written nor read
 * back.  Same bits as fp_rasterize + fp_depth_extents. */
int fp_rasterize_extents(fp_ctx* ctx, const fp_mesh* mesh, const float* d_poses, int Hn, float scale, float fx, float fy,
                         float cx, float cy, int W, int Hh, uint8_t* d_rgb, float* d_depth, double* d_ext, int32_t* d_boxes,
                         void* stream);
/* a9/K17: per view, bbox of depth>0 (with the <100 px fallback square) and metric extents of the
 * back-projected cloud (float64 like utils.py:122-145): out f64 [Hn,8] = {xmin,ymin,xmax,ymax (px), dx, dy (m),
 * count, 0}. */
int fp_depth_extents(fp_ctx* ctx, const float* d_depth, int Hn, int Hh, int W, float fx, float fy, float cx,
                     float cy, double* d_out, void* stream);

/* ---- e: multi-GPU.  One process per GPU; RCCL over xGMI.  The reference has no counterpart (SLURM array jobs +
 * files: scripts/dino_inference.py:51-54, merge_results.py); Python hosts use torch.distributed (freepose_amd/parallel.py), these
 * entry points serve hosts without it.  librccl is opened lazily on first use. ------------------------------------------- */
/* rank 0 creates the 128-byte RCCL unique id (ncclGetUniqueId) and hands it to the other ranks through the host's own channel */
int fp_comm_unique_id(void* out_id128);
/* collective over all `nranks` ranks (ncclCommInitRank).  Fails — status + fp_last_error(), never a hang — on a bad rank / nranks,
 * an all-zero id, a context that already has a communicator, or when the rendezvous does not complete within "comm_timeout_s"
 * (ranks that disagree on nranks or on the id, a rank that never arrives). */
int fp_comm_init(fp_ctx* ctx, int nranks, int rank, const void* unique_id128);
int fp_comm_destroy(fp_ctx* ctx);
int fp_comm_size(const fp_ctx* ctx);
int fp_comm_rank(const fp_ctx* ctx);
/* all-gather `bytes` device bytes from every rank into d_recv [nranks * bytes] (rank order); a copy without a communicator */
int fp_allgather_bytes(fp_ctx* ctx, const void* d_send, size_t bytes, void* d_recv, void* stream);
/* bank-row sharding: every rank passes its local top-k [Q,k] with GLOBAL row indices; the candidates of all ranks
 * are gathered and merged with the canonical (score desc, index asc) rule -> identical [Q,k_out] on every rank */
int fp_allgather_topk(fp_ctx* ctx, const float* d_scores, const int32_t* d_idx, int Q, int k, int k_out, float* d_out_scores,
                      int32_t* d_out_idx, void* stream);
/* proposal / frame / object sharding: gather n_rows fixed-length f64 result rows per rank (every rank passes the same n_rows;
 * pad with rows the caller can recognise) into d_out [nranks * n_rows, row_len] */
int fp_allgather_poses(fp_ctx* ctx, const double* d_rows, int n_rows, int row_len, double* d_out, void* stream);

/* ---- kernel-level entry points (unit parity tests, microbenchmarks; the ViT forward is built from these) */
/* C[M,N] = epi(X[M,K] W[N,K]^T + bias): epi 0 = bias, 1 = bias+GELU(erf), 2 = resid + gamma*(.) ; bf16, ld* in
 * elements (multiples of 8), K % 64 == 0, N % 16 == 0. */
int fp_op_gemm(fp_ctx* ctx, const void* d_X, int ldx, const void* d_W, int ldw, void* d_C, int ldc, const void* d_bias,
               const void* d_gamma, const void* d_resid, int ldr, int M, int N, int K, int epi, void* stream);
/* V part of qkv stored transposed per head: Vt[b,h,d,t] for rows m = b*npad + t, n = h*64 + d */
int fp_op_gemm_vt(fp_ctx* ctx, const void* d_X, int ldx, const void* d_W, int ldw, void* d_Vt, const void* d_bias, int M, int N,
                  int K, int npad, int heads, void* stream);
/* LayerNorm folded into the consuming linear layer, the way fp_vit_forward runs LN1 -> qkv and LN2 -> fc1 (hub DINOv2 block:
 * x + ls1 * attn(norm1(x)); x + ls2 * mlp(norm2(x)) — the nn.LayerNorm + nn.Linear pairs behind src/pipeline/retrieval/dino.py:18-19):
 *   LN(x) W^T + b  =  rstd (x W'^T - mean colsum(W')) + b',   W' = W diag(gamma_ln),  b' = b + W beta_ln
 * d_X bf16 [M,K] raw rows, d_g_ln / d_b_ln bf16 [K], d_W bf16 [N,K], d_bias bf16 [N].  mode 0: d_out bf16 [M,N] = linear,
 * 1: GELU(linear), 2: transposed per head like fp_op_gemm_vt (npad, heads).  Output features n < n_scaled are multiplied by row_scale
 * inside the fold (W' and b' of those rows; one rounding) — the ViT's q rows with log2(e) / sqrt(64); n_scaled = 0 for none.
 * Kernel-level entry used by the tests. */
int fp_op_ln_linear(fp_ctx* ctx, const void* d_X, int M, int K, const void* d_g_ln, const void* d_b_ln, float eps, const void* d_W,
                    int N, const void* d_bias, int mode, int npad, int heads, int n_scaled, float row_scale, void* d_out, void* stream);
/* fp_op_gemm with epilogue 2 (LayerScale + residual) that also emits the row statistics of what it wrote — the producer side of the
 * folded LayerNorm (per-64-column partial sums in the epilogue, summed in block order).  d_stat u32/f32 [M,6]: words 0-3 the 16-byte
 * init-MFMA record of the row as the consuming GEMM reads it — bf16 {sh, sl, sh, -mh, -ml, -mh, 0, 0}, sigma = sqrt(var + eps) and
 * -mean as two-piece bf16 splits —, word 4 rstd = 1 / sigma (f32), word 5 unused. */
int fp_op_gemm_stats(fp_ctx* ctx, const void* d_X, int ldx, const void* d_W, int ldw, void* d_C, int ldc, const void* d_bias,
                     const void* d_gamma, const void* d_resid, int ldr, int M, int N, int K, float eps, float* d_stat, void* stream);
/* d_y[i] = bf16(0.5 x (1 + erf(x / sqrt 2))) elementwise on bf16 values in fp32: the direct expression the fc1 epilogue's GELU table
 * is filled from (hub DINOv2 Mlp act_layer = nn.GELU behind src/pipeline/retrieval/dino.py:18-19); the table-GELU GEMM is tested
 * against it on all 65 536 bf16 inputs */
int fp_op_gelu(const void* d_x, void* d_y, size_t n, void* stream);
/* flash attention forward on QK [B*npad, 2*H*64] (ldqk elements) + Vt [B,H,64,npad] -> O [B*npad, H*64].
 * q_prescaled != 0: the q columns already hold q * log2(e) / sqrt(64) (fp_vit_forward folds that factor into the q rows of its
 * LayerNorm-folded qkv weights — fp_op_ln_linear's n_scaled / row_scale — so the softmax needs no multiply-add per element) */
int fp_op_attention(const void* d_QK, int ldqk, const void* d_Vt, void* d_O, int ldo, int B, int H, int n_tok,
                    int npad, int q_prescaled, void* stream);
int fp_op_layernorm(const void* d_X, void* d_Y, const void* d_gamma, const void* d_beta, int rows, int D, float eps,
                    void* stream);
/* ImageNet normalise (torchvision Normalize on a bf16 tensor: subtract, round, divide, round — src/pipeline/retrieval/dino.py:12,16) +
 * patch unfold of bf16 crops [B,3,H,W] in [0,1] into the patch-embed GEMM's A operand [B*(H/ps)*(W/ps), KP], k = c*ps*ps + dy*ps + dx,
 * columns >= 3*ps*ps zero (the Conv2d patch_embed.proj of hub DINOv2 behind dino.py:18).  Kernel-level entry used by the tests. */
int fp_op_im2col_norm(const void* d_img, void* d_A, int B, int H, int W, int ps, int KP, void* stream);

/* ---- measurement helpers (bench.py: HIP-event timing on the launch stream) ----------------------------- */
int fp_timer_create(void** out);
int fp_timer_start(void* timer, void* stream);
int fp_timer_stop(void* timer, void* stream);
int fp_timer_elapsed_ms(void* timer, float* h_ms); /* synchronises on the stop event */
int fp_timer_destroy(void* timer);
/* toggles per-kernel-class event timing inside fp_vit_forward (gemm / attention / other), read back in ms */
int fp_vit_profile(fp_vit* vit, int enable);
int fp_vit_profile_read(fp_vit* vit, float* h_ms_gemm, float* h_ms_attn, float* h_ms_other, double* h_gemm_flops);
/* number of GEMM kernel launches recorded since fp_vit_profile(vit, 1) (events are recorded without host syncs;
 * fp_vit_profile_read synchronises once and sums) */
long fp_vit_profile_gemm_launches(const fp_vit* vit);

#ifdef __cplusplus
}
#endif
#endif /* FREEPOSE_HIP_H */
