/*
 * dmnerf_hip.h -- C ABI of libdmnerf_hip.so: the MI355X (gfx950) implementation of the
 * DM-NeRF ray-rendering hot path.
 *
 * The reference (vLAR-group/DM-NeRF) has no FFI/plugin layer: its boundary is a set of Python
 * callables.  Each entry point below replaces the ATen op sequence of
 * one of them and cites it (paths relative to the reference checkout).  The Python mirror of
 * those callables (dm_nerf_amd/networks/) binds these symbols with ctypes; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer marked d_ is a DEVICE pointer to contiguous row-major float32 (int64 where
 *     stated); h_ pointers are HOST pointers.  No torch / ATen types cross this boundary.
 *   - the caller owns all memory; the library never allocates, frees or synchronises on the hot
 *     path, and launches only on the `stream` it is given (hipStream_t passed as void*),
 *     so every call is graph-capturable.
 *   - return 0 on success; negative on error (DMNERF_E_*), message via dmnerf_last_error().
 *   - sizes are validated first; an EMPTY batch (N == 0 rays / M == 0 rows) is legal, returns 0 without launching
 *     and without looking at the data pointers (an empty tensor's pointer is null).
 *   - N rays, S samples per ray, C = ins_num + 1 object logits, raw row = [r g b sigma | C logits].
 */
#ifndef DMNERF_HIP_H
#define DMNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMNERF_ABI_VERSION 8

#define DMNERF_OK 0
#define DMNERF_E_ARG (-1)     /* bad size / null pointer / unsupported shape */
#define DMNERF_E_LAUNCH (-2)  /* hipGetLastError() after launch */
#define DMNERF_E_NODEV (-3)   /* no HIP device visible */

#define DMNERF_POS_CH 63   /* 3 + 2*10*3, get_embedder(multires=10)  networks/dm_nerf.py:41-55 */
#define DMNERF_DIR_CH 27   /* 3 + 2*4*3,  get_embedder(multires_views=4) */
#define DMNERF_W 256       /* netwidth  (config.py:33 default; every shipped config) */
#define DMNERF_D 8         /* netdepth  (config.py:31 default), skips=[4] (config.py:133) */
#define DMNERF_MAX_LOGITS 128 /* C = ins_num+1 <= 128 (Replica room_0: 94) */
#define DMNERF_MAX_TRAIN_SAMPLES 1048576LL /* samples per training launch: 32-bit byte offsets into a 256-row tensor */

int dmnerf_abi_version(void);
const char* dmnerf_last_error(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int dmnerf_device_count(void);

/* ---- weights: reference state_dict -> kernel layout -------------------------------------
 * Flat parameter order = DM_NeRF.__init__ / state_dict order (networks/dm_nerf.py:59-78):
 *   mlps.0..7, rgb_feature_linear, ins_feature_linear, rgb_feature_linears.0,
 *   ins_feature_linears.0, density_linear, ins_linear, rgb_linear; each as weight[out,in]
 *   row-major followed by bias[out].
 * The kernel "blob" holds the same numbers permuted into MFMA A-operand order (DESIGN.md
 * section 3), zero padded.  blob[i] = idx[i] >= 0 ? flat[idx[i]] : 0.
 */
int64_t dmnerf_param_count(int ins_num);                    /* 696338 for ins_num=13 */
int64_t dmnerf_blob_floats(int ins_num);
int dmnerf_build_pack_index(int ins_num, int32_t* h_idx, int64_t n_idx);   /* host only, no GPU */
int dmnerf_pack_weights(const float* d_flat, const int32_t* d_idx, float* d_blob,
                        int64_t n_blob, void* stream);

/* ---- helpers.py --------------------------------------------------------------------------
 * get_rays_k (networks/helpers.py:50-61) for image rows [row0, row0+nrows):
 *   dirs = [(i-K02)/K00, (j-K12)/K11, K22]; rays_d = R dirs; rays_o = c2w[:3,3].
 *   h_intr = {K00, K11, K02, K12, K22} already rounded to float32 (what ATen does with the
 *   numpy float64 scalars); h_c2w = first 3 rows of c2w, row-major [3][4].
 *   d_rays_o / d_rays_d: [nrows*W, 3].                                                     */
int dmnerf_raygen(int H, int W, const float* h_intr, const float* h_c2w, int row0, int nrows,
                  float* d_rays_o, float* d_rays_d, void* stream);

/* Rays of selected pixels only: what get_select_full (networks/helpers.py:99-111) gathers out of a
 * full-frame get_rays_k.  d_idx: int64 flat pixel indices k = row*W + col; outputs [n,3].           */
int dmnerf_raygen_select(int H, int W, const float* h_intr, const float* h_c2w, const int64_t* d_idx,
                         int64_t n, float* d_rays_o, float* d_rays_d, void* stream);

/* z_val_sample (networks/helpers.py:114-119): z[n,s] = near + t[s]*(far-near); d_t = linspace(0,1,S). */
int dmnerf_z_val_sample(const float* d_t, float near_, float far_, int64_t N, int S, float* d_z,
                        void* stream);

/* stratified jitter (networks/render.py:42-47): mids/upper/lower, lower + (upper-lower)*t_rand. */
int dmnerf_stratify(const float* d_z_in, const float* d_t_rand, int64_t N, int S, float* d_z_out,
                    void* stream);

/* sample_pdf (networks/helpers.py:123-155).  bins [N,nb], weights [N,nb-1], u [N,n_samples]
 * (u_row_stride = n_samples) or one shared row (u_row_stride = 0, the det=True linspace).
 * Optional outputs (may be NULL): d_cdf [N,nb] float32, d_inds [N,n_samples] int64.          */
int dmnerf_sample_pdf(const float* d_bins, const float* d_weights, const float* d_u,
                      int64_t u_row_stride, int64_t N, int nb, int n_samples, float* d_samples,
                      float* d_cdf, int64_t* d_inds, void* stream);

/* stage-isolated inverse-CDF step (helpers.py:139-153): same (cdf,u) => same inds, samples. */
int dmnerf_sample_from_cdf(const float* d_bins, const float* d_cdf, const float* d_u,
                           int64_t u_row_stride, int64_t N, int nb, int n_samples,
                           float* d_samples, int64_t* d_inds, void* stream);

/* hierarchical resample + merge (networks/render.py:66-70):
 *   z_mid, sample_pdf(z_mid, weights[...,1:-1], n_imp, u), sort(cat(z_coarse, z_samples)).
 * d_z_samples may be NULL.  d_z_fine [N, S + n_imp].                                         */
int dmnerf_importance_resample(const float* d_z_coarse, const float* d_weights_coarse,
                               const float* d_u, int64_t u_row_stride, int64_t N, int S, int n_imp,
                               float* d_z_fine, float* d_z_samples, void* stream);

/* ---- dm_nerf.py ---------------------------------------------------------------------------
 * Embedder.embed (networks/dm_nerf.py:37-38): x [M,3] -> [M, 3+6L], L frequencies 2^0..2^(L-1). */
int dmnerf_embed(const float* d_x, int64_t M, int L, float* d_out, void* stream);

/* DM_NeRF.forward (networks/dm_nerf.py:80-106) on pre-embedded rows x [M, 63+27] -> raw [M, 4+C]. */
int dmnerf_mlp_fwd_embedded(const float* d_blob, int ins_num, const float* d_x, int64_t M,
                            float* d_raw, void* stream);

/* Fused points + positional encoding + DM_NeRF.forward (networks/render.py:49-61 / :71-83):
 *   pts = o + d*z; embed(pts) | embed(d/|d|); MLP.  rays_o/rays_d [N,3], z [N,S] -> raw [N,S,4+C]. */
int dmnerf_mlp_fwd_rays(const float* d_blob, int ins_num, const float* d_rays_o,
                        const float* d_rays_d, const float* d_z, int64_t N, int S, float* d_raw,
                        void* stream);

/* ---- render.py ----------------------------------------------------------------------------
 * render_train (networks/render.py:6-28): raw [N,S,4+C], z [N,S], rays_d [N,3] ->
 *   rgb_map [N,3], weights [N,S], depth_map [N], ins_map [N,C-1] (sigmoid after the sum, last
 *   channel dropped).                                                                       */
int dmnerf_composite_fwd(const float* d_raw, const float* d_z, const float* d_rays_d, int64_t N,
                         int S, int C, float* d_rgb_map, float* d_weights, float* d_depth_map,
                         float* d_ins_map, void* stream);

/* ---- training (autograd of the two entry points above) ------------------------------------
 * Backward of render_train: given dL/d{rgb_map [N,3], ins_map [N,C-1], depth_map [N] (nullable),
 * weights [N,S] (nullable)} -> dL/draw [N,S,4+C] (every element written).  d_ins_map = the forward's
 * ins_map.  Honours weights.detach() on the ins path (render.py:22-23).                        */
int dmnerf_composite_bwd(const float* d_raw, const float* d_z, const float* d_rays_d,
                         const float* d_ins_map, const float* d_g_rgb, const float* d_g_ins,
                         const float* d_g_depth, const float* d_g_weights, int64_t N, int S, int C,
                         float* d_grad_raw, void* stream);

/* Training forward of dmnerf_mlp_fwd_rays: additionally saves every layer input / relu output into
 * d_save (dmnerf_train_save_floats(M) floats, M = N*S, Mp = M rounded up to 32).  Each tensor with R
 * rows is stored block-major: addr(sample m, row) = ((m/32 * R + row) * 32 + m%32); tensors in order:
 *   embed(pts) R=63 | embed(dirs) R=27 | h_0..h_7 8 x R=256 | rgb hidden 128 | ins hidden 128,
 *   each R*Mp floats, then the 1-bit ReLU masks.  (rgb_feature / ins_feature are not saved: they have no
 *   activation, dm_nerf.py:89,96, and the backward re-associates around them -- see dmnerf_head_product.)
 * In the 256- and 128-row tensors (not the two encodings) memory row rho holds feature
 * (rho & ~7) | ((rho & 7) >> 1) | ((rho & 1) << 2), i.e. inside each group of 8 features the rows are
 * 0,4,1,5,2,6,3,7 (the kernels' TID-addressed stores; only dmnerf_mlp_bwd_* read these buffers, and the
 * gradients they return are in the reference's parameter order).   M <= DMNERF_MAX_TRAIN_SAMPLES.       */
int64_t dmnerf_train_save_floats(int64_t M);
int dmnerf_mlp_fwd_rays_train(const float* d_blob, int ins_num, const float* d_rays_o,
                              const float* d_rays_d, const float* d_z, int64_t N, int S,
                              float* d_raw, float* d_save, void* stream);
/* OPT-IN: the training forward on the fused-heads blob (dmnerf_fuse_heads + dmnerf_build_pack_index_fused): same outputs
 * and the same d_save contents as dmnerf_mlp_fwd_rays_train up to f32 re-association of the two feature linears. */
int dmnerf_mlp_fwd_rays_train_fused(const float* d_blob_fused, int ins_num, const float* d_rays_o,
                                    const float* d_rays_d, const float* d_z, int64_t N, int S,
                                    float* d_raw, float* d_save, void* stream);
/* OPT-IN: the training forward on the split-bf16 MFMA path (blob of dmnerf_pack_split): f32-class values (not the bitwise
 * f32 chain), the same f32 d_save contents for the same backward. */
int dmnerf_mlp_fwd_rays_train_split(const float* d_blob_split, int ins_num, const float* d_rays_o, const float* d_rays_d,
                                    const float* d_z, int64_t N, int S, float* d_raw, float* d_save, void* stream);
/* DM_NeRF.forward on pre-embedded rows (networks/dm_nerf.py:80-106 called directly, as mesh / third-party code does) in
 * training mode: d_raw as dmnerf_mlp_fwd_embedded, plus the activation workspace d_save (dmnerf_train_save_floats(M))
 * that dmnerf_mlp_bwd_data / dmnerf_mlp_bwd_weights consume.  Parameter gradients only: the kernels produce no
 * gradient for the embedded rows themselves. */
int dmnerf_mlp_fwd_embedded_train(const float* d_blob, int ins_num, const float* d_x, int64_t M, float* d_raw,
                                  float* d_save, void* stream);

/* Backward blob (W^T as MFMA A operand) and the data-gradient pass: dL/draw [M,4+C] + d_save ->
 * d_dsave (same layout as d_save): dy of mlps.0..7 in the h rows, d(rgb hidden pre-act), d(ins hidden pre-act).
 * Weight gradients are dW = dy . x^T over M.
 * d_graw_t (nullable): receives dL/draw in block-major form [block][4+C][32] (zero padding columns),
 * the third operand of dmnerf_mlp_bwd_weights.
 * Gradient barriers: h.detach() on the ins branch (dm_nerf.py:95); none to the encodings.
 * The two feature linears carry no activation, so their gradients are never materialised: with
 * A = rgb_feature_linears.0.weight[:, :256] and F = A . rgb_feature_linear.weight [128,256], d h_7 = F^T dg1 is ONE
 * GEMM (instead of 128 -> 256 -> 256).  dmnerf_head_product forms F (f32 fmaf chain, k ascending) from the flat
 * parameter vector; the W^T blob gathers from [flat parameters | F]: the index of dmnerf_build_pack_index_t addresses
 * dmnerf_param_count(ins_num) + 32768 source floats.                                             */
int64_t dmnerf_blob_t_floats(int ins_num);
int dmnerf_build_pack_index_t(int ins_num, int32_t* h_idx, int64_t n_idx);
int dmnerf_head_product(const float* d_params_flat, int ins_num, float* d_F, void* stream);
/* Inference-only head fusion (the blobs of dmnerf_mlp_fwd_rays_fused / _split): d_flat_fused = the flat parameter vector
 * with the two hidden layers replaced by hidden . feature (products accumulated in float64, rounded once); it must not
 * alias d_params_flat.  Pack it with the index of dmnerf_build_pack_index_fused.                                    */
int dmnerf_fuse_heads(const float* d_params_flat, int ins_num, float* d_flat_fused, void* stream);
int dmnerf_mlp_bwd_data(const float* d_blob, const float* d_blob_t, int ins_num, const float* d_save,
                        const float* d_graw, int64_t M, float* d_dsave, float* d_graw_t, void* stream);

/* Weight / bias gradients dW = dy . x^T over the batch (split-K f32 MFMA, deterministic 2-stage sum).
 * The plan (which workgroup does which job slices) depends only on (ins_num, M, max_wgs): build it once
 * on the host, upload the two tables, reuse every step.  n_jobs = the number of WORKGROUPS (<= max_wgs; the grid of the launch,
 * all filled to the same modelled time); the job table (n_job_bytes) holds one leader item per workgroup followed by the further
 * items some of them continue with (a job's last slice and the next job's first), each item with its own partial tile.  d_graw_t = dL/draw in the same block-major
 * form, R = 4+C rows, zero in the padding columns.  d_grad_flat: dmnerf_param_count(ins_num) floats in the
 * flat parameter order above.  d_part: workspace of `part_floats` floats.
 * The gradients of rgb_feature_linear, ins_feature_linear and of the two hidden layers that consume them are formed from
 * G = dg1 . h_7^T, Q = dg2 . h_7^T and the weights themselves (d_params_flat: the flat parameter vector the forward
 * used), e.g. d rgb_feature_linear.weight = A^T G: exact algebra, f32 re-association (csrc/heads.hip).        */
int dmnerf_wgrad_plan_sizes(int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes,
                            int64_t* n_out_bytes, int64_t* part_floats, int* n_jobs, int* n_outs);
int dmnerf_wgrad_plan(int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes,
                      void* h_outs, int64_t out_bytes);
int dmnerf_mlp_bwd_weights(const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M,
                           const void* d_jobs, int n_jobs, const void* d_outs, int n_outs,
                           const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat, void* stream);
/* OPT-IN split-bf16 weight gradients (training with args.mfma_split; csrc/wgrad_split.hip) -- the parameter gradients of
 * networks/dm_nerf.py:80-106 that torch autograd forms in the reference's train loops (train_dmsr.py:62-64): the same tables, workspace and
 * second stage; both f32 operands are split on the fly into three bf16 planes and a product is six bf16 MFMAs accumulated
 * in f32 (f32-class, not the bitwise chain of dmnerf_mlp_bwd_weights).  The _split plan balances the slices for this
 * kernel's chunk times (its 256 x 256 jobs are HBM-bound); either plan is valid for either kernel.                  */
int dmnerf_wgrad_plan_sizes_split(int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes,
                                  int64_t* n_out_bytes, int64_t* part_floats, int* n_jobs, int* n_outs);
int dmnerf_wgrad_plan_split(int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes,
                            void* h_outs, int64_t out_bytes);
int dmnerf_mlp_bwd_weights_split(const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M,
                                 const void* d_jobs, int n_jobs, const void* d_outs, int n_outs,
                                 const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat, void* stream);
/* Diagnostic: when d_ticks != NULL every workgroup of the following dmnerf_mlp_bwd_weights launches writes its
 * {start, end} 100 MHz wall-clock ticks to d_ticks[2*wg], d_ticks[2*wg+1] (n_jobs pairs); NULL turns it off.
 * Used by scripts/diag_wgrad.py to fit the split-K cost model. */
int dmnerf_wgrad_set_trace(int64_t* d_ticks);

/* ---- inference-only algebraic fusion (opt-in) -------------------------------------------
 * rgb_feature_linear and ins_feature_linear have no activation (dm_nerf.py:89,96), so for inference they can be
 * folded into the hidden layers that consume them: W' = W_hidden[:, :256] . W_feature, b' = W_hidden[:, :256] .
 * b_feature + b_hidden (the caller forms them, e.g. in float64, and writes them over the rgb_feature_linears.0 /
 * ins_feature_linears.0 slots of a copy of the flat parameter vector).  The fused blob has the same table and the
 * same stream without the two 256x256 stages (36 quarters instead of 44: -18.9 % MACs at ins_num 13).  Results
 * differ from the layer-by-layer evaluation by f32 re-association only (|d raw| ~ 1e-6), hence opt-in.           */
int64_t dmnerf_blob_fused_floats(int ins_num);
int dmnerf_build_pack_index_fused(int ins_num, int32_t* h_idx, int64_t n_idx);
int dmnerf_mlp_fwd_rays_fused(const float* d_blob_fused, int ins_num, const float* d_rays_o,
                              const float* d_rays_d, const float* d_z, int64_t N, int S,
                              float* d_raw, void* stream);

/* ---- opt-in split-bf16 ("bf16x3") inference (docs/EXPERIMENTS.md section 8) -----------------------------------------
 * The MLP on v_mfma_f32_32x32x16_bf16 with every f32 operand split by truncation into three bf16 planes
 * (x = hi + mid + lo exactly) and the six leading products accumulated in f32: the rounding class of an f32 GEMM
 * at 2.7x fewer MFMA cycles; not the bitwise fmaf chain of dmnerf_mlp_fwd_rays, hence opt-in.  The blob is
 * [the fused blob's 4096-float table | split stream]: dmnerf_build_pack_index_split gives one int32 per bf16
 * element of the stream (source parameter | plane << 28, -1 = zero) over the FUSED flat parameter vector
 * (see above), dmnerf_pack_split writes the stream words (n_words = dmnerf_blob_split_words - 4096).          */
int64_t dmnerf_blob_split_words(int ins_num);
int dmnerf_build_pack_index_split(int ins_num, int32_t* h_idx, int64_t n_idx);
int dmnerf_pack_split(const float* d_flat, const int32_t* d_idx, float* d_stream_words, int64_t n_words, void* stream);
/* OPT-IN split-bf16 data-gradient kernel (training with args.mfma_split): the blob is [the first 1024 floats of the W^T
 * blob (dmnerf_build_pack_index_t: the two VALU heads' table) | W^T split stream]; the index (two int32 per stream word,
 * as above) addresses [flat parameters | F] like dmnerf_build_pack_index_t.  Reads and writes exactly what
 * dmnerf_mlp_bwd_data reads and writes (networks/dm_nerf.py:80-106 backward), f32-class, not its bitwise chain. */
int64_t dmnerf_blob_t_split_words(int ins_num);
int dmnerf_build_pack_index_t_split(int ins_num, int32_t* h_idx, int64_t n_idx);
int dmnerf_mlp_bwd_data_split(const float* d_blob_t_split, int ins_num, const float* d_save, const float* d_graw, int64_t M,
                              float* d_dsave, float* d_graw_t, void* stream);
int dmnerf_mlp_fwd_rays_split(const float* d_blob_split, int ins_num, const float* d_rays_o, const float* d_rays_d,
                              const float* d_z, int64_t N, int S, float* d_raw, void* stream);

/* OPT-IN split-f16 ("f16x2") inference (args.mfma_split = "f16x2"): the same function as dmnerf_mlp_fwd_rays_fused
 * (networks/dm_nerf.py:80-106 with the two activation-free feature linears folded in) on v_mfma_f32_32x32x16_f16 with every f32
 * operand split into two f16 planes and THREE products per f32 product (csrc/mlp_f16_impl.h, split_f16.h): f32-class values
 * (|d raw| <= 1e-5 (1 + |raw|), like the f32 kernels; not their bitwise chain), half the MFMAs of the bf16x3 mode; activations
 * saturate at 65 504 (f16 range).  Blob: dmnerf_blob_f16_words 32-bit words = [4096-float bias table | stream of 16 KiB groups];
 * dmnerf_build_pack_index_f16 fills the table's float gather index (h_idx_tab[4096], for dmnerf_pack_weights) and the stream's
 * element index (h_idx_stream[(words - 4096) * 2]: source parameter | plane << 28, or -1), both addressing the FUSED flat
 * parameter vector (dmnerf_fuse_heads); dmnerf_pack_f16 writes the stream words (n_words = dmnerf_blob_f16_words - 4096). */
int64_t dmnerf_blob_f16_words(int ins_num);
int dmnerf_build_pack_index_f16(int ins_num, int32_t* h_idx_tab, int64_t n_tab, int32_t* h_idx_stream, int64_t n_stream);
int dmnerf_pack_f16(const float* d_flat, const int32_t* d_idx, float* d_stream_words, int64_t n_words, void* stream);
int dmnerf_mlp_fwd_rays_f16(const float* d_blob_f16, int ins_num, const float* d_rays_o, const float* d_rays_d,
                            const float* d_z, int64_t N, int S, float* d_raw, void* stream);

/* OPT-IN training forward on the split-f16 path (args.mfma_split = "f16x2" with grad enabled): dmnerf_mlp_fwd_rays_f16 plus
 * the f32 workspace d_save of dmnerf_mlp_fwd_rays_train (dmnerf_train_save_floats(N S) floats: pe, de, h_0..h_7, g1, g2 as
 * block-major rows + 1-bit ReLU masks), which every backward kernel of this library consumes. */
int dmnerf_mlp_fwd_rays_train_f16(const float* d_blob_f16, int ins_num, const float* d_rays_o, const float* d_rays_d,
                                  const float* d_z, int64_t N, int S, float* d_raw, float* d_save, void* stream);

/* OPT-IN split-f16 data-gradient kernel (training with args.mfma_split = "f16x2"): the blob is [the first 1024 floats of the
 * W^T blob (dmnerf_build_pack_index_t: the two VALU heads' table) | W^T stream of 16 KiB groups, two f16 planes]; the index (two
 * int32 per stream word: source | plane << 28) addresses [flat parameters | F] like dmnerf_build_pack_index_t; dmnerf_pack_f16
 * writes the stream.  Reads and writes exactly what dmnerf_mlp_bwd_data reads and writes (networks/dm_nerf.py:80-106 backward),
 * f32-class, not its bitwise chain.
 */
int64_t dmnerf_blob_t_f16_words(int ins_num);
int dmnerf_build_pack_index_t_f16(int ins_num, int32_t* h_idx, int64_t n_idx);
/* Gradient scaling.  f16 has 5 exponent bits and the data gradients of a real training step sit far below its normal range, so
 * the split-f16 backward runs on 2^s dL/draw (the whole backward is linear in it; s = 6 - ceil(log2 max|dL/draw|), exact) and
 * the weight-gradient reduction multiplies by 2^-s: dmnerf_grad_scale reduces max|d_graw[0..n)| and writes d_scale4 = {2^s,
 * 2^-s, scratch, scratch} (4 floats the caller zeroes ONCE; the kernel leaves the scratch words zero); d_scale = that buffer,
 * or null for s = 0.  With a scale, every row the kernel writes (d_dsave, d_graw_t) carries the factor 2^s. */
int dmnerf_grad_scale(const float* d_graw, int64_t n, float* d_scale4, void* stream);
int dmnerf_mlp_bwd_data_f16(const float* d_blob_t_f16, int ins_num, const float* d_save, const float* d_graw, int64_t M,
                            float* d_dsave, float* d_graw_t, const float* d_scale, void* stream);
/* OPT-IN split-f16 weight-gradient kernel: dmnerf_mlp_bwd_weights on the f16 MFMA (both operands split on the fly into two f16
 * planes, three products per f32 product).  The _f16 plan balances the slices for this kernel's chunk times (every job class is
 * HBM-bound); any of the three plans is valid for any of the three kernels.  The dy-side operands
 * (d_dsave, d_graw_t) carry the factor 2^s of dmnerf_grad_scale; the second stage multiplies every gradient by d_scale[1] = 2^-s
 * (d_scale null: 1). */
int dmnerf_wgrad_plan_sizes_f16(int ins_num, int64_t M, int max_wgs, int64_t* n_job_bytes,
                                int64_t* n_out_bytes, int64_t* part_floats, int* n_jobs, int* n_outs);
int dmnerf_wgrad_plan_f16(int ins_num, int64_t M, int max_wgs, void* h_jobs, int64_t job_bytes,
                          void* h_outs, int64_t out_bytes);
/* Run-time honesty of the f16x2 mode (opt-in diagnostic: DMNERF_CHECK_F16=1 / args.check_f16 in the Python mirror).  The
 * conversions saturate silently at 65 504; every converted operand is also a row of the f32 workspace the training forward saves
 * (gradients = 0) or the data-gradient kernel writes (gradients = 1, values carry the factor 2^s of dmnerf_grad_scale), so one
 * pass over the workspace of M samples finds them: ORs DMNERF_F16_ACT_SATURATED / DMNERF_F16_GRAD_SATURATED into the int32 at
 * d_flags (sticky: the caller zeroes it when it has read it) if any |x| >= 65 504 or non-finite.  No sync. */
#define DMNERF_F16_ACT_SATURATED 1
#define DMNERF_F16_GRAD_SATURATED 2
int dmnerf_f16x2_range_flags(const float* d_workspace, int64_t M, int gradients, int32_t* d_flags, void* stream);
int dmnerf_mlp_bwd_weights_f16(const float* d_save, const float* d_dsave, const float* d_graw_t, int64_t M,
                               const void* d_jobs, int n_jobs, const void* d_outs, int n_outs,
                               const float* d_params_flat, int ins_num, float* d_part, float* d_grad_flat,
                               const float* d_scale, void* stream);

/* ---- evaluator.py (the object-code loss, no host round trip) ---------------------------
 * ins_criterion (networks/evaluator.py:19-74): pred [N, ins_num] (rendered object codes in (0,1)), labels [N]
 * (int32, values 0..ins_num; other values are ignored) -> out4 = {ins_loss_sum, valid_ce, invalid_ce,
 * valid_siou}.  The cost matrices (:57-69), the label -> channel assignment that the reference delegates to
 * scipy.optimize.linear_sum_assignment (:43-54; same optimum, solved on the device by shortest augmenting
 * paths) and the three means (:28-37) run on `stream`; d_work (dmnerf_ins_criterion_work_bytes) carries the
 * assignment to _bwd, which writes d loss / d pred [N, ins_num] for upstream gradients gout4 of the four outputs.
 * ins_num <= 128; at most ins_num distinct labels may occur (the reference's own limit, :21-26).
 * Where the reference RAISES, the stream cannot: more distinct labels than channels (its one-hot column assignment
 * fails, :24) keeps the first ins_num of them, and a label outside [0, ins_num] (its F.one_hot / indexing fails, :23)
 * joins no row.  Both conditions are recorded in an int32 flags word inside d_work, at byte offset
 * dmnerf_ins_criterion_flags_offset(N, ins_num), valid once _fwd has run on the stream: a caller that wants the
 * reference's behaviour reads it back and raises (the Python mirror does under check=True / DMNERF_CHECK_LABELS=1). */
#define DMNERF_CRIT_TOO_MANY_LABELS 1
#define DMNERF_CRIT_LABEL_RANGE 2
int64_t dmnerf_ins_criterion_work_bytes(int64_t N, int ins_num);
int64_t dmnerf_ins_criterion_flags_offset(int64_t N, int ins_num);
int dmnerf_ins_criterion_fwd(const float* d_pred, const int32_t* d_labels, int64_t N, int ins_num, void* d_work,
                             int64_t work_bytes, float* d_out4, void* stream);
int dmnerf_ins_criterion_bwd(const float* d_pred, const int32_t* d_labels, int64_t N, int ins_num, const void* d_work,
                             const float* d_gout4, float* d_grad_pred, void* stream);
/* The same two calls for TWO predictions against the same labels -- the fine and the coarse level of a training step
 * (train_dmsr.py:38-47) -- in one launch per kernel; each with its own work buffer (work_bytes each) and outputs. */
int dmnerf_ins_criterion_fwd2(const float* d_pred_a, const float* d_pred_b, const int32_t* d_labels, int64_t N, int ins_num,
                              void* d_work_a, void* d_work_b, int64_t work_bytes, float* d_out4_a, float* d_out4_b, void* stream);
int dmnerf_ins_criterion_bwd2(const float* d_pred_a, const float* d_pred_b, const int32_t* d_labels, int64_t N, int ins_num,
                              const void* d_work_a, const void* d_work_b, const float* d_gout4_a, const float* d_gout4_b,
                              float* d_grad_a, float* d_grad_b, void* stream);

/* ---- the scalar tail of the training step (train_dmsr.py:33-61; extension: the reference forms it from tensor operations) ----
 * loss = sum over the levels a (fine), b (coarse) of img2mse(rgb, target) (evaluator.py:11) + ins_criterion (out4[0] of the
 * calls above; nullable) + emptiness penalizer (penalizer.py:43-55 from the four batch sums of dmnerf_penalizer_sums / _sums2;
 * nullable), added in f32 in the training loop's order.  _fwd: d_terms8 = {mse_a, crit_a, pen_a, mse_b, crit_b, pen_b, total, 0},
 * d_pen_inv4 = the penalizer's two normalisers per level.  _bwd, for the upstream scalar d_g_total: d loss / d rgb of both
 * levels ((g / 3N) * (2 (rgb - target)), the products autograd forms), d_gout8 = {g, 0, 0, 0} x 2 for dmnerf_ins_criterion_bwd2
 * and d_pen_scales4 = d_pen_inv4 * g for dmnerf_penalizer_bwd; d_g_partials8 (nullable) = the same factors as the 4-double rows
 * {scale_b, 0, scale_m, 0} per level that dmnerf_composite_pen_bwd reads.  One launch each. */
int dmnerf_loss_tail_fwd(const float* d_rgb_a, const float* d_rgb_b, const float* d_target, int64_t N,
                         const float* d_crit_out4_a, const float* d_crit_out4_b, const double* d_pen_sums4_a,
                         const double* d_pen_sums4_b, int C, float* d_terms8, float* d_pen_inv4, void* stream);
int dmnerf_loss_tail_bwd(const float* d_rgb_a, const float* d_rgb_b, const float* d_target, int64_t N,
                         const float* d_g_total, const float* d_pen_inv4, float* d_grad_rgb_a, float* d_grad_rgb_b,
                         float* d_gout8, float* d_pen_scales4, double* d_g_partials8, void* stream);

/* ---- manipulator.py (scene editing at render time) -----------------------------------
 * manipulator_render (networks/manipulator.py:86-105): render_train whose object map keeps all C channels
 * (d_ins_map [N,C]).  z_val_lerp: the grid of manipulator_nerf (:117-119), near (1-t) + far t.
 * sort_rows: torch.sort(x, -1).values per row (K <= 2048), exact permutation (:191,:195).
 * exchanger (:18-83): per-sample / per-ray argmax labels and the masked swaps of raw rows, d_ori_raw modified in
 * place; h_tar_raws / h_tar_accs: HOST arrays of T device pointers ([N,S,4+C] / [N,C]); h_labels: T moved labels;
 * outputs (nullable): int64 labels [N,S] of the original rays and of the last target.                 */
int dmnerf_manipulator_render(const float* d_raw, const float* d_z, const float* d_rays_d, int64_t N, int S,
                              int C, float* d_rgb_map, float* d_weights, float* d_depth_map,
                              float* d_ins_map, void* stream);
int dmnerf_z_val_lerp(const float* d_t, float near_, float far_, int64_t N, int S, float* d_z, void* stream);
int dmnerf_sort_rows(const float* d_in, int64_t N, int K, float* d_out, void* stream);
int dmnerf_exchanger(float* d_ori_raw, const float* const* h_tar_raws, const float* d_ori_acc,
                     const float* const* h_tar_accs, const int* h_labels, int T, int64_t N, int S, int C,
                     int64_t* d_ori_label, int64_t* d_tar_label, void* stream);

/* ---- evaluator.py ins_eval, the part every frame needs (networks/evaluator.py:127-137) ----------------
 * d_label [N] int64 = argmax over the C object channels of d_ins [N,C] (first maximum, like torch.argmax on CPU);
 * d_conf [N] (nullable) = that maximum (np.max(pred_ins, -1)).                                                    */
int dmnerf_ins_label_conf(const float* d_ins, int64_t N, int C, int64_t* d_label, float* d_conf, void* stream);

/* ---- penalizer.py (the consumer of raw / z_vals / depth) ---------------------------
 * emptiness_penalizer (networks/penalizer.py:5-55) fused: _fwd writes per-ray partial sums
 * d_partials [N,4] (double): {sum BCE*w_before, sum m_before, sum loss_middle*w_middle, sum m_middle};
 *   loss = S0 / (C max(S1,1e-8)) + S2 / max(S3,1e-8) with Sk = sum over rays.
 * _bwd writes dL/draw [N,S,4+C] (zero in channels 0..3) given d_scales = {up/(C max(S1,1e-8)), up/max(S3,1e-8)}.
 * two_deta_w_sq = 2*deta_w^2 and gauss_norm = 0.4*sqrt(2 pi), both rounded to float32 as the reference forms them.
 * depth is a constant (the reference detaches it, penalizer.py:59).                                   */
int dmnerf_penalizer_fwd(const float* d_raw, const float* d_z, const float* d_depth, const float* d_rays_d,
                         int64_t N, int S, int C, float tolerance, float two_deta_w_sq, float gauss_norm,
                         double* d_partials, void* stream);
int dmnerf_penalizer_bwd(const float* d_raw, const float* d_z, const float* d_depth, const float* d_rays_d,
                         int64_t N, int S, int C, float tolerance, float two_deta_w_sq, float gauss_norm,
                         const float* d_scales, float* d_grad_raw, void* stream);
/* The scalar tail of the same function (penalizer.py:44-55) without a chain of scalar tensor operations: _sums reduces the per-ray
 * partials to d_sums4 = {S0, S1, S2, S3} (doubles; one block, fixed order; a ray-sharded caller all-reduces them here), _finish
 * writes d_loss1 = (float)(S0 / (C max(S1,1e-8)) + S2 / max(S3,1e-8)) and d_inv2 = {1 / (C max(S1,1e-8)), 1 / max(S3,1e-8)}
 * (float; d_scales of _bwd = d_inv2 * upstream gradient). */
int dmnerf_penalizer_sums(const double* d_partials, int64_t N, double* d_sums4, void* stream);
/* render_train AND the penalizer's per-ray partial sums in one pass over the ray (extension, SURVEY 8(f)-1: the penalizer's only
 * other input is the ray's own depth, which the compositing pass has just produced): the outputs of dmnerf_composite_fwd plus
 * d_partials [N,4] of dmnerf_penalizer_fwd called with that depth map -- bit-equal to the two calls.  _bwd: dmnerf_composite_bwd
 * plus the penalizer's gradient (dmnerf_penalizer_bwd's values) ADDED in the same pass; d_g_partials = d loss / d partials of ONE
 * ray, 4 doubles, the same for every ray ({up / (C max(sum m_b, 1e-8)), -, up / max(sum m_m, 1e-8), -}: the normalisers are
 * batch sums).  One kernel and one write of d raw instead of two kernels, two writes and an elementwise add. */
int dmnerf_composite_pen_fwd(const float* d_raw, const float* d_z, const float* d_rays_d, int64_t N, int S, int C,
                             float tolerance, float two_deta_w_sq, float gauss_norm, float* d_rgb_map, float* d_weights,
                             float* d_depth_map, float* d_ins_map, double* d_partials, void* stream);
int dmnerf_composite_pen_bwd(const float* d_raw, const float* d_z, const float* d_rays_d, const float* d_ins_map,
                             const float* d_depth_map, const float* d_g_rgb, const float* d_g_ins, const float* d_g_depth,
                             const float* d_g_weights, const double* d_g_partials, int64_t N, int S, int C, float tolerance,
                             float two_deta_w_sq, float gauss_norm, float* d_grad_raw, void* stream);
/* ... of two levels in one launch: d_sums8 = the four sums of a, then of b. */
int dmnerf_penalizer_sums2(const double* d_partials_a, int64_t N_a, const double* d_partials_b, int64_t N_b, double* d_sums8,
                           void* stream);
int dmnerf_penalizer_finish(const double* d_sums4, int C, float* d_loss1, float* d_inv2, void* stream);

/* dm_nerf inference (networks/render.py:31-96, perturb handled by the caller passing t_rand/u):
 * all stages on `stream`, outputs = the 10 tensors of the reference dict (ins_* are [N, C-1]).
 *   d_t_rand: [N,S] or NULL (no jitter); d_u / u_row_stride as in dmnerf_sample_pdf.
 * Scratch: d_weights_ws [N, S+n_imp] floats.                                                  */
typedef struct {
    const float* d_blob_coarse;
    const float* d_blob_fine;
    int ins_num;
    const float* d_rays_o;
    const float* d_rays_d;
    const float* d_z_in;        /* [N,S] coarse depths before jitter */
    const float* d_t_rand;      /* nullable */
    const float* d_u;
    int64_t u_row_stride;
    int64_t N;
    int S;
    int n_imp;
    /* outputs */
    float* d_z_coarse;          /* [N,S]        'z_vals_coarse' (jittered) */
    float* d_raw_coarse;        /* [N,S,4+C]    'raw_coarse'   */
    float* d_rgb_coarse;        /* [N,3]        'rgb_coarse'   */
    float* d_depth_coarse;      /* [N]          'depth_coarse' */
    float* d_ins_coarse;        /* [N,C-1]      'ins_coarse'   */
    float* d_z_fine;            /* [N,S+n_imp]  'z_vals_fine'  */
    float* d_raw_fine;          /* [N,S+n_imp,4+C] 'raw_fine'  */
    float* d_rgb_fine;          /* [N,3]        'rgb_fine'     */
    float* d_depth_fine;        /* [N]          'depth_fine'   */
    float* d_ins_fine;          /* [N,C-1]      'ins_fine'     */
    float* d_weights_ws;        /* scratch [N, S+n_imp] */
    /* optional hipEvent_t pair recorded on `stream` around the fine-network MLP launch (the dominant
     * kernel), so a caller can time it live without splitting the call; NULL = not recorded */
    void* ev_fine_mlp_begin;
    void* ev_fine_mlp_end;
    /* 1: d_blob_coarse / d_blob_fine are fused-heads blobs (dmnerf_build_pack_index_fused); 2: split-bf16 blobs; 3: split-f16 blobs */
    int fused_heads;
} dmnerf_render_args;
int dmnerf_render_rays_fwd(const dmnerf_render_args* args, void* stream);

/* ---- network shapes other than D = 8, W = 256, multires 10 / 4 (config.py:126-138 passes args.netdepth / netwidth /
 * multires* through; no shipped config changes them): the layer-by-layer path of csrc/generic.hip.  One strided f32-MFMA
 * GEMM serves the three products of a linear layer; the Python mirror (dm_nerf_amd/generic.py) chains them as
 * DM_NeRF.forward does (networks/dm_nerf.py:80-106) and as its autograd would.
 *   dmnerf_gemm:   C[i*ldc + j] (+)= sum_k A[i*sai + k*sak] * B[k*sbk + j*sbj]   (+ bias[j], ReLU, * [mask[i*ldm + j] > 0]);
 *                  splits > 1: split-K over k with a workspace of splits*I*J floats, partials added in slice order
 *                  (then no bias / relu / mask).
 *   dmnerf_colsum: out[j] = sum_m X[m*ldx + j] (bias gradient), workspace slices*J floats.
 *   dmnerf_ray_points: pts [N*S,3] = o + d z, dirs [N*S,3] = d / |d| per sample (render.py:37,49-57).
 *   dmnerf_copy_cols:  dst[m*ld_dst + c] = src[m*ld_src + c], c < n (the cat of dm_nerf.py:87,90 into a column slice). */
int dmnerf_gemm(const float* d_A, int64_t sai, int64_t sak, const float* d_B, int64_t sbk, int64_t sbj, float* d_C, int64_t ldc,
                int64_t I, int J, int64_t K, const float* d_bias, int relu, const float* d_mask, int64_t ldm, int accumulate,
                float* d_ws, int splits, void* stream);
int dmnerf_colsum(const float* d_X, int64_t ldx, int64_t M, int J, float* d_out, float* d_ws, int slices, void* stream);
int dmnerf_ray_points(const float* d_rays_o, const float* d_rays_d, const float* d_z, int64_t N, int S, float* d_pts,
                      float* d_dirs, void* stream);
int dmnerf_copy_cols(const float* d_src, int64_t ld_src, float* d_dst, int64_t ld_dst, int64_t M, int n, void* stream);

/* The same path's forward and data-gradient products on the machine's own terms (csrc/gemm_nt.hip, ABI 7): both operands k-fast,
 * global -> LDS by LDS-DMA into a swizzled ring, ds_read_b128 operand reads, one workgroup = 128 samples x all outputs of the layer.
 *   dmnerf_pack_nt: the layer's weight matrix in the kernel's form -- d_out [rows_pad][ldb], rows_pad = 32 x dmnerf_gemm_nt_blocks(n)
 *     x tiles, ldb = 32 (ceil(k0 / 32) + ceil(k1 / 32)), zero-filled padding: row n = source row n, columns [c0, c0 + k0) then (from
 *     column 32 ceil(k0 / 32)) [c1, c1 + k1) -- the two K ranges of a layer whose input is a cat (dm_nerf.py:87 [h, pts], :90
 *     [rgb_feature, dirs]); transposed != 0: the source is read as W^T (row n = source COLUMN n, column k = source ROW c + k): the
 *     data-gradient form.  d_bias_out [rows_pad] (nullable) = the bias, zero beyond n_rows.
 *   dmnerf_gemm_nt: C[m*ldc + n] = act(sum_k A(m, k) Wp[n][k] + bias[n]) for m < M, n < n_store, zeros in columns [n_store, n_zero)
 *     (the pad columns that keep the NEXT layer's operand rows 16-byte aligned); A(m, k) = A0[m*lda0 + k] for k < k0, then
 *     A1[m*lda1 + (k - k0)] (k1 = 0: one range); lda0 / lda1 multiples of 4, base pointers 16-byte aligned, a*_floats = floats from the
 *     pointer to the end of its allocation (the kernel's descriptor bound: a row's tail may be read past k0 into the next row -- those
 *     products meet zero weights -- but never past the allocation).  relu; d_mask (nullable): C = mask[m*ldm + n] > 0 ? v : 0 (the
 *     ReLU derivative of the data gradient); accumulate: C += before relu / mask.  Bias is the accumulator's initial value.
 *   dmnerf_ray_embed: pts = o + d z, viewdirs = d / |d| per sample (render.py:37,49-57) and Embedder.embed of both (dm_nerf.py:37-38)
 *     into row-padded buffers x_pos [N*S][ldp], x_dir [N*S][ldv], pad columns zeroed.
 *   dmnerf_copy_cols_pad: dst[m*ld_dst + c] = c < n ? src[m*ld_src + c] : 0 for c < n_pad -- a column slice as such an operand.      */
int dmnerf_gemm_nt_blocks(int n_out);
int dmnerf_pack_nt(const float* d_W, int64_t ldw, int n_rows, int c0, int k0, int c1, int k1, int transposed, const float* d_bias,
                   float* d_out, int rows_pad, int ldb, float* d_bias_out, void* stream);
int dmnerf_gemm_nt(const float* d_A0, int64_t lda0, int64_t a0_floats, int k0, const float* d_A1, int64_t lda1, int64_t a1_floats, int k1,
                   const float* d_B, int64_t b_floats, int ldb, const float* d_bias, float* d_C, int64_t ldc, int n_store, int n_zero,
                   int64_t M, int relu, const float* d_mask, int64_t ldm, int accumulate, void* stream);
/* The same path's WEIGHT gradient (csrc/gemm_tn.hip, ABI 8): dW[i*ldw + j] = sum_m dy[m*ldy + i] x[m*ldx + j] for i < n_out, j < n_in
 * and (d_db non-null) db[i] = sum_m dy[m*ldy + i] -- both operands sample-major rows as gemm_nt reads and writes them (ldy / ldx multiples
 * of 4, base pointers 16-byte aligned, *_floats = floats from the pointer to the end of its allocation), 32-sample chunks through an
 * LDS-DMA ring, split over the samples with a workspace of dmnerf_gemm_tn_ws_floats(n_out, n_in, M) floats whose partials are added in
 * slice order (deterministic).  Replaces dmnerf_gemm's splits > 1 form + dmnerf_colsum on this path (those remain for operands whose
 * rows are not 16-byte aligned).  torch.autograd of nn.Linear in the reference: dm_nerf.py:66-83 layers under train_dmsr.py:62-64.   */
int64_t dmnerf_gemm_tn_ws_floats(int n_out, int n_in, int64_t M);
int dmnerf_gemm_tn(const float* d_dy, int64_t ldy, int64_t dy_floats, int n_out, const float* d_x, int64_t ldx, int64_t x_floats, int n_in,
                   int64_t M, float* d_dW, int64_t ldw, float* d_db, float* d_ws, int64_t ws_floats, void* stream);
/* The TRUNK of a narrow network (width 32 .. 160, a multiple of 32) as ONE launch in inference (csrc/gemm_chain.hip): the activations
 * of a 128-sample tile stay in LDS from layer to layer, the weights of all layers stream through an LDS ring.  Layer l: h_l = relu?(
 * [h_{l-1} if from_act | x if from_x] W_l^T + b_l) with d_B packed by dmnerf_pack_nt (range 0 = the `width` columns of h, range 1 = the
 * x_cols columns of the encoding; layer 0: from_x only), d_bias [width]; d_x = the row-padded encoding [M][ldx] of dmnerf_ray_embed,
 * d_out = the last layer's output [M][ldo].  dmnerf_mlp_chain_supported: whether (width, x_cols) fits the CU's LDS.                    */
#define DMNERF_CHAIN_MAX_LAYERS 16
typedef struct dmnerf_chain_layer {
    const float* d_B;
    const float* d_bias;
    int ldb;          /* 32 x (width / 32 if from_act) + 32 x ceil(x_cols / 32) if from_x) */
    int from_act, from_x, relu;
} dmnerf_chain_layer;
int dmnerf_mlp_chain_supported(int width, int x_cols);
int dmnerf_mlp_chain(const float* d_x, int64_t ldx, int64_t x_floats, int x_cols, const dmnerf_chain_layer* layers, int n_layers,
                     int width, float* d_out, int64_t ldo, int64_t M, void* stream);
int dmnerf_copy_cols_pad(const float* d_src, int64_t ld_src, float* d_dst, int64_t ld_dst, int64_t M, int n, int n_pad, void* stream);
int dmnerf_ray_embed(const float* d_rays_o, const float* d_rays_d, const float* d_z, int64_t N, int S, int Lp, int Lv,
                     float* d_x_pos, int ldp, float* d_x_dir, int ldv, void* stream);

/* ---- EXTENSION: the optimizer update and the weight re-packing of a training step as two launches (csrc/optim.hip).
 * The reference steps torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr, betas=(0.9, 0.999))
 * (train_dmsr.py:124-125) after total_loss.backward() (:62-64); the drop-in path keeps exactly that.  These two entries are what
 * dm_nerf_amd.optim.FlatAdam runs instead when a caller opts in: at small per-rank batches the ~9 multi-tensor launches of the
 * torch optimizer and the 8 pack / cat launches after it are a measurable part of the step.
 *   dmnerf_adam_step: ONE pass over flat f32 vectors of n elements -- parameters, gradients (the gradient arena the
 *     weight-gradient kernels wrote), first and second moments -- applying torch.optim.Adam's update (amsgrad off, no weight
 *     decay) operation for operation, each rounded to f32 on its own, scalar factors formed in double:
 *       m <- m + (1-b1)(g - m);  v <- v b2;  v <- v + (1-b2)(g g);  p <- p + (-(lr / (1 - b1^t))) (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
 *     t = d_state2[0] + 1 is read on the device and stored back by the last workgroup to finish (d_state2[1] is its ticket
 *     counter; both start at 0): no host synchronisation, HIP-graph capturable.  d_lr (nullable): a device f32 scalar that
 *     overrides lr (a captured graph changes the learning rate by writing it).
 *   dmnerf_repack_train: for up to DMNERF_REPACK_MAX_MODELS models in one launch, from each model's (updated) flat parameter
 *     vector: d_flat_copy (nullable; dmnerf_param_count floats, must not alias the source), the forward blob (what
 *     dmnerf_pack_weights gathers with dmnerf_build_pack_index) and the W^T blob (dmnerf_build_pack_index_t over
 *     [parameters | F], F = A . rgb_feature_linear.weight formed inline with dmnerf_head_product's fmaf chain): bit-identical to
 *     dmnerf_head_product + two dmnerf_pack_weights calls.                                                                  */
#define DMNERF_REPACK_MAX_MODELS 4
typedef struct dmnerf_repack_model {
    const float* d_params_flat;
    int ins_num;
    float* d_flat_copy;
    const int32_t* d_idx;
    float* d_blob;
    const int32_t* d_idx_t;
    float* d_blob_t;
    const int32_t* d_f_pos;   /* nullable; 32768 entries: position in d_blob_t of F[i][j] (i * 256 + j), i.e. the inverse of d_idx_t on
                               * its entries >= dmnerf_param_count (each occurs once).  Given, F is formed in its natural order
                               * (coalesced) and scattered; NULL: every blob slot that holds an F element computes it itself. */
} dmnerf_repack_model;
int dmnerf_adam_step(float* d_params, const float* d_grads, float* d_exp_avg, float* d_exp_avg_sq, int64_t n,
                     double lr, const float* d_lr, double beta1, double beta2, double eps, int64_t* d_state2, void* stream);
int dmnerf_repack_train(const dmnerf_repack_model* models, int n_models, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMNERF_HIP_H */
