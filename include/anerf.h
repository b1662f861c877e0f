/*
 * anerf.h -- C ABI of the MI355X-native A-NeRF ray-march hot path (libanerf_hip.so).
 *
 * Drop-in boundary for the reference's caster call
 *     ray_caster(rays_flat[i:i+chunk], **batch_kwargs)          core/trainer.py:70-72
 *     RayCaster.forward / render_rays                           core/raycasters.py:349-474
 *     NeRF.forward / forward_batchify / raw2outputs             core/networks/nerf.py:90,133,150
 *     get_near_far_in_cylinder / sample_from_lineseg /
 *     isample_from_lineseg / sample_pdf                         core/utils/ray_utils.py:157-344
 *     CutoffEmbedder._embed, transform_batch_pts/rays,
 *     RelDistEncoder, VecNormEncoder                            core/cutoff_embedder.py:111, core/encoders.py:8-193
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless the name says host.  The library never allocates,
 *     frees or retains device memory: the caller owns all inputs, outputs and workspaces.
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises the
 *     device or reads device memory on the host.
 *   - Return value: 0 = ok, negative = error (see ANERF_E_*).  No exceptions cross the ABI.
 *   - Re-entrant, no global mutable state.
 *   - fp32 everywhere; `sorted_idx` is int64 as in torch.sort.
 */
#ifndef ANERF_H
#define ANERF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANERF_OK 0
#define ANERF_E_CONFIG (-1)   /* configuration outside the built template set            */
#define ANERF_E_SHAPE (-2)    /* bad sizes (N_samples < 8, > 512, ...)                   */
#define ANERF_E_NULL (-3)     /* required pointer is NULL                                */
#define ANERF_E_WORKSPACE (-4)/* workspace too small                                     */
#define ANERF_E_LAUNCH (-5)   /* hipLaunch failed (hipGetLastError text via anerf_last_error) */

/* Static configuration of the path; reference: create_raycaster(), core/raycasters.py:17-184. */
typedef struct AnerfConfig {
  int32_t n_joints;        /* 24 (SMPL)                                                   */
  int32_t multires;        /* 7   -> 24*(1+14)  = 360 distance-PE channels                */
  int32_t multires_views;  /* 4   -> 72*(1+8)   = 648 view-PE channels (0 -> 72)          */
  int32_t framecode_ch;    /* 0, or 16 with opt_framecode (core/networks/embedding.py)    */
  int32_t netdepth;        /* 8                                                           */
  int32_t netwidth;        /* 256                                                         */
  int32_t skip;            /* 4                                                           */
  int32_t density_act;     /* 0 = relu, 1 = softplus(x - softplus_shift)  raycasters.py:230 */
  float density_scale;     /* B in raw2outputs, nerf.py:150                               */
  float softplus_shift;
  int32_t cutoff_bones;    /* ABI revision 5.  != 0: --cutoff_bones (raycasters.py:54-57 with multires_bones = 0): the bone-direction
                            * block r [72] is gated too, r_j * w_j with the distance gate w_j = 1 - sigmoid(tau_v * (v_j - cutoff_v[j]))
                            * (the bone embedder is a CutoffEmbedder fed the same distances, with the same tau and cutoff) */
} AnerfConfig;

/* One network's parameters in the reference's state_dict order (torch Linear [out,in] row-major):
 *   w[0..7]  pts_linears.{0..7}.weight   w[8] alpha_linear  w[9] feature_linear
 *   w[10]    views_linears.0             w[11] rgb_linear   ;  b[] likewise.
 *   codes    framecodes.codes.weight [n_codes, framecode_ch] or NULL.                      */
typedef struct AnerfNetParams {
  const float* w[12];
  const float* b[12];
  const float* codes;
  int32_t n_codes;
  /* ABI revision 5: the embedders' frequency schedule (--freq_schedule; CutoffEmbedder.get_schedule_w, core/cutoff_embedder.py:
   * 159,191-197: every sin / cos band k of the encoding is multiplied by w_k(alpha) = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2
   * before it reaches the network) as per-column factors FOLDED INTO THE WEIGHT IMAGES: W (s * e) = (W diag(s)) e, so the kernels
   * keep producing the unscaled encoding e and the pack kernels multiply the columns that consume it -- sched_x [sched_dim_x =
   * distance block + bone block, torch column order of pts_linears.0's input] for pts_linears.0 and the input columns of the skip
   * layer pts_linears.5, sched_u [sched_dim_u = view block (+ frame-code columns)] for the columns of views_linears.0 behind its
   * 256 feature columns.  DEVICE pointers, NULL (with dim 0) = no schedule: every factor 1, the images of revisions <= 4. */
  int32_t sched_dim_x, sched_dim_u, reserved_;
  const float *sched_x, *sched_u;
} AnerfNetParams;

/* Sizes of the packed weight images the kernels consume (see DESIGN.md "weight stream"). */
typedef struct AnerfLayout {
  int64_t stream_floats;   /* MFMA-fragment-ordered weights, multiple of 8192 (one 32 KiB stage)  */
  int64_t aux_floats;      /* biases + alpha/rgb head weights, natural order                      */
  int32_t n_stages;        /* stream_floats / 8192                                                */
  int32_t x_width;         /* MLP input width (1080, 1081 with frame code column, 504 ...)        */
} AnerfLayout;

const char* anerf_last_error(void);
int anerf_version(void);

/* which: 0 = forward image (W), 1 = backward-data image (W^T of the hidden trunk, feature and view layers),
 * 2 = input-gradient image (W^T of the encoded-input columns of pts_linears.0/.5 and views_linears.0),
 * 3 = bf16x3 forward image (hi/lo bf16 pairs; see anerf_mlp_raw_b3). */
int anerf_layout(const AnerfConfig* cfg, int which, AnerfLayout* out);
/* HOST: fill table[stream_floats + aux_floats]: entry = (tensor_id << 24) | element offset, -1 = 0.0f;
 * tensor_id = index into {w[0..11], b[0..11]} (0..23).  Upload once per config. */
int anerf_build_pack_table(const AnerfConfig* cfg, int which, int32_t* host_table);
/* Gather the parameters into the packed image: out[i] = table[i] < 0 ? 0 : tensor[id][off]. */
int anerf_pack_params(const AnerfNetParams* params, const int32_t* table, int64_t n, float* out, void* stream);

/* A2: get_near_far_in_cylinder (ray_utils.py:292-344).  rays [N, ray_stride] = (o3,d3,near,far,...),
 * cyls [N,5].  near_far [N,2]; stats_ws: 32 bytes of 8-byte-aligned scratch (sum_near, sum_far in 2^-32 fixed point,
 * cnt_near, cnt_far as four 64-bit integers: exact, order-independent accumulation), zeroed by this call.  Rows whose ray misses the circle take the nan-mean of this call's rays. */
int anerf_ray_bounds(const float* rays, int32_t ray_stride, const float* cyls, int32_t n_rays,
                     float* near_far, float* stats_ws, void* stream);
/* A3: sample_from_lineseg (ray_utils.py:204-251) on the bounds of anerf_ray_bounds, NaN rows patched with the
 * call's nan-mean (or the placeholder bounds in rays[:,6:8] when every row is NaN).  t_rand [N,S] or NULL
 * (perturb == 0).  z_vals [N,S]; near_far_fixed [N,2] optional (the bounds actually used). */
int anerf_coarse_z(const float* near_far, const float* stats_ws, const float* rays, int32_t ray_stride,
                   int32_t n_rays, int32_t n_samples, const float* t_rand, int32_t lindisp, float* z_vals,
                   float* near_far_fixed, void* stream);

/* A4-A9 fused: per sample, world->bone transform, skeleton-relative features, cutoff PE and the whole
 * MLP; raw [N*S,4] = (r,g,b logits, sigma logit).  skts [N,24,4,4] (skt_ray_stride = 384) or one shared
 * pose (skt_ray_stride = 0).  cam_idx [N] float or NULL.  packed/aux from anerf_pack_params(which=0). */
int anerf_mlp_raw(const AnerfConfig* cfg, const float* packed, const float* aux,
                  const float* rays, int32_t ray_stride, const float* z_vals,
                  const float* skts, int64_t skt_ray_stride, const float* cam_idx,
                  const float* codes, int32_t n_codes,
                  float tau_v, float tau_d, const float* cutoff_v, const float* cutoff_d,
                  int32_t n_rays, int32_t n_samples, float* raw, void* stream);

/* A9 alone (NeRF.forward seam, nerf.py:133): x [P, x_width] already encoded -> raw [P,4]. */
int anerf_mlp_forward(const AnerfConfig* cfg, const float* packed, const float* aux,
                      const float* x, int64_t n_points, const float* codes, int32_t n_codes,
                      float* raw, void* stream);

/* A10: raw2outputs (nerf.py:150-205).  noise [N,S] or NULL (added to raw_sigma / B before the activation).
 * Outputs: rgb_map [N,3], disp_map [N], acc_map [N], weights [N,S], alpha [N,S]; depth_map [N] optional. */
int anerf_composite(const AnerfConfig* cfg, const float* raw, const float* z_vals, const float* rays,
                    int32_t ray_stride, const float* noise, int32_t n_rays, int32_t n_samples,
                    float* rgb_map, float* disp_map, float* acc_map, float* weights, float* alpha,
                    float* depth_map, void* stream);

/* A11: isample_from_lineseg + sample_pdf + sort (ray_utils.py:157-201,255-289).  u [N,Ni] or NULL
 * (deterministic linspace).  z_samples [N,Ni], z_merged [N,S+Ni], sorted_idx [N,S+Ni] int64 or NULL. */
int anerf_importance(const float* z_vals, const float* weights, int32_t n_rays, int32_t n_samples,
                     int32_t n_importance, const float* u, int32_t single_net,
                     float* z_samples, float* z_merged, int64_t* sorted_idx, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Training path (A12: autograd of A9/A10 -- Trainer.optimize -> loss.backward(), core/trainer.py:451-483).
 * forward:  anerf_mlp_raw_train (= anerf_mlp_raw + saved activations)
 * backward: anerf_composite_backward -> anerf_mlp_backward -> anerf_weight_grads
 * ------------------------------------------------------------------------------------------------------------- */

/* Activations the training forward saves for the backward; caller-allocated row-major planes of p_pad rows
 * (p_pad = multiple of 128 >= N*S; rows >= N*S must be ZERO).  x/u are in "stream column order" (see
 * anerf_build_perm_tables). */
typedef struct AnerfSaved {
  float* h;       /* [8][p_pad][256]   h0..h7, post-ReLU                       */
  float* f;       /* [p_pad][256]      feature_linear output                    */
  float* g;       /* [p_pad][128]      views_linears.0 output, post-ReLU        */
  float* x;       /* [p_pad][432]      density-net input                        */
  float* u;       /* [p_pad][u_width]  view-net input without the feature part  */
  int64_t p_pad;
} AnerfSaved;

/* Gradient tensors, same order and shapes as AnerfNetParams (torch layout); all written (not accumulated). */
typedef struct AnerfNetGrads {
  float* w[12];
  float* b[12];
  /* ABI revision 5: the factors the weight images were packed with (AnerfNetParams.sched_x / sched_u; [dim_x] / [u_width] DEVICE
   * floats or NULL).  d loss / d W = (d loss / d (W diag(s))) diag(s): the reduction multiplies the same gradient columns. */
  const float *sched_x, *sched_u;
} AnerfNetGrads;

typedef struct AnerfTrainLayout {
  int64_t p_pad;          /* rows per saved plane for n_points                          */
  int32_t x_width;        /* 432                                                         */
  int32_t u_width;        /* 648 (+16 with frame codes), 72 for multires_views = 0       */
  int32_t gemm_chunks;    /* p-chunks the weight-gradient GEMM splits the sample axis in */
  int64_t gemm_ws_floats; /* workspace floats anerf_weight_grads needs                   */
} AnerfTrainLayout;

int anerf_train_layout(const AnerfConfig* cfg, int64_t n_points, AnerfTrainLayout* out);
/* HOST: perm_x[432], perm_u[u_width]: stream column -> torch input column (of pts_linears.0 / of
 * views_linears.0 minus 256). */
int anerf_build_perm_tables(const AnerfConfig* cfg, int32_t* perm_x, int32_t* perm_u);

int anerf_mlp_raw_train(const AnerfConfig* cfg, const float* packed, const float* aux,
                        const float* rays, int32_t ray_stride, const float* z_vals,
                        const float* skts, int64_t skt_ray_stride, const float* cam_idx,
                        const float* codes, int32_t n_codes,
                        float tau_v, float tau_d, const float* cutoff_v, const float* cutoff_d,
                        int32_t n_rays, int32_t n_samples, float* raw, const AnerfSaved* saved, void* stream);

/* d(loss)/d(outputs of anerf_composite) -> draw [N,S,4].  Any of g_acc, g_disp, g_alpha, g_weights may be NULL. */
int anerf_composite_backward(const AnerfConfig* cfg, const float* raw, const float* z_vals, const float* rays,
                             int32_t ray_stride, const float* noise, int32_t n_rays, int32_t n_samples,
                             const float* g_rgb, const float* g_acc, const float* g_disp, const float* g_alpha,
                             const float* g_weights, float* draw, void* stream);

/* draw [P,4] -> d(pre-activations): dz [8][p_pad][256], df [p_pad][256], dzv [p_pad][128] (rows >= P untouched:
 * caller zeroes them).  packed_t: image which=1 (W^T); aux: the forward aux image. */
int anerf_mlp_backward(const AnerfConfig* cfg, const float* packed_t, const float* aux, const float* draw,
                       const AnerfSaved* saved, float* dz, float* df, float* dzv, int64_t n_points, void* stream);

/* Weight and bias gradients of all 12 Linear layers: one grouped fp32-MFMA GEMM over the sample axis + a
 * deterministic chunk reduction.  perm_x / perm_u: DEVICE copies of anerf_build_perm_tables. */
int anerf_weight_grads(const AnerfConfig* cfg, const AnerfSaved* saved, const float* dz, const float* df,
                       const float* dzv, const float* draw, int64_t n_points, const int32_t* perm_x,
                       const int32_t* perm_u, const AnerfNetGrads* grads, float* workspace, int64_t ws_floats,
                       void* stream);

/* ---- pose optimisation / frame codes (A12: d(loss)/d(skts), d(loss)/d(framecodes)) --------------------------------
 * anerf_layout / anerf_build_pack_table / anerf_pack_params with which=2 give the input-gradient weight image. */

/* dz planes + dzv (from anerf_mlp_backward) -> gradients w.r.t. the encoded inputs in stream column order:
 * dx [p_pad][432], du [p_pad][u_width]. */
int anerf_input_grads(const AnerfConfig* cfg, const float* packed_i, const float* dz, const float* dzv,
                      int64_t p_pad, int64_t n_points, float* dx, float* du, void* stream);

/* Backward of the fused encoding: dx/du -> dskts [N,24,4,4] (every element written: rows 0..2 the gradient, row 3 zeros; no zero
 * fill needed -- round 6; an earlier revision left row 3 to the caller).
 * skts must be per ray (stride 384).  dy_ws, dq_ws: scratch [N*S][72] each. */
int anerf_encode_backward(const AnerfConfig* cfg, const float* dx, const float* du, const float* rays,
                          int32_t ray_stride, const float* z_vals, const float* skts, int64_t skt_ray_stride,
                          float tau_v, float tau_d, const float* cutoff_v, const float* cutoff_d,
                          int32_t n_rays, int32_t n_samples, float* dy_ws, float* dq_ws, float* dskts, void* stream);

/* dcodes [n_codes,16] += sums of du's code columns over the rays of each cam_idx and their samples (caller zero-fills
 * dcodes).  Two fixed-order stages (per-ray sums into rowsum_ws [N,16], then one block per code): bit-reproducible. */
int anerf_code_grads(const AnerfConfig* cfg, const float* du, const float* cam_idx, int32_t n_rays, int32_t n_samples,
                     float* dcodes, int32_t n_codes, float* rowsum_ws, void* stream);

/* ---- next rows of SURVEY 8(f) ----------------------------------------------------------------------------------- */

/* Density query (RayCaster.render_pts_density / fwd_type='density'|'mesh', core/raycasters.py:579-648):
 * pts [P,3] world points under ONE pose skts [24,4,4] -> sigma_raw [P] = alpha_linear(forward_density(...)). */
int anerf_density(const AnerfConfig* cfg, const float* packed, const float* aux, const float* pts, const float* skts,
                  float tau_v, const float* cutoff_v, int64_t n_points, float* sigma_raw, void* stream);

/* get_rays (core/utils/ray_utils.py:6-28) for the pixels [x0,x1) x [y0,y1) that kp_to_valid_rays selects (:83-136),
 * written as the [N,11] ray batch of render() (core/trainer.py:116-135); valid_idx[i] = y*W + x.  c2w [3,4]. */
int anerf_gen_rays(int32_t H, int32_t W, float focal_x, float focal_y, float center_x, float center_y, const float* c2w,
                   int32_t x0, int32_t y0, int32_t x1, int32_t y1, float near, float far, float* ray_batch,
                   int64_t* valid_idx, void* stream);

/* render_path's composite + scatter (run_nerf.py:118-131): rgb_img [H*W,3] holds the background on entry and
 * rgb + (1-acc)*bg at valid_idx on exit; disp_img / acc_img [H*W] optional. */
int anerf_assemble_frame(const float* rgb_map, const float* acc_map, const float* disp_map, const int64_t* valid_idx,
                         int32_t n_rays, float* rgb_img, float* disp_img, float* acc_img, void* stream);

/* ---- bf16x3 render path (BASELINE config 5 "bf16 MFMA path", held to the fp32 parity bar) ------------------------
 * Weight image which=3: every weight split into two bf16 (hi, lo); the kernel evaluates W x as
 * Whi*Xhi + Whi*Xlo + Wlo*Xhi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-17 relative per product).
 * anerf_build_pack_table(which=3) fills 2*stream_floats + aux_floats entries (one per bf16 element, then aux). */
int anerf_pack_params_b3(const AnerfNetParams* params, const int32_t* table, int64_t stream_floats, int64_t aux_floats,
                         float* out, void* stream);
/* Same contract as anerf_mlp_raw, on the which=3 image. */
int anerf_mlp_raw_b3(const AnerfConfig* cfg, const float* packed, const float* aux,
                     const float* rays, int32_t ray_stride, const float* z_vals,
                     const float* skts, int64_t skt_ray_stride, const float* cam_idx,
                     const float* codes, int32_t n_codes,
                     float tau_v, float tau_d, const float* cutoff_v, const float* cutoff_d,
                     int32_t n_rays, int32_t n_samples, float* raw, void* stream);

/* ---- loss + optimiser step ---------------------------------------------------------------------
 * _compute_nerf_loss (core/trainer.py:353-380) for the fine and (optional) coarse head, with img2mse / img2l1 / img2huber
 * (:8-60), AND its gradient w.r.t. the rendered maps in one pass:
 *   pred = rgb + (1 - acc) * bg  (bgs != NULL; bg_stride 0 = one [3] colour, >= 3 = per ray)   else pred = rgb
 *   loss = mean_{N x 3} (pred - target)^2   (loss_type 0),   mean |pred - target|   (loss_type 1)   or
 *          img2huber = F.smooth_l1_loss(beta = huber_beta) (loss_type 2; trainer.py:57,152; huber_beta 0 = L1)
 * out4 = {fine + coarse_weight * coarse, fine, coarse, fine MSE (mse2psnr input)};  g_* = d(out4[0]) / d(map), any
 * may be NULL.  partials: workspace of 4 * anerf_loss_blocks(n_rays) floats.  Deterministic (fixed-order sums). */
int anerf_loss_blocks(int32_t n_rays);
int anerf_loss(const float* rgb, const float* acc, const float* rgb0, const float* acc0, const float* target,
               const float* bgs, int32_t bg_stride, int32_t n_rays, int32_t loss_type, float huber_beta,
               float coarse_weight, float* out4, float* g_rgb, float* g_acc, float* g_rgb0, float* g_acc0,
               float* partials, void* stream);

/* torch.optim.Adam (no amsgrad, no weight decay: trainer.py:173-183) over ONE flat fp32 buffer of n elements
 * (16-byte aligned), step = 1-based step count; grads are multiplied by grad_scale first (1/world after a summed
 * all-reduce), optionally zeroed afterwards (optimizer.zero_grad()).  If norms2 != NULL it receives get_gradnorm's
 * (total_norm, avg_norm) (trainer.py:192-203; n_tensors = tensors with a gradient) -- no host sync; partials:
 * anerf_adam_blocks(n) floats. */
int anerf_adam_blocks(int64_t n);
int anerf_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grads, int32_t n_tensors,
                    float* partials, float* norms2, void* stream);

/* ---- PoseOptLayer forward kinematics ------------------------------------------------------------
 * PoseOptLayer.calculate_kinematic / get_kinematic_chain_T + unrolled_kinematic_chain (core/pose_opt.py:372-445,
 * 482-566) for the SMPL tree (skeleton_utils.py:98-105):
 *   bones [U,24,rot_dim]: rot_dim 3 = axis-angle (pytorch3d.transforms.axis_angle_to_matrix, restated; see anerf_fk.hip),
 *                         rot_dim 6 = 6D rotation, the first two columns of R row-major (opt_rot6d: pose_opt.py:284-289,
 *                         391-392; rot6d_to_rotmat skeleton_utils.py:420-436),
 *   pelvis [U,3] or NULL, rest_pose [24,3] (rest_pose_stride 0) or [U,24,3] (stride 72)
 *   -> l2ws [U,24,4,4], skts [U,24,4,4] = inverse(l2ws), rots [U,24,3,3] (local), kp [U,24,3]; any output may be NULL. */
int anerf_fk_forward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose,
                     int64_t rest_pose_stride, int32_t n_poses, float* l2ws, float* skts, float* rots, float* kp,
                     void* stream);
/* Gradients w.r.t. the FK outputs (any may be NULL; only rows 0..2 of the 4x4 matrices are read) -> g_bones
 * [U,24,rot_dim], g_pelvis [U,3] (may be NULL).  With g_skts = the hot path's dskts this closes the pose-refinement loop. */
int anerf_fk_backward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose,
                      int64_t rest_pose_stride, int32_t n_poses, const float* g_skts, const float* g_l2ws,
                      const float* g_kp, const float* g_rots, float* g_bones, float* g_pelvis, void* stream);

/* ---- split-bf16 TRAINING forward: same contract as anerf_mlp_raw_train on the which=3 weight image.  Activations
 * are saved in fp32 exactly as the fp32 forward saves them (the backward kernels and the weight-gradient GEMM stay
 * fp32 and are unchanged); only the column order of saved->x / saved->u differs, so pass the DEVICE copies of
 * anerf_build_perm_tables_b3 (not anerf_build_perm_tables) to anerf_weight_grads. */
int anerf_build_perm_tables_b3(const AnerfConfig* cfg, int32_t* perm_x, int32_t* perm_u);
int anerf_mlp_raw_train_b3(const AnerfConfig* cfg, const float* packed, const float* aux,
                           const float* rays, int32_t ray_stride, const float* z_vals,
                           const float* skts, int64_t skt_ray_stride, const float* cam_idx,
                           const float* codes, int32_t n_codes,
                           float tau_v, float tau_d, const float* cutoff_v, const float* cutoff_d,
                           int32_t n_rays, int32_t n_samples, float* raw, const AnerfSaved* saved, void* stream);

/* anerf_mlp_backward on split-bf16 MFMAs: packed_t = weight image which=4 (W^T as (hi, lo) bf16 fragment pairs, built
 * with anerf_build_pack_table(which=4) + anerf_pack_params_b3), dz values split in registers; same outputs. */
int anerf_mlp_backward_b3(const AnerfConfig* cfg, const float* packed_t, const float* aux, const float* draw,
                          const AnerfSaved* saved, float* dz, float* df, float* dzv, int64_t n_points, void* stream);

/* anerf_input_grads on split-bf16 MFMAs: packed_i = weight image which=5 (anerf_build_pack_table(which=5) +
 * anerf_pack_params_b3); same outputs (stream column order of the fp32 path: anerf_encode_backward is unchanged). */
int anerf_input_grads_b3(const AnerfConfig* cfg, const float* packed_i, const float* dz, const float* dzv,
                         int64_t p_pad, int64_t n_points, float* dx, float* du, void* stream);

/* anerf_weight_grads with the products on split-bf16 MFMAs (operands split hi + lo in registers, fp32 accumulate):
 * same arguments, workspace and (deterministic) reduction; ~1e-6 relative on the gradients. */
int anerf_weight_grads_b3(const AnerfConfig* cfg, const AnerfSaved* saved, const float* dz, const float* df,
                          const float* dzv, const float* draw, int64_t n_points, const int32_t* perm_x,
                          const int32_t* perm_u, const AnerfNetGrads* grads, float* workspace, int64_t ws_floats,
                          void* stream);

/* ---- one-call forward: RayCaster.render_rays (core/raycasters.py:361-474) for one caster call ------------------------
 * The staged entry points above, enqueued back to back on `stream` with every intermediate (bounds, depths, raw logits,
 * weights, importance samples, merged depths / sort indices) in ONE caller-provided workspace of
 * anerf_workspace_size(...) bytes (16-byte aligned; nothing is allocated, nothing synchronises).
 *   packed_c/aux_c, packed_f/aux_f: weight images of the coarse / fine network, which = 0 (precision 0, exact fp32 MFMA) or
 *     which = 3 (precision 1, split-bf16); packed_f/aux_f may be NULL when n_importance == 0; single_net != 0 evaluates the
 *     importance samples with the coarse image only (raycasters.py:411-456) and packed_f is ignored.
 *   t_rand [N,S], u_imp [N,Ni], noise [N,S], noise_fine [N,S+Ni]: optional randomness, NULL = the deterministic variants.
 *   Outputs as RayCaster._collect_outputs (:711-724): rgb_map [N,3], disp_map/acc_map [N], alpha [N,S+Ni] of the last
 *   pass; rgb0/disp0/acc0/alpha0 of the coarse pass when n_importance > 0 (each may be NULL). */
/* ABI revision 3: optional per-kernel timing of the one-call training step.  `ev` holds hipEvent_t handles created by the
 * caller (hipEventCreate, timing enabled); the library only hipEventRecord()s them on `stream` around the MFMA kernels of
 * a pass -- it never creates, waits on or reads an event.  NULL entries are skipped.  Slots (pass p = 0 coarse, 1 fine):
 *   ANERF_PROF_FWD(p)     + {0,1}  before / after k_mlp_fwd<TRAIN>            (anerf_train_forward)
 *   ANERF_PROF_BWD(p)     + {0,1}  before / after k_mlp_bwd                   (anerf_backward)
 *   ANERF_PROF_GEMM(p)    + {0,1}  before / after k_gemm_tn + k_reduce_dw
 *   ANERF_PROF_BWD_IN(p)  + {0,1}  before / after k_mlp_bwd_in (only when input gradients are requested)
 * Used by bench.py for the executed-FLOP roofline of each training kernel; an un-profiled call passes profile = NULL. */
#define ANERF_PROF_FWD(p) (0 + 2 * (p))
#define ANERF_PROF_BWD(p) (4 + 2 * (p))
#define ANERF_PROF_GEMM(p) (8 + 2 * (p))
#define ANERF_PROF_BWD_IN(p) (12 + 2 * (p))
#define ANERF_PROF_SLOTS 16
typedef struct AnerfProfile {
  void* ev[ANERF_PROF_SLOTS];
} AnerfProfile;

/* ---- ABI revision 6: the per-step scalars of a training iteration in DEVICE memory -----------------------------------------
 * Every entry point is stream-only (no allocation, no synchronisation), so a whole training iteration -- anerf_rand_fill,
 * anerf_pack_params_multi, anerf_train_forward, anerf_loss, anerf_backward, anerf_adam_step, the pose layer -- can be captured
 * ONCE into a hipGraph and replayed.  What changes from one iteration to the next besides the buffers' contents are a few scalars
 * that are kernel ARGUMENTS in the calls above, i.e. frozen into a captured graph: the Philox offset of the random inputs, the gate
 * temperatures tau (CutoffEmbedder.update_tau, core/cutoff_embedder.py:181-183: a new value every global step), Adam's step
 * count / learning rate (decay_optimizer_lrate, core/trainer.py:173-183) and the 1/world gradient scale.  They live in an
 * AnerfStepBlock instead; the *_dev forms below and AnerfForwardIO.step read it, so a replay needs no node update:
 *   per iteration:  anerf_step_block_write(block, &values, stream)      one launch, the values travel as kernel arguments
 *                   hipGraphLaunch(exec, stream)
 * The arithmetic is that of the by-value forms (the bias corrections are computed on the host in double by both), so a captured
 * step is bit-identical to the same step issued call by call (tests/test_graph_step.py). */
#define ANERF_MAX_ADAM_GROUPS 4
typedef struct AnerfStepBlock {          /* DEVICE memory, 16-byte aligned; written only by anerf_step_block_write */
  uint64_t rng_seed, rng_offset;         /* anerf_rand_fill_dev                                                     */
  float tau_v, tau_d;                    /* AnerfForwardIO.step (adjacent: the kernels read them as a pair)         */
  float adam_step_size[ANERF_MAX_ADAM_GROUPS];    /* lr / (1 - beta1^step)                                          */
  float adam_sqrt_bc2[ANERF_MAX_ADAM_GROUPS];     /* sqrt(1 - beta2^step)                                           */
  float adam_grad_scale[ANERF_MAX_ADAM_GROUPS];   /* 1/world after a summed all-reduce, else 1                      */
  float reserved_[2];
} AnerfStepBlock;
typedef struct AnerfStepValues {         /* HOST: what the host mirror knows before it enqueues iteration i         */
  uint64_t rng_seed, rng_offset;
  float tau_v, tau_d;
  int32_t n_groups;                      /* optimiser groups in use, <= ANERF_MAX_ADAM_GROUPS                       */
  float lr[ANERF_MAX_ADAM_GROUPS], beta1[ANERF_MAX_ADAM_GROUPS], beta2[ANERF_MAX_ADAM_GROUPS];
  int32_t adam_step[ANERF_MAX_ADAM_GROUPS];       /* 1-based step count this iteration applies; <= 0: group not stepped, its
                                                   * block entries are left as they are                             */
  float grad_scale[ANERF_MAX_ADAM_GROUPS];
} AnerfStepValues;
int anerf_step_block_write(AnerfStepBlock* block, const AnerfStepValues* values_host, void* stream);
/* anerf_rand_fill / anerf_adam_step with (seed, offset) / (lr, step, grad_scale) taken from the block (group = optimiser group).
 * call_index: the fill's index inside the iteration (one fill per caster call: 0 for the first chunk, 1 for the second ...);
 * it draws what anerf_rand_fill(jobs, n_jobs, block->rng_seed, block->rng_offset + call_index) draws. */
struct AnerfRandJob;
int anerf_rand_fill_dev(const struct AnerfRandJob* jobs, int32_t n_jobs, const AnerfStepBlock* block, int32_t call_index, void* stream);
int anerf_adam_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2, float eps,
                        const AnerfStepBlock* block, int32_t group, int32_t zero_grads, int32_t n_tensors, float* partials,
                        float* norms2, void* stream);

typedef struct AnerfForwardIO {
  const float *packed_c, *aux_c, *packed_f, *aux_f;
  const float *rays; int32_t ray_stride;
  const float *skts; int64_t skt_ray_stride;
  const float *cyls, *cam_idx, *codes_c, *codes_f; int32_t n_codes;
  const float *t_rand, *u_imp, *noise, *noise_fine;
  const float *cutoff_v, *cutoff_d; float tau_v, tau_d;
  int32_t n_rays, n_samples, n_importance, lindisp, single_net, precision;
  float *rgb_map, *disp_map, *acc_map, *alpha, *rgb0, *disp0, *acc0, *alpha0;
  /* ABI revision 2: additive offsets of the sample points, `pts + randn_like(pts) * ray_noise_std` of
   * RayCaster.sample_pts / sample_pts_is (core/raycasters.py:650-677).  pts_noise [N,S,3] for the coarse samples, pts_noise_is
   * [N,Ni,3] for the importance samples (required with pts_noise when n_importance > 0); the fine pass gathers both by the
   * sort order of the merged depths.  NULL (the default of every shipped config: ray_noise_std = 0) = no offsets. */
  const float *pts_noise, *pts_noise_is;
  const AnerfProfile* profile;   /* ABI revision 3 (HOST pointer, may be NULL): see AnerfProfile */
  /* ABI revision 4: != 0 = `cyls` holds ONE cylinder [5] shared by every ray of the call -- what run_nerf.render_path's
   * `reuse_input(cyls, expand)` (run_nerf.py:62-72: a stride-0 expand of the frame's cylinder) means; 0 = per-ray [N,5]. */
  int32_t cyl_shared;
  /* ABI revision 6 (DEVICE pointer, may be NULL): the step block of a captured training step.  When given, the TRAINING kernels
   * (anerf_train_forward, and anerf_backward's pose-gradient kernel) read tau_v / tau_d from it instead of from the two fields
   * above; anerf_forward (rendering) ignores it. */
  const struct AnerfStepBlock* step;
} AnerfForwardIO;
int64_t anerf_workspace_size(const AnerfConfig* cfg, int32_t n_rays, int32_t n_samples, int32_t n_importance);
int anerf_forward(const AnerfConfig* cfg, const AnerfForwardIO* io, void* workspace, int64_t ws_bytes, void* stream);

/* ---- one-call training step: the anerf_forward / anerf_backward pair of the boundary -----------------
 * anerf_train_forward = anerf_forward with the training kernels: same io (t_rand / u_imp / noise as the trainer draws
 * them: perturb = 1, raw_noise_std = 1), and a workspace of anerf_train_workspace_size(...) bytes that additionally holds
 * the saved activations of both network passes.  The caller keeps io's buffers and the workspace untouched until
 * anerf_backward has been enqueued (autograd's saved-tensor contract).  Two-network configurations only; single_net
 * trains through the staged entry points above.
 * anerf_backward: gradients of the rendered maps (of the last pass: g_rgb [N,3] required, g_disp / g_acc [N], g_alpha
 * [N,S+Ni] optional; of the coarse pass when n_importance > 0: g_rgb0 required, others optional) -> every parameter
 * gradient of both networks (written, torch layout), and optionally d(loss)/d(skts) [N,24,4,4] (per-ray skts; zero-filled
 * inside, both passes summed) and d(loss)/d(frame codes) [n_codes,16] per network (zero-filled inside).
 * packed_t_*: which = 1 (fp32) / 4 (bf16x3) images; packed_i_*: which = 2 / 5, only read when g_skts or g_codes_* is
 * given; perm_x / perm_u: DEVICE copies of anerf_build_perm_tables (fp32) / anerf_build_perm_tables_b3 (bf16x3).
 * scratch: anerf_backward_scratch_size(...) bytes (input_grads != 0 when g_skts / g_codes_* will be requested; the size includes
 * a second region for the fine pass's GEMM partials: when both passes run in one call, without per-kernel timing, they are reduced
 * together with the coarse pass's by ONE k_reduce_dw2 launch -- a smaller scratch of the first region only still works), free
 * to reuse after the call is enqueued and executed.  Same kernels, same order, same results as the staged sequence
 * composite_backward -> mlp_backward -> weight_grads (-> input_grads -> encode_backward / code_grads) per pass. */
typedef struct AnerfBackwardIO {
  const float *g_rgb, *g_disp, *g_acc, *g_alpha;
  const float *g_rgb0, *g_disp0, *g_acc0, *g_alpha0;
  const float *packed_t_c, *packed_t_f, *packed_i_c, *packed_i_f;
  const int32_t *perm_x, *perm_u;
  AnerfNetGrads grads_c, grads_f;
  float *g_skts, *g_codes_c, *g_codes_f;
  int32_t accumulate;   /* 0: grads_c / grads_f are written; 1: added to their current contents (param.grad in place) -- and so are
                         * g_codes_c / g_codes_f (ABI revision 4: no zero fill of them inside) */
  int32_t passes;       /* ABI revision 2.  0 or 3: both network passes (fine, then coarse).  1: the fine pass only; 2: the
                         * coarse pass only -- the two halves of one backward, enqueued as two calls so that the caller can
                         * start reducing the fine network's gradients (complete after the first call) over RCCL while the
                         * coarse pass runs (replaces nn.DataParallel's reduce-add, core/raycasters.py:157).  g_skts is
                         * zero-filled by the call that runs the fine pass; a coarse-only call ADDS to it.  Ignored (both
                         * passes = the one pass) when n_importance == 0.
                         * ABI revision 6 -- the coarse pass in two calls (n_importance > 0 and g_skts only): 4 = everything that ends
                         * in PARAMETER gradients (weights, biases, frame codes), 8 = its pose-gradient tail into g_skts, reading the
                         * input gradients the passes = 4 call left in the SAME scratch.  On iterations where the pose group is not
                         * stepped the tail produces nothing that is all-reduced, so the coarse network's collective (started after
                         * the passes = 4 call) runs under it instead of after the backward.
                         * ABI revision 7 -- the passes = 4 part in two calls (n_importance > 0 and input gradients requested): 16 = the
                         * coarse pass up to its WEIGHT gradients (composite backward, k_mlp_bwd, GEMM + reduction), 32 = its
                         * input-gradient part (k_mlp_bwd_in[_enc] + frame-code gradients), reading dz / dzv of the passes = 16 call
                         * from the SAME scratch; then 8 as before.  The coarse network's 3.46 MB of weight gradients can be
                         * all-reduced from behind the passes = 16 call, under the input-gradient kernel; only its frame-code
                         * table (n_codes x 16 floats) is left for the tail's window.
                         * (g_skts is written in full by the first pose-gradient pass: "zero-filled" above is history.) */
  const AnerfProfile* profile;   /* ABI revision 3 (HOST pointer, may be NULL): see AnerfProfile */
} AnerfBackwardIO;
int64_t anerf_train_workspace_size(const AnerfConfig* cfg, int32_t n_rays, int32_t n_samples, int32_t n_importance);
int64_t anerf_backward_scratch_size(const AnerfConfig* cfg, int32_t n_rays, int32_t n_samples, int32_t n_importance,
                                    int32_t input_grads);
int anerf_train_forward(const AnerfConfig* cfg, const AnerfForwardIO* io, void* workspace, int64_t ws_bytes, void* stream);
int anerf_backward(const AnerfConfig* cfg, const AnerfForwardIO* io, const AnerfBackwardIO* b, void* workspace,
                   int64_t ws_bytes, void* scratch, int64_t scratch_bytes, void* stream);

/* ABI revision 4: the pose layer's per-iteration work as ONE launch each way (PoseOptLayer.forward, core/pose_opt.py:318-331,
 * 372-445: parameter lookup for the batch's poses, forward kinematics once per DISTINCT pose, per-ray expansion `x[inverse]` of
 * the five outputs; and its autograd).  bones [P,24,rot_dim] / pelvis [P,3] (may be NULL) are the layer's FULL parameter tensors,
 * pose_idx [U] (int64, device) the distinct pose rows of the batch, inverse [N] (int32, device) each ray's slot in 0..U-1, rest_pose
 * [24,3] one rest pose shared by all poses.  Outputs, any may be NULL: per distinct pose u_kp [U,24,3], u_bones [U,24,rot_dim] (the
 * gathered parameters), u_rots [U,24,3,3]; per ray r_kp [N,24,3], r_bones [N,24,rot_dim], r_skts / r_l2ws [N,24,4,4], r_rots
 * [N,24,3,3].
 * backward: gradients w.r.t. any of those outputs (NULL = none) -> g_bones [P,24,rot_dim], g_pelvis [P,3] (may be NULL) at the rows
 * pose_idx[u] only (other rows untouched): written, or with accumulate != 0 ADDED to their contents (the parameters' .grad in
 * place).  Per-ray gradients of a pose are summed in ray order inside 64-ray chunks and the chunks in chunk order: bit-reproducible.
 * scratch: anerf_pose_batch_scratch_size(n_unique, n_rays) bytes (only read when a per-ray gradient is given).  Launches: one
 * forward; backward one (no per-ray gradients) or two. */
int anerf_pose_batch_forward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose, const int64_t* pose_idx,
                             int32_t n_unique, const int32_t* inverse, int32_t n_rays, float* u_kp, float* u_bones, float* u_rots,
                             float* r_kp, float* r_bones, float* r_skts, float* r_l2ws, float* r_rots, void* stream);
int anerf_pose_batch_backward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose, const int64_t* pose_idx,
                              int32_t n_unique, const int32_t* inverse, int32_t n_rays, const float* g_r_kp, const float* g_r_bones,
                              const float* g_r_skts, const float* g_r_l2ws, const float* g_r_rots, const float* g_u_kp,
                              const float* g_u_bones, const float* g_u_rots, float* g_bones, float* g_pelvis, int32_t accumulate,
                              void* scratch, int64_t scratch_bytes, void* stream);
int64_t anerf_pose_batch_scratch_size(int32_t n_unique, int32_t n_rays);

/* Pose regulariser of the pose-refinement step (Trainer._compute_kp_loss, core/trainer.py:382-403) and its gradient in one
 * launch:  loss = coef * sum_u w_u / 23 * sum_{j >= 1, c} max-thresholded (anchor - value)^2  ("d > tol ? d - tol : 0"),
 * over the U DISTINCT poses of the batch with pose_weights w_u = (rays of pose u) / N -- the value the reference computes
 * on the per-ray replicated batch.  rot6d != 0 (opt_rot6d): values = rots [U,24,3,3] (FK output, columns 0..1 are used),
 * anchors [U,24,6]; rot6d == 0: values = anchors-shaped axis-angle bones [U,24,3].  loss [1]; g_values (same shape as values,
 * may be NULL) = d loss / d values, every element written. */
int anerf_kp_loss(const float* values, int32_t rot6d, const float* anchors, const float* pose_weights, int32_t n_poses, float tol,
                  float coef, float* loss, float* g_values, void* stream);

/* ABI revision 7: the same launch also forms the trainer's sum `total_loss = rgb losses + kp_loss` (core/trainer.py:236-246):
 * total [1] = base [1] + loss (one fp32 add -- what the separate torch add kernel computed; that kernel was one of the launch-floor
 * dispatches of the 384-ray step).  base: DEVICE scalar (e.g. out4[0] of anerf_loss); base, loss, total must not be NULL. */
int anerf_kp_loss_add(const float* values, int32_t rot6d, const float* anchors, const float* pose_weights, int32_t n_poses, float tol,
                      float coef, const float* base, float* loss, float* total, float* g_values, void* stream);

/* ---- ABI revision 3: per-step host glue as single launches (the 384-rays-per-rank step) --------
 * Everything a training iteration does around the caster call used to be a string of small torch launches (4 weight-image
 * gathers, 2 x torch.rand + 2 x torch.randn + scaling, ones_like / norm / div / cat for the ray batch): ~0.3 ms of a
 * 2.6 ms step at 384 rays per rank.  Each group is one launch here. */

/* anerf_pack_params for several weight images at once (one launch, blockIdx.y = job).  kind 0: out is float[n]
 * (fp32 images which = 0 / 1 / 2, and the aux part of a bf16x3 image); kind 1: out is uint16[n], the hi/lo-split
 * stream part of a bf16x3 image (which = 3 / 4 / 5; table entries as in anerf_pack_params_b3).  n_jobs <= 8. */
typedef struct AnerfPackJob {
  AnerfNetParams params;
  const int32_t* table;
  int64_t n;
  void* out;
  int32_t kind;
} AnerfPackJob;
#define ANERF_MAX_PACK_JOBS 8
int anerf_pack_params_multi(const AnerfPackJob* jobs, int32_t n_jobs, void* stream);

/* The random inputs of one caster call in ONE launch: replaces torch.rand (t_rand of sample_from_lineseg,
 * ray_utils.py:240-246; u of sample_pdf, :171-180) and torch.randn * raw_noise_std (nerf.py:176-182) /
 * randn_like(pts) * ray_noise_std (raycasters.py:660,674).  Counter-based Philox4x32-10: element i of job j is a pure
 * function of (seed, offset, j, i) -- reproducible, independent of the launch geometry; the caller advances `offset` by 1
 * per call.  kind 0: uniform in [0, 1) (24 random bits, as torch.rand(float32)); kind 1: standard normal (Box-Muller) *
 * scale.  n_jobs <= 6. */
typedef struct AnerfRandJob {
  float* out;
  int64_t n;
  int32_t kind;
  float scale;
} AnerfRandJob;
#define ANERF_MAX_RAND_JOBS 6
int anerf_rand_fill(const AnerfRandJob* jobs, int32_t n_jobs, uint64_t seed, uint64_t offset, void* stream);

/* render()'s ray-batch assembly (core/trainer.py:116-135): rays_o / rays_d [N,3] -> ray_batch [N, 8 | 11] =
 * (o, d, near, far [, d / |d|]).  out_stride 8 (no view directions) or 11. */
int anerf_make_ray_batch(const float* rays_o, const float* rays_d, int32_t n_rays, float near, float far, int32_t out_stride,
                         float* ray_batch, void* stream);

/* 2-D pixel boxes of the projected bounding cylinders of F frames in one launch (cylinder_to_box_2d,
 * core/utils/skeleton_utils.py:607-690 + nerf_c2w_to_extrinsic :442, as kp_to_valid_rays uses them,
 * core/utils/ray_utils.py:83-136): per frame cyl [5], c2w [3,4] row-major -- DOUBLE: the reference projects the cap points in
 * float64 (it inverts the camera matrix in the dtype it arrives in; boxes agree unless a projected extreme lies within ~1e-5
 * px of an integer) --,
 * hwf [4] = (H, W, fx, fy), off [2] = integer principal point; circle [50][2] = (cos, sin) of linspace(0, 2 pi, 50)
 * computed on the host.  bbox [F,4] int32 = (x0, y0, x1, y1), clipped to the image as the reference does. */
int anerf_cyl_bbox(const double* cyls, const double* c2ws, const double* hwf, const int32_t* off, const double* circle,
                   int32_t n_frames, int32_t* bbox, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANERF_H */
