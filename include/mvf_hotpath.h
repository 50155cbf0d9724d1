/*
 * mvf_hotpath.h -- C ABI of the MI355X (gfx950) view-synthesis + photometric-loss hot path.
 *
 * The reference (LiuJF1226/Mono-ViFI) has no FFI: its boundary for this path is the Python
 * API of layers.py plus three Trainer methods.  Each entry point below is the native body
 * of one of those functions (cited as file:line relative to the reference root); the Python
 * mirror in mono-vifi_amd/layers.py and mono-vifi_amd/trainer.py binds them with ctypes.
 *
 * Conventions
 *  - all pointers are DEVICE pointers (HIP), fp32 unless stated, tensors contiguous in the
 *    reference's layout: images [B,C,H,W], disp/depth [B,1,H,W], cam points [B,4,H*W],
 *    sampling grid [B,H,W,2], K / inv_K / T [B,4,4] row-major;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *    enqueues work -- no host synchronisation, no allocation, no global state;
 *  - return value: hipError_t as int (0 = success); mvf_error_string() describes it;
 *  - "exact mode" arithmetic: IEEE fp32, no FMA contraction except the
 *    k-sequential fmaf chains of the two [3xk]@[kxN] matrix products, true divides -- the
 *    integer sampling indices are bit-identical to the reference's CPU run.
 *  - reductions are deterministic: per-workgroup partials in caller-provided workspace,
 *    folded by a finishing kernel in a fixed order.
 */
#ifndef MVF_HOTPATH_H
#define MVF_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVF_ABI_VERSION 14

#if defined(__GNUC__)
#define MVF_API __attribute__((visibility("default")))
#else
#define MVF_API
#endif

/* flags of compute_losses_base (reference: options.py --no_ssim / --avg_reprojection /
 * --disable_automasking, used at train.py:979,1001,1010,1016) */
#define MVF_NO_SSIM 1
#define MVF_AVG_REPROJ 2
#define MVF_NO_AUTOMASK 4

#define MVF_MAX_SRC 4
#define MVF_MAX_UNITS 8 /* hot-path units one mvf_units_fwdbwd launch can carry */

MVF_API int mvf_abi_version(void);
MVF_API const char *mvf_error_string(int err);

/* ---- a1: layers.disp_to_depth (layers.py:16-25) -------------------------------------
 * scaled = min_disp + range*disp ; depth = 1/scaled.  min_disp = (float)(1/max_depth),
 * range = (float)(1/min_depth - 1/max_depth) are rounded by the caller.  Either output may
 * be NULL. */
MVF_API int mvf_disp_to_depth_fwd(const float *disp, float *scaled, float *depth, int64_t n,
                          float min_disp, float range, void *stream);
/* g_disp = g_scaled*range - g_depth*range*depth^2 (either upstream may be NULL) */
MVF_API int mvf_disp_to_depth_bwd(const float *disp, const float *g_scaled, const float *g_depth,
                          float *g_disp, int64_t n, float min_disp, float range, void *stream);

/* ---- a2: BackprojectDepth.forward (layers.py:192-197) --------------------------------
 * cam[b,0:3,i] = depth[b,i] * (inv_K[b,:3,:3] @ [x,y,1]) ; cam[b,3,i] = 1 ; i = y*W+x */
MVF_API int mvf_backproject_fwd(const float *depth, const float *inv_K, float *cam, int B, int H, int W,
                        void *stream);
/* g_depth[b,i] = sum_c g_cam[b,c,i] * ray_c(i) */
MVF_API int mvf_backproject_bwd(const float *g_cam, const float *inv_K, float *g_depth, int B, int H,
                        int W, void *stream);

/* ---- a3: Project3D.forward (layers.py:211-222) ---------------------------------------
 * P = (K@T)[:3] ; c = P@cam ; pix = ((c.xy/(c.z+eps)) / (W-1,H-1) - 0.5)*2, [B,H,W,2] */
MVF_API int mvf_project_fwd(const float *cam, const float *K, const float *T, float *pix, int B, int H,
                    int W, float eps, void *stream);
/* g_cam [B,4,N] (may be NULL) and g_T [B,4,4] (may be NULL).  workspace: floats, size
 * mvf_workspace_floats(B,H,W). */
MVF_API int mvf_project_bwd(const float *cam, const float *K, const float *T, const float *g_pix,
                    float *g_cam, float *g_T, float *workspace, int B, int H, int W, float eps,
                    void *stream);

/* ---- a4: F.grid_sample(img, grid, "bilinear", "border", align_corners=True) ----------
 * call site train.py:966-969.  idx_xy (nullable) int32 [B,H,W,2] = top-left tap (x0,y0):
 * the bit-exact integers of the parity contract. */
MVF_API int mvf_grid_sample_fwd(const float *img, const float *grid, float *out, int32_t *idx_xy, int B,
                        int C, int H, int W, void *stream);
/* g_grid [B,H,W,2] (zero where the coordinate was clipped); g_img nullable, must be
 * zero-initialised by the caller (scatter-add). */
MVF_API int mvf_grid_sample_bwd(const float *img, const float *grid, const float *g_out, float *g_grid,
                        float *g_img, int B, int C, int H, int W, void *stream);

/* ---- a5: Trainer.generate_images_pred (train.py:956-971), fused per pixel -------------
 * disp->depth->backproject->project->grid_sample for ONE source.  pix / idx_xy nullable. */
MVF_API int mvf_warp_fwd(const float *disp, const float *inv_K, const float *K, const float *T,
                 const float *src, float *warped, float *pix, int32_t *idx_xy, int B, int H,
                 int W, float min_disp, float range, float eps, void *stream);
/* g_warped [B,3,H,W] -> g_disp [B,1,H,W] (overwritten, or added to when accumulate != 0)
 * and g_T [B,4,4].  g_src nullable, zero-initialised by the caller (scatter-add).
 * workspace: mvf_workspace_floats(B,H,W) floats. */
MVF_API int mvf_warp_bwd(const float *disp, const float *inv_K, const float *K, const float *T,
                 const float *src, const float *g_warped, float *g_disp, float *g_T, float *g_src,
                 float *workspace, int accumulate, int B, int H, int W, float min_disp,
                 float range, float eps, void *stream);

/* ---- a6: SSIM.forward (layers.py:277-290) ---------------------------------------------*/
MVF_API int mvf_ssim_fwd(const float *x, const float *y, float *out, int B, int C, int H, int W,
                 void *stream);
/* g_x, g_y: either may be NULL */
MVF_API int mvf_ssim_bwd(const float *x, const float *y, const float *g_out, float *g_x, float *g_y,
                 int B, int C, int H, int W, void *stream);

/* ---- a7: Trainer.compute_reprojection_loss (train.py:973-985) -------------------------
 * pred, target [B,3,H,W] -> out [B,1,H,W] = 0.85*mean_c SSIM + 0.15*mean_c |t-p| */
MVF_API int mvf_reprojection_fwd(const float *pred, const float *target, float *out, int B, int H, int W,
                         int no_ssim, void *stream);
MVF_API int mvf_reprojection_bwd(const float *pred, const float *target, const float *g_out, float *g_pred,
                         int B, int H, int W, int no_ssim, void *stream);

/* ---- a9: get_smooth_loss (layers.py:231-242) ------------------------------------------
 * out[0] = mean|dx disp|*exp(-mean_c|dx img|) + same for y.  normalise != 0 first divides
 * disp by (per-image mean + 1e-7) as train.py:1044-1045 does.
 * stats (nullable, [B,4] floats): per image {mean, den = mean+1e-7, sx_b, sy_b} saved for
 * the backward.  workspace: mvf_workspace_floats(B,H,W). */
MVF_API int mvf_smooth_fwd(const float *disp, const float *img, float *out, float *stats, float *workspace,
                   int normalise, int B, int H, int W, void *stream);
/* g_disp = (*g_loss) * scale * d smooth / d disp ; overwritten or accumulated */
MVF_API int mvf_smooth_bwd(const float *disp, const float *img, const float *stats, const float *g_loss,
                   float scale, float *g_disp, int accumulate, int normalise, int B, int H, int W,
                   void *stream);

/* ---- a8: Trainer.compute_losses_base (train.py:987-1051) ------------------------------
 * warped[k], src[k]: S device pointers each (host arrays of device pointers), [B,3,H,W].
 * noise: the standard-normal tie-break draw of train.py:1023-1024 BEFORE the 1e-5 scale,
 * [B,n_id,H,W] with n_id = 1 if AVG_REPROJ else S (ignored when NO_AUTOMASK);
 * mask_rec nullable [B,1,H,W].
 * outputs: loss[0] = mean(to_optimise) + smoothness*smooth ; loss[1] = photometric mean ;
 * loss[2] = smooth ; argmin uint8 [B,H,W] (candidate index, identity candidates first; 255
 * when there is a single candidate) ; auto_mask nullable float [B,1,H,W] ; to_opt nullable
 * float [B,H,W] ; stats [B,4] (see mvf_smooth_fwd).
 * workspace: mvf_workspace_floats(B,H,W). */
MVF_API int mvf_photo_fwd(const float *disp, const float *tgt, const float *const *warped,
                  const float *const *src, const float *noise, const float *mask_rec, int S,
                  int flags, float smoothness, float *loss, uint8_t *argmin, float *auto_mask,
                  float *to_opt, float *stats, float *workspace, int B, int H, int W, void *stream);
/* g_loss: device scalar.  g_warped[k] [B,3,H,W] overwritten; g_disp [B,1,H,W] overwritten
 * with the smoothness gradient (nullable). */
MVF_API int mvf_photo_bwd(const float *disp, const float *tgt, const float *const *warped,
                  const uint8_t *argmin, const float *mask_rec, const float *stats,
                  const float *g_loss, int S, int flags, float smoothness, float *const *g_warped,
                  float *g_disp, int B, int H, int W, void *stream);

/* ---- a5+a8 fused: one hot-path UNIT (S x generate_images_pred + compute_losses_base) ---
 * The warped images never touch HBM: algorithmic traffic is disp 4 + tgt 12 + S*12 B/px
 * read, 1 B/px (argmin) [+4 auto_mask] written.  T: [S,B,4,4].  src: S device pointers.
 * idx_xy nullable int32 [S,B,H,W,2] (parity tests). */
MVF_API int mvf_unit_fwd(const float *disp, const float *tgt, const float *const *src, const float *T,
                 const float *K, const float *inv_K, const float *noise, const float *mask_rec,
                 int S, int flags, float smoothness, float min_disp, float range, float eps,
                 float *loss, uint8_t *argmin, float *auto_mask, float *to_opt, float *stats,
                 int32_t *idx_xy, float *workspace, int B, int H, int W, void *stream);
/* g_disp [B,1,H,W] overwritten ; g_T [S,B,4,4] overwritten */
MVF_API int mvf_unit_bwd(const float *disp, const float *tgt, const float *const *src, const float *T,
                 const float *K, const float *inv_K, const uint8_t *argmin, const float *mask_rec,
                 const float *stats, const float *g_loss, int S, int flags, float smoothness,
                 float min_disp, float range, float eps, float *g_disp, float *g_T,
                 float *workspace, int B, int H, int W, void *stream);
/* Forward AND backward of a unit in one tile kernel (S <= 2: one source pair), for the
 * training step, where both always run: loss[3], stats[B,4] as mvf_unit_fwd, plus the
 * gradients for an upstream gradient of 1 in RAW form -- g_disp_raw [B,1,H,W] lacks the
 * per-image constant of the mean-normalised smoothness term (it needs the per-image smoothness
 * sum, known only after the launch) and g_T_raw [S,B,4,4] is final; mvf_unit_fwdbwd_scale
 * applies the constant and the upstream gradient in one pass (the backward is linear in it).
 * The warp, the staging and the window statistics are done once instead of once per direction.
 * argmin / auto_mask / to_opt nullable; idx_xy nullable int32 [S,B,H,W,2] as for mvf_unit_fwd
 * (the parity tests read the sampling indices of the kernel that trains).
 * noise == NULL (with auto-masking on): the tie-break draw of train.py:1023-1024 is generated
 * in the kernel from a counter-based generator keyed by noise_seed and the element index, and
 * written to noise_out (nullable, same layout as noise) so that it can be replayed.
 * disp_mean_partials (nullable) [B,32]: the per-image partial sums of disp as mvf_disp_head_fwd
 * emits them; NULL = computed here (one more small launch).
 * workspace: mvf_workspace_floats(B,H,W) floats. */
MVF_API int mvf_unit_fwdbwd(const float *disp, const float *tgt, const float *const *src, const float *T,
                    const float *K, const float *inv_K, const float *noise, const float *mask_rec,
                    int S, int flags, float smoothness, float min_disp, float range, float eps,
                    float *loss, uint8_t *argmin, float *auto_mask, float *to_opt, float *stats,
                    int32_t *idx_xy, float *g_disp_raw, float *g_T_raw, float *workspace, int B, int H,
                    int W, uint64_t noise_seed, float *noise_out, const float *disp_mean_partials,
                    void *stream);
/* ---- several units in ONE launch ------------------------------------------------------------
 * Trainer.process_batch (train.py:747-883) runs nine units per step in three groups of three
 * mutually independent ones (single-frame train.py:747-760, multi-frame 795-810, affine 837-882);
 * the units of a group share B, H, W, S and the flags and go out as one launch of
 * n_units * B * tiles workgroups.  Images are addressed as base + b * stride (strides in FLOATS,
 * 0 = contiguous), so the interleaved [B*G,...] output of a grouped decoder call is read in
 * place; inside an image the reference's planar layout is required, and a plane must stay below
 * 2^28 bytes (H*W < 67,108,864: tap offsets travel with four flag bits above them) -- larger shapes
 * return hipErrorInvalidValue.
 *
 * ident_out / ident_in ([B,H,W,2], nullable): the two identity-reprojection maps of the unit
 * (compute_reprojection_loss of the raw sources against the target, train.py:1020-1022, BEFORE
 * the tie-break noise).  The multi-frame unit of a target shares target and sources with the
 * single-frame one (train.py:747-749 vs 795-797), so it can take the maps the first wrote
 * instead of staging the sources and re-evaluating their SSIM: same values, bit for bit.
 *
 * One finishing launch folds the tile partials of all units (loss[3], stats[B,4], g_T_raw;
 * deterministic: fixed fold order); the last block of a unit to arrive folds its images.
 * tickets: mvf_units_ticket_ints(n_units, B) int32 counters, ZERO on entry; the call leaves them
 * zero, so one persistent buffer per stream serves every call.
 * workspace: mvf_units_workspace_floats(...) floats, 8-byte aligned.
 * Everything else as for mvf_unit_fwdbwd (which is this call with one contiguous unit). */
typedef struct mvf_unit_desc {
    const float *disp;     int64_t disp_stride;       /* [B,1,H,W] */
    const float *tgt;      int64_t tgt_stride;        /* [B,3,H,W] */
    const float *src[2];   int64_t src_stride[2];     /* [B,3,H,W] each; src[1] unused when S == 1 */
    const float *T;                                   /* [S,B,4,4] */
    const float *K, *inv_K;                           /* [B,4,4] */
    const float *mask_rec; int64_t mask_stride;       /* nullable [B,1,H,W] */
    const float *noise;                               /* nullable [B,n_id,H,W] contiguous */
    const float *disp_mean_partials;                  /* nullable [B,32] */
    const float *ident_in;                            /* nullable [B,H,W,2] */
    uint64_t noise_seed;
    float *ident_out;                                 /* nullable [B,H,W,2] */
    float *loss;                                      /* [3] */
    float *stats;                                     /* [B,4] */
    float *g_disp_raw;     int64_t g_stride;          /* [B,1,H,W] */
    float *g_T_raw;                                   /* [S,B,4,4] */
    uint8_t *argmin;                                  /* nullable [B,H,W] */
    float *auto_mask, *to_opt;                        /* nullable [B,1,H,W] contiguous */
    int32_t *idx_xy;                                  /* nullable [S,B,H,W,2] */
    float *noise_out;                                 /* nullable, layout of noise */
    float *loss_sum;                                  /* nullable [1]; read from units[0] only: the SUM of loss[0] over
                                                         the units of the launch (unit order), written by the last
                                                         finishing block -- the `losses.sum()` of a group of
                                                         process_batch (train.py:760, 812, 882) without a launch */
    const float *loss_sum_in;                         /* nullable [1]; read from units[0] only: a running total the sum
                                                         is added to (loss_sum = *loss_sum_in + the units' losses in unit
                                                         order): `loss_base +=` of train.py:760, 812, 882 without a launch */
} mvf_unit_desc;
MVF_API size_t mvf_units_workspace_floats(int n_units, int B, int H, int W);
MVF_API size_t mvf_units_ticket_ints(int n_units, int B);
MVF_API int mvf_units_fwdbwd(const mvf_unit_desc *units, int n_units, int S, int flags, float smoothness,
                     float min_disp, float range, float eps, float *workspace, int32_t *tickets, int B,
                     int H, int W, void *stream);
/* backward() of mvf_units_fwdbwd, one launch for the units' grad_disp and grad_T (formula below);
 * upstream gradient of a unit = *g_loss (device scalar, nullable) + *g_sum (device scalar, nullable: the gradient
 * of the launch's loss_sum); at least one of the two.  g_disp == NULL: only grad_T of that unit is formed (its
 * disparity gradient is consumed raw by mvf_disp_head_bwd_units). */
typedef struct mvf_unit_scale_desc {
    const float *g_disp_raw; int64_t in_stride;
    const float *g_T_raw, *stats, *g_loss;
    float *g_disp;           int64_t out_stride;
    float *g_T;
    const float *g_sum;
} mvf_unit_scale_desc;
MVF_API int mvf_units_fwdbwd_scale(const mvf_unit_scale_desc *units, int n_units, float smoothness, int B,
                           int S, int H, int W, void *stream);

/* backward() of mvf_unit_fwdbwd: g_disp = (g_disp_raw - shift_b) * (*g_loss), g_T = g_T_raw * (*g_loss),
 * shift_b = (smoothness * (stats[b,2] + stats[b,3]) / (H*W)) / stats[b,1]; g_loss device scalar.
 * In-place use (g_disp == g_disp_raw) is allowed. */
MVF_API int mvf_unit_fwdbwd_scale(const float *g_disp_raw, const float *g_T_raw, const float *stats,
                          const float *g_loss, float smoothness, float *g_disp, float *g_T, int B,
                          int S, int H, int W, void *stream);


/* ---- a10: layers.transformation_from_parameters (layers.py:28-103) --------------------
 * axisangle, translation [B,3] -> M [B,4,4]; M = T*R, or R^T*T(-t) when invert != 0 */
MVF_API int mvf_pose_fwd(const float *axisangle, const float *translation, float *M, int invert, int B,
                 void *stream);
MVF_API int mvf_pose_bwd(const float *axisangle, const float *translation, const float *g_M,
                 float *g_axisangle, float *g_translation, int invert, int B, void *stream);

/* ---- f1: flow warp -- IFRNet.warp (networks/IFRNet.py:7-15) and
 * FusionModule.warp_features (networks/fusion_module.py:80-90).  out[b,c,y,x] = bilinear /
 * border / align_corners=True sample of img[b,c] at grid = (xs[x] + flow_x/((W-1)/2),
 * ys[y] + flow_y/((H-1)/2)); xs = linspace(-1,1,W), ys = linspace(-1,1,H) are passed in so
 * that they are the reference's own fp32 values.  img [B,C,H,W], flow [B,2,H,W] (pixels).
 * idx_xy nullable int32 [B,H,W,2]; out nullable. */
MVF_API int mvf_flow_warp_fwd(const float *img, const float *flow, const float *xs, const float *ys,
                      float *out, int32_t *idx_xy, int B, int C, int H, int W, void *stream);
/* g_img nullable, zero-initialised by the caller (scatter-add); g_flow nullable [B,2,H,W]
 * (needs workspace of mvf_flow_warp_workspace_floats floats). */
MVF_API int mvf_flow_warp_bwd(const float *img, const float *flow, const float *xs, const float *ys,
                      const float *g_out, float *g_img, float *g_flow, float *workspace, int B, int C,
                      int H, int W, void *stream);
MVF_API size_t mvf_flow_warp_workspace_floats(int B, int C, int H, int W);
/* Deterministic g_img of IFRNet.warp (networks/IFRNet.py:7-15): the scatter of mvf_flow_warp_bwd
 * turned into a gather through sorted inverse tap lists (no float atomics, bit-reproducible).
 * g_img [B,C,H,W] is fully written.  workspace: mvf_fusion_bwd_workspace_ints(B, H, W) int32. */
MVF_API int mvf_flow_warp_bwd_gather(const float *flow, const float *xs, const float *ys, const float *g_out,
                             float *g_img, int32_t *workspace, int B, int C, int H, int W, void *stream);

/* ---- f1: FusionModule (networks/fusion_module.py:65-130) -----------
 * The tensor that enters the 1x1 convolution of one pyramid level,
 *   out [B, 2*(C+42), h, w] = cat[ feat_0 | emb(0) | m*(warp(feat_n1, fl_n1) | emb(e_n1))
 *                                                  + (1-m)*(warp(feat_p1, fl_p1) | emb(e_p1)) ],
 * in one launch (get_embedding_flow :65-78, warp_features :80-90, merge_features :92-103 and
 * the cats of forward :105-130).  flows [B,2,Hf,Wf] and merge mask [B,1,Hf,Wf] are the frozen
 * teacher's full-resolution outputs.
 * mvf_fusion_prep fills the per-level side tensor prep [B,9,h,w] = {e_n1 (2), e_p1 (2): cascaded
 * half-resolution flows, halved at every step; fl_n1 (2), fl_p1 (2): F.interpolate(flow, (h,w))
 * scaled by (w/Wf, h/Hf); m (1): F.interpolate(mask, (h,w))}.  The cascade continues from
 * prev_prep ([B,9,prev_h,prev_w], the level above) or starts at the flows when prev_prep is NULL;
 * halvings = 2 only for Lite-Mono's first level (fusion_module.py:71-74), else 1. */
MVF_API size_t mvf_fusion_prep_floats(int B, int h, int w);
MVF_API int mvf_fusion_prep(const float *flow_n1, const float *flow_p1, const float *mask,
                    const float *prev_prep, float *prep, int B, int h, int w, int Hf, int Wf, int prev_h,
                    int prev_w, int halvings, void *stream);
/* feat_* [B,C,h,w]; xs = linspace(-1,1,w), ys = linspace(-1,1,h) (the reference's fp32 values) */
MVF_API int mvf_fusion_level_fwd(const float *feat_0, const float *feat_n1, const float *feat_p1,
                         const float *prep, const float *xs, const float *ys, float *out, int B, int C,
                         int h, int w, void *stream);
/* g_out [B,2*(C+42),h,w] -> g_feat_n1 / g_feat_p1 [B,C,h,w] (nullable; zero-initialised by the
 * caller: scatter-add).  grad of feat_0 is g_out[:, :C] itself; flows and mask carry no gradient. */
MVF_API int mvf_fusion_level_bwd(const float *g_out, const float *prep, const float *xs, const float *ys,
                         float *g_feat_n1, float *g_feat_p1, int B, int C, int h, int w, void *stream);
/* The same gradients WITHOUT float atomics (bit-reproducible): an inverse tap list per destination
 * pixel is built from the flows (count / scan / fill / sort, integer work only) and every channel
 * gathers through it in a fixed order.  g_feat_* are overwritten (no zero-initialisation needed).
 * workspace: mvf_fusion_bwd_workspace_ints(B,h,w) int32. */
MVF_API size_t mvf_fusion_bwd_workspace_ints(int B, int h, int w);
MVF_API int mvf_fusion_level_bwd_gather(const float *g_out, const float *prep, const float *xs, const float *ys,
                                float *g_feat_n1, float *g_feat_p1, int32_t *workspace, int B, int C, int h,
                                int w, void *stream);
/* Round 5: the deterministic adjoint through ANCHOR lists (an output pixel listed once, under the top-left cell of the
 * 2x2 block it touches; its channel values fetched once instead of once per touched cell), the lists of ALL pyramid
 * levels of a step built by one count / scan / fill / sort pass: they depend only on prep (the teacher's flows).
 * lists[i]: mvf_fusion_lists_level_ints(B, hs[i], ws[i]) int32, 16-byte aligned, kept until the level's backward;
 * scratch: mvf_fusion_lists_scratch_ints(...) int32, free after the call (stream order).
 * mvf_fusion_level_bwd_lists == mvf_fusion_level_bwd_gather within rounding (another fixed order of the same
 * additions), bit-reproducible from run to run. */
MVF_API size_t mvf_fusion_lists_level_ints(int B, int h, int w);
MVF_API size_t mvf_fusion_lists_scratch_ints(int B, int n_levels, const int32_t *hs, const int32_t *ws);
MVF_API int mvf_fusion_lists_build(const float *const *preps, const float *const *xs, const float *const *ys,
                           const int32_t *hs, const int32_t *ws, int n_levels, int B, int32_t *const *lists,
                           int32_t *scratch, void *stream);
MVF_API int mvf_fusion_level_bwd_lists(const float *g_out, const int32_t *lists, float *g_feat_n1, float *g_feat_p1,
                               int B, int C, int h, int w, void *stream);

/* ---- f2: Trainer.compute_SI_log_depth_loss (train.py:924-941) ----
 * pred, target [B,1,H,W] (N = H*W), mask nullable [B,1,H,W] (same shape; any batch size).
 * loss[0] = mean_b( sum ld^2/n - beta*(sum ld)^2/n^2 ), ld = log(pred+1e-7)*m - log(target+1e-7)*m,
 * n = sum m + 1e-8.  sums [B,4] = {sum ld, sum ld^2, n, -} saved for the backward.
 * workspace: B*64*4 floats. */
MVF_API int mvf_silog_fwd(const float *pred, const float *target, const float *mask, float *loss,
                  float *sums, float *workspace, int B, int N, float beta, void *stream);
/* g_pred / g_target nullable; g_loss device scalar */
MVF_API int mvf_silog_bwd(const float *pred, const float *target, const float *mask, const float *sums,
                  const float *g_loss, float *g_pred, float *g_target, int B, int N, float beta,
                  void *stream);

/* The SI-log losses of a step (train.py:813-815, 868-882: nine per step) as one partial + one finishing launch forward and ONE launch backward.
 * A job reads image b of pred / target / mask at base + b * stride (floats; 0 = N: contiguous), so the depth views of a
 * grouped decoder call are read in place.  losses [n_jobs], total [1] = their sum in job order, sums [n_jobs,B,4];
 * workspace: mvf_silog_many_workspace_floats(n_jobs, B) floats; tickets: 1 int32, ZERO on entry, left zero.
 * Backward: upstream gradient of job j = *g_total (nullable) + g_losses[j] (nullable), g_pred / g_target [B,N]
 * contiguous, nullable per job.  Same arithmetic per job as mvf_silog_fwd / mvf_silog_bwd. */
#define MVF_MAX_SILOG_JOBS 16
typedef struct mvf_silog_job {
    const float *pred;   int64_t pred_stride;
    const float *target; int64_t target_stride;
    const float *mask;   int64_t mask_stride;         /* nullable */
    float *g_pred, *g_target;                         /* backward only */
} mvf_silog_job;
MVF_API size_t mvf_silog_many_workspace_floats(int n_jobs, int B);
MVF_API int mvf_silog_many_fwd(const mvf_silog_job *jobs, int n_jobs, float *losses, float *total, float *sums,
                       float *workspace, int32_t *tickets, int B, int N, float beta, void *stream);
MVF_API int mvf_silog_many_bwd(const mvf_silog_job *jobs, int n_jobs, const float *sums, const float *g_total,
                       const float *g_losses, int B, int N, float beta, void *stream);

/* ---- f2: the affine-augmentation glue ---------------------------
 * Trainer.affine_transform (train.py:888-902): per sample rotate(img, angle) (torchvision
 * functional.rotate, bilinear, zero fill), crop box = (x0, y0, w, h), F.interpolate back to
 * (H,W) (bilinear, align_corners=False).  The reference loops over the batch with five
 * .item() host syncs per sample; here angle_deg [B] (degrees), box [B,4] int32 and ratio [B]
 * are device arrays and one launch covers the batch.  img/out [B,C,H,W].  Forward only
 * (the trainer applies it to teacher frames, which carry no gradient).  The box must lie
 * inside the image (the reference's slicing / paste assume it too). */
MVF_API int mvf_affine_transform_fwd(const float *img, const float *angle_deg, const int32_t *box,
                             float *out, int B, int C, int H, int W, void *stream);
/* the same for B images that are views of B_meta samples (image b uses angle / box of sample b % B_meta): the frames
 * train.py:832-833 transforms one call each, as one launch over their concatenation */
MVF_API int mvf_affine_transform_views_fwd(const float *img, const float *angle_deg, const int32_t *box, float *out,
                                   int B, int B_meta, int C, int H, int W, void *stream);
/* depth_restore of Trainer.compute_depth_consistency_loss_affine (train.py:909-916):
 * out = ratio[b] * rotate( zeros(H,W) with F.interpolate(depth -> (h,w)) pasted at (x0,y0),
 * -angle[b] ).  depth/out [B,C,H,W]. */
MVF_API int mvf_affine_restore_fwd(const float *depth, const float *angle_deg, const int32_t *box,
                           const float *ratio, float *out, int B, int C, int H, int W, void *stream);
/* the same with the C planes of image b at depth + b * image_stride (floats; 0 = C*H*W): several depth maps that lie
 * one plane apart -- consecutive groups of a grouped decoder call's interleaved output -- restored by ONE launch as the
 * channels of one image (they share angle / box / ratio: train.py:868-882 restores three per step) */
MVF_API int mvf_affine_restore_strided_fwd(const float *depth, int64_t image_stride, const float *angle_deg,
                                   const int32_t *box, const float *ratio, float *out, int B, int C, int H,
                                   int W, void *stream);
/* g_depth = adjoint of the above applied to g_out; deterministic (two gather passes, no
 * atomics).  workspace: B*C*H*W floats. */
MVF_API int mvf_affine_restore_bwd(const float *g_out, const float *angle_deg, const int32_t *box,
                           const float *ratio, float *workspace, float *g_depth, int B, int C,
                           int H, int W, void *stream);

/* ---- Conv3x3's ReflectionPad2d(1) (layers.py:121-138) -------------------------------------
 * in [planes,H,W] -> out [planes,H+2,W+2] (planes = B*C); backward is a deterministic gather.
 * H, W >= 2 (ATen's reflection_pad2d requires pad < size). */
MVF_API int mvf_reflect_pad1_fwd(const float *in, float *out, int planes, int H, int W, void *stream);
MVF_API int mvf_reflect_pad1_bwd(const float *g_out, float *g_in, int planes, int H, int W, void *stream);

/* ---- Regrouping of interleaved group batches (section 8f-3; reference: train.py:745-747, 788-797, 830-868 hand
 * each encoder call's feature pyramid to the decoder / fusion calls that use it) ---------------
 * src [B*G, chunk] holds G independent invocations interleaved (sample n = b*G + g).  Output k is
 * the interleaved batch of counts[k] of those groups: dst[k] sample b*counts[k] + j = src sample
 * b*G + groups[offset_k + j] (groups = the per-output lists concatenated; a group may appear in
 * several outputs and several times in one).  dst / counts / groups are HOST arrays (n_out <= 8,
 * at most 32 slots in total, G <= 32); ONE launch for all outputs. */
MVF_API int mvf_regroup_fwd(const float *src, int G, int B, int64_t chunk, int n_out, float *const *dst,
                    const int32_t *counts, const int32_t *groups, void *stream);
/* adjoint: g_src [B*G, chunk] = per source group the sum of the gradients of the slots that read
 * it, in (output, position) order; groups nobody read (or whose outputs have g_dst[k] == NULL:
 * not differentiated) get zeros.  Written once, deterministic.  g_strides (HOST, nullable = chunk):
 * elements between consecutive samples of g_dst[k] (>= chunk: a gradient that is a channel slice
 * of a wider tensor is read in place). */
MVF_API int mvf_regroup_bwd(const float *const *g_dst, const int64_t *g_strides, int G, int B, int64_t chunk,
                    int n_out, const int32_t *counts, const int32_t *groups, float *g_src, void *stream);
/* Input side of a grouped call: n_slots separate tensors src[s] [B, len[s]] -> dst [B*G, total],
 * sample b*G + group[s] holding part s at element offset[s] (e.g. the two frames of each of the
 * six pose pairs of a step, train.py:943-946 / 724-731, as one [B*6, 6*H*W] batch).  HOST arrays,
 * n_slots <= 32; forward only (the parts are images). */
MVF_API int mvf_interleave_fwd(const float *const *src, const int64_t *len, const int64_t *offset,
                       const int32_t *group, int n_slots, float *dst, int G, int B, int64_t total,
                       void *stream);

/* Grouped batch norm (networks/grouped.py; the reference normalises every encoder call on its
 * own, train.py:745-868 through nn.BatchNorm2d / SyncBatchNorm train.py:207): the C-vectors of a
 * layer repeated for the G interleaved calls, tiled [4][G*C] = weight | bias | running_mean |
 * running_var.  One launch instead of stack + repeat. */
MVF_API int mvf_bn_tile(const float *weight, const float *bias, const float *running_mean,
                const float *running_var, float *tiled, int C, int G, void *stream);
/* running <- beta * running + sum_g coef[g] * upd[g*C + c] for both statistics (upd = the tiled
 * statistics after the batch-norm call: one momentum update per group from the common start;
 * coef, beta as GroupedBatchNorm2d._fold_running derives them: the G sequential updates of the
 * per-call form), *num_batches_tracked += G (nullable).  coef: HOST array of G <= 32 floats. */
MVF_API int mvf_bn_fold_running(float *running_mean, float *running_var, const float *upd_mean,
                        const float *upd_var, const float *coef, float beta, int C, int G,
                        int64_t *num_batches_tracked, void *stream);
/* adjoint of the tiling: g_out [2][C] = per channel the sum over the G groups of the tiled weight /
 * bias gradients, in group order. */
MVF_API int mvf_bn_untile(const float *g_weight_tiled, const float *g_bias_tiled, float *g_out, int C, int G,
                  void *stream);

/* out = act(((t0 + t1) + t2) + ...) over `total` elements, act in {0 none, 2 relu}: the branch sum of
 * an HRNet fuse layer (networks/hrnet_encoder.py:267-285: y = y + term per branch, self.relu(y)) in one
 * pass.  terms: HOST array of n_terms <= 8 device pointers.  Left-to-right sum (bit-identical to
 * the term-at-a-time form); the adjoint is mvf_bias_act_bwd's activation pass, shared by all terms. */
MVF_API int mvf_sum_act_fwd(const float *const *terms, int n_terms, float *out, int64_t total, int act,
                    void *stream);

/* The same two steps for ALL grouped batch-norm layers of a network at once.  rows: DEVICE array of
 * n_layers records of eight 8-byte fields { const float *weight, *bias; float *running_mean,
 * *running_var, *tiled [4][G*C]; int64_t *num_batches_tracked (nullable); int64_t C; int64_t pad }.
 * mvf_bn_tile_many fills every layer's `tiled` (when the grouped call begins); mvf_bn_fold_many
 * folds rows 2, 3 of every `tiled` into the running statistics and advances the counters (when
 * it ends).  max_channels = the largest C of the table. */
MVF_API int mvf_bn_tile_many(const void *rows, int n_layers, int max_channels, int G, void *stream);
MVF_API int mvf_bn_fold_many(const void *rows, int n_layers, int max_channels, const float *coef, float beta,
                     int G, void *stream);

/* ---- f4: step glue either side of the hot path ------------------
 * Decoder stage glue (networks/monodepth2.py:84-90 with layers.py:121-138, 225-228): the padded
 * input of upconv_1, out [B, C1+C2, 2h+2, 2w+2] = ReflectionPad2d(1)(cat([upsample_nearest_x2(x),
 * skip], 1)), written once from x [B,C1,h,w] and skip [B,C2,2h,2w] (skip NULL when C2 == 0). */
MVF_API int mvf_up2cat_pad_fwd(const float *x, const float *skip, float *out, int B, int C1, int C2, int h,
                       int w, void *stream);
/* adjoint: g_x [B,C1,h,w] and g_skip [B,C2,2h,2w] (either nullable), deterministic gathers; h, w >= 2 */
MVF_API int mvf_up2cat_pad_bwd(const float *g_out, float *g_x, float *g_skip, int B, int C1, int C2, int h,
                       int w, void *stream);
/* Round 5: the same two pad kernels with the PRODUCER's epilogue applied on load -- the input is a raw convolution
 * output y [B,C,...] and the padded tensor holds ELU(y + bias[c]) (ConvBlock, layers.py:106-118, followed by the next
 * Conv3x3's pad, layers.py:121-138, or by upsample + cat + pad, networks/monodepth2.py:84-90): the activated tensor
 * is written once, already padded.  Backward: g_in = pad_adjoint(g_padded) * ELU'(interior of padded), g_bias[c] = its
 * sum over batch and pixels (deterministic fold); for up2cat only the x part is activated, g_skip as before.
 * Wide shapes only: mvf_pad_act_supported(H, W) of the PADDED operation's unpadded size (2h, 2w for up2cat) and
 * 16-byte aligned bases; anything else returns hipErrorInvalidValue (the caller keeps the separate kernels).
 * workspace: mvf_pad_act_workspace_floats(B, C, H, W) / mvf_up2cat_pad_act_workspace_floats(B, C1, h, w) floats. */
MVF_API int mvf_pad_act_supported(int H, int W);
MVF_API size_t mvf_pad_act_workspace_floats(int B, int C, int H, int W);
MVF_API int mvf_reflect_pad1_act_fwd(const float *in, const float *bias, float *out, int B, int C, int H, int W,
                             void *stream);
MVF_API int mvf_reflect_pad1_act_bwd(const float *g_padded, const float *padded, float *g_in, float *g_bias,
                             float *workspace, int B, int C, int H, int W, void *stream);
MVF_API int mvf_up2cat_pad_act_fwd(const float *x, const float *bias, const float *skip, float *out, int B, int C1,
                           int C2, int h, int w, void *stream);
MVF_API size_t mvf_up2cat_pad_act_workspace_floats(int B, int C1, int h, int w);
MVF_API int mvf_up2cat_pad_act_bwd(const float *g_padded, const float *padded, float *g_x, float *g_bias, float *g_skip,
                           float *workspace, int B, int C1, int C2, int h, int w, void *stream);
/* Disparity head (networks/monodepth2.py:93 followed by layers.py:16-25): disp = sigmoid(logit),
 * depth = 1/(min_disp + range*disp) (nullable), mean_partials (nullable) [B,32] = the per-image
 * partial sums of disp that mvf_unit_fwdbwd otherwise computes itself.  logit [B,N]. */
MVF_API int mvf_disp_head_fwd(const float *logit, float *disp, float *depth, float *mean_partials, int B, int N,
                      float min_disp, float range, void *stream);
/* g_logit = (g_disp - g_depth*range*depth^2) * disp*(1-disp); g_disp / g_depth nullable */
MVF_API int mvf_disp_head_bwd(const float *disp, const float *g_disp, const float *g_depth, float *g_logit,
                      int64_t n, float min_disp, float range, void *stream);
/* The same adjoint fed by the hot-path units directly (reference: the autograd edge between compute_losses_base,
 * train.py:987-1051, and the decoder's sigmoid, networks/monodepth2.py:93): a unit's disparity gradient is taken RAW,
 * as mvf_units_fwdbwd left it in g_disp_raw, and g_disp = (raw - shift_b) * (*g_loss + *g_sum) -- the formula of
 * mvf_units_fwdbwd_scale below, same operations, same bits -- is applied on load; the unit's B images are images
 * first, first + step, ... of the head's batch [B_head, N] (the interleaved batch of a grouped decoder call).
 * g_disp (nullable): gradient of disp from any other consumer, added first.  n_units <= MVF_MAX_UNITS. */
typedef struct mvf_head_unit_grad {
    const float *g_disp_raw; int64_t raw_stride;      /* [count, N]; stride in floats, 0 = N */
    const float *stats;                               /* [count, 4] of the unit (mvf_units_fwdbwd) */
    const float *g_loss, *g_sum;                      /* device scalars, at least one */
    float smoothness;
    int32_t first, step, count;
} mvf_head_unit_grad;
MVF_API int mvf_disp_head_bwd_units(const float *disp, const float *g_disp, const float *g_depth, float *g_logit,
                            int B, int N, float min_disp, float range, const mvf_head_unit_grad *units,
                            int n_units, void *stream);
/* Convolution epilogue: out = act(x + bias[c] (+ res)) in one pass over [N,C,HW] (x may alias out).
 * Replaces the bias add_ + activation (+ residual add) that follow every biased convolution:
 * decoder ConvBlock (layers.py:106-118: ELU), IFRNet convrelu / ResBlock (networks/IFRNet.py:128-157:
 * PReLU, block input added before the last one), transposed convolutions / 1x1 merges (bias only).
 * act: 0 none, 1 ELU(alpha=1), 2 ReLU, 3 PReLU with slope [slope_n], slope_n == C or 1.
 * bias, slope, res nullable. */
MVF_API int mvf_bias_act_fwd(const float *x, const float *bias, const float *slope, const float *res, float *out,
                     int N, int C, int HW, int act, int slope_n, void *stream);
/* adjoint from the RESULT (act 0..2): g_x = g * act'(out) (act none: g_x is g, not written; out / g_x
 * nullable) and g_bias[c] = sum of g_x over n, hw -- one pass + a fold of fixed-order partials
 * (deterministic).  g_bias NULL: only g_x (workspace unused).  workspace:
 * mvf_bias_act_workspace_floats(N, C, HW) floats. */
MVF_API size_t mvf_bias_act_workspace_floats(int N, int C, int HW);
MVF_API int mvf_bias_act_bwd(const float *g, const float *out, float *g_x, float *g_bias, float *workspace, int N,
                     int C, int HW, int act, void *stream);
/* F.interpolate(x, mode="bilinear", align_corners=...) over [planes, ih, iw] -> [planes, oh, ow]
 * (HRNet fuse layers networks/hrnet_encoder.py:275-280: align_corners=True, coarse -> fine branch;
 * Lite-Mono decoder networks/LiteMono.py:495, 502 through layers.py:225-228 `upsample`:
 * scale_factor=2, align_corners=False).  scale_h / scale_w are ATen's area_pixel_compute_scale values:
 * align_corners ? (in-1)/(out-1) (0 when out == 1) : (1/scale_factor when one was given, else
 * in/out), as fp32.  One lane per output element; planes <= 262,140. */
MVF_API int mvf_resize_bilinear_fwd(const float *x, float *out, int planes, int ih, int iw, int oh, int ow,
                            float scale_h, float scale_w, int align_corners, void *stream);
/* adjoint: deterministic gather (no float atomics), g_x [planes, ih, iw] fully written.  With a
 * workspace of mvf_resize_bilinear_bwd_workspace_floats(planes, ih, ow) floats: two separable
 * passes (rows of g_out folded into the workspace, then its columns); workspace NULL: one pass. */
MVF_API size_t mvf_resize_bilinear_bwd_workspace_floats(int planes, int ih, int ow);
MVF_API int mvf_resize_bilinear_bwd(const float *g_out, float *g_x, float *workspace, int planes, int ih, int iw,
                            int oh, int ow, float scale_h, float scale_w, int align_corners, void *stream);
/* F.interpolate(x, scale_factor=f, mode="nearest") for an integer factor (layers.py:225-228
 * `upsample`; DHRNet decoder networks/DHRNet.py branch merges): [planes, ih, iw] -> [planes, ih*f, iw*f];
 * adjoint = the f x f block sum (gather, deterministic). */
MVF_API int mvf_upsample_nearest_fwd(const float *x, float *out, int planes, int ih, int iw, int factor, void *stream);
MVF_API int mvf_upsample_nearest_bwd(const float *g_out, float *g_x, int planes, int ih, int iw, int factor,
                             void *stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet trunks (networks/monodepth2.py:39
 * `self.encoder.maxpool`, networks/posenet.py:21, 87): x [planes, H, W] -> out [planes, OH, OW],
 * OH = (H-1)/2 + 1, OW = (W-1)/2 + 1; idx [planes, OH, OW] uint8 = window-local position (kh*3 + kw
 * relative to (2*oy-1, 2*ox-1)) of the maximum, selected as ATen does (first maximum in kh, kw scan
 * order; a NaN wins).  Backward: g_x [planes, H, W] fully written, deterministic gather over the
 * <= 4 windows of an input pixel in ATen's accumulation order.  planes <= 262,140. */
MVF_API int mvf_maxpool3s2_fwd(const float *x, float *out, uint8_t *idx, int planes, int H, int W, void *stream);
MVF_API int mvf_maxpool3s2_bwd(const float *g_out, const uint8_t *idx, float *g_x, int planes, int H, int W,
                       void *stream);
/* the same adjoint with the gradient the pooled tensor received from its OTHER consumer added in
 * the pass: g_x = addend + gather (the stem output of the depth encoder is both pooled and a
 * feature of the pyramid, monodepth2.py:36-41: autograd's separate accumulation pass read both
 * full-size tensors again).  addend [planes,H,W], may alias nothing else. */
MVF_API int mvf_maxpool3s2_bwd_add(const float *g_out, const uint8_t *idx, const float *addend, float *g_x,
                           int planes, int H, int W, void *stream);
/* On-device colour augmentation of the data pipeline (datasets/mono_dataset.py:102-184, 214-256:
 * do_flip, do_color_aug with one torchvision ColorJitter draw per sample applied to all of its
 * frames).  img [samples*frames,3,H,W] (frame-minor), factors [samples,4] = {brightness, contrast,
 * saturation, hue}, order [samples,4] = the permutation of {0 brightness, 1 contrast, 2 saturation,
 * 3 hue}, apply / flip [samples] int32 flags.  out_raw (nullable) = the flipped frames, out_aug =
 * flipped + jittered.  workspace: mvf_color_jitter_workspace_floats(samples*frames) floats. */
MVF_API size_t mvf_color_jitter_workspace_floats(int images);
MVF_API int mvf_color_jitter(const float *img, const float *factors, const int32_t *order, const int32_t *apply,
                     const int32_t *flip, float *out_raw, float *out_aug, float *workspace, int samples,
                     int frames, int H, int W, void *stream);

/* ---- measurement hooks (bench.py) ------------------------------------------------------
 * When enabled, the library brackets launches of its kernels with a pair of HIP events recorded on the
 * launch stream.  mvf_profile_read() synchronises the recorded events and returns the summed kernel time
 * and launch count for one kernel id since the last reset.  Off by default; the only process-global state
 * in the library.  Levels: 1 = the hot-path tile kernels (ids 0..6: what the timed region of bench.py runs
 * with -- six event records per step), 2 = every kernel id below (the glue kernels either side of the path
 * too: some hundred event pairs per step, used by bench.py's separate kernel-roofline leg).
 * `work`: ids 0..6 count PIXELS (images x H x W summed over the units of each launch); ids >= 7 count the
 * ALGORITHMIC BYTES of each launch (every input element read once + every output element written once). */
#define MVF_PROF_UNIT_FWD 0  /* fused unit forward tile kernel  */
#define MVF_PROF_UNIT_BWD 1  /* fused unit backward tile kernel */
#define MVF_PROF_PHOTO_FWD 2 /* staged compute_losses_base forward */
#define MVF_PROF_PHOTO_BWD 3
#define MVF_PROF_WARP_FWD 4  /* staged generate_images_pred */
#define MVF_PROF_WARP_BWD 5
#define MVF_PROF_UNIT_FWDBWD 6 /* forward+backward of a unit in one tile kernel */
#define MVF_PROF_UNITS_FINISH 7      /* k_units_finish */
#define MVF_PROF_FB_SCALE 8          /* k_fb_scale */
#define MVF_PROF_DISP_MEAN 9         /* k_disp_mean (only when the disparity head did not supply the partials) */
#define MVF_PROF_BIAS_ACT_FWD 10     /* k_bias_act_fwd */
#define MVF_PROF_BIAS_ACT_BWD 11     /* k_bias_act_bwd + k_bias_grad_finish */
#define MVF_PROF_ACT_BWD_FLAT 12     /* k_act_bwd_flat */
#define MVF_PROF_UP2CAT_FWD 13       /* k_up2cat_pad_fwd */
#define MVF_PROF_UP2CAT_BWD_X 14     /* k_up2cat_pad_bwd_x */
#define MVF_PROF_UP2CAT_BWD_SKIP 15  /* k_up2cat_pad_bwd_skip */
#define MVF_PROF_REFLECT_PAD_FWD 16  /* k_reflect_pad1_fwd */
#define MVF_PROF_REFLECT_PAD_BWD 17  /* k_reflect_pad1_bwd */
#define MVF_PROF_MAXPOOL_FWD 18      /* k_maxpool3s2_fwd */
#define MVF_PROF_MAXPOOL_BWD 19      /* k_maxpool3s2_bwd */
#define MVF_PROF_FUSION_FWD 20       /* k_fusion_level_fwd */
#define MVF_PROF_FUSION_BWD_GATHER 21 /* inverse tap lists + k_fusion_level_bwd_gather */
#define MVF_PROF_FLOW_WARP_FWD 22    /* k_flow_warp_fwd */
#define MVF_PROF_DISP_HEAD_FWD 23    /* k_disp_head_fwd */
#define MVF_PROF_DISP_HEAD_BWD 24    /* k_disp_head_bwd */
#define MVF_PROF_RESIZE_FWD 25       /* k_resize_bilinear_fwd */
#define MVF_PROF_RESIZE_BWD 26       /* k_resize_bilinear_bwd */
#define MVF_PROF_NEAREST_FWD 27      /* k_upsample_nearest_fwd */
#define MVF_PROF_NEAREST_BWD 28      /* k_upsample_nearest_bwd */
#define MVF_PROF_SILOG_FWD 29        /* k_silog_partial + k_silog_finish */
#define MVF_PROF_SILOG_BWD 30        /* k_silog_bwd */
#define MVF_PROF_AFFINE 31           /* affine transform / restore kernels */
#define MVF_PROF_REGROUP_FWD 32      /* k_regroup_fwd */
#define MVF_PROF_REGROUP_BWD 33      /* k_regroup_bwd */
#define MVF_PROF_INTERLEAVE_FWD 34   /* k_interleave_fwd */
#define MVF_PROF_SUM_ACT_FWD 35      /* k_sum_act_fwd */
#define MVF_PROF_COUNT 36
/* launch tags of MVF_PROF_UNIT_FWDBWD (mvf_profile_read_launches): what kind of unit group a launch carried */
#define MVF_TAG_SINGLE_FRAME 0  /* identity candidates evaluated (and possibly handed over: ident_out) */
#define MVF_TAG_MULTI_FRAME 1   /* identity maps taken from another unit (ident_in) */
#define MVF_TAG_AFFINE 2        /* mask_rec supplied */
#define MVF_TAG_MIXED 3         /* units with and without mask_rec in one launch (single-frame + affine groups) */
MVF_API int mvf_profile_enable(int level);
MVF_API int mvf_profile_reset(void);
MVF_API int mvf_profile_read(int kernel_id, double *total_ms, int64_t *launches);
/* work the recorded launches processed: pixels for ids 0..6, algorithmic bytes for ids >= 7 */
MVF_API int mvf_profile_read_work(int kernel_id, int64_t *work);
/* per-launch records of one kernel id since the last reset, oldest first: duration (ms), work, tag.  Returns
 * the number of records copied (<= cap), or a negative HIP error.  (Medians per launch type: SURVEY 8d.) */
MVF_API int64_t mvf_profile_read_launches(int kernel_id, double *ms, int64_t *work, int32_t *tag, int64_t cap);
/* kernel name of a profile id (static string) */
MVF_API const char *mvf_profile_name(int kernel_id);

/* floats of scratch the reducing entry points need for a [B,*,H,W] problem */
MVF_API size_t mvf_workspace_floats(int B, int H, int W);

#ifdef __cplusplus
}
#endif
#endif /* MVF_HOTPATH_H */
