/*
 * sfmwarp.h -- C ABI of libsfmwarp.so: the MI355X (gfx950) implementation of SfM-Learner's
 * photometric view-synthesis loss path.
 *
 * Every entry point replaces a piece of pfnet/sfm-learner-chainer (citations are
 * file:line into that repository).  Conventions:
 *
 *   - all tensors are float32, C-contiguous, NCHW, resident in device (HBM) memory and owned
 *     by the caller; the library never allocates, frees or keeps a pointer after return;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call is
 *     asynchronous with respect to the host, re-entrant, and keeps no global mutable state
 *     (the reference's module globals `filler` / `meshgrid`, models/transform.py:62,135, are
 *     deliberately not reproduced).  Per THREAD the library remembers the launch plans of the last
 *     four distinct (descriptor bytes, device, entry point) combinations of sfm_loss_*: host-side
 *     integers only, no device memory, never a pointer that is used without being passed in again;
 *   - image values: the reference's input contract is uint8 / 127.5 - 1, i.e. [-1, 1]
 *     (datasets/kitti/kitti_raw_dataset.py:12-14).  Warp, zero mask (models/base_model.py:96), L1 and smoothness hold for finite
 *     images of any range.  The SSIM terms (models/base_model.py:126-142) form variances as E[x^2] - mu^2 in fp32, as the
 *     reference does: beyond a range of about +-16 those cancel, SSIM's denominator can reach 0, and where the reference's
 *     F.clip backward then yields a zero gradient this library may yield a non-finite one (the loss scalars stay right).
 *     A NaN anywhere in an image makes the loss NaN, as in the reference.  Image values are taken to be zero or NORMAL
 *     floats: the zero mask of models/base_model.py:96 is formed as clamp(max_c |I^_c| * 2^127, 0, 1), which is exactly 0 / 1 for those
 *     and a fraction of 1 for a pixel whose three warped channels are all subnormal (|.| < 1.2e-38; uint8 / 127.5 - 1 has none);
 *   - return value: 0 on success; SFM_ERR_* (<0) for a rejected argument; a positive value is
 *     a hipError_t from the launch.  sfm_last_error() returns a thread-local message.
 *     No exception or abort crosses the ABI.
 */
#ifndef SFMWARP_H_
#define SFMWARP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_ABI_VERSION 5

#define SFM_OK 0
#define SFM_ERR_NULL (-1)      /* a required pointer is NULL                       */
#define SFM_ERR_SHAPE (-2)     /* a dimension is out of the supported range        */
#define SFM_ERR_CONFIG (-3)    /* inconsistent loss configuration                  */
#define SFM_ERR_WORKSPACE (-4) /* workspace missing or too small                   */

#define SFM_MAX_SCALES 8
#define SFM_MAX_SRC 8

int sfm_abi_version(void);
const char *sfm_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Pose -> projection.  proj_tgt_to_src(), models/transform.py:64-91 (euler2mat :11-40,
 * pose_vec2mat :43-59), kept on the device (the reference hops to the CPU, :76-80,89-90).
 *   pose6 (N,6) rx,ry,rz,tx,ty,tz ; K (N,3,3)  ->  proj (N,4,4) = [[K,0],[0,1]] . [[R,t],[0,1]]
 * Backward: g_proj (N,4,4) -> d_pose6 (N,6) (overwritten).
 * ---------------------------------------------------------------------------------------- */
int sfm_pose_proj_fwd(const float *pose6, const float *K, float *proj, int N, void *stream);
int sfm_pose_proj_bwd(const float *pose6, const float *K, const float *g_proj, float *d_pose6, int N,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * projective_inverse_warp(imgs, depthes, poses, K), models/transform.py:156-193
 * (pixel2cam :94-109, cam2pixel :111-133 incl. the x2 rule :128-131, and the
 * F.spatial_transformer_sampler call :189).
 *   src (N,C,H,W) ; depth (N,depth_rows,H*W) ; pose6 (N,6) ; K (N,3,3)  ->  warped (N,C,H,W).
 *   depth_rows = 3: the reference's `depthes` (N,3,H*W), one depth per camera coordinate (:107);
 *   depth_rows = 1: ONE row of it, for the case the caller knows the three rows are the
 *   broadcast of models/base_model.py:82-84 (d_depth is then the sum over the three rows).
 * Backward for an upstream gradient g_warped (N,C,H,W):
 *   d_depth (N,depth_rows,H*W) overwritten
 *   d_pose6 (N,6)     overwritten
 *   d_src   (N,C,H,W) or NULL; ACCUMULATED into (zero it first) with float atomics
 *   ws: sfm_warp_bwd_workspace_bytes(N,H,W) bytes of scratch.
 * H, W >= 3 (below that the reference's x2 rule no longer implies zero fill).
 * ---------------------------------------------------------------------------------------- */
int sfm_warp_fwd(const float *src, const float *depth, int depth_rows, const float *pose6, const float *K,
                 float *warped, int N, int C, int H, int W, void *stream);
size_t sfm_warp_bwd_workspace_bytes(int N, int H, int W);
int sfm_warp_bwd(const float *src, const float *depth, int depth_rows, const float *pose6, const float *K,
                 const float *g_warped, float *d_depth, float *d_pose6, float *d_src, void *ws,
                 size_t ws_bytes, int N, int C, int H, int W, void *stream);

/* ------------------------------------------------------------------------------------------
 * F.spatial_transformer_sampler(x, grid) as called at models/transform.py:189 (Chainer
 * built-in): normalized grid (N,2,oH,oW) with [:,0]=x, [:,1]=y in [-1,1]; bilinear; the image
 * is zero-padded by one pixel and coordinates are clipped to the padded image.
 *   fwd: x (N,C,H,W), grid -> y (N,C,oH,oW)
 *   bwd: gy -> ggrid (N,2,oH,oW) overwritten ; gx (N,C,H,W) or NULL, ACCUMULATED (atomics)
 * ---------------------------------------------------------------------------------------- */
int sfm_sampler_fwd(const float *x, const float *grid, float *y, int N, int C, int H, int W, int oH, int oW,
                    void *stream);
int sfm_sampler_bwd(const float *x, const float *grid, const float *gy, float *ggrid, float *gx, int N,
                    int C, int H, int W, int oH, int oW, void *stream);

/* ------------------------------------------------------------------------------------------
 * SpatialTransformerSamplerInterp, models/spational_transformer_sampler_interp.py:9-159:
 * grid in PIXEL coordinates; weights formed after clipping the taps (:46-55), so the result is
 * exactly 0 outside [0,W-1) x [0,H-1); backward returns ggrid (:129-147) and gx == 0 (:148).
 *   fwd: _forward :32-78 ; bwd: _backward :86-149 (gx, if not NULL, is zero-filled).
 * ---------------------------------------------------------------------------------------- */
int sfm_sampler_interp_fwd(const float *x, const float *grid, float *y, int N, int C, int H, int W, int oH,
                           int oW, void *stream);
int sfm_sampler_interp_bwd(const float *x, const float *grid, const float *gy, float *ggrid, float *gx,
                           int N, int C, int H, int W, int oH, int oW, void *stream);

/* ------------------------------------------------------------------------------------------
 * The fused multi-scale loss: the loop of SFMLearner.__call__, models/base_model.py:69-124,
 * from the image pyramid onwards -- depth = 1/disp (:60), projective_inverse_warp (:90-94),
 * L1 + zero mask (:95-100,:111), SSIM (:112-115, compute_ssim :126-142), smoothness
 * (:75-77 compute_smooth_loss :169-185, or the edge-aware compute_disp_smooth :144-155 that
 * the reference leaves commented out at :78-80), explainability (:103-109, :157-167) and the
 * assembly of the five reported scalars (:117-123).
 * ---------------------------------------------------------------------------------------- */
#define SFM_SMOOTH_NONE 0
#define SFM_SMOOTH_SECOND_ORDER 1 /* base_model.py:169-185 (the live one) */
#define SFM_SMOOTH_EDGE_AWARE 2   /* base_model.py:144-155                */

/* Memory layout of the image pyramids handed to sfm_loss_* (tgt[s], src[s]):
 *   SFM_LAYOUT_PLANAR  the reference's: tgt (B,3,h,w), src (B,3*n_src,h,w)   (base_model.py:71-72)
 *   SFM_LAYOUT_HWC     pixel-interleaved: tgt (B,h,w,3), src (B,n_src,h,w,3) -- what sfm_pyramid_hwc_fwd writes.
 * Same values, same results (bit for bit); HWC lets the kernel fetch the three channels of a tap with one
 * 12-byte load (6 instead of 10 vector loads per pixel row) and is the layout the fused loss runs fastest on.
 * Disparities, masks and every gradient output (d_src included) are planar in both cases. */
#define SFM_LAYOUT_PLANAR 0
#define SFM_LAYOUT_HWC 1

/* How the fused kernels evaluate the per-pixel projection of projective_inverse_warp (models/transform.py:94-133,189).  (ABI v5)
 * Both take the pose -> projection products of proj_tgt_to_src and batch_inv in the reference's own roundings
 * (R = (X . Y) . Z, K4 . T and adj(K) / det as separate multiplies, adds and correctly rounded quotients: transform.py:11-91,105).
 *   SFM_PROJECTION_FAST             q = D (M . pix) + P[:,3] with M = P[:, :3] . K^-1 folded once per wavefront; the strict in-view
 *                                   test of transform.py:129 on U = q0 / z directly and the sample taken at (U, V).  16 vector
 *                                   instructions per pixel row.  GUARANTEES: the five loss scalars to 1e-4 relative; the warped pixels
 *                                   within 1e-4 of the image range of the reference's up to 128 x 416 frames; at 256 x 832, where one
 *                                   ulp of a sampling position is 6e-5 px, within 2e-4 with fewer than 0.01 % of the pixels beyond 1e-4.
 *   SFM_PROJECTION_REFERENCE_ORDER  the reference's own chain per pixel: ray = K^-1 . pix, c = D ray, q = Pm . (c, 1), U = q0 / z,
 *                                   xn = U / ((W-1)/2.) - 1, the strict test on xn, the sampler's position (xn + 1) (W-1) / 2 --
 *                                   every multiply, add and quotient rounded where the reference rounds it (quotients as v_rcp +
 *                                   residual correction: the IEEE result except in rare double-rounding cases).  45 vector
 *                                   instructions per pixel row: the launch is about 10 % longer.  GUARANTEES: warped pixels within
 *                                   1e-4 of the image range at ANY frame size, seams, large motion and samples behind the camera
 *                                   included, and gradients that need no second opinion (d_pose within 1e-5 of its maximum).
 * Both produce d_src (a second launch re-projects the pixels by the same chain). */
#define SFM_PROJECTION_FAST 0
#define SFM_PROJECTION_REFERENCE_ORDER 1

typedef struct SfmLossDesc {
  int32_t B;        /* samples held by this call (a batch shard)                              */
  int32_t norm_B;   /* batch size used in every mean: the GLOBAL batch when sharded, else B   */
  int32_t n_src;    /* seq_len - 1 (base_model.py:34)                                         */
  int32_t n_scales; /* len(pred_disps) (:66)                                                  */
  int32_t H[SFM_MAX_SCALES];
  int32_t W[SFM_MAX_SCALES];
  float smooth_reg;    /* config['smooth_reg'] (:37); 0 disables (:75)                        */
  float exp_reg;       /* config['exp_reg'] (:38); 0 disables (:86,:103)                      */
  float ssim_rate;     /* config.get('ssim_rate', 0) (:39)                                    */
  int32_t smooth_mode; /* SFM_SMOOTH_*                                                        */
  /* inputs */
  const float *tgt[SFM_MAX_SCALES];         /* (B,3,h,w)        curr_tgt_img  (:71)  [layout] */
  const float *src[SFM_MAX_SCALES];         /* (B,3*n_src,h,w)  curr_src_imgs (:72)  [layout] */
  const float *disp[SFM_MAX_SCALES];        /* (B,1,h,w)        pred_disps    (:59)           */
  const float *mask_logits[SFM_MAX_SCALES]; /* (B,n_src,h,w)    pred_maskes (:62) or NULL     */
  const float *intrinsics;                  /* (B,n_scales,3,3) (:85)                         */
  const float *pose[SFM_MAX_SRC];           /* (B,6)            pred_poses[i] (:62)           */
  /* gradient outputs (backward only) */
  float *d_disp[SFM_MAX_SCALES]; /* (B,1,h,w)     overwritten                                 */
  float *d_pose[SFM_MAX_SRC];    /* (B,6)         overwritten                                 */
  float *d_mask[SFM_MAX_SCALES]; /* (B,n_src,h,w) overwritten; required iff exp_reg != 0      */
  /* optional: dL/d(curr_src_imgs) (the reference computes it -- the sampler's gx, models/transform.py:189 -- and drops it,
   * base_model.py:71-72 `.data`).  Any subset of the scales may bind it.  Binding it adds a second kernel launch to
   * sfm_loss_bwd / sfm_loss_fwd_bwd (the pixels re-projected, their taps summed in on-chip memory) and, per bound scale,
   * B*3*n_src*h*w floats to sfm_loss_workspace_bytes (dL/d(warped pixel), recorded by the first launch). */
  float *d_src[SFM_MAX_SCALES];  /* (B,3*n_src,h,w) or NULL; ACCUMULATED (atomics)            */
  int32_t image_layout;          /* SFM_LAYOUT_* of tgt[] and src[]                           */
  /* optional output of sfm_loss_fwd and sfm_loss_fwd_bwd (ignored by sfm_loss_bwd): the warped source images the loss
   * was computed on, curr_proj_img of models/base_model.py:90-94, i.e. what projective_inverse_warp returns for source i
   * at scale s -- exactly 0 where the sample is not in view (:96).  Planar in both image layouts. (ABI v4) */
  float *warped[SFM_MAX_SCALES]; /* (B,n_src,3,h,w) or NULL; overwritten                      */
  int32_t projection;            /* SFM_PROJECTION_* (ABI v5); 0 = SFM_PROJECTION_FAST        */
} SfmLossDesc;

/* scratch needed by the three calls below for this descriptor (0 on a bad descriptor).  It depends on the shapes, on n_src and on
 * WHICH d_src[] are bound: query with the descriptor the calls will get. */
size_t sfm_loss_workspace_bytes(const SfmLossDesc *desc);

/* loss5 (device, 5 floats): total, pixel, smooth, exp, ssim -- the chainer.report keys
 * (:119-123) in that order.  With norm_B > B the values are this shard's additive share. */
int sfm_loss_fwd(const SfmLossDesc *desc, float *loss5, void *ws, size_t ws_bytes, void *stream);
/* gradients of total_loss scaled by the upstream gradient gy (loss.backward() => gy = 1) */
int sfm_loss_bwd(const SfmLossDesc *desc, float gy, void *ws, size_t ws_bytes, void *stream);
/* forward and backward in one launch: loss5 and the gradients for gy = 1 */
int sfm_loss_fwd_bwd(const SfmLossDesc *desc, float *loss5, void *ws, size_t ws_bytes, void *stream);

/* Measurement hook (the reference times the same loop with CUDA events, models/utils.py:16-30,
 * models/base_model.py:67): the NEXT sfm_loss_* call of the calling thread records the two
 * hipEvent_t handles on its stream immediately before and after its main kernel, then forgets
 * them.  NULL disables. */
int sfm_loss_profile_events(void *ev_start, void *ev_stop);
/* Host-side only (no device needed): the work decomposition the library chooses for `desc` and the entry point given by
 * (grad, loss) = (0,1) sfm_loss_fwd, (1,0) sfm_loss_bwd, (1,1) sfm_loss_fwd_bwd.  out[0] = wavefront items of the launch; then
 * per scale four ints: strips, row chunks per strip, rows of a chunk, items per sample (n_out >= 1 + 4 * n_scales).  For tests
 * and tuning; nothing is launched. */
int sfm_loss_plan_info(const SfmLossDesc *desc, int grad, int loss, int *out, int n_out);

/* Development / test hook, consumed by the NEXT sfm_loss_* call of the calling thread (whatever becomes of that call), then back
 * to 0.  3 = the kernels read the header of their argument block from the struct instead of taking it as preloaded scalar arguments
 * -- the path of a batch or a tile count beyond 16 bits, which no test could reach otherwise (results are bit-identical to 0).
 * 4 / 5 = one source per pass / two sources per pass wherever the latter exists (SSIM gradient launches in SFM_LAYOUT_HWC with an
 * even number of sources, FAST projection, no d_src), whatever the library would choose: the same arithmetic per pixel, the two
 * sources' shares of d_disp added in another order (in-process A/B timing and tests).
 * (Rounds 4-5 selected the projection with values 1 and 2 here: that is SfmLossDesc.projection since ABI v5.) */
int sfm_loss_variant(int variant);
/* Diagnostics: the NEXT sfm_loss_* call of this thread makes every wavefront of its main kernel
 * write {start, end (100 MHz realtime counter), HW_ID, XCC_ID} as 4 x uint64 per work item into
 * buf (device memory, 32 bytes * number of items; items <= workspace_bytes / 64). NULL disables. */
int sfm_loss_debug_trace(void *buf);

/* ------------------------------------------------------------------------------------------
 * F.resize_images(x, (oH,oW)) as used for the pyramid, models/base_model.py:70-72:
 * bilinear, align-corners.  x (N,C,H,W) -> y (N,C,oH,oW).
 * ---------------------------------------------------------------------------------------- */
int sfm_resize_fwd(const float *x, float *y, int N, int C, int H, int W, int oH, int oW, void *stream);

/* The image pyramid of one step in ONE launch: the loop head models/base_model.py:69-72 for one
 * tensor.  x (N,C,H,W) -> y[s] (N,C,H>>s,W>>s) for s = 1..n_scales-1, every scale resampled from
 * the full-resolution input (as the reference does); y[0] is ignored (scale 0 is x itself). */
int sfm_pyramid_fwd(const float *x, float *const *y, int N, int C, int H, int W, int n_scales, void *stream);
/* The same pyramid written pixel-interleaved for SFM_LAYOUT_HWC: x (N,3*G,H,W) planar, G images per sample
 * (1 for the target, n_src for the sources, base_model.py:50-57) -> y[s] (N,G,H>>s,W>>s,3) for s = 0..n_scales-1
 * (scale 0 is the re-laid-out input).  Values are identical to sfm_pyramid_fwd's. */
int sfm_pyramid_hwc_fwd(const float *x, float *const *y, int N, int G, int H, int W, int n_scales, void *stream);
/* Development / test hook: which kernel the NEXT sfm_pyramid_hwc_fwd / sfm_pyramid_pair_hwc_fwd call of the calling thread runs,
 * then back to automatic.  0 = automatic (the band kernel that reads every input pixel once, where the shape allows),
 * 1 = the one-thread-per-output-pixel kernel.  Same values bit for bit.  The hook is consumed by that next call whatever becomes
 * of it (empty batch, rejected argument); sfm_pyramid_fwd (planar) does not take it. */
int sfm_pyramid_variant(int variant);
/* Both pyramids of a step in ONE launch: tgt (N,3,H,W) and src (N,3*n_src,H,W) (base_model.py:50-57) ->
 * y_tgt[s] (N,1,h,w,3), y_src[s] (N,n_src,h,w,3), s = 0..n_scales-1: the whole loop head :69-72. */
int sfm_pyramid_pair_hwc_fwd(const float *tgt, const float *src, float *const *y_tgt, float *const *y_src, int N, int n_src,
                             int H, int W, int n_scales, void *stream);

/* One step of SFMLearner.__call__ from the FULL-RESOLUTION frames in one call (models/base_model.py:48-124; ABI v5): the loop head
 * :69-72 -- sfm_pyramid_pair_hwc_fwd of tgt_full (B,3,H,W) and src_full (B,3*n_src,H,W) into the buffers desc->tgt[s] /
 * desc->src[s], which the descriptor must bind as SFM_LAYOUT_HWC with H[s] = H[0] >> s, W[s] = W[0] >> s -- followed by
 * sfm_loss_fwd (sfm_step_fwd) or sfm_loss_fwd_bwd (sfm_step_fwd_bwd) on the same stream.  Exactly the two calls it replaces, same
 * results bit for bit; it exists for callers whose step is host-bound (the reference trains at B = 4, experiments/sfm_learner_v1.yml:43:
 * 25 us of GPU work per step): one trip through the FFI, one argument conversion, one plan look-up. */
int sfm_step_fwd(const float *tgt_full, const float *src_full, const SfmLossDesc *desc, float *loss5, void *ws, size_t ws_bytes,
                 void *stream);
int sfm_step_fwd_bwd(const float *tgt_full, const float *src_full, const SfmLossDesc *desc, float *loss5, void *ws, size_t ws_bytes,
                     void *stream);

/* ------------------------------------------------------------------------------------------
 * DispNet's output activation for all scales in one launch, models/disp_net.py:7-8 and
 * :104,:110,:116,:122:  disp = 10 * sigmoid(x) + 0.01.  numel[s] = elements of scale s.
 * Backward: g_x = g_disp * 10 * s * (1 - s), s recovered from disp (overwrites g_x).
 * ---------------------------------------------------------------------------------------- */
int sfm_disp_act_fwd(const float *const *x, float *const *disp, const long long *numel, int n_scales, void *stream);
int sfm_disp_act_bwd(const float *const *disp, const float *const *g_disp, float *const *g_x, const long long *numel,
                     int n_scales, void *stream);

/* ------------------------------------------------------------------------------------------
 * data_augmentation(), datasets/kitti/kitti_raw_transformed.py:23-74, image side: random scaling
 * (:32-45, F.resize_images to (scaled_h, scaled_w)), random crop back to (H,W) at (offset_y,
 * offset_x) (:48-59) and horizontal flip (:62-67) as ONE gather per output pixel.
 *   imgs (B,F,C,H,W): target + sources of each sample ; out the same shape ;
 *   params (B,5) float: scaled_h, scaled_w, offset_y, offset_x, flip (0/1), drawn on the host in the
 *   reference's order (the intrinsics update :41-44,:54-57,:66 is 4 scalars per sample, host side).
 * ---------------------------------------------------------------------------------------- */
int sfm_augment_fwd(const float *imgs, const float *params, float *out, int B, int F, int C, int H, int W, void *stream);

/* ------------------------------------------------------------------------------------------
 * DEVELOPMENT ONLY -- environment variables the library reads ONCE per process (at the first call that needs them; setting them
 * later has no effect).  They move work between wavefronts or pick another kernel for the same arithmetic; none changes a result
 * beyond the summation order of the per-wave partial sums.  Not part of the interface: names and meaning may change.
 *   fused loss (csrc/sfm_loss.hip, struct Tuning):
 *     SFM_CHUNK_ROWS=n            rows of a wave's chunk at every scale (4..28) instead of the planned heights
 *     SFM_CHUNK_ROWS_LIST=a,b,..  the same per scale
 *     SFM_NO_FILL                 no slot-filling refinement of the chunk heights (plan_chunks)
 *     SFM_NO_WIDE                 small L1 launches on the four-waves-per-SIMD build too
 *     SFM_PAIR=0|1                never / wherever possible two sources per pass (as sfm_loss_variant 4 / 5, for a whole process)
 *     SFM_PRIO_TABLE=abc,def      issue-priority levels of the dispatch rounds in the first / second half of the sources
 *     SFM_DEAL_ITEMS_BELOW=n      batches smaller than n (default 8) have their items, not whole samples, dealt over the XCDs
 *   image pyramid (csrc/sfm_ops.hip, struct PyramidTuning):
 *     SFM_PYRAMID_BAND_ROWS=n     input rows per band of the band kernel
 *     SFM_PYRAMID_THREADS=n       threads per workgroup of the band kernel
 * One-call hooks with the same purpose are entry points above: sfm_loss_variant, sfm_pyramid_variant, sfm_loss_profile_events,
 * sfm_loss_debug_trace.  Diagnostic BUILDS (never the product): -DSFM_STAMPS (`make stamps`), -DSFM_FIN_STAMPS.
 * ---------------------------------------------------------------------------------------- */

#ifdef __cplusplus
}
#endif
#endif /* SFMWARP_H_ */
