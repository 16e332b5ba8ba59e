/*
 * xrdslam_hip.h — C-ABI of the MI355X (gfx950) tracking/mapping engine.
 *
 * Drop-in boundary for the render/optimise hot path of openxrlab/xrdslam
 * (SURVEY.md §8b).  Plain C: raw device pointers, sizes, a hipStream_t passed
 * as void*; no torch types.  Every entry point returns XRD_OK (0) or an error
 * code and never calls exit() (the reference's CUDA_CHECK_ERRORS does,
 * third_party/sparse_voxels/include/cuda_utils.h:37-48).  Tensors are owned by
 * the caller; the engine borrows them for the duration of the launch.
 *
 * Each block cites the reference interface it replaces (paths relative to the
 * reference root).
 */
#ifndef XRDSLAM_HIP_H
#define XRDSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* xrd_stream_t; /* hipStream_t */

enum {
  XRD_OK = 0,
  XRD_ERR_ARG = 1,         /* bad argument (null pointer, bad size)          */
  XRD_ERR_LAUNCH = 2,      /* hip launch / runtime error                     */
  XRD_ERR_UNSUPPORTED = 3  /* configuration outside what the kernels cover   */
};

/* library/ABI version; bumps when a signature changes */
int xrd_abi_version(void);
/* last hip error string for XRD_ERR_LAUNCH (thread-unsafe, diagnostic only) */
const char* xrd_last_error(void);

/* ------------------------------------------------------------------------
 * NICE-SLAM fused render  (replaces, per iteration, the chain
 *   slam/models/conv_onet.py:377-524   ConvOnet.render_batch_ray
 *   slam/models/conv_onet.py:339-375   ConvOnet.eval_points
 *   slam/model_components/decoder_nice.py:386-414  NICE.forward (+ MLP,
 *       MLP_no_xyz, GaussianFourierFeatureTransform, F.grid_sample lookups)
 *   slam/model_components/utils.py:189-244  raw2outputs_nerf_color
 * and their autograd backward).
 * ---------------------------------------------------------------------- */
enum { XRD_STAGE_COARSE = 0, XRD_STAGE_MIDDLE = 1, XRD_STAGE_FINE = 2,
       XRD_STAGE_COLOR = 3 };
/* decoder kinds: MLP_no_xyz coarse, MLP(c_dim 32,out 1) middle,
 * MLP(c_dim 64,out 1) fine, MLP(c_dim 32,out 4) color */
enum { XRD_DEC_COARSE = 0, XRD_DEC_MIDDLE = 1, XRD_DEC_FINE = 2,
       XRD_DEC_COLOR = 3 };

typedef struct {
  /* bounding box AFTER ConvOnet.load_bound (conv_onet.py:324-337), float64 as
   * the reference holds it: x0,x1,y0,y1,z0,z1 */
  double bound[6];
  /* feature grids coarse,middle,fine,color; CHANNEL-LAST [Z][Y][X][32] f32,
   * i.e. the reference's [1,32,Z,Y,X] tensor (feature_grid_nice.py:4-12) in
   * torch.channels_last_3d memory format.  NULL when the stage does not use
   * the grid. */
  const float* grid[4];
  int32_t gdim[12]; /* Z,Y,X for each of the four grids */
  /* decoders in the packed MFMA-fragment layout produced by gathering the
   * flat state_dict-ordered parameter vector through xrd_nice_pack_index */
  const float* dec[4];
  /* optional per-cell byte masks (frustum feature selection,
   * slam/model_components/utils.py:298-375): the backward only accumulates
   * gradients into cells whose mask byte is non-zero.  NULL = every cell. */
  const uint8_t* gmask[4];
  int32_t n_samples;       /* rendering_n_samples  (conv_onet.py:46) */
  int32_t n_surface;       /* rendering_n_surface  (conv_onet.py:47) */
  const float* t_uniform;  /* [n_samples] torch.linspace(0,1,n) f32           */
  const double* t_surface; /* [n_surface] torch.linspace(0,1,n).double()      */
  double coarse_enlarge;   /* model_coarse_bound_enlarge (conv_onet.py:36)    */
} xrd_nice_scene;

/* number of floats of the flat (state_dict order) / packed parameter vector */
int xrd_nice_flat_len(int dec_kind);
int xrd_nice_pack_len(int dec_kind);
/* HOST function: idx[pack_len]; packed[i] = idx[i] < 0 ? 0 : flat[idx[i]] */
int xrd_nice_pack_index(int dec_kind, int32_t* idx_host);

/* Forward.  n_rays rays; S = n_samples + (gt_depth && stage!=coarse ?
 * n_surface : 0) samples per ray (S must be 32 or 48).
 *   rays_o, rays_d [n,3] f32; gt_depth [n] f32 or NULL; dmax: device scalar =
 *   max(gt_depth) (conv_onet.py:418,455), ignored when gt_depth is NULL.
 * Outputs depth,var [n] f64 (the reference returns float64 here), rgb [n,3]
 * f32; raw_out [n,S,4] (rgb_raw, occupancy logit after the out-of-bound
 * override) may be NULL — it is what the backward needs besides the inputs. */
int xrd_nice_render_fwd(const xrd_nice_scene* scene, int stage, int n_rays,
                        const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        double* depth, double* var, float* rgb, float* raw_out,
                        xrd_stream_t stream);

/* stage COARSE with a grid gradient: optional workspace of private replicas of
 * the coarse-grid gradient (the coarse grid has ~1e3 cells and every ray starts
 * in the camera's cell: blocks scatter into 32 replicas that are summed into
 * g_grid[0] afterwards).  The buffer must be ZERO on entry and is left zero.
 * Pass it as `workspace`; NULL = scatter straight into g_grid[0]. */
/* Point queries for the mesher (ConvOnet.query_fn / color_func,
 * slam/models/conv_onet.py:213-240 -> NICE.forward, stage 'fine' / 'color';
 * out-of-bound points get occupancy 100 like ConvOnet.eval_points :358-370):
 * points [n,3] f32 -> raw [n,4] = (rgb raw (stage colour) or 0, occupancy
 * logit = fine + middle).  stage: XRD_STAGE_FINE or XRD_STAGE_COLOR. */
int xrd_nice_eval_points(const xrd_nice_scene* scene, int stage,
                         int64_t n_points, const float* points, float* raw,
                         xrd_stream_t stream);
/* Point-SLAM geometry path — neighbour interpolation + geometry decoder
 * (MLP_geometry.get_feature_at_pos / forward,
 * slam/model_components/decoder_pointslam.py:162-273; half of the
 * xrd_point_render_* pair of SURVEY §8b, the colour half is not built yet):
 *   points [n,3] f32 sample positions, neighbors [n,8] i64 (ids into the
 *   cloud, -1 = none; xrd_knn_search), n_neighbors [n] i32 (neighbours inside
 *   the radius, as the search reports them), cloud [N,3], geo_feats [N,32],
 *   feat_mask [N] u8 or NULL (the frustum mask of get_geo_feats), radius [n]
 *   per-sample query radius or NULL -> radius_all, min_nn (samples with fewer
 *   neighbours take empty_feat [32]), packed_decoder = the decoder in the
 *   NICE 'middle' packing (xrd_nice_pack_index(XRD_STAGE_MIDDLE)) with
 *   embedder._B scaled by 2 pi.
 * fwd: occ [n] (occupancy logit), has [n] u8, relu_masks [n,4] u64 (for the
 *   backward; NULL = inference).
 * bwd: g_occ [n] -> g_points [n,3] (through the Fourier features AND the
 *   recomputed neighbour distances; NULL = not wanted), g_geo_feats [N,32]
 *   (ACCUMULATED with atomics, masked; NULL = not wanted). */
int xrd_point_geo_fwd(int64_t n_points, const float* points,
                      const int64_t* neighbors, const int32_t* n_neighbors,
                      const float* cloud, const float* geo_feats,
                      const uint8_t* feat_mask, const float* radius,
                      float radius_all, int min_nn, const float* empty_feat,
                      const float* packed_decoder, float* occ, uint8_t* has,
                      uint64_t* relu_masks, xrd_stream_t stream);
int xrd_point_geo_bwd(int64_t n_points, const float* points,
                      const int64_t* neighbors, const int32_t* n_neighbors,
                      const float* cloud, const float* geo_feats,
                      const uint8_t* feat_mask, const float* radius,
                      float radius_all, int min_nn, const float* empty_feat,
                      const float* packed_decoder, const uint64_t* relu_masks,
                      const float* g_occ, float* g_points, float* g_geo_feats,
                      xrd_stream_t stream);
/* Point-SLAM colour path (MLP_color with its defaults: per-neighbour F_theta
 * on [rel-pos embedding, colour feature], inverse-distance interpolation,
 * 5 x 128 softplus trunk with the feature added after every layer, skip after
 * the third, sigmoid; no view direction, no exposure code) replacing
 * slam/model_components/decoder_pointslam.py:276-291,408-542.
 *   flat parameter order (xrd_point_color_flat_len floats; the first
 *   xrd_point_color_grad_len are trainable and form the flat gradient):
 *     embedder_rel_pos._B [3,10], mlp_col_neighbor.linear1.weight [128,52],
 *     .bias, linear2.weight [32,128], .bias, fc_c.{0..4}.weight [128,32]/.bias,
 *     pts_linears.{0..4}.weight/.bias ([128,40], [128,128] x2, [128,168],
 *     [128,128]), output_linear.weight [3,128], .bias  (= the order of
 *     MLP_color.parameters()), then embedder._B [3,20].
 *   packed = flat gathered through xrd_point_color_pack_index (int32
 *   [xrd_point_color_pack_len], -1 = 0.0).
 * fwd: rgb [n,3]; save_c [n,32], save_h [5,n,128], save_y [n,8,32] (the
 *   interpolated feature, the trunk's layer outputs and F_theta's outputs, for
 *   the backward; all NULL = inference).
 * bwd: g_rgb [n,3] -> g_points [n,3] (Fourier features of p, relative-position
 *   features and the recomputed neighbour distances; NULL = not wanted),
 *   g_col_feats [N,32] (ACCUMULATED with atomics; NULL = not wanted), g_flat
 *   [xrd_point_color_grad_len] (overwritten; NULL = no parameter gradients).
 *   With g_flat: the weight gradients are contracted inside the backward's
 *   blocks (MFMA accumulators over the block's 128 points);
 *   ops = xrd_point_color_ops_floats(n) floats of scratch (the narrow operands
 *   read back as B fragments: 492 floats a point), workspace =
 *   xrd_point_color_ws_floats() floats (one partial per block, summed by a
 *   second launch). */
int xrd_point_color_flat_len(void);
int xrd_point_color_grad_len(void);
int xrd_point_color_pack_len(void);
int xrd_point_color_pack_index(int32_t* index);
int64_t xrd_point_color_ops_floats(int64_t n_points);
int64_t xrd_point_color_ws_floats(void);
int xrd_point_color_fwd(int64_t n_points, const float* points,
                        const int64_t* neighbors, const int32_t* n_neighbors,
                        const float* cloud, const float* col_feats,
                        const float* radius, float radius_all, int min_nn,
                        const float* empty_feat, const float* packed,
                        float* rgb, float* save_c, float* save_h,
                        float* save_y, xrd_stream_t stream);
int xrd_point_color_bwd(int64_t n_points, const float* points,
                        const int64_t* neighbors, const int32_t* n_neighbors,
                        const float* cloud, const float* col_feats,
                        const float* radius, float radius_all, int min_nn,
                        const float* packed, const float* rgb,
                        const float* save_c, const float* save_h,
                        const float* save_y, const float* g_rgb,
                        float* g_points, float* g_col_feats, float* g_flat,
                        float* ops, float* workspace, xrd_stream_t stream);
/* Point-SLAM mapping: compositing + mapping loss + their backward in one
 * launch (raw2outputs_nerf_color2, slam/model_components/utils.py:247-294;
 * get_loss_dict's mapping branch, slam/models/conv_onet_pointslam.py:190-204).
 *   raw [n_rays*S,4] = [rgb, occupancy logit] per sample, point_mask [n_rays*S]
 *   u8 (sample has neighbours; others composite with logit -100), z_vals
 *   [n_rays,S], target_d [n_rays], target_rgb [n_rays,3] or NULL (geometry
 *   stage: no colour term), ray_valid [n_rays] u8 or NULL.  A ray counts when
 *   target_d > 0, at least min_valid_points samples have neighbours, its depth
 *   is not NaN and ray_valid keeps it.
 *   -> loss [2] = (sum |target_d - depth|, w_color sum |target_rgb - colour|),
 *   g_raw [n_rays*S,4] = d (loss[0] + loss[1]) / d raw.  S <= 16. */
int xrd_point_map_loss(int n_rays, int n_samples, const float* raw,
                       const uint8_t* point_mask, const float* z_vals,
                       const float* target_d, const float* target_rgb,
                       const uint8_t* ray_valid, float sigmoid_coef,
                       float w_color, int min_valid_points, float* loss,
                       float* g_raw, xrd_stream_t stream);
/* Point-SLAM batch selection + sample placement, one launch (single block):
 * the filter of PointSLAM.get_model_input (slam/algorithms/point_slam.py:
 * 246-300: valid = d > 0, keep = valid & d <= min(10 median(d[valid]),
 * 1.2 max(d[valid])), torch.median = LOWER median) — as a mask, the batch
 * keeps its shape — and render_batch_ray's samples
 * (slam/models/conv_onet_pointslam.py:330-347): z = near d (1 - t) + far d t,
 * t = linspace(0, 1, S), pts = o + dir z.  radius_stack [F, image_pixels]
 * (NULL: no radii): the rays' dynamic query radii, looked up at pixel
 * (hedge + idx / crop_width, wedge + idx % crop_width) of frame
 * ray / rays_per_frame, per ray and repeated per sample.  stats [2] (NULL or)
 * = (median, max). */
int xrd_point_batch(int n_rays, int n_samples, const float* rays_o,
                    const float* rays_d, const float* target_d,
                    const float* radius_stack, const int64_t* pixel_idx,
                    int rays_per_frame, int crop_width, int hedge, int wedge,
                    int image_width, int64_t image_pixels, float near_coef,
                    float far_coef, uint8_t* keep, float* radius,
                    float* z_vals, float* pts, float* radius_pts, float* stats,
                    xrd_stream_t stream);
/* Point-SLAM tracking loss, one launch (single block): the tracking branch of
 * ConvOnet2.get_loss_dict (slam/models/conv_onet_pointslam.py:144-189) on a
 * batch that kept its shape (ray_valid [n] = the batch filter, or NULL):
 * uncertainty-normalised depth error with the batch-median outlier rejection
 * (LOWER median, NaN-propagating like torch.median) and the masked L1 colour
 * term.  -> loss [2] = (geo, w_color * rgb), g_depth [n], g_color [n,3] =
 * gradients of loss[0] + loss[1] (the variance is detached). */
int xrd_point_track_loss(int n_rays, int handle_dynamic, int use_color,
                         float w_color, const float* depth, const float* var,
                         const float* color, const float* target_d,
                         const float* target_rgb, const uint8_t* ray_valid,
                         float* loss, float* g_depth, float* g_color,
                         xrd_stream_t stream);
/* Point-SLAM compositing alone (raw2outputs_nerf_color2, utils.py:247-294, with
 * the no-neighbour override of render_batch_ray, conv_onet_pointslam.py:441):
 *   rgb [n_rays*S rows, stride rgb_stride >= 3] or NULL, occ [n_rays*S rows,
 *   stride occ_stride] (interleaved [rgb, occ] rows: rgb = raw, stride 4,
 *   occ = raw + 3, stride 4), point_mask [n_rays*S] u8 or NULL, z_vals
 *   [n_rays,S] -> depth [n_rays], var [n_rays], color [n_rays,3].
 * bwd: g_depth / g_var [n_rays], g_color [n_rays,3] (each may be NULL = 0) ->
 *   g_rgb (strided like rgb, may be NULL) and g_occ (strided).  S <= 16. */
int xrd_point_composite_fwd(int n_rays, int n_samples, const float* rgb,
                            int rgb_stride, const float* occ, int occ_stride,
                            const uint8_t* point_mask, const float* z_vals,
                            float sigmoid_coef, float* depth, float* var,
                            float* color, xrd_stream_t stream);
int xrd_point_composite_bwd(int n_rays, int n_samples, const float* rgb,
                            int rgb_stride, const float* occ, int occ_stride,
                            const uint8_t* point_mask, const float* z_vals,
                            float sigmoid_coef, const float* g_depth,
                            const float* g_var, const float* g_color,
                            float* g_rgb, int g_rgb_stride, float* g_occ,
                            int g_occ_stride, xrd_stream_t stream);
/* Point-SLAM render as one call each way (render_batch_ray given the sample
 * points [n_rays*S,3] and their neighbours, conv_onet_pointslam.py:302-461) =
 * xrd_point_geo_* -> xrd_point_color_* (col_feats != NULL: colour stage) ->
 * xrd_point_composite_*.  The per-point buffers (occ [m], has [m] u8,
 * relu_masks [m,4] u64, rgb [m,3], save_c / save_h / save_y as in
 * xrd_point_color_fwd; m = n_rays*S) are the caller's and carry the forward to
 * the backward.  bwd: g_depth / g_var [n_rays], g_color [n_rays,3] (NULL = 0)
 * -> g_points [m,3] (NULL = not wanted), g_geo_feats / g_col_feats
 * (ACCUMULATED, NULL = not wanted), g_flat / ops / workspace as in
 * xrd_point_color_bwd; scratch = xrd_point_render_scratch_floats(m) floats. */
int64_t xrd_point_render_scratch_floats(int64_t n_points);
int xrd_point_render_fwd(
    int n_rays, int n_samples, const float* points, const int64_t* neighbors,
    const int32_t* n_neighbors, const float* cloud, const float* geo_feats,
    const uint8_t* feat_mask, const float* col_feats, const float* radius,
    float radius_all, int min_nn, const float* empty_geo,
    const float* empty_col, const float* packed_geo, const float* packed_col,
    const float* z_vals, float sigmoid_coef, float* occ, uint8_t* has,
    uint64_t* relu_masks, float* rgb, float* save_c, float* save_h,
    float* save_y, float* depth, float* var, float* color,
    xrd_stream_t stream);
int xrd_point_render_bwd(
    int n_rays, int n_samples, const float* points, const int64_t* neighbors,
    const int32_t* n_neighbors, const float* cloud, const float* geo_feats,
    const uint8_t* feat_mask, const float* col_feats, const float* radius,
    float radius_all, int min_nn, const float* empty_geo,
    const float* packed_geo, const float* packed_col, const float* z_vals,
    float sigmoid_coef, const float* occ, const uint8_t* has,
    const uint64_t* relu_masks, const float* rgb, const float* save_c,
    const float* save_h, const float* save_y, const float* g_depth,
    const float* g_var, const float* g_color, float* scratch, float* g_points,
    float* g_geo_feats, float* g_col_feats, float* g_flat, float* ops,
    float* workspace, xrd_stream_t stream);
int64_t xrd_nice_coarse_ws_floats(const xrd_nice_scene* scene);
/* Backward of the above.  g_depth,g_var [n] f64, g_rgb [n,3] f32 (any may be
 * NULL = zero).  Requested gradients (each may be NULL = not needed):
 *   g_rays_o, g_rays_d [n,3] f32 (overwritten; per-tile f64 partial sums in
 *     `ws`, added in a fixed order);
 *   g_grid[4] channel-last like the grids (ACCUMULATED with atomics: caller
 *     zeroes);
 *   g_dec[4]  flat state_dict-ordered decoder gradients (overwritten).  The
 *     weight gradients are contracted inside the render backward (MFMA
 *     accumulators kept across tiles, operands exchanged through LDS) and
 *     added to 8 replicas in `ws` that a finishing launch sums.  Only
 *     XRD_DEC_COLOR is supported in this version (mapping_fix_fine=True
 *     default, conv_onet.py:62,190-195); others -> XRD_ERR_UNSUPPORTED.
 *   ws: float workspace of xrd_nice_bwd_ws_floats(n_rays) floats, 16-byte
 *     aligned, contents arbitrary on entry (needed when ray or decoder
 *     gradients are requested; the coarse stage takes its replica buffer
 *     here instead, see above).
 * Launches: the middle / fine / colour stages are ONE kernel over every decoder
 * of the stage (+ one finishing launch when ray or decoder gradients are
 * wanted, + one fill of the replicas for decoder gradients). */
int64_t xrd_nice_bwd_ws_floats(int n_rays);
int xrd_nice_render_bwd(const xrd_nice_scene* scene, int stage, int n_rays,
                        const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        const float* raw, const double* g_depth,
                        const double* g_var, const float* g_rgb,
                        float* g_rays_o, float* g_rays_d,
                        float* const g_grid[4], float* const g_dec[4],
                        float* ws, xrd_stream_t stream);
/* Tracking (colour stage, 48 samples a ray, <= 340 rays, ray gradients only):
 * the forward keeps the ReLU masks of the three decoders — masks:
 * xrd_nice_fwd_masks_words(n_rays) 64-bit words: [(ray*3 + tile)*3 + decoder]
 * [64 lanes], then n_rays * 32 words the forward's launches hand over in —
 * and the backward that receives them skips its forward recompute.  Both run
 * one decoder per block (three blocks a group of tiles, csrc/nice_render.hip)
 * and a finishing launch; raw_out is required (the decoder blocks meet in it).
 * Same results as
 * xrd_nice_render_fwd / xrd_nice_render_bwd (the forward bit for bit, the ray
 * gradients up to the summation order of the three decoders' parts).  Other
 * shapes -> XRD_ERR_UNSUPPORTED (use the pair above).  Replaces the same
 * reference lines: the renderer (conv_onet.py:339-524) under autograd in
 * Algorithm.optimize_update's tracking loop (base_algorithm.py:243-275). */
int64_t xrd_nice_fwd_masks_words(int n_rays);
int xrd_nice_render_fwd_masks(const xrd_nice_scene* scene, int stage,
                              int n_rays, const float* rays_o,
                              const float* rays_d, const float* gt_depth,
                              const float* dmax, double* depth, double* var,
                              float* rgb, float* raw_out, uint64_t* masks,
                              xrd_stream_t stream);
int xrd_nice_render_bwd_masks(const xrd_nice_scene* scene, int stage,
                              int n_rays, const float* rays_o,
                              const float* rays_d, const float* gt_depth,
                              const float* dmax, const float* raw,
                              const double* g_depth, const double* g_var,
                              const float* g_rgb, const uint64_t* masks,
                              float* g_rays_o, float* g_rays_d, float* ws,
                              xrd_stream_t stream);


/* One NICE-SLAM MAPPING iteration of a stage as one launch (+ one finishing
 * launch): forward render, the mapping loss and the backward of everything it
 * reaches — replaces, for Algorithm.optimize_update's mapping iterations
 * (slam/algorithms/base_algorithm.py:239-275), the chain Model.get_outputs ->
 * get_loss_dict -> loss.backward of slam/models/conv_onet.py:339-524 (render),
 * :178-184 (the mapping losses: sum |gt_d - depth| over the rays with a valid
 * sensor depth, + w_color * sum |gt_rgb - rgb| in the colour stage — plain
 * sums over rays, so the gradient of a ray is known once it is composited)
 * and slam/model_components/utils.py:189-244.  Inputs as xrd_nice_render_fwd
 * (gt_depth [n] REQUIRED: the loss needs it also in the coarse stage, which
 * does not sample with it; dmax [1] device, unused in the coarse stage),
 * tgt_rgb [n,3] (colour stage), keep [n] uint8 or NULL: rays with keep == 0
 * are rendered but contribute neither loss nor gradient (the reference drops
 * them from the batch, slam/algorithms/nice_slam.py:181-194).  Outputs (each
 * may be NULL): g_rays_o / g_rays_d [n,3] (overwritten; not in the coarse
 * stage), g_grid[4] (ACCUMULATED, cells masked by scene->gmask),
 * g_dec_color (flat colour-decoder gradient, overwritten; colour stage),
 * loss [1] f64 = the summed loss terms.  ws: xrd_nice_map_ws_floats(scene,
 * stage, n) floats, 16-byte aligned, ZERO before its first use (the coarse
 * stage's gradient replicas are handed back zeroed; the colour stage's partial
 * rows of the decoder gradient are rewritten in full by every call).
 * 48 samples a ray (32 + 16) only: other sampling configs ->
 * XRD_ERR_UNSUPPORTED (use xrd_nice_render_fwd / xrd_nice_loss /
 * xrd_nice_render_bwd). */
/* xrd_nice_map_iter that also hands out, per sample (48 a ray, ray-major),
 * the sample point [n*48,3] and d loss / d occupancy logit [n*48] (0 for
 * samples outside the bound and rays with keep == 0) — what a caller needs to
 * form the weight gradient of the fine / middle decoder outside the launch
 * (mapping_fix_fine = False, slam/models/conv_onet.py:62,190-195: the fused
 * launch contracts the colour decoder's weight gradient only).  Both NULL =
 * xrd_nice_map_iter; coarse stage -> XRD_ERR_UNSUPPORTED. */
int xrd_nice_map_iter_export(const xrd_nice_scene* scene, int stage,
                             int n_rays, const float* rays_o,
                             const float* rays_d, const float* gt_depth,
                             const float* dmax, const float* tgt_rgb,
                             const uint8_t* keep, float w_color,
                             float* g_rays_o, float* g_rays_d,
                             float* const g_grid[4], float* g_dec_color,
                             float* sample_points, float* g_occ, float* ws,
                             double* loss, xrd_stream_t stream);
/* One NICE-SLAM TRACKING iteration (colour stage, 48 samples a ray) as one
 * launch + one finishing launch: forward render, the robust tracking loss of
 * slam/models/conv_onet.py:145-176 (residual |d - depth| / sqrt(var), rays
 * below 10 x the batch's lower median when handle_dynamic, + w_color x L1
 * colour on the same rays when use_color; the variance is detached) and the
 * backward to the rays.  The batch median sits between forward and backward:
 * the blocks meet at one grid barrier, so every block must be resident —
 * n_rays <= 1024 (4 rays a block, one block a CU), else XRD_ERR_UNSUPPORTED
 * (use xrd_nice_render_fwd / xrd_nice_loss / xrd_nice_render_bwd).  Same
 * arithmetic as that chain (tests/test_nice_hip.py).  ws:
 * xrd_nice_track_ws_floats(n_rays) floats, 16-byte aligned, ZERO before its
 * first use and handed back usable (the barrier counter is re-zeroed).
 * -> g_rays_o / g_rays_d [n,3] (overwritten), loss [1] f64. */
int64_t xrd_nice_track_ws_floats(int n_rays);
int xrd_nice_track_iter(const xrd_nice_scene* scene, int n_rays,
                        const float* rays_o, const float* rays_d,
                        const float* gt_depth, const float* dmax,
                        const float* tgt_rgb, const uint8_t* keep,
                        int use_color, int handle_dynamic, float w_color,
                        float* g_rays_o, float* g_rays_d, float* ws,
                        double* loss, xrd_stream_t stream);
int64_t xrd_nice_map_ws_floats(const xrd_nice_scene* scene, int stage,
                               int n_rays);
int xrd_nice_map_iter(const xrd_nice_scene* scene, int stage, int n_rays,
                      const float* rays_o, const float* rays_d,
                      const float* gt_depth, const float* dmax,
                      const float* tgt_rgb, const uint8_t* keep,
                      float w_color, float* g_rays_o, float* g_rays_d,
                      float* const g_grid[4], float* g_dec_color, float* ws,
                      double* loss, xrd_stream_t stream);
/* kernel attributes (dynamic LDS) of the above; once per process, before
 * capturing launches into a hipGraph */
int xrd_nice_map_warmup(void);

/* NICE-SLAM frustum feature selection on the device — replaces, per mapping
 * call, ConvOnet.pre_precessing -> get_mask_from_c2w
 * (slam/models/conv_onet.py:94-130, slam/model_components/utils.py:298-375:
 * numpy + cv2.remap on the host) for n_grids <= 4 grids in two launches.
 * Per grid g: dims_zyx[3g..] = (Z, Y, X); axes[3g + a] = the lattice
 * coordinates along x, y, z (float32, lengths X, Y, Z: the reference's
 * torch.linspace of the bound); sampled[g] float [ZYX] scratch; outputs:
 * mask[g] uint8 [Z][Y][X] (1 = the cell is optimised), cells[g] int32 (room
 * for ZYX entries; the selected cells in ARBITRARY order), count[g] int32 [1].
 * c2w: device float[16] row-major camera-to-world; depth: device float
 * [H, W]; ws: device int32[4] scratch. */
int xrd_nice_frustum_cells(int n_grids, const int32_t* dims_zyx,
                           const float* const* axes, const float* c2w,
                           const float* depth, int H, int W, float fx,
                           float fy, float cx, float cy,
                           float* const* sampled, uint8_t* const* mask,
                           int32_t* const* cells, int32_t* const* count,
                           int32_t* ws, xrd_stream_t stream);

/* Fused Adam over a subset of 32-float cells of a channel-last grid
 * (frustum feature selection: conv_onet.py:94-130,187-211 optimise
 * val[mask] as a 1-D Parameter and write it back every iteration; here the
 * masked cells are updated in place).  cell_idx [n_cells] int32 lists the
 * selected cells (NULL = all n_cells cells); m, v are the Adam moments,
 * COMPACT [n_cells][cell_floats] (row i belongs to cell_idx[i]).  Matches torch.optim.Adam (no amsgrad, no weight decay):
 * step is the 1-based step count.  zero_grad != 0 clears g after use. */
int xrd_adam_cells(float* param, float* g, float* m, float* v,
                   const int32_t* cell_idx, int64_t n_cells, int cell_floats,
                   float lr, float beta1, float beta2, float eps, int step,
                   int zero_grad, xrd_stream_t stream);

/* same, with the 1-based step count read from device memory at execution time
 * (hipGraph replay: a preceding node increments *step_dev) */
int xrd_adam_cells_devstep(float* param, float* g, float* m, float* v,
                           const int32_t* cell_idx, int64_t n_cells,
                           int cell_floats, float lr, float beta1, float beta2,
                           float eps, const int32_t* step_dev, int zero_grad,
                           xrd_stream_t stream);

/* same, for a launch that is replayed across mapping calls: cell_idx has room
 * for `capacity` entries (m, v: capacity x cell_floats), the number of valid
 * entries is read from *n_cells_dev at execution time */
int xrd_adam_cells_devcount(float* param, float* g, float* m, float* v,
                            const int32_t* cell_idx, int64_t capacity,
                            int cell_floats, float lr, float beta1,
                            float beta2, float eps, const int32_t* step_dev,
                            const int32_t* n_cells_dev, int zero_grad,
                            xrd_stream_t stream);

/* xrd_adam_cells with a self-advancing device step counter: step_ticket =
 * {steps taken so far, ticket (0 between launches)} int32[2]; this launch is
 * step step_ticket[0] + 1 and its last block to finish stores that number —
 * replaces the separate increment launch in front of xrd_adam_cells_devstep /
 * _devcount (the reference: torch.optim.Adam's per-step counter,
 * slam/engine/optimizers.py:63-171).  n_cells_dev NULL: n_cells is the exact
 * count; else the capacity, with the valid count read on the device. */
int xrd_adam_cells_tick(float* param, float* g, float* m, float* v,
                        const int32_t* cell_idx, int64_t n_cells,
                        int cell_floats, float lr, float beta1, float beta2,
                        float eps, int32_t* step_ticket,
                        const int32_t* n_cells_dev, int zero_grad,
                        xrd_stream_t stream);

/* xrd_adam_cells_tick for up to XRD_ADAM_MAX_SETS grids in ONE launch (the
 * feature grids a mapping stage optimises share betas / eps and differ in
 * learning rate, selection and step count: slam/engine/optimizers.py:63-171
 * steps them one torch.optim.Adam after the other).  Same arithmetic per grid
 * as xrd_adam_cells_tick; sets with n_cells == 0 are skipped. */
#define XRD_ADAM_MAX_SETS 4
typedef struct {
  float* param;
  float* grad;
  float* m;
  float* v;
  const int32_t* cell_idx;
  int64_t n_cells;
  float lr;
  int32_t* step_ticket;
  const int32_t* n_cells_dev;
} xrd_adam_cells_set;
int xrd_adam_cells_multi(int n_sets, const xrd_adam_cells_set* sets,
                         int cell_floats, float beta1, float beta2, float eps,
                         int zero_grad, xrd_stream_t stream);

/* one-time set-up of kernel attributes (dynamic LDS sizes); call once per
 * process before capturing launches into a hipGraph */
int xrd_nice_warmup(void);

/* ------------------------------------------------------------------------
 * Co-SLAM encodings — replace the tiny-cuda-nn modules the reference
 * instantiates in slam/model_components/encodings_coslam.py:43-53 (HashGrid,
 * 16 levels x 2 features) and :68-75 (OneBlob, 16 bins); used by
 * slam/models/joint_encoding.py:212-234,439-481.  fp32 parameters
 * (dtype=torch.float).  tiny-cuda-nn is an unvendored dependency of the
 * reference: arithmetic per SURVEY.md Appendix C.1 / C.2.
 * ---------------------------------------------------------------------- */
/* HOST: per-level table of a multi-resolution grid.  scale_l =
 * exp2f(l*log2f(per_level_scale))*base_resolution - 1; res_l = ceilf(scale)+1;
 * size_l = min(align8(res_l^3), 2^log2_hashmap_size) (dense != 0: no cap);
 * offsets are in entries (each entry = 2 floats). */
int xrd_hashgrid_levels(int n_levels, int base_resolution,
                        float per_level_scale, int log2_hashmap_size,
                        int dense, float* scales, uint32_t* res,
                        uint32_t* sizes, uint32_t* offsets,
                        uint32_t* total_entries);
/* y[n,2*L] = encode(x[n,3] in [0,1]); level table arrays are HOST pointers */
int xrd_hashgrid_fwd(int n_levels, const float* scales, const uint32_t* res,
                     const uint32_t* sizes, const uint32_t* offsets,
                     int64_t n_points, const float* x, const float* params,
                     float* y, xrd_stream_t stream);
/* dparams (ACCUMULATED: LDS-privatised per 8192-entry chunk, then coalesced
 * atomic adds of the non-zero entries; may be NULL), dx[n,3] (overwritten; may be
 * NULL) from dy[n,2*L] */
int xrd_hashgrid_bwd(int n_levels, const float* scales, const uint32_t* res,
                     const uint32_t* sizes, const uint32_t* offsets,
                     int64_t n_points, const float* x, const float* params,
                     const float* dy, float* dparams, float* dx,
                     xrd_stream_t stream);
/* Co-SLAM's smoothness term (slam/models/joint_encoding.py:165-197) on the
 * hash grid: total variation of the 2*L features on a random lattice of
 * side^3 points voxel_size apart inside bound6 (xmin,xmax,ymin,ymax,zmin,zmax;
 * HOST doubles).  rand_offset / rand_shift: the reference's two torch.rand
 * draws (DEVICE doubles [3] each: lattice origin inside the volume, sub-voxel
 * shift).  points [side^3,3] (normalised to the unit cube, f64 arithmetic like
 * the reference's bbox), feat / dfeat [side^3, 2L]: the features and
 * d loss / d features, loss [1] f64 = scale * sum of squared neighbour
 * differences along the three axes / (side + 1)^3.  Three launches. */
int xrd_hashgrid_tv(int n_levels, const float* scales, const uint32_t* res,
                    const uint32_t* sizes, const uint32_t* offsets,
                    const float* params, int side, const double* bound6,
                    double voxel_size, double margin,
                    const double* rand_offset, const double* rand_shift,
                    float scale, float* points, float* feat, float* dfeat,
                    double* loss, xrd_stream_t stream);
/* OneBlob: y[n,dims*n_bins] (dimension-major), quartic kernel, periodic */
int xrd_oneblob_fwd(int64_t n_points, int dims, int n_bins, const float* x,
                    float* y, xrd_stream_t stream);
int xrd_oneblob_bwd(int64_t n_points, int dims, int n_bins, const float* x,
                    const float* dy, float* dx, xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * Vox-Fusion sparse voxel octree (HOST) — replaces the TorchScript class
 * torch.classes.svo.Octree (third_party/sparse_octree/src/bindings.cpp:8-31,
 * src/octree.cpp).  Node ids are the creation order of a PROCESS-GLOBAL
 * counter like the reference's static Octant::next_index_ (octree.cpp:9);
 * xrd_octree_reset_id_counter() is what the reference's unpickle constructor
 * does (octree.cpp:22).  Not thread-safe (the reference neither: callers hold
 * SparseVoxel.map_lock).  voxels are int32 [n,3] HOST arrays.
 * ---------------------------------------------------------------------- */
void* xrd_octree_create(int grid_dim, int feat_dim, double voxel_size);
void xrd_octree_destroy(void* tree);
void xrd_octree_reset_id_counter(void);
/* insert(Tensor[N,3] int32): *created_any = 1 when a node was created */
int xrd_octree_insert(void* tree, const int32_t* voxels, int64_t n,
                      int* created_any);
/* try_insert: fraction of the batch's corner keys already in the tree */
double xrd_octree_try_insert(void* tree, const int32_t* voxels, int64_t n);
int xrd_octree_has_voxel(void* tree, const int32_t* xyz);
int64_t xrd_octree_count_nodes(void* tree);
int64_t xrd_octree_count_leaf_nodes(void* tree);
/* get_centres_and_children: voxels f32[T,4], children f32[T,8] (ids, -1),
 * features i32[T,8] (corner-leaf ids of SURFACE leaves), T = count_nodes */
int xrd_octree_export(void* tree, float* voxels, float* children,
                      int32_t* features);
/* get_voxels / get_leaf_voxels: return the row count; fill up to cap_rows */
int64_t xrd_octree_get_voxels(void* tree, float* out_xyzs, int64_t cap_rows);
int64_t xrd_octree_get_leaf_voxels(void* tree, float* out_xyz,
                                   int64_t cap_rows);

/* ------------------------------------------------------------------------
 * Vox-Fusion ray/octree kernels — replace the two live functions of the
 * pybind module `grid` (third_party/sparse_voxels/src/binding.cpp:10-21):
 *   svo_intersect        (src/intersect.cpp:83-112, intersect_gpu.cu:191-270)
 *   inverse_cdf_sampling (src/sample.cpp:56-95,     sample_gpu.cu:133-239)
 * Same shapes, hit order, sentinels and quirks; voxel ids bit-exact.
 * ---------------------------------------------------------------------- */
/* ray_start, ray_dir [B,M,3] f32; points [B,N,3] f32 node centres, children
 * [B,N,9] i32 (8 child ids, side) — or ONE tree [N,..] shared by all batches
 * when tree_shared != 0 (the reference needs B replicas); out idx [B,M,n_max]
 * i32 (-1 padded, DFS hit order), min_depth/max_depth [B,M,n_max] f32 (only
 * hit slots are written: pre-zero like intersect.cpp:98-106).  overflow_flag
 * (optional device int) is set if a ray's DFS stack exceeded 128 entries
 * (the reference asserts < 256; 256^3 trees need 57). */
int xrd_svo_intersect(int b, int n_nodes, int m_rays, float voxelsize,
                      int n_max, int tree_shared, const float* ray_start,
                      const float* ray_dir, const float* points,
                      const int32_t* children, int32_t* idx, float* min_depth,
                      float* max_depth, int32_t* overflow_flag,
                      xrd_stream_t stream);
/* pts_idx [G,R,P] i32, min/max_depth, probs [G,R,P] f32, steps [G,R] f32,
 * uniform_noise [G,R,S] f32 -> sampled_idx i32, sampled_depth, sampled_dists
 * f32 [G,R,S]; outputs must be pre-filled (-1, 0, 0) like sample.cpp:77-86. */
int xrd_inverse_cdf_sampling(int b, int num_rays, int max_hits, int max_steps,
                             float fixed_step_size, const int32_t* pts_idx,
                             const float* min_depth, const float* max_depth,
                             const float* uniform_noise, const float* probs,
                             const float* steps, int32_t* sampled_idx,
                             float* sampled_depth, float* sampled_dists,
                             xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * Vox-Fusion fused "voxel features + decoder" — replaces, per sample point,
 * get_features (F.embedding x2 + trilinear_interp,
 * slam/model_components/voxel_helpers_voxfusion.py:97-123) and
 * Decoder.get_values (slam/model_components/decoder_voxfusion.py:123-149; the
 * model's defaults sparse_voxel.py:59-62: in_dim 16, width 128, depth 2, no
 * positional encoding) and their autograd backward.
 *   xyz [P,3] f32 sample positions, voxel_idx [P] i32 leaf-voxel ids (>= 0),
 *   centres [Nv,3] f32, vertex_idx [Nv,8] i32, embeddings [E,16] f32
 *   (map_states of sparse_voxel.py:342-357), packed = the decoder's
 *   state_dict, flattened in key order and gathered through
 *   xrd_vox_pack_index (xrd_vox_pack_len floats).
 * fwd: sdf [P], rgb [P,3] (after the sigmoid); optional saves for the
 *   backward and the weight-gradient GEMMs: save_x [P,16], save_h1/h2/f/hc
 *   [P,128] (post-ReLU layer outputs, f = sdf feature), masks [P,3,4] u32.
 * bwd: g_sdf [P], g_rgb [P,3] (NULL = zero) -> g_xyz [P,3] (NULL = not
 *   wanted), g_embeddings [E,16] (ACCUMULATED with atomics; NULL = not
 *   wanted) and the per-point gradient operands g_c3 [P,4] (d/d rgb logits,
 *   slot 3 = g_sdf), g_hc/g_f/g_h2/g_h1 [P,128] (pre-activation gradients;
 *   NULL = not wanted).  The five weight gradients are then plain GEMMs over
 *   the points: dW = G^T A (engine/vox.py runs them in rocBLAS).
 * n_points_dev (NULL = unused): static-capacity launches (captured graphs) —
 *   n_points is the CAPACITY of the arrays and the live count is read from
 *   this device int (clamped to [0, n_points]); rows beyond it are neither
 *   read nor written.
 * ---------------------------------------------------------------------- */
int xrd_vox_flat_len(void);
int xrd_vox_pack_len(void);
int xrd_vox_pack_index(int32_t* packed_from_flat);
int xrd_vox_points_fwd(int64_t n_points, const float* xyz,
                       const int32_t* voxel_idx, const float* centres,
                       const int32_t* vertex_idx, const float* embeddings,
                       float voxel_size, const float* packed, float* sdf,
                       float* rgb, float* save_x, float* save_h1,
                       float* save_h2, float* save_f, float* save_hc,
                       uint32_t* masks, const int32_t* n_points_dev,
                       xrd_stream_t stream);
int xrd_vox_points_bwd(int64_t n_points, const float* xyz,
                       const int32_t* voxel_idx, const float* centres,
                       const int32_t* vertex_idx, const float* embeddings,
                       float voxel_size, const float* packed, const float* rgb,
                       const uint32_t* masks, const float* g_sdf,
                       const float* g_rgb, float* g_xyz, float* g_embeddings,
                       float* g_c3, float* g_hc, float* g_f, float* g_h2,
                       float* g_h1, const int32_t* n_points_dev,
                       xrd_stream_t stream);

/* The decoder's weight gradients from the operands above (autograd of
 * decoder_voxfusion.py:123-149): dW = G^T A contracted over the points on
 * MFMA, bias gradients = column sums.  g_flat [xrd_vox_flat_len()] receives
 * the ten tensors in state_dict order (pts_linears.0.weight [128,16], .bias,
 * pts_linears.1.weight [128,128], .bias, sdf_out.weight [129,128], .bias,
 * color_out.0.weight [128,144], .bias, color_out.2.weight [3,128], .bias);
 * workspace: xrd_vox_dw_ws_floats() floats.  n_points_dev as above. */
int64_t xrd_vox_dw_ws_floats(void);
int xrd_vox_dw(int64_t n_points, const int32_t* n_points_dev,
               const float* save_x, const float* save_h1, const float* save_h2,
               const float* save_f, const float* save_hc, const float* g_c3,
               const float* g_hc, const float* g_f, const float* g_h2,
               const float* g_h1, float* workspace, float* g_flat,
               xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * Vox-Fusion ray pipeline with STATIC capacities (csrc/vox_rays.hip) — what
 * SparseVoxel.render_rays / get_loss_dict do around the decoder
 * (slam/models/sparse_voxel.py:103-143,160-304) and ray_intersect /
 * ray_sample do around the two `grid` kernels
 * (slam/model_components/voxel_helpers_voxfusion.py:399-481,647-714), without
 * a host sync: the data-dependent sizes the reference trims its tensors to
 * (largest hit count, hit rays, sampler row length, longest sample row, valid
 * samples, the loss's sample counts) are kept in meta [xrd_vox_meta_len()]
 * i32 on the device.  meta[5] = overflow bits (1: a sample row needed more
 * than s_cap slots, 2: more than p_cap points, 4: a sample row with a hole),
 * meta[10] = traversal stack overflow; a caller checks them once per frame.
 * meta must be ZERO before its first use and stay with its pipeline: beyond
 * the record it holds the blocks' partial counts.
 *
 * xrd_vox_sample_rays is THREE launches (intersection + hit sort; hit-ray
 * ranks + sampling; point offsets + compaction): the batch-wide prefixes are
 * sums over per-block partial counts taken by the next launch's blocks.
 *
 * xrd_vox_sample_rays: rays [n_rays,3] (+ target_d [n_rays]) -> hits (octree
 *   traversal when centres != NULL, n_max <= 64 hits a ray; else hit_idx /
 *   hit_min / hit_max [n_rays,n_max] are inputs), sorted by entry depth and cut
 *   at max_distance IN PLACE; probs [n_rays,n_max], steps [n_rays], hit
 *   [n_rays] (0/1), rank / hit_rays [n_rays] (hit-ray compaction both ways);
 *   samples s_idx / s_depth [n_rays,s_cap] (-1 / 10.0 padded), cnt [n_rays],
 *   offs [n_rays+1] (exclusive scan); points xyz [p_cap,3], vox [p_cap] in
 *   ray-major order.  noise [n_rays,s_cap] uniform draws (NULL = 0.5, the
 *   deterministic mode), clamped to [0.001, 0.999] like the reference.  Sample
 *   ids and depths are those of xrd_inverse_cdf_sampling on the reference's
 *   [200, R, P] regrouping of the hit rays.  loss_acc [4] f64 is zeroed.
 * xrd_vox_render_fwd: per-point sdf_pt [p_cap] / rgb_pt [p_cap,3] -> depth
 *   [n_rays], rgb [n_rays,3] (0 for rays without a hit), optional z_min
 *   [n_rays] and weights [n_rays,s_cap]; with loss_acc != NULL also the four
 *   loss terms: loss [5] = rgb, depth, sdf, fs (weighted) and their sum,
 *   scale [4] = the normalisers the backward needs.
 * xrd_vox_render_bwd: d loss[4] / d sdf_pt, d rgb_pt (g_sdf [p_cap], g_rgb
 *   [p_cap,3]; rows >= the point count are not written), times *g_loss (device
 *   float, NULL = 1).
 * xrd_vox_ray_grads: g_xyz [p_cap,3] -> g_rays_o, g_rays_d [n_rays,3].
 * ---------------------------------------------------------------------- */
int xrd_vox_meta_len(void);
int xrd_vox_sample_rays(int n_rays, int n_max, int s_cap, int64_t p_cap,
                        int n_nodes, const float* centres,
                        const int32_t* children, float voxel_size,
                        float max_distance, float step_size, float trunc,
                        float max_depth, const float* rays_o,
                        const float* rays_d, const float* target_d,
                        const float* noise, int32_t* hit_idx, float* hit_min,
                        float* hit_max, float* probs, float* steps,
                        int32_t* hit, int32_t* rank, int32_t* hit_rays,
                        int32_t* s_idx, float* s_depth, int32_t* cnt,
                        int32_t* offs, float* xyz, int32_t* vox, int32_t* meta,
                        double* loss_acc, xrd_stream_t stream);
/* Sharded mapping (exact against the single-process iteration): every
 * rank passes the WHOLE batch (same draws) and ray_keep [n_rays] u8 marking
 * its slice.  All rays are intersected, sorted, regrouped ([200, R, P]) and
 * sampled — so the sampler's rows and the size record (hit rays, longest row,
 * free-space / band sample counts, usable depths: the loss normalisers) are
 * those of the whole batch on every rank, no exchange needed — but only kept
 * rays leave points; the others read as rays without a hit afterwards (hit =
 * 0, cnt = 0).  ray_keep NULL = xrd_vox_sample_rays. */
int xrd_vox_sample_rays_shard(
    int n_rays, int n_max, int s_cap, int64_t p_cap, int n_nodes,
    const float* centres, const int32_t* children, float voxel_size,
    float max_distance, float step_size, float trunc, float max_depth,
    const float* rays_o, const float* rays_d, const float* target_d,
    const float* noise, const uint8_t* ray_keep, int32_t* hit_idx,
    float* hit_min, float* hit_max, float* probs, float* steps, int32_t* hit,
    int32_t* rank, int32_t* hit_rays, int32_t* s_idx, float* s_depth,
    int32_t* cnt, int32_t* offs, float* xyz, int32_t* vox, int32_t* meta,
    double* loss_acc, xrd_stream_t stream);
int xrd_vox_render_fwd(int n_rays, int s_cap, int64_t p_cap, float trunc,
                       float max_depth, const int32_t* hit,
                       const int32_t* cnt, const int32_t* offs,
                       const float* s_depth, const float* sdf_pt,
                       const float* rgb_pt, const float* target_d,
                       const float* target_rgb, const int32_t* meta,
                       float* depth, float* rgb, float* z_min, float* weights,
                       double* loss_acc, float w_rgb, float w_depth,
                       float w_sdf, float w_fs, float* loss, float* scale,
                       xrd_stream_t stream);
int xrd_vox_render_bwd(int n_rays, int s_cap, int64_t p_cap, float trunc,
                       float max_depth, const int32_t* hit,
                       const int32_t* cnt, const int32_t* offs,
                       const float* s_depth, const float* sdf_pt,
                       const float* rgb_pt, const float* target_d,
                       const float* target_rgb, const int32_t* meta,
                       const float* scale, const float* g_loss, float* g_sdf,
                       float* g_rgb, xrd_stream_t stream);
int xrd_vox_ray_grads(int n_rays, int s_cap, int64_t p_cap, const int32_t* hit,
                      const int32_t* cnt, const int32_t* offs,
                      const float* s_depth, const float* g_xyz,
                      float* g_rays_o, float* g_rays_d, xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * Multi-GPU gradient exchange (the reference is single-GPU and
 * has no collective).  One process per GPU; mapping rays are sharded over the
 * ranks and ONE all-reduce (SUM) of a flat fp32 bucket (map + decoder + pose
 * gradients) per iteration keeps the replicated Adam steps identical.  RCCL
 * (over xGMI) is bound at run time: xrd_comm_load(path) dlopens the given
 * librccl (NULL: "librccl.so.1"; a host process that already carries one —
 * PyTorch bundles its own — passes that path and shares the instance).
 * xrd_comm_unique_id fills xrd_comm_unique_id_bytes() bytes on ONE rank, the
 * caller distributes them (any side channel) and every rank calls
 * xrd_comm_create(id, rank, world) with its GPU current; NULL on failure
 * (xrd_last_error).  xrd_allreduce_grads sums `bucket` [n] f32 in place over
 * the ranks, enqueued on `stream` (ordered with the kernels around it; no
 * host synchronisation).  xrd_allreduce_max_i32: element-wise MAX of small
 * int32 records (the Vox-Fusion size record).
 * ---------------------------------------------------------------------- */
int xrd_comm_load(const char* rccl_path);
int xrd_comm_unique_id_bytes(void);
int xrd_comm_unique_id(void* out_id);
void* xrd_comm_create(const void* id, int rank, int world);
int xrd_comm_world(void* comm);
int xrd_allreduce_grads(void* comm, float* bucket, int64_t n,
                        xrd_stream_t stream);
int xrd_allreduce_max_i32(void* comm, int32_t* values, int64_t n,
                          xrd_stream_t stream);
void xrd_comm_destroy(void* comm);

/* ------------------------------------------------------------------------
 * SplaTAM Gaussian rasteriser — replaces the unvendored CUDA module
 * diff_gaussian_rasterization (-w-depth @ cb65e4b): GaussianRasterizer(
 * raster_settings)(means3D, means2D, opacities, colors_precomp, scales,
 * rotations) -> (color[3,H,W], radii[N], depth[1,H,W]); reference call sites
 * slam/model_components/gaussian_cloud_splatam.py:63-69,267-268 and
 * slam/common/common.py:592-619 (GaussianRasterizationSettings).
 * The pipeline is exposed phase by phase; the inclusive scan of
 * tiles_touched and the 64-bit key sort between phases are the caller's
 * (rocPRIM via torch.cumsum / torch.sort in the Python shim).
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t image_height, image_width;
  float tanfovx, tanfovy;
  float bg[3];
  float scale_modifier;
  float viewmatrix[16]; /* w2c TRANSPOSED, as common.py:599 passes it */
  float projmatrix[16]; /* (proj @ w2c) TRANSPOSED, common.py:605       */
} xrd_gs_camera;

/* per Gaussian: view depth, pixel centre xy[N,2], conic+opacity[N,4], 3-sigma
 * radius (0 = culled), tile rectangle rect[N,4]=(x0,y0,x1,y1), tile count */
int xrd_gs_preprocess(const xrd_gs_camera* cam, int n, const float* means3D,
                      const float* scales, const float* rotations,
                      const float* opacities, float* depths, float* xy,
                      float* conic_opacity, int32_t* radii, int32_t* rect,
                      int32_t* tiles_touched, xrd_stream_t stream);
/* Tile-band sharding of one image over ranks (the reference
 * rasterises the whole image in one process,
 * slam/model_components/gaussian_cloud_splatam.py:47-78): clip rect to the tile
 * rows [tile_row0, tile_row1) and recount tiles_touched, in place, after
 * xrd_gs_preprocess.  Binning / blend / backward then cover the band only
 * (tiles outside it composite nothing: background). */
int xrd_gs_band_clip(int n, int tile_row0, int tile_row1, int32_t* rect,
                     int32_t* tiles_touched, xrd_stream_t stream);
/* keys[i] = (tile_id << 32) | float_bits(depth), values[i] = Gaussian id, at
 * the offsets given by the INCLUSIVE scan of tiles_touched */
int xrd_gs_duplicate_keys(int n, int image_width, const int32_t* rect,
                          const int64_t* offsets_inclusive,
                          const float* depths, int64_t* keys, int32_t* values,
                          xrd_stream_t stream);
/* ranges[tile] = [start,end) in the sorted list; pre-zero ranges */
int xrd_gs_tile_ranges(int64_t n_keys, const int64_t* sorted_keys,
                       int32_t* ranges, xrd_stream_t stream);
/* The same binning as ONE call without a host sync (csrc/gs_bin.hip):
 * inclusive scan of tiles_touched, key duplication into a static-capacity
 * array (key_capacity pairs; slots past the live count sort last and are
 * ignored), 64-bit radix sort restricted to the live key bits, per-tile
 * ranges.  point_list [key_capacity] i32, ranges [tiles,2] i32 (zeroed here),
 * n_keys: device i64 = the TRUE pair count of this pass (> key_capacity: the
 * pass dropped pairs; the caller re-sizes).  workspace:
 * xrd_gs_bin_ws_bytes(n, key_capacity, W, H) bytes. */
int64_t xrd_gs_bin_ws_bytes(int n, int64_t key_capacity, int image_width,
                            int image_height);
int xrd_gs_bin(int n, int image_width, int image_height, const int32_t* rect,
               const int32_t* tiles_touched, const float* depths,
               int64_t key_capacity, void* workspace, int32_t* point_list,
               int32_t* ranges, int64_t* n_keys, xrd_stream_t stream);
/* xrd_gs_bin that also returns what a per-Gaussian gradient reduction needs:
 * offsets [n] int64, the inclusive scan of tiles_touched — the (Gaussian,
 * tile) pairs of Gaussian i carry the PRE-SORT indices [offsets[i-1],
 * offsets[i]) — and key_pos [2 * key_capacity] int32: entries [0, cap) map a
 * sorted position to its pair's pre-sort index, entries [cap, 2 cap) map a
 * pre-sort index to the Gaussian id, or -1 for a pair a full list dropped. */
int xrd_gs_bin2(int n, int image_width, int image_height, const int32_t* rect,
                const int32_t* tiles_touched, const float* depths,
                int64_t key_capacity, void* workspace, int32_t* point_list,
                int32_t* ranges, int64_t* n_keys, int32_t* key_pos,
                int64_t* offsets, xrd_stream_t stream);

/* Tile blend, fourth formulation (csrc/gs_blend.hip; same call sites of the
 * reference as xrd_gs_render_fwd / _bwd: the forward / backward of
 * GaussianRasterizer, slam/model_components/gaussian_cloud_splatam.py:63-69,
 * 267-268).  colors_b / out_color_b / dL_dcolor_b / dL_dcolors_b NULL: one
 * colour set; else two sets blended with the same weights (SplaTAM's rgb and
 * depth / silhouette renders).
 *   forward: as xrd_gs_render_fwd[2]; ckpt (xrd_gs_blend_ckpt_floats floats:
 *     one packed 64-byte record per sorted key) is written by the forward
 *     (point_list non-NULL) and read by both directions.
 *   backward: front to back, one wave per 8x8 sub-tile; key_grad
 *     [key_capacity][12] scratch (one gradient row per (Gaussian, tile) pair,
 *     filed under the pair's pre-sort index: key_pos / offsets of
 *     xrd_gs_bin2); the per-Gaussian outputs are OVERWRITTEN (no atomics,
 *     nothing to zero): dL_dmean2D [n,2], dL_dconic [n,3], dL_dopacity [n],
 *     dL_dcolors_a/_b [n,3].  out_color_a/_b: the forward's images. */
int64_t xrd_gs_blend_ckpt_floats(int64_t key_capacity, int image_width,
                                 int image_height);
int xrd_gs_blend_fwd(const xrd_gs_camera* c, const int32_t* ranges,
                     const int32_t* point_list, const float* xy,
                     const float* colors_a, const float* colors_b,
                     const float* conic_opacity, const float* depths,
                     float* out_color_a, float* out_color_b, float* out_depth,
                     float* final_T, int32_t* n_contrib, float* ckpt,
                     xrd_stream_t stream);
int xrd_gs_blend_bwd(const xrd_gs_camera* c, int n, int64_t key_capacity,
                     const int32_t* ranges, const int32_t* point_list,
                     const int32_t* key_pos, const int64_t* offsets,
                     const float* xy, const float* conic_opacity,
                     const float* colors_a, const float* colors_b,
                     const float* final_T, const int32_t* n_contrib,
                     const float* out_color_a, const float* out_color_b,
                     const float* dL_dcolor_a, const float* dL_dcolor_b,
                     const float* ckpt, float* key_grad, float* dL_dmean2D,
                     float* dL_dconic, float* dL_dopacity, float* dL_dcolors_a,
                     float* dL_dcolors_b, xrd_stream_t stream);
/* SplaTAM per-iteration glue around the raster pass, one launch each way.
 *
 * xrd_gs_prepare_fwd: transform_to_frame + both render-variable dictionaries
 * (slam/model_components/slam_helpers_splatam.py:263-292, 205-260; call site
 * gaussian_cloud_splatam.py:47-62).  pose [4,4] row-major is w2c
 * (pose_is_c2w = 0) or the frame's c2w (pose_is_c2w = 1: the rigid inverse
 * [R^T | -R^T t] is evaluated in the kernel, replacing torch.inverse of
 * slam/algorithms/splatam.py:63).  Outputs: pts [N,3] camera-frame centres,
 * rotations [N,4] = normalize(unnorm_rot), opacities [N] = sigmoid,
 * scales [N,3] = exp(log_scales) tiled, ds_colors [N,3] = (z, 1, z^2) with z
 * the depth of pts under first_w2c (third row used).
 * xrd_gs_prepare_bwd: any upstream gradient may be NULL (zero); Gaussian
 * gradients are written when their output pointer is non-NULL; g_pose [16]
 * (gradient w.r.t. the matrix that was passed in, bottom row 0) when non-NULL,
 * then acc [16] (zero on entry, zero again on return) is required. */
int xrd_gs_prepare_fwd(int n, const float* means3D, const float* unnorm_rot,
                       const float* logit_opacities, const float* log_scales,
                       const float* pose, int pose_is_c2w,
                       const float* first_w2c, float* pts, float* rotations,
                       float* opacities, float* scales, float* ds_colors,
                       xrd_stream_t stream);
int xrd_gs_prepare_bwd(int n, const float* means3D, const float* unnorm_rot,
                       const float* logit_opacities, const float* log_scales,
                       const float* pose, int pose_is_c2w,
                       const float* first_w2c, const float* g_pts,
                       const float* g_rotations, const float* g_opacities,
                       const float* g_scales, const float* g_ds_colors,
                       float* g_means3D, float* g_unnorm_rot,
                       float* g_logit_opacities, float* g_log_scales,
                       float* acc, float* g_pose, xrd_stream_t stream);
/* GaussianSplatting.get_loss_dict (slam/models/gaussian_splatting.py:102-160)
 * for use_l1 = True, ignore_outlier_depth_loss = False: masked L1 depth and
 * L1 colour — tracking: sums (colour over the same mask when use_sil);
 * mapping: means (colour x rgb_l1_scale, 0.8 in the reference; the SSIM part
 * is the caller's) — times the weights.  rgb / depth_sil [3,H,W] renders,
 * target_d [H,W], target_rgb [H,W,3].  stats [8] doubles (sums and counts)
 * is kept for the backward; no host read-back, no boolean-mask indexing. */
int xrd_gs_loss_fwd(int H, int W, int is_mapping, int use_sil, float sil_thres,
                    float w_depth, float w_rgb, float rgb_l1_scale,
                    const float* rgb, const float* depth_sil,
                    const float* target_d, const float* target_rgb,
                    double* stats, float* loss_depth, float* loss_rgb,
                    xrd_stream_t stream);
int xrd_gs_loss_bwd(int H, int W, int is_mapping, int use_sil, float sil_thres,
                    float w_depth, float w_rgb, float rgb_l1_scale,
                    const float* rgb, const float* depth_sil,
                    const float* target_d, const float* target_rgb,
                    const double* stats, const float* g_loss_depth,
                    const float* g_loss_rgb, float* g_rgb, float* g_depth_sil,
                    xrd_stream_t stream);
int xrd_gs_render_fwd(const xrd_gs_camera* cam, const int32_t* ranges,
                      const int32_t* point_list, const float* xy,
                      const float* colors, const float* conic_opacity,
                      const float* depths, float* out_color, float* out_depth,
                      float* final_T, int32_t* n_contrib, xrd_stream_t stream);
/* per-Gaussian gradients are ACCUMULATED (pre-zero): dL_dmean2D[N,2] (w.r.t.
 * ndc, i.e. what the CUDA module reports as means2D.grad), dL_dconic[N,3]
 * (true partials w.r.t. conic a,b,c), dL_dopacity[N], dL_dcolors[N,3] */
int xrd_gs_render_bwd(const xrd_gs_camera* cam, const int32_t* ranges,
                      const int32_t* point_list, const float* xy,
                      const float* conic_opacity, const float* colors,
                      const float* final_T, const int32_t* n_contrib,
                      const float* dL_dcolor, float* dL_dmean2D,
                      float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                      xrd_stream_t stream);
/* The same with TWO colour sets blended by the same weights in one pass —
 * SplaTAM's rgb render and its (z, 1, z^2) depth / silhouette render share
 * every Gaussian's geometry and opacity
 * (slam/model_components/gaussian_cloud_splatam.py:63-69): one preprocess,
 * one binning, one forward and one backward instead of two of each. */
int xrd_gs_render_fwd2(const xrd_gs_camera* c, const int32_t* ranges,
                       const int32_t* point_list, const float* xy,
                       const float* colors_a, const float* colors_b,
                       const float* conic_opacity, const float* depths,
                       float* out_color_a, float* out_color_b,
                       float* out_depth, float* final_T, int32_t* n_contrib,
                       xrd_stream_t stream);
int xrd_gs_render_bwd2(const xrd_gs_camera* c, const int32_t* ranges,
                       const int32_t* point_list, const float* xy,
                       const float* conic_opacity, const float* colors_a,
                       const float* colors_b, const float* final_T,
                       const int32_t* n_contrib, const float* dL_dcolor_a,
                       const float* dL_dcolor_b, float* dL_dmean2D,
                       float* dL_dconic, float* dL_dopacity,
                       float* dL_dcolors_a, float* dL_dcolors_b,
                       xrd_stream_t stream);
int xrd_gs_preprocess_bwd(const xrd_gs_camera* cam, int n,
                          const float* means3D, const float* scales,
                          const float* rotations, const int32_t* radii,
                          const float* dL_dmean2D, const float* dL_dconic,
                          float* dL_dmeans3D, float* dL_dscales,
                          float* dL_drotations, xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * Point-SLAM neighbour search — replaces the FAISS index the reference builds
 * in slam/model_components/neural_point_cloud.py:46-52 and queries at :255
 * (`index.search(x, 8)` -> squared L2 distances + ids).  Exact kNN within
 * max_radius on a uniform grid (FAISS-IVF is approximate),
 * ties by smaller id, (FLT_MAX, -1) for missing neighbours.  Build = cell ids
 * -> caller sorts points by cell id (torch.sort) -> cell ranges.
 * ---------------------------------------------------------------------- */
/* origin[3], dims[3] are HOST arrays; cell_ids[n] i64 = linear cell index
 * (z major) of every point, coordinates clamped into the grid */
int xrd_knn_cell_ids(int64_t n, const float* points, const float* origin,
                     float cell, const int32_t* dims, int64_t* cell_ids,
                     xrd_stream_t stream);
/* cell_start/cell_end [n_cells] i32, pre-zeroed; from the SORTED cell ids */
int xrd_knn_cell_ranges(int64_t n, const int64_t* sorted_cell_ids,
                        int32_t* cell_start, int32_t* cell_end,
                        xrd_stream_t stream);
/* queries [m,3]; sorted_points [n,3] + sorted_ids [n] (original ids) in cell
 * order; k must be 8 (Point-SLAM nn_num); out_d2 [m,8] f32 ascending,
 * out_idx [m,8] i64 */
int xrd_knn_search(int64_t m, const float* queries, const float* sorted_points,
                   const int32_t* sorted_ids, const float* origin, float cell,
                   const int32_t* dims, const int32_t* cell_start,
                   const int32_t* cell_end, int k, float max_radius,
                   float* out_d2, int64_t* out_idx, xrd_stream_t stream);

/* xrd_knn_search that also counts, per query, the neighbours strictly inside
 * the query's own radius (radius_q [m], or radius_all when NULL):
 * NeuralPointCloud.find_neighbors_faiss's (D < r^2).sum(-1)
 * (slam/model_components/neural_point_cloud.py:268-274) without the four
 * torch launches. */
int xrd_knn_search_count(int64_t m, const float* queries,
                         const float* sorted_points, const int32_t* sorted_ids,
                         const float* origin, float cell, const int32_t* dims,
                         const int32_t* cell_start, const int32_t* cell_end,
                         int k, float max_radius, float* out_d2,
                         int64_t* out_idx, const float* radius_q,
                         float radius_all, int32_t* n_within,
                         xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * One-launch replacements for the small-op chains around the render call of a
 * NICE-SLAM iteration (each is ~20-60 tiny torch kernels in the reference).
 * ---------------------------------------------------------------------- */
/* get_samples + bbox filter (slam/common/common.py:39-122,188-227,
 * slam/algorithms/nice_slam.py:181-194): crop_idx[n] i64 are the indices drawn
 * in the cropped row-major pixel grid (rows [h0,..), cols [w0, w0+crop_width));
 * depth_img [H*W], rgb_img [H*W,3], c2w [16] device row-major.  keep[i]=1 when
 * the bbox exit distance >= sensor depth; dmax (optional device float, caller
 * zeroes) receives max(depth) over kept rays (conv_onet.py:418,455). */
int xrd_sample_rays(int n, int image_width, int h0, int w0, int crop_width,
                    float fx, float fy, float cx, float cy,
                    const double* bound6, const int64_t* crop_idx,
                    const float* depth_img, const float* rgb_img,
                    const float* c2w, float* rays_o, float* rays_d,
                    float* tgt_d, float* tgt_rgb, uint8_t* keep, float* dmax,
                    xrd_stream_t stream);
/* the same for the n_frames (<= 16) frames of a mapping window in ONE launch,
 * poses given as parameters: pose_t[f] -> t[3], pose_q[f] -> quaternion
 * (r,i,j,k) of frame f (host arrays of device pointers; OptimizablePose with
 * rot_rep 'quat', slam/utils/opt_pose.py:51-69).  crop_idx [n_frames*n], all
 * outputs frame-major [n_frames*n, ...]; c2w_out [n_frames*16] receives the
 * matrices.  Equals n_frames x (xrd_pose_quat_fwd + xrd_sample_rays) up to
 * the last ulp of rays_d (the matrix stays in registers). */
int xrd_sample_rays_multi(int n_frames, int n, int image_width, int h0, int w0,
                          int crop_width, float fx, float fy, float cx,
                          float cy, const double* bound6,
                          const int64_t* crop_idx,
                          const float* const* depth_imgs,
                          const float* const* rgb_imgs,
                          const float* const* pose_t,
                          const float* const* pose_q, float* c2w_out,
                          float* rays_o, float* rays_d, float* tgt_d,
                          float* tgt_rgb, uint8_t* keep, float* dmax,
                          xrd_stream_t stream);
/* g_pose [n_frames*7] = per frame [d/dt (3), d/dq (4)]: pose-parameter
 * gradients (= xrd_sample_rays_bwd followed by xrd_pose_quat_bwd, per frame) */
int xrd_sample_rays_multi_bwd(int n_frames, int n, int image_width, int h0,
                              int w0, int crop_width, float fx, float fy,
                              float cx, float cy, const int64_t* crop_idx,
                              const float* const* pose_t,
                              const float* const* pose_q,
                              const float* g_rays_o, const float* g_rays_d,
                              float* g_pose, xrd_stream_t stream);
/* g_c2w[16] = d loss / d c2w from the ray gradients (rotation via rays_d,
 * translation via rays_o) */
int xrd_sample_rays_bwd(int n, int image_width, int h0, int w0, int crop_width,
                        float fx, float fy, float cx, float cy,
                        const int64_t* crop_idx, const float* g_rays_o,
                        const float* g_rays_d, float* g_c2w,
                        xrd_stream_t stream);
/* ConvOnet.get_loss_dict (slam/models/conv_onet.py:145-185) and its gradient
 * in one launch: loss (f64 scalar) + g_depth[n] f64 + g_rgb[n,3] f32.
 * tracking: sum |d-d^|/sqrt(var+1e-10) over (res < 10*median(res)) & d>0 &
 * keep, + w_color*L1 colour on the same rays; mapping: L1 depth over d>0 &
 * keep, + w_color*L1 colour over keep when use_color.  n <= 8192. */
int xrd_nice_loss(int n, int is_mapping, int use_color, int handle_dynamic,
                  float w_color, const double* depth, const double* var,
                  const float* rgb, const float* tgt_d, const float* tgt_rgb,
                  const uint8_t* keep, double* loss, double* g_depth,
                  float* g_rgb, xrd_stream_t stream);
/* OptimizablePose.matrix() for rot_rep='quat' (slam/utils/opt_pose.py:51-76):
 * c2w = [R(q) | t], q = (r,i,j,k) not necessarily unit */
int xrd_pose_quat_fwd(const float* t3, const float* q4, float* c2w16,
                      xrd_stream_t stream);
int xrd_pose_quat_bwd(const float* q4, const float* g_c2w16, float* g_t3,
                      float* g_q4, xrd_stream_t stream);
/* OptimizablePose.matrix() for rot_rep='axis_angle' (slam/utils/opt_pose.py:
 * 51-95, Rodrigues; exactly I when |r| <= 1e-8), n poses per launch:
 * r3[n,3], t3[n,3] -> c2w16[n,16]; backward to g_r3, g_t3 */
int xrd_pose_aa_fwd(int n, const float* r3, const float* t3, float* c2w16,
                    xrd_stream_t stream);
int xrd_pose_aa_bwd(int n, const float* r3, const float* g_c2w16, float* g_r3,
                    float* g_t3, xrd_stream_t stream);
/* Pose hand-over between frames on the device (no host round trip between the
 * last tracking iteration of a frame and the first of the next).
 * xrd_pose_from_matrix: OptimizablePose.from_matrix (slam/utils/opt_pose.py:
 * 97-110; frame.py:24-36): c2w16 -> vec = [t(3), rot], rot = unit quaternion
 * (r,i,j,k), r >= 0 (XRD_ROT_QUAT, 7 floats) or axis-angle (XRD_ROT_AXIS_ANGLE,
 * 6 floats).  xrd_pose_from_matrix_checked: the same + Frame's consistency
 * check of an initial pose (slam/common/frame.py:24-29, |c2w - matrix of the
 * parameters| <= 1e-3) without its host read: the largest deviation of the
 * rotation rebuilt from the quaternion is folded into dev_max[0] (a device
 * float the caller zeroes and reads when it reads poses back anyway; NaN
 * sticks).  xrd_pose_predict: the tracker's constant-velocity start
 * (slam/pipeline/tracker.py:185-199): next = (prev @ inv(prev2)) @ prev. */
enum { XRD_ROT_AXIS_ANGLE = 0, XRD_ROT_QUAT = 1 };
int xrd_pose_from_matrix(int rot_rep, const float* c2w16, float* vec,
                         xrd_stream_t stream);
int xrd_pose_from_matrix_checked(int rot_rep, const float* c2w16, float* vec,
                                 float* dev_max, xrd_stream_t stream);
int xrd_pose_predict(const float* prev16, const float* prev2_16,
                     float* next16, xrd_stream_t stream);
/* torch.optim.Adam step on a small dense tensor, step count on the device */
int xrd_adam_dense(float* param, const float* grad, float* m, float* v,
                   int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, const int32_t* step_dev,
                   xrd_stream_t stream);
/* xrd_adam_dense with the self-advancing counter of xrd_adam_cells_tick: this
 * launch is step step_ticket[0] + 1; advance != 0: its last block stores that
 * number (the LAST launch of a group of parameters sharing the counter passes
 * 1, the others 0) */
int xrd_adam_dense_tick(float* param, const float* grad, float* m, float* v,
                        int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int32_t* step_ticket,
                        int advance, xrd_stream_t stream);
/* xrd_adam_dense_tick for up to XRD_ADAM_DENSE_MAX_SETS tensors in ONE launch:
 * the reference steps one torch.optim.Adam per parameter group
 * (slam/engine/optimizers.py:63-171: SplaTAM's five Gaussian tensors, a
 * model's table / decoder / pose groups), each with its own learning rate and
 * step count.  Same arithmetic per tensor as xrd_adam_dense_tick (this launch
 * is step step_ticket[0] + 1 of a set; advance != 0 stores it).  Every set
 * must have its OWN step_ticket (XRD_ERR_ARG otherwise); sets with n == 0 are
 * skipped. */
#define XRD_ADAM_DENSE_MAX_SETS 8
typedef struct {
  float* param;
  const float* grad;
  float* m;
  float* v;
  int64_t n;
  float lr;
  float weight_decay;
  int32_t* step_ticket;
  int32_t advance;
} xrd_adam_dense_set;
int xrd_adam_dense_multi(int n_sets, const xrd_adam_dense_set* sets,
                         float beta1, float beta2, float eps,
                         xrd_stream_t stream);
/* keep the pose with the lowest loss (base_algorithm.py:262-265) on device */
int xrd_track_best(const double* loss, const float* c2w16, double* best_loss,
                   float* best_c2w16, uint8_t* valid, xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * Co-SLAM fused ray renderer — replaces, for the reference's default model
 * (oneGrid, HashGrid 16x2 + OneBlob 16 bins, 2x32 bias-free MLPs, depth-guided
 * sampling), JointEncoding.render_rays / run_network / query_color_sdf /
 * sdf2weights / raw2outputs (slam/models/joint_encoding.py:250-344, 483-507,
 * 463-481, 346-374, 376-407) and ColorSDFNet_v2.forward
 * (slam/model_components/decoder_coslam.py), forward and backward.
 * One ray = n_range_d + n_sample_d <= 48 samples.
 * ---------------------------------------------------------------------- */
#define XRD_COSLAM_LEVELS 16
typedef struct xrd_coslam_scene {
  double bound[6];            /* xmin,xmax, ymin,ymax, zmin,zmax (float64 as
                                 in run_network's normalisation) */
  float lv_scale[XRD_COSLAM_LEVELS];   /* xrd_hashgrid_levels() table */
  uint32_t lv_res[XRD_COSLAM_LEVELS];
  uint32_t lv_size[XRD_COSLAM_LEVELS];
  uint32_t lv_offset[XRD_COSLAM_LEVELS];
  const float* table;         /* hash-grid parameters [entries][2] (device) */
  const float* pack;          /* packed decoder weights, xrd_coslam_pack_len()
                                 floats (device) */
  const float* t_near;        /* [n_range_d] linspace(-range_d, range_d) */
  const float* t_far;         /* [n_range_d] linspace(near, far) (depth<=0) */
  const float* t_uniform;     /* [n_sample_d] linspace(near, far) */
  int32_t n_range_d, n_sample_d;
  int32_t perturb, white_bkgd;
  float trunc;                /* training_trunc */
  float sc_factor;            /* data_sc_factor */
} xrd_coslam_scene;

/* HOST: length of the flat decoder vector (state_dict order: color_net.model.0
 * [32,63], color_net.model.2 [3,32], sdf_net.model.0 [32,80], sdf_net.model.2
 * [16,32]), of the packed MFMA-fragment buffer, and of the slot-space
 * weight-gradient buffer the backward pass produces */
int xrd_coslam_flat_len(void);
int xrd_coslam_pack_len(void);
int xrd_coslam_dw_len(void);
/* HOST: pack_idx[pack_len]: flat index feeding each packed float (-1 = 0);
 * dw_idx[flat_len]: position of each flat parameter's gradient in the
 * slot-space buffer.  Either pointer may be NULL. */
int xrd_coslam_index(int32_t* pack_idx, int32_t* dw_idx);

/* forward.  target_d[n] (>0 valid), rnd[n,S] uniform draws (NULL iff
 * perturb == 0).  out: z_vals[n,S], raw[n,S,4] (rgb logits, sdf),
 * maps[n,8] = rgb(3), depth, depth_var, acc, disp, 0 */
int xrd_coslam_render_fwd(const xrd_coslam_scene* scene, int n_rays,
                          const float* rays_o, const float* rays_d,
                          const float* target_d, const float* rnd,
                          float* z_vals, float* raw, float* maps,
                          xrd_stream_t stream);
/* floats of workspace the backward pass needs for n_rays rays */
int64_t xrd_coslam_bwd_ws_floats(int n_rays);
/* backward from g_maps[n,8] (rgb, depth, depth_var, acc; disp ignored) and
 * g_raw[n,S,4] (may be NULL), with z_vals/raw as produced by the forward.
 * g_rays_o/g_rays_d [n,3] overwritten (both NULL: no ray gradients);
 * g_table (whole table) and g_dw[dw_len] OVERWRITTEN (both NULL: map and
 * decoder frozen, the tracking case).  The table gradient is scattered through
 * LDS-privatised 8192-entry chunks (no random global atomics). */
int xrd_coslam_render_bwd(const xrd_coslam_scene* scene, int n_rays,
                          const float* rays_o, const float* rays_d,
                          const float* z_vals, const float* raw,
                          const float* g_maps, const float* g_raw,
                          float* g_rays_o, float* g_rays_d, float* g_table,
                          float* g_dw, float* workspace, xrd_stream_t stream);
/* ... with n_extra further points (extra_x [n_extra,3] normalised like the
 * samples, extra_dfeat [n_extra, 32] = d loss / d their hash features, e.g.
 * xrd_hashgrid_tv's lattice) whose table gradient joins the samples' in the
 * SAME scatter launch (a second scatter costs as much as the first).
 * workspace: xrd_coslam_bwd_ws_floats_extra(n_rays, n_extra) floats. */
int64_t xrd_coslam_bwd_ws_floats_extra(int n_rays, int64_t n_extra);
int xrd_coslam_render_bwd_extra(
    const xrd_coslam_scene* scene, int n_rays, const float* rays_o,
    const float* rays_d, const float* z_vals, const float* raw,
    const float* g_maps, const float* g_raw, float* g_rays_o, float* g_rays_d,
    float* g_table, float* g_dw, int64_t n_extra, const float* extra_x,
    const float* extra_dfeat, float* workspace, xrd_stream_t stream);

/* Co-SLAM mapping batch (slam/algorithms/coslam.py:139-150 sample_global_rays,
 * :152-210 get_model_input).
 * xrd_sample_distinct: n_out DISTINCT indices in [0, n_total) — the device
 * counterpart of python's random.sample(range(n_total), n_out): a keyed
 * pseudo-random permutation (4-round Feistel, cycle-walking) evaluated at
 * 0..n_out-1; keys4 = 4 int64 round keys in DEVICE memory (drawn by the
 * caller's RNG).
 * xrd_pose_rays_*: rays_d[i] = R[id_i] dirs[i], rays_o[i] = t[id_i] for per-ray
 * pose ids into c2w[n_pose,4,4] (the reference's poses[ids] gather, multiply
 * and sum); backward accumulates d/dc2w per pose (g_c2w overwritten, rows 0..2
 * of every 4x4 used). dirs rows are dir_stride floats apart (bank rows: 7). */
/* ..._dev: n_total is read from DEVICE memory at execution time (a captured
 * mapping iteration samples a bank that grows between its replays); same
 * permutation as xrd_sample_distinct for the same n_total and keys. */
int xrd_sample_distinct_dev(const int64_t* n_total, int n_out,
                            const int64_t* keys4, int64_t* out_idx,
                            xrd_stream_t stream);
int xrd_sample_distinct(int64_t n_total, int n_out, const int64_t* keys4,
                        int64_t* out_idx, xrd_stream_t stream);
/* The mapping batch in one launch: bank rows at bank_idx [n_bank] (int64) of
 * the keyframe ray bank [*,7] + the current frame's pixels pix [n_cur] (int64,
 * flat) gathered from ray_dirs [H*W,3], rgb [H*W,3], depth [H*W] -> rows
 * [n_bank+n_cur,7] = (direction, rgb, depth), ids [n_bank+n_cur] int64 = pose
 * row of every ray (bank row / rays_per_keyframe; *cur_id, a device scalar,
 * for the current frame's) — coslam.py:139-150 (sample_global_rays) and the
 * concatenations of :152-210. */
int xrd_coslam_map_rows(int n_bank, const int64_t* bank_idx, const float* bank,
                        int rays_per_keyframe, int n_cur, const int64_t* pix,
                        const float* ray_dirs, const float* rgb,
                        const float* depth, const int64_t* cur_id, float* rows,
                        int64_t* ids, xrd_stream_t stream);
int xrd_pose_rays_fwd(int n, const float* dirs, int dir_stride,
                      const int64_t* pose_ids, const float* c2w, float* rays_o,
                      float* rays_d, xrd_stream_t stream);
int xrd_pose_rays_bwd(int n, int n_pose, const float* dirs, int dir_stride,
                      const int64_t* pose_ids, const float* g_rays_o,
                      const float* g_rays_d, float* g_c2w, xrd_stream_t stream);

/* JointEncoding.get_loss_dict without the smoothness term
 * (slam/models/joint_encoding.py:94-147; get_sdf_loss / get_masks of
 * slam/model_components/utils.py:100-186) on the renderer's outputs:
 * loss5 = {total, rgb, depth, sdf, fs} (weighted terms), g_maps[n,8] and
 * g_raw[n,S,4] = d total / d (maps, raw).  trunc = training_trunc *
 * data_sc_factor.  The depth term averages over valid-depth rays (0 when there
 * is none).  workspace: n*8 floats. */
int xrd_coslam_loss(int n_rays, int n_samples, float w_rgb, float w_depth,
                    float w_sdf, float w_fs, float trunc, float depth_trunc,
                    float rgb_missing, const float* maps, const float* z_vals,
                    const float* raw, const float* target_d,
                    const float* target_rgb, float* loss5, float* g_maps,
                    float* g_raw, float* workspace, xrd_stream_t stream);
/* xrd_coslam_loss with the batch size read on the device: the first *n_live
 * of the n_rays rows are the batch (normalisers, balancing weights), the rows
 * behind it get zero gradients — a persistent mapping graph renders a
 * capacity batch whose current-frame part (mapping_sample // n_keyframes rays,
 * slam/algorithms/coslam.py:166-170) shrinks as keyframes are added. */
int xrd_coslam_loss_live(int n_rays, int n_samples, float w_rgb, float w_depth,
                         float w_sdf, float w_fs, float trunc,
                         float depth_trunc, float rgb_missing,
                         const float* maps, const float* z_vals,
                         const float* raw, const float* target_d,
                         const float* target_rgb, const int32_t* n_live,
                         float* loss5, float* g_maps, float* g_raw,
                         float* workspace, xrd_stream_t stream);
/* The same in two steps, for a batch SHARDED over ranks (multi-GPU mapping):
 * stats[n,8] per ray = {n_fs, n_sdf, S_fs, S_sdf, valid, depth err^2, rgb
 * err^2, w}; the caller sums columns 0..6 over its rays, all-reduces the seven
 * sums over ranks and hands them in as totals7 (device, float64) with the
 * global ray count: every normaliser and the batch-global balancing weights
 * are then those of the whole batch, so the all-reduced gradient equals the
 * single-GPU one.  totals7 == NULL: this rank's rays are the whole batch. */
int xrd_coslam_loss_stats(int n_rays, int n_samples, float trunc,
                          float depth_trunc, float rgb_missing,
                          const float* maps, const float* z_vals,
                          const float* raw, const float* target_d,
                          const float* target_rgb, float* stats,
                          xrd_stream_t stream);
int xrd_coslam_loss_grads(int n_rays, int n_samples, float w_rgb, float w_depth,
                          float w_sdf, float w_fs, float trunc,
                          float depth_trunc, float rgb_missing,
                          const float* maps, const float* z_vals,
                          const float* raw, const float* target_d,
                          const float* target_rgb, const float* stats,
                          const double* totals7, int64_t n_rays_total,
                          float* loss5, float* g_maps, float* g_raw,
                          xrd_stream_t stream);

/* SSIM map of SplaTAM's mapping colour loss (calc_ssim / _ssim,
 * slam/model_components/slam_external_splatam.py:59-96): 11x11 Gaussian window
 * (sigma 1.5), zero padding, per channel; img*, ssim_map [C,H,W] f32.  The
 * three d_* maps (may all be NULL) keep d ssim / d (mu1, E[x^2], E[xy]) for
 * xrd_ssim_bwd, which returns d loss / d img1 from g_map = d loss / d ssim_map
 * (img2, the target image, gets no gradient). */
int xrd_ssim_fwd(int channels, int height, int width, const float* img1,
                 const float* img2, float* ssim_map, float* d_mu1, float* d_e11,
                 float* d_e12, xrd_stream_t stream);
int xrd_ssim_bwd(int channels, int height, int width, const float* img1,
                 const float* img2, const float* g_map, const float* d_mu1,
                 const float* d_e11, const float* d_e12, float* g_img1,
                 xrd_stream_t stream);

/* ------------------------------------------------------------------------
 * Map maintenance between iterations (csrc/map_ops.hip).
 * ---------------------------------------------------------------------- */
/* Stable compaction of n rows of up to 24 arrays by ONE keep mask — the
 * reference's chains of `tensor[mask]` (one compaction and one size read-back
 * per tensor): SplaTAM remove_points / get_pointcloud(mask)
 * (slam/model_components/gaussian_cloud_splatam.py:84-111,355-399), Point-SLAM
 * and Vox-Fusion row selections.  keep [n] u8; array j has row_words[j] 4-byte
 * words a row, src[j] -> dst[j] (dst rows [0, count) are written; dst may be
 * sized n); ws: xrd_compact_ws_ints(n) i32; count: device i32 = rows kept.
 * n_arrays == 0 only counts. */
int64_t xrd_compact_ws_ints(int64_t n);
int xrd_compact_rows(int64_t n, const uint8_t* keep, int n_arrays,
                     const void* const* src, void* const* dst,
                     const int32_t* row_words, int32_t* ws, int32_t* count,
                     xrd_stream_t stream);
/* first[i] = 1 when row i of voxels [n,3] i32 is the first occurrence of its
 * voxel — replaces torch.unique(dim=0) + first-index scatter in front of the
 * octree insertion (slam/models/sparse_voxel.py:333-340; the octree creates a
 * node for a voxel's first occurrence only, so first-occurrence order keeps
 * node and vertex ids).  Open-addressing table: table_keys [table_size] u64,
 * table_rows [table_size] i32, table_size a power of two >= 2 n; err: device
 * i32, 0 on success (1: coordinate outside +-2^20, 2: table full). */
int xrd_voxel_first_flags(int64_t n, const int32_t* voxels, uint8_t* first,
                          uint64_t* table_keys, int32_t* table_rows,
                          int64_t table_size, int32_t* err,
                          xrd_stream_t stream);
/* Point-SLAM cal_dynamic_radius (slam/algorithms/point_slam.py:326-354 with
 * slam/common/common.py:74-88): luma of the f32 image rgb [H,W,3], Sobel
 * magnitude with reflected borders in f64, clipped to [0, thresh],
 * np.interp over (0, 0.01, thresh) -> (add_max, add_max, add_min) and the same
 * times query_ratio; r_add, r_query [H,W] f64. */
int xrd_point_dynamic_radius(int H, int W, const float* rgb, double thresh,
                             double add_max, double add_min,
                             double query_ratio, double* r_add,
                             double* r_query, xrd_stream_t stream);
/* Point-SLAM add_neural_points (slam/model_components/neural_point_cloud.py:
 * 109-221) in two launches around the neighbour count: sensor points
 * pts = o + d * depth [n,3]; then, with n_within [n] i32 = neural points
 * inside the add radius of each sensor point (xrd_knn_search_count; NULL: the
 * cloud is empty), rays with depth > 0 and no neighbour append, in ray order,
 * their sensor point (out_pos), colour * 255 (out_rgb) and n_add points along
 * the ray (out_pts [.., n_add, 3]): z = depth + lin[j] (fix_interval) or
 * near_end * depth * (1 - lin[j]) + far_end * depth * lin[j].  Outputs sized
 * for n rays; count: device i32 = rays kept. */
int xrd_point_sensor_points(int64_t n, const float* rays_o, const float* rays_d,
                            const float* depth, float* pts,
                            xrd_stream_t stream);
int xrd_point_insert(int n, const float* rays_o, const float* rays_d,
                     const float* depth, const float* color,
                     const float* pts_gt, const int32_t* n_within,
                     const float* lin, int n_add, int fix_interval,
                     float near_end, float far_end, float* out_pos,
                     float* out_rgb, float* out_pts, int32_t* count,
                     xrd_stream_t stream);
/* Point-SLAM get_mask_from_c2w (slam/algorithms/point_slam.py:356-420): mask[i]
 * = neural point i [n,3] f32 projects inside the image minus `edge` pixels, in
 * front of the camera and not behind the bilinearly sampled depth (zero
 * border; pixels without depth take the maximum sampled depth) + 0.5 m.
 * w2c [3,4] f64 device row-major (top rows of the inverse pose); depth [H,W];
 * ws_f [n*3] f32, ws_d [n] f64, ws_i [1] i32 scratch; mask [n] u8. */
int xrd_point_frustum_mask(int64_t n, const float* points, const double* w2c,
                           const float* depth, int H, int W, double fx,
                           double fy, double cx, double cy, int edge,
                           float* ws_f, double* ws_d, int32_t* ws_i,
                           uint8_t* mask, xrd_stream_t stream);

/* self test of the MFMA operand/accumulator lane mapping the kernels rely on
 * (v_mfma_f32_16x16x4_f32); out[16*16] f32 device = A(16x4)·B(4x16) */
int xrd_selftest_mfma(const float* a16x4, const float* b4x16, float* out,
                      xrd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* XRDSLAM_HIP_H */
