/*
 * gof_train_hip.h -- C ABI of the training-iteration epilogue in libgof_hip.so:
 * the per-iteration work of the reference's train.py that surrounds the rasterizer and becomes the
 * dominant cost once the rasterizer takes < 5 ms -- SSIM, the depth->normal map of the depth-normal
 * consistency loss, and the Adam update of the 59 floats per Gaussian.
 *
 * In the reference these are pure-torch functions (no native symbol):
 *   utils/loss_utils.py:30-63    ssim / _ssim            (5 depthwise 11x11 conv2d + elementwise, autograd)
 *   utils/depth_utils.py:6-35    depths_to_points / depth_to_normal (meshgrid, 2 matmuls, cross, normalize, autograd)
 *   scene/gaussian_model.py:360  torch.optim.Adam(l, lr=0.0, eps=1e-15)   (torch 2.x foreach implementation)
 * The Python mirrors (`loss_utils.ssim`, `depth_utils.depth_to_normal`, `optim.FusedAdam` in the package)
 * keep the reference's names, arguments and error behaviour and are rebound into the unchanged train.py by
 * launch/run_reference_script.py.
 *
 * Conventions as in gof_hip.h: extern "C", device pointers unless the name ends in `_host`, caller-owned
 * buffers, asynchronous on `stream` (a hipStream_t passed as void*), 0 = ok / negative GOF_E_* with text
 * in gof_last_error().
 */
#ifndef GOF_TRAIN_HIP_H_INCLUDED
#define GOF_TRAIN_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOF_SSIM_WINDOW 11   /* utils/loss_utils.py:30 (window_size default; the only value the reference uses) */

/* ---- SSIM (utils/loss_utils.py:43-63) ---------------------------------------------------- */
/* img1, img2: [planes,H,W] (planes = batch*channels; every plane is convolved with the same window, the
 * `groups=channel` depthwise conv2d of loss_utils.py:44-52, zero padding 5).  window_host: the 11 fp32
 * taps of loss_utils.py:22-24 (host memory).  Writes plane_sums[planes] = sum over the plane of ssim_map
 * (the caller divides: .mean() of loss_utils.py:60-63).  If dmaps != NULL also writes the three partial
 * derivative maps [3,planes,H,W] (d ssim_map / d mu1, d sigma1_sq, d sigma12) the backward convolves.
 * scratch: gof_ssim_scratch_bytes(). */
size_t gof_ssim_scratch_bytes(int32_t planes, int32_t W, int32_t H);
int gof_ssim_forward(int32_t planes, int32_t W, int32_t H,
                     const float* img1, const float* img2, const float* window_host,
                     float* plane_sums, float* dmaps,
                     void* scratch, size_t scratch_bytes, void* stream);
/* dL_dimg1[plane] = plane_scale[plane] * (conv(dm_dmu1) + 2 img1 conv(dm_dsigma1_sq) + img2 conv(dm_dsigma12)):
 * what autograd produces for `ssim(img1, img2)` w.r.t. img1 when d loss / d ssim_map = plane_scale[plane]
 * for every pixel of the plane.  img2 (the ground-truth image) gets no gradient. */
int gof_ssim_backward(int32_t planes, int32_t W, int32_t H,
                      const float* img1, const float* img2, const float* window_host,
                      const float* dmaps, const float* plane_scale,
                      float* dL_dimg1, void* stream);

/* ---- depth -> points -> normals (utils/depth_utils.py:6-35) ----------------------------- */
/* depth [H,W]; world_view_transform [16] as stored by scene/cameras.py:56 (the transposed world-to-camera
 * matrix; depth_utils.py:7 inverts its transpose); fx, fy as depth_utils.py:9-10 computes them.
 * Writes points [H,W,3] (depth_utils.py:20) and normals [H,W,3] (zero on the 1-pixel border, depth_utils.py:30-34). */
int gof_depth_to_normal(int32_t W, int32_t H, const float* depth, const float* world_view_transform,
                        float fx, float fy, float* normals, float* points, void* stream);
/* Gradient of the above w.r.t. depth given dL_dnormals [H,W,3] and (nullable) dL_dpoints [H,W,3]. */
int gof_depth_to_normal_backward(int32_t W, int32_t H, const float* depth, const float* world_view_transform,
                                 float fx, float fy, const float* dL_dnormals, const float* dL_dpoints,
                                 float* dL_ddepth, void* stream);

/* ---- l1_loss (utils/loss_utils.py:17-18: torch.abs(a - b).mean(); train.py:156 and the evaluation loop :328) -------------
 *      out_mean (device, 1 float) = sum|a - b| / n with a fixed summation order; backward: dL_da = grad_out[0] * sign(a - b) / n
 *      (grad_out a device scalar; the gradient w.r.t. b is its negative). */
size_t gof_l1_scratch_bytes(uint64_t n);
int gof_l1_forward(uint64_t n, const float* a, const float* b, float* out_mean, void* scratch, size_t scratch_bytes, void* stream);
int gof_l1_backward(uint64_t n, const float* a, const float* b, const float* grad_out, float* dL_da, void* stream);

/* ---- the loss of one training iteration (train.py:150-188) as ONE call ("fused L1 + SSIM + depth-normal
 *      + distortion loss kernels").  train.py composes it inline from ~60 torch launches and their autograd:
 *          Ll1 = l1_loss(image, gt); rgb_loss = (1 - l_dssim) Ll1 + l_dssim (1 - ssim(image, gt))                 :156-161
 *          distortion_loss = rendering[8].mean()                                                                  :164-167
 *          depth_normal = depth_to_normal(view, rendering[6]); render_normal = normalize(rendering[3:6], dim=0)
 *          depth_normal_loss = mean(1 - <c2w[:3,:3] @ render_normal, depth_normal>)                               :170-182
 *          loss = rgb_loss + depth_normal_loss l_dn + distortion_loss l_dist                                      :188
 *      (the decoupled-appearance variant of Ll1, :158-159, goes through a network and stays torch code).
 *      rendering [9,H,W] is the rasterizer's output, gt_image [3,H,W].  terms (device, 6 floats) receives
 *      {loss, Ll1, ssim, rgb_loss, depth_normal_loss, distortion_loss}; dL_drendering [9,H,W] (nullable: values only) receives
 *      d loss / d rendering, every element written.  Five launches, sums in a fixed order (deterministic).
 *      A train.py that calls this instead of :150-188 is shown in INTEGRATION.md; the drop-in mirrors above stay the default. */
size_t gof_train_loss_scratch_bytes(int32_t W, int32_t H);
int gof_train_loss(int32_t W, int32_t H, const float* rendering, const float* gt_image, const float* window_host /* 11 taps */,
                   const float* world_view_transform, float fx, float fy, double lambda_dssim, double lambda_depth_normal,
                   double lambda_distortion, float* terms, float* dL_drendering, void* scratch, size_t scratch_bytes, void* stream);

/* ---- Adam (scene/gaussian_model.py:360; torch/optim/adam.py _multi_tensor_adam, no weight decay / amsgrad) */
typedef struct GofAdamTensor {
    float* param;             /* [n] updated in place                          */
    const float* grad;        /* [n]                                           */
    float* exp_avg;           /* [n] updated in place                          */
    float* exp_avg_sq;        /* [n] updated in place                          */
    uint64_t n;
    float step_size;          /* -lr / (1 - beta1^step)  (host computes, as adam.py does in Python floats) */
    float bias_correction2_sqrt; /* sqrt(1 - beta2^step)                        */
} GofAdamTensor;
#define GOF_ADAM_MAX_TENSORS 16
/* One launch updates up to GOF_ADAM_MAX_TENSORS tensors (the six parameter groups of the reference). */
/* beta1, beta2, eps are doubles: torch forms (1 - beta) in double and rounds each scalar to fp32 once. */
int gof_adam_step(int32_t n_tensors, const GofAdamTensor* tensors_host,
                  double beta1, double beta2, double eps, void* stream);

/* ---- 3D smoothing filter (scene/gaussian_model.py:262-311, GaussianModel.compute_3D_filter) ------
 * For every Gaussian centre: the smallest camera-space depth over all training cameras that see it (depth > 0.2 and inside the
 * image enlarged by 15 % per side), divided by the largest focal length and scaled by sqrt(0.2); centres no camera sees get the
 * largest such depth.  The reference runs ~25 torch kernels and two boolean-mask index ops (host syncs) PER CAMERA, after every
 * densification (train.py:118,261,269); here: one launch over (points x all cameras) + one finishing launch.
 * cameras: [num_cams][GOF_FILTER_CAM_FLOATS] fp32 in DEVICE memory: R (9, row-major, used as xyz @ R exactly like the reference's
 * `xyz @ R`), T (3), focal_x, focal_y, image_width, image_height.
 * any_valid_host (host, nullable): receives 1 if at least one point is seen by a camera, else 0 (the reference raises on the
 * empty `distance[valid_points].max()`); when non-NULL the call SYNCHRONISES `stream`. */
#define GOF_FILTER_CAM_FLOATS 16
size_t gof_filter3d_ws_bytes(int64_t num_points);
int gof_compute_3d_filter(int64_t num_points, const float* xyz, int32_t num_cams, const float* cameras,
                          float* filter_3D /* [num_points] */, void* ws, size_t ws_bytes, int32_t* any_valid_host, void* stream);

/* ---- densification statistics (scene/gaussian_model.py:709-714, GaussianModel.add_densification_stats) -------------------
 * For every Gaussian i with update_filter[i] != 0:  n2 = ||grad[i,0:2]||, n1 = ||grad[i,2:3]|| (torch.norm: sqrt of the sum of
 * squares, also for the single element);  accum[i] += n2;  accum_abs[i] += n1;  accum_abs_max[i] = max(accum_abs_max[i], n1);
 * denom[i] += 1.  The reference does this with four boolean-mask read-modify-writes (each a nonzero + gather + scatter and a
 * host sync) every iteration of the densification phase; here one launch.  grad is the [P,3] gradient of the screen-space
 * points (x, y signed, z = sum of |.|, rasterizer backward); update_filter is a [P] byte/bool mask. */
int gof_add_densification_stats(int64_t num_points, const float* viewspace_grad, const uint8_t* update_filter,
                                float* xyz_gradient_accum, float* xyz_gradient_accum_abs, float* xyz_gradient_accum_abs_max,
                                float* denom, void* stream);

/* ---- densification on the device (scene/gaussian_model.py:631-707, GaussianModel.densify_and_prune) ------
 * The reference clones / splits / prunes with boolean-mask indexing: every `x[mask]` is a nonzero + gather with a host sync, and
 * the six parameter tensors and their two Adam moments are re-concatenated and re-indexed four times per densification (clone
 * postfix, split postfix, removal of the split originals, final prune).  Here the decisions become ordered INDEX LISTS on the
 * device and every tensor is rebuilt ONCE by a row gather; two 12-byte read-backs (list lengths, final row count) remain.
 *   gof_densify_select : role of every Gaussian (0 stays, 1 cloned, 2 split) from the accumulated gradients, with the reference's
 *                        thresholds (`norm(grads) >= max_grad or norm(grads_abs) >= Q`, clone if max(get_scaling) <= size_threshold,
 *                        else split; grads = accum / denom with NaN -> 0), and the ordered lists keep_idx (role != 2), clone_idx,
 *                        split_idx (each sized for num_points by the caller); counts_host[3] = their lengths.  q_abs_dev is the
 *                        torch.quantile value as a DEVICE scalar.  SYNCHRONISES the stream (the caller allocates from the counts).
 *   gof_compact_rows   : out_rows = (src_rows ? src_rows[i] : i) for the rows with keep[i] != 0, in order; *count_host = how many.
 *                        SYNCHRONISES the stream.
 *   gof_rows_gather    : out[r, :] = src[rows[r], :] if rows[r] >= 0 else extra[-rows[r] - 1, :] (zeros if extra is NULL):
 *                        parameters take their new rows from `extra`, Adam moments get zeros there.
 * ws: gof_densify_ws_bytes(n) bytes for n rows. */
size_t gof_densify_ws_bytes(int64_t n);
int gof_densify_select(int64_t num_points, const float* xyz_gradient_accum, const float* xyz_gradient_accum_abs, const float* denom,
                       const float* scale_max, float max_grad, const float* q_abs_dev, float size_threshold, uint8_t* role,
                       int32_t* keep_idx, int32_t* clone_idx, int32_t* split_idx, void* ws, size_t ws_bytes, int64_t* counts_host, void* stream);
int gof_compact_rows(int64_t n, const uint8_t* keep, const int32_t* src_rows, int32_t* out_rows, void* ws, size_t ws_bytes,
                     int64_t* count_host, void* stream);
int gof_rows_gather(int64_t n_rows, int32_t floats_per_row, const int32_t* rows, const float* src, const float* extra, float* out, void* stream);

/* ---- parameter activations of the render path (scene/gaussian_model.py:157-166, 183-194) -------------------------------------
 * render() reads three derived tensors per iteration (gaussian_renderer/__init__.py:60,70-71); in the reference each is a chain
 * of 3-10 torch elementwise kernels plus their autograd (~50 launches per iteration).  One forward and one backward launch each:
 *   scaling  [P,3]: sqrt(exp(_scaling)^2 + filter_3D^2)                                   (get_scaling_with_3D_filter, :157-162)
 *   opacity  [P,1]: sigmoid(_opacity) * sqrt(prod(s^2) / prod(s^2 + filter_3D^2)), s = exp(_scaling)  (get_opacity_with_3D_filter, :183-194)
 *   rotation [P,4]: _rotation / max(||_rotation||, 1e-12)                                 (get_rotation, :165-166: F.normalize)
 * filter_3D is [P] (the reference's (P,1) column); it receives no gradient (it is recomputed, never optimised).
 * The *_backward calls WRITE (not accumulate) the gradient w.r.t. their raw inputs. */
int gof_act_scaling(int64_t P, const float* raw_scaling, const float* filter_3D, float* out, void* stream);
int gof_act_scaling_backward(int64_t P, const float* raw_scaling, const float* filter_3D, const float* grad_out,
                             float* grad_raw_scaling, void* stream);
int gof_act_opacity(int64_t P, const float* raw_opacity, const float* raw_scaling, const float* filter_3D, float* out, void* stream);
int gof_act_opacity_backward(int64_t P, const float* raw_opacity, const float* raw_scaling, const float* filter_3D, const float* grad_out,
                             float* grad_raw_opacity, float* grad_raw_scaling, void* stream);
int gof_act_rotation(int64_t P, const float* raw_rotation, float* out, void* stream);
int gof_act_rotation_backward(int64_t P, const float* raw_rotation, const float* grad_out, float* grad_raw_rotation, void* stream);

/* ---- train.py:177-179, `c2w[:3, :3] @ render_normal.reshape(3, -1)`: a 3x3 matrix applied to the columns of X [3, N] ----------
 * Y [3, N] = A X with A(i, j) = M[i * row_stride + j * col_stride] (device memory; transpose != 0: A(i, j) = M[j * row_stride +
 * i * col_stride], the backward w.r.t. X).  One streaming launch where torch's matmul runs a GEMM of tile shape 256x16x16 on a K = 3
 * product (128 us each way at 1600x1063 on MI355X).  Bound by the launcher through train_epilogue.pose.SmallMatrix. */
int gof_rot3_apply(int64_t N, const float* M, int32_t row_stride, int32_t col_stride, int32_t transpose, const float* X, float* Y, void* stream);

#ifdef __cplusplus
}
#endif
#endif
