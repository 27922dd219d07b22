/* clslam_hip.h -- C ABI of libclslam_hip.so, the MI355X (gfx950) native library behind the
 * CL-SLAM depth_pose_prediction hot path.
 *
 * The reference (robot-learning-freiburg/CL-SLAM) is pure Python on PyTorch; its "FFI" for this
 * path is the set of ATen/cuDNN ops it calls.  Each entry point below names the reference
 * call site(s) it replaces (paths relative to the reference root; dpp.py =
 * depth_pose_prediction/depth_pose_prediction.py).  The host-side binding is
 * cl-slam_amd/clslam_hip/_lib.py (ctypes); INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions: every pointer is a DEVICE pointer to fp32 data unless stated otherwise; tensors
 * inside the library are NHWC (channels last) except the planar NCHW images / depth maps that
 * the reference's sample dict and output dict define.  Functions never allocate, never
 * synchronise, enqueue on `stream` (a hipStream_t passed as void*), and return CLSLAM_OK or a
 * negative error code; clslam_last_error() gives the message.  No exceptions cross the ABI.
 */
#ifndef CLSLAM_HIP_H
#define CLSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLSLAM_OK 0
#define CLSLAM_ERR_INVALID (-1)
#define CLSLAM_ERR_LAUNCH (-2)

#define CLSLAM_ACT_NONE 0
#define CLSLAM_ACT_RELU 1
#define CLSLAM_ACT_ELU 2
#define CLSLAM_ACT_HSWISH 3   /* x * relu6(x+3) / 6   (MobileNetV3) */
#define CLSLAM_ACT_HSIGMOID 4 /* relu6(x+3) / 6 */

#define CLSLAM_PAD_ZERO 0
#define CLSLAM_PAD_REFLECT 1

/* Library identification / error text (thread-local). */
/* ABI version: 101 = clslam_conv_desc.weight_wino appended, clslam_wino_weight_*; 100 -> 101 also covers the double* dp_partial of
 * clslam_warp_bwd / clslam_pose_bwd / clslam_loss_bwd*_pyramid (round 4); 102 = clslam_conv_desc.cu_limit appended; 103 = clslam_handoff_* added; 104 = ..._pyramid_range entry points added.
 * Bindings check it before the first call.                                                                                    */
#define CLSLAM_ABI_VERSION 104
int clslam_version(void);
const char* clslam_last_error(void);
const char* clslam_last_error_string(void); /* = clslam_last_error (the name SURVEY.md 8b lists) */
/* 16 hex digits identifying the kernel sources the library was built from (csrc/build.py source_id()); "unstamped" for a
 * build that did not go through build.py.  bench.py pairs its timings with counter files of the same id only.           */
const char* clslam_build_id(void);
/* 1 when the library was built from the HIP sources for gfx950, 0 for the CPU emulator used
 * by the host-logic tests (tests/emu); the product loader refuses anything but 1. */
int clslam_is_device_build(void);

/* ---------------------------------------------------------------------------------------------
 * clslam_conv2d: fp32 implicit-GEMM convolution on MFMA (3x3 / 1x1, NHWC, OHWI weights).
 * Replaces nn.Conv2d + BatchNorm2d(eval) + ReLU/ELU + residual add + ReflectionPad2d +
 * F.interpolate(nearest) + torch.cat at: networks/resnet_encoder.py:118-125 (torchvision
 * BasicBlock bodies), networks/layers.py:9-48, networks/depth_decoder.py:51-66,
 * networks/pose_decoder.py:40-47; and their autograd data-gradients (dgrad runs the same kernel
 * on transposed weights, see clslam_weight_transpose).
 *   out[b,oy,ox,n] = act(scale[n]*sum_{ky,kx,c} in(b, oy*stride-pad+ky, ox*stride-pad+kx, c)
 *                        * weight[n][ky][kx][c] + shift[n] + residual[b,oy,ox,n])
 *   in(...) = channel-concat of src_a (optionally nearest-2x upsampled) and src_b, padded by
 *   zeros or reflection.                                                                      */
typedef struct clslam_conv_desc {
    const float* src_a;    /* [B][in_h(/2)][in_w(/2)][ch_a]                                   */
    const float* src_b;    /* [B][in_h][in_w][ch_b] or NULL                                   */
    const float* weight;   /* [ch_out][ksize*ksize][ch_a+ch_b]                                */
    const float* scale;    /* [ch_out] or NULL (=1)   folded BatchNorm gamma/sqrt(var+eps)     */
    const float* shift;    /* [ch_out] or NULL (=0)   folded BatchNorm shift / conv bias       */
    const float* residual; /* [B][out_h][out_w][ch_out] or NULL                               */
    float* out;            /* [B][out_h][out_w][ch_out]                                       */
    int32_t batch, in_h, in_w, ch_a, ch_b, out_h, out_w, ch_out;
    int32_t ksize, stride, pad, pad_mode, upsample_a, act;
    int32_t config;        /* tile configuration, -1 = choose                                  */
    /* dgrad epilogue: out *= act'(y) where y = actgrad_src[b,oy,ox,n] is the OUTPUT of the
     * activation being differentiated (ReLU' = y>0, ELU' = y>0 ? 1 : y+1). NULL = off.        */
    const float* actgrad_src;
    int32_t actgrad_kind;
    /* Optional split-K scratch for the small-M 3x3 layers (layer3/4, pose decoder, upconv_4_x at 192x640:
     * fewer output tiles than CUs).  Device memory, ZERO-FILLED ONCE by the caller, private to the
     * stream the conv is launched on (convs on one stream may share it: the first 64 KiB are per-tile
     * arrival counters that every launch leaves at zero).  NULL/0: no split-K.  Results do not depend on
     * which workgroup finishes last: partial tiles are summed in split order.                       */
    void* workspace;
    size_t workspace_bytes;
    /* Optional (ABI version >= 101): the same filter pre-transformed for the Winograd F(2x2,3x3) kernel by
     * clslam_wino_weight_transform (3x3, stride 1, zero padding, one source; needs `workspace`).  NULL: the direct kernels.
     * Frozen weights (the two ResNet encoders) are transformed once per load; `weight` must still be set.  The kernel needs
     * 64 KiB + 64 KiB per workgroup of `workspace` (hand-off flags -- eight per workgroup, holding the launch's epoch, never
     * reset -- and one partial slab each); with config < 0 a launch that does not fit falls back to the direct kernels.   */
    const float* weight_wino;
    /* Optional (ABI version >= 102): how many compute units a PERSISTENT launch (the stream-K and Winograd kernels: one or two
     * resident workgroups per CU that walk the whole layer) may occupy; 0 = all of them.  A caller that runs independent
     * branches on several streams (the depth and the pose network of one step) gives each launch half of the chip: two such
     * launches then run side by side, each workgroup walks twice the units and the per-launch prologue / hand-off / epilogue
     * phases cost half.  Measured on MI355X at 192x640: -2.9 % per step at 5 triplets, +3.7 % at 33.  Results
     * do not depend on it beyond the summation order of the hand-off (fixed per (shape, cu_limit)).                       */
    int32_t cu_limit;
} clslam_conv_desc;
int clslam_conv2d(const clslam_conv_desc* desc, void* stream);
/* the tile configuration clslam_conv2d uses for desc->config < 0 (profiling / reporting) */
int clslam_conv2d_pick_config(const clslam_conv_desc* desc);
/* U = G g G^T of Winograd F(2x2,3x3) for a 3x3 filter w [ch_out][9][ch_in] (ch_in a multiple of 16), formed in double and
 * rounded once, in the layout the kernel stages through LDS: clslam_wino_weight_size(ch_out, ch_in) floats
 * ([ceil(ch_out/64)][ch_in/8][16 positions][64][8], zero beyond ch_out).  Replaces nothing in the reference: cuDNN does the
 * same transformation inside its own Winograd convolution (networks/resnet_encoder.py:118-125 on the GPU).            */
size_t clslam_wino_weight_size(int ch_out, int ch_in);
int clslam_wino_weight_transform(const float* w, float* u, int ch_out, int ch_in, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the trainable convolutions -- replaces autograd's convolution_backward,
 * reflection_pad2d_backward, upsample_nearest2d_backward, cat backward, elu/relu backward for
 * networks/depth_decoder.py:51-71, networks/layers.py:9-48, networks/pose_decoder.py:37-54
 * (entered from dpp.py:312 `losses['loss'].backward()`).                                      */

/* wt[ci][taps-1-t][co] = w[co][t][ci] for ci < ch_in_sel: the weights of the dgrad-as-conv.    */
int clslam_weight_transpose(const float* w, float* wt, int ch_out, int taps, int ch_in, int ch_in_sel,
                            void* stream);
/* The same for nitems weight tensors in ONE launch (items is a HOST array, passed to the kernel by value): the eleven
 * dgrad weight sets of a backward pass (nine depth_decoder upconvs, pose_decoder.py:40-47 pose_0 / pose_1) depend on
 * the weights only and are produced together at its start.                                                          */
typedef struct clslam_transpose_item {
    const float* w;
    float* wt;
    int ch_out, taps, ch_in, ch_in_sel;
} clslam_transpose_item;
int clslam_weight_transpose_multi(const clslam_transpose_item* items, int nitems, void* stream);
/* dz = act'(yout) * fold(dxp): dxp is the gradient w.r.t. the PADDED conv input
 * [B][h+2*border][w+2*border][ch_stride]; border=1 folds the reflection border back
 * (ReflectionPad2d backward), pool=1 sums each 2x2 block (nearest-2x upsample backward, output
 * is [B][h/2][w/2][ch]); only channels [0,ch) are used (the skip half of a concat is dead:
 * encoders are frozen, dpp.py:308,813-819); yout may be NULL (no activation).                 */
/* bias_partial (optional) receives [clslam_fold_blocks(...)][ch] per-block column sums of dz, i.e.
 * the bias-gradient partials of the conv that produced yout (reduce with clslam_reduce_partials). */
int clslam_fold_blocks(int batch, int h, int w, int ch, int pool);
/* disp_dz (B,h,w) / disp_w (9,ch), optional (pool = 0, border = 1): additionally adds the data gradient of
 * the dispconv head that reads the same activation (what clslam_dispconv_bwd_data would accumulate into
 * dxp), evaluated per folded position; dxp may then be NULL.                                        */
int clslam_fold_act_grad(const float* dxp, const float* yout, float* dz, float* bias_partial, int batch, int h, int w,
                         int ch, int ch_stride, int border, int pool, int act, const float* disp_dz, const float* disp_w,
                         void* stream);
/* Weight gradient as an MFMA GEMM reducing over pixels; desc = the forward conv's descriptor. */
int clslam_wgrad_splits(const clslam_conv_desc* desc, int target_blocks);
int clslam_conv_wgrad(const clslam_conv_desc* desc, const float* dz, float* partial, int splits, void* stream);
/* Same contract as clslam_conv_wgrad for 3x3 stride-1 convs on images wider than 40 px, with the dZ
 * tile and the input patch resident in LDS (wgrad_patch.hip); splits from clslam_wgrad_patch_splits. */
int clslam_wgrad_patch_supported(const clslam_conv_desc* desc);
int clslam_wgrad_patch_splits(const clslam_conv_desc* desc, int target_blocks);
int clslam_conv_wgrad_patch(const clslam_conv_desc* desc, const float* dz, float* partial, int splits, void* stream);
/* out[i] = scale * sum_s partial[s*n + i] in a fixed order (deterministic).                    */
int clslam_reduce_partials(const float* partial, float* out, size_t n, int splits, float scale, void* stream);
/* Batched clslam_reduce_partials: items_dev is a DEVICE array of nitems records
 * {const float* partial; float* out; uint64_t n; int32_t splits; float scale} (32 bytes each).     */
int clslam_reduce_multi(const void* items_dev, int nitems, int blocks_per_item, void* stream);
/* clslam_reduce_multi followed, element by element, by the optimizer step of clslam_adam_step (torch.optim.Adam.step,
 * dpp.py:313) -- the single-GPU path, where nothing sits between the reduction and the update.  grad_base / param /
 * exp_avg / exp_avg_sq: the flat arenas (an item's `out` lies inside grad_base's; the same offset addresses the others);
 * every trainable element must be the output of exactly one item.  guard: see clslam_adam_step.                       */
int clslam_reduce_multi_adam(const void* items_dev, int nitems, int blocks_per_item, const float* grad_base, float* param,
                             float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps, int step,
                             const float* guard, void* stream);
/* bias gradient: column sums of x[rows][ch], stage 1 (follow with clslam_reduce_partials).     */
int clslam_colsum_blocks(int rows);
int clslam_colsum(const float* x, float* partial, int rows, int ch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder stem (frozen): (x-0.45)/0.225 -> conv7x7 s2 p3 -> eval BN -> ReLU, and maxpool 3x3 s2 p1.
 * Replaces networks/resnet_encoder.py:117-121.  img_a/img_b: planar (B,3,h,w) frames exactly as
 * the sample dict holds them (img_b = second frame of the pose pair, dpp.py:951-955, or NULL);
 * weight: the conv1 weight (64, 3*num_images, 7, 7), checkpoint OIHW order, re-packed ONCE per load by
 * clslam_stem_pack_weight into clslam_stem_packed_size(num_images) floats; out: NHWC (B,h/2,w/2,64). */
int clslam_stem_packed_size(int num_images);
int clslam_stem_pack_weight(const float* weight, float* packed, int num_images, void* stream);
int clslam_stem_conv(const float* img_a, const float* img_b, const float* weight, const float* scale,
                     const float* shift, float* out, int batch, int h, int w, int num_images, void* stream);
int clslam_maxpool3x3s2(const float* in, float* out, int batch, int h, int w, int ch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Disparity head: reflection-padded 3x3 conv C->1 + sigmoid (networks/depth_decoder.py:67-69) and
 * its backward.  x NHWC (B,h,w,ch); w [9][ch]; disp / dz planar (B,h,w); dz = dL/d(pre-sigmoid).
 * dispconv_bwd_data writes (accumulate=0) or adds (1) into the padded-domain gradient
 * dxp (B,h+2,w+2,ch) that clslam_fold_act_grad consumes.                                        */
int clslam_dispconv_fwd(const float* x, const float* w, const float* bias, float* disp, int batch, int h, int wd,
                        int ch, void* stream);
int clslam_dispconv_bwd_data(const float* dz, const float* w, float* dxp, int batch, int h, int wd, int ch,
                             int accumulate, void* stream);
int clslam_dispconv_wgrad_blocks(int pixels);
int clslam_dispconv_wgrad(const float* dz, const float* x, float* partial, int batch, int h, int wd, int ch,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pose head: pose_2 (1x1, 256->12) + spatial mean + x0.01 (networks/pose_decoder.py:44-54) and
 * its backward.  x: pose_1 output NHWC (n,hw,256) post-ReLU; pose (n,12); mean (n,256) is kept
 * for the backward; dz1 = dL/d(pre-ReLU pose_1 output).                                         */
int clslam_pose_head_fwd(const float* x, const float* w2, const float* b2, float* mean, float* pose, int n, int hw,
                         void* stream);
int clslam_pose_head_bwd(const float* dpose, const float* x, const float* w2, const float* mean, float* dz1,
                         float* dw2, float* db2, int n, int hw, float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pose -> matrices, view synthesis (forward and backward).  Replaces
 * depth_pose_prediction/utils.py:34-117 (transformation_from_parameters, invert for frame -1),
 * networks/layers.py:51-104 (BackprojectDepth, Project3D), utils.py:120-142 (disp_to_depth),
 * dpp.py:986-1017 (F.interpolate bilinear + F.grid_sample border/align_corners=True) and autograd.
 * pose (2B,12): row fi*B+b = pose-decoder output of frame idx fi (0: frame -1, 1: frame +1);
 * cam_t_cam (2,B,4,4); proj (2,B,3,4) = (K*T)[:3]; images planar (B,3,H,W); depth (B,H,W);
 * warped / dpred (2,B,3,H,W).  min_depth/max_depth <= 0 stand for None.                          */
int clslam_pose_to_proj(const float* pose, const float* kmat, float* cam_t_cam, float* proj, int batch, void* stream);
int clslam_warp_fwd(const float* disp_s, int h, int w, const float* src_m1, const float* src_p1, const float* inv_k,
                    const float* proj, float* depth, float* warped, int batch, int H, int W, float min_depth,
                    float max_depth, void* stream);
/* Pyramid forms (ONE launch for the four scales): disp = host array of 4 device pointers,
 * disp[s] (B,H>>s,W>>s); depth (4,B,H,W); warped (4,2,B,3,H,W).                                     */
int clslam_warp_fwd_pyramid(const float* const* disp, const float* src_m1, const float* src_p1, const float* inv_k,
                            const float* proj, float* depth, float* warped, int batch, int H, int W, float min_depth,
                            float max_depth, void* stream);
/* ..._range: scales [scale_lo, scale_lo + scale_count) of the same pyramid only (the same buffers, the same per-scale
 * arithmetic: four single-scale launches write bit for bit what the one launch writes).  The engine issues the coarse
 * scales' view synthesis + photometric stage + loss backward as soon as their disparity exists, beside the depth decoder's
 * remaining levels (round 5; no reference counterpart: dpp.py:981-1076 loops over the scales after the decoder).      */
int clslam_warp_fwd_pyramid_range(const float* const* disp, const float* src_m1, const float* src_p1, const float* inv_k,
                                  const float* proj, float* depth, float* warped, int batch, int H, int W, float min_depth,
                                  float max_depth, int scale_lo, int scale_count, void* stream);
/* Diagnostic (parity tests): the bilinear cell and border-clip flags the path uses per (scale, source frame, sample,
 * pixel) -- what F.grid_sample decides internally at dpp.py:1013-1017.  cells (4,2,B,H,W) int32 =
 * x0 | y0 << 12 | (x not clipped) << 24 | (y not clipped) << 25; same arguments as clslam_warp_fwd_pyramid.          */
int clslam_warp_cells_pyramid(const float* const* disp, const float* inv_k, const float* proj, int* cells, int batch, int H,
                              int W, float min_depth, float max_depth, void* stream);
/* Diagnostic: the sampling positions themselves -- coords (4, 2, batch, H, W, 2) float = (ix, iy) in pixels of the source frame
 * after grid_sample's border clip (the value the four taps are weighted by); same arguments as clslam_warp_cells_pyramid.
 * tests/test_warp_positions.py bounds them against the float64 oracle, which is what the warped-image tolerance derives from. */
int clslam_warp_coords_pyramid(const float* const* disp, const float* inv_k, const float* proj, float* coords, int batch, int H,
                               int W, float min_depth, float max_depth, void* stream);
int clslam_warp_bwd_blocks(int H, int W);
/* ddisp_up (B,H,W) = dL/d(upsampled disparity); dp_partial [B][nblk][24] block sums of dL/dproj */
int clslam_warp_bwd(const float* dpred, const float* disp_s, int h, int w, const float* src_m1, const float* src_p1,
                    const float* inv_k, const float* proj, float* ddisp_up, double* dp_partial, int batch, int H, int W,
                    float min_depth, float max_depth, void* stream);
/* dp_partial [nscale][B][nblk][24]; dpose (2B,12) = dL/d(pose-decoder output) incl. the velocity
 * term (dpp.py:1125-1146; dist0/dist1 = relative_distance(0|1) as float64, sample_w (B)).        */
int clslam_pose_bwd(const double* dp_partial, int nscale, int nblk, const float* pose, const float* kmat,
                    const double* dist0, const double* dist1, const float* sample_w, float vel_scale, float* dpose,
                    int batch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Photometric / smoothness / velocity loss and its backward.  Replaces networks/layers.py:107-137
 * (SSIM), dpp.py:1019-1192 (_compute_loss, _compute_smooth_loss incl. its flattening behaviour,
 * _compute_velocity_loss, _compute_reprojection_loss) and autograd.                               */
/* map[n] = 0.85*mean_c SSIM + 0.15*mean_c L1 of pred image n (npred x (3,H,W)) vs target n % B;
 * coef (npred,9,H,W) or NULL: SSIM derivative coefficients kept for clslam_photo_grad.           */
int clslam_photo_map(const float* pred, const float* target, float* map, float* coef, int npred, int batch, int H,
                     int W, void* stream);
int clslam_automask_blocks(int H, int W);
/* idmap/rpmap (2,B,H,W); noise (B,2,H,W) or NULL; sel (B,H,W) u8 argmin of
 * [id(-1)+noise, id(+1)+noise, reproj(-1), reproj(+1)]; partial [B][nblk] sums of the minimum.  */
int clslam_automask(const float* idmap, const float* noise, const float* rpmap, unsigned char* sel, float* partial,
                    int batch, int H, int W, void* stream);
/* psum [B][clslam_disp_mean_chunks()] partial sums of each sample's disparity map (the mean is
 * formed inside clslam_loss_finalize).                                                            */
int clslam_automask_pyramid(const float* idmap, const float* noise, const float* rpmap, unsigned char* sel, float* partial,
                            int nscale, int batch, int H, int W, void* stream);
int clslam_disp_mean_pyramid(const float* const* disp, float* psum, int batch, int H, int W, void* stream);
/* Fused clslam_photo_map(warped) + clslam_automask over the pyramid: both reprojection maps are evaluated in
 * registers and only the SELECTED frame's 9 SSIM coefficients are stored: coef_sel (4,B,9,H,W) or NULL.      */
int clslam_photo_automask_pyramid(const float* warped, const float* target, const float* idmap, const float* noise,
                                  unsigned char* sel, float* coef_sel, float* partial, int batch, int H, int W,
                                  void* stream);
/* The same with the tie-break noise of dpp.py:1055-1056 drawn INSIDE the kernel (Philox4x32-10 + Box-Muller, x 1e-5):
 * element ((scale * B + b) * H * W + pixel) of draw `offset`, key `seed` (non-zero).  Timing / production runs; parity
 * runs inject captured tensors through clslam_photo_automask_pyramid.  clslam_tie_break_noise writes the pairs
 * (n_id(-1), n_id(+1)) of elements [0, npix) of the same stream to out[2 * npix] (tests, inspection).            */
int clslam_photo_automask_pyramid_rng(const float* warped, const float* target, const float* idmap, unsigned long long seed,
                                      unsigned long long offset, unsigned char* sel, float* coef_sel, float* partial, int batch,
                                      int H, int W, void* stream);
/* scales [scale_lo, scale_lo + scale_count): noise injected (noise != NULL), drawn in the kernel (seed != 0) or absent (both 0) */
int clslam_photo_automask_pyramid_range(const float* warped, const float* target, const float* idmap, const float* noise,
                                        unsigned long long seed, unsigned long long offset, unsigned char* sel, float* coef_sel,
                                        float* partial, int batch, int H, int W, int scale_lo, int scale_count, void* stream);
int clslam_tie_break_noise(float* out, size_t npix, unsigned long long seed, unsigned long long offset, void* stream);
/* Opt-in "intended" smoothness (the per-sample edge-aware term of monodepth2 instead of the flattened-batch
 * behaviour of dpp.py:1148-1176 that the default mode reproduces).  fwd: partial[4][B][clslam_smooth_intended_chunks()]
 * sums of |d disp| * exp(-mean_c |d I|) / count over the x and y edges; finalize (after clslam_loss_finalize run with
 * n_smooth = 0): adds sum_b w_b * inv_b * T_b per scale to losses[s*4+1..3] and losses[17], keeps aux[4][B][2] =
 * (1/(mean disp + 1e-7), T); bwd: ADDS the term's gradient (through the mean normalisation and the head's sigmoid) to
 * dz[s] (B,h_s,w_s).  means = the (4,B,clslam_disp_mean_chunks()) output of clslam_disp_mean_pyramid.               */
int clslam_smooth_intended_chunks(void);
int clslam_smooth_intended_fwd(const float* const* disp, const float* const* rgb0, float* partial, int batch, int H, int W,
                               void* stream);
int clslam_smooth_intended_finalize(const float* partial, const float* means, const float* sample_w, float* losses, float* aux,
                                    int batch, int H, int W, float smooth_scale, void* stream);
int clslam_smooth_intended_bwd(const float* const* disp, const float* const* rgb0, const float* aux, const float* sample_w,
                               float* const* dz, int batch, int H, int W, float smooth_scale, void* stream);
/* LDS-tiled fused loss backward on coef_sel; dp_partial [4][B][clslam_loss_bwd2_blocks][24].                */
int clslam_loss_bwd2_blocks(int H, int W);
int clslam_loss_bwd2_pyramid(const float* const* disp, const unsigned char* sel, const float* coef_sel, const float* warped,
                             const float* target, const float* src_m1, const float* src_p1, const float* inv_k,
                             const float* proj, const float* sample_w, float* ddisp_up, double* dp_partial, int batch,
                             int H, int W, float min_depth, float max_depth, void* stream);
int clslam_loss_bwd2_pyramid_range(const float* const* disp, const unsigned char* sel, const float* coef_sel, const float* warped,
                                   const float* target, const float* src_m1, const float* src_p1, const float* inv_k,
                                   const float* proj, const float* sample_w, float* ddisp_up, double* dp_partial, int batch,
                                   int H, int W, float min_depth, float max_depth, int scale_lo, int scale_count, void* stream);
/* Fused clslam_photo_grad + clslam_warp_bwd for all four scales: sel (4,B,H,W), coef (4,2,B,9,H,W),
 * warped (4,2,B,3,H,W) -> ddisp_up (4,B,H,W), dp_partial [4][B][clslam_loss_bwd_blocks][24].          */
int clslam_loss_bwd_blocks(int H, int W);
int clslam_loss_bwd_pyramid(const float* const* disp, const unsigned char* sel, const float* coef, const float* warped,
                            const float* target, const float* src_m1, const float* src_p1, const float* inv_k,
                            const float* proj, const float* sample_w, float* ddisp_up, double* dp_partial, int batch, int H,
                            int W, float min_depth, float max_depth, void* stream);
int clslam_disp_grad_pyramid(const float* ddisp_up, const float* const* disp, const float* smooth_aux, int n_smooth,
                             float* const* dz, int batch, int H, int W, void* stream);
int clslam_disp_grad_pyramid_range(const float* ddisp_up, const float* const* disp, const float* smooth_aux, int n_smooth,
                                   float* const* dz, int batch, int H, int W, int scale_lo, int scale_count, void* stream);
int clslam_disp_mean_chunks(void);
int clslam_disp_mean(const float* disp, float* psum, int batch, int hw, void* stream);
typedef struct clslam_loss_desc {
    const float* partial[4];  /* automask partials per scale [B][nblk]                             */
    const float* disp[4];     /* ('disp',s) (B,H>>s,W>>s)                                          */
    const float* rgb0[4];     /* ('rgb',0,s) (B,3,H>>s,W>>s)                                       */
    const float* means[4];    /* clslam_disp_mean outputs [B][chunks]                              */
    const float* pose;        /* (2B,12)                                                           */
    const double* dist0;      /* ('relative_distance',0) (B) float64                               */
    const double* dist1;
    const float* sample_w;    /* (B) loss weights of the local samples (dpp.py:1031-1032)          */
    const float* smooth_w;    /* (n_smooth) weights of the smoothness terms                        */
    float* losses;            /* [18]: 4 x {reprojection, smooth, reg, depth_loss/scale}, velocity, total */
    float* smooth_aux;        /* [4][2+2*n_smooth] kept for clslam_disp_grad                        */
    int32_t batch, nblk, H, W, n_smooth;
    float smooth_scale, vel_scale;
} clslam_loss_desc;
int clslam_loss_finalize(const clslam_loss_desc* desc, void* stream);
/* dpred (2,B,3,H,W) = dL/d(warped images) of one scale.                                          */
int clslam_photo_grad(const unsigned char* sel, const float* coef, const float* pred, const float* target,
                      const float* sample_w, float* dpred, int batch, int H, int W, void* stream);
/* dz (B,h,w) = dL/d(pre-sigmoid disparity logits): transposed bilinear upsample of ddisp_up plus the
 * smoothness gradient (smooth_aux slice of this scale), times sigmoid'.                           */
int clslam_disp_grad(const float* ddisp_up, const float* disp, const float* smooth_aux, int n_smooth, float* dz,
                     int batch, int h, int w, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused Adam over a flat fp32 arena.  Replaces torch.optim.Adam.step() (dpp.py:203,313; torch
 * defaults, same op order as the single-tensor CPU implementation).  grad is multiplied by
 * grad_scale first (1 for single GPU; data-parallel ranks all-reduce with SUM, so it stays 1).
 * guard (device pointer or NULL): when *guard is NaN the launch leaves param and both moments
 * untouched -- the NaN-loss abort of dpp.py:1115-1118 without a host sync between forward and
 * backward (the host checks the loss once per step, after the optimizer launch).               */
int clslam_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                     double beta1, double beta2, double eps, int step, float grad_scale, const float* guard, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hook (bench.py's roofline leg; no reference counterpart).  Between _begin and _end every
 * clslam_conv2d launch issued by THIS host thread carries its own start/stop timestamps (hipExtLaunchKernel):
 * the kernel's execution time, the quantity `rocprofv3 --kernel-trace` reports -- an event pair recorded around
 * a launch also counts the ~3 us dispatch gap.  _end waits for the launches, writes their durations in
 * milliseconds in launch order (at most `capacity`), stores how many there were in *count and disarms.   */
int clslam_conv_profile_begin(int max_launches);
int clslam_conv_profile_end(float* ms, int capacity, int* count);

/* ---------------------------------------------------------------------------------------------
 * Cross-stream hand-off without a marker packet on the PRODUCER's stream (no reference counterpart: the reference
 * has one stream).  The step's dependency chain (decoder forward, data-gradient chain of the backward) releases work to
 * the side streams ~17 times per step; hipEventRecord() puts a barrier packet between two kernels of the chain every
 * time (measured on MI355X / ROCm 7.2, tools/micro/event_gap.hip: +5.0 us per hand-off on the producer), while an event
 * that IS the producing kernel's completion signal (hipExtLaunchKernel stopEvent) costs +1.3 us.
 *   clslam_handoff_arm(ev):   the next clslam_conv2d / clslam_fold_act_grad launch of THIS host thread completes `ev`
 *   clslam_handoff_wait(ev, producer, consumer):  `consumer` stream waits for `ev`; if no launch has taken the armed
 *                             event since clslam_handoff_arm (an op that does not support it, an empty batch) it is
 *                             recorded on `producer` first -- never a missing dependency.
 * Events come from clslam_handoff_event_create (timing disabled) and may be re-armed once waited on.  Arming while another
 * event is still armed (its launch failed before it went out) replaces that event: the stale one was never recorded.     */
void* clslam_handoff_event_create(void);
void clslam_handoff_event_destroy(void* event);
int clslam_handoff_arm(void* event);
int clslam_handoff_wait(void* event, void* producer_stream, void* consumer_stream);

/* ---------------------------------------------------------------------------------------------
 * nitems independent device-to-device copies in one launch (items is a HOST array; sizes in bytes,
 * any alignment).  No reference counterpart: it replaces the per-tensor copies a hipGraph-replayed
 * step needs around the replay -- the sample dict of dpp.py:916-917 into the graph's static inputs and
 * the static output planes into the fresh tensors adapt()/predict() hand back (dpp.py:319).          */
typedef struct clslam_copy_item {
    const void* src;
    void* dst;
    size_t bytes;
} clslam_copy_item;
int clslam_copy_multi(const clslam_copy_item* items, int nitems, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Exact inner-product search = what the reference asks of faiss's
 * index_factory(d, 'Flat', METRIC_INNER_PRODUCT): loop_closure_detection/loop_closure_detection.py:35-36
 * (.add :48, .reconstruct/.search(features, 100) :55-57) and slam/replay_buffer.py:96-98 (.search(f, 1) :110,
 * .search(features, ntotal) :121-122,130).  Rows are L2-normalised by the caller (faiss.normalize_L2).
 * clslam_ip_scores: scores[q][i] = <db[i], queries[q]>, db (n,d) / queries (nq,d) / scores (nq,n) row-major.
 * clslam_topk_desc: the k largest per query in descending order, equal scores by ascending position;
 * out_val/out_idx (nq,k), missing entries -FLT_MAX / -1 like faiss.  n > 4096 needs cand_val/cand_idx of
 * nq * clslam_topk_chunks(n) * k elements (and chunks * k <= 4096).                                  */
int clslam_ip_scores(const float* db, const float* queries, float* scores, int n, int d, int nq, void* stream);
int clslam_topk_chunks(int n);
int clslam_topk_desc(const float* scores, int n, int nq, int k, float* cand_val, int* cand_idx, float* out_val,
                     int* out_idx, void* stream);
/* faiss.normalize_L2 on device rows (loop_closure_detection.py:47, replay_buffer.py:102,215): x_i *= 1/sqrt(<x_i,x_i>),
 * zero rows untouched. */
int clslam_l2_normalize_rows(float* x, int n, int d, void* stream);
/* One candidate of the replay buffer's diversity bookkeeping (slam/replay_buffer.py:104-152, maximize_diversity):
 * db (max_slots,d) vectors and sim (ld,ld) similarity matrix in SLOT order, occupied[max_slots] flags, `scores`
 * = clslam_ip_scores(db, query) over the first nslots slots.  Accepts the candidate when its largest similarity to
 * an occupied slot is < threshold (0 for an empty buffer), writes it to the first free slot (or slot nslots) with
 * its row/column of sim, and when more than `capacity` slots are occupied evicts argmax_j(sum_i sim[i][j] -
 * sim[j][j]) (row/column set to -1, slot freed).  result[5] = {accepted, slot written, slot evicted, occupied
 * count, bits of the nearest similarity}; similarity[0] = the nearest stored similarity as a float.  One workgroup; no host decision between the steps. */
int clslam_diversity_commit(float* db, float* sim, int ld, unsigned char* occupied, int nslots, int max_slots, int d,
                            int capacity, float threshold, const float* query, const float* scores, int* result,
                            float* similarity, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image-pyramid ingest: the reference's datasets resize every pyramid level from the
 * previous one with torchvision.transforms.Resize(LANCZOS) on PIL images (datasets/utils.py:62-66,154-163)
 * = Pillow's ImagingResample 8-bit path, then ToTensor (datasets/utils.py:213-215).  Bit-exact on uint8.
 * clslam_lanczos_plan is a HOST function (double precision + libm, as Pillow): bounds[out][2] = (first tap,
 * tap count), coeffs[out][clslam_lanczos_ksize] 22-bit fixed point.  clslam_resize_pass_u8 runs one
 * separable pass on interleaved uint8 images (B,in_h,in_w,ch<=4) on the device: axis 1 = width -> out_size,
 * axis 0 = height -> out_size (Pillow's order: horizontal first); planar != NULL additionally writes
 * ToTensor(result) as float (B,ch,oh,ow).  clslam_u8_to_planar_f32 is ToTensor alone.                 */
int clslam_lanczos_ksize(int in_size, int out_size);
int clslam_lanczos_plan(int in_size, int out_size, int* bounds, int* coeffs);
int clslam_resize_pass_u8(const unsigned char* src, unsigned char* dst, float* planar, const int* bounds, const int* coeffs,
                          int ksize, int batch, int in_h, int in_w, int ch, int out_size, int axis, void* stream);
int clslam_u8_to_planar_f32(const unsigned char* src, float* planar, int batch, int h, int w, int ch, void* stream);
/* Colour augmentation of the datasets (datasets/utils.py:236-259: torchvision adjust_brightness / _contrast /
 * _saturation / _hue on PIL images, in the order drawn by random.shuffle) with Pillow's exact arithmetic, bit-
 * exact on uint8 RGB (B,h,w,3).  order[n_ops] (0 brightness, 1 contrast, 2 saturation, 3 hue) and factors[4]
 * (doubles as Python draws them, indexed by op id) are HOST arrays; scratch = second image buffer, lsum = batch uint64 device scratch.  */
int clslam_color_jitter_u8(const unsigned char* src, unsigned char* dst, unsigned char* scratch, unsigned long long* lsum,
                           int batch, int h, int w, const int* order, int n_ops, const double* factors, void* stream);

/* The replay buffer's colour jitter (slam/replay_buffer.py:264-265,281-283, enabled by slam/slam.py:98): torchvision's TENSOR code path
 * (transforms/functional_tensor.py 0.11.1: adjust_brightness / _contrast / _saturation / _hue of float images in [0,1]) applied AFTER
 * ToTensor, one drawn transform per replayed sample for its three frames and four scales.  src / dst: n_images planar (3,h,w) float
 * images of one size (dst may be src); params: n_images device records {int32 order[4] (op ids 0 brightness, 1 contrast, 2 saturation,
 * 3 hue in application order, -1 = end); float factor[4] (by op id); float one_minus_factor[4] ((float)(1.0 - factor), formed in
 * double like torch does)} = 48 bytes each; partial: n_images * clslam_color_jitter_f32_blocks(h,w) floats of scratch (the
 * per-image mean gray value of `contrast`, summed in a fixed order).  PARITY UNPINNED against real torchvision (source absent).   */
int clslam_color_jitter_f32_blocks(int h, int w);
int clslam_color_jitter_f32(const float* src, float* dst, const void* params, float* partial, int n_images, int h, int w,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loop-closure feature encoder: MobileNetV3-small forward (loop_closure_detection/encoder.py:13-33:
 * torchvision mobilenet_v3_small cut at 'flatten', ImageNet mean/std normalisation).  The 1x1 convs
 * run through clslam_conv2d (channels zero-padded to multiples of 16); these are the other ops.
 * PARITY UNPINNED: torchvision's source/weights are not available to the build. */
/* normalise + conv3x3 s2 p1 (3->16) + BN + hardswish; img planar (B,3,h,w), weight OIHW (16,3,3,3) */
int clslam_mbv3_stem(const float* img, const float* weight, const float* scale, const float* shift, float* out,
                     int batch, int h, int w, void* stream);
/* depthwise k x k (3|5), stride 1|2, pad k/2, NHWC, weight [k*k][ch], BN scale/shift, activation   */
int clslam_dwconv(const float* x, const float* weight, const float* scale, const float* shift, float* out, int batch,
                  int h, int w, int ch, int ksize, int stride, int act, void* stream);
/* out[b][c] = mean over pixels of x[b][p][c]                                                      */
/* partial (optional): batch * clslam_avgpool_chunks(hw) * ch floats -- two-stage reduction over pixel chunks
 * (deterministic order); NULL = one block per (sample, 64 channels).                                  */
int clslam_avgpool_chunks(int hw);
int clslam_global_avgpool(const float* x, float* out, float* partial, int batch, int hw, int ch, void* stream);
/* squeeze-excitation gates: gate[b][c] = hardsigmoid(w2 * relu(w1 * pool[b] + b1) + b2)            */
int clslam_se_gate(const float* pool, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                   int batch, int ch, int squeeze, void* stream);
/* x[b][p][c] *= gate[b][c] (in place)                                                             */
int clslam_channel_scale(float* x, const float* gate, int batch, int hw, int ch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLSLAM_HIP_H */
