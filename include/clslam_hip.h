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

#define CLSLAM_PAD_ZERO 0
#define CLSLAM_PAD_REFLECT 1

/* Library identification / error text (thread-local). */
int clslam_version(void);
const char* clslam_last_error(void);
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
} clslam_conv_desc;
int clslam_conv2d(const clslam_conv_desc* desc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the trainable convolutions -- replaces autograd's convolution_backward,
 * reflection_pad2d_backward, upsample_nearest2d_backward, cat backward, elu/relu backward for
 * networks/depth_decoder.py:51-71, networks/layers.py:9-48, networks/pose_decoder.py:37-54
 * (entered from dpp.py:312 `losses['loss'].backward()`).                                      */

/* wt[ci][taps-1-t][co] = w[co][t][ci] for ci < ch_in_sel: the weights of the dgrad-as-conv.    */
int clslam_weight_transpose(const float* w, float* wt, int ch_out, int taps, int ch_in, int ch_in_sel,
                            void* stream);
/* dz = act'(yout) * fold(dxp): dxp is the gradient w.r.t. the PADDED conv input
 * [B][h+2*border][w+2*border][ch_stride]; border=1 folds the reflection border back
 * (ReflectionPad2d backward), pool=1 sums each 2x2 block (nearest-2x upsample backward, output
 * is [B][h/2][w/2][ch]); only channels [0,ch) are used (the skip half of a concat is dead:
 * encoders are frozen, dpp.py:308,813-819); yout may be NULL (no activation).                 */
int clslam_fold_act_grad(const float* dxp, const float* yout, float* dz, int batch, int h, int w, int ch,
                         int ch_stride, int border, int pool, int act, void* stream);
/* Weight gradient as an MFMA GEMM reducing over pixels; desc = the forward conv's descriptor. */
int clslam_wgrad_splits(const clslam_conv_desc* desc, int target_blocks);
int clslam_conv_wgrad(const clslam_conv_desc* desc, const float* dz, float* partial, int splits, void* stream);
/* out[i] = scale * sum_s partial[s*n + i] in a fixed order (deterministic).                    */
int clslam_reduce_partials(const float* partial, float* out, size_t n, int splits, float scale, void* stream);
/* bias gradient: column sums of x[rows][ch], stage 1 (follow with clslam_reduce_partials).     */
int clslam_colsum_blocks(int rows);
int clslam_colsum(const float* x, float* partial, int rows, int ch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder stem (frozen): (x-0.45)/0.225 -> conv7x7 s2 p3 -> eval BN -> ReLU, and maxpool 3x3 s2 p1.
 * Replaces networks/resnet_encoder.py:117-121.  img_a/img_b: planar (B,3,h,w) frames exactly as
 * the sample dict holds them (img_b = second frame of the pose pair, dpp.py:951-955, or NULL);
 * weight: (64, 3*num_images, 7, 7) in the checkpoint's OIHW order; out: NHWC (B,h/2,w/2,64).     */
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

#ifdef __cplusplus
}
#endif
#endif /* CLSLAM_HIP_H */
