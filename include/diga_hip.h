/*
 * diga_hip.h -- C ABI of libdiga_hip.so: the MI355X (gfx950) kernels of the DiGA
 * training hot path.
 *
 * The reference (fy-vision/DiGA) has no FFI: every function below replaces a piece of
 * Python/torch code of the reference, cited as file:line under /root/reference with
 *   G5/ = domain_adaptation/GTA5/.
 * The host-side mirror of the reference interface (diga_amd/util/loss.py, util/utils.py,
 * calc_centroids.py, model/model_noaux.py) binds these entry points with ctypes; see
 * INTEGRATION.md for the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name starts with h_;
 *   - tensors are contiguous; image tensors NCHW fp32 unless the name says nhwc;
 *     labels are int64 as the reference's loaders produce them (255 = ignore);
 *   - the caller owns every buffer, including workspaces (size queries below);
 *     kernels never allocate, free or keep pointers past return;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *     no entry point synchronises;
 *   - return 0 on success, a negative DIGA_E* code on a bad argument, or the positive
 *     hipError_t of a failed launch; diga_last_error_string() describes the last failure
 *     of the calling thread.  Nothing throws across the ABI.
 */
#ifndef DIGA_HIP_H
#define DIGA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libdiga_hip.so is built with -fvisibility=hidden: the functions declared in this header are its whole dynamic symbol table */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define DIGA_ABI_VERSION 1

#define DIGA_OK 0
#define DIGA_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define DIGA_EALIGN (-2)   /* pointer not aligned as required             */
#define DIGA_EWORKSPACE (-3) /* workspace too small                       */

#define DIGA_IGNORE_LABEL 255

int diga_version(void);
const char* diga_last_error_string(void);

/* ------------------------------------------------------------------------------------
 * Losses
 * ---------------------------------------------------------------------------------- */

/* Bytes of workspace the loss entry points need for `n_pixels` = N*H*W pixels. */
size_t diga_loss_workspace_bytes(int64_t n_pixels);

/* cross_entropy2d, G5/util/loss.py:48-62.
 *   loss      = sum over pixels with target != 255 of -log_softmax(logits)[target] / (N*H*W)
 *   grad      = d loss / d logits * grad_scale      (nullable: loss only)
 * logits [N,C,H,W], target [N,H,W] int64, loss_out [1], C <= 32. */
int diga_ce2d_fwd_bwd(const float* logits, const int64_t* target, float* grad, float* loss_out,
                      void* workspace, size_t workspace_bytes,
                      int64_t N, int64_t C, int64_t H, int64_t W, float grad_scale, void* stream);

/* OhemCrossEntropy (hard-pixel CE of the Synthia / semi-supervised trees), G5/util/loss.py:65-122
 * (`_ohem_forward` :91-109): over pixels with target != ignore_label, p_t = softmax(logits)[target];
 * kth = the min(min_kept, n_valid-1)-th smallest p_t (0-based); thr = max(kth, thresh);
 *   loss = mean of -log_softmax(logits)[target] over the valid pixels with p_t < thr   (NaN if there are none)
 *   grad = d loss / d logits * grad_scale                                               (nullable: loss only)
 * logits [N,C,H,W] at label resolution, target [N,H,W] int64, loss_out [1].  The order statistic is found by an
 * exact radix select on the device (no sort, no host sync). */
size_t diga_ohem_ce_workspace_bytes(int64_t N, int64_t H, int64_t W);
int diga_ohem_ce_fwd_bwd(const float* logits, const int64_t* target, float* grad, float* loss_out,
                         void* workspace, size_t workspace_bytes, int64_t N, int64_t C, int64_t H, int64_t W,
                         int64_t ignore_label, float thresh, int64_t min_kept, float grad_scale, void* stream);

/* distillation_loss, G5/util/loss.py:125-143.  teacher/student [2B,C,H,W]; the two halves of
 * the batch are the two views.  loss = mean_{B,H,W} sum_c -q0 log p1 + scale * mean sum_c -q1 log p0
 * with q = softmax(teacher), p = softmax(student); grad (nullable) = d loss / d student * grad_scale. */
int diga_distill_fwd_bwd(const float* teacher, const float* student, float* grad, float* loss_out,
                         void* workspace, size_t workspace_bytes,
                         int64_t B2, int64_t C, int64_t H, int64_t W, float scale, float grad_scale,
                         void* stream);

/* x *= *scale_dev, skipped entirely (no traffic) when *scale_dev == 1.  Used by the autograd
 * wrappers to apply the upstream gradient without a host sync. */
int diga_scale_inplace(float* x, const float* scale_dev, int64_t n, void* stream);

/* Loss block of the training step at the LOW-RES boundary: bilinear(align_corners=True)
 * upsampling of student and teacher logits to label size fused with both losses and their
 * backward (G5/train_DiGA_gta2city_warm_up.py:173-176,267-282,299).
 *   stu_lr, tea_lr [2B,C,h,w]; labels [B,H,W] int64
 *   losses_out[0] = cross_entropy2d(up(stu)[:B], labels); losses_out[1] = distillation_loss(up(tea), up(stu), scale)
 *   grad_stu_lr [2B,C,h,w] = d(lambda_seg*ce + lambda_distil*distil)/d stu_lr
 * Deterministic (no float atomics).  Requires h,w >= 2. */
size_t diga_upsample_loss_workspace_bytes(int64_t B2, int64_t C, int64_t h, int64_t w);
int diga_upsample_ce_distill_fwd_bwd(const float* stu_lr, const float* tea_lr, const int64_t* labels,
                                     float* grad_stu_lr, float* losses_out,
                                     void* workspace, size_t workspace_bytes,
                                     int64_t B, int64_t C, int64_t h, int64_t w, int64_t H, int64_t W,
                                     float lambda_seg, float lambda_distil, float scale, void* stream);

/* Same fusion for one cross_entropy2d term only (self-training: CE of the cross-domain mix,
 * G5/train_DiGA_gta2city_self_training.py:343-351).  logits_lr [N,C,h,w], labels [N,H,W]. */
int diga_upsample_ce_fwd_bwd(const float* logits_lr, const int64_t* labels, float* grad_lr,
                             float* loss_out, void* workspace, size_t workspace_bytes,
                             int64_t N, int64_t C, int64_t h, int64_t w, int64_t H, int64_t W,
                             float lambda_seg, void* stream);

/* nn.Upsample(size, 'bilinear', align_corners=True) forward (used for eval / drop-in callers). */
int diga_upsample_bilinear_ac(const float* x, float* y, int64_t NC, int64_t h, int64_t w,
                              int64_t H, int64_t W, void* stream);

/* ------------------------------------------------------------------------------------
 * EMA teacher and SGD (multi-tensor; tables are device arrays prepared by the caller)
 * ---------------------------------------------------------------------------------- */

/* update_teacher_params, G5/util/utils.py:103-116:  t <- alpha*t + (1-alpha)*s  (bit-exact to the
 * two-multiply-one-add fp32 arithmetic of the reference).  one_minus_alpha is passed explicitly
 * because the reference rounds (1 - alpha) in double before casting. */
int diga_ema_update_flat(float* teacher, const float* student, int64_t n, float alpha,
                         float one_minus_alpha, void* stream);

/* Same over a list of tensors.  Chunk c covers elements [chunk_start[c], +chunk_elems) of tensor
 * chunk_tensor[c]; sizes[i] = numel of tensor i. */
int diga_ema_update_multi(float* const* teacher_ptrs, const float* const* student_ptrs,
                          const int64_t* sizes, const int32_t* chunk_tensor, const int64_t* chunk_start,
                          int64_t n_chunks, int64_t chunk_elems, float alpha, float one_minus_alpha,
                          void* stream);

/* torch.optim.SGD(momentum, weight_decay) as the reference drives it (G5/model/model_noaux.py:48-77,
 * G5/train_DiGA_gta2city_warm_up.py:156,301-305): tensor i occurs mult[i] times in its group, so one
 * step() is mult[i] sequential micro-steps  d = g + wd*p; buf = first_step ? d : momentum*buf + d;
 * p -= lr[i]*buf  fused in registers.  lr[i] is a device array (per tensor; poly-LR is written there
 * by the host each step).  skip_flag (nullable, device int32): when skip_flag[0] != 0 the launch changes nothing -- the
 * "found_inf" of a loss-scaled fp16 backward (the MiT student, diga_mit.h), decided on the device without a host sync. */
int diga_sgd_momentum_multi(float* const* param_ptrs, const float* const* grad_ptrs, float* const* buf_ptrs,
                            const int64_t* sizes, const int32_t* mult, const float* lr,
                            const int32_t* chunk_tensor, const int64_t* chunk_start,
                            int64_t n_chunks, int64_t chunk_elems,
                            float momentum, float weight_decay, int first_step, float grad_scale,
                            const int32_t* skip_flag, void* stream);
/* flag[0] = 1 and flag[1] += 1 when the n floats at x hold an inf or a NaN (flag: two device int32, zeroed by the caller at
 * the start of a step; [1] counts the overflowed steps).  The reference is fp32 throughout and has no such failure mode
 * (torch.cuda.amp.GradScaler is what a mixed-precision port of it would use: same found_inf / skipped-step semantics). */
int diga_nonfinite_flag_f32(const float* x, int64_t n, int32_t* flag, void* stream);

/* ------------------------------------------------------------------------------------
 * ClassMix (G5/train_DiGA_gta2city_warm_up.py:240-259, ..._self_training.py:259-275,306-325)
 * ---------------------------------------------------------------------------------- */

/* hist[b][v] = number of pixels of image b with label v, v in 0..255 (other values are not
 * counted).  hist [B,256] uint32 must be zeroed by the caller (hipMemsetAsync). */
int diga_label_hist256(const int64_t* labels, uint32_t* hist, int64_t B, int64_t HW, void* stream);

/* out[b,c,y,x] = lut[b][label[b,y,x]] ? fg : bg.  lut [B,256] uint8; bg, fg, out [B,CH,H*W].
 * If labels_out != NULL also labels_out = lut ? labels : bg_labels (label paste of the
 * self-training mix).  Labels outside 0..255 select bg. */
int diga_classmix_paste(const float* bg, const float* fg, const int64_t* labels, const uint8_t* lut,
                        float* out, const int64_t* bg_labels, int64_t* labels_out,
                        int64_t B, int64_t CH, int64_t HW, void* stream);

/* ------------------------------------------------------------------------------------
 * Centroid pseudo-labeler (G5/calc_centroids.py:120-176, ..._self_training.py:298-341)
 * ---------------------------------------------------------------------------------- */

/* Class_Features.get_centroid_weight: w[n,k,p] = softmax_k( -|| centroids[k] - feat[n,:,p] ||_2 ).
 * feat [N,D,HW], centroids [K,D], weights [N,K,HW]; K <= 32.  neg_dist (nullable) receives -d. */
int diga_centroid_softmax_weights(const float* feat, const float* centroids, float* weights,
                                  float* neg_dist, int64_t N, int64_t D, int64_t K, int64_t HW,
                                  void* stream);

/* Bilateral consensus: argmax_k of the bilinear(align_corners) upsampling of weights to [H,W],
 * pseudo_out = pseudo_in where it equals that argmax else 255; feat_pseudo (nullable) = argmax. */
int diga_upsample_argmax_consensus(const float* weights, const int64_t* pseudo_in, int64_t* pseudo_out,
                                   int64_t* feat_pseudo, int64_t N, int64_t K, int64_t h, int64_t w,
                                   int64_t H, int64_t W, void* stream);

/* Class_Features.calculate_mean_vector.  ids[n,p] = argmax_k out[n,k,p], kept only where it equals
 * the label (labels given) and < K.  Labels come either already at feature resolution
 * (labels_lr [N,hw] float, as the reference passes them) or at full resolution (labels_full
 * [N,H,W] int64, nearest-downsampled here with src = floor(dst*in/out)); both NULL = no labels.
 * sums [N,K,D] = sum of feat over member pixels, counts [N,K] (int32). */
size_t diga_class_mean_workspace_bytes(int64_t N, int64_t hw);
int diga_class_mean_vectors(const float* feat, const float* out, const float* labels_lr,
                            const int64_t* labels_full, float* sums, int32_t* counts,
                            void* workspace, size_t workspace_bytes,
                            int64_t N, int64_t D, int64_t K, int64_t h, int64_t w, int64_t H, int64_t W,
                            void* stream);

/* update_objective_SingleVector applied sequentially in image-major / class-minor order:
 * for n, for t: skip if counts<min_pixels or sum(v)==0; v = sums/counts;
 *   mode 0 (moving average): c_t = c_t*(1-m) + m*v ; num_t = min(num_t+1, 3000)
 *   mode 1 (mean):           c_t = (c_t*num_t + v)/(num_t+1) ; num_t = min(num_t+1, 3000)
 * centroids [K,D], nums [K] float. */
int diga_centroid_ema_apply(float* centroids, float* nums, const float* sums, const int32_t* counts,
                            int64_t N, int64_t K, int64_t D, int64_t hw, float momentum, int min_pixels,
                            int mode, void* stream);

/* ------------------------------------------------------------------------------------
 * Metrics (G5/util/metrics.py:32-44)
 * ---------------------------------------------------------------------------------- */

/* hist[gt*K+pred] += 1 for pixels with 0 <= gt < K.  hist [K*K] int64, accumulated into. */
int diga_confusion_matrix(const int64_t* gt, const int64_t* pred, int64_t* hist, int64_t n, int64_t K,
                          void* stream);

/* Two-scale evaluation (G5/train_DiGA_gta2city_warm_up.py:346-359, G5/evaluate_val.py:73-93): per label pixel,
 * argmax_k max(up(pred_a)[k], up(pred_b)[k]) with up = bilinear align_corners upsampling to [H,W]; pred_out
 * (nullable) receives it, hist (nullable, [K*K] int64, accumulated) the confusion against gt (nullable). */
int diga_two_scale_confusion(const float* pred_a, int64_t ha, int64_t wa, const float* pred_b, int64_t hb, int64_t wb,
                             const int64_t* gt, int64_t* pred_out, int64_t* hist, int64_t N, int64_t K, int64_t H,
                             int64_t W, void* stream);

/* ------------------------------------------------------------------------------------
 * Convolution on the fp32 matrix cores (nn.Conv2d layers of G5/model/seg_model_noaux.py:57-101,
 * :140-172; cuDNN in the reference).  Activations NHWC fp32 with a leading dimension (`*_ld` floats
 * between pixels, so channel slices of wider tensors can be read/written in place); weights
 * [Cout][R][S][Cin] = the channels_last image of torch's [Cout,Cin,R,S].
 * ---------------------------------------------------------------------------------- */

/* out[n,ho,wo,k] = bias[k] + sum_{r,s,c} in[n, ho*stride_y + off_y0 + r*off_dy, wo*stride_x + off_x0 + s*off_dx, c]
 *                                        * wgt[k][r][s][c]          (taps outside the image read zero)
 * Forward conv: off_*0 = -padding, off_d* = dilation.  Backward-data of a stride-1 conv: in = dy,
 * wgt = the [Cin][R][S][Cout] transpose (diga_weight_transpose), off_*0 = +padding, off_d* = -dilation.
 * Cin % 32 == 0, in_ld % 4 == 0; bias nullable.  prof_tag: DIGA_PROF_CONV_FWD or DIGA_PROF_CONV_BWD_DATA. */
int diga_conv2d_nhwc_f32(const float* in, const float* wgt, const float* bias, float* out,
                         int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld,
                         int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                         int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                         int64_t off_dy, int64_t off_dx, float* stats_partial, int prof_tag, void* stream);

/* stats_partial (nullable, diga_conv2d_stats_floats(N,Ho,Wo,Cout) floats): the epilogue also writes, per 128-pixel
 * tile and channel, sum(y - s), sum((y - s)^2), s -- the column-statistics partials of the BatchNorm that follows the
 * conv (diga_bn_fwd_partials), which then needs no statistics pass of its own over y. */
size_t diga_conv2d_stats_floats(int64_t N, int64_t Ho, int64_t Wo, int64_t Cout);
/* Rows per statistics chunk the forward entry point will use for this layer: 128, or 64 where an fp32 stride-1 pointwise layer
 * runs on the persistent GEMM (each wave reduces its own 64 rows in registers).  Pass it to diga_bn_fwd_partials as chunk_rows;
 * diga_conv2d_stats_floats sizes the buffer for either. */
int diga_conv2d_stats_chunk_rows(int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t R,
                                 int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int math);
/* ... and of the `partials` records a backward-data epilogue (`_epi` entry points) will write for this geometry. */
int diga_conv2d_epi_chunk_rows(int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t R,
                               int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int math);

/* ------------------------------------------------------------------------------------
 * Backward-data with a fused epilogue.  The gradient a backward-data convolution produces is the input gradient of the
 * layer in front of it -- in the DeepLab trunk always a train-mode BatchNorm (+ReLU, + the residual junction of
 * G5/model/seg_model_noaux.py:81-101).  The `_epi` forms of the three convolution entry points finish that tensor on its
 * way out of the matrix cores instead of in separate passes over HBM:
 *     out = acc                               the convolution (bias-free, no forward statistics)
 *     out += addend                           the gradient reaching the same tensor through the residual branch
 *     out  = out where (mask_y > 0)           ReLU of a BatchNorm with residual (mask_y = its output)          -- or --
 *     out  = out where bit c of mask_bits     the same mask as one bit per channel (relu_bits of diga_bn_fwd*) -- or --
 *     out  = out where fma(x, a, b) > 0       ReLU of a BatchNorm without residual (relu_ab = its forward coefficients)
 *     partials[chunk] = { sum out, sum out*xhat } per 128-row chunk and channel, xhat = (x - mean) * invstd
 * `out` is then the MASKED gradient g of that BatchNorm and `partials` what diga_bn_bwd_partials finalises -- its
 * reduce pass over (g, x), the separate add and the residual-gradient copy are gone.  All tensors [M][ld] fp32 with
 * M = N*Ho*Wo rows, ld % 4 == 0, 16-byte aligned; Cout % 4 == 0; partials holds ceil(M/64)*2*Cout floats: one {sum, sum*xhat}
 * record per chunk of diga_conv2d_epi_chunk_rows(...) rows (128; 64 on the persistent fp32 GEMM) -- pass that chunk size
 * to diga_bn_bwd_partials.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    const float* addend;   /* nullable */
    int64_t addend_ld;
    const float* mask_y;   /* nullable */
    int64_t mask_ld;
    const float* x;        /* nullable (needed by relu_ab and partials) */
    int64_t x_ld;
    const float* relu_ab;  /* nullable [2][Cout] */
    const float* mean;     /* [Cout] (with partials) */
    const float* invstd;   /* [Cout] (with partials) */
    float* partials;       /* nullable */
    const unsigned char* mask_bits;   /* nullable: [M][mask_bits_ld bytes], bit (c & 7) of byte (c >> 3) = keep channel c */
    int64_t mask_bits_ld;
} diga_bwd_epilogue_t;

int diga_conv2d_nhwc_f32_epi(const float* in, const float* wgt, float* out, int64_t N, int64_t Hi, int64_t Wi, int64_t Cin,
                             int64_t in_ld, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                             int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy,
                             int64_t off_dx, const diga_bwd_epilogue_t* epi, int prof_tag, void* stream);
int diga_conv2d_nhwc_bf16x3_epi(const float* in, const uint16_t* wgt_hi, const uint16_t* wgt_lo, float* out, int64_t N,
                                int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo, int64_t Cout,
                                int64_t out_ld, int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0,
                                int64_t off_x0, int64_t off_dy, int64_t off_dx, const diga_bwd_epilogue_t* epi, int prof_tag,
                                void* stream);
int diga_conv2d_nhwc_twin_epi(const void* in_twin, const void* wgt_img, float* out, int64_t N, int64_t Hi, int64_t Wi,
                              int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                              int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy,
                              int64_t off_dx, const diga_bwd_epilogue_t* epi, int prof_tag, void* stream);

/* Options of ONE forward convolution call (the `_opts` entry points below) -- the non-conv ops around the translator's
 * convolutions (G5/model/model_util.py:21-61: ReflectionPad2d -> Conv2d -> [norm] -> [activation];
 * G5/model/model_noaux.py:100-117: nn.Upsample(scale_factor=2) in front of a block) folded into the kernel's addressing
 * instead of separate passes over HBM:
 *   reflect_pad     1: taps outside the image read the mirrored pixel (index -i -> i, H-1+i -> H-1-i) instead of zero
 *   upsample_shift  s: the conv reads the 2^s nearest-neighbour upsampling of `in` ([N,Hi,Wi,Cin] stays the SOURCE tensor;
 *                      Ho/Wo and the tap offsets are those of the upsampled image)
 *   activation      1: tanh on the (biased) output
 * Passed explicitly with the call; the library keeps no per-thread or per-process mode.  Not combinable with BatchNorm
 * statistics output or a backward epilogue. */
typedef struct diga_conv_options {
    int reflect_pad;
    int upsample_shift;
    int activation;
} diga_conv_options_t;

/* Conv arithmetic.  The exact-fp32 and the split-bf16 kernels are SEPARATE entry points (diga_conv2d_nhwc_f32 vs
 * diga_conv2d_nhwc_bf16x3 / _twin); the one entry point that serves both, diga_conv2d_wgrad_nhwc_f32, takes the arithmetic
 * as an argument.  Which one a model runs in is host-side policy (diga_amd/_lib.py: set_conv_math, default from the
 * environment variable DIGA_CONV_MATH) -- the library holds no process-wide mode.
 *   DIGA_CONV_MATH_F32    v_mfma_f32_32x32x2_f32, exact fp32 (k-ordered fmaf chain)
 *   DIGA_CONV_MATH_BF16X3 operands split into bf16 hi+lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
 *                         accumulate (~1e-5 relative per product) */
#define DIGA_CONV_MATH_F32 0
#define DIGA_CONV_MATH_BF16X3 1

/* Tuned split-bf16 forward / backward-data: same contract as diga_conv2d_nhwc_f32 in DIGA_CONV_MATH_BF16X3, but the
 * weights are passed already split (diga_split_bf16 of the [Cout][R][S][Cin] array, once per step), so the kernel
 * stages them without arithmetic.  hi/lo are bf16 bit patterns; n % 4 == 0. */
int diga_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, void* stream);
int diga_conv2d_nhwc_bf16x3(const float* in, const uint16_t* wgt_hi, const uint16_t* wgt_lo, const float* bias, float* out,
                            int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld,
                            int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                            int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                            int64_t off_dy, int64_t off_dx, float* stats_partial, int prof_tag, void* stream);

/* The same convolution with BOTH operands pre-split and staged global -> LDS by LDS-DMA loads (no registers, no split
 * arithmetic, no ds_write in the kernel; three-stage LDS ring, one barrier per K-step; multi-tap layers run with two
 * MFMA waves per SIMD and skip the K-steps of taps that lie outside the image for a whole tile):
 *   in_twin  = diga_make_twin of the [N,Hi,Wi,Cin] fp32 activations: per pixel and group of 8 channels 16 B of bf16 hi
 *              + 16 B of bf16 lo (4*Cin bytes per pixel, dense);
 *   wgt_img  = diga_split_bf16_image of the [Cout][R][S][Cin] weights (diga_split_bf16_image_bytes bytes): the hi / lo
 *              LDS images of every (output-channel tile, K-step).
 * Cin % 32 == 0.  Pays when the activation tensor is read by many tiles (3x3 taps, wide Cout, several convs on one
 * input): the twin costs one extra pass (read 4 B + write 4 B per element). */
int diga_make_twin(const float* x, int64_t ld, void* twin, int64_t M, int64_t C, void* stream);
size_t diga_split_bf16_image_bytes(int64_t Cout, int64_t RS, int64_t Cin);
int diga_split_bf16_image(const float* w, void* img, int64_t Cout, int64_t RS, int64_t Cin, void* stream);
int diga_conv2d_nhwc_twin(const void* in_twin, const void* wgt_img, const float* bias, float* out, int64_t N, int64_t Hi,
                          int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                          int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx,
                          float* stats_partial, int prof_tag, void* stream);

/* Winograd in fp32 for a stride-1 3x3 convolution with padding = dilation (output H x W = input H x W): same result as
 * diga_conv2d_nhwc_f32 with R = S = 3, offsets (-dilation, +dilation) up to fp32 rounding.
 *   tile = 2: F(2x2,3x3), 16 instead of 36 multiplications per 2x2 outputs, transforms with coefficients 0, +-1, +-1/2
 *             (as accurate as the direct fmaf chain);
 *   tile = 4: F(4x4,3x3), 36 instead of 144 multiplications per 4x4 outputs, points 0, +-1, +-2, inf (rounding error about
 *             10x the direct kernel's: max 1e-5 of the output scale on a 256-channel layer) -- the choice for the
 *             dilation 1 / 2 / 4 layers, whose d*d sub-images are large against a 4x4 tile.
 * The caller picks `tile` per call (every entry point of one layer -- workspace queries, forward, kept V, weight gradient --
 * must be given the same value); the library keeps no mode.  flip = 1 reads the taps in reverse order
 * (w[2-r][2-s]): with wgt = the [Cin][3][3][Cout] transpose this is the backward-data convolution.  No statistics / backward
 * epilogue.  Cin % 32 == 0, Cout % 4 == 0, Cout > 64.  Replaces the cuDNN call behind nn.Conv2d(3x3, dilation = padding) of
 * G5/model/seg_model_noaux.py:66-70,143-150,166-170.  workspace: diga_conv2d_winograd_workspace_bytes (16-byte aligned). */
size_t diga_conv2d_winograd_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t dilation, int64_t tile);
/* Round 5, two optional arguments of every Winograd entry point:
 *   tile_table (nullable): the layer's tile table -- a function of (N, H, W, dilation, tile) only -- built ONCE by
 *     diga_conv2d_winograd_tile_table into a caller-owned buffer of diga_conv2d_winograd_tile_table_bytes bytes and reused by every
 *     call on that geometry (forward, backward-data, backward-weight of all layers that share it); NULL: the call builds it in its
 *     workspace, one more launch (136 per C2 step before).
 *   stats_partial (nullable; forward of 4x4 / 6x6 tiles only): diga_conv2d_winograd_stats_floats floats; the output transform then
 *     also leaves the BatchNorm behind the layer its column statistics as diga_conv2d_winograd_stats_records RECORDS of unequal
 *     size -- [R][3][Cout] {sum (y - s), sum (y - s)^2, s} followed by [R] pixel counts -- for diga_bn_fwd_records
 *     (counts = stats_partial + R * 3 * Cout): no statistics pass over y. */
size_t diga_conv2d_winograd_tile_table_bytes(int64_t N, int64_t H, int64_t W, int64_t dilation, int64_t tile);
int diga_conv2d_winograd_tile_table(void* table, int64_t N, int64_t H, int64_t W, int64_t dilation, int64_t tile, void* stream);
size_t diga_conv2d_winograd_stats_records(int64_t N, int64_t H, int64_t W, int64_t Cout, int64_t dilation, int64_t tile);
size_t diga_conv2d_winograd_stats_floats(int64_t N, int64_t H, int64_t W, int64_t Cout, int64_t dilation, int64_t tile);
int diga_conv2d_winograd_f32(const float* in, const float* wgt, const float* bias, float* out, void* workspace,
                             size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld, int64_t Cout,
                             int64_t out_ld, int64_t dilation, int64_t tile, int flip, float* stats_partial, const void* tile_table,
                             int prof_tag, void* stream);
/* Forward with a diga_conv_options_t (inference-only, as the `_opts` forms of the direct kernels): reflect_pad folds
 * nn.ReflectionPad2d(dilation) into the input transform's tap addressing (the translator's 3x3 ResBlock convs,
 * G5/model/model_util.py:21-61: 54 % of its FLOPs); upsample_shift / activation must be 0 here (direct kernels).  tile = 4 or 6. */
int diga_conv2d_winograd_f32_opts(const float* in, const float* wgt, const float* bias, float* out, void* workspace,
                                  size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld, int64_t Cout,
                                  int64_t out_ld, int64_t dilation, int64_t tile, const diga_conv_options_t* opts, const void* tile_table,
                                  int prof_tag, void* stream);
/* ... with the backward-data epilogue of diga_bwd_epilogue_t (declared above; same arithmetic per element as the `_epi` forms
 * of the direct kernels; `partials` rows are filled per group of tiles instead of per 128 pixel rows -- the finaliser
 * diga_bn_bwd_partials only adds the rows up). */
int diga_conv2d_winograd_f32_epi(const float* in, const float* wgt, float* out, void* workspace, size_t workspace_bytes, int64_t N,
                                 int64_t H, int64_t W, int64_t Cin, int64_t in_ld, int64_t Cout, int64_t out_ld, int64_t dilation,
                                 int64_t tile, int flip, const diga_bwd_epilogue_t* epi, const void* tile_table, int prof_tag,
                                 void* stream);
/* Backward-weight of the same layer through Winograd (dw [Cout][3][3][Cin] = G^T [sum over tiles (A dY A^T) (.) (B^T d B)] G):
 * transforms of dy and x, the 16 (tile 2) / 36 (tile 4) products contracted over the tiles in one launch of the fp32 LDS-DMA
 * backward-weight kernel (fixed-order split-K: bit-reproducible), the transform back to 3x3.  Cout % 256 == 0, Cin % 128 == 0. */
size_t diga_conv2d_wgrad_winograd_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t dilation,
                                                  int64_t tile, int v_kept /* 1: the call will pass the forward's kept V */);
int diga_conv2d_wgrad_winograd_f32(const float* dy, const float* x, const float* v_kept, float* dw, void* workspace,
                                   size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t x_ld, int64_t Cout,
                                   int64_t dy_ld, int64_t dilation, int64_t tile, const void* tile_table, void* stream);
/* Forward that leaves its transformed input V (diga_conv2d_winograd_v_floats floats, 4x the input tensor) in `v_keep` for the
 * weight gradient of the same layer: pass it as `v_kept` above (x may then be null) and the backward skips the input transform
 * -- HBM is 288 GB: the transform is a bandwidth pass of 5x the input's bytes per 3x3 layer. */
size_t diga_conv2d_winograd_v_floats(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t dilation, int64_t tile);
int diga_conv2d_winograd_f32_keep(const float* in, const float* wgt, const float* bias, float* out, float* v_keep, void* workspace,
                                  size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld, int64_t Cout,
                                  int64_t out_ld, int64_t dilation, int64_t tile, float* stats_partial, const void* tile_table, int prof_tag,
                                  void* stream);

/* diga_conv2d_nhwc_f32 / _bf16x3 / _twin with a diga_conv_options_t (non-null; inference-only: no statistics output). */
int diga_conv2d_nhwc_f32_opts(const float* in, const float* wgt, const float* bias, float* out, int64_t N, int64_t Hi, int64_t Wi,
                              int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                              int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx,
                              const diga_conv_options_t* opts, int prof_tag, void* stream);
int diga_conv2d_nhwc_bf16x3_opts(const float* in, const uint16_t* wgt_hi, const uint16_t* wgt_lo, const float* bias, float* out,
                                 int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo, int64_t Cout,
                                 int64_t out_ld, int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0,
                                 int64_t off_x0, int64_t off_dy, int64_t off_dx, const diga_conv_options_t* opts, int prof_tag,
                                 void* stream);
int diga_conv2d_nhwc_twin_opts(const void* in_twin, const void* wgt_img, const float* bias, float* out, int64_t N, int64_t Hi,
                               int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                               int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx,
                               const diga_conv_options_t* opts, int prof_tag, void* stream);

/* dw[k][r][s][c] = sum_{n,ho,wo} dy[n,ho,wo,k] * x[n, ho*stride_y + off_y0 + r*off_dy, wo*stride_x + off_x0 + s*off_dx, c]
 * Split over pixel ranges into fp32 slabs in `workspace`, summed in fixed order (deterministic).
 * math = DIGA_CONV_MATH_F32 (exact fp32 MFMA) or DIGA_CONV_MATH_BF16X3 (operands split in the kernel).  Cin % 4 == 0,
 * Cout % 4 == 0. */
size_t diga_conv2d_wgrad_workspace_bytes(int64_t N, int64_t Ho, int64_t Wo, int64_t Cout, int64_t Cin,
                                         int64_t R, int64_t S);
int diga_conv2d_wgrad_nhwc_f32(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes,
                               int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t x_ld,
                               int64_t Ho, int64_t Wo, int64_t Cout, int64_t dy_ld, int64_t R, int64_t S,
                               int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                               int64_t off_dy, int64_t off_dx, int math, void* stream);

/* Backward-weight of the same convolution on the split twins (diga_make_twin of dy [N,Ho,Wo,Cout] and of x
 * [N,Hi,Wi,Cin], both dense): operands staged by LDS-DMA, fragments read with transposing LDS reads.  dw [Cout][R][S][Cin].
 * Pays where the twins exist anyway (multi-tap layers: x twin from the forward pass, dy twin from backward-data). */
size_t diga_conv2d_wgrad_twin_workspace_bytes(int64_t N, int64_t Ho, int64_t Wo, int64_t Cout, int64_t Cin, int64_t R, int64_t S);
int diga_conv2d_wgrad_twin(const void* dy_twin, const void* x_twin, float* dw, void* workspace, size_t workspace_bytes,
                           int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t R,
                           int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy,
                           int64_t off_dx, void* stream);

/* Stem (7x7/2 on the 3-channel NCHW image, G5/model/seg_model_noaux.py:221): out[n,ho,wo][(r*S+s)*C + c] =
 * x[n,c,ho*stride-pad+r,wo*stride-pad+s] (zero outside / beyond R*S*C up to Kpad), after which the conv is a
 * 1x1 conv with Cin = Kpad on the kernels above. */
int diga_im2col_nchw(const float* x, float* out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t R, int64_t S,
                     int64_t stride, int64_t pad, int64_t Ho, int64_t Wo, int64_t Kpad, void* stream);

/* w [K][RS][C] -> wt [C][RS][K] (weights for backward-data). */
int diga_weight_transpose(const float* w, float* wt, int64_t K, int64_t RS, int64_t C, void* stream);

/* ------------------------------------------------------------------------------------
 * Normalisation / pooling on NHWC fp32 (nn.BatchNorm2d with frozen affine in train mode, nn.GroupNorm(32),
 * SEBlock, nn.Dropout2d, nn.MaxPool2d(3,2,1,ceil_mode) of G5/model/seg_model_noaux.py:57-101,122-137,
 * 140-214,216-232).  x/y/... are [rows][C] with `ld_*` floats between rows; C % 4 == 0.
 * ---------------------------------------------------------------------------------- */
size_t diga_norm_workspace_bytes(int64_t rows_per_segment, int64_t n_segments, int64_t C);

/* y = [relu]( fma(x, a, b) [+ residual] ), a = invstd*gamma, b = beta - mean*a.  training: batch statistics over the
 * M rows (biased variance), running_mean/var updated with `momentum` (unbiased variance), statistics saved for the
 * backward pass; eval: running statistics.  residual nullable.  save_ab (nullable, [2][C]) receives a and b: passed
 * back to diga_bn_bwd as `relu_ab` it lets the backward of a residual-free BN+ReLU re-derive the ReLU mask from x
 * instead of reading y.  y_twin != 0: y (dense, ld_y == C, C % 8 == 0) receives the split twin of the result (the
 * format of diga_make_twin, 4 bytes per element) instead of fp32 -- for a tensor read only by the twin conv kernels.
 * relu_bits (nullable; relu, C % 32 == 0): also receives the ReLU mask as [M][C/8] bytes, bit (c & 7) of byte (c >> 3) =
 * (y[c] > 0) -- `mask_bits` of diga_bwd_epilogue_t: the backward-data epilogue of the conv that consumes y reads 1 bit
 * per element instead of y itself. */
int diga_bn_fwd(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* residual, int64_t ld_r,
                const float* gamma, const float* beta, float* running_mean, float* running_var,
                float* save_mean, float* save_invstd, float* save_ab, int64_t M, int64_t C, int training, int relu,
                int y_twin, unsigned char* relu_bits, float momentum, float eps, void* workspace, size_t workspace_bytes,
                void* stream);

/* Train-mode diga_bn_fwd whose statistics pass is replaced by partials the producing conv already wrote
 * (`partial` = stats_partial of diga_conv2d_nhwc_*, `chunk_rows` = 128). */
int diga_bn_fwd_partials(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* residual, int64_t ld_r,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float* save_mean, float* save_invstd, float* save_ab, int64_t M, int64_t C, int relu,
                         int y_twin, unsigned char* relu_bits, float momentum, float eps, const float* partial,
                         int64_t chunk_rows, void* workspace, size_t workspace_bytes, void* stream);

/* ... replaced by RECORDS of unequal size: partial [n_records][3][C] {sum (x - s), sum (x - s)^2, s}, counts [n_records] = the rows
 * behind each record (0 allowed), sum of counts = M -- what the Winograd forward's output transform writes (stats_partial of
 * diga_conv2d_winograd_f32: its tile groups hold different numbers of in-image pixels).  workspace as diga_bn_fwd_partials + 96 floats. */
int diga_bn_fwd_records(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* residual, int64_t ld_r,
                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float* save_mean, float* save_invstd, float* save_ab, int64_t M, int64_t C, int relu,
                        int y_twin, unsigned char* relu_bits, float momentum, float eps, const float* partial,
                        const float* counts, int64_t n_records, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the above wrt x (gamma/beta are frozen on this path): g = dy*mask, mask = [y>0] when y is given,
 * [fma(x, a, b) > 0] when relu_ab = save_ab of the forward is given instead (BN without residual), 1 when both are
 * null (no ReLU); dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dres (nullable) = g.  dx_twin != 0: dx (dense,
 * C % 8 == 0) receives the split twin instead of fp32 (the conv before this BN runs backward on the twin kernels). */
int diga_bn_bwd(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                const float* relu_ab, const float* gamma, const float* save_mean, const float* save_invstd, float* dx,
                int64_t ld_dx, float* dres, int64_t ld_dr, int64_t M, int64_t C, int training, int dx_twin, void* workspace,
                size_t workspace_bytes, void* stream);

/* diga_bn_bwd for a BatchNorm whose affine pair TRAINS -- the `linear_fuse` ConvModule of the SegFormer decode head,
 * G5/model/networks/segformer_head.py:63-68 (norm_cfg BN, requires_grad=True): additionally dgamma[c] = sum g*xhat and
 * dbeta[c] = sum g (in train mode the two column sums the input gradient needs anyway: no extra pass).  training = 0 (the module in
 * eval(): BN-frozen fine-tuning, test-time adaptation): save_mean / save_invstd are the RUNNING statistics the forward used,
 * dx = gamma*invstd*g without the mean terms, dgamma / dbeta as above -- what nn.BatchNorm2d returns in eval mode. */
int diga_bn_bwd_affine(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                       const float* relu_ab, const float* gamma, const float* save_mean, const float* save_invstd, float* dx,
                       int64_t ld_dx, float* dgamma, float* dbeta, int64_t M, int64_t C, int training, void* workspace,
                       size_t workspace_bytes, void* stream);

/* diga_bn_bwd for a gradient that arrives already masked and reduced: `g` and `partial` ([ceil(M/chunk_rows)][2][C]:
 * sum g, sum g*xhat per chunk) come out of the epilogue of the backward-data convolution that produced g
 * (diga_conv2d_nhwc_*_epi, chunk_rows = 128): one finalise launch + the apply pass (read g, x; write dx).
 * workspace >= 67*C floats. */
int diga_bn_bwd_partials(const float* g, int64_t ld_g, const float* x, int64_t ld_x, const float* gamma,
                         const float* save_mean, const float* save_invstd, float* dx, int64_t ld_dx, int64_t M, int64_t C,
                         int dx_twin, const float* partial, int64_t chunk_rows, void* workspace, size_t workspace_bytes,
                         void* stream);

/* GroupNorm over (HW x C/G) per image and group, then y = [relu](chan_scale[n,c] * (xhat*gamma + beta));
 * chan_scale (nullable, [N][C]) carries the Dropout2d keep/(1-p) pattern.  save_mean/invstd [N][G]. */
int diga_gn_fwd(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* gamma, const float* beta,
                const float* chan_scale, float* save_mean, float* save_invstd, int64_t N, int64_t HW, int64_t C,
                int64_t G, int relu, float eps, void* workspace, size_t workspace_bytes, void* stream);
int diga_gn_bwd(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                const float* gamma, const float* chan_scale, const float* save_mean, const float* save_invstd,
                float* dx, int64_t ld_dx, float* dgamma, float* dbeta, int64_t N, int64_t HW, int64_t C, int64_t G,
                void* workspace, size_t workspace_bytes, void* stream);

/* SE block pieces: out[n][c] = mean_hw x;  y = x*a[n][c] + b[n][c] (b nullable);  out[n][c] = sum_hw dy*x. */
int diga_avgpool_nhwc(const float* x, int64_t ld_x, float* out, int64_t N, int64_t HW, int64_t C, void* workspace,
                      size_t workspace_bytes, void* stream);
int diga_channel_affine(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* a, const float* b,
                        int64_t N, int64_t HW, int64_t C, void* stream);
/* The dense layers of the SE block (G5/model/seg_model_noaux.py:122-137: nn.Linear(1280, 80) -> ReLU -> nn.Linear(80, 1280) -> Sigmoid
 * on the pooled [N, 1280] vector; torch runs them through a GEMM library):  y[N][O] = act(x[N][K] . W[O][K]^T + b),
 * act 0 = none, 1 = ReLU, 2 = sigmoid.  Backward from the ACTIVATED output y: dz = dy * act'(y) (workspace dz [N][O], caller-owned),
 * db[O] = sum_n dz, dW[O][K] = dz^T x, dx[N][K] = dz W (db / dW / dx nullable). */
int diga_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int64_t N, int64_t K, int64_t O, int act, void* stream);
int diga_small_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dz, float* dx, float* dw, float* db,
                          int64_t N, int64_t K, int64_t O, int act, void* stream);
/* Bias gradient of a convolution with bias (the ASPP branches / bottleneck, G5/model/seg_model_noaux.py:143-170; torch computes
 * grad_output.sum((0, 2, 3))): out[c] = sum over the M rows of x [M][ld_x] (channels contiguous), shifted sums merged in double.
 * C % 4 == 0 and ld_x % 4 == 0, or any C <= 64 (the 19-class prediction conv of the SegFormer head, segformer_head.py:70).
 * workspace: diga_norm_workspace_bytes(M, 1, C). */
int diga_colsum_nhwc(const float* x, int64_t ld_x, float* out, int64_t M, int64_t C, void* workspace, size_t workspace_bytes,
                     void* stream);
int diga_channel_dot(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, float* out, int64_t N, int64_t HW,
                     int64_t C, void* workspace, size_t workspace_bytes, void* stream);

/* Multi-scale fusion of the SegFormer decode head (G5/model/networks/segformer_head.py:145-159: every embedded stage output is
 * resized to the 1/4-scale grid with F.interpolate(mode='bilinear', align_corners=False), :472-488, concatenated and reduced by the
 * 1x1 `linear_fuse` conv; resize and 1x1 conv commute, so the host applies the fuse weights at each map's own resolution and only the
 * SUM is formed at the fine scale): dst [N][H][W][C] (fp32, channels contiguous, C % 4 == 0, holds the finest map on entry)
 * += bias[c] (nullable) + sum_k resize(s_k [N][h_k][w_k][C]) for the non-null sources -- torch's half-pixel source index and tap order.
 * _bwd: the adjoint for ONE source, ds [N][h][w][C] = resize^T(dout) (the gradient wrt the finest map is dout itself); a gather
 * over each coarse pixel's footprint with the forward's weights, deterministic. */
int diga_pyramid_sum_fwd(float* dst, int64_t H, int64_t W, const float* bias, const float* s0, int64_t h0, int64_t w0,
                         const float* s1, int64_t h1, int64_t w1, const float* s2, int64_t h2, int64_t w2, int64_t N, int64_t C,
                         void* stream);
int diga_pyramid_sum_bwd(const float* dout, int64_t H, int64_t W, float* ds, int64_t h, int64_t w, int64_t N, int64_t C, void* stream);
/* The three adjoints of the SegFormer geometry in one pass over dout: ds2 [N][H/2][W/2][C], ds4 [N][H/4][W/4][C], ds8 [N][H/8][W/8][C]
 * (the stage maps of segformer_head.py:141-147 for a crop whose side is a multiple of 64); H % 16 == 0, W % 16 == 0, C % 16 == 0.
 * Same weights as diga_pyramid_sum_bwd, another summation order (fixed: deterministic). */
/* diga_pyramid_sum_fwd for that geometry (s2 / s4 / s8: the maps at 1/2, 1/4, 1/8 of dst's grid; H % 16 == 0, W % 16 == 0,
 * C % 32 == 0), tiled through LDS, the same taps in the same order; stats (nullable, [N * H * W / 64][3][C]) receives what the BatchNorm behind
 * `linear_fuse` needs -- per 64-pixel chunk and channel {sum (y - s), sum (y - s)^2, s}, the layout diga_bn_fwd_partials finalises
 * with chunk_rows = 64 (a chunk = four rows of a 16 x 16 tile; the finaliser only needs every chunk to hold 64 values). */
int diga_pyramid_sum_fwd3(float* dst, int64_t H, int64_t W, const float* bias, const float* s2, const float* s4, const float* s8,
                          float* stats, int64_t N, int64_t C, void* stream);
int diga_pyramid_sum_bwd3(const float* dout, int64_t H, int64_t W, float* ds2, float* ds4, float* ds8, int64_t N, int64_t C, void* stream);

/* 3x3 stride-2 pad-1 max-pool with ceil_mode (Ho = ceil((H-1)/2)+1 clipped so the last window starts inside);
 * idx [N,Ho,Wo,C] uint8 = winning tap (first maximum); backward gathers, no atomics. */
int diga_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C,
                          int64_t Ho, int64_t Wo, void* stream);
int diga_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t N, int64_t H, int64_t W, int64_t C,
                          int64_t Ho, int64_t Wo, void* stream);

/* ------------------------------------------------------------------------------------
 * Colour-augmentation view: out = beta*Normalize(extra_aug(x)) + (1-beta)*x in one pass,
 * G5/train_DiGA_gta2city_warm_up.py:105-111,233 with extra_aug = kornia 0.5.8 ColorJitter -> RandomGrayscale ->
 * RandomGaussianBlur(3x3, sigma 2, reflect) -> RandomSharpness and Normalize = G5/util/utils.py:141-156.  kornia is not
 * part of the reference tree: PARITY UNPINNED against kornia, pinned against oracle/coloraug.py's restatement.
 * x, out [B,3,H,W] fp32 NCHW (device).  params [B][12] fp32 (device): {apply jitter, apply grayscale, apply blur,
 * apply sharpness} as 0/1, then brightness, contrast, saturation, hue factors, sharpness factor.  order (HOST int32[4]):
 * the permutation of {0 brightness, 1 contrast, 2 saturation, 3 hue} applied to every jittered sample.  mean, std:
 * HOST float[3].
 * ---------------------------------------------------------------------------------- */
int diga_color_aug_view(const float* x, float* out, const float* params, const int32_t* order_host, int64_t B, int64_t H,
                        int64_t W, float beta, const float* mean_host, const float* std_host, void* stream);

/* ------------------------------------------------------------------------------------
 * Kernel-family timing (bench.py's roofline leg): when enabled, every entry point brackets its
 * launches with HIP events recorded on the launch stream.  Not for production steps.
 * ---------------------------------------------------------------------------------- */
enum {
    DIGA_PROF_CE2D = 0,
    DIGA_PROF_DISTILL,
    DIGA_PROF_UPSAMPLE_LOSS,
    DIGA_PROF_EMA,
    DIGA_PROF_SGD,
    DIGA_PROF_CLASSMIX_HIST,
    DIGA_PROF_CLASSMIX_PASTE,
    DIGA_PROF_CENTROID_WEIGHTS,
    DIGA_PROF_CONSENSUS,
    DIGA_PROF_CLASS_MEANS,
    DIGA_PROF_CENTROID_APPLY,
    DIGA_PROF_CONV_FWD,
    DIGA_PROF_CONV_BWD_DATA,
    DIGA_PROF_CONV_BWD_WEIGHT,
    DIGA_PROF_NORM,
    DIGA_PROF_ELEMENTWISE,
    DIGA_PROF_MIT_GEMM,        /* fp16 Linear / patch-embedding GEMMs, forward and backward-data (include/diga_mit.h) */
    DIGA_PROF_MIT_WGRAD,       /* fp16 weight-gradient GEMMs */
    DIGA_PROF_MIT_ATTN_FWD,    /* spatial-reduction attention, forward */
    DIGA_PROF_MIT_ATTN_BWD,    /* spatial-reduction attention, backward (dQ, dK/dV) */
    DIGA_PROF_MIT_NORM,        /* LayerNorm forward / backward */
    DIGA_PROF_MIT_DWCONV,      /* depthwise 3x3 + GELU (Mix-FFN), forward / backward */
    DIGA_PROF_MIT_MISC,        /* im2col / col2im, casts, column sums */
    DIGA_PROF_NTAGS
};
int diga_prof_enable(int on);
int diga_prof_reset(void);
/* Waits for the recorded events of `tag`; h_count = launches seen, h_total_ms = summed duration. */
int diga_prof_query(int tag, int64_t* h_count, double* h_total_ms);
/* Summed ALGORITHMIC work the recorded calls of `tag` declared: bytes (HBM-bound families: what the call must move
 * given its arguments, e.g. 8 B per element for a BatchNorm apply, +4 with a residual) or FLOPs (convolutions:
 * 2*M*Cout*R*S*Cin). */
int diga_prof_query_work(int tag, double* h_work);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DIGA_HIP_H */
