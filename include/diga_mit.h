/*
 * diga_mit.h -- C ABI of the MiT / SegFormer student kernels in libdiga_hip.so (BASELINE.json configs[4]: "SegFormer-B5
 * (MiT transformer) backbone variant, fp16 -- MFMA attention path for the distillation student").
 *
 * Reference: /root/reference/domain_adaptation/GTA5/model/networks/MixTransfomer.py (G5/.../MixTransfomer.py below), a
 * Python/torch module tree with no FFI; each entry point replaces the torch ops of the cited lines.  The host-side mirror
 * is diga_amd/model/networks/MixTransfomer.py (same class names and state-dict keys), which binds these with ctypes.
 *
 * Conventions (as include/diga_hip.h): device pointers, caller-owned buffers and workspaces, asynchronous on `stream`,
 * no allocation, no sync, 0 / DIGA_E* / hipError_t return codes.  Token matrices are row-major [rows = B*H*W][channels]
 * (channels contiguous = NHWC); "16" = IEEE fp16 storage, "32" = fp32.  All arithmetic accumulates in fp32.
 */
#ifndef DIGA_MIT_H
#define DIGA_MIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libdiga_hip.so is built with -fvisibility=hidden: the functions declared in this header are its whole dynamic symbol table */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* nn.Linear forward and backward-data (MixTransfomer.py:52-55,72-79 Mlp.fc1/fc2; :97-99,122,130,135 Attention.q/kv/proj),
 * and -- on rows gathered by diga_mit_im2col -- the patch-embedding / spatial-reduction convolutions (:200,105).
 *   out[M,N] = residual + seg_scale[m / rows_per_seg] * (alpha * A[M,K] . B[N,K]^T + bias)
 * A, B fp16 (K contiguous, K % 32 == 0, N % 4 == 0); out fp16 (out_f32 = 0; `accumulate` adds to what is there) or fp32;
 * bias [N] fp32, residual [M][ldr] fp32 and seg_scale (DropPath, :176-177) nullable. */
int diga_mit_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* out, int64_t ldc,
                     int out_f32, const float* residual, int64_t ldr, const float* seg_scale, int64_t rows_per_seg,
                     int accumulate, float alpha, int64_t M, int64_t N, int64_t K, void* stream);

/* nn.Linear backward-weight: dw[N,K] (fp32) = scale * sum_m A[m,N] * B[m,K]  (+ dw when accumulate); A = dY, B = X, fp16,
 * N % 8 == 0, K % 8 == 0.  dbias [N] (nullable) = scale * column sums of A, from the same pass (the bias gradient).
 * Split over rows, fp32 slabs in the workspace summed in fixed order (deterministic). */
size_t diga_mit_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K);
int diga_mit_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* dw, float* dbias, float scale, int accumulate,
                     void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K, void* stream);

/* Bias gradient: out[C] (fp32) = scale * column sums of the fp16 matrix x[M][ld] (+ out when accumulate).  C % 8 == 0. */
size_t diga_mit_colsum_workspace_bytes(int64_t M, int64_t C);
int diga_mit_colsum(const void* x, int64_t ld, float* out, float scale, int accumulate, void* workspace, size_t workspace_bytes,
                    int64_t M, int64_t C, void* stream);

/* fp32 weight [R][C] -> fp16 copy and/or fp16 transpose [C][R] (once per optimizer step; either output nullable). */
int diga_mit_cast_transpose(const float* w, void* w16, void* wt16, int64_t R, int64_t C, void* stream);

/* Every weight of a model in ONE launch (per forward; the parameters change in place at every optimizer / EMA step): entry t
 * describes a [Co][Ci][R][S] fp32 parameter (nn.Linear: R = S = 1; RS = R*S) whose GEMM form is rows[co][k],
 * k = (r*S + s)*Ci + ci, zero-padded to Kp columns.
 *   mode 0: out_a = fp16 [Co][Kp], out_b = fp16 [Kp][Co]                        (forward / backward-data operands)
 *   mode 1: out_a = fp32 [RS][Co], out_b = the same with the taps reversed      (depthwise 3x3 forward / backward, Ci = 1)
 * `table` [n_tensors] and `tile_start` [n_tensors] (first 32x32 tile of tensor t; tiles = ceil(Co/32) * tiles_k,
 * tiles_k = ceil(Kp/32)) are DEVICE arrays built once by the caller; total_tiles = grid size. */
typedef struct diga_mit_weight_prep {
    const float* src;
    void* out_a;
    void* out_b;
    int Co, Ci, RS, Kp, mode, tiles_k;
} diga_mit_weight_prep_t;
int diga_mit_weight_prep_multi(const diga_mit_weight_prep_t* table, const int64_t* tile_start, int64_t n_tensors,
                               int64_t total_tiles, void* stream);

/* fp32 -> fp16 with a scale (a gradient entering the fp16 backward pass picks up the loss scale).  n % 4 == 0. */
int diga_mit_cast_scale(const float* x, void* y16, int64_t n, float scale, void* stream);

/* y16[m][:] = x16[m][:] * seg_scale[m / rows_per_seg] on dense fp16 [M][C] (C % 8 == 0): the gradient of a branch that the
 * forward scaled per image (DropPath, MixTransfomer.py:176-177). */
int diga_mit_row_scale(const void* x16, void* y16, const float* seg_scale, int64_t rows_per_seg, int64_t M, int64_t C, void* stream);

/* nn.LayerNorm(C, eps) over the channel axis (MixTransfomer.py:152,160,176-177 Block.norm1/norm2; :106,128 Attention.norm;
 * :202,220 OverlapPatchEmbed.norm; :273-301,380-406 stage norms): x fp32 [M][ldx] -> y16 (fp16) and/or y32 (fp32), both
 * [M][ldy]; mean / rstd [M] saved for backward (nullable together).  C % 4 == 0, C <= 512. */
int diga_mit_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y16, float* y32, int64_t ldy,
                           float* mean, float* rstd, int64_t M, int64_t C, float eps, void* stream);

/* LayerNorm backward fused with the residual junction x + f(norm(x)):
 *   dx = dres + LN'(gscale * dy)     -> dx32 (fp32) and/or dx16 (fp16), [M][ldo];  dres (fp32, [M][ldr]) nullable
 *   dgamma / dbeta [C] = param_scale * sums (+ existing when accumulate)
 * dy is fp16 (dy_is_f32 = 0) or fp32. */
size_t diga_mit_layernorm_bwd_workspace_bytes(int64_t M, int64_t C);
int diga_mit_layernorm_bwd(const void* dy, int dy_is_f32, int64_t ldg, float gscale, const float* x, int64_t ldx, const float* gamma,
                           const float* mean, const float* rstd, const float* dres, int64_t ldr, float* dx32, void* dx16, int64_t ldo,
                           float* dgamma, float* dbeta, float param_scale, int accumulate, void* workspace, size_t workspace_bytes,
                           int64_t M, int64_t C, void* stream);

/* Mix-FFN middle: depthwise 3x3 conv (stride 1, padding 1, bias) + GELU (MixTransfomer.py:72-74, DWConv :409-423), on
 * [B][H][W][C] fp16.  wt9 = the [C,1,3,3] weight re-laid as [9][C] fp32.  u16 (pre-activation, for backward; nullable) and
 * h16 = gelu(u) are written.  C % 8 == 0. */
int diga_mit_dwconv_gelu_fwd(const void* x, const float* wt9, const float* bias, void* u16, void* h16, int64_t B, int64_t H, int64_t W,
                             int64_t C, void* stream);

/* Its backward: du = dh * gelu'(u) (scratch du16), dx16 = conv with the flipped taps (wt9_flipped[t] = wt9[8 - t]),
 * dw [C][9] and db [C] (fp32) = param_scale * sums (+ existing when accumulate).  dx16 may alias dh. */
size_t diga_mit_dwconv_bwd_workspace_bytes(int64_t B, int64_t H, int64_t C);
int diga_mit_dwconv_gelu_bwd(const void* dh, const void* u, const void* x, const float* wt9_flipped, void* du16, void* dx16, float* dw,
                             float* db, float param_scale, int accumulate, void* workspace, size_t workspace_bytes, int64_t B, int64_t H,
                             int64_t W, int64_t C, void* stream);

/* Row gather for convolutions run as GEMMs: cols[(b,oy,ox)][(ky*S + kx)*C + c] (fp16, zero outside the image, zero-padded to
 * Kp columns) from src_kind 0 = fp32 [B][H][W][C], 1 = fp16 [B][H][W][C], 2 = fp32 NCHW (the input image).
 * OverlapPatchEmbed.proj: 7x7/4 pad 3 and 3x3/2 pad 1 (MixTransfomer.py:245-252); Attention.sr: k = stride = sr_ratio (:105). */
int diga_mit_im2col(const void* src, int src_kind, void* cols, int64_t B, int64_t H, int64_t W, int64_t C, int64_t R, int64_t S,
                    int64_t stride, int64_t pad, int64_t Ho, int64_t Wo, int64_t Kp, void* stream);

/* Adjoint of diga_mit_im2col (gather form, no atomics): dst fp32 [B][H][W][C] = gscale * sum (overwrite) when dst_f32,
 * else dst fp16 += gscale * sum. */
int diga_mit_col2im(const void* dcols, void* dst, int dst_f32, float gscale, int64_t B, int64_t H, int64_t W, int64_t C, int64_t R,
                    int64_t S, int64_t stride, int64_t pad, int64_t Ho, int64_t Wo, int64_t Kp, void* stream);

/* Spatial-reduction attention (MixTransfomer.py:120-137): per image b and head h (head_dim 64)
 *   out[b, n, h, :] = softmax_k(scale * q[b,n,h,:] . k[b,k,h,:]) @ v[b,:,h,:]
 * q [B*N][ldq] (head h at columns 64h..), kv [B*Nk][ldkv] (K at columns 64h.., V at 64*heads + 64h..): the layouts the
 * q / kv Linear layers write (:122,130-132).  out [B*N][ldo]; lse [B][heads][N] (log2-domain log-sum-exp, for backward;
 * nullable).  Scores never reach memory. */
int diga_mit_attention_fwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, void* out, int64_t ldo, float* lse, int64_t B,
                           int64_t heads, int64_t N, int64_t Nk, float scale, void* stream);

/* Its backward: d_out -> dq [B*N][ldq], dkv [B*Nk][2*heads*64] (same column layout as kv; ldkv must equal 2*heads*64). */
size_t diga_mit_attention_bwd_workspace_bytes(int64_t B, int64_t heads, int64_t N, int64_t Nk);
int diga_mit_attention_bwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const void* out, const void* d_out, int64_t ldo,
                           const float* lse, void* dq, void* dkv, void* workspace, size_t workspace_bytes, int64_t B, int64_t heads,
                           int64_t N, int64_t Nk, float scale, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DIGA_MIT_H */
