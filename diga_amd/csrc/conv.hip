// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s
// peak on MI355X) for the DeepLabV2/ResNet-101 trunk and the ASPP head.
// Reference: the nn.Conv2d layers of G5/model/seg_model_noaux.py:57-101 (Bottleneck 1x1 / dilated 3x3),
// :140-172 (ASPP 1x1 + four dilated 3x3 2048->256, 3x3 1280->256, 1x1 256->19), run there through cuDNN.
//
// Data layout: activations NHWC fp32 ([pixel][channel], `ld` floats between pixels so that a kernel can
// read or write a channel slice of a wider tensor, e.g. the 1280-channel ASPP concat); weights
// [Cout][R][S][Cin] (the channels_last image of torch's [Cout,Cin,R,S]).  GEMM view: M = N*Ho*Wo output
// pixels, N = Cout, K = R*S*Cin walked tap by tap in 32-channel steps, so one K-step reads, for every
// output pixel of the tile, 128 contiguous bytes of one input pixel (or zeros outside the image).
//
//   conv_fwd_kernel   : forward, and backward-data with the roles swapped (in = dy, weights transposed
//                       to [Cin][R][S][Cout], tap offsets negated).
//   conv_wgrad_kernel : backward-weight, dw[k][tap][c] = sum_p dy[p][k] * x[p + off(tap)][c], split over
//                       pixel ranges into fp32 slabs that a second kernel sums in fixed order
//                       (deterministic: no float atomics).
//
// Tile: 128 x BN x 32 per 256-thread block (4 waves as 2x2, each wave TM x TN tiles of 32x32, 16
// accumulator registers per tile); operands are staged global -> registers -> LDS (two LDS buffers, one
// barrier per K-step; the global loads of step k+1 are in flight while step k's MFMAs issue).  LDS rows
// are padded to 36 floats so that the ds_read_b128 fragment reads of 16 different rows land on 16
// different 4-bank slots (conflict-free).  MFMA fragment mapping: in the 8-wide k group t of a K-step,
// lane (i = lane&31, h = lane>>5) holds A[i][8t+4h .. 8t+4h+3] (one ds_read_b128) and feeds element j to
// MFMA step j; B is read the same way, so both halves of the wave agree on which k they multiply.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>

#include "common.h"

namespace diga {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4nt = __attribute__((ext_vector_type(4))) float;

// 16-byte streaming store: conv outputs (150-600 MB per layer) are never re-read before they have left the caches, and a
// default-policy store keeps its line in L2.  Measured with in-kernel stamps on a pointwise layer (persistent-kernel experiment, 256 -> 1024
// channels): epilogue 17.7 k -> 10.7 k cycles per 128 KB tile and the loaders' waits 3.9 k -> 1.0 k (their L2-resident
// weight panels are no longer evicted by the output stream).
__device__ __forceinline__ void store4_stream(float* p, float x, float y, float z, float w) {
#if defined(DIGA_PROBE_NOSTORE)          // diagnostic builds only (tools/diag/build_probe.sh): what the output stream costs a tile
    if (x == 1.2345e38f) *p = y + z + w;
#elif defined(DIGA_PROBE_PLAINSTORE)
    *reinterpret_cast<float4*>(p) = make_float4(x, y, z, w);
#else
    __builtin_nontemporal_store((f32x4nt){x, y, z, w}, reinterpret_cast<f32x4nt*>(p));
#endif
}

constexpr int kBK = 32;

struct ConvArgs {
    const float* in;
    const float* wgt;
    const uint16_t* wgt_hi;   // split-bf16 weights (conv_fwd_x3p_kernel), else null
    const uint16_t* wgt_lo;
    const unsigned char* wgt_img;   // conv_fwd_x3t_kernel: split weights pre-arranged as LDS images (split_image_kernel)
    const float* bias;
    float* out;
    float* stats;             // nullable: per-(128-row tile, channel) shifted sums for the BatchNorm that follows
    int N, Hi, Wi, Cin, in_ld;
    int Ho, Wo, Cout, out_ld;
    int R, S, sy, sx;            // input coordinate = out*s + off0 + tap*doff
    int oy0, ox0, ody, odx;
    int M, tiles_m, tiles_n;
    int all_inside;              // every tap of every output pixel reads inside the image (no zero padding needed)
    // backward-data epilogue fusion (diga_bwd_epilogue_t): out = [mask](acc + addend), + the BatchNorm-backward column
    // sums of the result -- all null for a plain convolution
    const float* e_add;          // [M][e_add_ld]
    const float* e_masky;        // [M][e_masky_ld]: keep where > 0
    const unsigned char* e_maskbits;   // [M][e_maskbits_ld bytes]: the same mask as one bit per channel (bit c & 7 of byte c >> 3)
    const float* e_x;            // [M][e_x_ld]: input of the BatchNorm whose backward consumes `out`
    const float* e_relu_ab;      // [2][Cout]: keep where fma(x, a, b) > 0
    const float* e_mean;         // [Cout]
    const float* e_invstd;       // [Cout]
    float* e_partials;           // [ceil(M/128)][2][Cout]: sum g, sum g*xhat per 128-row chunk
    int e_add_ld, e_masky_ld, e_x_ld, e_maskbits_ld;
    // input map / output activation (diga_conv2d_next_options; 0 = plain zero-padded convolution)
    int pad_reflect;             // out-of-image taps read the mirrored pixel (nn.ReflectionPad2d in front of the conv)
    int up_shift;                // the conv reads the 2^up_shift nearest-neighbour upsampling of `in` (nn.Upsample in front)
    int act;                     // 1: tanh on the output
    // batched GEMM on conv_fwd_dma_kernel (winograd.hip): 256-row tile tm multiplies weight panel tm / wb_tiles
    int wb_tiles = 0;            // 0: one weight array for every tile
    int64_t wb_stride = 0;       // floats between weight panels
};

// Logical input coordinate (in the optionally upsampled image, before padding) -> source pixel of `in`.
// Returns false for a tap that reads zero padding (then (py, px) is a clamped, valid pixel).
__device__ __forceinline__ bool map_tap(const ConvArgs& a, int iy, int ix, int& py, int& px) {
    const int Hl = a.Hi << a.up_shift, Wl = a.Wi << a.up_shift;
    bool ok = (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
    if (a.pad_reflect) {
        iy = iy < 0 ? -iy : (iy >= Hl ? 2 * (Hl - 1) - iy : iy);
        ix = ix < 0 ? -ix : (ix >= Wl ? 2 * (Wl - 1) - ix : ix);
        ok = true;
    }
    py = min(max(iy, 0), Hl - 1) >> a.up_shift;
    px = min(max(ix, 0), Wl - 1) >> a.up_shift;
    return ok;
}

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    // blocks b and b+8 share an XCD (observed round-robin dispatch): give every XCD a contiguous range
    // of tiles so that neighbouring tiles (shared activation rows / weight panels) hit the same L2.
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Epilogue shared by the forward kernels: the block's 128 x BN accumulator tile is staged through LDS (the operand
// buffers are free after the K loop) and written with 16-byte stores, whole 512-byte (256-byte for BN=64) row
// segments at a time, instead of 64 scalar stores per lane that touch 128-byte pieces.  On the way out it can emit
// what the BatchNorm after the conv needs -- per channel sum(y - s), sum((y - s)^2) and s = the tile's first row,
// over the tile's valid rows -- in exactly the layout colstats_partial_kernel produces with 128-row chunks, so the
// BN forward skips its own statistics pass over the conv output.
template <int TM, int TN, bool EPI = false, int NTHR = 256>
__device__ __forceinline__ void drain_stage(const float* stage_in, const ConvArgs& a, int m0, int n0, int t, int tile_m,
                                            bool active = true);

template <int TM, int TN, bool EPI = false>
__device__ __forceinline__ void epilogue_tile(f32x16 (&acc)[TM][TN], float* __restrict__ stage, const ConvArgs& a,
                                              int m0, int n0, int wm, int wn, int lane, int t, int tile_m) {
    constexpr int BN = 64 * TN, LDS_LD = BN + 4;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[(wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * LDS_LD + wn * 32 * TN + j * 32 + li] =
                    acc[i][j][e];
    __syncthreads();
    drain_stage<TM, TN, EPI>(stage, a, m0, n0, t, tile_m);
}

// Row loop of the backward-data epilogue (drain_stage, EPI), one instantiation per operand combination:
//   ADD: out = acc + addend;  MK: 0 no mask, 1 keep where mask_y > 0, 2 mask bits, 3 keep where fma(x, a, b) > 0;  XX: x is given
//   (needed by MK == 3 and by the BatchNorm-backward sums).  Same expressions in the same order as the generic loop it replaces.
template <int TM, int TN, int NTHR, bool ADD, int MK, bool XX>
__device__ __forceinline__ void epi_rows(const float* __restrict__ stage, const ConvArgs& a, int m0, int n, bool col_ok, int cq, int rg,
                                         const float (&ra)[4], const float (&rb)[4], const float (&mu)[4], const float (&is)[4],
                                         float (&sd)[4], float (&sd2)[4]) {
    constexpr int BN = 64 * TN, LDS_LD = BN + 4, CQ = BN / 4, RG = NTHR / CQ, RPT = 128 / RG;
    constexpr int RB = RPT >= 8 ? 8 : RPT;
    static_assert(MK != 3 || XX, "the relu_ab mask reads x");
    const int nn = col_ok ? n : 0;
    const int mlast = a.M - 1;
    // running row pointers (rows rg, rg + RG, ...): one 64-bit add per row and operand instead of a 64-bit multiply
    const int mfirst = m0 + rg;
    const int64_t r0 = mfirst < a.M ? mfirst : mlast;
    const float* pa = ADD ? a.e_add + r0 * a.e_add_ld + nn : nullptr;
    const float* px = XX ? a.e_x + r0 * a.e_x_ld + nn : nullptr;
    const float* py = MK == 1 ? a.e_masky + r0 * a.e_masky_ld + nn : nullptr;
    const unsigned char* pb = MK == 2 ? a.e_maskbits + r0 * a.e_maskbits_ld + (nn >> 3) : nullptr;
    float* po = a.out + (int64_t)mfirst * a.out_ld + n;
    const int64_t sa = (int64_t)RG * a.e_add_ld, sx = (int64_t)RG * a.e_x_ld, sy = (int64_t)RG * a.e_masky_ld,
                  sb = (int64_t)RG * a.e_maskbits_ld, so = (int64_t)RG * a.out_ld;
    const bool whole = m0 + 128 <= a.M;                         // (uniform) every row of this half-tile exists
    const int shb = nn & 4;
#pragma unroll
    for (int k0 = 0; k0 < RPT; k0 += RB) {
        float4 va[RB], vx[RB], vy[RB];
        unsigned vb[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            // rows beyond M (last tile only): the pointers stop advancing past the last row -- loads stay in bounds, nothing is stored
            if (ADD) va[u] = *reinterpret_cast<const float4*>(pa);
            if (XX) vx[u] = *reinterpret_cast<const float4*>(px);
            if (MK == 1) vy[u] = *reinterpret_cast<const float4*>(py);
            if (MK == 2) vb[u] = (unsigned)*pb >> shb;
            if (k0 + u + 1 < RPT) {
                const bool nxt = whole || m0 + rg + RG * (k0 + u + 1) <= mlast;
                if (ADD) pa += nxt ? sa : 0;
                if (XX) px += nxt ? sx : 0;
                if (MK == 1) py += nxt ? sy : 0;
                if (MK == 2) pb += nxt ? sb : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = rg + RG * (k0 + u);
            const float4 v4 = *reinterpret_cast<const float4*>(stage + r * LDS_LD + cq * 4);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            if (ADD) {
                v[0] += va[u].x; v[1] += va[u].y; v[2] += va[u].z; v[3] += va[u].w;
            } else {
                v[0] += 0.f; v[1] += 0.f; v[2] += 0.f; v[3] += 0.f;      // (the generic loop added a zero addend: -0 -> +0)
            }
            float xx[4] = {0.f, 0.f, 0.f, 0.f};
            if (XX) { xx[0] = vx[u].x; xx[1] = vx[u].y; xx[2] = vx[u].z; xx[3] = vx[u].w; }
            if (MK == 1) {
                v[0] = vy[u].x > 0.f ? v[0] : 0.f; v[1] = vy[u].y > 0.f ? v[1] : 0.f;
                v[2] = vy[u].z > 0.f ? v[2] : 0.f; v[3] = vy[u].w > 0.f ? v[3] : 0.f;
            } else if (MK == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((vb[u] >> e) & 1u) ? v[e] : 0.f;
            } else if (MK == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(xx[e], ra[e], rb[e]) > 0.f ? v[e] : 0.f;
            }
            if (col_ok && (whole || m0 + r < a.M)) {
                store4_stream(po, v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sd[e] += v[e];
                    sd2[e] += v[e] * ((xx[e] - mu[e]) * is[e]);
                }
            }
            po += so;
        }
    }
}

// Second half of the epilogue: 128 staged rows x 64*TN columns (row stride 64*TN + 4 floats) -> global memory
// (+ bias, + BatchNorm partials for 128-row chunk `tile_m`).  Entered after a barrier that follows the stage writes.
// active = false: a thread group whose 128-row half lies beyond M still walks the barriers (nothing is stored).
// EPI: the backward-data epilogue of diga_bwd_epilogue_t (its own instantiation of every kernel, so that the plain
// kernels keep their register budget: inlined into the 256-register kernels the extra row buffers spilled).
template <int TM, int TN, bool EPI, int NTHR>
__device__ __forceinline__ void drain_stage(const float* stage_in, const ConvArgs& a, int m0, int n0, int t, int tile_m,
                                            bool active) {
    float* stage = const_cast<float*>(stage_in);
    constexpr int BN = 64 * TN, LDS_LD = BN + 4;
    constexpr int CQ = BN / 4;            // column quads per row
    constexpr int RG = NTHR / CQ;         // row groups (threads sharing a column quad)
    constexpr int RPT = 128 / RG;         // rows per thread
    const int cq = t % CQ, rg = t / CQ;
    const int n = n0 + cq * 4;
    const bool vec_ok = (a.out_ld & 3) == 0 && n + 3 < a.Cout && ((uintptr_t)a.out & 15u) == 0;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias != nullptr) {
        bv.x = n + 0 < a.Cout ? a.bias[n + 0] : 0.f;
        bv.y = n + 1 < a.Cout ? a.bias[n + 1] : 0.f;
        bv.z = n + 2 < a.Cout ? a.bias[n + 2] : 0.f;
        bv.w = n + 3 < a.Cout ? a.bias[n + 3] : 0.f;
    }
    const float4 s0 = *reinterpret_cast<const float4*>(stage + cq * 4);       // tile row 0 (always a valid row)
    const float4 sh = make_float4(s0.x + bv.x, s0.y + bv.y, s0.z + bv.z, s0.w + bv.w);
    float sd[4] = {0.f, 0.f, 0.f, 0.f}, sd2[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI) {
        // Backward-data of a convolution whose input came out of a BatchNorm(+ReLU)(+residual): the gradient this tile
        // produces is, on its way out, (1) summed with the gradient that reaches the same tensor through the residual
        // branch, (2) masked by that BatchNorm's ReLU, (3) reduced into the column sums its backward needs
        // (sum g, sum g*xhat per 128-row chunk) -- instead of an add pass, a mask inside two later passes and a reduce
        // pass over the tensor.  Entry points guarantee Cout % 4 == 0, 16-byte aligned pointers, lds % 4 == 0.
        float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f};
        const bool col_ok = n + 3 < a.Cout;
        if (col_ok && a.e_relu_ab != nullptr) {
            const float4 t0 = *reinterpret_cast<const float4*>(a.e_relu_ab + n), t1 = *reinterpret_cast<const float4*>(a.e_relu_ab + a.Cout + n);
            ra[0] = t0.x; ra[1] = t0.y; ra[2] = t0.z; ra[3] = t0.w;
            rb[0] = t1.x; rb[1] = t1.y; rb[2] = t1.z; rb[3] = t1.w;
        }
        if (col_ok && a.e_partials != nullptr) {
            const float4 t0 = *reinterpret_cast<const float4*>(a.e_mean + n), t1 = *reinterpret_cast<const float4*>(a.e_invstd + n);
            mu[0] = t0.x; mu[1] = t0.y; mu[2] = t0.z; mu[3] = t0.w;
            is[0] = t1.x; is[1] = t1.y; is[2] = t1.z; is[3] = t1.w;
        }
        // The epilogue moves 3-4x the bytes of the tile itself: rows are taken four at a time with all their loads
        // (up to 12 x 16 B per thread, ~48 KB per block) in flight before the first use.  Which operands exist is uniform over the
        // grid: one instantiation of the row loop per combination (addend?, mask kind, x?), entered through scalar branches, with
        // running row pointers -- written with the nullable pointers tested per element the loop was 3000 VALU instructions per
        // thread (selects, 64-bit multiplies for every row address): 24 000 cycles per tile and SIMD against the 65 000 of a
        // K = 256 tile's MFMAs, during which the matrix pipe idles (the 60 % of conv_fwd_dma_kernel<true>).
        const int mk = a.e_masky != nullptr ? 1 : a.e_maskbits != nullptr ? 2 : a.e_relu_ab != nullptr ? 3 : 0;
        const bool has_add = a.e_add != nullptr, has_x = a.e_x != nullptr;
#define DIGA_EPI_ROWS(ADD, MK, XX) epi_rows<TM, TN, NTHR, ADD, MK, XX>(stage, a, m0, n, col_ok, cq, rg, ra, rb, mu, is, sd, sd2)
        if (has_add) {
            if (mk == 2) { if (has_x) DIGA_EPI_ROWS(true, 2, true); else DIGA_EPI_ROWS(true, 2, false); }
            else if (mk == 3) DIGA_EPI_ROWS(true, 3, true);
            else if (mk == 1) { if (has_x) DIGA_EPI_ROWS(true, 1, true); else DIGA_EPI_ROWS(true, 1, false); }
            else { if (has_x) DIGA_EPI_ROWS(true, 0, true); else DIGA_EPI_ROWS(true, 0, false); }
        } else {
            if (mk == 2) { if (has_x) DIGA_EPI_ROWS(false, 2, true); else DIGA_EPI_ROWS(false, 2, false); }
            else if (mk == 3) DIGA_EPI_ROWS(false, 3, true);
            else if (mk == 1) { if (has_x) DIGA_EPI_ROWS(false, 1, true); else DIGA_EPI_ROWS(false, 1, false); }
            else { if (has_x) DIGA_EPI_ROWS(false, 0, true); else DIGA_EPI_ROWS(false, 0, false); }
        }
#undef DIGA_EPI_ROWS
        if (a.e_partials == nullptr) return;   // uniform over the grid
        __syncthreads();
        float* red = stage;                    // [2][RG][BN]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[(0 * RG + rg) * BN + cq * 4 + e] = sd[e];
            red[(1 * RG + rg) * BN + cq * 4 + e] = sd2[e];
        }
        __syncthreads();
        if (active && t < BN && n0 + t < a.Cout) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                t1 += red[(0 * RG + g) * BN + t];
                t2 += red[(1 * RG + g) * BN + t];
            }
            float* sp = a.e_partials + (int64_t)tile_m * 2 * a.Cout + n0 + t;
            sp[0] = t1;
            sp[a.Cout] = t2;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = rg + RG * k;
        const int m = m0 + r;
        if (m < a.M) {
            float4 v = *reinterpret_cast<const float4*>(stage + r * LDS_LD + cq * 4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (a.act == 1) {
                v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
            }
            float* o = a.out + (int64_t)m * a.out_ld + n;
            if (vec_ok) {
                store4_stream(o, v.x, v.y, v.z, v.w);
            } else {
                if (n + 0 < a.Cout) o[0] = v.x;
                if (n + 1 < a.Cout) o[1] = v.y;
                if (n + 2 < a.Cout) o[2] = v.z;
                if (n + 3 < a.Cout) o[3] = v.w;
            }
            const float d0 = v.x - sh.x, d1 = v.y - sh.y, d2 = v.z - sh.z, d3 = v.w - sh.w;
            sd[0] += d0; sd[1] += d1; sd[2] += d2; sd[3] += d3;
            sd2[0] += d0 * d0; sd2[1] += d1 * d1; sd2[2] += d2 * d2; sd2[3] += d3 * d3;
        }
    }
    if (a.stats == nullptr) return;        // uniform over the grid
    __syncthreads();                       // everyone is done reading the staged tile
    float* red = stage;                    // [2][RG][BN]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[(0 * RG + rg) * BN + cq * 4 + e] = sd[e];
        red[(1 * RG + rg) * BN + cq * 4 + e] = sd2[e];
    }
    if (rg == 0) *reinterpret_cast<float4*>(red + 2 * RG * BN + cq * 4) = sh;      // the shift of these 4 columns
    __syncthreads();
    if (active && t < BN && n0 + t < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            t1 += red[(0 * RG + g) * BN + t];
            t2 += red[(1 * RG + g) * BN + t];
        }
        float* sp = a.stats + (int64_t)tile_m * 3 * a.Cout + n0 + t;
        sp[0] = t1;
        sp[a.Cout] = t2;
        sp[2 * a.Cout] = red[2 * RG * BN + t];
    }
}

template <int TM, int TN, int BK>
__device__ __forceinline__ void mma_kstep(const float* __restrict__ As, const float* __restrict__ Bs, int a_row0,
                                          int b_row0, int lane, f32x16 (&acc)[TM][TN]) {
    constexpr int LD = BK + 4;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[i] = *reinterpret_cast<const float4*>(As + (a_row0 + i * 32 + li) * LD + t * 8 + lh * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b[j] = *reinterpret_cast<const float4*>(Bs + (b_row0 + j * 32 + li) * LD + t * 8 + lh * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float av = e == 0 ? a[i].x : e == 1 ? a[i].y : e == 2 ? a[i].z : a[i].w;
                    const float bv = e == 0 ? b[j].x : e == 1 ? b[j].y : e == 2 ? b[j].z : b[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward / backward-data
// ---------------------------------------------------------------------------------------------
// block tile 128 x (64*TN) x BK; wave tile 64 x (32*TN).  BK = 32: 2 blocks per CU (74 KB LDS each);
// BK = 16: 37 KB LDS, 3 blocks per CU (register-limited) -> a third wave per SIMD to cover barrier stalls.
template <int TN, int BK, bool EPI = false>
__global__ __launch_bounds__(256, (BK == 16 ? 3 : 2)) void conv_fwd_kernel(ConvArgs a) {
    constexpr int BM = 128, BN = 64 * TN, TM = 2;
    constexpr int LD = BK + 4;
    constexpr int CPR = BK / 4;          // float4 chunks per LDS row
    constexpr int RPP = 256 / CPR;       // rows staged per pass of the block
    constexpr int NPA = BM / RPP, NPB = BN / RPP;
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                    // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;      // [2][BN][LD]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // loader geometry: thread owns column chunk c4 (4 floats) of rows lr + RPP*i
    const int lr = t / CPR, c4 = (t % CPR) * 4;
    int pixbase[NPA], iy0[NPA], ix0[NPA];
    bool mok[NPA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + lr + RPP * i;
        mok[i] = m < a.M;
        const int mm = mok[i] ? m : 0;
        const int img = mm / HoWo, rem = mm - img * HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pixbase[i] = img * a.Hi * a.Wi;
        iy0[i] = ho * a.sy + a.oy0;
        ix0[i] = wo * a.sx + a.ox0;
    }
    const int RS = a.R * a.S;
    const int cchunks = a.Cin / BK;
    const int ksteps = RS * cchunks;

    // Branch-free loader: out-of-image taps / rows beyond M read a clamped (valid) address and are zeroed by a
    // select afterwards, so the 8 loads of a K-step issue back to back instead of through 8 exec-masked branches.
    float4 ra[NPA], rb[NPB];
    float fa[NPA], fb[NPB];
    int cob[NPB];
    bool cok[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int co = n0 + lr + RPP * i;
        cok[i] = co < a.Cout;
        cob[i] = cok[i] ? co : a.Cout - 1;
        fb[i] = cok[i] ? 1.f : 0.f;
    }
    auto gload = [&](int ks) {
        const int tap = ks / cchunks, c0 = (ks - tap * cchunks) * BK;
        const int r = tap / a.S, s = tap - r * a.S;
        const int dy = r * a.ody, dx = s * a.odx;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            int cy, cx;
            const bool ok = map_tap(a, iy, ix, cy, cx) && mok[i];
            const float* p = a.in + (int64_t)(pixbase[i] + cy * a.Wi + cx) * a.in_ld + c0 + c4;
            ra[i] = *reinterpret_cast<const float4*>(p);
            fa[i] = ok ? 1.f : 0.f;               // applied at LDS-store time (a multiply: hipcc turns a select
                                                  // back into a branch around the load)
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const float* p = a.wgt + ((int64_t)cob[i] * RS + tap) * a.Cin + c0 + c4;
            rb[i] = *reinterpret_cast<const float4*>(p);
        }
    };
    auto lstore = [&](int buf) {
        float* ad = As + buf * BM * LD;
        float* bd = Bs + buf * BN * LD;
#pragma unroll
        for (int i = 0; i < NPA; ++i)
            *reinterpret_cast<float4*>(ad + (lr + RPP * i) * LD + c4) =
                make_float4(ra[i].x * fa[i], ra[i].y * fa[i], ra[i].z * fa[i], ra[i].w * fa[i]);
#pragma unroll
        for (int i = 0; i < NPB; ++i)
            *reinterpret_cast<float4*>(bd + (lr + RPP * i) * LD + c4) =
                make_float4(rb[i].x * fb[i], rb[i].y * fb[i], rb[i].z * fb[i], rb[i].w * fb[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) gload(ks + 1);
        mma_kstep<TM, TN, BK>(As + cur * BM * LD, Bs + cur * BN * LD, wm * 64, wn * 32 * TN, lane, acc);
        if (ks + 1 < ksteps) lstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: accumulator register e of a 32x32 tile is row (e&3) + 8*(e>>2) + 4*(lane>>5), col lane&31
    if constexpr (BK == 32) {
        epilogue_tile<TM, TN, EPI>(acc, smem, a, m0, n0, wm, wn, lane, t, tile_m);
    } else {
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 32 * TN + j * 32 + li;
            const bool nok = n < a.Cout;
            const float bv = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (nok && m < a.M) a.out[(int64_t)m * a.out_ld + n] = acc[i][j][e] + bv;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward / backward-data on the bf16 matrix cores with split operands ("bf16x3")
//
// Every fp32 operand x is split while it is staged into LDS into hi = bf16(x) and lo = bf16(x - hi)
// (x = hi + lo up to 2^-18 relative) and a*b is evaluated as hi*hi + hi*lo + lo*hi with
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 3 MFMAs of 32 cycles per 16 k instead of 8 fp32 MFMAs of
// 64 cycles -- 5.3x less matrix-pipe time at ~1e-5 relative error per product (the dropped lo*lo term
// and the split residual are each <= 2^-18), well inside the path's 1e-3 logit tolerance.
// Same tiling, loader and epilogue as conv_fwd_kernel; LDS rows are 32 bf16 (64 B) + 16 B of padding so
// that the 16 lanes of a ds_read_b128 group fall on 16 different 16-byte slots.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);   // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(uint32_t, h);
}

// 4 fp32 -> 4 bf16 hi (8 B) + 4 bf16 lo (8 B)
__device__ __forceinline__ void split4(const float4 v, const float f, uint2& hi, uint2& lo) {
    const float x0 = v.x * f, x1 = v.y * f, x2 = v.z * f, x3 = v.w * f;
    hi.x = pack_bf16(x0, x1);
    hi.y = pack_bf16(x2, x3);
    const float h0 = __uint_as_float(hi.x << 16), h1 = __uint_as_float(hi.x & 0xffff0000u);
    const float h2 = __uint_as_float(hi.y << 16), h3 = __uint_as_float(hi.y & 0xffff0000u);
    lo.x = pack_bf16(x0 - h0, x1 - h1);
    lo.y = pack_bf16(x2 - h2, x3 - h3);
}

// unmasked variant (all taps inside the image): 12 VALU instead of 16
__device__ __forceinline__ void split4_nomask(const float4 v, uint2& hi, uint2& lo) {
    hi.x = pack_bf16(v.x, v.y);
    hi.y = pack_bf16(v.z, v.w);
    const float h0 = __uint_as_float(hi.x << 16), h1 = __uint_as_float(hi.x & 0xffff0000u);
    const float h2 = __uint_as_float(hi.y << 16), h3 = __uint_as_float(hi.y & 0xffff0000u);
    lo.x = pack_bf16(v.x - h0, v.y - h1);
    lo.y = pack_bf16(v.z - h2, v.w - h3);
}

constexpr int kRowB = 80;   // bytes per LDS row: 32 bf16 + 16 B pad

// ---------------------------------------------------------------------------------------------
// Split-bf16 forward / backward-data, tuned variant: weights arrive already split (two bf16 arrays written
// once per step by split_bf16_kernel, so the B operand is staged with two 8-byte copies and no VALU work),
// and the K loop walks tap-major with per-row pointers that are recomputed only when the tap changes
// (one 64-bit add per load instead of a clamp + multiply-add chain).  Rows beyond M / columns beyond Cout
// read clamped addresses and are simply never stored; only out-of-image taps are zeroed.
// ---------------------------------------------------------------------------------------------
template <int TN, bool EPI = false>
__global__ __launch_bounds__(256, 2) void conv_fwd_x3p_kernel(ConvArgs a) {
    constexpr int BM = 128, BN = 64 * TN, TM = 2, BK = 32;
    constexpr int CPR = BK / 4, RPP = 256 / CPR, NPA = BM / RPP, NPB = BN / RPP;
    constexpr int A_PLANE = BM * kRowB, B_PLANE = BN * kRowB, BUF = 2 * A_PLANE + 2 * B_PLANE;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int lr = t / CPR, c4 = (t % CPR) * 4;
    int pixbase[NPA], iy0[NPA], ix0[NPA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = min(m0 + lr + RPP * i, a.M - 1);
        const int img = m / HoWo, rem = m - img * HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pixbase[i] = img * a.Hi * a.Wi;
        iy0[i] = ho * a.sy + a.oy0;
        ix0[i] = wo * a.sx + a.ox0;
    }
    const int RS = a.R * a.S;
    const int cchunks = a.Cin / BK;
    const int ksteps = RS * cchunks;
    const int64_t wrow = (int64_t)RS * a.Cin;       // elements per output channel of the weight arrays
    int64_t wbase[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) wbase[i] = (int64_t)min(n0 + lr + RPP * i, a.Cout - 1) * wrow + c4;

    // loader state: (l_tap, l_cc) of the NEXT K-step to fetch
    const float* pa[NPA];
    float ftap[NPA];
    int l_tap = 0, l_cc = 0;
    auto set_tap = [&](int tap) {
        const int r = tap / a.S, s = tap - r * a.S;
        const int dy = r * a.ody, dx = s * a.odx;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const int cy = min(max(iy, 0), a.Hi - 1), cx = min(max(ix, 0), a.Wi - 1);
            pa[i] = a.in + (int64_t)(pixbase[i] + cy * a.Wi + cx) * a.in_ld + c4;
            ftap[i] = ok ? 1.f : 0.f;
        }
    };
    float4 ra[NPA];
    float fa[NPA];
    uint2 rbh[NPB], rbl[NPB];
    auto gload = [&]() {
        const int koff = l_tap * a.Cin + l_cc * BK;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(pa[i] + l_cc * BK);
            fa[i] = ftap[i];
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            rbh[i] = *reinterpret_cast<const uint2*>(a.wgt_hi + wbase[i] + koff);
            rbl[i] = *reinterpret_cast<const uint2*>(a.wgt_lo + wbase[i] + koff);
        }
        if (++l_cc == cchunks) {
            l_cc = 0;
            if (++l_tap < RS) set_tap(l_tap);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* base = smem_b + buf * BUF;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            uint2 hi, lo;
            split4(ra[i], fa[i], hi, lo);
            const int off = (lr + RPP * i) * kRowB + c4 * 2;
            *reinterpret_cast<uint2*>(base + off) = hi;
            *reinterpret_cast<uint2*>(base + A_PLANE + off) = lo;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int off = (lr + RPP * i) * kRowB + c4 * 2;
            *reinterpret_cast<uint2*>(base + 2 * A_PLANE + off) = rbh[i];
            *reinterpret_cast<uint2*>(base + 2 * A_PLANE + B_PLANE + off) = rbl[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    set_tap(0);
    gload();
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) gload();
        const unsigned char* Ah = smem_b + cur * BUF;
        const unsigned char* Al = Ah + A_PLANE;
        const unsigned char* Bh = Ah + 2 * A_PLANE;
        const unsigned char* Bl = Bh + B_PLANE;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = (wm * 64 + i * 32 + li) * kRowB + s * 32 + lh * 16;
                ah[i] = *reinterpret_cast<const bf16x8_t*>(Ah + off);
                al[i] = *reinterpret_cast<const bf16x8_t*>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = (wn * 32 * TN + j * 32 + li) * kRowB + s * 32 + lh * 16;
                bh[j] = *reinterpret_cast<const bf16x8_t*>(Bh + off);
                bl[j] = *reinterpret_cast<const bf16x8_t*>(Bl + off);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (ks + 1 < ksteps) lstore(cur ^ 1);
        __syncthreads();
    }

    epilogue_tile<TM, TN, EPI>(acc, reinterpret_cast<float*>(smem_b), a, m0, n0, wm, wn, lane, t, tile_m);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 forward / backward-data, wide variant: 256 x (64*TN) block tile, 4 waves as 2 x 2, wave tile
// 128 x (32*TN) built from v_mfma_f32_16x16x32_bf16 (8 x 2*TN accumulator tiles of 4 registers).
//
// Why: on the 128^2 kernel the LDS is as busy as the matrix pipe (per 32-deep K-step: 32 KB of ds_write_b64 at
// ~75 B/clk + 64 KB of ds_read_b128 at 256 B/clk ~= the 768 MFMA cycles of a wave).  Doubling the rows a wave owns
// halves the B-operand traffic per flop and cuts the A-operand re-reads by the wave grid: per flop 0.75x the
// LDS writes and 0.75x the reads; the 16x16x32 shape also holds a higher clock than 32x32x16 on this chip.
//
// LDS image: four planes (A hi, A lo, B hi, B lo) of 64-byte rows (one pixel / output channel x 32 k), the four
// 16-byte k-slots of row r stored at slot ^ swz(r), swz(r) = G[(r>>2)&3] ^ 2*((r>>1)&1), G = {0,2,3,1}: the
// ds_read_b128 of a 16x16x32 fragment (lane l: row l&15, slot l>>4) then touches 16 distinct slots of the
// 256-byte bank row in each of its four lane groups, and the staging ds_write_b64 (16 lanes = 4 rows x 4 half
// slots) is conflict-free too.  One LDS buffer (two barriers per K-step), two blocks per CU.
//
// Measured alternatives (C2 layer shapes, sum of forward convs of one pass: 128^2 kernel 85.2 ms, this kernel
// 68.2 ms): the same tile on 32x32x16 MFMAs 88.9 ms; a wave-specialised build (4 MFMA waves + 4 loader waves, double
// buffered, one barrier per step) 87.9 ms; tap-inner K order (L2 reuse across taps) no gain; s_setprio around the
// staging phase no gain; 4 (3) instead of 2 global loads per MFMA group 77.5 (73.8) ms; weight tiles pre-arranged as LDS
// images and copied by LDS-DMA (global_load_lds_dwordx4, double-buffered B, no registers / ds_writes on the weight
// side) 71.8 vs 70.9 ms -- the weight side of the staging is not what bounds the kernel.
// ---------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ int lds_swz(int r) { return ((0x78 >> (((r >> 2) & 3) * 2)) & 3) ^ (((r >> 1) & 1) << 1); }

// STAMP = diagnostic build (DIGA_CONV_STAMP=1): per block, wave 0 accumulates s_memtime deltas of the K-loop phases
// {issue loads + MFMA, barrier 1, wait for the global loads, split + ds_write, barrier 2, steps} into a.stats[8 * block]
// instead of BatchNorm partials.  Never used by the product path.
__device__ __forceinline__ uint64_t stamp() {
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t v = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return v;
}

template <int TN, bool STAMP = false, bool EPI = false>
__global__ __launch_bounds__(256, 2) void conv_fwd_x3w_kernel(ConvArgs a) {
    constexpr int BM = 256, BN = 64 * TN, NT = 2 * TN, MT = 8;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // loader: thread = (row lr + 64 p, float4 chunks q and q + 4 of the 32-deep K slice).  Per-row state is kept
    // small (the accumulators own half the register file): image base, packed (y, x) origin, 32-bit element offsets.
    const int q = t & 3, lr = t >> 2;
    int pixbase[4], yx0[4];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = min(m0 + lr + 64 * p, a.M - 1);
        const int img = m / HoWo, rem = m - img * HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pixbase[p] = img * a.Hi * a.Wi;
        yx0[p] = ((ho * a.sy + a.oy0) << 16) | ((wo * a.sx + a.ox0) & 0xffff);
    }
    const int RS = a.R * a.S;
    const int cchunks = a.Cin / 32;
    const int ksteps = RS * cchunks;
    const int wrow = RS * a.Cin;
    int wbase[TN];
#pragma unroll
    for (int p = 0; p < TN; ++p) wbase[p] = min(n0 + lr + 64 * p, a.Cout - 1) * wrow + 4 * q;

    int po[4];                       // element offset of (pixel of the current tap, channel 4 q)
    unsigned tapmask = 0, ldmask = 0; // bit p: the tap lies inside the image for row p (current tap / loaded data)
    int l_tap = 0, l_cc = 0;
    auto set_tap = [&](int tap) {
        const int r = tap / a.S, s = tap - r * a.S;
        const int dy = r * a.ody, dx = s * a.odx;
        tapmask = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int iy = (yx0[p] >> 16) + dy, ix = (int)(short)(yx0[p] & 0xffff) + dx;
            int cy, cx;
            const bool ok = map_tap(a, iy, ix, cy, cx);
            po[p] = (pixbase[p] + cy * a.Wi + cx) * a.in_ld + 4 * q;
            tapmask |= ok ? (1u << p) : 0u;
        }
    };
    float4 ra[4][2];
    uint2 rbh[TN][2], rbl[TN][2];
    // The loads of a K-step are issued unconditionally (the last step re-reads valid addresses) so that they sit
    // in the same basic block as the MFMAs and can be spread between them: a wave issues in order, and 16 loads
    // back to back at the loop top stall on the CU's memory queue with the matrix pipe idle behind them.
    auto gload = [&](bool real) {
        const int koff = real ? l_tap * a.Cin + l_cc * 32 : 0;
        const int aoff = real ? l_cc * 32 : 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float* src = a.in + po[p] + aoff;
            ra[p][0] = *reinterpret_cast<const float4*>(src);
            ra[p][1] = *reinterpret_cast<const float4*>(src + 16);
        }
        ldmask = tapmask;
#pragma unroll
        for (int p = 0; p < TN; ++p) {
            rbh[p][0] = *reinterpret_cast<const uint2*>(a.wgt_hi + wbase[p] + koff);
            rbh[p][1] = *reinterpret_cast<const uint2*>(a.wgt_hi + wbase[p] + koff + 16);
            rbl[p][0] = *reinterpret_cast<const uint2*>(a.wgt_lo + wbase[p] + koff);
            rbl[p][1] = *reinterpret_cast<const uint2*>(a.wgt_lo + wbase[p] + koff + 16);
        }
    };
    auto advance = [&]() {           // loader state -> the K-step after the one just loaded
        if (++l_cc == cchunks) {
            l_cc = 0;
            if (++l_tap < RS) set_tap(l_tap);
        }
    };
    // staging offsets: row lr + 64 p keeps lr's swizzle (64 p does not touch bits 1..3)
    const int wsw = lds_swz(lr);
    const int woff0 = lr * 64 + ((((q >> 1)) ^ wsw) << 4) + ((q & 1) << 3);
    const int woff1 = lr * 64 + ((((q >> 1) | 2) ^ wsw) << 4) + ((q & 1) << 3);
    auto lstore = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint2 hi, lo;
            const float f = (ldmask >> p) & 1u ? 1.f : 0.f;
            if (a.all_inside) {                // uniform
                split4_nomask(ra[p][0], hi, lo);
            } else {
                split4(ra[p][0], f, hi, lo);
            }
            *reinterpret_cast<uint2*>(smem_b + p * 4096 + woff0) = hi;
            *reinterpret_cast<uint2*>(smem_b + A_PLANE + p * 4096 + woff0) = lo;
            if (a.all_inside) {
                split4_nomask(ra[p][1], hi, lo);
            } else {
                split4(ra[p][1], f, hi, lo);
            }
            *reinterpret_cast<uint2*>(smem_b + p * 4096 + woff1) = hi;
            *reinterpret_cast<uint2*>(smem_b + A_PLANE + p * 4096 + woff1) = lo;
        }
#pragma unroll
        for (int p = 0; p < TN; ++p) {
            *reinterpret_cast<uint2*>(smem_b + 2 * A_PLANE + p * 4096 + woff0) = rbh[p][0];
            *reinterpret_cast<uint2*>(smem_b + 2 * A_PLANE + p * 4096 + woff1) = rbh[p][1];
            *reinterpret_cast<uint2*>(smem_b + 2 * A_PLANE + B_PLANE + p * 4096 + woff0) = rbl[p][0];
            *reinterpret_cast<uint2*>(smem_b + 2 * A_PLANE + B_PLANE + p * 4096 + woff1) = rbl[p][1];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offset of this lane inside a 16-row tile
    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ lds_swz(frow)) << 4);
    const unsigned char* Ah = smem_b + wm * 128 * 64 + foff;
    const unsigned char* Al = Ah + A_PLANE;
    const unsigned char* Bh = smem_b + 2 * A_PLANE + wn * 32 * TN * 64 + foff;
    const unsigned char* Bl = Bh + B_PLANE;

    uint64_t tacc[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    set_tap(0);
    gload(true);
    advance();
    lstore();
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        if constexpr (STAMP) t0 = stamp();
        gload(more);
        // fragment reads run one 16-row tile ahead of the MFMAs that consume them (hipcc otherwise parks every read
        // directly in front of its first use and drains lgkmcnt(0) sixteen times per K-step)
        bf16x8_t bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = *reinterpret_cast<const bf16x8_t*>(Bh + j * 1024);
            bl[j] = *reinterpret_cast<const bf16x8_t*>(Bl + j * 1024);
        }
        bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(Ah);
        bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(Al);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT + 2, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            bf16x8_t ahn = ah, aln = al;
            if (i + 1 < MT) {
                ahn = *reinterpret_cast<const bf16x8_t*>(Ah + (i + 1) * 1024);
                aln = *reinterpret_cast<const bf16x8_t*>(Al + (i + 1) * 1024);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            if (i * 2 < 8 + 4 * TN) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // 2 global loads
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT, 0);
            ah = ahn;
            al = aln;
        }
        if constexpr (STAMP) {
            t1 = stamp();
            tacc[0] += t1 - t0;
        }
        __syncthreads();
        if constexpr (STAMP) {
            t0 = stamp();
            tacc[1] += t0 - t1;
        }
        if (more) {
            if constexpr (STAMP) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t1 = stamp();
                tacc[2] += t1 - t0;
            }
            lstore();
            advance();
            if constexpr (STAMP) {
                t0 = stamp();
                tacc[3] += t0 - t1;
            }
            __syncthreads();
            if constexpr (STAMP) {
                t1 = stamp();
                tacc[4] += t1 - t0;
            }
        }
    }
    if constexpr (STAMP) {
        if (t == 0) {
            float* d = a.stats + (int64_t)blockIdx.x * 8;
#pragma unroll
            for (int e = 0; e < 5; ++e) d[e] = (float)tacc[e];
            d[5] = (float)ksteps;
        }
        a.stats = nullptr;
    }

    // epilogue: the two 128-row halves of the tile go through the 128 x (BN + 4) float stage one after the other
    float* stage = reinterpret_cast<float*>(smem_b);
    constexpr int LDS_LD = BN + 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (m0 + h * 128 >= a.M) break;              // uniform over the block
        if (h) __syncthreads();
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        stage[(i * 16 + (lane >> 4) * 4 + e) * LDS_LD + wn * 32 * TN + j * 16 + (lane & 15)] = acc[i][j][e];
        }
        __syncthreads();
        drain_stage<2, TN, EPI>(stage, a, m0 + h * 128, n0, t, tile_m * 2 + h);
    }
}

// x[n] -> hi[n] = bf16(x), lo[n] = bf16(x - hi)   (weights, once per optimizer step)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        uint2 h, l;
        split4(reinterpret_cast<const float4*>(x)[i], 1.f, h, l);
        reinterpret_cast<uint2*>(hi)[i] = h;
        reinterpret_cast<uint2*>(lo)[i] = l;
    }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 forward / backward-data without register staging ("twin" path).
//
// In-kernel stamps of conv_fwd_x3w_kernel: a loop that only reads fragments and issues MFMAs runs a K-step in 2.4 k
// cycles, the real kernel in 5.5 k -- the global -> register -> split -> ds_write -> barrier staging costs more than
// the MFMAs.  Here both operands arrive already split and are copied global -> LDS by LDS-DMA loads
// (global_load_lds_dwordx4: no registers, no VALU, no ds_write) into a two-stage ring, one barrier per K-step:
//   * activations as a "twin": per pixel and group of 8 channels 16 B of bf16 hi + 16 B of bf16 lo (4 B per element,
//     the size of the fp32 tensor), written by make_twin_kernel;
//   * weights as LDS images (split_image_kernel): per 128/64-channel tile and K-step the hi and lo planes exactly as
//     they sit in LDS.
// An LDS-DMA instruction writes 64 x 16 B linearly from a wave-uniform LDS base, so the swizzle of the image is applied
// to the per-lane SOURCE address (lane l fills row l >> 2, slot l & 3 and therefore fetches k-slot (l & 3) ^ swz(row));
// outside taps fetch 16 zero bytes.  Same tile, MFMA loop and epilogue as conv_fwd_x3w_kernel; 144 KB of LDS, one block
// (4 waves) per CU.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned char g_zero16[16];

// x [M][ld] fp32 (C channels) -> twin [M][C/8][hi8 | lo8]
__global__ __launch_bounds__(256) void make_twin_kernel(const float* __restrict__ x, int64_t ld, unsigned char* __restrict__ twin,
                                                        int64_t M, int C8) {
    const int64_t total = M * C8, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / C8;
        const int g = (int)(i - m * C8);
        const float* src = x + m * ld + g * 8;
        uint2 h0, l0, h1, l1;
        split4_nomask(*reinterpret_cast<const float4*>(src), h0, l0);
        split4_nomask(*reinterpret_cast<const float4*>(src + 4), h1, l1);
        uint4* dst = reinterpret_cast<uint4*>(twin + i * 32);
        dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// Weights [K][RS][C] fp32 -> LDS images: for every tile of `bn` output channels and every 32-channel K-step (tap-major,
// the kernel's K order) 2 * bn * 64 bytes = hi plane then lo plane, one 64-byte row per output channel (rows past K
// repeat the last channel: never stored), the four 16-byte k-slots of row r at slot ^ lds_swz(r).
__global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ w, unsigned char* __restrict__ img,
                                                          int K, int RS, int C, int bn, int64_t total) {
    const int cchunks = C / 32, ksteps = RS * cchunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int s = (int)(i & 3);
        const int r = (int)((i >> 2) % bn);
        const int64_t tk = (i >> 2) / bn;               // tile * ksteps + ks
        const int ks = (int)(tk % ksteps), tile = (int)(tk / ksteps);
        const int tap = ks / cchunks, cc = ks - tap * cchunks;
        const int n = min(tile * bn + r, K - 1);
        const float* src = w + ((int64_t)n * RS + tap) * C + cc * 32 + 8 * s;
        uint2 h0, l0, h1, l1;
        split4_nomask(*reinterpret_cast<const float4*>(src), h0, l0);
        split4_nomask(*reinterpret_cast<const float4*>(src + 4), h1, l1);
        unsigned char* dst = img + tk * (2 * (int64_t)bn * 64) + r * 64 + ((s ^ lds_swz(r)) << 4);
        *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(dst + (int64_t)bn * 64) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// Taps of which at least one row of the block's M-tile reads inside the image (bit r*S + s), conservatively: a tile
// inside one image covers the output rows ho0..ho1 (and every column once it spans a full row); a tap whose input rows
// all fall outside contributes exact zeros and its K-steps are skipped (dilated ASPP branches near the image border:
// 15 % of the K-steps at dilation 24 on a 97-row map).  Uniform over the block.  RS <= 64.
__device__ __forceinline__ uint64_t live_taps(const ConvArgs& a, int m0, int BM) {
    const int RS = a.R * a.S;
    const uint64_t all = RS >= 64 ? ~0ull : ((1ull << RS) - 1ull);
    if (a.pad_reflect) return all;                       // mirrored taps are never zero
    const int HoWo = a.Ho * a.Wo;
    const int m_last = min(m0 + BM, a.M) - 1;
    const int img0 = m0 / HoWo, img1 = m_last / HoWo;
    if (img0 != img1) return all;
    const int rem0 = m0 - img0 * HoWo, rem1 = m_last - img0 * HoWo;
    const int ho0 = rem0 / a.Wo, ho1 = rem1 / a.Wo;
    int wo0 = 0, wo1 = a.Wo - 1;
    if (ho0 == ho1) {
        wo0 = rem0 - ho0 * a.Wo;
        wo1 = rem1 - ho0 * a.Wo;
    }
    uint64_t live = 0;
    for (int r = 0; r < a.R; ++r) {
        const int ylo = ho0 * a.sy + a.oy0 + r * a.ody, yhi = ho1 * a.sy + a.oy0 + r * a.ody;
        if (yhi < 0 || ylo >= (a.Hi << a.up_shift)) continue;
        for (int q = 0; q < a.S; ++q) {
            const int xlo = wo0 * a.sx + a.ox0 + q * a.odx, xhi = wo1 * a.sx + a.ox0 + q * a.odx;
            if (xhi < 0 || xlo >= (a.Wi << a.up_shift)) continue;
            live |= 1ull << (r * a.S + q);
        }
    }
    return live != 0 ? live : all;
}

// ---------------------------------------------------------------------------------------------
// conv_fwd_x3t_kernel with TWO MFMA waves per SIMD: 12 waves = 8 MFMA waves as 4 (M) x 2 (N), wave tile 64 x (32*TN)
// (64 accumulator registers, ~150 VGPRs: three waves fit a SIMD), + 4 LDS-DMA loader waves (one per SIMD).  Same
// 256 x (64*TN) x 32 block tile, same three-stage ring, same split / MFMA order per accumulator (bit-identical results).
// Why: with one MFMA wave per SIMD nothing covers that wave's barrier wait and the LDS latency of the first fragment
// reads of every K-step (matrix pipe 66 % busy); a second wave's MFMAs fill those holes.  K-steps of taps that lie
// outside the image for the whole tile are skipped (live_taps).  Both 128-row halves of the epilogue drain at once
// (two 256-thread groups, two stage areas).
// ---------------------------------------------------------------------------------------------
template <int TN, bool EPI = false>
__global__ __launch_bounds__(768, 3) void conv_fwd_x3t8_kernel(ConvArgs a) {
    constexpr int BM = 256, BN = 64 * TN, NT = 2 * TN, MT = 4;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, STAGE = 2 * A_PLANE + 2 * B_PLANE;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool loader = wv >= 8;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RS = a.R * a.S;
    const int cchunks = a.Cin / 32;
    const uint64_t live = RS <= 64 ? live_taps(a, m0, BM) : ~0ull;
    const int ntaps = RS <= 64 ? __builtin_popcountll(live) : RS;
    const int ksteps = ntaps * cchunks;

    if (loader) {
        const int lw = wv - 8;                                   // 0..3: A rows lw*64 + 16 j + (lane >> 2)
        const unsigned char* twin = reinterpret_cast<const unsigned char*>(a.in);
        const int lrow = lane >> 2;
        const int kslot = (lane & 3) ^ lds_swz(lrow);
        int pixbase[4], yx0[4];
        const int HoWo = a.Ho * a.Wo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(m0 + lw * 64 + 16 * j + lrow, a.M - 1);
            const int img = m / HoWo, rem = m - img * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            pixbase[j] = img * a.Hi * a.Wi;
            yx0[j] = ((ho * a.sy + a.oy0) << 16) | ((wo * a.sx + a.ox0) & 0xffff);
        }
        const int64_t rowb = (int64_t)a.in_ld * 4;
        const unsigned char* pa[4];
        uint64_t todo = live;
        int l_tap = 0, l_cc = 0;
        auto next_tap = [&]() {                                  // -> l_tap = next live tap
            if (RS <= 64) {
                l_tap = __builtin_ctzll(todo);
                todo &= todo - 1;
            } else {
                ++l_tap;
            }
        };
        auto set_tap = [&](int tap) {
            const int r = tap / a.S, q = tap - r * a.S;
            const int dy = r * a.ody, dx = q * a.odx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = (yx0[j] >> 16) + dy, ix = (int)(short)(yx0[j] & 0xffff) + dx;
                int cy, cx;
                const bool ok = map_tap(a, iy, ix, cy, cx);
                pa[j] = ok ? twin + (int64_t)(pixbase[j] + cy * a.Wi + cx) * rowb + kslot * 32 : nullptr;
            }
        };
        const unsigned char* bimg = a.wgt_img + (int64_t)tile_n * (RS * cchunks) * (2 * B_PLANE) + (lw * 2 * TN) * 1024 + lane * 16;
        int issued = 0;
        auto issue = [&](int buf) {
            unsigned char* stage = smem_b + buf * STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned char* src = pa[j] != nullptr ? pa[j] + l_cc * 128 : g_zero16;
                const unsigned char* src_lo = pa[j] != nullptr ? src + 16 : g_zero16;
                unsigned char* dst = stage + (lw * 64 + 16 * j) * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lo,
                                                 (__attribute__((address_space(3))) void*)(dst + A_PLANE), 16, 0, 0);
            }
            const unsigned char* bsrc = bimg + (int64_t)(l_tap * cchunks + l_cc) * (2 * B_PLANE);
            unsigned char* bdst = stage + 2 * A_PLANE + (lw * 2 * TN) * 1024;
#pragma unroll
            for (int c = 0; c < 2 * TN; ++c)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + c * 1024),
                                                 (__attribute__((address_space(3))) void*)(bdst + c * 1024), 16, 0, 0);
            ++issued;
            if (++l_cc == cchunks) {
                l_cc = 0;
                if (issued < ksteps) {
                    next_tap();
                    set_tap(l_tap);
                }
            }
        };
        auto wait_next = [&](bool newest_in_flight) {
            if (newest_in_flight) {
                if constexpr (TN == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        };
        if (RS <= 64) {
            next_tap();
        }
        set_tap(l_tap);
        issue(0);
        if (ksteps > 1) issue(1);
        wait_next(ksteps > 1);
        int nx = 2;
        for (int ks = 0; ks < ksteps; ++ks) {
            const bool ahead = ks + 2 < ksteps;
            if (ahead) issue(nx);
            wait_next(ahead);
            nx = nx == 2 ? 0 : nx + 1;
        }
        return;
    }

    const int wm = wv >> 1, wn = wv & 1;                          // 4 x 2 MFMA waves
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ lds_swz(frow)) << 4);
    const int aoff = wm * 64 * 64 + foff;
    const int boff = 2 * A_PLANE + wn * 32 * TN * 64 + foff;

    __builtin_amdgcn_s_barrier();                                // stage 0 has landed
    int cur = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        const unsigned char* Ah = smem_b + cur * STAGE + aoff;
        const unsigned char* Al = Ah + A_PLANE;
        const unsigned char* Bh = smem_b + cur * STAGE + boff;
        const unsigned char* Bl = Bh + B_PLANE;
        bf16x8_t bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = *reinterpret_cast<const bf16x8_t*>(Bh + j * 1024);
            bl[j] = *reinterpret_cast<const bf16x8_t*>(Bl + j * 1024);
        }
        bf16x8_t fa[MT][2];
        fa[0][0] = *reinterpret_cast<const bf16x8_t*>(Ah);
        fa[0][1] = *reinterpret_cast<const bf16x8_t*>(Al);
        fa[1][0] = *reinterpret_cast<const bf16x8_t*>(Ah + 1024);
        fa[1][1] = *reinterpret_cast<const bf16x8_t*>(Al + 1024);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT + 4, 0);
#pragma unroll
        for (int i = 0; i < MT; i += 2) {
            if (i + 2 < MT) {
                fa[i + 2][0] = *reinterpret_cast<const bf16x8_t*>(Ah + (i + 2) * 1024);
                fa[i + 2][1] = *reinterpret_cast<const bf16x8_t*>(Al + (i + 2) * 1024);
                fa[i + 3][0] = *reinterpret_cast<const bf16x8_t*>(Ah + (i + 3) * 1024);
                fa[i + 3][1] = *reinterpret_cast<const bf16x8_t*>(Al + (i + 3) * 1024);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i + u][1], bh[j], acc[i + u][j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i + u][0], bl[j], acc[i + u][j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i + u][0], bh[j], acc[i + u][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * NT, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }
    __syncthreads();                                             // (8 surviving waves) everyone is out of the ring

    // epilogue: thread group h = wv >> 2 (waves 4h .. 4h+3 = rows 128h .. 128h+127) stages and drains its half
    constexpr int LDS_LD = BN + 4;
    const int h = wv >> 2, t = threadIdx.x & 255;
    float* stage = reinterpret_cast<float*>(smem_b) + h * (128 * LDS_LD);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                stage[((wm & 1) * 64 + i * 16 + (lane >> 4) * 4 + e) * LDS_LD + wn * 32 * TN + j * 16 + (lane & 15)] = acc[i][j][e];
    __syncthreads();
    drain_stage<2, TN, EPI>(stage, a, m0 + h * 128, n0, t, tile_m * 2 + h, m0 + h * 128 < a.M);
}

// ---------------------------------------------------------------------------------------------
// Exact-fp32 forward / backward-data with LDS-DMA operands (round 3): the structure of conv_fwd_x3t8_kernel on the fp32
// matrix cores.  256 x 128 x 32 block tile, 12 waves = 8 MFMA waves as 4 (M) x 2 (N), wave tile 64 x 64 = 2 x 2 tiles of
// v_mfma_f32_32x32x2_f32 (64 accumulator registers), + 4 loader waves (one per SIMD) that only issue
// global_load_lds_dwordx4: activations and weights go global -> LDS with no register staging, no ds_write and no VALU
// on the way, into a three-stage ring (3 x 48 KB; counted vmcnt, one raw barrier per K-step).
//
// Why (conv_fwd_kernel, 128 x 128, register-staged, two LDS buffers, measured on the C2 shapes: matrix pipe 72 % busy):
// a 128 x 128 x 32 fp32 step is 32 FLOP per operand byte, so at the fp32 MFMA peak the CUs pull 4.9 TB/s out of L2 --
// every load passes through VGPRs and is written to LDS by the MFMA waves themselves, and a load has one K-step to
// land.  Here a step is 42.7 FLOP/byte (3.7 TB/s at peak), loads have two K-steps (~16 k cycles) to land, and the MFMA
// waves do nothing but ds_read_b128 + MFMA.
//
// LDS image: rows of 32 floats (128 B, one pixel / output channel x 32 k), unpadded (LDS-DMA writes 1 KB = 8 whole rows
// per wave instruction); the eight 16-byte chunks of row r live at chunk ^ ((r >> 1) & 7) -- applied to the per-lane
// SOURCE address -- so that the ds_read_b128 of a fragment (16 lanes = 16 consecutive rows, same logical chunk) covers
// the 64 banks exactly once.  Fragment / MFMA order per accumulator is conv_fwd_kernel's (k-group t, element e): results
// are bit-identical to it (skipped dead taps only ever added exact zeros).
// ---------------------------------------------------------------------------------------------
#if defined(DIGA_PROBE_STAMP)      // diagnostic build (tools/diag/f32_tile_stamps.py): phase time stamps of MFMA wave 0 of every block
__device__ unsigned long long g_probe_stamps[8192 * 6];
#define DIGA_STAMP(slot_)                                                                                                  \
    do {                                                                                                                   \
        if (wv == 0 && lane == 0 && blockIdx.x < 8192) {                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            g_probe_stamps[blockIdx.x * 6 + (slot_)] = __builtin_amdgcn_s_memrealtime();                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
        }                                                                                                                  \
    } while (0)
#else
#define DIGA_STAMP(slot_) do { } while (0)
#endif

#ifndef DIGA_FWD_DMA_MINW
#define DIGA_FWD_DMA_MINW 3           /* (A/B knob: 4 = a 128-register cap, so that apply-pass waves of the other stream fit next to it) */
#endif
template <bool EPI = false>
__global__ __launch_bounds__(768, DIGA_FWD_DMA_MINW) void conv_fwd_dma_kernel(ConvArgs a) {
    constexpr int BM = 256, BN = 128;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool loader = wv >= 8;
    DIGA_STAMP(0);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RS = a.R * a.S;
    const int cchunks = a.Cin / 32;
    const uint64_t live = RS <= 64 ? live_taps(a, m0, BM) : ~0ull;
    const int ntaps = RS <= 64 ? __builtin_popcountll(live) : RS;
    const int ksteps = ntaps * cchunks;

    if (loader) {
        const int lw = wv - 8;                                   // A rows lw*64 + 8 j + (lane >> 3), B rows lw*32 + 8 c + (lane >> 3)
        const unsigned char* in_b = reinterpret_cast<const unsigned char*>(a.in);
        const int lrow = lane >> 3;
        int pixbase[8], yx0[8];
        bool mok[8];
        const int HoWo = a.Ho * a.Wo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = m0 + lw * 64 + 8 * j + lrow;
            mok[j] = m < a.M;
            const int mm = mok[j] ? m : a.M - 1;
            const int img = mm / HoWo, rem = mm - img * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            pixbase[j] = img * a.Hi * a.Wi;
            yx0[j] = ((ho * a.sy + a.oy0) << 16) | ((wo * a.sx + a.ox0) & 0xffff);
        }
        const int64_t rowb = (int64_t)a.in_ld * 4;
        const unsigned char* pa[8];
        const unsigned char* pb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = lw * 32 + 8 * c + lrow, co = n0 + r;
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            const float* wbase = a.wb_tiles > 0 ? a.wgt + (int64_t)(tile_m / a.wb_tiles) * a.wb_stride : a.wgt;
            pb[c] = co < a.Cout ? reinterpret_cast<const unsigned char*>(wbase) + (int64_t)co * RS * a.Cin * 4 + chunk * 16 : nullptr;
        }
        uint64_t todo = live;
        int l_tap = 0, l_cc = 0;
        auto next_tap = [&]() {
            if (RS <= 64) {
                l_tap = __builtin_ctzll(todo);
                todo &= todo - 1;
            } else {
                ++l_tap;
            }
        };
        auto set_tap = [&](int tap) {
            const int r = tap / a.S, q = tap - r * a.S;
            const int dy = r * a.ody, dx = q * a.odx;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int iy = (yx0[j] >> 16) + dy, ix = (int)(short)(yx0[j] & 0xffff) + dx;
                int cy, cx;
                const bool ok = map_tap(a, iy, ix, cy, cx) && mok[j];
                const int chunk = (lane & 7) ^ (((8 * j + lrow) >> 1) & 7);
                pa[j] = ok ? in_b + (int64_t)(pixbase[j] + cy * a.Wi + cx) * rowb + chunk * 16 : nullptr;
            }
        };
        int issued = 0;
        auto issue = [&](int buf) {
            unsigned char* stage = smem_b + buf * STAGE;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned char* src = pa[j] != nullptr ? pa[j] + l_cc * 128 : g_zero16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + (lw * 64 + 8 * j) * 128), 16, 0, 0);
            }
            const int64_t boff = ((int64_t)l_tap * a.Cin + l_cc * 32) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned char* src = pb[c] != nullptr ? pb[c] + boff : g_zero16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + A_BYTES + (lw * 32 + 8 * c) * 128), 16, 0, 0);
            }
            ++issued;
            if (++l_cc == cchunks) {
                l_cc = 0;
                if (issued < ksteps) {
                    next_tap();
                    set_tap(l_tap);
                }
            }
        };
        auto wait_next = [&](bool newest_in_flight) {            // 12 loads per stage and loader wave
            if (newest_in_flight) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        if (RS <= 64) next_tap();
        set_tap(l_tap);
        issue(0);
        if (ksteps > 1) issue(1);
        wait_next(ksteps > 1);
        int nx = 2;
        for (int ks = 0; ks < ksteps; ++ks) {
            const bool ahead = ks + 2 < ksteps;
            if (ahead) issue(nx);                                // that stage was last read in step ks - 1 (barrier since)
            wait_next(ahead);
            nx = nx == 2 ? 0 : nx + 1;
        }
        return;                                                  // the epilogue's barriers count the surviving waves
    }

    const int wm = wv >> 1, wn = wv & 1;                         // 4 x 2 MFMA waves
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    const int fsw = (li >> 1) & 7;
    int foff[4];                                                 // byte offset of k-group t inside row li
#pragma unroll
    for (int t = 0; t < 4; ++t) foff[t] = li * 128 + (((2 * t + lh) ^ fsw) << 4);
    const int abase = wm * 64 * 128, bbase = A_BYTES + wn * 64 * 128;

    DIGA_STAMP(1);
    __builtin_amdgcn_s_barrier();                                // stage 0 has landed
    DIGA_STAMP(2);
    int cur = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        const unsigned char* As = smem_b + cur * STAGE + abase;
        const unsigned char* Bs = smem_b + cur * STAGE + bbase;
        float4 fa[2][2], fb[2][2];                               // [parity of t][tile]
        fa[0][0] = *reinterpret_cast<const float4*>(As + foff[0]);
        fa[0][1] = *reinterpret_cast<const float4*>(As + foff[0] + 32 * 128);
        fb[0][0] = *reinterpret_cast<const float4*>(Bs + foff[0]);
        fb[0][1] = *reinterpret_cast<const float4*>(Bs + foff[0] + 32 * 128);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = t & 1;
            if (t + 1 < 4) {
                fa[p ^ 1][0] = *reinterpret_cast<const float4*>(As + foff[t + 1]);
                fa[p ^ 1][1] = *reinterpret_cast<const float4*>(As + foff[t + 1] + 32 * 128);
                fb[p ^ 1][0] = *reinterpret_cast<const float4*>(Bs + foff[t + 1]);
                fb[p ^ 1][1] = *reinterpret_cast<const float4*>(Bs + foff[t + 1] + 32 * 128);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float av = e == 0 ? fa[p][i].x : e == 1 ? fa[p][i].y : e == 2 ? fa[p][i].z : fa[p][i].w;
                        const float bv = e == 0 ? fb[p][j].x : e == 1 ? fb[p][j].y : e == 2 ? fb[p][j].z : fb[p][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this step's fragment reads are done before the stage is released
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }
    DIGA_STAMP(3);
    __syncthreads();                                             // (8 surviving waves) everyone is out of the ring

    // epilogue: thread group h = wv >> 2 (rows 128h .. 128h+127) stages and drains its half
    constexpr int LDS_LD = BN + 4;
    const int h = wv >> 2, t = threadIdx.x & 255;
    float* stage = reinterpret_cast<float*>(smem_b) + h * (128 * LDS_LD);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[((wm & 1) * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * LDS_LD + wn * 64 + j * 32 + li] = acc[i][j][e];
    __syncthreads();
    DIGA_STAMP(4);
    drain_stage<2, 2, EPI>(stage, a, m0 + h * 128, n0, t, tile_m * 2 + h, m0 + h * 128 < a.M);
    DIGA_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// backward-weight
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dy;     // [M][Cout] (dy_ld)
    const float* x;      // [N][Hi][Wi][Cin] (x_ld)
    float* slab;         // [splits][Cout][RS][Cin]
    int N, Hi, Wi, Cin, x_ld;
    int Ho, Wo, Cout, dy_ld;
    int R, S, sy, sx, oy0, ox0, ody, odx;
    int M, tiles_m, tiles_n, splits, steps_per_split;
    const int* ptab;     // conv_wgrad_x3w_kernel: [R*S][M_pad] x-pixel index of (tap, output pixel), -1 = outside the image
    const float* zeros;  // 16 bytes of zeros (what an outside tap loads)
    int M_pad;
    // batched products on conv_wgrad_dma_kernel (winograd.hip): "tap" t reads dy + t * dy_tap_stride, x + t * x_tap_stride
    int64_t dy_tap_stride = 0, x_tap_stride = 0;
};

constexpr int kLDW = 128 + 4;   // wgrad LDS rows: [pixel][channel], channel contiguous

template <int TM, int TN>   // block tile (64*TM) couts x (64*TN) cins; wave tile (32*TM) x (32*TN)
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                      // [2][kBK][kLDW]  dy tile   (pixel-major)
    float* Bs = smem + 2 * kBK * kLDW;     // [2][kBK][kLDW]  x tile
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int RS = a.R * a.S;
    // tiles fastest, pixel-range split slowest: blocks that run together (and, after the XCD remap, share an
    // L2) read the same dy / x pixel range for different taps and channel tiles
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;
    wg /= a.tiles_n;
    const int tile_m = wg % a.tiles_m;
    wg /= a.tiles_m;
    const int tap = wg % RS;
    const int split = wg / RS;
    const int k0 = tile_m * BM, c0 = tile_n * BN;

    // loader: 32 pixels x BM (or BN) channels per K-step; thread owns 4-float chunks
    constexpr int CA = BM / 4, CB = BN / 4;          // float4 chunks per pixel row
    constexpr int NA = (kBK * CA) / 256, NB = (kBK * CB) / 256;
    float4 ra[NA], rb[NB];
    const int p_begin = split * a.steps_per_split * kBK;
    int p_end = p_begin + a.steps_per_split * kBK;
    if (p_end > a.M) p_end = a.M;
    const int ksteps = p_end > p_begin ? (p_end - p_begin + kBK - 1) / kBK : 0;

    // x-operand pixel of (tap, output pixel): from the table wgrad_pixtab_kernel built (one int per pixel and tap, -1 outside
    // the image) or, for an identity map (1x1, stride 1, no offset), the output pixel itself -- no divisions or coordinate
    // arithmetic in the loop (round 3: they sat in front of every K-step's MFMAs; matrix pipe 59 % busy).  Indices are
    // fetched one K-step ahead of the loads that use them.
    const int* tab = a.ptab != nullptr ? a.ptab + (int64_t)tap * a.M_pad : nullptr;
    int bidx[NB];
    auto load_idx = [&](int ks) {
        const int pb = p_begin + ks * kBK;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = t + 256 * i;
            const int p = pb + idx / CB;
            bidx[i] = p < p_end ? (tab != nullptr ? tab[p] : p) : -1;
        }
    };
    auto gload = [&](int ks) {
        const int pb = p_begin + ks * kBK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = t + 256 * i;
            const int pr = idx / CA, ch = (idx - pr * CA) * 4;
            const int p = pb + pr;
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < p_end && k0 + ch < a.Cout)
                ra[i] = *reinterpret_cast<const float4*>(a.dy + (int64_t)p * a.dy_ld + k0 + ch);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = t + 256 * i;
            const int ch = (idx % CB) * 4;
            rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bidx[i] >= 0 && c0 + ch < a.Cin)
                rb[i] = *reinterpret_cast<const float4*>(a.x + (int64_t)bidx[i] * a.x_ld + c0 + ch);
        }
    };
    auto lstore = [&](int buf) {
        float* ad = As + buf * kBK * kLDW;
        float* bd = Bs + buf * kBK * kLDW;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = t + 256 * i;
            const int pr = idx / CA, ch = (idx - pr * CA) * 4;
            *reinterpret_cast<float4*>(ad + pr * kLDW + ch) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = t + 256 * i;
            const int pr = idx / CB, ch = (idx - pr * CB) * 4;
            *reinterpret_cast<float4*>(bd + pr * kLDW + ch) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    if (ksteps > 0) {
        load_idx(0);
        gload(0);
        lstore(0);
        if (ksteps > 1) load_idx(1);
    }
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) {
            gload(ks + 1);
            if (ks + 2 < ksteps) load_idx(ks + 2);
        }
        const float* Ac = As + cur * kBK * kLDW;
        const float* Bc = Bs + cur * kBK * kLDW;
        // fragment reads run one k-pair AHEAD of their MFMAs in a second register set (the compiler's own schedule read each
        // pair into the same registers right before its four MFMAs and waited for LDS every time: matrix pipe 59 % busy)
        float av[2][TM], bv[2][TN];
        auto frag = [&](int kk, int set) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[set][i] = Ac[(kk + lh) * kLDW + wm * 32 * TM + i * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[set][j] = Bc[(kk + lh) * kLDW + wn * 32 * TN + j * 32 + li];
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const int set = (kk >> 1) & 1;
            if (kk + 2 < kBK) frag(kk + 2, set ^ 1);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);           // the next pair's LDS reads first ...
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][i], bv[set][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);           // ... then this pair's MFMAs
        }
        if (ks + 1 < ksteps) lstore(cur ^ 1);
        __syncthreads();
    }

    float* out = a.slab + (int64_t)split * a.Cout * RS * a.Cin;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + wn * 32 * TN + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (k < a.Cout && c < a.Cin) out[((int64_t)k * RS + tap) * a.Cin + c] = acc[i][j][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weight with split-bf16 operands.  The contraction index is the pixel, but both operands sit
// pixel-major in memory ([pixel][channel]); a thread therefore loads a 4-pixel x 4-channel block (4 float4),
// transposes it in registers and writes, per channel, the 4 consecutive pixels as 8 B of hi and 8 B of lo
// into LDS rows [channel][32 pixels]: the MFMA fragment reads are then exactly those of conv_fwd_x3_kernel.
// ---------------------------------------------------------------------------------------------
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(WgradArgs a) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_PLANE = BM * kRowB, B_PLANE = BN * kRowB, BUF = 2 * A_PLANE + 2 * B_PLANE;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int RS = a.R * a.S;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;
    wg /= a.tiles_n;
    const int tile_m = wg % a.tiles_m;
    wg /= a.tiles_m;
    const int tap = wg % RS;
    const int split = wg / RS;
    const int r = tap / a.S, s = tap - r * a.S;
    const int dyo = a.oy0 + r * a.ody, dxo = a.ox0 + s * a.odx;
    const int k0 = tile_m * BM, c0 = tile_n * BN;
    const int HoWo = a.Ho * a.Wo;

    // loader: thread -> (channel quad cq, pixel quad pq) of each operand tile (BM or BN channels x 32 pixels)
    const int pq = t & 7, cq = t >> 3;                // 8 pixel quads x 32 channel quads
    const bool a_on = cq < BM / 4, b_on = cq < BN / 4;
    const int p_begin = split * a.steps_per_split * kBK;
    int p_end = p_begin + a.steps_per_split * kBK;
    if (p_end > a.M) p_end = a.M;
    const int ksteps = p_end > p_begin ? (p_end - p_begin + kBK - 1) / kBK : 0;
    const int ka = min(k0 + cq * 4, a.Cout - 4), cb = min(c0 + cq * 4, a.Cin - 4);

    float4 ra[4], rb[4];
    float fa[4], fb[4];
    // General loader (any geometry, ragged last step): per pixel two integer divisions and clamps -- ~90 VALU
    // instructions per pixel, 3x the matrix-pipe time of a K-step.  Used for the first step, the ragged last step
    // and tiny maps only.
    auto gload_slow = [&](int ks) {
        const int pb = p_begin + ks * kBK + pq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = pb + j;
            const bool inr = p < p_end;
            const int pc = inr ? p : p_begin;            // clamped: always a valid pixel
            ra[j] = *reinterpret_cast<const float4*>(a.dy + (int64_t)pc * a.dy_ld + ka);
            fa[j] = (inr && a_on && k0 + cq * 4 < a.Cout) ? 1.f : 0.f;
            const int img = pc / HoWo, rem = pc - img * HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const int iy = ho * a.sy + dyo, ix = wo * a.sx + dxo;
            const bool ok = inr && b_on && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi &&
                            c0 + cq * 4 < a.Cin;
            const int cy = min(max(iy, 0), a.Hi - 1), cx = min(max(ix, 0), a.Wi - 1);
            rb[j] = *reinterpret_cast<const float4*>(a.x + (int64_t)((img * a.Hi + cy) * a.Wi + cx) * a.x_ld + cb);
            fb[j] = ok ? 1.f : 0.f;
        }
    };
    // Fast loader (full steps of maps at least 32 wide): the thread's first pixel is tracked as (image, row, column)
    // and advanced by 32 per step; its three neighbours are derived with one wrap test each.  Only the x operand is
    // masked (a zero on either side kills the product; rows / columns beyond Cout / Cin are never stored), and a
    // pointwise layer (1x1, stride 1, no offset: x pixel == dy pixel) needs neither coordinates nor a mask.
    const bool pointwise = RS == 1 && a.sy == 1 && a.sx == 1 && dyo == 0 && dxo == 0 && a.Hi == a.Ho && a.Wi == a.Wo;
    // (measured on the C2 layer shapes: pointwise layers +7..10 % with the fast loader, 3x3 layers -5..10 % -- their
    // coordinate arithmetic in front of the pinned load slots delays the loads -- so only pointwise layers take it)
    const bool fast_ok = pointwise && a.Wo >= kBK && a.Ho >= 2;
    int f_img, f_ho, f_wo;        // coordinates of pixel f_p = p_begin + (step) * 32 + pq * 4
    int64_t f_p;
    auto fast_init = [&](int ks) {
        f_p = p_begin + ks * kBK + pq * 4;
        const int pc = (int)min((int64_t)a.M - 1, f_p);
        f_img = pc / HoWo;
        const int rem = pc - f_img * HoWo;
        f_ho = rem / a.Wo;
        f_wo = rem - f_ho * a.Wo;
    };
    auto gload_fast = [&]() {
        // addresses first (the only control flow), then the eight loads in straight-line code so that they can be
        // scheduled between the MFMAs of the step
        const float* xa[4];
        if (pointwise) {
            const float* xp = a.x + f_p * a.x_ld + cb;
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[j] = xp + (int64_t)j * a.x_ld;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int wo = f_wo + j, ho = f_ho, img = f_img;
                const bool wrap = wo >= a.Wo;
                wo = wrap ? wo - a.Wo : wo;
                ho = wrap ? ho + 1 : ho;
                const bool wrap2 = ho >= a.Ho;
                ho = wrap2 ? 0 : ho;
                img = wrap2 ? img + 1 : img;
                const int iy = ho * a.sy + dyo, ix = wo * a.sx + dxo;
                const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
                const int cy = min(max(iy, 0), a.Hi - 1), cx = min(max(ix, 0), a.Wi - 1);
                xa[j] = a.x + (int64_t)((img * a.Hi + cy) * a.Wi + cx) * a.x_ld + cb;
                fb[j] = ok ? 1.f : 0.f;
            }
        }
        const float* dyp = a.dy + f_p * a.dy_ld + ka;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const float4*>(dyp + (int64_t)j * a.dy_ld);
            rb[j] = *reinterpret_cast<const float4*>(xa[j]);
        }
        f_p += kBK;
        f_wo += kBK;
        const bool w1 = f_wo >= a.Wo;
        f_wo = w1 ? f_wo - a.Wo : f_wo;
        f_ho = w1 ? f_ho + 1 : f_ho;
        const bool w2 = f_ho >= a.Ho;
        f_ho = w2 ? 0 : f_ho;
        f_img = w2 ? f_img + 1 : f_img;
    };
    // transposing stores: per channel e of the thread's channel quad, its 4 consecutive pixels -> 8 B hi + 8 B lo
    auto lstore = [&](int buf, bool fast) {
        unsigned char* base = smem_b + buf * BUF;
        if (a_on) {
            const float va[4][4] = {{ra[0].x, ra[1].x, ra[2].x, ra[3].x}, {ra[0].y, ra[1].y, ra[2].y, ra[3].y},
                                    {ra[0].z, ra[1].z, ra[2].z, ra[3].z}, {ra[0].w, ra[1].w, ra[2].w, ra[3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint2 hi, lo;
                if (fast) split4_nomask(make_float4(va[e][0], va[e][1], va[e][2], va[e][3]), hi, lo);
                else split4_nomask(make_float4(va[e][0] * fa[0], va[e][1] * fa[1], va[e][2] * fa[2], va[e][3] * fa[3]), hi, lo);
                const int off = (cq * 4 + e) * kRowB + pq * 8;
                *reinterpret_cast<uint2*>(base + off) = hi;
                *reinterpret_cast<uint2*>(base + A_PLANE + off) = lo;
            }
        }
        if (b_on) {
            const float vb[4][4] = {{rb[0].x, rb[1].x, rb[2].x, rb[3].x}, {rb[0].y, rb[1].y, rb[2].y, rb[3].y},
                                    {rb[0].z, rb[1].z, rb[2].z, rb[3].z}, {rb[0].w, rb[1].w, rb[2].w, rb[3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint2 hi, lo;
                if (fast && pointwise) split4_nomask(make_float4(vb[e][0], vb[e][1], vb[e][2], vb[e][3]), hi, lo);
                else split4_nomask(make_float4(vb[e][0] * fb[0], vb[e][1] * fb[1], vb[e][2] * fb[2], vb[e][3] * fb[3]), hi, lo);
                const int off = (cq * 4 + e) * kRowB + pq * 8;
                *reinterpret_cast<uint2*>(base + 2 * A_PLANE + off) = hi;
                *reinterpret_cast<uint2*>(base + 2 * A_PLANE + B_PLANE + off) = lo;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    auto compute = [&](int cur, bool with_loads) {
        const unsigned char* Ah = smem_b + cur * BUF;
        const unsigned char* Al = Ah + A_PLANE;
        const unsigned char* Bh = Ah + 2 * A_PLANE;
        const unsigned char* Bl = Bh + B_PLANE;
#pragma unroll
        for (int sl = 0; sl < kBK / 16; ++sl) {
            bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = (wm * 32 * TM + i * 32 + li) * kRowB + sl * 32 + lh * 16;
                ah[i] = *reinterpret_cast<const bf16x8_t*>(Ah + off);
                al[i] = *reinterpret_cast<const bf16x8_t*>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = (wn * 32 * TN + j * 32 + li) * kRowB + sl * 32 + lh * 16;
                bh[j] = *reinterpret_cast<const bf16x8_t*>(Bh + off);
                bl[j] = *reinterpret_cast<const bf16x8_t*>(Bl + off);
            }
            // issue order: this sub-step's fragment reads, half of the next step's global loads, then the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            if (with_loads) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM * TN, 0);
        }
    };

    if (ksteps > 0) {
        gload_slow(0);
        lstore(0, false);
    }
    __syncthreads();
    const bool ragged = (p_end - p_begin) % kBK != 0;
    // steps [0, nfast) prefetch their successor with the fast loader in the same basic block as the MFMAs
    const int nfast = fast_ok ? max(0, ksteps - (ragged ? 2 : 1)) : 0;
    if (nfast > 0) fast_init(1);
    int ks = 0;
    for (; ks < nfast; ++ks) {
        const int cur = ks & 1;
        gload_fast();
        compute(cur, true);
        lstore(cur ^ 1, true);
        __syncthreads();
    }
    for (; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) gload_slow(ks + 1);
        compute(cur, false);
        if (ks + 1 < ksteps) lstore(cur ^ 1, false);
        __syncthreads();
    }

    float* out = a.slab + (int64_t)split * a.Cout * RS * a.Cin;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + wn * 32 * TN + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (k < a.Cout && c < a.Cin) out[((int64_t)k * RS + tap) * a.Cin + c] = acc[i][j][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weight, split-bf16, wide variant: 256 output channels x (64*TN) input channels per block (wave tile
// 128 x 32*TN), one LDS buffer, two blocks per CU.  Both operands are split in-kernel, so per MFMA the 128^2 kernel
// does twice the split arithmetic of the forward kernel and is bound by the vector-issue port (it reaches 0.23 of
// the MFMA peak); doubling the Cout extent of the tile halves the x-operand work per MFMA: per 32-pixel K-step a
// thread stages 48 values for 48 MFMAs instead of 32 for 24.
// The (tap, output pixel) -> input pixel map comes from a table built by wgrad_pixtab_kernel (one int per pixel and
// tap, -1 outside the image), so the loader has no divisions, no coordinate arithmetic and no masks: an outside
// tap or a pixel past the block's range loads 16 bytes of zeros for the x operand (a zero on one side kills the
// product; rows / columns beyond Cout / Cin are never stored).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_pixtab_kernel(int* __restrict__ tab, float* __restrict__ zeros, int M,
                                                           int M_pad, int Ho, int Wo, int Hi, int Wi, int S, int sy, int sx,
                                                           int oy0, int ox0, int ody, int odx) {
    const int p = blockIdx.x * 256 + threadIdx.x, tap = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) zeros[threadIdx.x] = 0.f;
    if (p >= M_pad) return;
    int v = -1;
    if (p < M) {
        const int r = tap / S, s = tap - r * S;
        const int img = p / (Ho * Wo), rem = p - img * Ho * Wo;
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const int iy = ho * sy + oy0 + r * ody, ix = wo * sx + ox0 + s * odx;
        if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) v = (img * Hi + iy) * Wi + ix;
    }
    tab[(int64_t)tap * M_pad + p] = v;
}

template <int TN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3w_kernel(WgradArgs a) {
    constexpr int BM = 256, BN = 64 * TN, TM = 4;
    constexpr int A_PLANE = BM * kRowB, B_PLANE = BN * kRowB;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int RS = a.R * a.S;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;
    wg /= a.tiles_n;
    const int tile_m = wg % a.tiles_m;
    wg /= a.tiles_m;
    const int tap = wg % RS;
    const int split = wg / RS;
    const int k0 = tile_m * BM, c0 = tile_n * BN;

    const int pq = t & 7, cq = t >> 3;                // 8 pixel quads x 32 channel quads (two passes over dy)
    const bool b_on = cq < BN / 4;
    const int p_begin = split * a.steps_per_split * kBK;
    int p_end = p_begin + a.steps_per_split * kBK;
    if (p_end > a.M) p_end = a.M;
    const int ksteps = p_end > p_begin ? (p_end - p_begin + kBK - 1) / kBK : 0;
    const int ka0 = min(k0 + cq * 4, a.Cout - 4), ka1 = min(k0 + 128 + cq * 4, a.Cout - 4);
    const int cb = min(c0 + cq * 4, a.Cin - 4);
    const int* tab = a.ptab + (int64_t)tap * a.M_pad;

    float4 ra[2][4], rb[4];
    int4 idx;                       // x-pixel indices of the quad that the next gload() fetches
    auto load_idx = [&](int p) { idx = *reinterpret_cast<const int4*>(tab + min(p, a.M_pad - 4)); };
    auto gload = [&](int p) {       // p = first pixel of this thread's quad
        const int xi[4] = {idx.x, idx.y, idx.z, idx.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* dyp = a.dy + (int64_t)min(p + j, a.M - 1) * a.dy_ld;
            ra[0][j] = *reinterpret_cast<const float4*>(dyp + ka0);
            ra[1][j] = *reinterpret_cast<const float4*>(dyp + ka1);
            const bool ok = xi[j] >= 0 && p + j < p_end;
            const float* xp = ok ? a.x + (int64_t)xi[j] * a.x_ld + cb : a.zeros;
            rb[j] = *reinterpret_cast<const float4*>(xp);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float va[4][4] = {{ra[i][0].x, ra[i][1].x, ra[i][2].x, ra[i][3].x}, {ra[i][0].y, ra[i][1].y, ra[i][2].y, ra[i][3].y},
                                    {ra[i][0].z, ra[i][1].z, ra[i][2].z, ra[i][3].z}, {ra[i][0].w, ra[i][1].w, ra[i][2].w, ra[i][3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint2 hi, lo;
                split4_nomask(make_float4(va[e][0], va[e][1], va[e][2], va[e][3]), hi, lo);
                const int off = (i * 128 + cq * 4 + e) * kRowB + pq * 8;
                *reinterpret_cast<uint2*>(smem_b + off) = hi;
                *reinterpret_cast<uint2*>(smem_b + A_PLANE + off) = lo;
            }
        }
        if (b_on) {
            const float vb[4][4] = {{rb[0].x, rb[1].x, rb[2].x, rb[3].x}, {rb[0].y, rb[1].y, rb[2].y, rb[3].y},
                                    {rb[0].z, rb[1].z, rb[2].z, rb[3].z}, {rb[0].w, rb[1].w, rb[2].w, rb[3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint2 hi, lo;
                split4_nomask(make_float4(vb[e][0], vb[e][1], vb[e][2], vb[e][3]), hi, lo);
                const int off = (cq * 4 + e) * kRowB + pq * 8;
                *reinterpret_cast<uint2*>(smem_b + 2 * A_PLANE + off) = hi;
                *reinterpret_cast<uint2*>(smem_b + 2 * A_PLANE + B_PLANE + off) = lo;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const unsigned char* Ab = smem_b + (wm * 128 + li) * kRowB + lh * 16;
    const unsigned char* Bb = smem_b + 2 * A_PLANE + (wn * 32 * TN + li) * kRowB + lh * 16;
    int p = p_begin + pq * 4;
    if (ksteps > 0) {
        load_idx(p);
        gload(p);
        load_idx(p + kBK);
        lstore();
    }
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        p = more ? p + kBK : p;          // the last step re-reads its own pixels (loads stay unconditional so that they
        gload(p);                        // share a basic block with the MFMAs)
        load_idx(p + kBK);
#pragma unroll
        for (int sl = 0; sl < kBK / 16; ++sl) {
            bf16x8_t bh[TN], bl[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8_t*>(Bb + j * 32 * kRowB + sl * 32);
                bl[j] = *reinterpret_cast<const bf16x8_t*>(Bb + B_PLANE + j * 32 * kRowB + sl * 32);
            }
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                bf16x8_t ah[2], al[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ah[u] = *reinterpret_cast<const bf16x8_t*>(Ab + (ip * 2 + u) * 32 * kRowB + sl * 32);
                    al[u] = *reinterpret_cast<const bf16x8_t*>(Ab + A_PLANE + (ip * 2 + u) * 32 * kRowB + sl * 32);
                }
                // issue order: fragment reads, a quarter of the next step's 13 global loads, MFMAs
                if (ip == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * TN, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                if (sl == 0 && ip == 0) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
                else __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[ip * 2 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bh[j], acc[ip * 2 + u][j], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[ip * 2 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bl[j], acc[ip * 2 + u][j], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[ip * 2 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bh[j], acc[ip * 2 + u][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * TN, 0);
            }
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    float* out = a.slab + (int64_t)split * a.Cout * RS * a.Cin;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = c0 + wn * 32 * TN + j * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (k < a.Cout && c < a.Cin) out[((int64_t)k * RS + tap) * a.Cin + c] = acc[i][j][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weight on the split twins (multi-tap layers): both operands pre-split ([pixel][8-channel group][hi8 | lo8]
// twins of dy and of x), staged global -> LDS by LDS-DMA loads into a three-stage ring by four loader waves while four
// MFMA waves compute -- the structure of conv_fwd_x3t_kernel.  The contraction index is the pixel and the twins are
// pixel-major, so the LDS image is [32 pixels][channels] per plane and the MFMA fragments (8 consecutive pixels of one
// channel per lane) are read with the transposing ds_read_b64_tr_b16: per 16-lane group a block of 4 pixel rows x 16
// channels, lane i receiving channel i of the 4 rows (tools/experiments/tr_read_probe.hip), two reads per fragment.
// 16-byte chunks (8 channels) of pixel row r are stored at chunk ^ (key(r) << 1), key = (r & 3) | ((r >> 3) & 1) << 2,
// so that the 8 rows a 32-lane half touches fall on 8 different 32-byte positions of the 256-byte bank row; with
// LDS-DMA the permutation is applied to the source address.  256 (Cout) x 128 (Cin) tile per (tap, pixel range).
// ---------------------------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tr_key(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(512, 1) void conv_wgrad_x3t_kernel(WgradArgs a) {
    constexpr int BM = 256, BN = 128, MT = 8, NT = 4;
    constexpr int A_ROW = BM * 2, B_ROW = BN * 2;                       // bytes per pixel row and plane
    constexpr int A_PLANE = kBK * A_ROW, B_PLANE = kBK * B_ROW, STAGE = 2 * A_PLANE + 2 * B_PLANE;   // 48 KB
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int t = threadIdx.x & 255, lane = t & 63, wv = t >> 6;
    const bool loader = threadIdx.x >= 256;
    const int wm = wv >> 1, wn = wv & 1;
    const int RS = a.R * a.S;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;
    wg /= a.tiles_n;
    const int tile_m = wg % a.tiles_m;
    wg /= a.tiles_m;
    const int tap = wg % RS;
    const int split = wg / RS;
    const int k0 = tile_m * BM, c0 = tile_n * BN;
    const int p_begin = split * a.steps_per_split * kBK;
    int p_end = p_begin + a.steps_per_split * kBK;
    if (p_end > a.M) p_end = a.M;
    const int ksteps = p_end > p_begin ? (p_end - p_begin + kBK - 1) / kBK : 0;

    if (loader) {
        const unsigned char* dyt = reinterpret_cast<const unsigned char*>(a.dy);
        const unsigned char* xt = reinterpret_cast<const unsigned char*>(a.x);
        const int64_t dy_rowb = (int64_t)a.Cout * 4, x_rowb = (int64_t)a.Cin * 4;
        const int* tab = a.ptab + (int64_t)tap * a.M_pad;
        // A (dy): 32 rows x 32 chunks per plane = 16 LDS-DMA instructions, 4 per wave: instruction j of wave wv covers
        //         rows 8 wv + 2 j + (lane >> 5), destination chunk lane & 31.   B (x): 32 rows x 16 chunks = 8
        //         instructions per plane, 2 per wave: rows 8 wv + 4 j + (lane >> 4), destination chunk lane & 15.
        const int a_chunk_dst = lane & 31, b_chunk_dst = lane & 15;
        const int kgrp = k0 / 8, cgrp = c0 / 8, kmax = a.Cout / 8 - 1, cmax = a.Cin / 8 - 1;
        int bidx[2];                       // x-pixel indices of this lane's two B rows for the step being issued
        auto load_idx = [&](int ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = p_begin + ks * kBK + 8 * wv + 4 * j + (lane >> 4);
                bidx[j] = p < p_end ? tab[p] : -1;
            }
        };
        auto issue = [&](int ks, int stg) {
            unsigned char* stage = smem_b + stg * STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * wv + 2 * j + (lane >> 5);
                const int p = min(p_begin + ks * kBK + row, a.M - 1);
                const int chunk = min(kgrp + (a_chunk_dst ^ (tr_key(row) << 1)), kmax);
                const unsigned char* src = dyt + p * dy_rowb + (int64_t)chunk * 32;
                unsigned char* dst = stage + (8 * wv + 2 * j) * A_ROW;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16),
                                                 (__attribute__((address_space(3))) void*)(dst + A_PLANE), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 8 * wv + 4 * j + (lane >> 4);
                const int chunk = min(cgrp + (b_chunk_dst ^ (tr_key(row) << 1)), cmax);
                const unsigned char* src = bidx[j] >= 0 ? xt + bidx[j] * x_rowb + (int64_t)chunk * 32 : g_zero16;
                const unsigned char* src_lo = bidx[j] >= 0 ? src + 16 : g_zero16;
                unsigned char* dst = stage + 2 * A_PLANE + (8 * wv + 4 * j) * B_ROW;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lo,
                                                 (__attribute__((address_space(3))) void*)(dst + B_PLANE), 16, 0, 0);
            }
        };
        auto wait_next = [&](bool newest_in_flight) {      // 12 LDS-DMA loads per stage and wave
            if (newest_in_flight) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        if (ksteps > 0) {
            load_idx(0);
            issue(0, 0);
            if (ksteps > 1) {
                load_idx(1);
                issue(1, 1);
            }
        }
        wait_next(ksteps > 1);
        int nx = 2;
        for (int ks = 0; ks < ksteps; ++ks) {
            const bool ahead = ks + 2 < ksteps;
            if (ahead) {
                load_idx(ks + 2);
                issue(ks + 2, nx);
            }
            wait_next(ahead);
            nx = nx == 2 ? 0 : nx + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // transposing read of this lane: group g = lane >> 4 takes pixel rows 8 g + q (+ 4 for the second read), q = (lane & 15) >> 2,
    // channels 16 tile + 4 p, p = lane & 3: chunk = 2 tile + (p >> 1) (swizzled), byte (p & 1) * 8 inside the chunk
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int row0 = 8 * g + q, row1 = row0 + 4;
    const int key0 = tr_key(row0) << 1, key1 = tr_key(row1) << 1;
    auto a_off = [&](int tile, int row, int key) { return row * A_ROW + (((2 * (wm * 8 + tile) + (pp >> 1)) ^ key) << 4) + ((pp & 1) << 3); };
    auto b_off = [&](int tile, int row, int key) { return row * B_ROW + (((2 * (wn * 4 + tile) + (pp >> 1)) ^ key) << 4) + ((pp & 1) << 3); };

    __builtin_amdgcn_s_barrier();                          // stage 0 has landed
    int cur = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        const unsigned char* Ah = smem_b + cur * STAGE;
        const unsigned char* Al = Ah + A_PLANE;
        const unsigned char* Bh = Ah + 2 * A_PLANE;
        const unsigned char* Bl = Bh + B_PLANE;
        bf16x8_t bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = tr_frag(Bh + b_off(j, row0, key0), Bh + b_off(j, row1, key1));
            bl[j] = tr_frag(Bl + b_off(j, row0, key0), Bl + b_off(j, row1, key1));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bf16x8_t ah = tr_frag(Ah + a_off(i, row0, key0), Ah + a_off(i, row1, key1));
            const bf16x8_t al = tr_frag(Al + a_off(i, row0, key0), Al + a_off(i, row1, key1));
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }

    float* out = a.slab + (int64_t)split * a.Cout * RS * a.Cin;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int c = c0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = k0 + wm * 128 + i * 16 + (lane >> 4) * 4 + e;
                if (kk < a.Cout && c < a.Cin) out[((int64_t)kk * RS + tap) * a.Cin + c] = acc[i][j][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Exact-fp32 backward-weight with LDS-DMA operands (round 3): the structure of conv_fwd_dma_kernel / conv_wgrad_x3t_kernel.
// 256 (Cout) x 128 (Cin) tile per (tap, pixel range); 8 MFMA waves as 4 x 2 (wave tile 64 x 64 = 2 x 2 tiles of
// v_mfma_f32_32x32x2_f32) + 4 loader waves; K-step = 32 pixels; three-stage ring of 48 KB stages.  Both operands are
// pixel-major in memory and stay so in LDS ([pixel][channel], 1 KB / 512 B rows written whole by the LDS-DMA loads):
// an MFMA operand (one k = pixel per lane half, 32 consecutive channels over the lanes) is a ds_read_b32 with an
// immediate offset per k-pair.  Rows of odd pixels are stored with their 128-byte halves swapped pairwise (byte ^ 128,
// applied to the source address), so that the lanes of the two halves (pixels kk, kk + 1) hit the two halves of the bank row.
// Needs Cout % 256 == 0 and Cin % 128 == 0 (every layer of layer3 / layer4 / the ASPP head); the rest stays on
// conv_wgrad_kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768, 3) void conv_wgrad_dma_kernel(WgradArgs a) {
    constexpr int BM = 256, BN = 128;
    constexpr int A_ROW = BM * 4, B_ROW = BN * 4;
    constexpr int A_BYTES = kBK * A_ROW, B_BYTES = kBK * B_ROW, STAGE = A_BYTES + B_BYTES;      // 48 KB
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool loader = wv >= 8;
    const int RS = a.R * a.S;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n;
    wg /= a.tiles_n;
    const int tile_m = wg % a.tiles_m;
    wg /= a.tiles_m;
    const int tap = wg % RS;
    const int split = wg / RS;
    const int k0 = tile_m * BM, c0 = tile_n * BN;
    const int p_begin = split * a.steps_per_split * kBK;
    int p_end = p_begin + a.steps_per_split * kBK;
    if (p_end > a.M) p_end = a.M;
    const int ksteps = p_end > p_begin ? (p_end - p_begin + kBK - 1) / kBK : 0;

    if (loader) {
        const int lw = wv - 8;
        const unsigned char* dyb = reinterpret_cast<const unsigned char*>(a.dy) + ((int64_t)k0 + tap * a.dy_tap_stride) * 4;
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + ((int64_t)c0 + tap * a.x_tap_stride) * 4;
        const int64_t dy_rowb = (int64_t)a.dy_ld * 4, x_rowb = (int64_t)a.x_ld * 4;
        const int* tab = a.ptab != nullptr ? a.ptab + (int64_t)tap * a.M_pad : nullptr;
        // A (dy): 32 pixel rows x 1 KB = 32 instructions, 8 per wave: row 8 lw + j, destination chunk `lane`.
        // B (x):  32 pixel rows x 512 B = 16 instructions, 4 per wave: rows 8 lw + 2 j + (lane >> 5), chunk lane & 31.
        const int b_half = lane >> 5;
        const int b_src = (((lane & 31) ^ (b_half << 3)) << 4);
        int bidx[4];
        auto load_idx = [&](int ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = p_begin + ks * kBK + 8 * lw + 2 * j + b_half;
                bidx[j] = p < p_end ? (tab != nullptr ? tab[p] : p) : -1;
            }
        };
        auto issue = [&](int ks, int stg) {
            unsigned char* stage = smem_b + stg * STAGE;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 8 * lw + j;
                const int p = min(p_begin + ks * kBK + row, a.M - 1);     // rows past p_end meet zero x rows
                const unsigned char* src = dyb + p * dy_rowb + ((lane ^ ((j & 1) << 3)) << 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + row * A_ROW), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned char* src = bidx[j] >= 0 ? xb + bidx[j] * x_rowb + b_src : g_zero16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + A_BYTES + (8 * lw + 2 * j) * B_ROW), 16, 0, 0);
            }
        };
        auto wait_next = [&](bool newest_in_flight) {      // 12 LDS-DMA loads per stage and wave
            if (newest_in_flight) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        if (ksteps > 0) {
            load_idx(0);
            issue(0, 0);
            if (ksteps > 1) {
                load_idx(1);
                issue(1, 1);
            }
        }
        wait_next(ksteps > 1);
        int nx = 2;
        for (int ks = 0; ks < ksteps; ++ks) {
            const bool ahead = ks + 2 < ksteps;
            if (ahead) {
                load_idx(ks + 2);
                issue(ks + 2, nx);
            }
            wait_next(ahead);
            nx = nx == 2 ? 0 : nx + 1;
        }
        return;
    }

    const int wm = wv >> 1, wn = wv & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    int offa[2], offb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offa[i] = lh * A_ROW + (((wm * 64 + i * 32 + li) * 4) ^ (lh << 7));
        offb[i] = A_BYTES + lh * B_ROW + (((wn * 64 + i * 32 + li) * 4) ^ (lh << 7));
    }

    __builtin_amdgcn_s_barrier();                          // stage 0 has landed
    int cur = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        const unsigned char* St = smem_b + cur * STAGE;
        float av[2][2], bv[2][2];
        auto frag = [&](int kk, int set) {
#pragma unroll
            for (int i = 0; i < 2; ++i) av[set][i] = *reinterpret_cast<const float*>(St + offa[i] + kk * A_ROW);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[set][j] = *reinterpret_cast<const float*>(St + offb[j] + kk * B_ROW);
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const int set = (kk >> 1) & 1;
            if (kk + 2 < kBK) frag(kk + 2, set ^ 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][i], bv[set][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }

    float* out = a.slab + (int64_t)split * a.Cout * RS * a.Cin;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = c0 + wn * 64 + j * 32 + li;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                out[((int64_t)k * RS + tap) * a.Cin + c] = acc[i][j][e];
            }
        }
    }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                          int64_t n4, int splits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4*>(slab)[i];
    for (int k = 1; k < splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(slab)[(int64_t)k * n4 + i];
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    reinterpret_cast<float4*>(dw)[i] = s;
}

// w [K][RS][C] -> wt [C][RS][K]   (32x32 LDS tile per tap)
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                               int K, int RS, int C) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (k < K && c < C) ? w[((int64_t)k * RS + tap) * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, k = k0 + tx;
        if (c < C && k < K) wt[((int64_t)c * RS + tap) * K + k] = tile[tx][ty + 8 * i];
    }
}

// process-wide arithmetic of the forward / backward-data kernels (diga_set_conv_math); the default can be
// chosen with DIGA_CONV_MATH=bf16x3 in the environment
static int check_conv_common(const char* who, int64_t Cin, int64_t in_ld, int64_t out_ld, int64_t Cout,
                             const void* in, const void* w, const void* out) {
    DIGA_REQUIRE(Cin > 0 && Cin % 32 == 0, DIGA_EINVAL, "%s: Cin=%lld must be a multiple of 32", who, (long long)Cin);
    DIGA_REQUIRE(in_ld >= Cin && in_ld % 4 == 0 && out_ld >= Cout, DIGA_EINVAL, "%s: bad leading dimensions", who);
    DIGA_REQUIRE(aligned16(in) && aligned16(w), DIGA_EALIGN, "%s: pointers must be 16-byte aligned", who);
    (void)out;
    return DIGA_OK;
}

// input-map / activation options of a forward convolution (include/diga_hip.h, diga_conv_options_t); nullptr = none
static void set_options(ConvArgs& a, const diga_conv_options_t* o) {
    a.pad_reflect = o ? o->reflect_pad : 0;
    a.up_shift = o ? o->upsample_shift : 0;
    a.act = o ? o->activation : 0;
}

static int check_options(const diga_conv_options_t* o, const char* who) {
    if (o == nullptr) return DIGA_OK;
    DIGA_REQUIRE(o->reflect_pad >= 0 && o->reflect_pad <= 1 && o->upsample_shift >= 0 && o->upsample_shift <= 2 && o->activation >= 0 &&
                     o->activation <= 1, DIGA_EINVAL, "%s: bad options (reflect_pad 0/1, upsample_shift 0..2, activation 0/1)", who);
    return DIGA_OK;
}

// fills the backward-epilogue fields of ConvArgs from the public descriptor (nullptr = plain convolution)
static int set_bwd_epilogue(ConvArgs& a, const diga_bwd_epilogue_t* e, const char* who) {
    a.e_add = a.e_masky = a.e_x = a.e_relu_ab = a.e_mean = a.e_invstd = nullptr;
    a.e_partials = nullptr;
    a.e_maskbits = nullptr;
    a.e_add_ld = a.e_masky_ld = a.e_x_ld = a.e_maskbits_ld = 0;
    if (e == nullptr) return DIGA_OK;
    DIGA_REQUIRE(e->addend || e->mask_y || e->mask_bits || e->x, DIGA_EINVAL, "%s: empty epilogue descriptor", who);
    DIGA_REQUIRE(a.Cout % 4 == 0 && a.out_ld % 4 == 0 && aligned16(a.out) && a.bias == nullptr && a.stats == nullptr, DIGA_EINVAL,
                 "%s: a backward epilogue needs Cout %% 4 == 0, out_ld %% 4 == 0, a 16-byte aligned output, no bias, no forward statistics", who);
    DIGA_REQUIRE(!e->addend || (aligned16(e->addend) && e->addend_ld >= a.Cout && e->addend_ld % 4 == 0), DIGA_EINVAL, "%s: bad addend", who);
    DIGA_REQUIRE(!e->mask_y || (aligned16(e->mask_y) && e->mask_ld >= a.Cout && e->mask_ld % 4 == 0), DIGA_EINVAL, "%s: bad mask_y", who);
    DIGA_REQUIRE(!e->x || (aligned16(e->x) && e->x_ld >= a.Cout && e->x_ld % 4 == 0), DIGA_EINVAL, "%s: bad x", who);
    DIGA_REQUIRE((e->mask_y != nullptr) + (e->relu_ab != nullptr) + (e->mask_bits != nullptr) <= 1, DIGA_EINVAL,
                 "%s: give one of mask_y, mask_bits, relu_ab", who);
    DIGA_REQUIRE(!e->mask_bits || e->mask_bits_ld * 8 >= a.Cout, DIGA_EINVAL, "%s: bad mask_bits", who);
    DIGA_REQUIRE(!e->relu_ab || (e->x && aligned16(e->relu_ab)), DIGA_EINVAL, "%s: relu_ab needs x", who);
    DIGA_REQUIRE(!e->partials || (e->x && e->mean && e->invstd && aligned16(e->mean) && aligned16(e->invstd)), DIGA_EINVAL,
                 "%s: partials need x, mean and invstd", who);
    a.e_add = e->addend; a.e_add_ld = (int)e->addend_ld;
    a.e_masky = e->mask_y; a.e_masky_ld = (int)e->mask_ld;
    a.e_maskbits = e->mask_bits; a.e_maskbits_ld = (int)e->mask_bits_ld;
    a.e_x = e->x; a.e_x_ld = (int)e->x_ld;
    a.e_relu_ab = e->relu_ab; a.e_mean = e->mean; a.e_invstd = e->invstd; a.e_partials = e->partials;
    return DIGA_OK;
}

// ---------------------------------------------------------------------------------------------
// Plain fp32 GEMM out [M x Cout] = A [M x K] * W^T (W [Cout][K], one panel per `wb_tiles` 256-row tiles) as a PERSISTENT
// version of conv_fwd_dma_kernel: 256 blocks (one per CU), each walking its share of the 256 x 128 tiles in one continuous
// stream of K-steps through the three-stage LDS-DMA ring.  The loader waves run two steps ahead ACROSS tile boundaries, so
// the next tile's first stages land under the current tile's last steps (a fresh block waits 3.3-3.9 us for them, sets up for
// 0.6 us and follows its predecessor by 1.7 us: 13 % of a K = 256 tile); a finished tile leaves the accumulators by
// non-temporal stores straight from the registers (128-byte row pieces: a 32x32 accumulator tile holds 32 consecutive
// columns over the lanes) -- no LDS staging, no extra barrier, the stores drain under the next tile's MFMAs.  XCD x owns a
// contiguous range of tiles (as xcd_remap gives it); its 32 blocks take them round-robin.  Same MFMA order per accumulator
// as conv_fwd_dma_kernel: bit-identical results.  M % 256 == 0, K % 32 == 0, Cout % 128 == 0.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A;
    const float* W;
    float* out;
    int M, K, Cout, tiles_m, tiles_n, wb_tiles;      // M: valid rows (tiles_m = ceil(M / 256); rows beyond M load zeros, store nothing)
    int64_t wb_stride;
    int64_t a_ld, out_ld;                            // floats between rows of A / out
    const float* bias;                               // nullable [Cout]
    float* stats;                                    // nullable [ceil(M / 64)][3][Cout]: {sum (y - s), sum (y - s)^2, s = first row} per 64-row chunk
};

__global__ __launch_bounds__(768, 3) void gemm_f32_persistent_kernel(GemmArgs g) {
    constexpr int A_BYTES = 256 * 128, B_BYTES = 128 * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool loader = wv >= 8;
    const int total = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int q = total >> 3, r = total & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, len = q + (xcd < r ? 1 : 0);
    const int nmine = slot < len ? (len - slot + nslots - 1) / nslots : 0;
    if (nmine == 0) return;                                      // (uniform over the block)
    const int ksteps = g.K / 32;
    const int total_steps = nmine * ksteps;
    const int64_t rowb = g.a_ld * 4, wrowb = (int64_t)g.K * 4;

    if (loader) {
        const int lw = wv - 8;
        const int lrow = lane >> 3;
        const unsigned char* pa[8];
        const unsigned char* pb[4];
        int l_it = 0, l_cc = 0, issued = 0;
        auto set_tile = [&](int it) {
            const int t = start + slot + nslots * it;
            const int tile_n = t % g.tiles_n, tile_m = t / g.tiles_n;
            const int row0 = tile_m * 256 + lw * 64 + lrow;
            const unsigned char* ab = reinterpret_cast<const unsigned char*>(g.A) + (int64_t)row0 * rowb;
            const unsigned char* wb = reinterpret_cast<const unsigned char*>(g.W + (g.wb_tiles > 0 ? (int64_t)(tile_m / g.wb_tiles) * g.wb_stride : 0)) +
                                      (int64_t)(tile_n * 128 + lw * 32 + lrow) * wrowb;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                pa[j] = row0 + 8 * j < g.M ? ab + (int64_t)(8 * j) * rowb + (((lane & 7) ^ (((8 * j + lrow) >> 1) & 7)) << 4) : nullptr;
#pragma unroll
            for (int c = 0; c < 4; ++c) pb[c] = wb + (int64_t)(8 * c) * wrowb + (((lane & 7) ^ (((8 * c + lrow) >> 1) & 7)) << 4);
        };
        auto issue = [&](int buf) {
            unsigned char* stage = smem_b + buf * STAGE;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned char* src = pa[j] != nullptr ? pa[j] + l_cc * 128 : g_zero16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + (lw * 64 + 8 * j) * 128), 16, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[c] + l_cc * 128),
                                                 (__attribute__((address_space(3))) void*)(stage + A_BYTES + (lw * 32 + 8 * c) * 128), 16, 0, 0);
            ++issued;
            if (++l_cc == ksteps) {
                l_cc = 0;
                if (++l_it < nmine) set_tile(l_it);
            }
        };
        auto wait_next = [&](bool newest_in_flight) {            // 12 loads per stage and loader wave
            if (newest_in_flight) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
        set_tile(0);
        issue(0);
        if (total_steps > 1) issue(1);
        wait_next(total_steps > 1);
        int nx = 2;
        for (int gs = 0; gs < total_steps; ++gs) {
            const bool ahead = gs + 2 < total_steps;
            if (ahead) issue(nx);
            wait_next(ahead);
            nx = nx == 2 ? 0 : nx + 1;
        }
        return;
    }

    const int wm = wv >> 1, wn = wv & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    const int fsw = (li >> 1) & 7;
    // k-group t of row li sits at byte li * 128 + (((2 t + lh) ^ fsw) << 4) = F ^ (t << 5), F = li * 128 + ((lh ^ fsw) << 4): ONE
    // per-lane base per operand and an XOR immediate per k-group instead of four offset registers and eight per-step addresses
    // (round 5: 117 -> 112 VGPRs with the 32-bit output offsets below -- 176 instead of 152 registers per SIMD stay free for the
    // other stream's bandwidth kernels; same reads, same MFMA order)
    const int F = li * 128 + ((lh ^ fsw) << 4);
    const int FA = F + wm * 64 * 128, FB = F + A_BYTES + wn * 64 * 128;
#define DIGA_FOFF(t) ((t) << 5)

    __builtin_amdgcn_s_barrier();                                // stage 0 has landed
    int cur = 0, it = 0, ks_in_tile = 0;
    for (int gs = 0; gs < total_steps; ++gs) {
        const int sa = cur * STAGE + FA, sb = cur * STAGE + FB;      // (stage and wave bases are multiples of 8 KB: bits 5-6 stay F's)
#define As_AT(t) (smem_b + (sa ^ DIGA_FOFF(t)))
#define Bs_AT(t) (smem_b + (sb ^ DIGA_FOFF(t)))
        float4 fa[2][2], fb[2][2];
        fa[0][0] = *reinterpret_cast<const float4*>(As_AT(0));
        fa[0][1] = *reinterpret_cast<const float4*>(As_AT(0) + 32 * 128);
        fb[0][0] = *reinterpret_cast<const float4*>(Bs_AT(0));
        fb[0][1] = *reinterpret_cast<const float4*>(Bs_AT(0) + 32 * 128);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int p = t & 1;
            if (t + 1 < 4) {
                fa[p ^ 1][0] = *reinterpret_cast<const float4*>(As_AT(t + 1));
                fa[p ^ 1][1] = *reinterpret_cast<const float4*>(As_AT(t + 1) + 32 * 128);
                fb[p ^ 1][0] = *reinterpret_cast<const float4*>(Bs_AT(t + 1));
                fb[p ^ 1][1] = *reinterpret_cast<const float4*>(Bs_AT(t + 1) + 32 * 128);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float av = e == 0 ? fa[p][i].x : e == 1 ? fa[p][i].y : e == 2 ? fa[p][i].z : fa[p][i].w;
                        const float bv = e == 0 ? fb[p][j].x : e == 1 ? fb[p][j].y : e == 2 ? fb[p][j].z : fb[p][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
        if (++ks_in_tile == ksteps) {
            // tile finished: accumulator register e of a 32x32 tile is row (e & 3) + 8 (e >> 2) + 4 lh, column li
            const int t = start + slot + nslots * it;
            const int tile_n = t % g.tiles_n, tile_m = t / g.tiles_n;
            const int row_w = tile_m * 256 + wm * 64;                          // first row of this wave's 64 x 64 piece
            const int col_w = tile_n * 128 + wn * 64 + li;
            // 32-bit element offsets from the uniform output pointer (M * out_ld < 2^32: checked where the kernel is chosen): one
            // offset register instead of a 64-bit pointer per lane
            const unsigned ob = (unsigned)(row_w + 4 * lh) * (unsigned)g.out_ld + (unsigned)col_w;
#define DIGA_O_AT(r, j) (g.out + (size_t)(ob + (unsigned)(r) * (unsigned)g.out_ld + (unsigned)((j) * 32)))
            const int rows_left = g.M - (row_w + 4 * lh);                      // row offset r is valid iff r < rows_left
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bv = g.bias != nullptr ? g.bias[col_w + j * 32] : 0.f;
                // what the BatchNorm after the conv needs, per 64-row chunk (= this wave's rows) and column: shift s = the chunk's
                // first row (register 0 of the lower lane half), sum (y - s), sum (y - s)^2 over the valid rows
                const float sh = __shfl(acc[0][j][0] + bv, li, 64);
                float sd = 0.f, sd2 = 0.f;
                if (row_w + 64 <= g.M) {
                    // (uniform branch) all 64 rows of the wave's piece exist -- every tile but the last row tile: no per-element
                    // compare + exec-mask around each of the 64 stores
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int r = i * 32 + (e & 3) + 8 * (e >> 2);
                            const float v = acc[i][j][e] + bv;
                            __builtin_nontemporal_store(v, DIGA_O_AT(r, j));
                            const float d = v - sh;
                            sd += d;
                            sd2 += d * d;
                            acc[i][j][e] = 0.f;
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int r = i * 32 + (e & 3) + 8 * (e >> 2);
                            const float v = acc[i][j][e] + bv;
                            if (r < rows_left) {
                                __builtin_nontemporal_store(v, DIGA_O_AT(r, j));
                                const float d = v - sh;
                                sd += d;
                                sd2 += d * d;
                            }
                            acc[i][j][e] = 0.f;
                        }
                }
                if (g.stats != nullptr && row_w < g.M) {
                    sd += __shfl_xor(sd, 32, 64);
                    sd2 += __shfl_xor(sd2, 32, 64);
                    if (lh == 0) {
                        float* sp = g.stats + (int64_t)(row_w >> 6) * 3 * g.Cout + col_w + j * 32;
                        sp[0] = sd;
                        sp[g.Cout] = sd2;
                        sp[2 * g.Cout] = sh;
                    }
                }
            }
            ks_in_tile = 0;
            ++it;
        }
    }
}

// `batches` independent products out_b [rows x Cout] = A_b [rows x K] * W_b^T (W_b [Cout][K]) in one launch of
// conv_fwd_dma_kernel: the operands of all batches are stacked row-wise (A [batches * rows][K], out likewise,
// W [batches][Cout][K]); rows % 256 == 0 so that no 256-row tile straddles two batches.  Used by winograd.hip.
int gemm_batched_f32_dma(const float* A, int64_t rows_per_batch, int batches, int64_t K, const float* W, int64_t Cout,
                         float* out, hipStream_t st) {
    DIGA_REQUIRE(rows_per_batch > 0 && rows_per_batch % 256 == 0 && K % 32 == 0 && Cout > 0 && Cout % 4 == 0, DIGA_EINVAL,
                 "gemm_batched_f32_dma: rows %% 256, K %% 32, Cout %% 4 required");
    const int64_t M = rows_per_batch * batches;
    DIGA_REQUIRE(M / 256 < 32768 && M < (1ll << 31), DIGA_EINVAL, "gemm_batched_f32_dma: too many rows");
    ConvArgs a;
    a.in = A; a.wgt = W; a.wgt_hi = nullptr; a.wgt_lo = nullptr; a.wgt_img = nullptr; a.bias = nullptr; a.out = out; a.stats = nullptr;
    a.N = 1; a.Hi = (int)(M / 256); a.Wi = 256; a.Cin = (int)K; a.in_ld = (int)K;
    a.Ho = a.Hi; a.Wo = 256; a.Cout = (int)Cout; a.out_ld = (int)Cout;
    a.R = 1; a.S = 1; a.sy = 1; a.sx = 1; a.oy0 = 0; a.ox0 = 0; a.ody = 1; a.odx = 1;
    a.M = (int)M; a.tiles_m = (int)ceil_div(M, 128); a.tiles_n = (int)ceil_div(Cout, 128); a.all_inside = 1;
    a.pad_reflect = 0; a.up_shift = 0; a.act = 0;
    a.e_add = a.e_masky = a.e_x = a.e_relu_ab = a.e_mean = a.e_invstd = nullptr;
    a.e_partials = nullptr; a.e_maskbits = nullptr;
    a.e_add_ld = a.e_masky_ld = a.e_x_ld = a.e_maskbits_ld = 0;
    a.wb_tiles = (int)(rows_per_batch / 256);
    a.wb_stride = Cout * K;
    const unsigned grid = (unsigned)((M / 256) * a.tiles_n);
    const size_t sh = 3 * (256 + 128) * 128;
    if (Cout % 128 == 0 && grid >= 512 && M * Cout < (1ll << 32)) {        // (at least two rounds of tiles: the persistent walk; 32-bit output offsets)
        GemmArgs g;
        g.A = A; g.W = W; g.out = out; g.M = (int)M; g.K = (int)K; g.Cout = (int)Cout;
        g.tiles_m = (int)(M / 256); g.tiles_n = a.tiles_n; g.wb_tiles = a.wb_tiles; g.wb_stride = a.wb_stride;
        g.a_ld = K; g.out_ld = Cout; g.bias = nullptr; g.stats = nullptr;
        (void)hipFuncSetAttribute((const void*)gemm_f32_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL(gemm_f32_persistent_kernel, dim3(256), dim3(768), sh, st, g);
        return DIGA_OK;
    }
    (void)hipFuncSetAttribute((const void*)conv_fwd_dma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(conv_fwd_dma_kernel<false>, dim3(grid), dim3(768), sh, st, a);
    return DIGA_OK;
}

// A stride-1 pointwise convolution in fp32 with enough tiles for two rounds goes to gemm_f32_persistent_kernel (shape rule
// only: the statistics buffer's chunk size must be known to the caller, diga_conv2d_stats_chunk_rows).
bool pointwise_persistent_ok(int64_t M, int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t R,
                             int64_t S, int64_t sy, int64_t sx, int64_t oy0, int64_t ox0) {
    return R == 1 && S == 1 && sy == 1 && sx == 1 && oy0 == 0 && ox0 == 0 && Hi == Ho && Wi == Wo && Cin % 32 == 0 &&
           Cout % 128 == 0 && ceil_div(M, 256) * (Cout / 128) >= 512;
}

}  // namespace diga

using namespace diga;

static int conv2d_f32_impl(const float* in, const float* wgt, const float* bias, float* out, int64_t N,
                                    int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo,
                                    int64_t Cout, int64_t out_ld, int64_t R, int64_t S, int64_t stride_y,
                                    int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx,
                                    float* stats_partial, int prof_tag, void* stream, const diga_bwd_epilogue_t* epi,
                                    const diga_conv_options_t* opts = nullptr) {
    DIGA_REQUIRE(in && wgt && out, DIGA_EINVAL, "conv2d: null pointer");
    DIGA_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Cout > 0 && R > 0 && S > 0, DIGA_EINVAL, "conv2d: bad shape");
    int rc = check_conv_common("conv2d", Cin, in_ld, out_ld, Cout, in, wgt, out);
    if (rc) return rc;
    DIGA_REQUIRE(N * Hi * Wi < (1ll << 31) && N * Ho * Wo < (1ll << 31), DIGA_EINVAL, "conv2d: too many pixels for 32-bit tile indices");
    ConvArgs a;
    a.in = in; a.wgt = wgt; a.wgt_hi = nullptr; a.wgt_lo = nullptr; a.bias = bias; a.out = out; a.stats = stats_partial;
    a.N = (int)N; a.Hi = (int)Hi; a.Wi = (int)Wi; a.Cin = (int)Cin; a.in_ld = (int)in_ld;
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.Cout = (int)Cout; a.out_ld = (int)out_ld;
    a.R = (int)R; a.S = (int)S; a.sy = (int)stride_y; a.sx = (int)stride_x;
    a.oy0 = (int)off_y0; a.ox0 = (int)off_x0; a.ody = (int)off_dy; a.odx = (int)off_dx;
    a.M = (int)(N * Ho * Wo);
    a.all_inside = 0;
    a.tiles_m = (int)ceil_div(a.M, 128);
    rc = check_options(opts, "conv2d");
    if (rc) return rc;
    set_options(a, opts);
    rc = set_bwd_epilogue(a, epi, "conv2d");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(prof_tag == DIGA_PROF_CONV_BWD_DATA ? DIGA_PROF_CONV_BWD_DATA : DIGA_PROF_CONV_FWD, st,
                   2.0 * (double)a.M * (double)Cout * (double)(R * S) * (double)Cin);
#define DIGA_LAUNCH_K(KERNEL_, THREADS_, SH_)                                                                          \
    do {                                                                                                               \
        (void)hipFuncSetAttribute((const void*)KERNEL_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SH_));        \
        hipLaunchKernelGGL(KERNEL_, dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(THREADS_), (SH_), st, a);             \
    } while (0)
#define DIGA_FWD_LAUNCH(TN_, BK_)                                                                                      \
    do {                                                                                                               \
        const size_t sh = (size_t)(2 * 128 * (BK_ + 4) + 2 * 64 * TN_ * (BK_ + 4)) * sizeof(float);                     \
        if (epi != nullptr) DIGA_LAUNCH_K((conv_fwd_kernel<TN_, 32, true>), 256, sh);                                   \
        else DIGA_LAUNCH_K((conv_fwd_kernel<TN_, BK_, false>), 256, sh);                                                \
    } while (0)
    // 256 x 128 tiles, LDS-DMA operands (conv_fwd_dma_kernel): one block per CU, so nothing overlaps a tile's epilogue --
    // measured on the C2 shapes (tools/bench_conv.py, same box) it wins 5-12 % (22 % with dead taps) from K >= 256 into
    // >= 256 channels and loses 5-10 % on the 128-channel / K = 64 layers, which stay on the 128 x 128 kernel at two
    // blocks per CU.
    if (pointwise_persistent_ok(N * Ho * Wo, Hi, Wi, Cin, Ho, Wo, Cout, R, S, stride_y, stride_x, off_y0, off_x0) &&
        epi == nullptr && !(opts && (opts->reflect_pad || opts->upsample_shift || opts->activation))) {
        // 1x1, stride 1: the persistent GEMM (statistics per 64-row chunk: diga_conv2d_stats_chunk_rows)
        DIGA_REQUIRE((int64_t)a.M * out_ld < (1ll << 32), DIGA_EINVAL, "conv2d_nhwc_f32: output beyond 2^32 elements (the pointwise kernel "
                     "addresses it with 32-bit element offsets)");
        GemmArgs g;
        g.A = in; g.W = wgt; g.out = out; g.M = a.M; g.K = (int)Cin; g.Cout = (int)Cout;
        g.tiles_m = (int)ceil_div(a.M, 256); g.tiles_n = (int)(Cout / 128); g.wb_tiles = 0; g.wb_stride = 0;
        g.a_ld = in_ld; g.out_ld = out_ld; g.bias = bias; g.stats = stats_partial;
        const size_t shp = 3 * (256 + 128) * 128;
        (void)hipFuncSetAttribute((const void*)gemm_f32_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp);
        hipLaunchKernelGGL(gemm_f32_persistent_kernel, dim3(256), dim3(768), shp, st, g);
        return launch_status("diga_conv2d_nhwc_f32");
    }
    if (a.M >= 256 && Cout >= 256 && R * S * Cin >= 256) {
        // (the BatchNorm partials keep their 128-row chunks)
        a.tiles_n = (int)ceil_div(Cout, 128);
        const unsigned grid = (unsigned)(ceil_div(a.M, 256) * a.tiles_n);
        const size_t sh = 3 * (256 + 128) * 128;
        if (epi != nullptr) {
            (void)hipFuncSetAttribute((const void*)conv_fwd_dma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL(conv_fwd_dma_kernel<true>, dim3(grid), dim3(768), sh, st, a);
        } else {
            (void)hipFuncSetAttribute((const void*)conv_fwd_dma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL(conv_fwd_dma_kernel<false>, dim3(grid), dim3(768), sh, st, a);
        }
    } else if (Cout > 64) {
        a.tiles_n = (int)ceil_div(Cout, 128);
        DIGA_FWD_LAUNCH(2, 32);
    } else {
        a.tiles_n = 1;
        DIGA_FWD_LAUNCH(1, 32);
    }
#undef DIGA_FWD_LAUNCH
    return launch_status("diga_conv2d_nhwc_f32");
}


extern "C" int diga_conv2d_nhwc_f32(const float* in, const float* wgt, const float* bias, float* out, int64_t N,
                                    int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo,
                                    int64_t Cout, int64_t out_ld, int64_t R, int64_t S, int64_t stride_y,
                                    int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx,
                                    float* stats_partial, int prof_tag, void* stream) {
    return conv2d_f32_impl(in, wgt, bias, out, N, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x, off_y0, off_x0,
                           off_dy, off_dx, stats_partial, prof_tag, stream, nullptr);
}

extern "C" int diga_conv2d_nhwc_f32_epi(const float* in, const float* wgt, float* out, int64_t N, int64_t Hi, int64_t Wi,
                                        int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld,
                                        int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0,
                                        int64_t off_x0, int64_t off_dy, int64_t off_dx, const diga_bwd_epilogue_t* epi,
                                        int prof_tag, void* stream) {
    DIGA_REQUIRE(epi != nullptr, DIGA_EINVAL, "conv2d_epi: null epilogue descriptor");
    return conv2d_f32_impl(in, wgt, nullptr, out, N, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x, off_y0,
                           off_x0, off_dy, off_dx, nullptr, prof_tag, stream, epi);
}

extern "C" int diga_split_bf16(const float* x, uint16_t* hi, uint16_t* lo, int64_t n, void* stream) {
    DIGA_REQUIRE(x && hi && lo && n > 0 && n % 4 == 0, DIGA_EINVAL, "split_bf16: n must be a positive multiple of 4");
    DIGA_REQUIRE(aligned16(x) && ((uintptr_t)hi & 7u) == 0 && ((uintptr_t)lo & 7u) == 0, DIGA_EALIGN, "split_bf16: alignment");
    int64_t blocks = ceil_div(n / 4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, hi, lo, n / 4);
    return launch_status("diga_split_bf16");
}

static int conv2d_bf16x3_impl(const float* in, const uint16_t* wgt_hi, const uint16_t* wgt_lo, const float* bias,
                                       float* out, int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld,
                                       int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                                       int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy,
                                       int64_t off_dx, float* stats_partial, int prof_tag, void* stream,
                                       const diga_bwd_epilogue_t* epi, const diga_conv_options_t* opts = nullptr) {
    DIGA_REQUIRE(in && wgt_hi && wgt_lo && out, DIGA_EINVAL, "conv2d_bf16x3: null pointer");
    DIGA_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Cout > 0 && R > 0 && S > 0, DIGA_EINVAL, "conv2d_bf16x3: bad shape");
    int rc = check_conv_common("conv2d_bf16x3", Cin, in_ld, out_ld, Cout, in, in, out);
    if (rc) return rc;
    DIGA_REQUIRE(((uintptr_t)wgt_hi & 7u) == 0 && ((uintptr_t)wgt_lo & 7u) == 0, DIGA_EALIGN, "conv2d_bf16x3: weight alignment");
    DIGA_REQUIRE(N * Hi * Wi < (1ll << 31) && N * Ho * Wo < (1ll << 31), DIGA_EINVAL, "conv2d_bf16x3: too many pixels");
    ConvArgs a;
    a.in = in; a.wgt = nullptr; a.wgt_hi = wgt_hi; a.wgt_lo = wgt_lo; a.wgt_img = nullptr; a.bias = bias; a.out = out; a.stats = stats_partial;
    a.N = (int)N; a.Hi = (int)Hi; a.Wi = (int)Wi; a.Cin = (int)Cin; a.in_ld = (int)in_ld;
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.Cout = (int)Cout; a.out_ld = (int)out_ld;
    a.R = (int)R; a.S = (int)S; a.sy = (int)stride_y; a.sx = (int)stride_x;
    a.oy0 = (int)off_y0; a.ox0 = (int)off_x0; a.ody = (int)off_dy; a.odx = (int)off_dx;
    a.M = (int)(N * Ho * Wo);
    a.tiles_m = (int)ceil_div(a.M, 128);
    {
        const int64_t y_lo = off_y0 + std::min<int64_t>(0, (R - 1) * off_dy), y_hi = (Ho - 1) * stride_y + off_y0 + std::max<int64_t>(0, (R - 1) * off_dy);
        const int64_t x_lo = off_x0 + std::min<int64_t>(0, (S - 1) * off_dx), x_hi = (Wo - 1) * stride_x + off_x0 + std::max<int64_t>(0, (S - 1) * off_dx);
        a.all_inside = y_lo >= 0 && y_hi < Hi && x_lo >= 0 && x_hi < Wi;
    }
    rc = check_options(opts, "conv2d_bf16x3");
    if (rc) return rc;
    set_options(a, opts);
    if (a.pad_reflect || a.up_shift) a.all_inside = 0;
    rc = set_bwd_epilogue(a, epi, "conv2d_bf16x3");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(prof_tag == DIGA_PROF_CONV_BWD_DATA ? DIGA_PROF_CONV_BWD_DATA : DIGA_PROF_CONV_FWD, st,
                   2.0 * (double)a.M * (double)Cout * (double)(R * S) * (double)Cin);
    const int tn = Cout > 64 ? 2 : 1;
    a.tiles_n = (int)ceil_div(Cout, 64 * tn);
    // the wide kernel addresses the input and the weights with 32-bit element offsets
    const bool fits32 = N * Hi * Wi * in_ld < (1ll << 31) && Cout * R * S * Cin < (1ll << 31);
    DIGA_REQUIRE(fits32 || !(a.pad_reflect || a.up_shift), DIGA_EINVAL,
                 "conv2d_bf16x3: reflect padding / fused upsampling are implemented by the 256-row kernel only");
    if (fits32) {
        a.tiles_m = (int)ceil_div(a.M, 256);
        const size_t loop = (size_t)2 * 256 * 64 + (size_t)2 * 64 * tn * 64, stage = (size_t)128 * (64 * tn + 4) * sizeof(float);
        const size_t sh = loop > stage ? loop : stage;
#ifdef DIGA_PROBE_STAMP                                                         /* diagnostic build only (tools/diag/build_probe.sh) */
        if (tn == 2 && a.stats != nullptr) {
            (void)hipFuncSetAttribute((const void*)conv_fwd_x3w_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
            hipLaunchKernelGGL((conv_fwd_x3w_kernel<2, true>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), sh, st, a);
        } else
#endif
        if (tn == 2) {
            if (epi != nullptr) DIGA_LAUNCH_K((conv_fwd_x3w_kernel<2, false, true>), 256, sh);
            else DIGA_LAUNCH_K((conv_fwd_x3w_kernel<2, false, false>), 256, sh);
        } else {
            if (epi != nullptr) DIGA_LAUNCH_K((conv_fwd_x3w_kernel<1, false, true>), 256, sh);
            else DIGA_LAUNCH_K((conv_fwd_x3w_kernel<1, false, false>), 256, sh);
        }
    } else if (tn == 2) {
        const size_t sh = (size_t)2 * (2 * 128 * kRowB + 2 * 128 * kRowB);
        if (epi != nullptr) DIGA_LAUNCH_K((conv_fwd_x3p_kernel<2, true>), 256, sh);
        else DIGA_LAUNCH_K((conv_fwd_x3p_kernel<2, false>), 256, sh);
    } else {
        const size_t sh = (size_t)2 * (2 * 128 * kRowB + 2 * 64 * kRowB);
        if (epi != nullptr) DIGA_LAUNCH_K((conv_fwd_x3p_kernel<1, true>), 256, sh);
        else DIGA_LAUNCH_K((conv_fwd_x3p_kernel<1, false>), 256, sh);
    }
    return launch_status("diga_conv2d_nhwc_bf16x3");
}

extern "C" int diga_conv2d_nhwc_bf16x3(const float* in, const uint16_t* wgt_hi, const uint16_t* wgt_lo, const float* bias,
                                       float* out, int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld,
                                       int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                                       int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy,
                                       int64_t off_dx, float* stats_partial, int prof_tag, void* stream) {
    return conv2d_bf16x3_impl(in, wgt_hi, wgt_lo, bias, out, N, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x,
                              off_y0, off_x0, off_dy, off_dx, stats_partial, prof_tag, stream, nullptr);
}

extern "C" int diga_conv2d_nhwc_bf16x3_epi(const float* in, const uint16_t* wgt_hi, const uint16_t* wgt_lo, float* out, int64_t N,
                                           int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo,
                                           int64_t Cout, int64_t out_ld, int64_t R, int64_t S, int64_t stride_y,
                                           int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx,
                                           const diga_bwd_epilogue_t* epi, int prof_tag, void* stream) {
    DIGA_REQUIRE(epi != nullptr, DIGA_EINVAL, "conv2d_bf16x3_epi: null epilogue descriptor");
    return conv2d_bf16x3_impl(in, wgt_hi, wgt_lo, nullptr, out, N, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x,
                              off_y0, off_x0, off_dy, off_dx, nullptr, prof_tag, stream, epi);
}

// ---- "twin" path: activations and weights pre-split, staged by LDS-DMA (conv_fwd_x3t_kernel)
static int64_t image_bn(int64_t K) { return K > 64 ? 128 : 64; }

extern "C" int diga_make_twin(const float* x, int64_t ld, void* twin, int64_t M, int64_t C, void* stream) {
    DIGA_REQUIRE(x && twin && M > 0 && C > 0 && C % 8 == 0 && ld >= C && ld % 4 == 0, DIGA_EINVAL, "make_twin: C must be a multiple of 8");
    DIGA_REQUIRE(aligned16(x) && aligned16(twin), DIGA_EALIGN, "make_twin: pointers must be 16-byte aligned");
    int64_t blocks = ceil_div(M * (C / 8), 256);
    if (blocks > 16384) blocks = 16384;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, (hipStream_t)stream, (double)M * C * 8.0);
    hipLaunchKernelGGL(make_twin_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ld, (unsigned char*)twin, M,
                       (int)(C / 8));
    return launch_status("diga_make_twin");
}

extern "C" size_t diga_split_bf16_image_bytes(int64_t K, int64_t RS, int64_t C) {
    if (K <= 0 || RS <= 0 || C <= 0 || C % 32 != 0) return 0;
    const int64_t bn = image_bn(K);
    return (size_t)(ceil_div(K, bn) * RS * (C / 32) * 2 * bn * 64);
}

extern "C" int diga_split_bf16_image(const float* w, void* img, int64_t K, int64_t RS, int64_t C, void* stream) {
    DIGA_REQUIRE(w && img && K > 0 && RS > 0 && C > 0 && C % 32 == 0, DIGA_EINVAL, "split_bf16_image: C must be a multiple of 32");
    DIGA_REQUIRE(aligned16(w) && aligned16(img), DIGA_EALIGN, "split_bf16_image: pointers must be 16-byte aligned");
    const int64_t bn = image_bn(K);
    const int64_t total = ceil_div(K, bn) * RS * (C / 32) * bn * 4;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)img, (int)K,
                       (int)RS, (int)C, (int)bn, total);
    return launch_status("diga_split_bf16_image");
}

static int conv2d_twin_impl(const void* in_twin, const void* wgt_img, const float* bias, float* out, int64_t N,
                                     int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld,
                                     int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                                     int64_t off_dy, int64_t off_dx, float* stats_partial, int prof_tag, void* stream,
                                     const diga_bwd_epilogue_t* epi, const diga_conv_options_t* opts = nullptr) {
    DIGA_REQUIRE(in_twin && wgt_img && out, DIGA_EINVAL, "conv2d_twin: null pointer");
    DIGA_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Cout > 0 && R > 0 && S > 0, DIGA_EINVAL, "conv2d_twin: bad shape");
    DIGA_REQUIRE(Cin > 0 && Cin % 32 == 0 && out_ld >= Cout, DIGA_EINVAL, "conv2d_twin: Cin must be a multiple of 32");
    DIGA_REQUIRE(aligned16(in_twin) && aligned16(wgt_img) && ((uintptr_t)out & 3u) == 0, DIGA_EALIGN, "conv2d_twin: alignment");
    DIGA_REQUIRE(N * Hi * Wi < (1ll << 31) && N * Ho * Wo < (1ll << 31), DIGA_EINVAL, "conv2d_twin: too many pixels");
    ConvArgs a;
    a.in = reinterpret_cast<const float*>(in_twin); a.wgt = nullptr; a.wgt_hi = nullptr; a.wgt_lo = nullptr;
    a.wgt_img = reinterpret_cast<const unsigned char*>(wgt_img); a.bias = bias; a.out = out; a.stats = stats_partial;
    a.N = (int)N; a.Hi = (int)Hi; a.Wi = (int)Wi; a.Cin = (int)Cin; a.in_ld = (int)Cin;
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.Cout = (int)Cout; a.out_ld = (int)out_ld;
    a.R = (int)R; a.S = (int)S; a.sy = (int)stride_y; a.sx = (int)stride_x;
    a.oy0 = (int)off_y0; a.ox0 = (int)off_x0; a.ody = (int)off_dy; a.odx = (int)off_dx;
    a.M = (int)(N * Ho * Wo);
    a.tiles_m = (int)ceil_div(a.M, 256);
    a.all_inside = 0;
    const int tn = Cout > 64 ? 2 : 1;
    a.tiles_n = (int)ceil_div(Cout, 64 * tn);
    {
        int rc = check_options(opts, "conv2d_twin");
        if (rc) return rc;
        set_options(a, opts);
        rc = set_bwd_epilogue(a, epi, "conv2d_twin");
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(prof_tag == DIGA_PROF_CONV_BWD_DATA ? DIGA_PROF_CONV_BWD_DATA : DIGA_PROF_CONV_FWD, st,
                   2.0 * (double)a.M * (double)Cout * (double)(R * S) * (double)Cin);
    const size_t sh = (size_t)3 * (2 * 256 * 64 + 2 * 64 * tn * 64);
    // two MFMA waves per SIMD + dead-tap skipping (conv_fwd_x3t8_kernel).  Its predecessor with one MFMA wave per SIMD measured
    // 3-6 % slower on multi-tap layers (ASPP dilation 24: 17 %) and 3-12 % on pointwise layers (tools/experiments/).
    DIGA_REQUIRE(R * S <= 64, DIGA_EINVAL, "conv2d_twin: at most 64 taps (the dead-tap mask is one 64-bit word)");
    {
        // K order: tap-major (the weight layout's order, bit-identical to the register-staged kernel).  Walking channel
        // chunks outer / taps inner instead (to line up in time the re-reads of an input row that tiles running together
        // make through different vertical taps) measured 8-14 % SLOWER on every shape, ASPP included: the loader's
        // per-step tap switch costs more than the locality returns (commit 78ff4b5 has the switch).
        const size_t stg = (size_t)2 * 128 * (64 * tn + 4) * sizeof(float);
        const size_t sh8 = sh > stg ? sh : stg;
        if (tn == 2) {
            if (epi != nullptr) DIGA_LAUNCH_K((conv_fwd_x3t8_kernel<2, true>), 768, sh8);
            else DIGA_LAUNCH_K((conv_fwd_x3t8_kernel<2, false>), 768, sh8);
        } else {
            if (epi != nullptr) DIGA_LAUNCH_K((conv_fwd_x3t8_kernel<1, true>), 768, sh8);
            else DIGA_LAUNCH_K((conv_fwd_x3t8_kernel<1, false>), 768, sh8);
        }
    }
    return launch_status("diga_conv2d_nhwc_twin");
}

extern "C" int diga_conv2d_nhwc_twin(const void* in_twin, const void* wgt_img, const float* bias, float* out, int64_t N,
                                     int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld,
                                     int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                                     int64_t off_dy, int64_t off_dx, float* stats_partial, int prof_tag, void* stream) {
    return conv2d_twin_impl(in_twin, wgt_img, bias, out, N, Hi, Wi, Cin, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x, off_y0, off_x0,
                            off_dy, off_dx, stats_partial, prof_tag, stream, nullptr);
}

extern "C" int diga_conv2d_nhwc_twin_epi(const void* in_twin, const void* wgt_img, float* out, int64_t N, int64_t Hi, int64_t Wi,
                                         int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R, int64_t S,
                                         int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy,
                                         int64_t off_dx, const diga_bwd_epilogue_t* epi, int prof_tag, void* stream) {
    DIGA_REQUIRE(epi != nullptr, DIGA_EINVAL, "conv2d_twin_epi: null epilogue descriptor");
    return conv2d_twin_impl(in_twin, wgt_img, nullptr, out, N, Hi, Wi, Cin, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x, off_y0,
                            off_x0, off_dy, off_dx, nullptr, prof_tag, stream, epi);
}

// The same three forward entry points with the translator's input map / output activation folded in (diga_conv_options_t):
// explicit per call -- no state survives a call.
extern "C" int diga_conv2d_nhwc_f32_opts(const float* in, const float* wgt, const float* bias, float* out, int64_t N, int64_t Hi,
                                         int64_t Wi, int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld,
                                         int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                                         int64_t off_dy, int64_t off_dx, const diga_conv_options_t* opts, int prof_tag, void* stream) {
    DIGA_REQUIRE(opts != nullptr, DIGA_EINVAL, "conv2d_opts: null options");
    return conv2d_f32_impl(in, wgt, bias, out, N, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x, off_y0, off_x0,
                           off_dy, off_dx, nullptr, prof_tag, stream, nullptr, opts);
}

extern "C" int diga_conv2d_nhwc_bf16x3_opts(const float* in, const uint16_t* wgt_hi, const uint16_t* wgt_lo, const float* bias, float* out,
                                            int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t in_ld, int64_t Ho, int64_t Wo,
                                            int64_t Cout, int64_t out_ld, int64_t R, int64_t S, int64_t stride_y, int64_t stride_x,
                                            int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx,
                                            const diga_conv_options_t* opts, int prof_tag, void* stream) {
    DIGA_REQUIRE(opts != nullptr, DIGA_EINVAL, "conv2d_bf16x3_opts: null options");
    return conv2d_bf16x3_impl(in, wgt_hi, wgt_lo, bias, out, N, Hi, Wi, Cin, in_ld, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x, off_y0,
                              off_x0, off_dy, off_dx, nullptr, prof_tag, stream, nullptr, opts);
}

extern "C" int diga_conv2d_nhwc_twin_opts(const void* in_twin, const void* wgt_img, const float* bias, float* out, int64_t N, int64_t Hi,
                                          int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout, int64_t out_ld, int64_t R,
                                          int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0, int64_t off_dy,
                                          int64_t off_dx, const diga_conv_options_t* opts, int prof_tag, void* stream) {
    DIGA_REQUIRE(opts != nullptr, DIGA_EINVAL, "conv2d_twin_opts: null options");
    return conv2d_twin_impl(in_twin, wgt_img, bias, out, N, Hi, Wi, Cin, Ho, Wo, Cout, out_ld, R, S, stride_y, stride_x, off_y0, off_x0,
                            off_dy, off_dx, nullptr, prof_tag, stream, nullptr, opts);
}

extern "C" size_t diga_conv2d_stats_floats(int64_t N, int64_t Ho, int64_t Wo, int64_t Cout) {
    return (size_t)ceil_div(N * Ho * Wo, 64) * 3 * (size_t)Cout;       // (room for 64-row chunks)
}

extern "C" int diga_conv2d_epi_chunk_rows(int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout,
                                          int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                                          int math) {
    // (a backward-data epilogue runs on the per-tile kernels: 128-row partial groups for every shape and arithmetic)
    (void)N; (void)Hi; (void)Wi; (void)Cin; (void)Ho; (void)Wo; (void)Cout; (void)R; (void)S; (void)stride_y; (void)stride_x;
    (void)off_y0; (void)off_x0; (void)math;
    return 128;
}

extern "C" int diga_conv2d_stats_chunk_rows(int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho, int64_t Wo, int64_t Cout,
                                            int64_t R, int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0, int64_t off_x0,
                                            int math) {
    return (math == DIGA_CONV_MATH_F32 && pointwise_persistent_ok(N * Ho * Wo, Hi, Wi, Cin, Ho, Wo, Cout, R, S, stride_y, stride_x, off_y0, off_x0))
               ? 64 : 128;
}


namespace {
struct WgradPlan {
    int tm, tn, tiles_m, tiles_n, splits, steps_per_split;
};
WgradPlan plan_wgrad(int64_t M, int64_t Cout, int64_t Cin, int64_t RS, bool x3) {
    WgradPlan p;
    p.tm = Cout > 64 ? 2 : 1;
    if (x3 && Cout >= 256) p.tm = 4;                   // conv_wgrad_x3w_kernel: 256 output channels per block
    p.tn = Cin > 64 ? 2 : 1;
    p.tiles_m = (int)ceil_div(Cout, 64 * p.tm);
    p.tiles_n = (int)ceil_div(Cin, 64 * p.tn);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * RS;
    const int64_t ksteps = ceil_div(M, kBK);
    int64_t splits = ceil_div(1024, tiles);            // ~4 blocks per CU in total
    if (p.tm == 4) splits = 512 / tiles > 0 ? 512 / tiles : 1;   // wide kernel: one full round of 2 blocks per CU
    const int64_t max_splits = ksteps / 8 > 0 ? ksteps / 8 : 1;   // at least 8 K-steps per block
    if (splits > max_splits) splits = max_splits;
    if (splits > 512) splits = 512;
    if (splits < 1) splits = 1;
    p.steps_per_split = (int)ceil_div(ksteps, splits);
    p.splits = (int)ceil_div(ksteps, p.steps_per_split);
    return p;
}
// 256 x 128 tiles at one block per CU (conv_wgrad_dma_kernel, conv_wgrad_x3t_kernel): the split count whose grid fills
// whole rounds of 256 blocks best (>= 16 K-steps per block)
WgradPlan plan_wgrad_wide(int64_t M, int64_t Cout, int64_t Cin, int64_t RS) {
    WgradPlan p;
    p.tm = 4;
    p.tn = 2;
    p.tiles_m = (int)ceil_div(Cout, 256);
    p.tiles_n = (int)ceil_div(Cin, 128);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * RS, ksteps = ceil_div(M, kBK);
    int64_t best = 1;
    double best_eff = 0.0;
    for (int64_t rounds = 1; rounds <= 4; ++rounds) {
        int64_t sp = 256 * rounds / tiles;
        if (sp < 1) sp = 1;
        if (ksteps / sp < 16) continue;
        const int64_t blocks = tiles * sp;
        const double eff = (double)blocks / (256.0 * (double)ceil_div(blocks, 256));
        if (eff > best_eff + 0.02) {
            best_eff = eff;
            best = sp;
        }
    }
    p.steps_per_split = (int)ceil_div(ksteps, best);
    p.splits = (int)ceil_div(ksteps, p.steps_per_split);
    return p;
}
bool wgrad_dma_ok(int64_t M, int64_t Cout, int64_t Cin) {
    return Cout % 256 == 0 && Cin % 128 == 0 && M >= 1024;
}
int64_t wgrad_mpad(int64_t M) { return ceil_div(M, kBK) * kBK + 2 * kBK; }
size_t wgrad_slab_bytes(const WgradPlan& p, int64_t Cout, int64_t Cin, int64_t RS) {
    return p.splits > 1 ? (size_t)p.splits * Cout * RS * Cin * sizeof(float) : 0;
}
}  // namespace

namespace diga {
// `batches` independent products dU_b [Cout x Cin] = Z_b^T [Cout x rows] * V_b [rows x Cin] (contraction over the rows) in
// one launch of conv_wgrad_dma_kernel, the batch riding on the kernel's tap index: Z [batches][rows][Cout],
// V [batches][rows][Cin], dU [Cout][batches][Cin].  rows % 32 == 0, Cout % 256 == 0, Cin % 128 == 0.  `slab`: scratch of
// wgrad_batched_slab_bytes() bytes for the split-K partial sums (fixed-order reduce).  Used by winograd.hip.
size_t wgrad_batched_slab_bytes(int64_t rows, int batches, int64_t Cout, int64_t Cin) {
    return wgrad_slab_bytes(plan_wgrad_wide(rows, Cout, Cin, batches), Cout, Cin, batches);
}
int wgrad_batched_f32_dma(const float* Z, const float* V, float* dU, float* slab, int64_t rows, int batches, int64_t Cout,
                          int64_t Cin, hipStream_t st) {
    DIGA_REQUIRE(rows > 0 && rows % 32 == 0 && Cout % 256 == 0 && Cin % 128 == 0 && batches > 0 && rows < (1ll << 31), DIGA_EINVAL,
                 "wgrad_batched_f32_dma: rows %% 32, Cout %% 256, Cin %% 128 required");
    const WgradPlan p = plan_wgrad_wide(rows, Cout, Cin, batches);
    WgradArgs a;
    a.dy = Z; a.x = V; a.slab = p.splits > 1 ? slab : dU;
    a.N = 1; a.Hi = 1; a.Wi = (int)rows; a.Cin = (int)Cin; a.x_ld = (int)Cin;
    a.Ho = 1; a.Wo = (int)rows; a.Cout = (int)Cout; a.dy_ld = (int)Cout;
    a.R = 1; a.S = batches; a.sy = 1; a.sx = 1; a.oy0 = 0; a.ox0 = 0; a.ody = 1; a.odx = 1;
    a.M = (int)rows; a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.splits = p.splits; a.steps_per_split = p.steps_per_split;
    a.ptab = nullptr; a.zeros = nullptr; a.M_pad = (int)rows;
    a.dy_tap_stride = rows * Cout;
    a.x_tap_stride = rows * Cin;
    const unsigned grid = (unsigned)((int64_t)p.tiles_m * p.tiles_n * batches * p.splits);
    const size_t shd = (size_t)3 * kBK * (256 + 128) * 4;
    (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shd);
    hipLaunchKernelGGL(conv_wgrad_dma_kernel, dim3(grid), dim3(768), shd, st, a);
    if (p.splits > 1) {
        const int64_t n4 = Cout * batches * Cin / 4;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, st, (const float*)slab, dU, n4, p.splits);
    }
    return DIGA_OK;
}

}  // namespace diga

// workspace = [split-K slabs][(tap, pixel) -> input pixel table of the wide split-bf16 kernel][16 zero bytes]; sized
// for either conv arithmetic so that a query and the call may straddle a diga_set_conv_math()
extern "C" size_t diga_conv2d_wgrad_workspace_bytes(int64_t N, int64_t Ho, int64_t Wo, int64_t Cout, int64_t Cin,
                                                    int64_t R, int64_t S) {
    const int64_t M = N * Ho * Wo, RS = R * S;
    const size_t s0 = wgrad_slab_bytes(plan_wgrad(M, Cout, Cin, RS, false), Cout, Cin, RS);
    size_t s1 = wgrad_slab_bytes(plan_wgrad(M, Cout, Cin, RS, true), Cout, Cin, RS);
    if (wgrad_dma_ok(M, Cout, Cin)) s1 = std::max(s1, wgrad_slab_bytes(plan_wgrad_wide(M, Cout, Cin, RS), Cout, Cin, RS));
    return (s0 > s1 ? s0 : s1) + (size_t)RS * wgrad_mpad(M) * sizeof(int) + 64;
}

extern "C" int diga_conv2d_wgrad_nhwc_f32(const float* dy, const float* x, float* dw, void* workspace,
                                          size_t workspace_bytes, int64_t N, int64_t Hi, int64_t Wi, int64_t Cin,
                                          int64_t x_ld, int64_t Ho, int64_t Wo, int64_t Cout, int64_t dy_ld, int64_t R,
                                          int64_t S, int64_t stride_y, int64_t stride_x, int64_t off_y0,
                                          int64_t off_x0, int64_t off_dy, int64_t off_dx, int math, void* stream) {
    DIGA_REQUIRE(dy && x && dw, DIGA_EINVAL, "conv2d_wgrad: null pointer");
    DIGA_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && R > 0 && S > 0, DIGA_EINVAL, "conv2d_wgrad: bad shape");
    DIGA_REQUIRE(Cin > 0 && Cin % 4 == 0 && Cout > 0 && Cout % 4 == 0 && x_ld >= Cin && x_ld % 4 == 0 && dy_ld >= Cout &&
                     dy_ld % 4 == 0,
                 DIGA_EINVAL, "conv2d_wgrad: channel counts and leading dimensions must be multiples of 4");
    DIGA_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dw), DIGA_EALIGN, "conv2d_wgrad: pointers must be 16-byte aligned");
    DIGA_REQUIRE(N * Hi * Wi < (1ll << 31) && N * Ho * Wo < (1ll << 31), DIGA_EINVAL, "conv2d_wgrad: too many pixels");
    const int64_t RS = R * S, M = N * Ho * Wo;
    DIGA_REQUIRE(math == DIGA_CONV_MATH_F32 || math == DIGA_CONV_MATH_BF16X3, DIGA_EINVAL, "conv2d_wgrad: math must be DIGA_CONV_MATH_F32 or _BF16X3");
    const bool x3 = math == DIGA_CONV_MATH_BF16X3;
    const bool dma = !x3 && wgrad_dma_ok(M, Cout, Cin);
    const WgradPlan p = dma ? plan_wgrad_wide(M, Cout, Cin, RS) : plan_wgrad(M, Cout, Cin, RS, x3);
    const bool wide = p.tm == 4 && !dma;
    const size_t slab_bytes = wgrad_slab_bytes(p, Cout, Cin, RS);
    const int64_t M_pad = wgrad_mpad(M);
    const bool identity = RS == 1 && stride_y == 1 && stride_x == 1 && off_y0 == 0 && off_x0 == 0 && Hi == Ho && Wi == Wo;
    const bool f32_tab = !x3 && !identity;               // conv_wgrad_kernel reads the pixel table too (round 3)
    const size_t need = slab_bytes + ((wide || f32_tab) ? (size_t)RS * M_pad * sizeof(int) + 64 : 0);
    DIGA_REQUIRE(workspace_bytes >= need && (need == 0 || (workspace && aligned16(workspace))), DIGA_EWORKSPACE,
                 "conv2d_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    WgradArgs a;
    a.dy = dy; a.x = x; a.slab = p.splits > 1 ? (float*)workspace : dw;
    a.N = (int)N; a.Hi = (int)Hi; a.Wi = (int)Wi; a.Cin = (int)Cin; a.x_ld = (int)x_ld;
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.Cout = (int)Cout; a.dy_ld = (int)dy_ld;
    a.R = (int)R; a.S = (int)S; a.sy = (int)stride_y; a.sx = (int)stride_x;
    a.oy0 = (int)off_y0; a.ox0 = (int)off_x0; a.ody = (int)off_dy; a.odx = (int)off_dx;
    a.M = (int)M; a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.splits = p.splits; a.steps_per_split = p.steps_per_split;
    a.ptab = nullptr; a.zeros = nullptr; a.M_pad = (int)M_pad;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_CONV_BWD_WEIGHT, st, 2.0 * (double)M * (double)Cout * (double)RS * (double)Cin);
    const unsigned grid = (unsigned)((int64_t)p.tiles_m * p.tiles_n * RS * p.splits);
    if (wide) {
        int* tab = reinterpret_cast<int*>(static_cast<char*>(workspace) + slab_bytes);
        float* zeros = reinterpret_cast<float*>(tab + RS * M_pad);
        hipLaunchKernelGGL(wgrad_pixtab_kernel, dim3((unsigned)ceil_div(M_pad, 256), (unsigned)RS), dim3(256), 0, st, tab, zeros,
                           (int)M, (int)M_pad, (int)Ho, (int)Wo, (int)Hi, (int)Wi, (int)S, (int)stride_y, (int)stride_x,
                           (int)off_y0, (int)off_x0, (int)off_dy, (int)off_dx);
        a.ptab = tab;
        a.zeros = zeros;
        const size_t shw = (size_t)2 * (256 + 64 * p.tn) * kRowB;
        if (p.tn == 2) {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_x3w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shw);
            hipLaunchKernelGGL((conv_wgrad_x3w_kernel<2>), dim3(grid), dim3(256), shw, st, a);
        } else {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_x3w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shw);
            hipLaunchKernelGGL((conv_wgrad_x3w_kernel<1>), dim3(grid), dim3(256), shw, st, a);
        }
    } else
    {
    if (f32_tab) {
        int* tab = reinterpret_cast<int*>(static_cast<char*>(workspace) + slab_bytes);
        float* zeros = reinterpret_cast<float*>(tab + RS * M_pad);
        hipLaunchKernelGGL(wgrad_pixtab_kernel, dim3((unsigned)ceil_div(M_pad, 256), (unsigned)RS), dim3(256), 0, st, tab, zeros,
                           (int)M, (int)M_pad, (int)Ho, (int)Wo, (int)Hi, (int)Wi, (int)S, (int)stride_y, (int)stride_x,
                           (int)off_y0, (int)off_x0, (int)off_dy, (int)off_dx);
        a.ptab = tab;
    }
    if (dma) {
        const size_t shd = (size_t)3 * kBK * (256 + 128) * 4;
        (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shd);
        hipLaunchKernelGGL(conv_wgrad_dma_kernel, dim3(grid), dim3(768), shd, st, a);
    } else {
    const size_t sh = (size_t)(4 * kBK * kLDW) * sizeof(float);
#define DIGA_WGRAD_LAUNCH(TM_, TN_)                                                                                   \
    do {                                                                                                               \
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<TM_, TN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL((conv_wgrad_kernel<TM_, TN_>), dim3(grid), dim3(256), sh, st, a);                            \
    } while (0)
#define DIGA_WGRAD_X3_LAUNCH(TM_, TN_)                                                                                \
    do {                                                                                                               \
        const size_t shx = (size_t)2 * (2 * 64 * TM_ * kRowB + 2 * 64 * TN_ * kRowB);                                   \
        (void)hipFuncSetAttribute((const void*)conv_wgrad_x3_kernel<TM_, TN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shx); \
        hipLaunchKernelGGL((conv_wgrad_x3_kernel<TM_, TN_>), dim3(grid), dim3(256), shx, st, a);                        \
    } while (0)
    if (x3) {
        if (p.tm == 2 && p.tn == 2) DIGA_WGRAD_X3_LAUNCH(2, 2);
        else if (p.tm == 2) DIGA_WGRAD_X3_LAUNCH(2, 1);
        else if (p.tn == 2) DIGA_WGRAD_X3_LAUNCH(1, 2);
        else DIGA_WGRAD_X3_LAUNCH(1, 1);
    } else {
        if (p.tm == 2 && p.tn == 2) DIGA_WGRAD_LAUNCH(2, 2);
        else if (p.tm == 2) DIGA_WGRAD_LAUNCH(2, 1);
        else if (p.tn == 2) DIGA_WGRAD_LAUNCH(1, 2);
        else DIGA_WGRAD_LAUNCH(1, 1);
    }
#undef DIGA_WGRAD_X3_LAUNCH
#undef DIGA_WGRAD_LAUNCH
    }
    }
    if (p.splits > 1) {
        const int64_t n4 = Cout * RS * Cin / 4;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, st, (const float*)workspace, dw,
                           n4, p.splits);
    }
    return launch_status("diga_conv2d_wgrad_nhwc_f32");
}

// ---- backward-weight on the split twins (conv_wgrad_x3t_kernel)
namespace {
WgradPlan plan_wgrad_twin(int64_t M, int64_t Cout, int64_t Cin, int64_t RS) { return plan_wgrad_wide(M, Cout, Cin, RS); }
}  // namespace

extern "C" size_t diga_conv2d_wgrad_twin_workspace_bytes(int64_t N, int64_t Ho, int64_t Wo, int64_t Cout, int64_t Cin,
                                                         int64_t R, int64_t S) {
    const int64_t M = N * Ho * Wo, RS = R * S;
    return wgrad_slab_bytes(plan_wgrad_twin(M, Cout, Cin, RS), Cout, Cin, RS) + (size_t)RS * wgrad_mpad(M) * sizeof(int) + 64;
}

extern "C" int diga_conv2d_wgrad_twin(const void* dy_twin, const void* x_twin, float* dw, void* workspace,
                                      size_t workspace_bytes, int64_t N, int64_t Hi, int64_t Wi, int64_t Cin, int64_t Ho,
                                      int64_t Wo, int64_t Cout, int64_t R, int64_t S, int64_t stride_y, int64_t stride_x,
                                      int64_t off_y0, int64_t off_x0, int64_t off_dy, int64_t off_dx, void* stream) {
    DIGA_REQUIRE(dy_twin && x_twin && dw && workspace, DIGA_EINVAL, "conv2d_wgrad_twin: null pointer");
    DIGA_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && R > 0 && S > 0, DIGA_EINVAL, "conv2d_wgrad_twin: bad shape");
    DIGA_REQUIRE(Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 8 == 0, DIGA_EINVAL, "conv2d_wgrad_twin: channel counts must be multiples of 8");
    DIGA_REQUIRE(aligned16(dy_twin) && aligned16(x_twin) && aligned16(dw) && aligned16(workspace), DIGA_EALIGN, "conv2d_wgrad_twin: alignment");
    DIGA_REQUIRE(N * Hi * Wi < (1ll << 31) && N * Ho * Wo < (1ll << 31), DIGA_EINVAL, "conv2d_wgrad_twin: too many pixels");
    const int64_t RS = R * S, M = N * Ho * Wo, M_pad = wgrad_mpad(M);
    const WgradPlan p = plan_wgrad_twin(M, Cout, Cin, RS);
    const size_t slab_bytes = wgrad_slab_bytes(p, Cout, Cin, RS);
    DIGA_REQUIRE(workspace_bytes >= slab_bytes + (size_t)RS * M_pad * sizeof(int) + 64, DIGA_EWORKSPACE, "conv2d_wgrad_twin: workspace too small");
    WgradArgs a;
    a.dy = reinterpret_cast<const float*>(dy_twin); a.x = reinterpret_cast<const float*>(x_twin);
    a.slab = p.splits > 1 ? (float*)workspace : dw;
    a.N = (int)N; a.Hi = (int)Hi; a.Wi = (int)Wi; a.Cin = (int)Cin; a.x_ld = (int)Cin;
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.Cout = (int)Cout; a.dy_ld = (int)Cout;
    a.R = (int)R; a.S = (int)S; a.sy = (int)stride_y; a.sx = (int)stride_x;
    a.oy0 = (int)off_y0; a.ox0 = (int)off_x0; a.ody = (int)off_dy; a.odx = (int)off_dx;
    a.M = (int)M; a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n; a.splits = p.splits; a.steps_per_split = p.steps_per_split;
    a.M_pad = (int)M_pad;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_CONV_BWD_WEIGHT, st, 2.0 * (double)M * (double)Cout * (double)RS * (double)Cin);
    int* tab = reinterpret_cast<int*>(static_cast<char*>(workspace) + slab_bytes);
    float* zeros = reinterpret_cast<float*>(tab + RS * M_pad);
    hipLaunchKernelGGL(wgrad_pixtab_kernel, dim3((unsigned)ceil_div(M_pad, 256), (unsigned)RS), dim3(256), 0, st, tab, zeros, (int)M,
                       (int)M_pad, (int)Ho, (int)Wo, (int)Hi, (int)Wi, (int)S, (int)stride_y, (int)stride_x, (int)off_y0, (int)off_x0,
                       (int)off_dy, (int)off_dx);
    a.ptab = tab;
    a.zeros = zeros;
    const unsigned grid = (unsigned)((int64_t)p.tiles_m * p.tiles_n * RS * p.splits);
    const size_t sh = (size_t)3 * (2 * kBK * 512 + 2 * kBK * 256);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_x3t_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(conv_wgrad_x3t_kernel, dim3(grid), dim3(512), sh, st, a);
    if (p.splits > 1) {
        const int64_t n4 = Cout * RS * Cin / 4;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, st, (const float*)workspace, dw, n4,
                           p.splits);
    }
    return launch_status("diga_conv2d_wgrad_twin");
}

extern "C" int diga_weight_transpose(const float* w, float* wt, int64_t K, int64_t RS, int64_t C, void* stream) {
    DIGA_REQUIRE(w && wt && K > 0 && RS > 0 && C > 0 && RS < 65536, DIGA_EINVAL, "weight_transpose: bad argument");
    dim3 grid((unsigned)ceil_div(C, 32), (unsigned)ceil_div(K, 32), (unsigned)RS);
    hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, (int)K, (int)RS, (int)C);
    return launch_status("diga_weight_transpose");
}

// ---------------------------------------------------------------------------------------------
// Stem (7x7/2 on the 3-channel image): gather the R*S*C inputs of every output pixel into one row of
// Kpad floats (k = (r*S + s)*C + c, zero padded), so that the conv becomes a 1x1 conv with Cin = Kpad
// on the kernels above (forward and backward-weight; the image needs no gradient).  Zero-padding the 3
// channels to the 32-channel K-step instead costs 10x the FLOPs.
// ---------------------------------------------------------------------------------------------
namespace diga {
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                          int C, int H, int W, int R, int S, int stride, int pad, int Ho,
                                                          int Wo, int Kpad) {
    const int kq = Kpad >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * kq;
    const int64_t gstride = (int64_t)gridDim.x * 256;
    const int K = R * S * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const int k0 = (int)(i % kq) * 4;
        int64_t m = i / kq;
        const int wo = (int)(m % Wo);
        m /= Wo;
        const int ho = (int)(m % Ho);
        const int n = (int)(m / Ho);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            v[e] = 0.f;
            if (k < K) {
                const int c = k % C, tap = k / C;
                const int r = tap / S, s = tap - r * S;
                const int hi = ho * stride - pad + r, wi = wo * stride - pad + s;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    v[e] = x[(((int64_t)n * C + c) * H + hi) * W + wi];
            }
        }
        *reinterpret_cast<float4*>(out + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
}  // namespace diga

extern "C" int diga_im2col_nchw(const float* x, float* out, int64_t N, int64_t C, int64_t H, int64_t W, int64_t R,
                                int64_t S, int64_t stride, int64_t pad, int64_t Ho, int64_t Wo, int64_t Kpad,
                                void* stream) {
    DIGA_REQUIRE(x && out && N > 0 && C > 0 && H > 0 && W > 0 && R > 0 && S > 0 && stride > 0 && Ho > 0 && Wo > 0,
                 DIGA_EINVAL, "im2col: bad argument");
    DIGA_REQUIRE(Kpad % 4 == 0 && Kpad >= R * S * C && aligned16(out), DIGA_EINVAL, "im2col: Kpad must be a multiple of 4 >= R*S*C");
    const int64_t total = N * Ho * Wo * (Kpad / 4);
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 16384) blocks = 16384;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)N * (C * H * W + Ho * Wo * Kpad) * 4.0);
    hipLaunchKernelGGL(im2col_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, out, (int)N, (int)C, (int)H, (int)W,
                       (int)R, (int)S, (int)stride, (int)pad, (int)Ho, (int)Wo, (int)Kpad);
    return launch_status("diga_im2col_nchw");
}

#if defined(DIGA_PROBE_STAMP)
extern "C" int diga_probe_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(diga::g_probe_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif
