// HBM-bound kernels of the MiT / SegFormer student (BASELINE configs[4]): LayerNorm forward / backward
// (G5/model/networks/MixTransfomer.py:152,160,176-177,202,220 -- nn.LayerNorm(eps=1e-6)), the Mix-FFN's depthwise 3x3
// convolution + bias + GELU (:54-55,72-74,409-423) forward / backward, and the row gathers (im2col / col2im) that turn the
// overlapping patch embeddings (:187-228) and the spatial-reduction convolution (:104-106,126-128) into GEMMs.
// Token matrices are [rows = B*H*W][channels], channels contiguous (= NHWC); fp32 residual stream, fp16 branch tensors.
#include "mit_common.h"

namespace diga {
namespace mit {

// ---------------------------------------------------------------------------------------------
// LayerNorm.  16 lanes per row, lane l of a row group handles float4 number l + 16 t (t < NV): a wave-instruction
// reads 4 rows x 256 contiguous bytes.  C <= 256 * NV, C % 4 == 0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, _Float16* __restrict__ y16,
                                                            float* __restrict__ y32, int64_t ldy, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int M, int C, float eps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l16 = lane & 15, rsub = lane >> 4;
    const int nq = C / 4;
    float4 g[NV], b[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int q = l16 + 16 * t;
        g[t] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        b[t] = q < nq ? reinterpret_cast<const float4*>(beta)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_c = 1.f / (float)C;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * 4; r0 < M; r0 += (int64_t)gridDim.x * 16) {
        const int64_t r = r0 + rsub;
        const bool ok = r < M;
        const int64_t rr = ok ? r : (int64_t)M - 1;
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int q = l16 + 16 * t;
            v[t] = q < nq ? reinterpret_cast<const float4*>(x + rr * ldx)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
        }
        const float mean = sum16(s) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int q = l16 + 16 * t;
            if (q < nq) {
                const float a0 = v[t].x - mean, a1 = v[t].y - mean, a2 = v[t].z - mean, a3 = v[t].w - mean;
                ss += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = rsqrtf(sum16(ss) * inv_c + eps);
        if (ok) {
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int q = l16 + 16 * t;
                if (q >= nq) continue;
                const float o0 = (v[t].x - mean) * rstd * g[t].x + b[t].x, o1 = (v[t].y - mean) * rstd * g[t].y + b[t].y;
                const float o2 = (v[t].z - mean) * rstd * g[t].z + b[t].z, o3 = (v[t].w - mean) * rstd * g[t].w + b[t].w;
                if (y16 != nullptr)
                    *reinterpret_cast<f16x4*>(y16 + r * ldy + 4 * q) = (f16x4){(_Float16)o0, (_Float16)o1, (_Float16)o2, (_Float16)o3};
                if (y32 != nullptr) *reinterpret_cast<float4*>(y32 + r * ldy + 4 * q) = make_float4(o0, o1, o2, o3);
            }
            if (l16 == 0 && mean_out != nullptr) {
                mean_out[r] = mean;
                rstd_out[r] = rstd;
            }
        }
    }
}

// Backward: g = gscale * dy (fp16 or fp32), xhat = (x - mean) * rstd,
//   dx = dres + rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat))      -> dx32 (fp32) and/or dx16 (fp16)
//   partial[block][0][c] = sum_rows g * xhat, partial[block][1][c] = sum_rows g      (dgamma / dbeta, reduced afterwards)
template <int NV, typename GT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const GT* __restrict__ dy, int64_t ldg, const float* __restrict__ x,
                                                            int64_t ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const float* __restrict__ dres, int64_t ldr, float* __restrict__ dx32,
                                                            _Float16* __restrict__ dx16, int64_t ldo, float* __restrict__ partial,
                                                            int M, int C, int rows_per_block, float gscale) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l16 = lane & 15, rsub = lane >> 4;
    const int nq = C / 4;
    float4 gm[NV];
    float4 sg[NV], sb[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int q = l16 + 16 * t;
        gm[t] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        sg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        sb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_c = 1.f / (float)C;
    const int64_t blk0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t blk1 = blk0 + rows_per_block < M ? blk0 + rows_per_block : (int64_t)M;
    for (int64_t r0 = blk0 + wv * 4; r0 < blk1; r0 += 16) {
        const int64_t r = r0 + rsub;
        const bool ok = r < blk1;
        const int64_t rr = ok ? r : blk1 - 1;
        const float mean = mean_in[rr], rstd = rstd_in[rr];
        float4 gv[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int q = l16 + 16 * t;
            if (q < nq) {
                float4 d;
                if constexpr (sizeof(GT) == 2) {
                    const f16x4 h = *reinterpret_cast<const f16x4*>(dy + rr * ldg + 4 * q);
                    d = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                } else {
                    d = *reinterpret_cast<const float4*>(dy + rr * ldg + 4 * q);
                }
                d.x *= gscale; d.y *= gscale; d.z *= gscale; d.w *= gscale;
                const float4 xv = reinterpret_cast<const float4*>(x + rr * ldx)[q];
                xh[t] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                gv[t] = d;
                const float a0 = d.x * gm[t].x, a1 = d.y * gm[t].y, a2 = d.z * gm[t].z, a3 = d.w * gm[t].w;
                s1 += (a0 + a1) + (a2 + a3);
                s2 += (a0 * xh[t].x + a1 * xh[t].y) + (a2 * xh[t].z + a3 * xh[t].w);
            } else {
                gv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                xh[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float m1 = sum16(s1) * inv_c, m2 = sum16(s2) * inv_c;
        if (ok) {
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int q = l16 + 16 * t;
                if (q >= nq) continue;
                float o0 = rstd * (gv[t].x * gm[t].x - m1 - xh[t].x * m2), o1 = rstd * (gv[t].y * gm[t].y - m1 - xh[t].y * m2);
                float o2 = rstd * (gv[t].z * gm[t].z - m1 - xh[t].z * m2), o3 = rstd * (gv[t].w * gm[t].w - m1 - xh[t].w * m2);
                if (dres != nullptr) {
                    const float4 dr = *reinterpret_cast<const float4*>(dres + r * ldr + 4 * q);
                    o0 += dr.x; o1 += dr.y; o2 += dr.z; o3 += dr.w;
                }
                if (dx32 != nullptr) *reinterpret_cast<float4*>(dx32 + r * ldo + 4 * q) = make_float4(o0, o1, o2, o3);
                if (dx16 != nullptr)
                    *reinterpret_cast<f16x4*>(dx16 + r * ldo + 4 * q) = (f16x4){(_Float16)o0, (_Float16)o1, (_Float16)o2, (_Float16)o3};
                sg[t].x += gv[t].x * xh[t].x; sg[t].y += gv[t].y * xh[t].y; sg[t].z += gv[t].z * xh[t].z; sg[t].w += gv[t].w * xh[t].w;
                sb[t].x += gv[t].x; sb[t].y += gv[t].y; sb[t].z += gv[t].z; sb[t].w += gv[t].w;
            }
        }
    }
    // fold the 4 row slots of a wave (lanes l, l+16, l+32, l+48), then the 4 waves through LDS, fixed order
    __shared__ float red[4][2][NV * 64];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        float* pg = reinterpret_cast<float*>(&sg[t]);
        float* pb = reinterpret_cast<float*>(&sb[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = pg[e], bsum = pb[e];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            bsum += __shfl_xor(bsum, 16, 64);
            bsum += __shfl_xor(bsum, 32, 64);
            if (rsub == 0) {
                red[wv][0][(l16 + 16 * t) * 4 + e] = a;
                red[wv][1][(l16 + 16 * t) * 4 + e] = bsum;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        partial[((int64_t)blockIdx.x * 2 + which) * C + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3 (stride 1, zero padding 1) on [B][H][W][C] fp16.  A thread owns 8 channels of a strip of PX pixels
// along x: its 72 weights live in registers and the 3 x (PX + 2) input vectors are loaded once.
//   MODE 0 (forward):        u = dw(x) + bias -> u16 ; h = gelu(u) -> out16                 (MixTransfomer.py:72-74)
//   MODE 1 (backward-data):  out = dw_flipped(x) (x = du), no bias, no activation
// Weights arrive as wt[9][C] fp32 (tap-major), MODE 1 passes the taps flipped.
// ---------------------------------------------------------------------------------------------
constexpr int kDwPX = 4;

template <int MODE>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const _Float16* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, _Float16* __restrict__ u16,
                                                        _Float16* __restrict__ out16, int B, int H, int W, int C) {
    const int cg = C / 8;
    const int strips = (W + kDwPX - 1) / kDwPX;
    const int64_t total = (int64_t)B * H * strips * cg;
    // one item per thread; blocks b, b + 8, ... share an XCD (round-robin dispatch): the remap hands every XCD a contiguous
    // band of image rows, so the rows y - 1 / y + 1 a strip re-reads were fetched by the same L2 moments before (without
    // it: 505 MB of HBM traffic per launch against 283 MB algorithmic on the stage-3 tensor -- profiles/r03_mit_b5_pmc_summary.json)
    {
        const int64_t idx = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
        if (idx >= total) return;
        const int c8 = (int)(idx % cg);
        int64_t rest = idx / cg;
        const int sx = (int)(rest % strips);
        rest /= strips;
        const int y = (int)(rest % H);
        const int b = (int)(rest / H);
        const int x0 = sx * kDwPX;
        float w[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 w0 = *reinterpret_cast<const float4*>(wt + (int64_t)t * C + c8 * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(wt + (int64_t)t * C + c8 * 8 + 4);
            w[t][0] = w0.x; w[t][1] = w0.y; w[t][2] = w0.z; w[t][3] = w0.w;
            w[t][4] = w1.x; w[t][5] = w1.y; w[t][6] = w1.z; w[t][7] = w1.w;
        }
        float acc[kDwPX][8];
#pragma unroll
        for (int p = 0; p < kDwPX; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
        // all 3 x (PX + 2) input vectors are fetched up front from clamped addresses (no branch between the loads: they
        // issue back to back) and zeroed afterwards where the tap lies outside the image
        f16x8 in[3][kDwPX + 2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const int yc = min(max(yy, 0), H - 1);
            const _Float16* row = x + (((int64_t)b * H + yc) * W) * C + c8 * 8;
#pragma unroll
            for (int j = 0; j < kDwPX + 2; ++j) {
                const int xx = x0 + j - 1;
                in[dy][j] = *reinterpret_cast<const f16x8*>(row + (int64_t)min(max(xx, 0), W - 1) * C);
            }
        }
        const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool yok = yy >= 0 && yy < H;
#pragma unroll
            for (int j = 0; j < kDwPX + 2; ++j) {
                const int xx = x0 + j - 1;
                const f16x8 v = (yok && xx >= 0 && xx < W) ? in[dy][j] : zero8;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int p = j - dx;                          // output pixel x0 + p reads input x0 + p + dx - 1 = x0 + j - 1
                    if (p < 0 || p >= kDwPX) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[p][e] += w[dy * 3 + dx][e] * (float)v[e];
                }
            }
        }
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (MODE == 0 && bias != nullptr) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + c8 * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(bias + c8 * 8 + 4);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
#pragma unroll
        for (int p = 0; p < kDwPX; ++p) {
            if (x0 + p >= W) break;
            const int64_t o = (((int64_t)b * H + y) * W + x0 + p) * C + c8 * 8;
            f16x8 uo, ho;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = acc[p][e] + bv[e];
                uo[e] = (_Float16)u;
                // GELU of the ROUNDED pre-activation: the backward pass re-derives gelu'(u) from the stored fp16 u
                ho[e] = MODE == 0 ? (_Float16)gelu_f((float)uo[e]) : (_Float16)u;
            }
            if (MODE == 0 && u16 != nullptr) *reinterpret_cast<f16x8*>(u16 + o) = uo;
            *reinterpret_cast<f16x8*>(out16 + o) = ho;
        }
    }
}

// The same convolution with a sliding window over ROWS: a thread owns 8 channels of a (PX wide, R tall) column strip, keeps the three
// input rows of the current output row in registers and fetches ONE new row of PX + 2 vectors per output row (issued before the
// current row is computed).  dwconv3x3_kernel fetches 3 (PX + 2) vectors per PX outputs (4.5 per output: every input element goes
// through the L1 / L2 path 4.5 times -- 424 MB per stage-3 launch against 94 MB of input); here it is (R + 2)(PX + 2) / (R PX): 1.9
// at R = 8, 2.25 at R = 4.  R is chosen at launch so that the grid keeps >= ~1000 blocks.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void dwconv3x3_rows_kernel(const _Float16* __restrict__ x, const float* __restrict__ wt,
                                                             const float* __restrict__ bias, _Float16* __restrict__ u16,
                                                             _Float16* __restrict__ out16, int B, int H, int W, int C, int R) {
    const int cg = C / 8;
    const int strips = (W + kDwPX - 1) / kDwPX;
    const int rgroups = (H + R - 1) / R;
    const int64_t total = (int64_t)B * rgroups * strips * cg;
    const int64_t idx = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c8 = (int)(idx % cg);
    int64_t rest = idx / cg;
    const int sx = (int)(rest % strips);
    rest /= strips;
    const int rg = (int)(rest % rgroups);
    const int b = (int)(rest / rgroups);
    const int x0 = sx * kDwPX, y0 = rg * R, y1 = min(y0 + R, H);
    float w[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 w0 = *reinterpret_cast<const float4*>(wt + (int64_t)t * C + c8 * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(wt + (int64_t)t * C + c8 * 8 + 4);
        w[t][0] = w0.x; w[t][1] = w0.y; w[t][2] = w0.z; w[t][3] = w0.w;
        w[t][4] = w1.x; w[t][5] = w1.y; w[t][6] = w1.z; w[t][7] = w1.w;
    }
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 && bias != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + c8 * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(bias + c8 * 8 + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const _Float16* img = x + ((int64_t)b * H * W) * C + c8 * 8;
    // column offsets (clamped: always loadable) and validity of the PX + 2 columns of this strip
    int64_t coff[kDwPX + 2];
    bool cok[kDwPX + 2];
#pragma unroll
    for (int j = 0; j < kDwPX + 2; ++j) {
        const int xx = x0 + j - 1;
        cok[j] = xx >= 0 && xx < W;
        coff[j] = (int64_t)min(max(xx, 0), W - 1) * C;
    }
    auto load_row = [&](int yy, f16x8 (&row)[kDwPX + 2]) {
        const _Float16* rp = img + (int64_t)min(max(yy, 0), H - 1) * W * C;
#pragma unroll
        for (int j = 0; j < kDwPX + 2; ++j) row[j] = *reinterpret_cast<const f16x8*>(rp + coff[j]);
    };
    auto mask_row = [&](int yy, f16x8 (&row)[kDwPX + 2]) {
        const bool yok = yy >= 0 && yy < H;
#pragma unroll
        for (int j = 0; j < kDwPX + 2; ++j) row[j] = (yok && cok[j]) ? row[j] : zero8;
    };
    // three row buffers: output row y is computed from (ra, rb, rc) = input rows y - 1, y, y + 1; then the window moves down by register
    // moves (48 v_mov against 288 FMAs) and input row y + 2 is fetched into rc -- the next row's dy = 0 / 1 taps run under that load
    f16x8 ra[kDwPX + 2], rb[kDwPX + 2], rc[kDwPX + 2];
    load_row(y0 - 1, ra);
    load_row(y0, rb);
    load_row(y0 + 1, rc);
    mask_row(y0 - 1, ra);
    mask_row(y0, rb);
#pragma unroll 1
    for (int y = y0; y < y1; ++y) {
        float acc[kDwPX][8];
#pragma unroll
        for (int p = 0; p < kDwPX; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if (dy == 2) mask_row(y + 1, rc);                 // (fetched raw)
            f16x8 (&row)[kDwPX + 2] = dy == 0 ? ra : dy == 1 ? rb : rc;
#pragma unroll
            for (int j = 0; j < kDwPX + 2; ++j) {
                // v_fma_mix_f32 reads the fp16 halves where they are (op_sel picks the half): written as acc += w * (float)v the
                // compiler converts every input vector to fp32 once for its three taps -- 144 more live registers, spills at 256
                const u32x4 pk = __builtin_bit_cast(u32x4, row[j]);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int p = j - dx;
                    if (p < 0 || p >= kDwPX) continue;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[p][e]) : "v"(pk[e >> 1]), "v"(w[dy * 3 + dx][e]));
                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[p][e + 1]) : "v"(pk[e >> 1]), "v"(w[dy * 3 + dx][e + 1]));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kDwPX + 2; ++j) {
            ra[j] = rb[j];
            rb[j] = rc[j];
        }
        if (y + 1 < y1) load_row(y + 2, rc);
#pragma unroll
        for (int p = 0; p < kDwPX; ++p) {
            if (x0 + p >= W) break;
            const int64_t o = (((int64_t)b * H + y) * W + x0 + p) * C + c8 * 8;
            f16x8 uo, ho;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = acc[p][e] + bv[e];
                uo[e] = (_Float16)u;
                ho[e] = MODE == 0 ? (_Float16)gelu_f((float)uo[e]) : (_Float16)u;     // (GELU of the ROUNDED pre-activation, as above)
            }
            if (MODE == 0 && u16 != nullptr) *reinterpret_cast<f16x8*>(u16 + o) = uo;
            *reinterpret_cast<f16x8*>(out16 + o) = ho;
        }
    }
}

static int dw_rows_per_thread(int64_t B, int64_t H, int64_t W, int64_t C) {
    // the tallest strip (8, 4, 2 rows) that still leaves ~1000 blocks of 256 threads; 0: the one-row kernel
    const int64_t per_row = B * ceil_div(W, kDwPX) * (C / 8);
    for (int r = 8; r >= 2; r >>= 1)
        if (ceil_div(per_row * ceil_div(H, r), 256) >= 1000) return r;
    return 0;
}

// du = dh * gelu'(u) -> du16; per-block partial sums of the depthwise weight / bias gradients:
//   dw[t][c] = sum_p du[p][c] * x[p + t][c],  db[c] = sum_p du[p][c]     (x = the conv input, zero outside the image)
// A block covers `rows_per_block` image rows of one image with gw * phases threads (gw = min(C / 8, 256) channel groups;
// wider tensors loop): thread (group, phase) walks the contiguous x range of its phase with a sliding 3 x 3 window of input
// vectors (3 new loads per pixel instead of 9).  partial[block][10][C].
__global__ __launch_bounds__(512) void dwconv_bwd_prep_kernel(const _Float16* __restrict__ dh, const _Float16* __restrict__ u,
                                                              const _Float16* __restrict__ x, _Float16* __restrict__ du16,
                                                              float* __restrict__ partial, int B, int H, int W, int C,
                                                              int rows_per_block, int gw, int phases) {
    const int cg = C / 8;
    const int blocks_per_img = (H + rows_per_block - 1) / rows_per_block;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);              // neighbouring row groups on one XCD (shared halo rows in its L2)
    const int b = blk / blocks_per_img;
    const int y0 = (blk - b * blocks_per_img) * rows_per_block;
    const int y1 = min(y0 + rows_per_block, H);
    extern __shared__ float red[];                                 // [phases * gw][8]
    const int gi = threadIdx.x % gw, ph = threadIdx.x / gw;
    const int xa = (int)(((int64_t)W * ph) / phases), xb = (int)(((int64_t)W * (ph + 1)) / phases);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int gbase = 0; gbase < cg; gbase += gw) {
        const bool live = gbase + gi < cg;
        const int c0 = (gbase + (live ? gi : 0)) * 8;
        float s[10][8];
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[t][e] = 0.f;
        if (live) {
            for (int y = y0; y < y1; ++y) {
                // rows outside the image: clamped address (always loadable), value zeroed by the select below
                const _Float16* rows[3];
                bool rok[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    rok[dy] = yy >= 0 && yy < H;
                    rows[dy] = x + (((int64_t)b * H + min(max(yy, 0), H - 1)) * W) * C + c0;
                }
                f16x8 win[3][3];                                   // win[dy][k] = x[y + dy - 1][xx + k - 1]
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const f16x8 a1 = *reinterpret_cast<const f16x8*>(rows[dy] + (int64_t)max(xa - 1, 0) * C);
                    const f16x8 a2 = *reinterpret_cast<const f16x8*>(rows[dy] + (int64_t)min(xa, W - 1) * C);
                    win[dy][0] = zero8;
                    win[dy][1] = (rok[dy] && xa - 1 >= 0) ? a1 : zero8;
                    win[dy][2] = (rok[dy] && xa < W) ? a2 : zero8;
                }
                for (int xx = xa; xx < xb; ++xx) {
                    const int64_t o = (((int64_t)b * H + y) * W + xx) * C + c0;
                    const f16x8 g8 = *reinterpret_cast<const f16x8*>(dh + o);
                    const f16x8 u8 = *reinterpret_cast<const f16x8*>(u + o);
                    f16x8 nw[3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) nw[dy] = *reinterpret_cast<const f16x8*>(rows[dy] + (int64_t)min(xx + 1, W - 1) * C);
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        win[dy][0] = win[dy][1];
                        win[dy][1] = win[dy][2];
                        win[dy][2] = (rok[dy] && xx + 1 < W) ? nw[dy] : zero8;
                    }
                    f16x8 d8;
                    float d[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        d8[e] = (_Float16)((float)g8[e] * gelu_grad_f((float)u8[e]));
                        d[e] = (float)d8[e];
                        s[9][e] += d[e];
                    }
                    *reinterpret_cast<f16x8*>(du16 + o) = d8;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int e = 0; e < 8; ++e) s[dy * 3 + dx][e] += d[e] * (float)win[dy][dx][e];
                }
            }
        }
        // fold the x phases in fixed order, one tap at a time through LDS
        for (int t = 0; t < 10; ++t) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[t][e];
            __syncthreads();
            if (ph == 0 && live) {
                float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int p = 0; p < phases; ++p)
#pragma unroll
                    for (int e = 0; e < 8; ++e) tot[e] += red[(p * gw + gi) * 8 + e];
#pragma unroll
                for (int e = 0; e < 8; ++e) partial[((int64_t)blk * 10 + t) * C + (gbase + gi) * 8 + e] = tot[e];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Row gathers.  Patch (oy, ox) of image b = the R x S window at (oy * stride - pad, ox * stride - pad); its row in `cols`
// holds k = (ky * S + kx) * C + c for the taps inside the image, zeros elsewhere, zero-padded to Kp columns.
//   SRC 0: src fp32 [B][H][W][C]     SRC 1: src fp16 [B][H][W][C]     SRC 2: src fp32 NCHW [B][C][H][W] (the image)
// ---------------------------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ src, _Float16* __restrict__ cols, int B, int H, int W,
                                                     int C, int R, int S, int stride, int pad, int Ho, int Wo, int Kp) {
    if constexpr (SRC == 2) {
        const int64_t total = (int64_t)B * Ho * Wo * Kp;
        const float* s = static_cast<const float*>(src);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int k = (int)(i % Kp);
            int64_t m = i / Kp;
            const int ox = (int)(m % Wo);
            m /= Wo;
            const int oy = (int)(m % Ho);
            const int b = (int)(m / Ho);
            float v = 0.f;
            if (k < R * S * C) {
                const int tap = k / C, c = k - tap * C;
                const int ky = tap / S, kx = tap - ky * S;
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = s[(((int64_t)b * C + c) * H + iy) * W + ix];
            }
            cols[i] = (_Float16)v;
        }
    } else {
        const int cg = C / 8, kg = Kp / 8;
        const int64_t total = (int64_t)B * Ho * Wo * kg;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            const int g = (int)(i % kg);
            int64_t m = i / kg;
            const int ox = (int)(m % Wo);
            m /= Wo;
            const int oy = (int)(m % Ho);
            const int b = (int)(m / Ho);
            f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (g < R * S * cg) {
                const int tap = g / cg, c8 = g - tap * cg;
                const int ky = tap / S, kx = tap - ky * S;
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const int64_t off = (((int64_t)b * H + iy) * W + ix) * C + c8 * 8;
                    if constexpr (SRC == 0) {
                        const float4 v0 = *reinterpret_cast<const float4*>(static_cast<const float*>(src) + off);
                        const float4 v1 = *reinterpret_cast<const float4*>(static_cast<const float*>(src) + off + 4);
                        o = (f16x8){(_Float16)v0.x, (_Float16)v0.y, (_Float16)v0.z, (_Float16)v0.w,
                                    (_Float16)v1.x, (_Float16)v1.y, (_Float16)v1.z, (_Float16)v1.w};
                    } else {
                        o = *reinterpret_cast<const f16x8*>(static_cast<const _Float16*>(src) + off);
                    }
                }
            }
            *reinterpret_cast<f16x8*>(cols + i * 8) = o;
        }
    }
}

// Adjoint gather (deterministic, no atomics): dst[b][iy][ix][c] (+)= sum over the taps (ky, kx) with
// (iy + pad - ky) % stride == 0, (ix + pad - kx) % stride == 0 and the patch inside [0,Ho)x[0,Wo) of dcols[patch][(ky*S+kx)*C + c].
//   OUT32: dst fp32, overwritten (x gscale)      else: dst fp16, accumulated (dst += value)
template <bool OUT32>
__global__ __launch_bounds__(256) void col2im_kernel(const _Float16* __restrict__ dcols, void* __restrict__ dst, int B, int H, int W,
                                                     int C, int R, int S, int stride, int pad, int Ho, int Wo, int Kp, float gscale) {
    const int cg = C / 8;
    const int64_t total = (int64_t)B * H * W * cg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(i % cg);
        int64_t m = i / cg;
        const int ix = (int)(m % W);
        m /= W;
        const int iy = (int)(m % H);
        const int b = (int)(m / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ky = (iy + pad) % stride; ky < R; ky += stride) {
            const int oy = (iy + pad - ky) / stride;
            if (oy < 0 || oy >= Ho) continue;
            for (int kx = (ix + pad) % stride; kx < S; kx += stride) {
                const int ox = (ix + pad - kx) / stride;
                if (ox < 0 || ox >= Wo) continue;
                const f16x8 v = *reinterpret_cast<const f16x8*>(dcols + (((int64_t)b * Ho + oy) * Wo + ox) * Kp + (ky * S + kx) * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        }
        if constexpr (OUT32) {
            float* o = static_cast<float*>(dst) + i * 8;
            *reinterpret_cast<float4*>(o) = make_float4(acc[0] * gscale, acc[1] * gscale, acc[2] * gscale, acc[3] * gscale);
            *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4] * gscale, acc[5] * gscale, acc[6] * gscale, acc[7] * gscale);
        } else {
            _Float16* o = static_cast<_Float16*>(dst) + i * 8;
            const f16x8 old = *reinterpret_cast<const f16x8*>(o);
            f16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (_Float16)((float)old[e] + acc[e] * gscale);
            *reinterpret_cast<f16x8*>(o) = r;
        }
    }
}

// fp32 -> fp16 with a scale (gradients entering the fp16 backward pass carry the loss scale)
__global__ __launch_bounds__(256) void cast_scale_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int64_t n4, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<f16x4*>(y)[i] = (f16x4){(_Float16)(v.x * scale), (_Float16)(v.y * scale), (_Float16)(v.z * scale), (_Float16)(v.w * scale)};
    }
}

// y[m][:] = x[m][:] * seg_scale[m / rows_per_seg]  (fp16; the backward of a DropPath-scaled branch, MixTransfomer.py:176-177)
__global__ __launch_bounds__(256) void row_scale_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y,
                                                        const float* __restrict__ seg_scale, int64_t n8, int c8, int rows_per_seg) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float s = seg_scale[(i / c8) / rows_per_seg];
        const f16x8 v = reinterpret_cast<const f16x8*>(x)[i];
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)v[e] * s);
        reinterpret_cast<f16x8*>(y)[i] = o;
    }
}

}  // namespace mit
}  // namespace diga

using namespace diga;
using namespace diga::mit;

static inline unsigned grid_for(int64_t items, int per_block = 256, int64_t cap = 256 * 32) {
    int64_t b = ceil_div(items, per_block);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int diga_mit_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y16, float* y32,
                                      int64_t ldy, float* mean, float* rstd, int64_t M, int64_t C, float eps, void* stream) {
    DIGA_REQUIRE(x && gamma && beta && (y16 || y32) && M > 0 && C > 0, DIGA_EINVAL, "mit_layernorm_fwd: null pointer / empty shape");
    DIGA_REQUIRE(C % 4 == 0 && C <= 512 && ldx % 4 == 0 && ldy % 4 == 0 && M < (1ll << 31), DIGA_EINVAL,
                 "mit_layernorm_fwd: C %% 4 == 0, C <= 512 (C=%lld)", (long long)C);
    DIGA_REQUIRE((mean == nullptr) == (rstd == nullptr), DIGA_EINVAL, "mit_layernorm_fwd: mean and rstd go together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_NORM, st, (double)M * C * (4.0 + (y16 ? 2.0 : 0.0) + (y32 ? 4.0 : 0.0)));
    const unsigned grid = grid_for(M, 16);
    const int nv = (int)ceil_div(C, 64);
    _Float16* y = static_cast<_Float16*>(y16);
#define DIGA_LN_FWD(NV_) hipLaunchKernelGGL(layernorm_fwd_kernel<NV_>, dim3(grid), dim3(256), 0, st, x, ldx, gamma, beta, y, y32, ldy, mean, rstd, (int)M, (int)C, eps)
    if (nv <= 1) DIGA_LN_FWD(1);
    else if (nv == 2) DIGA_LN_FWD(2);
    else if (nv <= 5) DIGA_LN_FWD(5);
    else DIGA_LN_FWD(8);
#undef DIGA_LN_FWD
    return launch_status("mit_layernorm_fwd");
}

static int ln_rows_per_block(int64_t M) {                // ~512 blocks where the matrix is big enough, 16..256 rows each
    // measured (tools/bench_mit_ops.py --cold, MiT-B5 sizes): ~512 blocks beat 1024 / 2048 / 4096 on every stage
    int64_t r = ceil_div(M, 512);
    r = ceil_div(r, 16) * 16;
    if (r < 16) r = 16;
    if (r > 256) r = 256;
    return (int)r;
}

extern "C" size_t diga_mit_layernorm_bwd_workspace_bytes(int64_t M, int64_t C) {
    if (M <= 0 || C <= 0) return 0;
    return (size_t)ceil_div(M, ln_rows_per_block(M)) * 2 * (size_t)C * sizeof(float);
}

extern "C" int diga_mit_layernorm_bwd(const void* dy, int dy_is_f32, int64_t ldg, float gscale, const float* x, int64_t ldx,
                                      const float* gamma, const float* mean, const float* rstd, const float* dres, int64_t ldr,
                                      float* dx32, void* dx16, int64_t ldo, float* dgamma, float* dbeta, float param_scale,
                                      int accumulate, void* workspace, size_t workspace_bytes, int64_t M, int64_t C, void* stream) {
    DIGA_REQUIRE(dy && x && gamma && mean && rstd && (dx32 || dx16) && dgamma && dbeta && workspace && M > 0 && C > 0, DIGA_EINVAL,
                 "mit_layernorm_bwd: null pointer / empty shape");
    DIGA_REQUIRE(C % 4 == 0 && C <= 512 && ldx % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && M < (1ll << 31), DIGA_EINVAL,
                 "mit_layernorm_bwd: C %% 4 == 0, C <= 512");
    const int rpb = ln_rows_per_block(M);
    const int blocks = (int)ceil_div(M, rpb);
    DIGA_REQUIRE(workspace_bytes >= (size_t)blocks * 2 * (size_t)C * sizeof(float), DIGA_EWORKSPACE, "mit_layernorm_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_NORM, st, (double)M * C * ((dy_is_f32 ? 4.0 : 2.0) + 4.0 + (dres ? 4.0 : 0.0) + (dx32 ? 4.0 : 0.0) + (dx16 ? 2.0 : 0.0)));
    float* part = static_cast<float*>(workspace);
    _Float16* d16 = static_cast<_Float16*>(dx16);
    const int nv = (int)ceil_div(C, 64);
#define DIGA_LN_BWD(NV_, GT_) hipLaunchKernelGGL((layernorm_bwd_kernel<NV_, GT_>), dim3(blocks), dim3(256), 0, st, static_cast<const GT_*>(dy), ldg, x, ldx, gamma, mean, rstd, dres, ldr, dx32, d16, ldo, part, (int)M, (int)C, rpb, gscale)
#define DIGA_LN_BWD_T(GT_)                \
    do {                                  \
        if (nv <= 1) DIGA_LN_BWD(1, GT_); \
        else if (nv == 2) DIGA_LN_BWD(2, GT_); \
        else if (nv <= 5) DIGA_LN_BWD(5, GT_); \
        else DIGA_LN_BWD(8, GT_);         \
    } while (0)
    if (dy_is_f32) DIGA_LN_BWD_T(float);
    else DIGA_LN_BWD_T(_Float16);
#undef DIGA_LN_BWD_T
#undef DIGA_LN_BWD
    if (blocks >= 64)             // few values, many chunks: 8 columns x 32 chunk phases per block
        hipLaunchKernelGGL((partial_reduce_kernel<1, 8>), dim3((unsigned)ceil_div(2 * C, 8)), dim3(256), 0, st, part, blocks, (int)(2 * C), dgamma, dbeta,
                           (int)C, param_scale, accumulate, (const float*)nullptr, 0);
    else
        hipLaunchKernelGGL((partial_reduce_kernel<1, 32>), dim3((unsigned)ceil_div(2 * C, 32)), dim3(256), 0, st, part, blocks, (int)(2 * C), dgamma, dbeta,
                           (int)C, param_scale, accumulate, (const float*)nullptr, 0);
    return launch_status("mit_layernorm_bwd");
}

extern "C" int diga_mit_dwconv_gelu_fwd(const void* x, const float* wt9, const float* bias, void* u16, void* h16, int64_t B, int64_t H,
                                        int64_t W, int64_t C, void* stream) {
    DIGA_REQUIRE(x && wt9 && h16 && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, DIGA_EINVAL, "mit_dwconv_gelu_fwd: bad argument");
    DIGA_REQUIRE(aligned16(x) && aligned16(h16) && aligned16(wt9), DIGA_EALIGN, "mit_dwconv_gelu_fwd: alignment");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t items = B * H * ceil_div(W, kDwPX) * (C / 8);
    ProfScope prof(DIGA_PROF_MIT_DWCONV, st, (double)B * H * W * C * (u16 ? 6.0 : 4.0));
    const int R = dw_rows_per_thread(B, H, W, C);
    if (R > 0) {
        const int64_t ritems = B * ceil_div(H, R) * ceil_div(W, kDwPX) * (C / 8);
        hipLaunchKernelGGL(dwconv3x3_rows_kernel<0>, dim3((unsigned)ceil_div(ritems, 256)), dim3(256), 0, st, static_cast<const _Float16*>(x), wt9,
                           bias, static_cast<_Float16*>(u16), static_cast<_Float16*>(h16), (int)B, (int)H, (int)W, (int)C, R);
    } else {
        hipLaunchKernelGGL(dwconv3x3_kernel<0>, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, st, static_cast<const _Float16*>(x), wt9, bias,
                           static_cast<_Float16*>(u16), static_cast<_Float16*>(h16), (int)B, (int)H, (int)W, (int)C);
    }
    return launch_status("mit_dwconv_gelu_fwd");
}

namespace diga {
namespace mit {

constexpr int kPrepPX = 2;      // strip width of the kernel below (4: 17 registers spilled at the 256 of two waves per SIMD)

// dwconv_bwd_prep_kernel in the row-sliding form of dwconv3x3_rows_kernel: a block = gw channel groups (lanes: 16 contiguous bytes
// each) x 256 / gw strip lanes over R image rows of one image; a thread walks its strips (4 pixels wide) down the R rows with the
// three-row window of x in registers -- per row 2 dh + 2 u + 4 new x vectors, ALL issued before the row's arithmetic (the walk along x
// of the kernel above has one dependent 5-load round trip per pixel in flight: 175 us on the stage-3 tensor, latency-bound).
// 144 v_fma_mix_f32 per row accumulate the nine tap sums; the strip lanes are folded through LDS in fixed order.
// partial[part = image, row group][10][C].
__global__ __launch_bounds__(256, 2) void dwconv_bwd_prep_rows_kernel(const _Float16* __restrict__ dh, const _Float16* __restrict__ u,
                                                                      const _Float16* __restrict__ x, _Float16* __restrict__ du16,
                                                                      float* __restrict__ partial, int B, int H, int W, int C, int R, int gw) {
    __shared__ float red[256 * 8];
    const int cg = C / 8;
    const int SL = 256 / gw;
    const int gi = threadIdx.x % gw, sl = threadIdx.x / gw;
    const int rgroups = (H + R - 1) / R;
    const int part = xcd_remap(blockIdx.y, gridDim.y);           // neighbouring row groups on one XCD (shared halo rows in its L2)
    const int b = part / rgroups, rg = part - b * rgroups;
    const int y0 = rg * R, y1 = min(y0 + R, H);
    const int c8 = blockIdx.x * gw + gi;
    const bool live = c8 < cg;
    const int strips = (W + kPrepPX - 1) / kPrepPX;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float s[10][8];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[t][e] = 0.f;
    if (live) {
        const int64_t img = ((int64_t)b * H * W) * C + c8 * 8;
        for (int sx = sl; sx < strips; sx += SL) {
            const int x0 = sx * kPrepPX;
            int coff[kPrepPX + 2];                               // (element offsets inside an image row: W * C < 2^31)
            bool cok[kPrepPX + 2];
#pragma unroll
            for (int j = 0; j < kPrepPX + 2; ++j) {
                const int xx = x0 + j - 1;
                cok[j] = xx >= 0 && xx < W;
                coff[j] = min(max(xx, 0), W - 1) * C;
            }
            auto load_row = [&](int yy, f16x8 (&row)[kPrepPX + 2]) {
                const _Float16* rp = x + img + (int64_t)min(max(yy, 0), H - 1) * W * C;
#pragma unroll
                for (int j = 0; j < kPrepPX + 2; ++j) row[j] = *reinterpret_cast<const f16x8*>(rp + coff[j]);
            };
            auto mask_row = [&](int yy, f16x8 (&row)[kPrepPX + 2]) {
                const bool yok = yy >= 0 && yy < H;
#pragma unroll
                for (int j = 0; j < kPrepPX + 2; ++j) row[j] = (yok && cok[j]) ? row[j] : zero8;
            };
            f16x8 ra[kPrepPX + 2], rb[kPrepPX + 2], rc[kPrepPX + 2];
            load_row(y0 - 1, ra);
            load_row(y0, rb);
            load_row(y0 + 1, rc);
            mask_row(y0 - 1, ra);
            mask_row(y0, rb);
#pragma unroll 1
            for (int y = y0; y < y1; ++y) {
                const int64_t o = img + (int64_t)y * W * C;
                f16x8 g8[kPrepPX], u8[kPrepPX];
#pragma unroll
                for (int p = 0; p < kPrepPX; ++p) {
                    g8[p] = *reinterpret_cast<const f16x8*>(dh + o + coff[p + 1]);
                    u8[p] = *reinterpret_cast<const f16x8*>(u + o + coff[p + 1]);
                }
                mask_row(y + 1, rc);                          // (fetched raw)
#pragma unroll
                for (int p = 0; p < kPrepPX; ++p) {
                    const bool pok = x0 + p < W;
                    f16x8 d8;
                    float d[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        d8[e] = pok ? (_Float16)((float)g8[p][e] * gelu_grad_f((float)u8[p][e])) : (_Float16)0.f;
                        d[e] = (float)d8[e];
                        s[9][e] += d[e];
                    }
                    if (pok) *reinterpret_cast<f16x8*>(du16 + o + coff[p + 1]) = d8;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        f16x8 (&row)[kPrepPX + 2] = dy == 0 ? ra : dy == 1 ? rb : rc;
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const u32x4 pk = __builtin_bit_cast(u32x4, row[p + dx]);     // x[y + dy - 1][x0 + p + dx - 1]
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(s[dy * 3 + dx][e]) : "v"(pk[e >> 1]), "v"(d[e]));
                                asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(s[dy * 3 + dx][e + 1]) : "v"(pk[e >> 1]), "v"(d[e + 1]));
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < kPrepPX + 2; ++j) {
                    ra[j] = rb[j];
                    rb[j] = rc[j];
                }
                if (y + 1 < y1) load_row(y + 2, rc);
            }
        }
    }
    // fold the strip lanes in fixed order, one tap at a time through LDS
    for (int t = 0; t < 10; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[t][e];
        __syncthreads();
        if (sl == 0 && live) {
            float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < SL; ++p)
#pragma unroll
                for (int e = 0; e < 8; ++e) tot[e] += red[(p * gw + gi) * 8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) partial[((int64_t)part * 10 + t) * C + c8 * 8 + e] = tot[e];
        }
    }
}

}  // namespace mit
}  // namespace diga

static int dw_rows_per_block(int64_t B, int64_t H) {     // ~1500+ blocks when the tensor has that many image rows
    int64_t r = (B * H) / 1536;
    if (r < 1) r = 1;
    if (r > 8) r = 8;
    return (int)r;
}

extern "C" size_t diga_mit_dwconv_bwd_workspace_bytes(int64_t B, int64_t H, int64_t C) {
    if (B <= 0 || H <= 0 || C <= 0) return 0;
    return (size_t)B * (size_t)ceil_div(H, dw_rows_per_block(B, H)) * 10 * (size_t)C * sizeof(float);
}

/* dh: gradient wrt gelu output, u: saved pre-activation, x: the conv input (fc1 output); du16: scratch for du [B,H,W,C];
 * wt9_flipped: the taps in reverse order ([8 - t][C]); dx16 = dw^T(du); dw [C][9] / db [C] get scale * sums. */
extern "C" int diga_mit_dwconv_gelu_bwd(const void* dh, const void* u, const void* x, const float* wt9_flipped, void* du16, void* dx16,
                                        float* dw, float* db, float param_scale, int accumulate, void* workspace, size_t workspace_bytes,
                                        int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    DIGA_REQUIRE(dh && u && x && wt9_flipped && du16 && dx16 && dw && db && workspace && B > 0 && H > 0 && W > 0 && C % 8 == 0 && C > 0,
                 DIGA_EINVAL, "mit_dwconv_gelu_bwd: bad argument");
    const int rpb = dw_rows_per_block(B, H);
    int blocks = (int)(B * ceil_div(H, rpb));
    DIGA_REQUIRE(workspace_bytes >= (size_t)blocks * 10 * (size_t)C * sizeof(float), DIGA_EWORKSPACE, "mit_dwconv_gelu_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_DWCONV, st, (double)B * H * W * C * 12.0);
    // row-sliding form where the tensor is large enough for >= 512 blocks of (32 or 64 channel groups) x (8 or 4 strip lanes) x R rows;
    // R >= rpb keeps the partial sums inside the workspace the caller sized with diga_mit_dwconv_bwd_workspace_bytes
    // (gw, R) by a cost model of the launch: two blocks per CU (206 registers) = 512 slots per round; a block's work is its threads'
    // strips x (R rows + the halo / prologue share of a walk); dead lanes of the last channel-group block and a mostly empty last round
    // are what the model avoids -- the stage-3 tensor (160 channel groups, 48 rows x 16 images) went 3 x 64 groups x 4 rows = 576 blocks
    // (two rounds for 1.1 rounds of work, a quarter of the third block's lanes dead) -> 5 x 32 groups x 8 rows = 480 blocks, one round
    const int cgn = (int)(C / 8);
    const int strips = (int)ceil_div(W, mit::kPrepPX);
    int gwr = 32, Rr = 0;
    double best = 0.0;
    for (int gw = 64; gw >= 32; gw >>= 1)
        for (int r = 8; r >= 1; r >>= 1) {
            if (r < rpb) continue;
            const int64_t nb = ceil_div(cgn, gw) * B * ceil_div(H, r);
            if (nb < 256) continue;                               // less than half a round: the other kernel form
            const double cost = (double)ceil_div(nb, 512) * (double)ceil_div(strips, 256 / gw) * ((double)r + 0.4);
            if (Rr == 0 || cost < best) {
                best = cost;
                gwr = gw;
                Rr = r;
            }
        }
    if (Rr > 0 && cgn >= 32) {
        blocks = (int)(B * ceil_div(H, Rr));
        hipLaunchKernelGGL(dwconv_bwd_prep_rows_kernel, dim3((unsigned)ceil_div(cgn, gwr), (unsigned)blocks), dim3(256), 0, st,
                           static_cast<const _Float16*>(dh), static_cast<const _Float16*>(u), static_cast<const _Float16*>(x),
                           static_cast<_Float16*>(du16), static_cast<float*>(workspace), (int)B, (int)H, (int)W, (int)C, Rr, gwr);
    } else {
        const int gw = (int)(C / 8 < 256 ? C / 8 : 256);
        int phases = 512 / gw;
        if (phases > W) phases = (int)W;
        if (phases < 1) phases = 1;
        hipLaunchKernelGGL(dwconv_bwd_prep_kernel, dim3(blocks), dim3(gw * phases), (size_t)gw * phases * 8 * sizeof(float), st,
                           static_cast<const _Float16*>(dh), static_cast<const _Float16*>(u), static_cast<const _Float16*>(x),
                           static_cast<_Float16*>(du16), static_cast<float*>(workspace), (int)B, (int)H, (int)W, (int)C, rpb, gw, phases);
    }
    if (blocks >= 64)
        hipLaunchKernelGGL((partial_reduce_kernel<2, 8>), dim3((unsigned)ceil_div(10 * C, 8)), dim3(256), 0, st, static_cast<const float*>(workspace), blocks,
                           (int)(10 * C), dw, db, (int)C, param_scale, accumulate, (const float*)nullptr, 0);
    else
        hipLaunchKernelGGL((partial_reduce_kernel<2, 32>), dim3((unsigned)ceil_div(10 * C, 32)), dim3(256), 0, st, static_cast<const float*>(workspace), blocks,
                           (int)(10 * C), dw, db, (int)C, param_scale, accumulate, (const float*)nullptr, 0);
    const int64_t items = B * H * ceil_div(W, kDwPX) * (C / 8);
    const int R = dw_rows_per_thread(B, H, W, C);
    if (R > 0) {
        const int64_t ritems = B * ceil_div(H, R) * ceil_div(W, kDwPX) * (C / 8);
        hipLaunchKernelGGL(dwconv3x3_rows_kernel<1>, dim3((unsigned)ceil_div(ritems, 256)), dim3(256), 0, st, static_cast<const _Float16*>(du16),
                           wt9_flipped, (const float*)nullptr, (_Float16*)nullptr, static_cast<_Float16*>(dx16), (int)B, (int)H, (int)W, (int)C, R);
    } else {
        hipLaunchKernelGGL(dwconv3x3_kernel<1>, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, st, static_cast<const _Float16*>(du16), wt9_flipped,
                           (const float*)nullptr, (_Float16*)nullptr, static_cast<_Float16*>(dx16), (int)B, (int)H, (int)W, (int)C);
    }
    return launch_status("mit_dwconv_gelu_bwd");
}

extern "C" int diga_mit_im2col(const void* src, int src_kind, void* cols, int64_t B, int64_t H, int64_t W, int64_t C, int64_t R, int64_t S,
                               int64_t stride, int64_t pad, int64_t Ho, int64_t Wo, int64_t Kp, void* stream) {
    DIGA_REQUIRE(src && cols && B > 0 && H > 0 && W > 0 && C > 0 && R > 0 && S > 0 && stride > 0 && Ho > 0 && Wo > 0, DIGA_EINVAL,
                 "mit_im2col: bad argument");
    DIGA_REQUIRE(Kp >= R * S * C && Kp % 8 == 0 && src_kind >= 0 && src_kind <= 2 && (src_kind == 2 || C % 8 == 0), DIGA_EINVAL,
                 "mit_im2col: Kp %% 8 == 0, Kp >= R*S*C, C %% 8 == 0 for NHWC sources");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_MISC, st, (double)B * Ho * Wo * Kp * 2.0 * 2.0);
    _Float16* c = static_cast<_Float16*>(cols);
    if (src_kind == 2) {
        hipLaunchKernelGGL(im2col_kernel<2>, dim3(grid_for(B * Ho * Wo * Kp)), dim3(256), 0, st, src, c, (int)B, (int)H, (int)W, (int)C, (int)R,
                           (int)S, (int)stride, (int)pad, (int)Ho, (int)Wo, (int)Kp);
    } else if (src_kind == 0) {
        hipLaunchKernelGGL(im2col_kernel<0>, dim3(grid_for(B * Ho * Wo * (Kp / 8))), dim3(256), 0, st, src, c, (int)B, (int)H, (int)W, (int)C,
                           (int)R, (int)S, (int)stride, (int)pad, (int)Ho, (int)Wo, (int)Kp);
    } else {
        hipLaunchKernelGGL(im2col_kernel<1>, dim3(grid_for(B * Ho * Wo * (Kp / 8))), dim3(256), 0, st, src, c, (int)B, (int)H, (int)W, (int)C,
                           (int)R, (int)S, (int)stride, (int)pad, (int)Ho, (int)Wo, (int)Kp);
    }
    return launch_status("mit_im2col");
}

extern "C" int diga_mit_col2im(const void* dcols, void* dst, int dst_f32, float gscale, int64_t B, int64_t H, int64_t W, int64_t C, int64_t R,
                               int64_t S, int64_t stride, int64_t pad, int64_t Ho, int64_t Wo, int64_t Kp, void* stream) {
    DIGA_REQUIRE(dcols && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && Kp >= R * S * C && Kp % 8 == 0 && stride > 0, DIGA_EINVAL,
                 "mit_col2im: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_MISC, st, (double)B * Ho * Wo * Kp * 2.0 + (double)B * H * W * C * (dst_f32 ? 4.0 : 4.0));
    const unsigned grid = grid_for(B * H * W * (C / 8));
    if (dst_f32)
        hipLaunchKernelGGL(col2im_kernel<true>, dim3(grid), dim3(256), 0, st, static_cast<const _Float16*>(dcols), dst, (int)B, (int)H, (int)W,
                           (int)C, (int)R, (int)S, (int)stride, (int)pad, (int)Ho, (int)Wo, (int)Kp, gscale);
    else
        hipLaunchKernelGGL(col2im_kernel<false>, dim3(grid), dim3(256), 0, st, static_cast<const _Float16*>(dcols), dst, (int)B, (int)H, (int)W,
                           (int)C, (int)R, (int)S, (int)stride, (int)pad, (int)Ho, (int)Wo, (int)Kp, gscale);
    return launch_status("mit_col2im");
}

extern "C" int diga_mit_cast_scale(const float* x, void* y16, int64_t n, float scale, void* stream) {
    DIGA_REQUIRE(x && y16 && n > 0 && n % 4 == 0, DIGA_EINVAL, "mit_cast_scale: n %% 4 == 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_MISC, st, (double)n * 6.0);
    hipLaunchKernelGGL(cast_scale_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, x, static_cast<_Float16*>(y16), n / 4, scale);
    return launch_status("mit_cast_scale");
}

extern "C" int diga_mit_row_scale(const void* x16, void* y16, const float* seg_scale, int64_t rows_per_seg, int64_t M, int64_t C,
                                  void* stream) {
    DIGA_REQUIRE(x16 && y16 && seg_scale && rows_per_seg > 0 && M > 0 && C > 0 && C % 8 == 0, DIGA_EINVAL, "mit_row_scale: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_MISC, st, (double)M * C * 4.0);
    hipLaunchKernelGGL(row_scale_kernel, dim3(grid_for(M * C / 8)), dim3(256), 0, st, static_cast<const _Float16*>(x16),
                       static_cast<_Float16*>(y16), seg_scale, M * C / 8, (int)(C / 8), (int)rows_per_seg);
    return launch_status("mit_row_scale");
}
