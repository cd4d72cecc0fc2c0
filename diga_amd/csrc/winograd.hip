// Winograd F(2x2, 3x3) in fp32 for the stride-1 "same" 3x3 convolutions of the trunk and the ASPP head (dilation d,
// padding d): layer3/layer4 conv2 (d = 2, 4), the ASPP branches (d = 6, 12, 18) and the ASPP bottleneck (d = 1) -- 64 %
// of the model's convolution FLOPs.  Reference: the nn.Conv2d(3x3) layers of G5/model/seg_model_noaux.py:66-70,
// 143-150,166-170, which the reference runs through cuDNN (whose fp32 algorithm choice for these shapes is Winograd too).
//
// A dilated 3x3 convolution with padding = dilation is d*d independent dense 3x3 convolutions on the sub-images
// {(a + d*i, b + d*j)} (phase (a, b)).  Each sub-image is cut into 2x2 output tiles; per tile and input channel the 4x4
// input patch becomes V = B^T d B (wino_input_kernel), per (output, input) channel the 3x3 filter becomes U = G g G^T
// (wino_weight_kernel), the 16 element-wise products summed over the input channels are 16 independent GEMMs
//   M_k [tiles x Cout] = V_k [tiles x Cin] * U_k^T [Cin x Cout]         (16 multiplications per 2x2 outputs instead of 36)
// run as ONE launch of conv_fwd_dma_kernel (exact-fp32 MFMA, LDS-DMA operands; the weight panel is picked per 256-row
// tile), and y = A^T M A (+ bias) (wino_output_kernel).  B, G, A hold 0, +-1, +-1/2 only: the result differs from the
// direct convolution by rounding alone (Lavin & Gray 2016 measure F(2x2,3x3) fp32 error BELOW direct convolution's);
// tests hold both against a float64 convolution with the same bound.
//
// Workspace: [tile table int4 x Tp][U 16 x Cout x Cin][V 16 x Tp x Cin][M 16 x Tp x Cout], Tp = tiles rounded up to
// 256 (rows of padding tiles are zero).  HBM traffic on top of the GEMM: V is written and read once (4x the input
// tensor), M likewise (4x the output) -- both transforms are plain bandwidth passes.
#include "common.h"

namespace diga {

int gemm_batched_f32_dma(const float* A, int64_t rows_per_batch, int batches, int64_t K, const float* W, int64_t Cout,
                         float* out, hipStream_t st);     // conv.hip
size_t wgrad_batched_slab_bytes(int64_t rows, int batches, int64_t Cout, int64_t Cin);                                   // conv.hip
int wgrad_batched_f32_dma(const float* Z, const float* V, float* dU, float* slab, int64_t rows, int batches, int64_t Cout,
                          int64_t Cin, hipStream_t st);                                                                 // conv.hip

namespace wino {

struct WinoGeom {
    int N, H, W, d;
    int tys, txs;          // tile rows / columns summed over the d phases
    int64_t T, Tp;
};

int phase_tiles(int len, int d) {
    int s = 0;
    for (int a = 0; a < d; ++a) {
        const int n = len > a ? (len - a + d - 1) / d : 0;
        s += (n + 1) / 2;
    }
    return s;
}

WinoGeom make_wino(int64_t N, int64_t H, int64_t W, int64_t d) {
    WinoGeom g;
    g.N = (int)N; g.H = (int)H; g.W = (int)W; g.d = (int)d;
    g.tys = phase_tiles((int)H, (int)d);
    g.txs = phase_tiles((int)W, (int)d);
    g.T = N * g.tys * g.txs;
    g.Tp = ceil_div(g.T, 256) * 256;
    return g;
}

// tab[t] = {image, oy, ox, 0}: top-left OUTPUT pixel of tile t (its 2x2 outputs are (oy + d i, ox + d j), its 4x4 input
// patch (oy + d (i - 1), ox + d (j - 1))); image = -1 for the padding tiles
__global__ __launch_bounds__(256) void wino_tiles_kernel(int4* __restrict__ tab, WinoGeom g) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Tp) return;
    if (t >= g.T) {
        tab[t] = make_int4(-1, 0, 0, 0);
        return;
    }
    const int timg = g.tys * g.txs;
    const int n = (int)(t / timg), r = (int)(t - (int64_t)n * timg);
    int R = r / g.txs, Cc = r - R * g.txs;
    int a = 0, b = 0;
    for (; a < g.d; ++a) {
        const int cnt = g.H > a ? (g.H - a + g.d - 1) / g.d : 0;
        const int tl = (cnt + 1) / 2;
        if (R < tl) break;
        R -= tl;
    }
    for (; b < g.d; ++b) {
        const int cnt = g.W > b ? (g.W - b + g.d - 1) / g.d : 0;
        const int tl = (cnt + 1) / 2;
        if (Cc < tl) break;
        Cc -= tl;
    }
    tab[t] = make_int4(n, a + 2 * R * g.d, b + 2 * Cc * g.d, 0);
}

using f32x4nt = __attribute__((ext_vector_type(4))) float;
// V and M are written once and read once by another kernel, hundreds of MB to GB each: keep them out of L2
__device__ __forceinline__ void nt_store4(float* p, float4 v) {
    __builtin_nontemporal_store((f32x4nt){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4nt*>(p));
}
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const f32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4half(float4 a) { return make_float4(0.5f * a.x, 0.5f * a.y, 0.5f * a.z, 0.5f * a.w); }

// U[k = 4 i + j][co][c] = (G g G^T)[i][j], g = w[co][.][.][c] (flip: g[r][s] = w[co][2 - r][2 - s][c], backward-data)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin,
                                                          int flip) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4n = Cin / 4;
    if (idx >= (int64_t)Cout * c4n) return;
    const int co = (int)(idx / c4n), c = (int)(idx - (int64_t)co * c4n) * 4;
    float4 g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int rr = flip ? 2 - r : r, ss = flip ? 2 - s : s;
            g[r][s] = *reinterpret_cast<const float4*>(w + ((int64_t)co * 9 + rr * 3 + ss) * Cin + c);
        }
    float4 t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float4 sum = f4add(g[0][s], g[2][s]);
        t[0][s] = g[0][s];
        t[1][s] = f4half(f4add(sum, g[1][s]));
        t[2][s] = f4half(f4sub(sum, g[1][s]));
        t[3][s] = g[2][s];
    }
    const int64_t plane = (int64_t)Cout * Cin;
    float* o = U + (int64_t)co * Cin + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 sum = f4add(t[i][0], t[i][2]);
        *reinterpret_cast<float4*>(o + (4 * i + 0) * plane) = t[i][0];
        *reinterpret_cast<float4*>(o + (4 * i + 1) * plane) = f4half(f4add(sum, t[i][1]));
        *reinterpret_cast<float4*>(o + (4 * i + 2) * plane) = f4half(f4sub(sum, t[i][1]));
        *reinterpret_cast<float4*>(o + (4 * i + 3) * plane) = t[i][2];
    }
}

// V[k][t][c] = (B^T d B)[i][j] of the 4x4 patch of tile t, channel c (zero outside the image / for padding tiles)
// ab != nullptr: the input is the PRE-activation tensor of a train-mode BatchNorm + ReLU without residual; the transform reads
// relu(fma(x, a[c], b[c])) (ab = [2][C], the BatchNorm's forward coefficients, the expression affine_apply_kernel evaluates) --
// the activated tensor is never written or re-read (out-of-image taps stay exact zeros).
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int64_t ld, const int4* __restrict__ tab,
                                                         float* __restrict__ V, int64_t Tp, int C, int H, int W, int d,
                                                         const float* __restrict__ ab) {
    const int c4n = C / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tp * c4n) return;
    const int64_t t = idx / c4n;
    const int c = (int)(idx - t * c4n) * 4;
    const int4 e = tab[t];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 av = make_float4(1.f, 1.f, 1.f, 1.f), bv = z;
    if (ab != nullptr) {
        av = *reinterpret_cast<const float4*>(ab + c);
        bv = *reinterpret_cast<const float4*>(ab + C + c);
    }
    float4 p[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = e.y + (i - 1) * d;
        const bool yok = e.x >= 0 && (unsigned)y < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = e.z + (j - 1) * d;
            const bool ok = yok && (unsigned)xx < (unsigned)W;
            float4 v = ok ? *reinterpret_cast<const float4*>(x + ((int64_t)(e.x * H + y) * W + xx) * ld + c) : z;
            if (ab != nullptr && ok) {
                v.x = fmaxf(__builtin_fmaf(v.x, av.x, bv.x), 0.f); v.y = fmaxf(__builtin_fmaf(v.y, av.y, bv.y), 0.f);
                v.z = fmaxf(__builtin_fmaf(v.z, av.z, bv.z), 0.f); v.w = fmaxf(__builtin_fmaf(v.w, av.w, bv.w), 0.f);
            }
            p[i][j] = v;
        }
    }
    float4 m[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[0][j] = f4sub(p[0][j], p[2][j]);
        m[1][j] = f4add(p[1][j], p[2][j]);
        m[2][j] = f4sub(p[2][j], p[1][j]);
        m[3][j] = f4sub(p[1][j], p[3][j]);
    }
    float* o = V + t * C + c;
    const int64_t plane = Tp * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        nt_store4(o + (4 * i + 0) * plane, f4sub(m[i][0], m[i][2]));
        nt_store4(o + (4 * i + 1) * plane, f4add(m[i][1], m[i][2]));
        nt_store4(o + (4 * i + 2) * plane, f4sub(m[i][2], m[i][1]));
        nt_store4(o + (4 * i + 3) * plane, f4sub(m[i][1], m[i][3]));
    }
}

// y[2x2 of tile t][co] = A^T M A + bias
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, const int4* __restrict__ tab,
                                                          const float* __restrict__ bias, float* __restrict__ y, int64_t ld,
                                                          int64_t T, int64_t Tp, int K, int H, int W, int d) {
    const int k4n = K / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * k4n) return;
    const int64_t t = idx / k4n;
    const int k = (int)(idx - t * k4n) * 4;
    const int4 e = tab[t];
    const int64_t plane = Tp * K;
    const float* src = Mb + t * K + k;
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = nt_load4(src + (4 * i + j) * plane);
    float4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
        s[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
    }
    const float4 b = bias != nullptr ? *reinterpret_cast<const float4*>(bias + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int yy = e.y + i * d;
        if (yy >= H) continue;
        const float4 o0 = f4add(f4add(f4add(s[i][0], s[i][1]), s[i][2]), b);
        const float4 o1 = f4add(f4sub(f4sub(s[i][1], s[i][2]), s[i][3]), b);
        float* row = y + ((int64_t)(e.x * H + yy) * W) * ld + k;
        *reinterpret_cast<float4*>(row + (int64_t)e.z * ld) = o0;
        if (e.z + d < W) *reinterpret_cast<float4*>(row + (int64_t)(e.z + d) * ld) = o1;
    }
}

// wino_output_kernel for a backward-data convolution with the fused epilogue of diga_bwd_epilogue_t (same arithmetic, element
// by element, as drain_stage<EPI> in conv.hip): out = mask(A^T M A + addend) and the BatchNorm-backward column sums
// { sum g, sum g * xhat }.  Block (g, s): tiles [g * tpb, (g + 1) * tpb) x channels [256 s, 256 s + 256), 64 channel quads x 4
// tile lanes; its sums go to row g of `partials` ([G][2][K], G = ceil(N H W / 128) rows as the direct kernel fills them --
// the finaliser only adds the rows up), reduced over the tile lanes in fixed order.
struct WinoEpi {
    const float* add;
    const float* masky;
    const unsigned char* maskbits;
    const float* x;
    const float* relu_ab;
    const float* mean;
    const float* invstd;
    float* partials;
    int64_t add_ld, masky_ld, x_ld, maskbits_ld;
};

__global__ __launch_bounds__(256) void wino_output_epi_kernel(const float* __restrict__ Mb, const int4* __restrict__ tab,
                                                              float* __restrict__ y, int64_t ld, int64_t T, int64_t Tp, int K,
                                                              int H, int W, int d, int tpb, WinoEpi ep) {
    __shared__ float red[2][4][256];
    const int q = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int k = (blockIdx.y * 64 + q) * 4;
    const bool kok = k < K;
    const int64_t t0 = (int64_t)blockIdx.x * tpb;
    int64_t t1 = t0 + tpb;
    if (t1 > T) t1 = T;
    const int64_t plane = Tp * K;
    float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f};
    if (kok && ep.relu_ab != nullptr) {
        const float4 a0 = *reinterpret_cast<const float4*>(ep.relu_ab + k), a1 = *reinterpret_cast<const float4*>(ep.relu_ab + K + k);
        ra[0] = a0.x; ra[1] = a0.y; ra[2] = a0.z; ra[3] = a0.w;
        rb[0] = a1.x; rb[1] = a1.y; rb[2] = a1.z; rb[3] = a1.w;
    }
    if (kok && ep.partials != nullptr) {
        const float4 a0 = *reinterpret_cast<const float4*>(ep.mean + k), a1 = *reinterpret_cast<const float4*>(ep.invstd + k);
        mu[0] = a0.x; mu[1] = a0.y; mu[2] = a0.z; mu[3] = a0.w;
        is[0] = a1.x; is[1] = a1.y; is[2] = a1.z; is[3] = a1.w;
    }
    float sd[4] = {0.f, 0.f, 0.f, 0.f}, sd2[4] = {0.f, 0.f, 0.f, 0.f};
    if (kok) {
        for (int64_t t = t0 + tl; t < t1; t += 4) {
            const int4 e = tab[t];
            const float* src = Mb + t * K + k;
            float4 m[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = nt_load4(src + (4 * i + j) * plane);
            float4 s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
                s[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = e.y + i * d;
                if (yy >= H) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int xx = e.z + j * d;
                    if (xx >= W) continue;
                    const float4 o = j == 0 ? f4add(f4add(s[i][0], s[i][1]), s[i][2]) : f4sub(f4sub(s[i][1], s[i][2]), s[i][3]);
                    const int64_t row = (int64_t)(e.x * H + yy) * W + xx;
                    float v[4] = {o.x, o.y, o.z, o.w};
                    float xv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (ep.add != nullptr) {
                        const float4 a4 = *reinterpret_cast<const float4*>(ep.add + row * ep.add_ld + k);
                        v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
                    }
                    if (ep.x != nullptr) {
                        const float4 x4 = *reinterpret_cast<const float4*>(ep.x + row * ep.x_ld + k);
                        xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
                    }
                    if (ep.masky != nullptr) {
                        const float4 y4 = *reinterpret_cast<const float4*>(ep.masky + row * ep.masky_ld + k);
                        v[0] = y4.x > 0.f ? v[0] : 0.f; v[1] = y4.y > 0.f ? v[1] : 0.f;
                        v[2] = y4.z > 0.f ? v[2] : 0.f; v[3] = y4.w > 0.f ? v[3] : 0.f;
                    } else if (ep.maskbits != nullptr) {
                        const unsigned b = ep.maskbits[row * ep.maskbits_ld + (k >> 3)] >> (k & 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = ((b >> c) & 1u) ? v[c] : 0.f;
                    } else if (ep.relu_ab != nullptr) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = __builtin_fmaf(xv[c], ra[c], rb[c]) > 0.f ? v[c] : 0.f;
                    }
                    *reinterpret_cast<float4*>(y + row * ld + k) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sd[c] += v[c];
                        sd2[c] += v[c] * ((xv[c] - mu[c]) * is[c]);
                    }
                }
            }
        }
    }
    if (ep.partials == nullptr) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        red[0][tl][q * 4 + c] = sd[c];
        red[1][tl][q * 4 + c] = sd2[c];
    }
    __syncthreads();
    const int ch = blockIdx.y * 256 + threadIdx.x;
    if (ch < K) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            a0 += red[0][l][threadIdx.x];
            a1 += red[1][l][threadIdx.x];
        }
        float* sp = ep.partials + (int64_t)blockIdx.x * 2 * K + ch;
        sp[0] = a0;
        sp[K] = a1;
    }
}

// ---- backward-weight: dL/dg = G^T [ sum over tiles (A dY A^T) (.) (B^T d B) ] G  (the transpose of the forward chain).
// Z[k][t][co] = (A dY A^T)[i][j] of the 2x2 output-gradient tile t (zero outside the image / for padding tiles)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int64_t ld, const int4* __restrict__ tab,
                                                      float* __restrict__ Z, int64_t Tp, int K, int H, int W, int d) {
    const int k4n = K / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tp * k4n) return;
    const int64_t t = idx / k4n;
    const int k = (int)(idx - t * k4n) * 4;
    const int4 e = tab[t];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int yy = e.y + i * d, xx = e.z + j * d;
            const bool ok = e.x >= 0 && yy < H && xx < W;
            g[i][j] = ok ? *reinterpret_cast<const float4*>(dy + ((int64_t)(e.x * H + yy) * W + xx) * ld + k) : z;
        }
    // rows of A = [1 0; 1 1; 1 -1; 0 -1]
    float4 r[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        r[0][j] = g[0][j];
        r[1][j] = f4add(g[0][j], g[1][j]);
        r[2][j] = f4sub(g[0][j], g[1][j]);
        r[3][j] = f4sub(z, g[1][j]);
    }
    float* o = Z + t * K + k;
    const int64_t plane = Tp * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        nt_store4(o + (4 * i + 0) * plane, r[i][0]);
        nt_store4(o + (4 * i + 1) * plane, f4add(r[i][0], r[i][1]));
        nt_store4(o + (4 * i + 2) * plane, f4sub(r[i][0], r[i][1]));
        nt_store4(o + (4 * i + 3) * plane, f4sub(z, r[i][1]));
    }
}

// dw[co][r][s][c] = (G^T dU G)[r][s], dU [co][16][c]
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Cout, int Cin) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4n = Cin / 4;
    if (idx >= (int64_t)Cout * c4n) return;
    const int co = (int)(idx / c4n), c = (int)(idx - (int64_t)co * c4n) * 4;
    const float* src = dU + (int64_t)co * 16 * Cin + c;
    float4 u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[i][j] = *reinterpret_cast<const float4*>(src + (int64_t)(4 * i + j) * Cin);
    // G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
    float4 t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 hs = f4half(f4add(u[1][j], u[2][j])), hd = f4half(f4sub(u[1][j], u[2][j]));
        t[0][j] = f4add(u[0][j], hs);
        t[1][j] = hd;
        t[2][j] = f4add(hs, u[3][j]);
    }
    float* o = dw + (int64_t)co * 9 * Cin + c;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float4 hs = f4half(f4add(t[r][1], t[r][2])), hd = f4half(f4sub(t[r][1], t[r][2]));
        *reinterpret_cast<float4*>(o + (int64_t)(3 * r + 0) * Cin) = f4add(t[r][0], hs);
        *reinterpret_cast<float4*>(o + (int64_t)(3 * r + 1) * Cin) = hd;
        *reinterpret_cast<float4*>(o + (int64_t)(3 * r + 2) * Cin) = f4add(hs, t[r][3]);
    }
}

struct WinoWgradLayout {
    size_t tab, V, Z, dU, slab, total;
};
WinoWgradLayout wino_wgrad_layout(const WinoGeom& g, int64_t Cin, int64_t Cout, bool with_v) {
    WinoWgradLayout l;
    size_t o = 0;
    l.tab = o; o += (size_t)g.Tp * sizeof(int4);
    l.V = o; o += with_v ? (size_t)16 * g.Tp * Cin * sizeof(float) : 0;        // (not when the forward's V was kept)
    l.Z = o; o += (size_t)16 * g.Tp * Cout * sizeof(float);
    l.dU = o; o += (size_t)16 * Cout * Cin * sizeof(float);
    l.slab = o; o += wgrad_batched_slab_bytes(g.Tp, 16, Cout, Cin);
    l.total = o + 64;
    return l;
}

struct WinoLayout {
    size_t tab, U, V, M, total;
};
WinoLayout wino_layout(const WinoGeom& g, int64_t Cin, int64_t Cout) {
    WinoLayout l;
    size_t o = 0;
    l.tab = o; o += (size_t)g.Tp * sizeof(int4);
    l.U = o; o += (size_t)16 * Cout * Cin * sizeof(float);
    l.V = o; o += (size_t)16 * g.Tp * Cin * sizeof(float);
    l.M = o; o += (size_t)16 * g.Tp * Cout * sizeof(float);
    l.total = o + 64;
    return l;
}

}  // namespace wino
}  // namespace diga

using namespace diga;
using namespace diga::wino;

extern "C" size_t diga_conv2d_winograd_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation <= 0) return 0;
    return wino_layout(make_wino(N, H, W, dilation), Cin, Cout).total;
}

static int winograd_impl(const float* in, const float* wgt, const float* bias, float* out, void* workspace,
                         size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld,
                         int64_t Cout, int64_t out_ld, int64_t dilation, int flip, const diga_bwd_epilogue_t* epi, int prof_tag,
                         void* stream, float* v_keep = nullptr, const float* in_ab = nullptr) {
    DIGA_REQUIRE(in && wgt && out && workspace, DIGA_EINVAL, "conv2d_winograd: null pointer");
    DIGA_REQUIRE(N > 0 && H > 0 && W > 0 && dilation > 0 && dilation < 4096, DIGA_EINVAL, "conv2d_winograd: bad shape");
    DIGA_REQUIRE(Cin % 32 == 0 && Cout % 4 == 0 && Cout > 64 && in_ld >= Cin && in_ld % 4 == 0 && out_ld >= Cout && out_ld % 4 == 0,
                 DIGA_EINVAL, "conv2d_winograd: Cin %% 32, Cout %% 4 (> 64) and leading dimensions %% 4 required");
    DIGA_REQUIRE(aligned16(in) && aligned16(wgt) && aligned16(out) && aligned16(workspace) && (!bias || aligned16(bias)), DIGA_EALIGN,
                 "conv2d_winograd: pointers must be 16-byte aligned");
    DIGA_REQUIRE(N * H * W < (1ll << 31), DIGA_EINVAL, "conv2d_winograd: too many pixels");
    const WinoGeom g = make_wino(N, H, W, dilation);
    DIGA_REQUIRE(16 * g.Tp / 256 < 32768, DIGA_EINVAL, "conv2d_winograd: too many tiles for one launch");
    const WinoLayout l = wino_layout(g, Cin, Cout);
    DIGA_REQUIRE(workspace_bytes >= l.total, DIGA_EWORKSPACE, "conv2d_winograd: workspace too small (%zu < %zu)", workspace_bytes, l.total);
    char* ws = static_cast<char*>(workspace);
    int4* tab = reinterpret_cast<int4*>(ws + l.tab);
    float* U = reinterpret_cast<float*>(ws + l.U);
    float* V = v_keep != nullptr ? v_keep : reinterpret_cast<float*>(ws + l.V);
    float* Mb = reinterpret_cast<float*>(ws + l.M);
    hipStream_t st = (hipStream_t)stream;
    // priced as the direct convolution it replaces (the algorithmic FLOPs of the layer)
    ProfScope prof(prof_tag == DIGA_PROF_CONV_BWD_DATA ? DIGA_PROF_CONV_BWD_DATA : DIGA_PROF_CONV_FWD, st,
                   2.0 * (double)(N * H * W) * (double)Cout * 9.0 * (double)Cin);
    hipLaunchKernelGGL(wino_tiles_kernel, dim3((unsigned)ceil_div(g.Tp, 256)), dim3(256), 0, st, tab, g);
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)ceil_div(Cout * (Cin / 4), 256)), dim3(256), 0, st, wgt, U, (int)Cout,
                       (int)Cin, flip);
    DIGA_REQUIRE(!in_ab || aligned16(in_ab), DIGA_EALIGN, "conv2d_winograd: in_ab must be 16-byte aligned");
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)ceil_div(g.Tp * (Cin / 4), 256)), dim3(256), 0, st, in, in_ld, tab, V, g.Tp,
                       (int)Cin, (int)H, (int)W, (int)dilation, in_ab);
    int rc = gemm_batched_f32_dma(V, g.Tp, 16, Cin, U, Cout, Mb, st);
    if (rc) return rc;
    if (epi == nullptr) {
        hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)ceil_div(g.T * (Cout / 4), 256)), dim3(256), 0, st, Mb, tab, bias, out,
                           out_ld, g.T, g.Tp, (int)Cout, (int)H, (int)W, (int)dilation);
    } else {
        WinoEpi ep;
        ep.add = epi->addend; ep.add_ld = epi->addend_ld;
        ep.masky = epi->mask_y; ep.masky_ld = epi->mask_ld;
        ep.maskbits = epi->mask_bits; ep.maskbits_ld = epi->mask_bits_ld;
        ep.x = epi->x; ep.x_ld = epi->x_ld;
        ep.relu_ab = epi->relu_ab; ep.mean = epi->mean; ep.invstd = epi->invstd; ep.partials = epi->partials;
        const int64_t G = ceil_div(N * H * W, 128);
        const int tpb = (int)ceil_div(g.T, G);
        hipLaunchKernelGGL(wino_output_epi_kernel, dim3((unsigned)G, (unsigned)ceil_div(Cout, 256)), dim3(256), 0, st, Mb, tab, out,
                           out_ld, g.T, g.Tp, (int)Cout, (int)H, (int)W, (int)dilation, tpb, ep);
    }
    return launch_status("diga_conv2d_winograd_f32");
}

extern "C" int diga_conv2d_winograd_f32(const float* in, const float* wgt, const float* bias, float* out, void* workspace,
                                        size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld,
                                        int64_t Cout, int64_t out_ld, int64_t dilation, int flip, int prof_tag, void* stream) {
    return winograd_impl(in, wgt, bias, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, flip, nullptr,
                         prof_tag, stream);
}

extern "C" int diga_conv2d_winograd_f32_ab(const float* in, const float* in_ab, const float* wgt, const float* bias, float* out,
                                           float* v_keep, void* workspace, size_t workspace_bytes, int64_t N, int64_t H, int64_t W,
                                           int64_t Cin, int64_t in_ld, int64_t Cout, int64_t out_ld, int64_t dilation, int prof_tag,
                                           void* stream) {
    DIGA_REQUIRE(in_ab != nullptr, DIGA_EINVAL, "conv2d_winograd_ab: null coefficients");
    DIGA_REQUIRE(!v_keep || aligned16(v_keep), DIGA_EALIGN, "conv2d_winograd_ab: v_keep must be 16-byte aligned");
    return winograd_impl(in, wgt, bias, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, 0, nullptr,
                         prof_tag, stream, v_keep, in_ab);
}

extern "C" size_t diga_conv2d_winograd_v_floats(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t dilation) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || dilation <= 0) return 0;
    return (size_t)16 * make_wino(N, H, W, dilation).Tp * Cin;
}

extern "C" int diga_conv2d_winograd_f32_keep(const float* in, const float* wgt, const float* bias, float* out, float* v_keep,
                                             void* workspace, size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                             int64_t in_ld, int64_t Cout, int64_t out_ld, int64_t dilation, int prof_tag,
                                             void* stream) {
    DIGA_REQUIRE(v_keep != nullptr && aligned16(v_keep), DIGA_EINVAL, "conv2d_winograd_keep: v_keep must be a 16-byte aligned buffer");
    return winograd_impl(in, wgt, bias, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, 0, nullptr,
                         prof_tag, stream, v_keep);
}

extern "C" int diga_conv2d_winograd_f32_epi(const float* in, const float* wgt, float* out, void* workspace, size_t workspace_bytes,
                                            int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld, int64_t Cout,
                                            int64_t out_ld, int64_t dilation, int flip, const diga_bwd_epilogue_t* e, int prof_tag,
                                            void* stream) {
    DIGA_REQUIRE(e != nullptr, DIGA_EINVAL, "conv2d_winograd_epi: null epilogue descriptor");
    DIGA_REQUIRE(e->addend || e->mask_y || e->mask_bits || e->x, DIGA_EINVAL, "conv2d_winograd_epi: empty epilogue descriptor");
    DIGA_REQUIRE(!e->addend || (aligned16(e->addend) && e->addend_ld >= Cout && e->addend_ld % 4 == 0), DIGA_EINVAL, "conv2d_winograd_epi: bad addend");
    DIGA_REQUIRE(!e->mask_y || (aligned16(e->mask_y) && e->mask_ld >= Cout && e->mask_ld % 4 == 0), DIGA_EINVAL, "conv2d_winograd_epi: bad mask_y");
    DIGA_REQUIRE(!e->x || (aligned16(e->x) && e->x_ld >= Cout && e->x_ld % 4 == 0), DIGA_EINVAL, "conv2d_winograd_epi: bad x");
    DIGA_REQUIRE((e->mask_y != nullptr) + (e->relu_ab != nullptr) + (e->mask_bits != nullptr) <= 1, DIGA_EINVAL,
                 "conv2d_winograd_epi: give one of mask_y, mask_bits, relu_ab");
    DIGA_REQUIRE(!e->mask_bits || e->mask_bits_ld * 8 >= Cout, DIGA_EINVAL, "conv2d_winograd_epi: bad mask_bits");
    DIGA_REQUIRE(!e->relu_ab || (e->x && aligned16(e->relu_ab)), DIGA_EINVAL, "conv2d_winograd_epi: relu_ab needs x");
    DIGA_REQUIRE(!e->partials || (e->x && e->mean && e->invstd && aligned16(e->mean) && aligned16(e->invstd)), DIGA_EINVAL,
                 "conv2d_winograd_epi: partials need x, mean and invstd");
    return winograd_impl(in, wgt, nullptr, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, flip, e, prof_tag,
                         stream);
}

extern "C" size_t diga_conv2d_wgrad_winograd_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                                             int64_t dilation, int v_kept) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation <= 0 || Cout % 256 != 0 || Cin % 128 != 0) return 0;
    return wino_wgrad_layout(make_wino(N, H, W, dilation), Cin, Cout, v_kept == 0).total;
}

static int wgrad_winograd_impl(const float* dy, const float* x, const float* x_ab, const float* v_kept, float* dw, void* workspace,
                               size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t x_ld,
                               int64_t Cout, int64_t dy_ld, int64_t dilation, void* stream) {
    DIGA_REQUIRE(dy && (x || v_kept) && dw && workspace, DIGA_EINVAL, "conv2d_wgrad_winograd: null pointer");
    DIGA_REQUIRE(!v_kept || aligned16(v_kept), DIGA_EALIGN, "conv2d_wgrad_winograd: v_kept must be 16-byte aligned");
    DIGA_REQUIRE(N > 0 && H > 0 && W > 0 && dilation > 0 && dilation < 4096 && N * H * W < (1ll << 31), DIGA_EINVAL,
                 "conv2d_wgrad_winograd: bad shape");
    DIGA_REQUIRE(Cout % 256 == 0 && Cin % 128 == 0 && x_ld >= Cin && x_ld % 4 == 0 && dy_ld >= Cout && dy_ld % 4 == 0, DIGA_EINVAL,
                 "conv2d_wgrad_winograd: Cout %% 256, Cin %% 128 and leading dimensions %% 4 required");
    DIGA_REQUIRE(aligned16(dy) && (!x || aligned16(x)) && aligned16(dw) && aligned16(workspace), DIGA_EALIGN,
                 "conv2d_wgrad_winograd: pointers must be 16-byte aligned");
    const WinoGeom g = make_wino(N, H, W, dilation);
    const WinoWgradLayout l = wino_wgrad_layout(g, Cin, Cout, v_kept == nullptr);
    DIGA_REQUIRE(workspace_bytes >= l.total, DIGA_EWORKSPACE, "conv2d_wgrad_winograd: workspace too small (%zu < %zu)", workspace_bytes,
                 l.total);
    char* ws = static_cast<char*>(workspace);
    int4* tab = reinterpret_cast<int4*>(ws + l.tab);
    const float* V = v_kept != nullptr ? v_kept : reinterpret_cast<float*>(ws + l.V);
    float* Z = reinterpret_cast<float*>(ws + l.Z);
    float* dU = reinterpret_cast<float*>(ws + l.dU);
    float* slab = reinterpret_cast<float*>(ws + l.slab);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_CONV_BWD_WEIGHT, st, 2.0 * (double)(N * H * W) * (double)Cout * 9.0 * (double)Cin);
    hipLaunchKernelGGL(wino_tiles_kernel, dim3((unsigned)ceil_div(g.Tp, 256)), dim3(256), 0, st, tab, g);
    if (v_kept == nullptr)
        hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)ceil_div(g.Tp * (Cin / 4), 256)), dim3(256), 0, st, x, x_ld, tab,
                           reinterpret_cast<float*>(ws + l.V), g.Tp, (int)Cin, (int)H, (int)W, (int)dilation, x_ab);
    hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)ceil_div(g.Tp * (Cout / 4), 256)), dim3(256), 0, st, dy, dy_ld, tab, Z, g.Tp,
                       (int)Cout, (int)H, (int)W, (int)dilation);
    int rc = wgrad_batched_f32_dma(Z, V, dU, slab, g.Tp, 16, Cout, Cin, st);
    if (rc) return rc;
    hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)ceil_div(Cout * (Cin / 4), 256)), dim3(256), 0, st, dU, dw, (int)Cout, (int)Cin);
    return launch_status("diga_conv2d_wgrad_winograd_f32");
}

extern "C" int diga_conv2d_wgrad_winograd_f32(const float* dy, const float* x, const float* v_kept, float* dw, void* workspace,
                                              size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t x_ld,
                                              int64_t Cout, int64_t dy_ld, int64_t dilation, void* stream) {
    return wgrad_winograd_impl(dy, x, nullptr, v_kept, dw, workspace, workspace_bytes, N, H, W, Cin, x_ld, Cout, dy_ld, dilation, stream);
}

extern "C" int diga_conv2d_wgrad_winograd_f32_ab(const float* dy, const float* x, const float* x_ab, const float* v_kept, float* dw,
                                                 void* workspace, size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                                 int64_t x_ld, int64_t Cout, int64_t dy_ld, int64_t dilation, void* stream) {
    DIGA_REQUIRE(x_ab != nullptr && aligned16(x_ab), DIGA_EINVAL, "conv2d_wgrad_winograd_ab: null / unaligned coefficients");
    return wgrad_winograd_impl(dy, x, x_ab, v_kept, dw, workspace, workspace_bytes, N, H, W, Cin, x_ld, Cout, dy_ld, dilation, stream);
}
