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
    int m;                 // output tile edge: 2 = F(2x2,3x3) (16 products), 4 = F(4x4,3x3) (36 products)
    int tys, txs;          // tile rows / columns summed over the d phases
    int64_t T, Tp;
};

int phase_tiles(int len, int d, int m) {
    int s = 0;
    for (int a = 0; a < d; ++a) {
        const int n = len > a ? (len - a + d - 1) / d : 0;
        s += (n + m - 1) / m;
    }
    return s;
}

WinoGeom make_wino(int64_t N, int64_t H, int64_t W, int64_t d, int64_t m) {
    WinoGeom g;
    g.N = (int)N; g.H = (int)H; g.W = (int)W; g.d = (int)d; g.m = (int)m;
    g.tys = phase_tiles((int)H, (int)d, (int)m);
    g.txs = phase_tiles((int)W, (int)d, (int)m);
    g.T = N * g.tys * g.txs;
    g.Tp = ceil_div(g.T, 256) * 256;
    return g;
}
static inline int products(int64_t m) { return (int)((m + 2) * (m + 2)); }
static inline bool tile_ok(int64_t m) { return m == 2 || m == 4 || m == 6; }

// tab[t] = {image, oy, ox, 0}: top-left OUTPUT pixel of tile t (its m x m outputs are (oy + d i, ox + d j), its (m+2) x (m+2)
// input patch (oy + d (i - 1), ox + d (j - 1))); image = -1 for the padding tiles
__global__ __launch_bounds__(256) void wino_tiles_kernel(int4* __restrict__ tab, WinoGeom g) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Tp) return;
    if (t >= g.T) {
        tab[t] = make_int4(-1, 0, 0, 0);
        return;
    }
    const int timg = g.tys * g.txs;
    const int n = (int)(t / timg), r = (int)(t - (int64_t)n * timg);
    // Order of an image's tiles (any order works: the table IS the order of the GEMM's rows and of every transform).  Row-major
    // (kBand = 1): horizontally adjacent tiles are neighbours in t and read the same 8 image rows as contiguous ~14 KB row pieces.
    // Round 5 tried bands of 4 tile rows, column-major inside a band (vertical neighbours adjacent in t, to keep the two shared patch
    // rows in L2): FETCH_SIZE unchanged (100 062 vs 100 283 KB raw per launch on l3.conv2) and the input transform SLOWER inside the
    // step, 177 vs 141 us per launch -- vertically adjacent patches touch separate DRAM pages.
    constexpr int kBand = 1;
    const int band = r / (kBand * g.txs), rem = r - band * kBand * g.txs;
    const int rows_in_band = min(kBand, g.tys - kBand * band);
    int Cc = rem / rows_in_band, R = kBand * band + (rem - Cc * rows_in_band);
    int a = 0, b = 0;
    for (; a < g.d; ++a) {
        const int cnt = g.H > a ? (g.H - a + g.d - 1) / g.d : 0;
        const int tl = (cnt + g.m - 1) / g.m;
        if (R < tl) break;
        R -= tl;
    }
    for (; b < g.d; ++b) {
        const int cnt = g.W > b ? (g.W - b + g.d - 1) / g.d : 0;
        const int tl = (cnt + g.m - 1) / g.m;
        if (Cc < tl) break;
        Cc -= tl;
    }
    tab[t] = make_int4(n, a + g.m * R * g.d, b + g.m * Cc * g.d, 0);
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
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int64_t ld, const int4* __restrict__ tab,
                                                         float* __restrict__ V, int64_t Tp, int C, int H, int W, int d) {
    const int c4n = C / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tp * c4n) return;
    const int64_t t = idx / c4n;
    const int c = (int)(idx - t * c4n) * 4;
    const int4 e = tab[t];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    // branch-free: every tap is loaded from a clamped (valid) address and zeroed afterwards when it lies outside the image, so the
    // 16 loads of a thread issue back to back (measured on the 4x4-tile twin of this kernel: 151 -> 109 us)
    const int img = max(e.x, 0);
    float4 p[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = e.y + (i - 1) * d;
        const bool yok = e.x >= 0 && (unsigned)y < (unsigned)H;
        const int yc = min(max(y, 0), H - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = e.z + (j - 1) * d;
            const bool ok = yok && (unsigned)xx < (unsigned)W;
            const float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)(img * H + yc) * W + min(max(xx, 0), W - 1)) * ld + c);
            p[i][j] = ok ? v : z;
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
            const float4 v = *reinterpret_cast<const float4*>(dy + ((int64_t)(max(e.x, 0) * H + min(yy, H - 1)) * W + min(xx, W - 1)) * ld + k);
            g[i][j] = ok ? v : z;
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

// ======================================================================================================================
// Larger tiles: F(m x m, 3x3) with m = 4 (36 products per 16 outputs: 2.25 multiplications per output instead of 9 direct / 4 with
// F(2x2); V is 2.25x the input, M 2.25x the output) and m = 6 (64 products per 36 outputs: 1.78 per output; V and M 1.78x).  The
// 1-D transforms are generated (tools/gen_winograd_xforms.py -> winograd_xforms.h) from the exact Toom-Cook matrices:
//   F(4,3): points 0, 1, -1, 2, -1/2, inf -- fp32 error of a 512-channel layer against float64: max 4.9e-6 of the output scale
//           (Lavin & Gray's 0, +-1, +-2: 1.2e-5; F(2x2): 7.5e-7; the direct fmaf chain 3e-7);
//   F(6,3): points 0, +-1, +-2, +-1/2, inf -- max 2.7e-5, rms 2.0e-6.
// The caller picks the tile per layer (diga_amd/model/conv.py::_wino_plan: fewest multiplications); tests hold every tile size
// to its own bound against float64 and the full-size captures to north_star's 1e-3.
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) {
    return make_float4(__builtin_fmaf(s, a.x, b.x), __builtin_fmaf(s, a.y, b.y), __builtin_fmaf(s, a.z, b.z), __builtin_fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
// the same helpers on channel pairs: the transforms hold (m+2)^2 vectors per thread, float2 halves the registers (twice the waves)
__device__ __forceinline__ float2 f4add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 f4sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 f4fma(float s, float2 a, float2 b) { return make_float2(__builtin_fmaf(s, a.x, b.x), __builtin_fmaf(s, a.y, b.y)); }
__device__ __forceinline__ float2 f4scale(float s, float2 a) { return make_float2(s * a.x, s * a.y); }
// ... and on single channels (the 6x6-tile output passes: 8 x 8 products per thread)
__device__ __forceinline__ float f4add(float a, float b) { return a + b; }
__device__ __forceinline__ float f4sub(float a, float b) { return a - b; }
__device__ __forceinline__ float f4fma(float s, float a, float b) { return __builtin_fmaf(s, a, b); }
__device__ __forceinline__ float f4scale(float s, float a) { return s * a; }
__device__ __forceinline__ void nt_store4(float* p, float v) { __builtin_nontemporal_store(v, p); }
using f32x2nt = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void nt_store4(float* p, float2 v) {
    __builtin_nontemporal_store((f32x2nt){v.x, v.y}, reinterpret_cast<f32x2nt*>(p));
}
template <typename V> __device__ __forceinline__ V nt_loadv(const float* p);
template <> __device__ __forceinline__ float4 nt_loadv<float4>(const float* p) { return nt_load4(p); }
template <> __device__ __forceinline__ float2 nt_loadv<float2>(const float* p) {
    const f32x2nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x2nt*>(p));
    return make_float2(v.x, v.y);
}
template <> __device__ __forceinline__ float nt_loadv<float>(const float* p) { return __builtin_nontemporal_load(p); }
template <typename V> __device__ __forceinline__ V vzero();
template <> __device__ __forceinline__ float vzero<float>() { return 0.f; }
template <> __device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float2 vzero<float2>() { return make_float2(0.f, 0.f); }

#include "winograd_xforms.h"      // (inside namespace diga::wino: the generated transforms use the helpers above)

template <int M> struct Xf;
template <> struct Xf<4> {
    template <typename V> static __device__ __forceinline__ void bt(const V* x, V* r) { wino4_bt(x, r); }
    template <typename V> static __device__ __forceinline__ void g(const V* x, V* r) { wino4_g(x, r); }
    template <typename V> static __device__ __forceinline__ void at(const V* x, V* r) { wino4_at(x, r); }
    template <typename V> static __device__ __forceinline__ void a(const V* x, V* r) { wino4_a(x, r); }
    template <typename V> static __device__ __forceinline__ void gt(const V* x, V* r) { wino4_gt(x, r); }
};
template <> struct Xf<6> {
    template <typename V> static __device__ __forceinline__ void bt(const V* x, V* r) { wino6_bt(x, r); }
    template <typename V> static __device__ __forceinline__ void g(const V* x, V* r) { wino6_g(x, r); }
    template <typename V> static __device__ __forceinline__ void at(const V* x, V* r) { wino6_at(x, r); }
    template <typename V> static __device__ __forceinline__ void a(const V* x, V* r) { wino6_a(x, r); }
    template <typename V> static __device__ __forceinline__ void gt(const V* x, V* r) { wino6_gt(x, r); }
};
// vector width of each pass (measured on l3.conv2, 16 images, 4x4 tiles: output 82 us with float2 vs 94 us with float4 -- 113
// instead of 215 VGPRs; input / dy: no difference; the 8x8-patch passes of 6x6 tiles hold 64 vectors per thread: pairs throughout)
template <int M> struct Vec;
template <> struct Vec<4> { using In = float4; using Out = float2; using Dy = float4; };
// round 5 re-measured the 6x6 input transform on float4: 108 vs 163 us per launch in a warm micro-benchmark (tools/bench_conv.py), but
// INSIDE the step (cold operands, serialised: tools/diag/r05_regress.sh) 195 vs 177 us per launch, and the two-stream step 427.7 vs
// 424.7 ms (three interleaved runs each on one box) -- the float4 form needs the whole register file (one wave per SIMD) and
// co-resides with nothing.  Pairs stay; dy on pairs as well (84 vs 88 us).
template <> struct Vec<6> { using In = float2; using Out = float; using Dy = float2; };

// U[k = A i + j][co][c] = (G g G^T)[i][j], A = M + 2
template <int M>
__global__ __launch_bounds__(256) void winoM_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin,
                                                           int flip) {
    constexpr int A = M + 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4n = Cin / 4;
    if (idx >= (int64_t)Cout * c4n) return;
    const int co = (int)(idx / c4n), c = (int)(idx - (int64_t)co * c4n) * 4;
    float4 t[A][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float4 g[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int rr = flip ? 2 - r : r, ss = flip ? 2 - s : s;
            g[r] = *reinterpret_cast<const float4*>(w + ((int64_t)co * 9 + rr * 3 + ss) * Cin + c);
        }
        float4 col[A];
        Xf<M>::g(g, col);
#pragma unroll
        for (int i = 0; i < A; ++i) t[i][s] = col[i];
    }
    const int64_t plane = (int64_t)Cout * Cin;
    float* o = U + (int64_t)co * Cin + c;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        float4 row[A];
        Xf<M>::g(t[i], row);
#pragma unroll
        for (int j = 0; j < A; ++j) *reinterpret_cast<float4*>(o + (A * i + j) * plane) = row[j];
    }
}

// V[k = A i + j][t][c] = (B^T d B)[i][j] of the A x A patch of tile t (zero outside the image / for padding tiles).
// (NOT `__launch_bounds__(256, 1)`: an explicit 1 makes the compiler plan for one wave per SIMD -- the float2 form went from 161 to
//  > 200 VGPRs and from 130 to 220 us per launch)
template <typename V> __device__ __forceinline__ void wino_in_store(float* p, V v) { nt_store4(p, v); }
template <int M, typename V, bool REFLECT = false>
__global__ __launch_bounds__(256) void winoM_input_kernel(const float* __restrict__ x, int64_t ld, const int4* __restrict__ tab,
                                                          float* __restrict__ Vo, int64_t Tp, int C, int H, int W, int d) {
    // REFLECT: taps outside the image read the MIRRORED pixel (nn.ReflectionPad2d(d) in front of the conv: the translator's
    // ResBlocks, G5/model/model_util.py:21-61) instead of zero; coordinates beyond the mirror's reach only feed discarded outputs.
    // A template parameter, not a kernel argument: as a run-time flag the two `if (reflect)` per tap cost the zero-padding form
    // its back-to-back loads (130 -> 220 us per launch on l3.conv2; found by the per-kernel diff against the round-4 tree,
    // tools/diag/r05_regress.sh)
    constexpr bool reflect = REFLECT;
    constexpr int A = M + 2;
    constexpr int VW = sizeof(V) / 4;
    const int c4n = C / VW;
    // Round 5 tried an XCD-aware block order here (XCD x walks a contiguous range of tiles: the (M + 2)^2 patches of neighbouring tiles
    // overlap, a pixel is read by 1.78 tiles at M = 6, and the round-robin block -> XCD dispatch sends those reads through eight
    // different L2s): FETCH_SIZE per launch 117 611 -> 100 062 KB raw on l3.conv2 (-15 %), but the kernel no faster (141 vs 132 us in
    // the serialised step) and the two-stream step 1.5 ms slower (425.6 vs 424.1 ms, three interleaved runs): not kept.
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tp * c4n) return;
    const int64_t t = idx / c4n;
    const int c = (int)(idx - t * c4n) * VW;
    const int4 e = tab[t];
    const V z = vzero<V>();
    const int img = max(e.x, 0);
    V m[A][A];
#pragma unroll
    for (int j = 0; j < A; ++j) {
        // branch-free: every tap is loaded from a clamped (valid) address and zeroed afterwards when it lies outside the image,
        // so the A * A loads of a thread issue back to back (measured on 4x4 tiles: 151 -> 109 us)
        int xx = e.z + (j - 1) * d;
        if (reflect) {
            xx = xx < 0 ? -xx : xx;
            xx = xx >= W ? 2 * (W - 1) - xx : xx;
        }
        const bool xok = e.x >= 0 && (reflect || (unsigned)xx < (unsigned)W);
        const int xc = min(max(xx, 0), W - 1);
        V p[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            int y = e.y + (i - 1) * d;
            if (reflect) {
                y = y < 0 ? -y : y;
                y = y >= H ? 2 * (H - 1) - y : y;
            }
            const bool ok = xok && (reflect || (unsigned)y < (unsigned)H);
            const int yc = min(max(y, 0), H - 1);
            const V v = *reinterpret_cast<const V*>(x + ((int64_t)(img * H + yc) * W + xc) * ld + c);
            p[i] = ok ? v : z;
        }
        V col[A];
        Xf<M>::bt(p, col);
#pragma unroll
        for (int i = 0; i < A; ++i) m[i][j] = col[i];
    }
    float* o = Vo + t * C + c;
    const int64_t plane = Tp * C;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        V row[A];
        Xf<M>::bt(m[i], row);
#pragma unroll
        for (int j = 0; j < A; ++j) wino_in_store<V>(o + (A * i + j) * plane, row[j]);
    }
}

// s[i][j] = (A^T Mt)[i][j] of tile t, channels k..: M x A values from the A * A product planes
template <int M, typename V>
__device__ __forceinline__ void winoM_load_rows(const float* __restrict__ src, int64_t plane, V (*s)[M + 2]) {
    constexpr int A = M + 2;
    // One running pointer per product row, advanced by a plane per column.  Callers that walk SEVERAL tiles in a loop hand in a pointer
    // the compiler cannot take apart (asm barrier at the call site): written as Mb + t * K + k + (A * i + j) * plane, the A * A
    // per-lane plane addresses are loop-invariant up to t * K, get hoisted out of the tile loop as 64 VGPR pairs and the kernel
    // lands at 240-256 VGPRs / one wave per SIMD (what round 4's winoM_output_epi_kernel suffered from: 2.0 TB/s).
    const float* p[A];
#pragma unroll
    for (int i = 0; i < A; ++i) p[i] = src + (int64_t)(A * i) * plane;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        V m[A];
#pragma unroll
        for (int i = 0; i < A; ++i) {
            m[i] = nt_loadv<V>(p[i]);
            p[i] += plane;
        }
        V col[M];
        Xf<M>::at(m, col);
#pragma unroll
        for (int i = 0; i < M; ++i) s[i][j] = col[i];
    }
}

// y[M x M of tile t][co] = A^T Mt A + bias
template <int M, typename V>
__global__ __launch_bounds__(256) void winoM_output_kernel(const float* __restrict__ Mb, const int4* __restrict__ tab,
                                                           const float* __restrict__ bias, float* __restrict__ y, int64_t ld,
                                                           int64_t T, int64_t Tp, int K, int H, int W, int d) {
    constexpr int A = M + 2;
    constexpr int VW = sizeof(V) / 4;
    const int k4n = K / VW;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * k4n) return;
    const int64_t t = idx / k4n;
    const int k = (int)(idx - t * k4n) * VW;
    const int4 e = tab[t];
    V s[M][A];
    winoM_load_rows<M, V>(Mb + t * K + k, Tp * K, s);
    const V b = bias != nullptr ? *reinterpret_cast<const V*>(bias + k) : vzero<V>();
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const int yy = e.y + i * d;
        if (yy >= H) continue;
        V o[M];
        Xf<M>::at(s[i], o);
        float* row = y + ((int64_t)(e.x * H + yy) * W) * ld + k;
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (e.z + j * d < W) *reinterpret_cast<V*>(row + (int64_t)(e.z + j * d) * ld) = f4add(o[j], b);
    }
}

// winoM_output_kernel that also leaves the BatchNorm behind the layer its column statistics (round 5: the Winograd layers used to take a
// separate statistics pass over y, 80 launches and ~9.5 GB per C2 step).  Block b = tiles [b * tpb, (b + 1) * tpb) x CG channel
// groups of VW channels; thread (tl, q) walks tiles t0 + tl, t0 + tl + TL, ... (TL = 256 / CG tile lanes; a wave never spans two
// lanes: CG >= 64) and keeps {sum (y - s), sum (y - s)^2, s = its first pixel, n = its pixels} -- record r = b * TL + tl of
// `stats` ([R][3][K]) and `counts` ([R]): the format of diga_bn_fwd_records (records of unequal, possibly zero, size).
template <int M, typename V>
// (register cap of the two looping output transforms: 4 blocks per CU = 128 VGPRs per wave)
__global__ __launch_bounds__(256, 4) void winoM_output_stats_kernel(const float* __restrict__ Mb, const int4* __restrict__ tab,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int64_t ld,
                                                                 int64_t T, int64_t Tp, int K, int H, int W, int d, int tpb, int CG,
                                                                 float* __restrict__ stats, float* __restrict__ counts) {
    constexpr int A = M + 2;
    constexpr int VW = sizeof(V) / 4;
    const int q = threadIdx.x % CG, tl = threadIdx.x / CG, TL = 256 / CG;
    const int k = (blockIdx.y * CG + q) * VW;
    if (k >= K) return;
    const int64_t t0 = (int64_t)blockIdx.x * tpb;
    int64_t t1 = t0 + tpb;
    if (t1 > T) t1 = T;
    float b[VW], s0[VW], sd[VW], sd2[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) b[c] = s0[c] = sd[c] = sd2[c] = 0.f;
    if (bias != nullptr) *reinterpret_cast<V*>(b) = *reinterpret_cast<const V*>(bias + k);
    float n = 0.f;
#pragma unroll 1
    for (int64_t t = t0 + tl; t < t1; t += TL) {
        const int4 e = tab[t];
        V s[M][A];
        const float* srcp = Mb + t * K + k;
        asm volatile("" : "+v"(srcp));
        winoM_load_rows<M, V>(srcp, Tp * K, s);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int yy = e.y + i * d;
            if (yy >= H) continue;
            V o[M];
            Xf<M>::at(s[i], o);
            float* row = y + ((int64_t)(e.x * H + yy) * W) * ld + k;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                if (e.z + j * d >= W) continue;
                float v[VW];
                *reinterpret_cast<V*>(v) = o[j];
#pragma unroll
                for (int c = 0; c < VW; ++c) v[c] += b[c];
                *reinterpret_cast<V*>(row + (int64_t)(e.z + j * d) * ld) = *reinterpret_cast<const V*>(v);
                if (n == 0.f) {
#pragma unroll
                    for (int c = 0; c < VW; ++c) s0[c] = v[c];
                }
                n += 1.f;
#pragma unroll
                for (int c = 0; c < VW; ++c) {
                    const float dv = v[c] - s0[c];
                    sd[c] += dv;
                    sd2[c] = __builtin_fmaf(dv, dv, sd2[c]);
                }
            }
        }
    }
    const int64_t r = (int64_t)blockIdx.x * TL + tl;
    float* o = stats + r * 3 * K + k;
    *reinterpret_cast<V*>(o) = *reinterpret_cast<const V*>(sd);
    *reinterpret_cast<V*>(o + K) = *reinterpret_cast<const V*>(sd2);
    *reinterpret_cast<V*>(o + 2 * K) = *reinterpret_cast<const V*>(s0);
    if (q == 0 && blockIdx.y == 0) counts[r] = n;
}

// Round 5: winoM_output_epi_kernel again, built for OCCUPANCY.  The kernel above it holds a tile's A * A products and the M * M pixels'
// epilogue operands (addend, x, mask: up to 3 * 36 values) in registers at once -- 256 VGPRs, one wave per SIMD, 2.0 TB/s on l3.conv2
// (626 MB in 316 us) where the plain output transform (95 VGPRs, five waves per SIMD) moves its bytes at 5.7 TB/s.  Here a thread
// column-transforms the products as they arrive (A * M values stay), then walks the tile's rows: the epilogue operands of ONE row are
// loaded, used and dropped, the next row's loads are in flight behind the current row's arithmetic only through the other resident
// waves (four to five per SIMD).  The operand combination is a template parameter (addend?, mask kind) -- no per-element pointer
// tests.  Same arithmetic per element, same partial-row layout (block (g, s) = tile group x channel slab, TL tile lanes reduced
// through LDS in lane order): bit-identical results.
template <int M, typename V, int TL, bool ADD, int MASK /* 0 none, 1 y > 0, 2 bits, 3 fma(x, a, b) > 0 */, bool SUMS>
__global__ __launch_bounds__(256, 4) void winoM_output_epi2_kernel(const float* __restrict__ Mb, const int4* __restrict__ tab,
                                                                float* __restrict__ y, int64_t ld, int64_t T, int64_t Tp, int K,
                                                                int H, int W, int d, int tpb, WinoEpi ep) {
    constexpr int A = M + 2;
    constexpr int VW = sizeof(V) / 4;
    constexpr int CG = 256 / TL;
    __shared__ float red[2][TL][CG * VW];
    const int q = threadIdx.x % CG, tl = threadIdx.x / CG;
    const int k = (blockIdx.y * CG + q) * VW;
    const bool kok = k < K;
    const int64_t t0 = (int64_t)blockIdx.x * tpb;
    int64_t t1 = t0 + tpb;
    if (t1 > T) t1 = T;
    const int64_t plane = Tp * K;
    float ra[VW], rb[VW], mu[VW], is[VW], sd[VW], sd2[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) ra[c] = rb[c] = mu[c] = is[c] = sd[c] = sd2[c] = 0.f;
    if (kok && MASK == 3) {
        *reinterpret_cast<V*>(ra) = *reinterpret_cast<const V*>(ep.relu_ab + k);
        *reinterpret_cast<V*>(rb) = *reinterpret_cast<const V*>(ep.relu_ab + K + k);
    }
    if (kok && SUMS) {
        *reinterpret_cast<V*>(mu) = *reinterpret_cast<const V*>(ep.mean + k);
        *reinterpret_cast<V*>(is) = *reinterpret_cast<const V*>(ep.invstd + k);
    }
    if (kok) {
#pragma unroll 1
        for (int64_t t = t0 + tl; t < t1; t += TL) {
            const int4 e = tab[t];
            V s[M][A];
            const float* srcp = Mb + t * K + k;
            asm volatile("" : "+v"(srcp));
            winoM_load_rows<M, V>(srcp, plane, s);
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const int yy = e.y + i * d;
                if (yy >= H) continue;
                const int64_t rowbase = (int64_t)(e.x * H + yy) * W + e.z;
                // this row's epilogue operands (clamped column: loads are unconditional, results of dead pixels unused)
                V va[M], vx[M], vy[M];
                unsigned vb[M];
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int64_t row = rowbase + min(j * d, W - 1 - e.z);
                    if (ADD) va[j] = *reinterpret_cast<const V*>(ep.add + row * ep.add_ld + k);
                    if (MASK == 3 || SUMS) vx[j] = *reinterpret_cast<const V*>(ep.x + row * ep.x_ld + k);
                    if (MASK == 1) vy[j] = *reinterpret_cast<const V*>(ep.masky + row * ep.masky_ld + k);
                    if (MASK == 2) vb[j] = (unsigned)ep.maskbits[row * ep.maskbits_ld + (k >> 3)] >> (k & 7);
                }
                V o[M];
                Xf<M>::at(s[i], o);
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    if (e.z + j * d >= W) continue;
                    float v[VW], xv[VW], a4[VW], y4[VW];
                    *reinterpret_cast<V*>(v) = o[j];
                    if (ADD) {
                        *reinterpret_cast<V*>(a4) = va[j];
#pragma unroll
                        for (int c = 0; c < VW; ++c) v[c] += a4[c];
                    }
                    if (MASK == 3 || SUMS) *reinterpret_cast<V*>(xv) = vx[j];
                    if (MASK == 1) {
                        *reinterpret_cast<V*>(y4) = vy[j];
#pragma unroll
                        for (int c = 0; c < VW; ++c) v[c] = y4[c] > 0.f ? v[c] : 0.f;
                    } else if (MASK == 2) {
#pragma unroll
                        for (int c = 0; c < VW; ++c) v[c] = ((vb[j] >> c) & 1u) ? v[c] : 0.f;
                    } else if (MASK == 3) {
#pragma unroll
                        for (int c = 0; c < VW; ++c) v[c] = __builtin_fmaf(xv[c], ra[c], rb[c]) > 0.f ? v[c] : 0.f;
                    }
                    *reinterpret_cast<V*>(y + (rowbase + j * d) * ld + k) = *reinterpret_cast<const V*>(v);
                    if (SUMS) {
#pragma unroll
                        for (int c = 0; c < VW; ++c) {
                            sd[c] += v[c];
                            sd2[c] += v[c] * ((xv[c] - mu[c]) * is[c]);
                        }
                    }
                }
            }
        }
    }
    if (!SUMS) return;
#pragma unroll
    for (int c = 0; c < VW; ++c) {
        red[0][tl][q * VW + c] = sd[c];
        red[1][tl][q * VW + c] = sd2[c];
    }
    __syncthreads();
    const int ch = blockIdx.y * CG * VW + threadIdx.x;
    if (threadIdx.x < CG * VW && ch < K) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int l = 0; l < TL; ++l) {
            a0 += red[0][l][threadIdx.x];
            a1 += red[1][l][threadIdx.x];
        }
        float* sp = ep.partials + (int64_t)blockIdx.x * 2 * K + ch;
        sp[0] = a0;
        sp[K] = a1;
    }
}

// Z[k = A i + j][t][co] = (A dY A^T)[i][j] of the M x M output-gradient tile t
template <int M, typename V>
__global__ __launch_bounds__(256) void winoM_dy_kernel(const float* __restrict__ dy, int64_t ld, const int4* __restrict__ tab,
                                                       float* __restrict__ Z, int64_t Tp, int K, int H, int W, int d) {
    constexpr int A = M + 2;
    constexpr int VW = sizeof(V) / 4;
    const int k4n = K / VW;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= Tp * k4n) return;
    const int64_t t = idx / k4n;
    const int k = (int)(idx - t * k4n) * VW;
    const int4 e = tab[t];
    const V z = vzero<V>();
    const int img = max(e.x, 0);
    V r[A][M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int xx = e.z + j * d, xc = min(xx, W - 1);
        V g[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int yy = e.y + i * d;
            const bool ok = e.x >= 0 && yy < H && xx < W;
            const V v = *reinterpret_cast<const V*>(dy + ((int64_t)(img * H + min(yy, H - 1)) * W + xc) * ld + k);     // (clamped address)
            g[i] = ok ? v : z;
        }
        V col[A];
        Xf<M>::a(g, col);
#pragma unroll
        for (int i = 0; i < A; ++i) r[i][j] = col[i];
    }
    float* o = Z + t * K + k;
    const int64_t plane = Tp * K;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        V row[A];
        Xf<M>::a(r[i], row);
#pragma unroll
        for (int j = 0; j < A; ++j) nt_store4(o + (A * i + j) * plane, row[j]);
    }
}

// dw[co][r][s][c] = (G^T dU G)[r][s], dU [co][A * A][c]
template <int M, typename V>
__global__ __launch_bounds__(256) void winoM_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Cout, int Cin) {
    constexpr int A = M + 2;
    constexpr int VW = sizeof(V) / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4n = Cin / VW;
    if (idx >= (int64_t)Cout * c4n) return;
    const int co = (int)(idx / c4n), c = (int)(idx - (int64_t)co * c4n) * VW;
    const float* src = dU + (int64_t)co * A * A * Cin + c;
    V t[3][A];
#pragma unroll
    for (int j = 0; j < A; ++j) {
        V u[A];
#pragma unroll
        for (int i = 0; i < A; ++i) u[i] = *reinterpret_cast<const V*>(src + (int64_t)(A * i + j) * Cin);
        V col[3];
        Xf<M>::gt(u, col);
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r][j] = col[r];
    }
    float* o = dw + (int64_t)co * 9 * Cin + c;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        V row[3];
        Xf<M>::gt(t[r], row);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<V*>(o + (int64_t)(3 * r + q) * Cin) = row[q];
    }
}

// ---- launches by tile size (2: the hand-written F(2x2) kernels above)
template <int M>
static void launch_input_m(const float* x, int64_t ld, const int4* tab, float* V, int64_t Tp, int64_t C, int64_t H, int64_t W, int64_t d,
                           hipStream_t st, int reflect) {
    using VT = typename Vec<M>::In;
    constexpr int VW = sizeof(VT) / 4;
    const dim3 grid((unsigned)ceil_div(Tp * (C / VW), 256));
    if (reflect)
        hipLaunchKernelGGL((winoM_input_kernel<M, VT, true>), grid, dim3(256), 0, st, x, ld, tab, V, Tp, (int)C, (int)H, (int)W, (int)d);
    else
        hipLaunchKernelGGL((winoM_input_kernel<M, VT, false>), grid, dim3(256), 0, st, x, ld, tab, V, Tp, (int)C, (int)H, (int)W, (int)d);
}
static void launch_input(int64_t tile, const float* x, int64_t ld, const int4* tab, float* V, int64_t Tp, int64_t C, int64_t H, int64_t W,
                         int64_t d, hipStream_t st, int reflect = 0) {
    if (tile == 6) launch_input_m<6>(x, ld, tab, V, Tp, C, H, W, d, st, reflect);
    else if (tile == 4) launch_input_m<4>(x, ld, tab, V, Tp, C, H, W, d, st, reflect);
    else
        hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)ceil_div(Tp * (C / 4), 256)), dim3(256), 0, st, x, ld, tab, V, Tp, (int)C,
                           (int)H, (int)W, (int)d);
}
template <int M>
static void launch_weight_m(const float* w, float* U, int64_t Cout, int64_t Cin, int flip, hipStream_t st) {
    hipLaunchKernelGGL(winoM_weight_kernel<M>, dim3((unsigned)ceil_div(Cout * (Cin / 4), 256)), dim3(256), 0, st, w, U, (int)Cout, (int)Cin, flip);
}
template <int M>
static void launch_output_m(const float* Mb, const int4* tab, const float* bias, float* out, int64_t out_ld, const WinoGeom& g, int64_t Cout,
                            hipStream_t st) {
    using VT = typename Vec<M>::Out;
    constexpr int VW = sizeof(VT) / 4;
    hipLaunchKernelGGL((winoM_output_kernel<M, VT>), dim3((unsigned)ceil_div(g.T * (Cout / VW), 256)), dim3(256), 0, st, Mb, tab, bias, out,
                       out_ld, g.T, g.Tp, (int)Cout, g.H, g.W, g.d);
}
// statistics records of the forward output transform: tiles per block and channel groups per block (see winoM_output_stats_kernel)
struct WinoStatsPlan {
    int tpb, CG, TL;
    int64_t blocks, records;
};
template <int M>
static WinoStatsPlan stats_plan_m(const WinoGeom& g, int64_t Cout) {
    using VT = typename Vec<M>::Out;
    constexpr int VW = sizeof(VT) / 4;
    WinoStatsPlan p;
    const int64_t groups = Cout / VW;
    p.CG = groups >= 256 ? 256 : groups >= 128 ? 128 : 64;
    p.TL = 256 / p.CG;
    p.tpb = 4 * p.TL;
    p.blocks = ceil_div(g.T, p.tpb);
    p.records = p.blocks * p.TL;
    return p;
}
static WinoStatsPlan stats_plan(const WinoGeom& g, int64_t Cout) {
    return g.m == 6 ? stats_plan_m<6>(g, Cout) : stats_plan_m<4>(g, Cout);
}
template <int M>
static void launch_output_stats_m(const float* Mb, const int4* tab, const float* bias, float* out, int64_t out_ld, const WinoGeom& g,
                                  int64_t Cout, float* stats, hipStream_t st) {
    using VT = typename Vec<M>::Out;
    constexpr int VW = sizeof(VT) / 4;
    const WinoStatsPlan p = stats_plan_m<M>(g, Cout);
    float* counts = stats + p.records * 3 * Cout;
    hipLaunchKernelGGL((winoM_output_stats_kernel<M, VT>), dim3((unsigned)p.blocks, (unsigned)ceil_div(Cout, (int64_t)p.CG * VW)), dim3(256), 0, st,
                       Mb, tab, bias, out, out_ld, g.T, g.Tp, (int)Cout, g.H, g.W, g.d, p.tpb, p.CG, stats, counts);
}

constexpr int kEpiTileLanes = 2;    // tile lanes of the backward-data output transform (round 4: 11.5 / 10.1 / 9.8 ms per step with 4 / 2 / 1)
template <int M, bool ADD, int MASK>
static void launch_output_epi2_sums(const float* Mb, const int4* tab, float* out, int64_t out_ld, const WinoGeom& g, int64_t Cout, int64_t G,
                                    int tpb, const WinoEpi& ep, hipStream_t st) {
    using VT = typename Vec<M>::Out;
    constexpr int VW = sizeof(VT) / 4;
    constexpr int TL = kEpiTileLanes;
    const dim3 grid((unsigned)G, (unsigned)ceil_div(Cout, (256 / TL) * VW));
    if (ep.partials != nullptr)
        hipLaunchKernelGGL((winoM_output_epi2_kernel<M, VT, TL, ADD, MASK, true>), grid, dim3(256), 0, st, Mb, tab, out, out_ld, g.T, g.Tp,
                           (int)Cout, g.H, g.W, g.d, tpb, ep);
    else
        hipLaunchKernelGGL((winoM_output_epi2_kernel<M, VT, TL, ADD, MASK, false>), grid, dim3(256), 0, st, Mb, tab, out, out_ld, g.T, g.Tp,
                           (int)Cout, g.H, g.W, g.d, tpb, ep);
}
template <int M, bool ADD>
static void launch_output_epi2_mask(const float* Mb, const int4* tab, float* out, int64_t out_ld, const WinoGeom& g, int64_t Cout, int64_t G,
                                    int tpb, const WinoEpi& ep, hipStream_t st) {
    if (ep.masky != nullptr) launch_output_epi2_sums<M, ADD, 1>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
    else if (ep.maskbits != nullptr) launch_output_epi2_sums<M, ADD, 2>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
    else if (ep.relu_ab != nullptr) launch_output_epi2_sums<M, ADD, 3>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
    else launch_output_epi2_sums<M, ADD, 0>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
}
template <int M>
static void launch_output_epi2_m(const float* Mb, const int4* tab, float* out, int64_t out_ld, const WinoGeom& g, int64_t Cout, int64_t G,
                                 int tpb, const WinoEpi& ep, hipStream_t st) {
    if (ep.add != nullptr) launch_output_epi2_mask<M, true>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
    else launch_output_epi2_mask<M, false>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
}

template <int M>
static void launch_dy_m(const float* dy, int64_t dy_ld, const int4* tab, float* Z, const WinoGeom& g, int64_t Cout, hipStream_t st) {
    using VT = typename Vec<M>::Dy;
    constexpr int VW = sizeof(VT) / 4;
    hipLaunchKernelGGL((winoM_dy_kernel<M, VT>), dim3((unsigned)ceil_div(g.Tp * (Cout / VW), 256)), dim3(256), 0, st, dy, dy_ld, tab, Z, g.Tp,
                       (int)Cout, g.H, g.W, g.d);
}
template <int M>
static void launch_dw_m(const float* dU, float* dw, int64_t Cout, int64_t Cin, hipStream_t st) {
    hipLaunchKernelGGL((winoM_dw_kernel<M, float2>), dim3((unsigned)ceil_div(Cout * (Cin / 2), 256)), dim3(256), 0, st, dU, dw, (int)Cout, (int)Cin);
}

struct WinoWgradLayout {
    size_t tab, V, Z, dU, slab, total;
};
WinoWgradLayout wino_wgrad_layout(const WinoGeom& g, int64_t Cin, int64_t Cout, bool with_v) {
    WinoWgradLayout l;
    size_t o = 0;
    l.tab = o; o += (size_t)g.Tp * sizeof(int4);
    l.V = o; o += with_v ? (size_t)products(g.m) * g.Tp * Cin * sizeof(float) : 0;        // (not when the forward's V was kept)
    l.Z = o; o += (size_t)products(g.m) * g.Tp * Cout * sizeof(float);
    l.dU = o; o += (size_t)products(g.m) * Cout * Cin * sizeof(float);
    l.slab = o; o += wgrad_batched_slab_bytes(g.Tp, products(g.m), Cout, Cin);
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
    l.U = o; o += (size_t)products(g.m) * Cout * Cin * sizeof(float);
    l.V = o; o += (size_t)products(g.m) * g.Tp * Cin * sizeof(float);
    l.M = o; o += (size_t)products(g.m) * g.Tp * Cout * sizeof(float);
    l.total = o + 64;
    return l;
}

}  // namespace wino
}  // namespace diga

using namespace diga;
using namespace diga::wino;

extern "C" size_t diga_conv2d_winograd_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t dilation, int64_t tile) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation <= 0 || !tile_ok(tile)) return 0;
    return wino_layout(make_wino(N, H, W, dilation, tile), Cin, Cout).total;
}

static int winograd_impl(const float* in, const float* wgt, const float* bias, float* out, void* workspace,
                         size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld,
                         int64_t Cout, int64_t out_ld, int64_t dilation, int64_t tile, int flip, const diga_bwd_epilogue_t* epi, int prof_tag,
                         void* stream, float* v_keep = nullptr, float* stats = nullptr,
                         const void* tile_table = nullptr, int reflect = 0) {
    DIGA_REQUIRE(!reflect || (tile != 2 && !flip && !epi && dilation < H && dilation < W), DIGA_EINVAL,
                 "conv2d_winograd: reflection padding comes with the forward of 4x4 / 6x6 tiles (pad < H, W)");
    DIGA_REQUIRE(in && wgt && out && workspace, DIGA_EINVAL, "conv2d_winograd: null pointer");
    DIGA_REQUIRE(N > 0 && H > 0 && W > 0 && dilation > 0 && dilation < 4096, DIGA_EINVAL, "conv2d_winograd: bad shape");
    DIGA_REQUIRE(tile_ok(tile), DIGA_EINVAL, "conv2d_winograd: tile must be 2, 4 or 6 (F(2x2,3x3) / F(4x4,3x3) / F(6x6,3x3))");
    DIGA_REQUIRE(Cin % 32 == 0 && Cout % 4 == 0 && Cout > 64 && in_ld >= Cin && in_ld % 4 == 0 && out_ld >= Cout && out_ld % 4 == 0,
                 DIGA_EINVAL, "conv2d_winograd: Cin %% 32, Cout %% 4 (> 64) and leading dimensions %% 4 required");
    DIGA_REQUIRE(aligned16(in) && aligned16(wgt) && aligned16(out) && aligned16(workspace) && (!bias || aligned16(bias)), DIGA_EALIGN,
                 "conv2d_winograd: pointers must be 16-byte aligned");
    DIGA_REQUIRE(N * H * W < (1ll << 31), DIGA_EINVAL, "conv2d_winograd: too many pixels");
    const WinoGeom g = make_wino(N, H, W, dilation, tile);
    const int P = products(tile);
    DIGA_REQUIRE(P * g.Tp / 256 < 32768, DIGA_EINVAL, "conv2d_winograd: too many tiles for one launch");
    const WinoLayout l = wino_layout(g, Cin, Cout);
    DIGA_REQUIRE(workspace_bytes >= l.total, DIGA_EWORKSPACE, "conv2d_winograd: workspace too small (%zu < %zu)", workspace_bytes, l.total);
    DIGA_REQUIRE(!stats || (epi == nullptr && tile != 2 && aligned16(stats)), DIGA_EINVAL,
                 "conv2d_winograd: statistics come with the forward output transform of 4x4 / 6x6 tiles (16-byte aligned buffer)");
    DIGA_REQUIRE(!tile_table || aligned16(tile_table), DIGA_EALIGN, "conv2d_winograd: tile_table must be 16-byte aligned");
    char* ws = static_cast<char*>(workspace);
    const int4* tab = tile_table != nullptr ? static_cast<const int4*>(tile_table) : reinterpret_cast<const int4*>(ws + l.tab);
    float* U = reinterpret_cast<float*>(ws + l.U);
    float* V = v_keep != nullptr ? v_keep : reinterpret_cast<float*>(ws + l.V);
    float* Mb = reinterpret_cast<float*>(ws + l.M);
    hipStream_t st = (hipStream_t)stream;
    // priced as the direct convolution it replaces (the algorithmic FLOPs of the layer)
    ProfScope prof(prof_tag == DIGA_PROF_CONV_BWD_DATA ? DIGA_PROF_CONV_BWD_DATA : DIGA_PROF_CONV_FWD, st,
                   2.0 * (double)(N * H * W) * (double)Cout * 9.0 * (double)Cin);
    // (the table depends on the geometry only: a caller that keeps one per geometry -- diga_conv2d_winograd_tile_table -- saves the launch)
    if (tile_table == nullptr)
        hipLaunchKernelGGL(wino_tiles_kernel, dim3((unsigned)ceil_div(g.Tp, 256)), dim3(256), 0, st, reinterpret_cast<int4*>(ws + l.tab), g);
    if (tile == 6) launch_weight_m<6>(wgt, U, Cout, Cin, flip, st);
    else if (tile == 4) launch_weight_m<4>(wgt, U, Cout, Cin, flip, st);
    else
        hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)ceil_div(Cout * (Cin / 4), 256)), dim3(256), 0, st, wgt, U, (int)Cout, (int)Cin,
                           flip);
    launch_input(tile, in, in_ld, tab, V, g.Tp, Cin, H, W, dilation, st, reflect);
    int rc = gemm_batched_f32_dma(V, g.Tp, P, Cin, U, Cout, Mb, st);
    if (rc) return rc;
    if (epi == nullptr && stats != nullptr) {
        if (tile == 6) launch_output_stats_m<6>(Mb, tab, bias, out, out_ld, g, Cout, stats, st);
        else launch_output_stats_m<4>(Mb, tab, bias, out, out_ld, g, Cout, stats, st);
    } else if (epi == nullptr) {
        if (tile == 6) launch_output_m<6>(Mb, tab, bias, out, out_ld, g, Cout, st);
        else if (tile == 4) launch_output_m<4>(Mb, tab, bias, out, out_ld, g, Cout, st);
        else
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
        if (tile == 6) launch_output_epi2_m<6>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
        else if (tile == 4) launch_output_epi2_m<4>(Mb, tab, out, out_ld, g, Cout, G, tpb, ep, st);
        else
            hipLaunchKernelGGL(wino_output_epi_kernel, dim3((unsigned)G, (unsigned)ceil_div(Cout, 256)), dim3(256), 0, st, Mb, tab, out,
                               out_ld, g.T, g.Tp, (int)Cout, (int)H, (int)W, (int)dilation, tpb, ep);
    }
    return launch_status("diga_conv2d_winograd_f32");
}

extern "C" size_t diga_conv2d_winograd_tile_table_bytes(int64_t N, int64_t H, int64_t W, int64_t dilation, int64_t tile) {
    if (N <= 0 || H <= 0 || W <= 0 || dilation <= 0 || !tile_ok(tile)) return 0;
    return (size_t)make_wino(N, H, W, dilation, tile).Tp * sizeof(int4);
}

extern "C" int diga_conv2d_winograd_tile_table(void* table, int64_t N, int64_t H, int64_t W, int64_t dilation, int64_t tile, void* stream) {
    DIGA_REQUIRE(table && aligned16(table), DIGA_EINVAL, "conv2d_winograd_tile_table: null / unaligned table");
    DIGA_REQUIRE(N > 0 && H > 0 && W > 0 && dilation > 0 && dilation < 4096 && tile_ok(tile) && N * H * W < (1ll << 31), DIGA_EINVAL,
                 "conv2d_winograd_tile_table: bad shape / tile");
    const WinoGeom g = make_wino(N, H, W, dilation, tile);
    hipLaunchKernelGGL(wino_tiles_kernel, dim3((unsigned)ceil_div(g.Tp, 256)), dim3(256), 0, (hipStream_t)stream, static_cast<int4*>(table), g);
    return launch_status("diga_conv2d_winograd_tile_table");
}

extern "C" size_t diga_conv2d_winograd_stats_records(int64_t N, int64_t H, int64_t W, int64_t Cout, int64_t dilation, int64_t tile) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || dilation <= 0 || (tile != 4 && tile != 6)) return 0;
    return (size_t)stats_plan(make_wino(N, H, W, dilation, tile), Cout).records;
}

extern "C" size_t diga_conv2d_winograd_stats_floats(int64_t N, int64_t H, int64_t W, int64_t Cout, int64_t dilation, int64_t tile) {
    const size_t r = diga_conv2d_winograd_stats_records(N, H, W, Cout, dilation, tile);
    return r * 3 * (size_t)Cout + r;
}

extern "C" int diga_conv2d_winograd_f32(const float* in, const float* wgt, const float* bias, float* out, void* workspace,
                                        size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld,
                                        int64_t Cout, int64_t out_ld, int64_t dilation, int64_t tile, int flip, float* stats_partial,
                                        const void* tile_table, int prof_tag, void* stream) {
    return winograd_impl(in, wgt, bias, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, tile, flip, nullptr,
                         prof_tag, stream, nullptr, stats_partial, tile_table);
}

extern "C" int diga_conv2d_winograd_f32_opts(const float* in, const float* wgt, const float* bias, float* out, void* workspace,
                                             size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld,
                                             int64_t Cout, int64_t out_ld, int64_t dilation, int64_t tile, const diga_conv_options_t* opts,
                                             const void* tile_table, int prof_tag, void* stream) {
    DIGA_REQUIRE(opts != nullptr, DIGA_EINVAL, "conv2d_winograd_opts: null options");
    DIGA_REQUIRE(opts->upsample_shift == 0 && opts->activation == 0, DIGA_EINVAL,
                 "conv2d_winograd_opts: only reflect_pad is folded on the Winograd path (upsampling / tanh: the direct `_opts` kernels)");
    return winograd_impl(in, wgt, bias, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, tile, 0, nullptr,
                         prof_tag, stream, nullptr, nullptr, tile_table, opts->reflect_pad ? 1 : 0);
}

extern "C" size_t diga_conv2d_winograd_v_floats(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t dilation, int64_t tile) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || dilation <= 0 || !tile_ok(tile)) return 0;
    return (size_t)products(tile) * make_wino(N, H, W, dilation, tile).Tp * Cin;
}

extern "C" int diga_conv2d_winograd_f32_keep(const float* in, const float* wgt, const float* bias, float* out, float* v_keep,
                                             void* workspace, size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin,
                                             int64_t in_ld, int64_t Cout, int64_t out_ld, int64_t dilation, int64_t tile,
                                             float* stats_partial, const void* tile_table, int prof_tag, void* stream) {
    DIGA_REQUIRE(v_keep != nullptr && aligned16(v_keep), DIGA_EINVAL, "conv2d_winograd_keep: v_keep must be a 16-byte aligned buffer");
    return winograd_impl(in, wgt, bias, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, tile, 0, nullptr,
                         prof_tag, stream, v_keep, stats_partial, tile_table);
}

extern "C" int diga_conv2d_winograd_f32_epi(const float* in, const float* wgt, float* out, void* workspace, size_t workspace_bytes,
                                            int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t in_ld, int64_t Cout,
                                            int64_t out_ld, int64_t dilation, int64_t tile, int flip, const diga_bwd_epilogue_t* e,
                                            const void* tile_table, int prof_tag, void* stream) {
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
    return winograd_impl(in, wgt, nullptr, out, workspace, workspace_bytes, N, H, W, Cin, in_ld, Cout, out_ld, dilation, tile, flip, e, prof_tag,
                         stream, nullptr, nullptr, tile_table);
}

extern "C" size_t diga_conv2d_wgrad_winograd_workspace_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                                             int64_t dilation, int64_t tile, int v_kept) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dilation <= 0 || Cout % 256 != 0 || Cin % 128 != 0 || !tile_ok(tile)) return 0;
    return wino_wgrad_layout(make_wino(N, H, W, dilation, tile), Cin, Cout, v_kept == 0).total;
}

static int wgrad_winograd_impl(const float* dy, const float* x, const float* v_kept, float* dw, void* workspace,
                               size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t x_ld,
                               int64_t Cout, int64_t dy_ld, int64_t dilation, int64_t tile, const void* tile_table, void* stream) {
    DIGA_REQUIRE(!tile_table || aligned16(tile_table), DIGA_EALIGN, "conv2d_wgrad_winograd: tile_table must be 16-byte aligned");
    DIGA_REQUIRE(dy && (x || v_kept) && dw && workspace, DIGA_EINVAL, "conv2d_wgrad_winograd: null pointer");
    DIGA_REQUIRE(!v_kept || aligned16(v_kept), DIGA_EALIGN, "conv2d_wgrad_winograd: v_kept must be 16-byte aligned");
    DIGA_REQUIRE(N > 0 && H > 0 && W > 0 && dilation > 0 && dilation < 4096 && N * H * W < (1ll << 31) && tile_ok(tile),
                 DIGA_EINVAL, "conv2d_wgrad_winograd: bad shape / tile");
    DIGA_REQUIRE(Cout % 256 == 0 && Cin % 128 == 0 && x_ld >= Cin && x_ld % 4 == 0 && dy_ld >= Cout && dy_ld % 4 == 0, DIGA_EINVAL,
                 "conv2d_wgrad_winograd: Cout %% 256, Cin %% 128 and leading dimensions %% 4 required");
    DIGA_REQUIRE(aligned16(dy) && (!x || aligned16(x)) && aligned16(dw) && aligned16(workspace), DIGA_EALIGN,
                 "conv2d_wgrad_winograd: pointers must be 16-byte aligned");
    const WinoGeom g = make_wino(N, H, W, dilation, tile);
    const WinoWgradLayout l = wino_wgrad_layout(g, Cin, Cout, v_kept == nullptr);
    DIGA_REQUIRE(workspace_bytes >= l.total, DIGA_EWORKSPACE, "conv2d_wgrad_winograd: workspace too small (%zu < %zu)", workspace_bytes,
                 l.total);
    char* ws = static_cast<char*>(workspace);
    const int4* tab = tile_table != nullptr ? static_cast<const int4*>(tile_table) : reinterpret_cast<const int4*>(ws + l.tab);
    const float* V = v_kept != nullptr ? v_kept : reinterpret_cast<float*>(ws + l.V);
    float* Z = reinterpret_cast<float*>(ws + l.Z);
    float* dU = reinterpret_cast<float*>(ws + l.dU);
    float* slab = reinterpret_cast<float*>(ws + l.slab);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_CONV_BWD_WEIGHT, st, 2.0 * (double)(N * H * W) * (double)Cout * 9.0 * (double)Cin);
    if (tile_table == nullptr)
        hipLaunchKernelGGL(wino_tiles_kernel, dim3((unsigned)ceil_div(g.Tp, 256)), dim3(256), 0, st, reinterpret_cast<int4*>(ws + l.tab), g);
    if (v_kept == nullptr)
        launch_input(tile, x, x_ld, tab, reinterpret_cast<float*>(ws + l.V), g.Tp, Cin, H, W, dilation, st);
    if (tile == 6) launch_dy_m<6>(dy, dy_ld, tab, Z, g, Cout, st);
    else if (tile == 4) launch_dy_m<4>(dy, dy_ld, tab, Z, g, Cout, st);
    else
        hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)ceil_div(g.Tp * (Cout / 4), 256)), dim3(256), 0, st, dy, dy_ld, tab, Z, g.Tp,
                           (int)Cout, (int)H, (int)W, (int)dilation);
    int rc = wgrad_batched_f32_dma(Z, V, dU, slab, g.Tp, products(tile), Cout, Cin, st);
    if (rc) return rc;
    if (tile == 6) launch_dw_m<6>(dU, dw, Cout, Cin, st);
    else if (tile == 4) launch_dw_m<4>(dU, dw, Cout, Cin, st);
    else
        hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)ceil_div(Cout * (Cin / 4), 256)), dim3(256), 0, st, dU, dw, (int)Cout, (int)Cin);
    return launch_status("diga_conv2d_wgrad_winograd_f32");
}

extern "C" int diga_conv2d_wgrad_winograd_f32(const float* dy, const float* x, const float* v_kept, float* dw, void* workspace,
                                              size_t workspace_bytes, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t x_ld,
                                              int64_t Cout, int64_t dy_ld, int64_t dilation, int64_t tile, const void* tile_table,
                                              void* stream) {
    return wgrad_winograd_impl(dy, x, v_kept, dw, workspace, workspace_bytes, N, H, W, Cin, x_ld, Cout, dy_ld, dilation, tile,
                               tile_table, stream);
}
