// Multi-scale feature fusion of the SegFormer decode head: the sum of bilinearly resized maps at the finest scale.
// Reference: G5/model/networks/segformer_head.py:145-159 -- each of the four embedded stage outputs is resized to the 1/4-scale
// grid (`resize(..., mode='bilinear', align_corners=False)`, :472-488 -> F.interpolate), concatenated (4 x 768 channels) and
// reduced by the 1x1 `linear_fuse` conv.  The resize acts per channel and the fuse conv per pixel, so they commute: the host side
// (diga_amd/model/networks/segformer_head.py) applies the fuse weights at each map's OWN resolution and these kernels add the
// three coarse results into the finest one -- the 3072-channel concatenation (7.2 GB in fp32 for 16 crops of 768x768) is never
// formed, and the fuse GEMM runs on 1/4 + 1/16 + 1/64 + ... of the pixels.
//
// Both kernels are bandwidth passes over the fine grid ([N][H][W][C] fp32, channels contiguous, one float4 per thread):
// forward  : dst (+bias) += sum_k resize(src_k)            -- 8 B per fine element + the (cached) coarse reads
// backward : d src_k[j][i] = sum over the fine pixels whose two-tap stencils touch (j, i), weights recomputed exactly as the
//            forward computes them (so the pair is an exact adjoint in float arithmetic); a gather, no atomics, deterministic.
#include "common.h"

namespace diga {

// torch's align_corners=False source index (ATen UpSample.h area_pixel_compute_source_index, cubic=false): negative clamps to 0
__device__ __forceinline__ void half_pixel_taps(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

struct PyrSrc {
    const float* p;
    int h, w;
    float sh, sw;
};

struct PyrSrcs {
    PyrSrc s[3];
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int NS>
__global__ __launch_bounds__(256) void pyramid_sum_fwd_kernel(float* __restrict__ dst, const float* __restrict__ bias, PyrSrcs srcs,
                                                              int H, int W, int C4, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const int64_t pix = idx / C4;
    const int x = (int)(pix % W);
    const int64_t t = pix / W;
    const int y = (int)(t % H);
    const int64_t n = t / H;
    const int C = C4 * 4;
    float4 acc = ld4(dst + idx * 4);
    if (bias != nullptr) {
        const float4 b = ld4(bias + c4 * 4);
        acc.x += b.x, acc.y += b.y, acc.z += b.z, acc.w += b.w;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const PyrSrc s = srcs.s[k];
        int y0, y1, x0, x1;
        float ly, lx;
        half_pixel_taps(y, s.sh, s.h, y0, y1, ly);
        half_pixel_taps(x, s.sw, s.w, x0, x1, lx);
        const float* base = s.p + (n * s.h * (int64_t)s.w) * C + c4 * 4;
        const float4 v00 = ld4(base + ((int64_t)y0 * s.w + x0) * C), v01 = ld4(base + ((int64_t)y0 * s.w + x1) * C);
        const float4 v10 = ld4(base + ((int64_t)y1 * s.w + x0) * C), v11 = ld4(base + ((int64_t)y1 * s.w + x1) * C);
        const float hy0 = 1.f - ly, hx0 = 1.f - lx;
        // torch's order: h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
        acc.x += hy0 * (hx0 * v00.x + lx * v01.x) + ly * (hx0 * v10.x + lx * v11.x);
        acc.y += hy0 * (hx0 * v00.y + lx * v01.y) + ly * (hx0 * v10.y + lx * v11.y);
        acc.z += hy0 * (hx0 * v00.z + lx * v01.z) + ly * (hx0 * v10.z + lx * v11.z);
        acc.w += hy0 * (hx0 * v00.w + lx * v01.w) + ly * (hx0 * v10.w + lx * v11.w);
    }
    *reinterpret_cast<float4*>(dst + idx * 4) = acc;
}

// fine rows / columns that can touch coarse index j: src = scale * (d + 0.5) - 0.5 in (j - 1, j + 1); two extra on each side
// absorb the float rounding of the bound (rows outside the true footprint get weight 0 from the exact test below)
__device__ __forceinline__ void footprint(int j, float scale, int n_out, int& lo, int& hi) {
    const float inv = 1.f / scale;
    int a = (int)floorf(((float)j - 0.5f) * inv - 0.5f) - 2;
    int b = (int)ceilf(((float)j + 1.5f) * inv - 0.5f) + 2;
    lo = a < 0 ? 0 : a;
    hi = b > n_out - 1 ? n_out - 1 : b;
}

__device__ __forceinline__ float tap_weight(int d, float scale, int n_in, int j) {
    int i0, i1;
    float l1;
    half_pixel_taps(d, scale, n_in, i0, i1, l1);
    return (i0 == j ? 1.f - l1 : 0.f) + (i1 == j ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void pyramid_sum_bwd_kernel(const float* __restrict__ dout, float* __restrict__ ds, int H, int W, int h,
                                                              int w, float sh, float sw, int C4, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    const int64_t pix = idx / C4;
    const int i = (int)(pix % w);
    const int64_t t = pix / w;
    const int j = (int)(t % h);
    const int64_t n = t / h;
    const int C = C4 * 4;
    int ylo, yhi, xlo, xhi;
    footprint(j, sh, H, ylo, yhi);
    footprint(i, sw, W, xlo, xhi);
    const float* base = dout + (n * H * (int64_t)W) * C + c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y = ylo; y <= yhi; ++y) {
        const float wy = tap_weight(y, sh, h, j);
        if (wy == 0.f) continue;
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int x = xlo; x <= xhi; ++x) {
            const float wx = tap_weight(x, sw, w, i);
            if (wx == 0.f) continue;
            const float4 g = ld4(base + ((int64_t)y * W + x) * C);
            row.x += wx * g.x, row.y += wx * g.y, row.z += wx * g.z, row.w += wx * g.w;
        }
        acc.x += wy * row.x, acc.y += wy * row.y, acc.z += wy * row.z, acc.w += wy * row.w;
    }
    *reinterpret_cast<float4*>(ds + idx * 4) = acc;
}


// The three adjoints of the SegFormer geometry (ratios 2, 4, 8: stage maps of a crop whose side is a multiple of 64) in ONE pass over
// the fine gradient: a block stages a 16 x 16 fine tile + a halo of 4 pixels (the footprint of a coarse pixel of ratio r reaches r / 2
// beyond its cell) x 16 channels in LDS -- 2.25x the tile's bytes instead of the 3 x ~4 L2-level reads of the per-ratio gathers --
// and computes its 64 + 16 + 4 coarse pixels from there.  Every phase gives each of the 256 threads 16 taps: ratio 2: one thread per
// (coarse pixel, channel quad), 4 x 4 taps; ratio 4: four lanes per output (two of its 8 rows each); ratio 8: sixteen lanes per output
// (one of its 16 rows each); the lane partial sums are folded by shuffles in fixed order.  Weights: tap_weight, as above.
constexpr int kPyrT = 16, kPyrHalo = 4, kPyrCQ = 4, kPyrS = kPyrT + 2 * kPyrHalo;      // 24 x 24 staged pixels x 4 channel quads

template <int R, int LANES>
__device__ __forceinline__ void pyr_tile_adjoint(const float4* __restrict__ tile, float* __restrict__ ds, int h, int w, int H, int W, int C,
                                                 int64_t n, int ty0, int tx0, int cq0, int t) {
    // outputs of this ratio inside the tile: (kPyrT / R)^2 coarse pixels x kPyrCQ quads, LANES lanes each (LANES * outputs == 256)
    constexpr int PER = kPyrT / R, ROWS = 2 * R / LANES;          // coarse pixels per tile side; footprint rows per lane
    const int sub = t % LANES;
    const int item = t / LANES;
    const int q = item % kPyrCQ;
    const int cpx = item / kPyrCQ;
    const int ci = cpx % PER, cj = cpx / PER;
    const int gj = ty0 / R + cj, gi = tx0 / R + ci;               // coarse pixel (global)
    const float sh = 1.f / (float)R;
    const int fy0 = R * gj - R / 2, fx0 = R * gi - R / 2;         // first fine row / column of the footprint (global, may be < 0)
    float wx[2 * R];
#pragma unroll
    for (int dx = 0; dx < 2 * R; ++dx) {
        const int x = fx0 + dx;
        wx[dx] = (x >= 0 && x < W) ? tap_weight(x, sh, w, gi) : 0.f;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        const int dy = sub * ROWS + rr;
        const int y = fy0 + dy;
        const float wy = (y >= 0 && y < H) ? tap_weight(y, sh, h, gj) : 0.f;
        const float4* row = tile + ((y - (ty0 - kPyrHalo)) * kPyrS + (fx0 - (tx0 - kPyrHalo))) * kPyrCQ + q;
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dx = 0; dx < 2 * R; ++dx) {
            const float4 g = row[dx * kPyrCQ];
            r4.x += wx[dx] * g.x, r4.y += wx[dx] * g.y, r4.z += wx[dx] * g.z, r4.w += wx[dx] * g.w;
        }
        acc.x += wy * r4.x, acc.y += wy * r4.y, acc.z += wy * r4.z, acc.w += wy * r4.w;
    }
#pragma unroll
    for (int o = 1; o < LANES; o <<= 1) {                          // lanes of one output are consecutive: xor-fold, fixed order
        acc.x += __shfl_xor(acc.x, o, 64);
        acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64);
        acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (sub == 0) *reinterpret_cast<float4*>(ds + ((n * h + gj) * (int64_t)w + gi) * C + (cq0 + q) * 4) = acc;
}

__global__ __launch_bounds__(256) void pyramid_sum_bwd3_kernel(const float* __restrict__ dout, float* __restrict__ ds2, float* __restrict__ ds4,
                                                               float* __restrict__ ds8, int H, int W, int C) {
    __shared__ float4 tile[kPyrS * kPyrS * kPyrCQ];               // 36 KB
    const int t = threadIdx.x;
    const int tx0 = blockIdx.x * kPyrT, ty0 = blockIdx.y * kPyrT;
    const int cqn = C / (4 * kPyrCQ);
    const int64_t n = blockIdx.z / cqn;
    const int cq0 = (int)(blockIdx.z % cqn) * kPyrCQ;
    const float* img = dout + (n * H * (int64_t)W) * C + cq0 * 4;
    for (int i = t; i < kPyrS * kPyrS * kPyrCQ; i += 256) {
        const int q = i % kPyrCQ, px = (i / kPyrCQ) % kPyrS, py = i / (kPyrCQ * kPyrS);
        const int y = ty0 - kPyrHalo + py, x = tx0 - kPyrHalo + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) v = ld4(img + ((int64_t)y * W + x) * C + q * 4);
        tile[i] = v;
    }
    __syncthreads();
    pyr_tile_adjoint<2, 1>(tile, ds2, H / 2, W / 2, H, W, C, n, ty0, tx0, cq0, t);
    pyr_tile_adjoint<4, 4>(tile, ds4, H / 4, W / 4, H, W, C, n, ty0, tx0, cq0, t);
    pyr_tile_adjoint<8, 16>(tile, ds8, H / 8, W / 8, H, W, C, n, ty0, tx0, cq0, t);
}


// Forward of the same geometry, tiled: a block owns a 16 x 16 fine tile x 32 channels, stages the coarse pixels its taps can reach
// (10 x 10, 6 x 6, 4 x 4 of ratios 2, 4, 8) in LDS once -- the flat kernel above fetches 12 coarse vectors per fine element through
// L1 / L2: 21.7 GB of cache traffic for a 1.8 GB map -- and, having every output in registers anyway, leaves the BatchNorm that follows
// its statistics: per 64-pixel chunk (4 tile rows) and channel {sum (y - s), sum (y - s)^2, s = the chunk's first pixel}, the format
// of the persistent GEMM's epilogue (diga_bn_fwd_partials finalises it: no statistics pass over the fused map).
// Thread t: channel quad t % 8, pixel lane t / 8; pass k covers tile pixels 32 k + t / 8 (two tile rows): 128 contiguous bytes per pixel.
constexpr int kFwdCQ = 8;

__global__ __launch_bounds__(256) void pyramid_sum_fwd3_kernel(float* __restrict__ dst, const float* __restrict__ bias,
                                                               const float* __restrict__ s2, const float* __restrict__ s4,
                                                               const float* __restrict__ s8, float* __restrict__ stats, int H, int W, int C) {
    __shared__ float4 c2[10 * 10 * kFwdCQ], c4[6 * 6 * kFwdCQ], c8[4 * 4 * kFwdCQ];     // 19 KB
    __shared__ float4 shift[kFwdCQ];
    __shared__ float4 red[2][4][kFwdCQ];
    const int t = threadIdx.x;
    const int tx0 = blockIdx.x * 16, ty0 = blockIdx.y * 16;
    const int cqn = C / (4 * kFwdCQ);
    const int64_t n = blockIdx.z / cqn;
    const int cq0 = (int)(blockIdx.z % cqn) * kFwdCQ;
    auto stage = [&](float4* tile, const float* src, int r, int S) {
        const int h = H / r, w = W / r, oy = ty0 / r - 1, ox = tx0 / r - 1;
        const float* img = src + (n * h * (int64_t)w) * C + cq0 * 4;
        for (int i = t; i < S * S * kFwdCQ; i += 256) {
            const int q = i % kFwdCQ, px = (i / kFwdCQ) % S, py = i / (kFwdCQ * S);
            const int y = min(max(oy + py, 0), h - 1), x = min(max(ox + px, 0), w - 1);
            tile[i] = ld4(img + ((int64_t)y * w + x) * C + q * 4);
        }
    };
    stage(c2, s2, 2, 10);
    stage(c4, s4, 4, 6);
    stage(c8, s8, 8, 4);
    __syncthreads();
    const int q = t % kFwdCQ, pl = t / kFwdCQ;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) bv = ld4(bias + (cq0 + q) * 4);
    auto taps = [&](const float4* tile, int r, int S, int y, int x, float4& acc) {
        const float sc = 1.f / (float)r;
        int y0, y1, x0, x1;
        float ly, lx;
        half_pixel_taps(y, sc, H / r, y0, y1, ly);
        half_pixel_taps(x, sc, W / r, x0, x1, lx);
        const int oy = ty0 / r - 1, ox = tx0 / r - 1;
        const float4 v00 = tile[((y0 - oy) * S + (x0 - ox)) * kFwdCQ + q], v01 = tile[((y0 - oy) * S + (x1 - ox)) * kFwdCQ + q];
        const float4 v10 = tile[((y1 - oy) * S + (x0 - ox)) * kFwdCQ + q], v11 = tile[((y1 - oy) * S + (x1 - ox)) * kFwdCQ + q];
        const float hy0 = 1.f - ly, hx0 = 1.f - lx;
        acc.x += hy0 * (hx0 * v00.x + lx * v01.x) + ly * (hx0 * v10.x + lx * v11.x);
        acc.y += hy0 * (hx0 * v00.y + lx * v01.y) + ly * (hx0 * v10.y + lx * v11.y);
        acc.z += hy0 * (hx0 * v00.z + lx * v01.z) + ly * (hx0 * v10.z + lx * v11.z);
        acc.w += hy0 * (hx0 * v00.w + lx * v01.w) + ly * (hx0 * v10.w + lx * v11.w);
    };
    const int64_t chunk0 = (((int64_t)n * (H / 16) + blockIdx.y) * (W / 16) + blockIdx.x) * 4;       // this tile's four 64-pixel chunks
    for (int sub = 0; sub < 4; ++sub) {
        float4 yv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = 64 * sub + 32 * k + pl;                  // tile pixel
            const int y = ty0 + p / 16, x = tx0 + p % 16;
            float* o = dst + ((n * H + y) * (int64_t)W + x) * C + (cq0 + q) * 4;
            float4 acc = ld4(o);
            acc.x += bv.x, acc.y += bv.y, acc.z += bv.z, acc.w += bv.w;
            taps(c2, 2, 10, y, x, acc);                            // same taps and order as the flat kernel
            taps(c4, 4, 6, y, x, acc);
            taps(c8, 8, 4, y, x, acc);
            *reinterpret_cast<float4*>(o) = acc;
            yv[k] = acc;
        }
        if (stats != nullptr) {
            if (pl == 0) shift[q] = yv[0];                         // the chunk's first pixel
            __syncthreads();
            const float4 sh = shift[q];
            float4 sd = make_float4(0.f, 0.f, 0.f, 0.f), sd2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float dx = yv[k].x - sh.x, dy = yv[k].y - sh.y, dz = yv[k].z - sh.z, dw = yv[k].w - sh.w;
                sd.x += dx, sd.y += dy, sd.z += dz, sd.w += dw;
                sd2.x += dx * dx, sd2.y += dy * dy, sd2.z += dz * dz, sd2.w += dw * dw;
            }
            // the 8 pixel lanes of a wave that share a quad are lanes q, q + 8, ..., q + 56: fold bits 3..5, then the 4 waves through LDS
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                sd.x += __shfl_xor(sd.x, o, 64), sd.y += __shfl_xor(sd.y, o, 64), sd.z += __shfl_xor(sd.z, o, 64), sd.w += __shfl_xor(sd.w, o, 64);
                sd2.x += __shfl_xor(sd2.x, o, 64), sd2.y += __shfl_xor(sd2.y, o, 64), sd2.z += __shfl_xor(sd2.z, o, 64), sd2.w += __shfl_xor(sd2.w, o, 64);
            }
            const int wv = t >> 6;
            if ((t & 63) < kFwdCQ) {
                red[0][wv][q] = sd;
                red[1][wv][q] = sd2;
            }
            __syncthreads();
            if (t < kFwdCQ) {
                float4 a = red[0][0][t], b = red[1][0][t];
#pragma unroll
                for (int w4 = 1; w4 < 4; ++w4) {
                    const float4 a2 = red[0][w4][t], b2 = red[1][w4][t];
                    a.x += a2.x, a.y += a2.y, a.z += a2.z, a.w += a2.w;
                    b.x += b2.x, b.y += b2.y, b.z += b2.z, b.w += b2.w;
                }
                float* sp = stats + (chunk0 + sub) * 3 * (int64_t)C + (cq0 + t) * 4;
                *reinterpret_cast<float4*>(sp) = a;
                *reinterpret_cast<float4*>(sp + C) = b;
                *reinterpret_cast<float4*>(sp + 2 * (int64_t)C) = shift[t];
            }
            __syncthreads();                                       // shift / red are rewritten by the next chunk
        }
    }
}

}  // namespace diga

using namespace diga;

extern "C" int diga_pyramid_sum_fwd(float* dst, int64_t H, int64_t W, const float* bias, const float* s0, int64_t h0, int64_t w0,
                                    const float* s1, int64_t h1, int64_t w1, const float* s2, int64_t h2, int64_t w2, int64_t N,
                                    int64_t C, void* stream) {
    DIGA_REQUIRE(dst && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, DIGA_EINVAL, "pyramid_sum_fwd: bad argument (C % 4 == 0)");
    DIGA_REQUIRE(aligned16(dst) && (!bias || aligned16(bias)), DIGA_EINVAL, "pyramid_sum_fwd: pointers must be 16-byte aligned");
    const float* ps[3] = {s0, s1, s2};
    const int64_t hs[3] = {h0, h1, h2}, ws[3] = {w0, w1, w2};
    PyrSrcs srcs;
    int ns = 0;
    for (int k = 0; k < 3; ++k) {
        if (!ps[k]) continue;
        DIGA_REQUIRE(hs[k] > 0 && ws[k] > 0 && aligned16(ps[k]), DIGA_EINVAL, "pyramid_sum_fwd: bad source %d", k);
        srcs.s[ns].p = ps[k];
        srcs.s[ns].h = (int)hs[k];
        srcs.s[ns].w = (int)ws[k];
        srcs.s[ns].sh = (float)hs[k] / (float)H;      // torch: area_pixel_compute_scale without an explicit scale factor
        srcs.s[ns].sw = (float)ws[k] / (float)W;
        ++ns;
    }
    for (int k = ns; k < 3; ++k) srcs.s[k] = PyrSrc{nullptr, 1, 1, 1.f, 1.f};
    hipStream_t st = (hipStream_t)stream;
    double coarse = 0.0;
    for (int k = 0; k < ns; ++k) coarse += (double)N * srcs.s[k].h * srcs.s[k].w * C * 4.0;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)N * H * W * C * 8.0 + coarse);
    const int64_t total = N * H * W * (C / 4);
    const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
    switch (ns) {
        case 0: hipLaunchKernelGGL(pyramid_sum_fwd_kernel<0>, grid, block, 0, st, dst, bias, srcs, (int)H, (int)W, (int)(C / 4), total); break;
        case 1: hipLaunchKernelGGL(pyramid_sum_fwd_kernel<1>, grid, block, 0, st, dst, bias, srcs, (int)H, (int)W, (int)(C / 4), total); break;
        case 2: hipLaunchKernelGGL(pyramid_sum_fwd_kernel<2>, grid, block, 0, st, dst, bias, srcs, (int)H, (int)W, (int)(C / 4), total); break;
        default: hipLaunchKernelGGL(pyramid_sum_fwd_kernel<3>, grid, block, 0, st, dst, bias, srcs, (int)H, (int)W, (int)(C / 4), total); break;
    }
    return launch_status("diga_pyramid_sum_fwd");
}

extern "C" int diga_pyramid_sum_bwd(const float* dout, int64_t H, int64_t W, float* ds, int64_t h, int64_t w, int64_t N, int64_t C,
                                    void* stream) {
    DIGA_REQUIRE(dout && ds && N > 0 && H > 0 && W > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, DIGA_EINVAL,
                 "pyramid_sum_bwd: bad argument (C % 4 == 0)");
    DIGA_REQUIRE(aligned16(dout) && aligned16(ds), DIGA_EINVAL, "pyramid_sum_bwd: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, ((double)N * H * W + (double)N * h * w) * C * 4.0);
    const int64_t total = N * h * w * (C / 4);
    hipLaunchKernelGGL(pyramid_sum_bwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, dout, ds, (int)H, (int)W, (int)h,
                       (int)w, (float)h / (float)H, (float)w / (float)W, (int)(C / 4), total);
    return launch_status("diga_pyramid_sum_bwd");
}

extern "C" int diga_pyramid_sum_bwd3(const float* dout, int64_t H, int64_t W, float* ds2, float* ds4, float* ds8, int64_t N, int64_t C,
                                     void* stream) {
    DIGA_REQUIRE(dout && ds2 && ds4 && ds8 && N > 0 && H > 0 && W > 0 && C > 0, DIGA_EINVAL, "pyramid_sum_bwd3: bad argument");
    DIGA_REQUIRE(H % 16 == 0 && W % 16 == 0 && C % 16 == 0, DIGA_EINVAL, "pyramid_sum_bwd3: H %% 16, W %% 16, C %% 16 required (H=%lld W=%lld C=%lld)",
                 (long long)H, (long long)W, (long long)C);
    DIGA_REQUIRE(aligned16(dout) && aligned16(ds2) && aligned16(ds4) && aligned16(ds8), DIGA_EINVAL, "pyramid_sum_bwd3: pointers must be 16-byte aligned");
    DIGA_REQUIRE(N * (C / 16) < 65536, DIGA_EINVAL, "pyramid_sum_bwd3: too many (image, channel group) pairs for one launch");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)N * H * W * C * 4.0 * (1.0 + 1.0 / 4 + 1.0 / 16 + 1.0 / 64));
    hipLaunchKernelGGL(pyramid_sum_bwd3_kernel, dim3((unsigned)(W / 16), (unsigned)(H / 16), (unsigned)(N * (C / 16))), dim3(256), 0, st, dout,
                       ds2, ds4, ds8, (int)H, (int)W, (int)C);
    return launch_status("diga_pyramid_sum_bwd3");
}

extern "C" int diga_pyramid_sum_fwd3(float* dst, int64_t H, int64_t W, const float* bias, const float* s2, const float* s4, const float* s8,
                                     float* stats, int64_t N, int64_t C, void* stream) {
    DIGA_REQUIRE(dst && s2 && s4 && s8 && N > 0 && H > 0 && W > 0 && C > 0, DIGA_EINVAL, "pyramid_sum_fwd3: bad argument");
    DIGA_REQUIRE(H % 16 == 0 && W % 16 == 0 && C % 32 == 0, DIGA_EINVAL, "pyramid_sum_fwd3: H %% 16, W %% 16, C %% 32 required (H=%lld W=%lld C=%lld)",
                 (long long)H, (long long)W, (long long)C);
    DIGA_REQUIRE(aligned16(dst) && aligned16(s2) && aligned16(s4) && aligned16(s8) && (!bias || aligned16(bias)) && (!stats || aligned16(stats)),
                 DIGA_EINVAL, "pyramid_sum_fwd3: pointers must be 16-byte aligned");
    DIGA_REQUIRE(N * (C / 32) < 65536, DIGA_EINVAL, "pyramid_sum_fwd3: too many (image, channel group) pairs for one launch");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)N * H * W * C * 4.0 * (2.0 + 1.0 / 4 + 1.0 / 16 + 1.0 / 64));
    hipLaunchKernelGGL(pyramid_sum_fwd3_kernel, dim3((unsigned)(W / 16), (unsigned)(H / 16), (unsigned)(N * (C / 32))), dim3(256), 0, st, dst,
                       bias, s2, s4, s8, stats, (int)H, (int)W, (int)C);
    return launch_status("diga_pyramid_sum_fwd3");
}
