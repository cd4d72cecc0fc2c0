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
