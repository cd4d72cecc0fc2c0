// Per-pixel losses of the DiGA step: cross_entropy2d, distillation_loss (full-res API
// boundary, HBM-bound) and the fused bilinear-upsample + CE + distillation forward/backward at
// the low-res boundary.  Reference: G5/util/loss.py:48-62,125-143 and
// G5/train_DiGA_gta2city_warm_up.py:173-176,267-282.
//
// Data layout: logits NCHW fp32, one thread owns VEC consecutive pixels of one image and walks
// the C class planes (each plane access is a coalesced 16-B-per-lane stream).  Loss sums are
// reduced wave -> block -> one partial per block, then summed in double by one finishing block:
// no float atomics, results are bit-reproducible run to run.
#include "common.h"

namespace diga {

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    using type = float;
};
template <>
struct VecT<2> {
    using type = float2;
};
template <>
struct VecT<4> {
    using type = float4;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
    using T = typename VecT<VEC>::type;
    T t = *reinterpret_cast<const T*>(p);
    const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = f[i];
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
    using T = typename VecT<VEC>::type;
    T t;
    float* f = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = v[i];
    *reinterpret_cast<T*>(p) = t;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Stage 1 of a two-level sum: block b reduces its slice of `partials` (value k < 2 of entry i at partials[i * stride + k * koff])
// to stage[b*2 + k]; finished by finalize_sums_kernel over the gridDim.x stage entries.
__global__ __launch_bounds__(256) void stage_sums_kernel(const float* __restrict__ partials, int64_t n, int stride, int64_t koff,
                                                         int kcount, float* __restrict__ stage) {
    __shared__ float sm[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < n ? lo + per : n;
    float a0 = 0.f, a1 = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        a0 += partials[i * stride];
        if (kcount > 1) a1 += partials[i * stride + koff];
    }
    const float t0 = block_sum<4>(a0, sm);
    const float t1 = block_sum<4>(a1, sm);
    if (threadIdx.x == 0) {
        stage[blockIdx.x * 2] = t0;
        stage[blockIdx.x * 2 + 1] = t1;
    }
}

// out[k] = scale_k * sum_i partials[i*stride + k], k < 2.  One block of 1024 threads, double accumulate.
__global__ __launch_bounds__(1024) void finalize_sums_kernel(const float* __restrict__ partials, int64_t n,
                                                             int stride, int kcount, float* __restrict__ out,
                                                             double s0, double s1) {
    __shared__ double sm[16][2];
    double a0 = 0.0, a1 = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        a0 += (double)partials[i * stride];
        if (kcount > 1) a1 += (double)partials[i * stride + 1];
    }
    a0 = wave_sum_d(a0);
    a1 = wave_sum_d(a1);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
        sm[wid][0] = a0;
        sm[wid][1] = a1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (int i = 0; i < 16; ++i) {
            t0 += sm[i][0];
            t1 += sm[i][1];
        }
        out[0] = (float)(t0 * s0);
        if (kcount > 1) out[1] = (float)(t1 * s1);
    }
}

// ------------------------------------------------------------------------------------------
// cross_entropy2d: C compile-time (registers hold the C x VEC logits of the thread's pixels)
// ------------------------------------------------------------------------------------------
template <int C, int VEC>
__global__ __launch_bounds__(256) void ce2d_kernel(const float* __restrict__ logits,
                                                   const long long* __restrict__ target,
                                                   float* __restrict__ grad, float* __restrict__ partials,
                                                   int64_t HW, int64_t groups_per_img, int64_t total_groups,
                                                   float gscale) {
    __shared__ float sm[4];
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (g < total_groups) {
        const int64_t n = g / groups_per_img;
        const int64_t p = (g - n * groups_per_img) * VEC;
        const float* base = logits + (n * C) * HW + p;
        float x[C][VEC];
#pragma unroll
        for (int c = 0; c < C; ++c) load_vec<VEC>(base + (int64_t)c * HW, x[c]);
        long long t[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = target[n * HW + p + v];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float m = x[0][v];
#pragma unroll
            for (int c = 1; c < C; ++c) m = fmaxf(m, x[c][v]);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) s += expf(x[c][v] - m);
            const float lse = m + logf(s);
            const bool valid = t[v] != DIGA_IGNORE_LABEL;
            const int tc = (int)t[v];
            float xt = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) xt = (c == tc) ? x[c][v] : xt;
            loss += valid ? (lse - xt) : 0.f;
            const float gs = valid ? gscale : 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) x[c][v] = (expf(x[c][v] - lse) - ((c == tc) ? 1.f : 0.f)) * gs;
        }
        if (grad != nullptr) {
            float* gb = grad + (n * C) * HW + p;
#pragma unroll
            for (int c = 0; c < C; ++c) store_vec<VEC>(gb + (int64_t)c * HW, x[c]);
        }
    }
    const float tot = block_sum<4>(loss, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// Any C <= 32: three passes over the class planes (re-reads are served by L1/L2).
__global__ __launch_bounds__(256) void ce2d_generic_kernel(const float* __restrict__ logits,
                                                           const long long* __restrict__ target,
                                                           float* __restrict__ grad, float* __restrict__ partials,
                                                           int C, int64_t HW, int64_t total, float gscale) {
    __shared__ float sm[4];
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (g < total) {
        const int64_t n = g / HW, p = g - n * HW;
        const float* base = logits + (n * C) * HW + p;
        float m = base[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, base[(int64_t)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(base[(int64_t)c * HW] - m);
        const float lse = m + logf(s);
        const long long t = target[g];
        const bool valid = t != DIGA_IGNORE_LABEL;
        if (valid) loss = lse - base[(int64_t)t * HW];
        if (grad != nullptr) {
            float* gb = grad + (n * C) * HW + p;
            const float gs = valid ? gscale : 0.f;
            for (int c = 0; c < C; ++c)
                gb[(int64_t)c * HW] = (expf(base[(int64_t)c * HW] - lse) - ((c == (int)t) ? 1.f : 0.f)) * gs;
        }
    }
    const float tot = block_sum<4>(loss, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------
// distillation_loss
// ------------------------------------------------------------------------------------------
template <int C, int VEC>
__global__ __launch_bounds__(256) void distill_kernel(const float* __restrict__ teacher,
                                                      const float* __restrict__ student,
                                                      float* __restrict__ grad, float* __restrict__ partials,
                                                      int64_t B, int64_t HW, int64_t groups_per_img,
                                                      int64_t total_groups, float scale, float gscale) {
    __shared__ float sm[4];
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (g < total_groups) {
        const int64_t n = g / groups_per_img;
        const int64_t p = (g - n * groups_per_img) * VEC;
        const int64_t partner = n < B ? n + B : n - B;   // the other view's teacher
        const float wgt = n < B ? scale : 1.f;           // q1 -> s0 carries `scale`, q0 -> s1 weight 1
        const float* sb = student + (n * C) * HW + p;
        const float* tb = teacher + (partner * C) * HW + p;
        float s[C][VEC], t[C][VEC];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            load_vec<VEC>(sb + (int64_t)c * HW, s[c]);
            load_vec<VEC>(tb + (int64_t)c * HW, t[c]);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float ms = s[0][v], mt = t[0][v];
#pragma unroll
            for (int c = 1; c < C; ++c) {
                ms = fmaxf(ms, s[c][v]);
                mt = fmaxf(mt, t[c][v]);
            }
            float zs = 0.f, zt = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                zs += expf(s[c][v] - ms);
                t[c][v] = expf(t[c][v] - mt);
                zt += t[c][v];
            }
            const float lse = ms + logf(zs);
            const float rzt = 1.f / zt;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float q = t[c][v] * rzt;
                acc -= q * (s[c][v] - lse);
                s[c][v] = (expf(s[c][v] - lse) - q) * (wgt * gscale);
            }
            loss += wgt * acc;
        }
        if (grad != nullptr) {
            float* gb = grad + (n * C) * HW + p;
#pragma unroll
            for (int c = 0; c < C; ++c) store_vec<VEC>(gb + (int64_t)c * HW, s[c]);
        }
    }
    const float tot = block_sum<4>(loss, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void distill_generic_kernel(const float* __restrict__ teacher,
                                                              const float* __restrict__ student,
                                                              float* __restrict__ grad,
                                                              float* __restrict__ partials, int C, int64_t B,
                                                              int64_t HW, int64_t total, float scale,
                                                              float gscale) {
    __shared__ float sm[4];
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f;
    if (g < total) {
        const int64_t n = g / HW, p = g - n * HW;
        const int64_t partner = n < B ? n + B : n - B;
        const float wgt = n < B ? scale : 1.f;
        const float* sb = student + (n * C) * HW + p;
        const float* tb = teacher + (partner * C) * HW + p;
        float ms = sb[0], mt = tb[0];
        for (int c = 1; c < C; ++c) {
            ms = fmaxf(ms, sb[(int64_t)c * HW]);
            mt = fmaxf(mt, tb[(int64_t)c * HW]);
        }
        float zs = 0.f, zt = 0.f;
        for (int c = 0; c < C; ++c) {
            zs += expf(sb[(int64_t)c * HW] - ms);
            zt += expf(tb[(int64_t)c * HW] - mt);
        }
        const float lse = ms + logf(zs), rzt = 1.f / zt;
        float acc = 0.f;
        float* gb = grad ? grad + (n * C) * HW + p : nullptr;
        for (int c = 0; c < C; ++c) {
            const float q = expf(tb[(int64_t)c * HW] - mt) * rzt;
            const float sv = sb[(int64_t)c * HW];
            acc -= q * (sv - lse);
            if (gb) gb[(int64_t)c * HW] = (expf(sv - lse) - q) * (wgt * gscale);
        }
        loss = wgt * acc;
    }
    const float tot = block_sum<4>(loss, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------
// x *= *scale (no-op, no traffic, when the device scalar is exactly 1)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ x,
                                                            const float* __restrict__ scale_dev, int64_t n4,
                                                            int64_t n) {
    const float s = *scale_dev;
    if (s == 1.f) return;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4*>(x)[i];
        v.x *= s;
        v.y *= s;
        v.z *= s;
        v.w *= s;
        reinterpret_cast<float4*>(x)[i] = v;
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) x[i] *= s;
    }
}

// ------------------------------------------------------------------------------------------
// Fused upsample(align_corners) + CE + distillation, forward and backward at the low-res
// boundary.  A *cell* is the set of full-res pixels whose bilinear taps are the low-res
// corners (ci,cj),(ci,cj+1),(ci+1,cj),(ci+1,cj+1).  One wave owns one cell of one student
// image: lanes walk the cell's pixels (8x8 at output stride 8), interpolate student and
// partner-teacher logits from the 4 corners, evaluate both softmaxes, and accumulate the
// pixel gradient times the 4 corner weights.  The 4*C corner sums (+2 loss sums) are reduced
// across the wave through a transposed LDS image and stored per cell; a second kernel sums the
// <=4 cells around every low-res pixel in fixed order.  No atomics: deterministic.
// ------------------------------------------------------------------------------------------
__global__ void cell_starts_kernel(int* __restrict__ ystart, int* __restrict__ xstart, int h, int w, int H,
                                   int W, float sy, float sx) {
    // ystart[i] = first Y whose cell index >= i  (i in [0,h-1]; ystart[h-1] = H)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < h) {
        int lo = 0, hi = H;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int i0;
            float w1;
            bilinear_cell(mid, sy, h, i0, w1);
            if (i0 >= t)
                hi = mid;
            else
                lo = mid + 1;
        }
        ystart[t] = (t == h - 1) ? H : lo;
    }
    if (t < w) {
        int lo = 0, hi = W;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int i0;
            float w1;
            bilinear_cell(mid, sx, w, i0, w1);
            if (i0 >= t)
                hi = mid;
            else
                lo = mid + 1;
        }
        xstart[t] = (t == w - 1) ? W : lo;
    }
}

template <int C>
constexpr int cell_rows() {
    return 4 * C + 2;
}

// SINGLE (round 6): no cell of the launch holds more pixels than its lane group has lanes (64 / cpw), so there is ONE pixel per lane and
// no loop: the 4 C gradient sums need no accumulators -- each product goes straight into its cross-lane reduction.  UNI (cpw == 1, the
// x8 upsampling of the DeepLab path): the cell index is uniform over the wave, so the 4 x C x 2 corner logits are wave-uniform loads
// (scalar loads: the generic form issued 118 vector loads per lane for them).  The generic form's 76 accumulators + 57 logit /
// exponential values + 152 gathered corners made it a 315-VGPR kernel: ONE wave per SIMD under ~60 dependent transcendentals and a chain
// of global round trips per block (1.0 ms per C2 step; the round-6 counter pass showed the occupancy, not the arithmetic, to be the bound).
template <int C, bool DISTILL, bool SINGLE, bool UNI>
__global__ __launch_bounds__(64) void upsample_loss_cells_kernel(
    const float* __restrict__ stu_lr, const float* __restrict__ tea_lr, const long long* __restrict__ labels,
    const int* __restrict__ ystart, const int* __restrict__ xstart, float* __restrict__ cellpart, int B,
    int n_ce /* images that carry a CE term */, int h, int w, int H, int W, float sy, float sx, float k_ce,
    float k_di, float scale, int cpw_rt /* cells per wave: 1, 2, 4 or 8 neighbours along x, 64 / cpw lanes each */) {
    constexpr int ROWS = cell_rows<C>();
    constexpr bool ONE = SINGLE;            // (one pixel per lane, no accumulators)
    const int cpw = UNI ? 1 : cpw_rt;
    // (round 6) the wave's 4 C + 2 sums are folded over each lane quad with two DPP steps BEFORE they cross LDS: 17 columns instead of
    // 65 -- 5.3 KB instead of 20 KB per one-wave block, which had capped the kernel at 8 waves per CU (two per SIMD) under ~57
    // dependent transcendentals per pixel
    __shared__ float red[ROWS * 17];
    // small cells (logits at 1/4 scale: 4 x 4 full-resolution pixels per cell) would leave most of a wave idle and pay the LDS fold
    // per 16 pixels: `cpw` neighbouring cells share the wave, lane group `sub` owns cell blockIdx.x * cpw + sub
    const int lpc = 64 / cpw;
    const int lane = UNI ? (int)threadIdx.x : (int)threadIdx.x % lpc, sub = UNI ? 0 : (int)threadIdx.x / lpc;
    const int cj_raw = blockIdx.x * cpw + sub, ci = blockIdx.y, n = blockIdx.z;
    const bool cell_ok = cj_raw < w - 1;
    const int cj = cell_ok ? cj_raw : w - 2;
    const int ylo = ystart[ci], yhi = ystart[ci + 1], xlo = xstart[cj], xhi = xstart[cj + 1];
    const int nx = xhi - xlo, npx = cell_ok ? (yhi - ylo) * nx : 0;
    const int64_t plane = (int64_t)h * w;
    const float* S = stu_lr + ((int64_t)n * C) * plane + (int64_t)ci * w + cj;
    const int partner = DISTILL ? (n < B ? n + B : n - B) : 0;
    const float* T = DISTILL ? tea_lr + ((int64_t)partner * C) * plane + (int64_t)ci * w + cj : nullptr;
    const float wgt = n < B ? scale : 1.f;
    const bool has_ce = n < n_ce;

    float acc[ONE ? 1 : 4][ONE ? 1 : C];
    if constexpr (!ONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[k][c] = 0.f;
    }
    float ce_sum = 0.f, di_sum = 0.f;
    const int wl = threadIdx.x;
    auto quad_sum = [](float v) {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
        return v;
    };
    const bool lead = (wl & 3) == 0;
    const int qcol = wl >> 2;

    for (int base = 0; base < (ONE ? 1 : npx); base += lpc) {
        const int idx = base + lane;
        const bool live = idx < npx;
        const int py = ylo + (live ? idx / nx : 0), px = xlo + (live ? idx % nx : 0);
        int i0, j0;
        float wy, wx;
        bilinear_cell(py, sy, h, i0, wy);
        bilinear_cell(px, sx, w, j0, wx);
        const float w00 = (1.f - wy) * (1.f - wx), w01 = (1.f - wy) * wx, w10 = wy * (1.f - wx), w11 = wy * wx;
        float s[C], q[C];
        float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* sc = S + (int64_t)c * plane;
            // same association as torch's upsample_bilinear2d: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)
            s[c] = (1.f - wy) * ((1.f - wx) * sc[0] + wx * sc[1]) + wy * ((1.f - wx) * sc[w] + wx * sc[w + 1]);
            ms = fmaxf(ms, s[c]);
            if (DISTILL) {
                const float* tc = T + (int64_t)c * plane;
                q[c] = (1.f - wy) * ((1.f - wx) * tc[0] + wx * tc[1]) + wy * ((1.f - wx) * tc[w] + wx * tc[w + 1]);
                mt = fmaxf(mt, q[c]);
            }
        }
        float zs = 0.f, zt = 0.f;
        float es[C];                       // exp(s - max): softmax(s) = es / zs below (round 5 evaluated exp(s - lse) again: C more expf per pixel)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            es[c] = expf(s[c] - ms);
            zs += es[c];
            if (DISTILL) {
                q[c] = expf(q[c] - mt);
                zt += q[c];
            }
        }
        const float lse = ms + logf(zs);
        const float rzs = 1.f / zs;
        const float rzt = DISTILL ? 1.f / zt : 0.f;
        int tc = -1;
        bool valid = false;
        if (has_ce && live) {
            const long long t = labels[((int64_t)n * H + py) * W + px];
            valid = t != DIGA_IGNORE_LABEL;
            tc = (int)t;
        }
        const float gce = valid ? k_ce : 0.f;
        const float gdi = (DISTILL && live) ? k_di * wgt : 0.f;
        float di_px = 0.f, xt = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float p = es[c] * rzs;
            float g = gce * (p - ((c == tc) ? 1.f : 0.f));
            xt = (c == tc) ? s[c] : xt;
            if (DISTILL) {
                const float qq = q[c] * rzt;
                di_px -= qq * (s[c] - lse);
                g += gdi * (p - qq);
            }
            if constexpr (ONE) {
                // (dead lanes carry g = 0: gce and gdi are zero there)
                const float r0 = quad_sum(w00 * g), r1 = quad_sum(w01 * g), r2 = quad_sum(w10 * g), r3 = quad_sum(w11 * g);
                if (lead) {
                    red[(0 * C + c) * 17 + qcol] = r0;
                    red[(1 * C + c) * 17 + qcol] = r1;
                    red[(2 * C + c) * 17 + qcol] = r2;
                    red[(3 * C + c) * 17 + qcol] = r3;
                }
            } else {
                acc[0][c] += w00 * g;
                acc[1][c] += w01 * g;
                acc[2][c] += w10 * g;
                acc[3][c] += w11 * g;
            }
        }
        ce_sum += valid ? (lse - xt) : 0.f;
        if (DISTILL && live) di_sum += wgt * di_px;
    }
    // transposed wave reduction: quad sums by DPP (a lane group is >= 8 lanes: a quad never spans two cells), then through LDS:
    // row r = value index, column = quad (of the whole wave); each cell folds the quads of its lane group
    if constexpr (!ONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float v = quad_sum(acc[k][c]);
                if (lead) red[(k * C + c) * 17 + qcol] = v;
            }
    }
    {
        const float v0 = quad_sum(ce_sum), v1 = quad_sum(di_sum);
        if (lead) {
            red[(4 * C) * 17 + qcol] = v0;
            red[(4 * C + 1) * 17 + qcol] = v1;
        }
    }
    __syncthreads();
    // cellpart is PLANAR (round 6): value r of cell q at cellpart[r * ncells + q] -- the gather kernel's four reads per gradient element
    // and the loss sums then walk contiguous memory (the cell-major layout made every one of them a 312-byte-strided access: 80 us)
    const int64_t ncells = (int64_t)gridDim.z * (h - 1) * (w - 1);
    float* out = cellpart + ((int64_t)n * (h - 1) + ci) * (w - 1) + (int64_t)blockIdx.x * cpw;
    const int ncell = min(cpw, (w - 1) - (int)blockIdx.x * cpw);
    const int qpc = lpc >> 2;              // quads per cell
    for (int o = wl; o < ROWS * ncell; o += 64) {
        const int r = o / ncell, sc = o - r * ncell;
        float t = 0.f;
        for (int k = 0; k < qpc; ++k) t += red[r * 17 + sc * qpc + k];
        out[(int64_t)r * ncells + sc] = t;
    }
}

template <int C>
__global__ __launch_bounds__(256) void upsample_loss_gather_kernel(const float* __restrict__ cellpart,
                                                                   float* __restrict__ grad_lr, int N, int h,
                                                                   int w) {
    const int64_t total = (int64_t)N * C * h * w;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % w);
    const int i = (int)((t / w) % h);
    const int c = (int)((t / ((int64_t)w * h)) % C);
    const int n = (int)(t / ((int64_t)w * h * C));
    const int64_t ncells = (int64_t)N * (h - 1) * (w - 1);
    const float* base = cellpart + ((int64_t)n * (h - 1)) * (w - 1);       // planar: value r of cell q at [r * ncells + q]
    float g = 0.f;
    // fixed order: (i-1,j-1) corner 11, (i-1,j) corner 10, (i,j-1) corner 01, (i,j) corner 00
    if (i > 0 && j > 0) g += base[(int64_t)(3 * C + c) * ncells + ((int64_t)(i - 1)) * (w - 1) + (j - 1)];
    if (i > 0 && j < w - 1) g += base[(int64_t)(2 * C + c) * ncells + ((int64_t)(i - 1)) * (w - 1) + j];
    if (i < h - 1 && j > 0) g += base[(int64_t)(1 * C + c) * ncells + ((int64_t)i) * (w - 1) + (j - 1)];
    if (i < h - 1 && j < w - 1) g += base[(int64_t)c * ncells + ((int64_t)i) * (w - 1) + j];
    grad_lr[t] = g;
}

__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int64_t NC, int h, int w, int H, int W, float sy,
                                                                float sx) {
    const int64_t total = NC * H * W;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int X = (int)(t % W), Y = (int)((t / W) % H);
        const int64_t nc = t / ((int64_t)W * H);
        int i0, j0;
        float wy, wx;
        bilinear_cell(Y, sy, h, i0, wy);
        bilinear_cell(X, sx, w, j0, wx);
        const float* p = x + (nc * h + i0) * w + j0;
        const int dj = (w > 1) ? 1 : 0, di = (h > 1) ? w : 0;
        y[t] = (1.f - wy) * ((1.f - wx) * p[0] + wx * p[dj]) + wy * ((1.f - wx) * p[di] + wx * p[di + dj]);
    }
}

static size_t partial_slots(int64_t n_pixels) { return (size_t)ceil_div(n_pixels, 256) + 8; }

}  // namespace diga

using namespace diga;

extern "C" size_t diga_loss_workspace_bytes(int64_t n_pixels) { return partial_slots(n_pixels) * sizeof(float); }

extern "C" int diga_ce2d_fwd_bwd(const float* logits, const int64_t* target, float* grad, float* loss_out,
                                 void* workspace, size_t workspace_bytes, int64_t N, int64_t C, int64_t H,
                                 int64_t W, float grad_scale, void* stream) {
    DIGA_REQUIRE(logits && target && loss_out && workspace, DIGA_EINVAL, "ce2d: null pointer");
    DIGA_REQUIRE(N > 0 && H > 0 && W > 0 && C >= 1 && C <= 32, DIGA_EINVAL, "ce2d: bad shape N=%lld C=%lld H=%lld W=%lld",
                 (long long)N, (long long)C, (long long)H, (long long)W);
    const int64_t HW = H * W, total = N * HW;
    DIGA_REQUIRE(workspace_bytes >= diga_loss_workspace_bytes(total), DIGA_EWORKSPACE, "ce2d: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // read C*4 + 8 (int64 label) per pixel, write the C*4-byte gradient
    ProfScope prof(DIGA_PROF_CE2D, st, (double)total * (C * 4.0 + 8.0 + (grad ? C * 4.0 : 0.0)));
    float* partials = (float*)workspace;
    const float gs = grad_scale / (float)total;
    const bool vec4 = (HW % 4 == 0) && aligned16(logits) && (!grad || aligned16(grad));
    int64_t blocks;
    const long long* tgt = (const long long*)target;
    if ((C == 19 || C == 16) && vec4) {
        const int64_t gpi = HW / 4, groups = N * gpi;
        blocks = ceil_div(groups, 256);
        if (C == 19)
            hipLaunchKernelGGL((ce2d_kernel<19, 4>), dim3((unsigned)blocks), dim3(256), 0, st, logits, tgt, grad, partials, HW, gpi, groups, gs);
        else
            hipLaunchKernelGGL((ce2d_kernel<16, 4>), dim3((unsigned)blocks), dim3(256), 0, st, logits, tgt, grad, partials, HW, gpi, groups, gs);
    } else if (C == 19 || C == 16) {
        blocks = ceil_div(total, 256);
        if (C == 19)
            hipLaunchKernelGGL((ce2d_kernel<19, 1>), dim3((unsigned)blocks), dim3(256), 0, st, logits, tgt, grad, partials, HW, HW, total, gs);
        else
            hipLaunchKernelGGL((ce2d_kernel<16, 1>), dim3((unsigned)blocks), dim3(256), 0, st, logits, tgt, grad, partials, HW, HW, total, gs);
    } else {
        blocks = ceil_div(total, 256);
        hipLaunchKernelGGL(ce2d_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logits, tgt, grad, partials, (int)C, HW, total, gs);
    }
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(1024), 0, st, partials, blocks, 1, 1, loss_out,
                       1.0 / (double)total, 0.0);
    return launch_status("diga_ce2d_fwd_bwd");
}

extern "C" int diga_distill_fwd_bwd(const float* teacher, const float* student, float* grad, float* loss_out,
                                    void* workspace, size_t workspace_bytes, int64_t B2, int64_t C, int64_t H,
                                    int64_t W, float scale, float grad_scale, void* stream) {
    DIGA_REQUIRE(teacher && student && loss_out && workspace, DIGA_EINVAL, "distill: null pointer");
    DIGA_REQUIRE(B2 > 0 && (B2 % 2) == 0 && H > 0 && W > 0 && C >= 1 && C <= 32, DIGA_EINVAL,
                 "distill: bad shape 2B=%lld C=%lld H=%lld W=%lld (batch must hold two views)", (long long)B2,
                 (long long)C, (long long)H, (long long)W);
    const int64_t HW = H * W, total = B2 * HW, B = B2 / 2;
    DIGA_REQUIRE(workspace_bytes >= diga_loss_workspace_bytes(total), DIGA_EWORKSPACE, "distill: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // per student-image pixel: read teacher and student logits, write the gradient
    ProfScope prof(DIGA_PROF_DISTILL, st, (double)total * (C * 8.0 + (grad ? C * 4.0 : 0.0)));
    float* partials = (float*)workspace;
    const float gs = grad_scale / (float)(B * HW);
    const bool vec2 = (HW % 2 == 0) && aligned16(teacher) && aligned16(student) && (!grad || aligned16(grad));
    int64_t blocks;
    if ((C == 19 || C == 16) && vec2) {
        const int64_t gpi = HW / 2, groups = B2 * gpi;
        blocks = ceil_div(groups, 256);
        if (C == 19)
            hipLaunchKernelGGL((distill_kernel<19, 2>), dim3((unsigned)blocks), dim3(256), 0, st, teacher, student, grad, partials, B, HW, gpi, groups, scale, gs);
        else
            hipLaunchKernelGGL((distill_kernel<16, 2>), dim3((unsigned)blocks), dim3(256), 0, st, teacher, student, grad, partials, B, HW, gpi, groups, scale, gs);
    } else if (C == 19 || C == 16) {
        blocks = ceil_div(total, 256);
        if (C == 19)
            hipLaunchKernelGGL((distill_kernel<19, 1>), dim3((unsigned)blocks), dim3(256), 0, st, teacher, student, grad, partials, B, HW, HW, total, scale, gs);
        else
            hipLaunchKernelGGL((distill_kernel<16, 1>), dim3((unsigned)blocks), dim3(256), 0, st, teacher, student, grad, partials, B, HW, HW, total, scale, gs);
    } else {
        blocks = ceil_div(total, 256);
        hipLaunchKernelGGL(distill_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, st, teacher, student, grad, partials, (int)C, B, HW, total, scale, gs);
    }
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(1024), 0, st, partials, blocks, 1, 1, loss_out,
                       1.0 / (double)(B * HW), 0.0);
    return launch_status("diga_distill_fwd_bwd");
}

extern "C" int diga_scale_inplace(float* x, const float* scale_dev, int64_t n, void* stream) {
    DIGA_REQUIRE(x && scale_dev && n >= 0, DIGA_EINVAL, "scale_inplace: bad argument");
    DIGA_REQUIRE(aligned16(x), DIGA_EALIGN, "scale_inplace: x must be 16-byte aligned");
    if (n == 0) return DIGA_OK;
    const int64_t n4 = n / 4;
    const int64_t blocks = n4 > 0 ? (ceil_div(n4, 256) < 2048 ? ceil_div(n4, 256) : 2048) : 1;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, scale_dev, n4, n);
    return launch_status("diga_scale_inplace");
}

namespace {
constexpr int kStageBlocks = 128;
struct CellWs {
    int* ystart;
    int* xstart;
    float* stage;
    float* cellpart;
    size_t bytes;
};
CellWs carve(void* ws, int64_t N, int64_t C, int64_t h, int64_t w) {
    CellWs r;
    const size_t tab = (size_t)((h + w + 63) / 64) * 64 * sizeof(int);
    const size_t stg = (size_t)kStageBlocks * 2 * sizeof(float);
    r.ystart = (int*)ws;
    r.xstart = r.ystart + h;
    r.stage = (float*)((char*)ws + tab);
    r.cellpart = (float*)((char*)ws + tab + stg);
    r.bytes = tab + stg + (size_t)N * (h - 1) * (w - 1) * (4 * C + 2) * sizeof(float);
    return r;
}

template <int C>
int run_upsample_loss(const float* stu_lr, const float* tea_lr, const int64_t* labels, float* grad_lr,
                      float* losses_out, void* workspace, int64_t N, int64_t B, int64_t n_ce, int64_t h, int64_t w,
                      int64_t H, int64_t W, float k_ce, float k_di, float scale, double ce_norm, double di_norm,
                      bool distill, hipStream_t st) {
    // low-res logits of both nets + full-res int64 labels in, low-res gradient out
    ProfScope prof(DIGA_PROF_UPSAMPLE_LOSS, st, (double)N * h * w * C * 4.0 * (distill ? 3.0 : 2.0) + (double)n_ce * H * W * 8.0);
    CellWs cw = carve(workspace, N, C, h, w);
    const float sy = ac_scale(h, H), sx = ac_scale(w, W);
    const int tmax = (int)(h > w ? h : w);
    hipLaunchKernelGGL(cell_starts_kernel, dim3((tmax + 255) / 256), dim3(256), 0, st, cw.ystart, cw.xstart, (int)h,
                       (int)w, (int)H, (int)W, sy, sx);
    // the largest cell per axis, counted with the kernels' own float arithmetic (bilinear_cell: i0 = min(int(scale * dst), n_in - 2);
    // 8 x 8 at the DeepLab path's x7.99 -- a bound like ceil(1 / s) + 1 says 9)
    auto max_cell = [](int64_t n_in, int64_t n_out, float sc) {
        int best = 0, run = 0, prev = -1;
        for (int64_t d = 0; d < n_out; ++d) {
            int i = (int)(sc * (float)d);
            if (i > (int)n_in - 2) i = (int)n_in - 2;
            if (i < 0) i = 0;
            run = (i == prev) ? run + 1 : 1;
            prev = i;
            best = run > best ? run : best;
        }
        return best;
    };
    // cells per wave: as many neighbouring cells along x as fit a wave with ONE pixel per lane (the accumulator-free form: 8 x 8 cells of
    // the DeepLab path -> 1; the 4 x 4 / 4 x 5 / 5 x 5 cells of logits at 1/4 scale -> 2); cells beyond 64 pixels: the looping form
    const int mc = (H <= 16384 && W <= 16384) ? max_cell(h, H, sy) * max_cell(w, W, sx) : 1 << 30;
    const bool single = mc <= 64;
    int cpw;
    if (single) {
        cpw = mc <= 8 ? 8 : mc <= 16 ? 4 : mc <= 32 ? 2 : 1;
    } else {
        const double cell_px = ((double)H / (double)(h - 1)) * ((double)W / (double)(w - 1));
        cpw = cell_px >= 48.0 ? 1 : cell_px >= 24.0 ? 2 : cell_px >= 12.0 ? 4 : 8;
    }
    dim3 grid((unsigned)ceil_div(w - 1, cpw), (unsigned)(h - 1), (unsigned)N);
#define DIGA_UL_LAUNCH(D_, S_, U_)                                                                                         \
    hipLaunchKernelGGL((upsample_loss_cells_kernel<C, D_, S_, U_>), grid, dim3(64), 0, st, stu_lr, tea_lr, (const long long*)labels, \
                       cw.ystart, cw.xstart, cw.cellpart, (int)B, (int)n_ce, (int)h, (int)w, (int)H, (int)W, sy, sx, k_ce, k_di,  \
                       scale, cpw)
    if (distill) {
        if (single && cpw == 1) DIGA_UL_LAUNCH(true, true, true);
        else if (single) DIGA_UL_LAUNCH(true, true, false);
        else DIGA_UL_LAUNCH(true, false, false);
    } else {
        if (single && cpw == 1) DIGA_UL_LAUNCH(false, true, true);
        else if (single) DIGA_UL_LAUNCH(false, true, false);
        else DIGA_UL_LAUNCH(false, false, false);
    }
#undef DIGA_UL_LAUNCH
    const int64_t total = N * C * h * w;
    hipLaunchKernelGGL((upsample_loss_gather_kernel<C>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                       cw.cellpart, grad_lr, (int)N, (int)h, (int)w);
    const int64_t ncells = N * (h - 1) * (w - 1);
    hipLaunchKernelGGL(stage_sums_kernel, dim3(kStageBlocks), dim3(256), 0, st, cw.cellpart + (int64_t)(4 * C) * ncells, ncells,
                       1, ncells, distill ? 2 : 1, cw.stage);
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(1024), 0, st, cw.stage, (int64_t)kStageBlocks, 2,
                       distill ? 2 : 1, losses_out, ce_norm, di_norm);
    return launch_status("diga_upsample_loss");
}
}  // namespace

extern "C" size_t diga_upsample_loss_workspace_bytes(int64_t B2, int64_t C, int64_t h, int64_t w) {
    if (h < 2 || w < 2 || B2 < 1 || C < 1) return 0;
    return carve(nullptr, B2, C, h, w).bytes;
}

extern "C" int diga_upsample_ce_distill_fwd_bwd(const float* stu_lr, const float* tea_lr, const int64_t* labels,
                                                float* grad_stu_lr, float* losses_out, void* workspace,
                                                size_t workspace_bytes, int64_t B, int64_t C, int64_t h, int64_t w,
                                                int64_t H, int64_t W, float lambda_seg, float lambda_distil,
                                                float scale, void* stream) {
    DIGA_REQUIRE(stu_lr && tea_lr && labels && grad_stu_lr && losses_out && workspace, DIGA_EINVAL,
                 "upsample_ce_distill: null pointer");
    DIGA_REQUIRE(B > 0 && h >= 2 && w >= 2 && H >= 1 && W >= 1, DIGA_EINVAL,
                 "upsample_ce_distill: bad shape B=%lld h=%lld w=%lld H=%lld W=%lld (need h,w >= 2)", (long long)B,
                 (long long)h, (long long)w, (long long)H, (long long)W);
    DIGA_REQUIRE(C == 19 || C == 16, DIGA_EINVAL, "upsample_ce_distill: C=%lld not built (19 and 16 are)", (long long)C);
    DIGA_REQUIRE(workspace_bytes >= diga_upsample_loss_workspace_bytes(2 * B, C, h, w), DIGA_EWORKSPACE,
                 "upsample_ce_distill: workspace too small");
    const double norm = 1.0 / ((double)B * H * W);
    const float k = (float)norm;
    hipStream_t st = (hipStream_t)stream;
    if (C == 19)
        return run_upsample_loss<19>(stu_lr, tea_lr, labels, grad_stu_lr, losses_out, workspace, 2 * B, B, B, h, w, H, W,
                                     lambda_seg * k, lambda_distil * k, scale, norm, norm, true, st);
    return run_upsample_loss<16>(stu_lr, tea_lr, labels, grad_stu_lr, losses_out, workspace, 2 * B, B, B, h, w, H, W,
                                 lambda_seg * k, lambda_distil * k, scale, norm, norm, true, st);
}

extern "C" int diga_upsample_ce_fwd_bwd(const float* logits_lr, const int64_t* labels, float* grad_lr,
                                        float* loss_out, void* workspace, size_t workspace_bytes, int64_t N,
                                        int64_t C, int64_t h, int64_t w, int64_t H, int64_t W, float lambda_seg,
                                        void* stream) {
    DIGA_REQUIRE(logits_lr && labels && grad_lr && loss_out && workspace, DIGA_EINVAL, "upsample_ce: null pointer");
    DIGA_REQUIRE(N > 0 && h >= 2 && w >= 2 && H >= 1 && W >= 1, DIGA_EINVAL, "upsample_ce: bad shape");
    DIGA_REQUIRE(C == 19 || C == 16, DIGA_EINVAL, "upsample_ce: C=%lld not built (19 and 16 are)", (long long)C);
    DIGA_REQUIRE(workspace_bytes >= diga_upsample_loss_workspace_bytes(N, C, h, w), DIGA_EWORKSPACE,
                 "upsample_ce: workspace too small");
    const double norm = 1.0 / ((double)N * H * W);
    hipStream_t st = (hipStream_t)stream;
    if (C == 19)
        return run_upsample_loss<19>(logits_lr, nullptr, labels, grad_lr, loss_out, workspace, N, N, N, h, w, H, W,
                                     lambda_seg * (float)norm, 0.f, 0.f, norm, 0.0, false, st);
    return run_upsample_loss<16>(logits_lr, nullptr, labels, grad_lr, loss_out, workspace, N, N, N, h, w, H, W,
                                 lambda_seg * (float)norm, 0.f, 0.f, norm, 0.0, false, st);
}

extern "C" int diga_upsample_bilinear_ac(const float* x, float* y, int64_t NC, int64_t h, int64_t w, int64_t H,
                                         int64_t W, void* stream) {
    DIGA_REQUIRE(x && y && NC > 0 && h > 0 && w > 0 && H > 0 && W > 0, DIGA_EINVAL, "upsample_bilinear: bad argument");
    const int64_t total = NC * H * W;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, NC,
                       (int)h, (int)w, (int)H, (int)W, ac_scale(h, H), ac_scale(w, W));
    return launch_status("diga_upsample_bilinear_ac");
}
