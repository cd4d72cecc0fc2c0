// Normalisation / pooling kernels of the DeepLabV2 model on NHWC fp32 tensors ([pixel][channel], `ld`
// floats between pixels): train-mode BatchNorm with frozen affine (+ReLU, +residual), GroupNorm(32)
// (+ReLU, + per-(n,c) Dropout2d scale), SE average pool / channel gate, 3x3/2 max-pool.
// Reference: nn.BatchNorm2d / nn.GroupNorm / SEBlock / nn.Dropout2d / nn.MaxPool2d(ceil_mode) inside
// G5/model/seg_model_noaux.py:57-101,122-137,140-214,216-232 (cuDNN / ATen kernels there).
//
// All of it is HBM-bound column statistics + elementwise work.  One family of kernels serves both norms:
//   colstats_partial : per (segment, row-chunk, channel) shifted sums  sum(x-s), sum((x-s)^2)
//   *_finalize       : merge partials in double (Chan) -> mean / invstd -> per-(segment,channel) a, b
//   affine_apply     : y = [relu](x*a[seg,c] + b[seg,c] [+ residual])
//   bwd_partial      : per (segment, chunk, channel)  sum(g), sum(g*xhat),  g = dy*[y>0]
//   *_bwd_finalize   : -> k1,k2,k3 with dx = k1*g - k2 - xhat*k3   (and dgamma/dbeta for GroupNorm)
//   bwd_apply        : dx (and the residual gradient g)
// A segment is the whole tensor for BatchNorm and one image for GroupNorm / SE.  Sums are merged in a
// fixed order: deterministic, no float atomics.
#include <initializer_list>

#include "common.h"

namespace diga {

constexpr int kNormThreads = 256;

// streaming (non-temporal) 16 / 8-byte stores and 16-byte loads for the apply passes: every tensor here is 150-600 MB and
// is touched once per pass
using nf4 = __attribute__((ext_vector_type(4))) float;
using nu2 = __attribute__((ext_vector_type(2))) unsigned int;
__device__ __forceinline__ void st4s(float* p, float4 v) { __builtin_nontemporal_store((nf4){v.x, v.y, v.z, v.w}, reinterpret_cast<nf4*>(p)); }
__device__ __forceinline__ void st2s(unsigned char* p, uint2 v) { __builtin_nontemporal_store((nu2){v.x, v.y}, reinterpret_cast<nu2*>(p)); }
__device__ __forceinline__ float4 ld4s(const float* p) { const nf4 v = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }

struct ColGeom {
    int64_t rows_per_seg;   // rows of one segment
    int nseg, C, chunk_rows, nchunk;
};

// thread -> (channel quad q, row lane rl): tq = C/4 quads; if tq >= 256 every thread walks quads q, q+256, ...
// over all rows of the chunk; else 256/tq row lanes share a quad and are summed through LDS.
struct NoPrep {
    __device__ __forceinline__ void operator()(int) const {}
};

// prep(c) runs once per column group a thread walks, before its rows: per-channel coefficients belong there -- loaded inside f they
// are re-fetched for every row (the compiler cannot hoist them past the stores of the reduction: bwd_partial_kernel issued 16 scalar
// loads next to the 2 row loads of every row and ran at a quarter of the HBM rate)
template <int NV, class F, class P = NoPrep>
__device__ __forceinline__ void chunk_walk(const ColGeom& g, int seg, int chunk, float* __restrict__ out, F&& f, P&& prep = NoPrep()) {
    // out: [NV][C] partial sums of this (seg, chunk); f(row_global, c, acc[NV][4]) accumulates one float4 column group
    __shared__ float red[kNormThreads][4];
    const int tq = g.C >> 2;
    const int64_t r0 = (int64_t)seg * g.rows_per_seg + (int64_t)chunk * g.chunk_rows;
    int64_t r1 = r0 + g.chunk_rows;
    const int64_t seg_end = (int64_t)(seg + 1) * g.rows_per_seg;
    if (r1 > seg_end) r1 = seg_end;
    if (tq >= kNormThreads) {
        for (int q = threadIdx.x; q < tq; q += kNormThreads) {
            float acc[NV][4];
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[v][e] = 0.f;
            prep(q * 4);
#pragma unroll 8
            for (int64_t r = r0; r < r1; ++r) f(r, r0, q * 4, acc);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) out[(int64_t)v * g.C + q * 4 + e] = acc[v][e];
        }
    } else {
        const int lanes = kNormThreads / tq;          // row lanes per quad (tq is a power of two <= 128 here)
        const int q = threadIdx.x % tq, rl = threadIdx.x / tq;
        float acc[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[v][e] = 0.f;
        if (rl < lanes) {
            prep(q * 4);
            // 8 independent row loads in flight per thread: these kernels are pure HBM streams
            int64_t r = r0 + rl;
            for (; r + 7 * (int64_t)lanes < r1; r += 8 * (int64_t)lanes) {
#pragma unroll
                for (int u = 0; u < 8; ++u) f(r + u * (int64_t)lanes, r0, q * 4, acc);
            }
            for (; r < r1; r += lanes) f(r, r0, q * 4, acc);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = acc[v][e];
            __syncthreads();
            if (threadIdx.x < tq) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                for (int l = 0; l < lanes; ++l)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += red[l * tq + threadIdx.x][e];
#pragma unroll
                for (int e = 0; e < 4; ++e) out[(int64_t)v * g.C + threadIdx.x * 4 + e] = s[e];
            }
        }
    }
}

// partial[(seg*nchunk + chunk)][3][C] = { sum(x - s), sum((x - s)^2), s }, s = x[first row of the chunk]
__global__ __launch_bounds__(kNormThreads) void colstats_partial_kernel(const float* __restrict__ x, int64_t ld,
                                                                         ColGeom g, float* __restrict__ partial) {
    const int chunk = blockIdx.x, seg = blockIdx.y;
    float* out = partial + ((int64_t)seg * g.nchunk + chunk) * 3 * g.C;
    const int64_t r0 = (int64_t)seg * g.rows_per_seg + (int64_t)chunk * g.chunk_rows;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);                    // the shift: the chunk's first row, fetched once per column group
    chunk_walk<2>(g, seg, chunk, out, [&](int64_t r, int64_t, int c, float (&acc)[2][4]) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        const float d0 = v.x - s.x, d1 = v.y - s.y, d2 = v.z - s.z, d3 = v.w - s.w;
        acc[0][0] += d0; acc[0][1] += d1; acc[0][2] += d2; acc[0][3] += d3;
        acc[1][0] += d0 * d0; acc[1][1] += d1 * d1; acc[1][2] += d2 * d2; acc[1][3] += d3 * d3;
    }, [&](int c) { s = *reinterpret_cast<const float4*>(x + r0 * ld + c); });
    for (int c = threadIdx.x; c < g.C; c += kNormThreads) out[2 * g.C + c] = x[r0 * ld + c];
}

__device__ __forceinline__ int64_t chunk_count(const ColGeom& g, int chunk) {
    const int64_t a = (int64_t)chunk * g.chunk_rows;
    int64_t b = a + g.chunk_rows;
    if (b > g.rows_per_seg) b = g.rows_per_seg;
    return b - a;
}

// merge the chunk partials of one (segment, channel) -> count, mean, M2  (Chan et al., double)
__device__ __forceinline__ void merge_channel(const float* __restrict__ partial, const ColGeom& g, int seg, int c,
                                              double& n, double& mean, double& m2) {
    n = 0.0; mean = 0.0; m2 = 0.0;
    for (int k = 0; k < g.nchunk; ++k) {
        const float* p = partial + ((int64_t)seg * g.nchunk + k) * 3 * g.C;
        const double nb = (double)chunk_count(g, k);
        const double sd = p[c], sd2 = p[g.C + c], sh = p[2 * g.C + c];
        const double mb = sh + sd / nb, m2b = sd2 - sd * sd / nb;
        const double tot = n + nb, delta = mb - mean;
        mean += delta * nb / tot;
        m2 += m2b + delta * delta * n * nb / tot;
        n = tot;
    }
}


// eval-mode BatchNorm: a, b from the running statistics
__global__ void bn_eval_ab_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                  float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ ab,
                                  int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(running_var[c] + eps);
    save_mean[c] = running_mean[c];
    save_invstd[c] = invstd;
    const float a = invstd * gamma[c];
    ab[c] = a;
    ab[C + c] = beta[c] - running_mean[c] * a;
}

// GroupNorm: one (image, group) population of rows_per_seg * cpg values.  save_* [N][G]; ab [2][N][C]
__global__ void gn_finalize_kernel(const float* __restrict__ partial, ColGeom g, int G, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ chan_scale,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ ab,
                                   float eps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (n, group)
    if (idx >= g.nseg * G) return;
    const int n_img = idx / G, grp = idx - n_img * G;
    const int cpg = g.C / G;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
        double nb, mb, m2b;
        merge_channel(partial, g, n_img, grp * cpg + j, nb, mb, m2b);
        const double tot = n + nb, delta = mb - mean;
        mean += delta * nb / tot;
        m2 += m2b + delta * delta * n * nb / tot;
        n = tot;
    }
    const float var = (float)(m2 / n);
    const float invstd = 1.f / sqrtf(var + eps);
    const float mu = (float)mean;
    save_mean[idx] = mu;
    save_invstd[idx] = invstd;
    const int64_t NC = (int64_t)g.nseg * g.C;
    for (int j = 0; j < cpg; ++j) {
        const int c = grp * cpg + j;
        const float s = chan_scale ? chan_scale[(int64_t)n_img * g.C + c] : 1.f;
        const float a = invstd * gamma[c];
        ab[(int64_t)n_img * g.C + c] = a * s;
        ab[NC + (int64_t)n_img * g.C + c] = (beta[c] - mu * a) * s;
    }
}

// y = [relu](x*a + b [+ res]); a,b indexed [seg*ab_seg_stride + c] (stride 0 = shared by all segments)
// 4 fp32 -> 4 bf16 hi + 4 bf16 lo (the split of csrc/conv.hip: hi = bf16(x) round-to-nearest-even, lo = bf16(x - hi))
__device__ __forceinline__ void norm_split4(const float4 v, uint2& hi, uint2& lo) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    hi.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){v.x, v.y}, bf2));
    hi.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){v.z, v.w}, bf2));
    const float h0 = __uint_as_float(hi.x << 16), h1 = __uint_as_float(hi.x & 0xffff0000u);
    const float h2 = __uint_as_float(hi.y << 16), h3 = __uint_as_float(hi.y & 0xffff0000u);
    lo.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){v.x - h0, v.y - h1}, bf2));
    lo.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){v.z - h2, v.w - h3}, bf2));
}

// twin_out: y receives the split twin of the result ([row][C/8][hi8 | lo8], dense rows of 4*C bytes) instead of fp32 --
// for a tensor whose only readers are convolutions on the twin kernels
// Round 5: the operand combination (residual?, ReLU?, mask bits?, twin output?, per-segment coefficients?) is a TEMPLATE parameter --
// with run-time flags every instantiation carried the registers of its heaviest path (the split-twin output: 54) and the flag tests per
// row; the forms the fp32 step runs need 30-40.  Same expression per element, bit for bit.
template <bool RES, bool RELU, bool BITS, bool TWIN, bool SEG>
__global__ __launch_bounds__(256) void affine_apply_kernel(const float* __restrict__ x, int64_t ld_x, float* __restrict__ y,
                                                           int64_t ld_y, const float* __restrict__ res, int64_t ld_r,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           int64_t ab_seg_stride, int64_t rows_per_seg, int64_t rows,
                                                           int C, unsigned char* __restrict__ relu_bits) {
    // a thread keeps its channel quad(s) and walks rows: no per-element division, coefficients in registers
    const int tq = C >> 2;
    const int tpr = tq < 256 ? tq : 256;          // threads per row
    const int rpb = 256 / tpr;                    // rows per block iteration
    const int q0 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    if (rl >= rpb) return;
    const int nrows = (int)rows, rps = (int)rows_per_seg;
    const int rstep = gridDim.x * rpb;
    for (int q = q0; q < tq; q += tpr) {
        const int c = q * 4;
        float4 av = *reinterpret_cast<const float4*>(a + c);
        float4 bv = b != nullptr ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        int cur_seg = 0;
        for (int r = blockIdx.x * rpb + rl; r < nrows; r += rstep) {
            if (SEG) {
                const int seg = r / rps;
                if (seg != cur_seg) {
                    cur_seg = seg;
                    av = *reinterpret_cast<const float4*>(a + (int64_t)seg * ab_seg_stride + c);
                    if (b != nullptr) bv = *reinterpret_cast<const float4*>(b + (int64_t)seg * ab_seg_stride + c);
                }
            }
            const float4 xv = ld4s(x + (int64_t)r * ld_x + c);
            float4 o;
            // explicit fma: the backward kernels re-derive the ReLU mask of a residual-free BatchNorm from x with the
            // same expression instead of re-reading y
            o.x = __builtin_fmaf(xv.x, av.x, bv.x); o.y = __builtin_fmaf(xv.y, av.y, bv.y);
            o.z = __builtin_fmaf(xv.z, av.z, bv.z); o.w = __builtin_fmaf(xv.w, av.w, bv.w);
            if (RES) {
                const float4 rv = ld4s(res + (int64_t)r * ld_r + c);
                o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
            }
            if (RELU) {
                if (BITS) {
                    // the ReLU mask as one bit per channel (C % 32 == 0): 8 lanes = 32 consecutive channels of one row OR
                    // their nibbles together (two quad permutes + a half-row mirror) and the first lane stores the word
                    unsigned v = ((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u))
                                 << (4 * (threadIdx.x & 7));
                    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
                    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
                    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);     // row_half_mirror
                    if ((threadIdx.x & 7) == 0)
                        *reinterpret_cast<unsigned*>(relu_bits + (int64_t)r * (C >> 3) + (c >> 3)) = v;
                }
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if (TWIN) {
                uint2 hi, lo;
                norm_split4(o, hi, lo);
                unsigned char* tw = reinterpret_cast<unsigned char*>(y) + ((int64_t)r * (C >> 3) + (c >> 3)) * 32 + ((c >> 2) & 1) * 8;
                st2s(tw, hi);
                st2s(tw + 16, lo);
            } else {
                st4s(y + (int64_t)r * ld_y + c, o);
            }
        }
    }
}

template <bool RES, bool RELU, bool BITS, bool TWIN>
static void launch_affine_seg(dim3 grid, hipStream_t st, const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* res, int64_t ld_r,
                              const float* a, const float* b, int64_t ab_seg_stride, int64_t rows_per_seg, int64_t rows, int C,
                              unsigned char* relu_bits) {
    if (ab_seg_stride != 0)
        hipLaunchKernelGGL((affine_apply_kernel<RES, RELU, BITS, TWIN, true>), grid, dim3(256), 0, st, x, ld_x, y, ld_y, res, ld_r, a, b, ab_seg_stride,
                           rows_per_seg, rows, C, relu_bits);
    else
        hipLaunchKernelGGL((affine_apply_kernel<RES, RELU, BITS, TWIN, false>), grid, dim3(256), 0, st, x, ld_x, y, ld_y, res, ld_r, a, b, ab_seg_stride,
                           rows_per_seg, rows, C, relu_bits);
}
// the run-time flags of the entry points -> the instantiation (relu_bits only with relu; a twin output only without mask bits)
static void launch_affine(dim3 grid, hipStream_t st, const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* res, int64_t ld_r,
                          const float* a, const float* b, int64_t ab_seg_stride, int64_t rows_per_seg, int64_t rows, int C, int relu,
                          int twin_out, unsigned char* relu_bits = nullptr) {
#define DIGA_AFF(RES, RELU, BITS, TWIN) \
    launch_affine_seg<RES, RELU, BITS, TWIN>(grid, st, x, ld_x, y, ld_y, res, ld_r, a, b, ab_seg_stride, rows_per_seg, rows, C, relu_bits)
    const bool bits = relu && relu_bits != nullptr;
    if (res != nullptr) {
        if (relu) {
            if (bits) { if (twin_out) DIGA_AFF(true, true, true, true); else DIGA_AFF(true, true, true, false); }
            else { if (twin_out) DIGA_AFF(true, true, false, true); else DIGA_AFF(true, true, false, false); }
        } else { if (twin_out) DIGA_AFF(true, false, false, true); else DIGA_AFF(true, false, false, false); }
    } else {
        if (relu) {
            if (bits) { if (twin_out) DIGA_AFF(false, true, true, true); else DIGA_AFF(false, true, true, false); }
            else { if (twin_out) DIGA_AFF(false, true, false, true); else DIGA_AFF(false, true, false, false); }
        } else { if (twin_out) DIGA_AFF(false, false, false, true); else DIGA_AFF(false, false, false, false); }
    }
#undef DIGA_AFF
}

// partial[(seg*nchunk+chunk)][2][C] = { sum g, sum g*xhat },  g = dy * [y > 0] (relu) , xhat = (x-mean)*invstd
// mean/invstd indexed [seg*st_seg_stride + c / cdiv]  (BatchNorm: stride 0, cdiv 1; GroupNorm: stride G, cdiv cpg)
__global__ __launch_bounds__(kNormThreads) void bwd_partial_kernel(const float* __restrict__ dy, int64_t ld_dy,
                                                                    const float* __restrict__ x, int64_t ld_x,
                                                                    const float* __restrict__ y, int64_t ld_y,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, int st_seg_stride,
                                                                    int cdiv, ColGeom g, float* __restrict__ partial,
                                                                    const float* __restrict__ relu_ab) {
    // relu_ab (BatchNorm without residual, y == nullptr): [2][C] forward coefficients; mask = fma(x, a, b) > 0
    const int chunk = blockIdx.x, seg = blockIdx.y;
    float* out = partial + ((int64_t)seg * g.nchunk + chunk) * 2 * g.C;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f}, ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
    chunk_walk<2>(g, seg, chunk, out, [&](int64_t r, int64_t, int c, float (&acc)[2][4]) {
        const float4 gv = *reinterpret_cast<const float4*>(dy + r * ld_dy + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ld_x + c);
        float gg[4] = {gv.x, gv.y, gv.z, gv.w};
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
        if (y != nullptr) {
            const float4 yv = *reinterpret_cast<const float4*>(y + r * ld_y + c);
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) gg[e] = yy[e] > 0.f ? gg[e] : 0.f;
        } else if (relu_ab != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gg[e] = __builtin_fmaf(xx[e], ra[e], rb[e]) > 0.f ? gg[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xx[e] - mu[e]) * is[e];                  // null statistics: mean 0, invstd 1
            acc[0][e] += gg[e];
            acc[1][e] += gg[e] * xh;
        }
    }, [&](int c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int si = seg * st_seg_stride + (c + e) / cdiv;
            mu[e] = mean != nullptr ? mean[si] : 0.f;
            is[e] = mean != nullptr ? invstd[si] : 1.f;
            ra[e] = relu_ab != nullptr ? relu_ab[c + e] : 0.f;
            rb[e] = relu_ab != nullptr ? relu_ab[g.C + c + e] : 0.f;
        }
    });
}


// GroupNorm backward: per (n,group) c1 = sum_c gamma*S1 / cnt, c2 = sum_c gamma*S2 / cnt; kk [3][N][C];
// S1/S2 already carry the Dropout2d scale through g_eff = scale*dy (applied here).  dgamma/dbeta [C].
__global__ void gn_bwd_finalize_kernel(const float* __restrict__ partial, ColGeom g, int G, const float* __restrict__ gamma,
                                       const float* __restrict__ chan_scale, const float* __restrict__ invstd,
                                       float* __restrict__ kk, float* __restrict__ chan_sums /* [2][N][C] */) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (n, group)
    if (idx >= g.nseg * G) return;
    const int n_img = idx / G, grp = idx - n_img * G;
    const int cpg = g.C / G;
    const int64_t NC = (int64_t)g.nseg * g.C;
    double c1 = 0.0, c2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
        const int c = grp * cpg + j;
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < g.nchunk; ++k) {
            const float* p = partial + ((int64_t)n_img * g.nchunk + k) * 2 * g.C;
            s1 += (double)p[c];
            s2 += (double)p[g.C + c];
        }
        const double sc = chan_scale ? (double)chan_scale[(int64_t)n_img * g.C + c] : 1.0;
        s1 *= sc;
        s2 *= sc;
        chan_sums[(int64_t)n_img * g.C + c] = (float)s1;          // d beta contribution of image n
        chan_sums[NC + (int64_t)n_img * g.C + c] = (float)s2;     // d gamma contribution
        c1 += (double)gamma[c] * s1;
        c2 += (double)gamma[c] * s2;
    }
    const double cnt = (double)g.rows_per_seg * cpg;
    const float is = invstd[idx];
    for (int j = 0; j < cpg; ++j) {
        const int c = grp * cpg + j;
        const float sc = chan_scale ? chan_scale[(int64_t)n_img * g.C + c] : 1.f;
        kk[(int64_t)n_img * g.C + c] = sc * gamma[c] * is;
        kk[NC + (int64_t)n_img * g.C + c] = (float)(is * (c1 / cnt));
        kk[2 * NC + (int64_t)n_img * g.C + c] = (float)(is * (c2 / cnt));
    }
}

__global__ void gn_param_grad_kernel(const float* __restrict__ chan_sums, int N, int C, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float b = 0.f, gm = 0.f;
    for (int n = 0; n < N; ++n) {
        b += chan_sums[(int64_t)n * C + c];
        gm += chan_sums[(int64_t)N * C + (int64_t)n * C + c];
    }
    dbeta[c] = b;
    dgamma[c] = gm;
}

// dx = k1*g - k2 - xhat*k3 ; dres = g   (g = dy*[y>0])
// Round 5: as affine_apply_kernel, the operand combination is a template parameter (MASK: 0 the gradient arrives masked / no ReLU,
// 1 mask = y > 0, 2 mask = fma(x, a, b) > 0; DRES: also store the masked gradient; TWIN: dx as the split twin; SEG: per-segment
// coefficients = GroupNorm) -- 92 registers with run-time flags, 44-64 per instantiation; same arithmetic per element.
template <int MASK, bool DRES, bool TWIN, bool SEG>
__global__ __launch_bounds__(256) void bwd_apply_kernel(const float* __restrict__ dy, int64_t ld_dy,
                                                        const float* __restrict__ x, int64_t ld_x,
                                                        const float* __restrict__ y, int64_t ld_y,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        int st_seg_stride, int cdiv, const float* __restrict__ kk,
                                                        int64_t kk_seg_stride, int64_t kk_plane, float* __restrict__ dx,
                                                        int64_t ld_dx, float* __restrict__ dres, int64_t ld_dr,
                                                        int64_t rows_per_seg, int64_t rows, int C,
                                                        const float* __restrict__ relu_ab) {
    // TWIN: dx receives the split twin ([row][C/8][hi8 | lo8]) instead of fp32 (its only readers are twin kernels)
    const int tq = C >> 2;
    const int tpr = tq < 256 ? tq : 256;
    const int rpb = 256 / tpr;
    const int q0 = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    if (rl >= rpb) return;
    const int nrows = (int)rows, rps = (int)rows_per_seg;
    const int rstep = gridDim.x * rpb;
    for (int q = q0; q < tq; q += tpr) {
        const int c = q * 4;
        float k1a[4], k2a[4], k3a[4], mu[4], is[4];
        float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
        if (MASK == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ra[e] = relu_ab[c + e];
                rb[e] = relu_ab[C + c + e];
            }
        }
        auto load_coeff = [&](int seg) {
            const float* kb = kk + (int64_t)seg * kk_seg_stride + c;
            const float4 k1 = *reinterpret_cast<const float4*>(kb);
            const float4 k2 = *reinterpret_cast<const float4*>(kb + kk_plane);
            const float4 k3 = *reinterpret_cast<const float4*>(kb + 2 * kk_plane);
            k1a[0] = k1.x; k1a[1] = k1.y; k1a[2] = k1.z; k1a[3] = k1.w;
            k2a[0] = k2.x; k2a[1] = k2.y; k2a[2] = k2.z; k2a[3] = k2.w;
            k3a[0] = k3.x; k3a[1] = k3.y; k3a[2] = k3.z; k3a[3] = k3.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int si = seg * st_seg_stride + (c + e) / cdiv;
                mu[e] = mean[si];
                is[e] = invstd[si];
            }
        };
        load_coeff(0);
        int cur_seg = 0;
        for (int r = blockIdx.x * rpb + rl; r < nrows; r += rstep) {
            if (SEG) {
                const int seg = r / rps;
                if (seg != cur_seg) {
                    cur_seg = seg;
                    load_coeff(seg);
                }
            }
            const float4 gv = ld4s(dy + (int64_t)r * ld_dy + c);
            const float4 xv = ld4s(x + (int64_t)r * ld_x + c);
            float gg[4] = {gv.x, gv.y, gv.z, gv.w};
            const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
            if (MASK == 1) {
                const float4 yv = ld4s(y + (int64_t)r * ld_y + c);
                const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) gg[e] = yy[e] > 0.f ? gg[e] : 0.f;
            } else if (MASK == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gg[e] = __builtin_fmaf(xx[e], ra[e], rb[e]) > 0.f ? gg[e] : 0.f;
            }
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xx[e] - mu[e]) * is[e];
                o[e] = k1a[e] * gg[e] - k2a[e] - xh * k3a[e];
            }
            if (TWIN) {
                uint2 hi, lo;
                norm_split4(make_float4(o[0], o[1], o[2], o[3]), hi, lo);
                unsigned char* tw = reinterpret_cast<unsigned char*>(dx) + ((int64_t)r * (C >> 3) + (c >> 3)) * 32 + ((c >> 2) & 1) * 8;
                st2s(tw, hi);
                st2s(tw + 16, lo);
            } else {
                st4s(dx + (int64_t)r * ld_dx + c, make_float4(o[0], o[1], o[2], o[3]));
            }
            if (DRES) st4s(dres + (int64_t)r * ld_dr + c, make_float4(gg[0], gg[1], gg[2], gg[3]));
        }
    }
}

template <int MASK, bool DRES, bool TWIN>
static void launch_bwd_apply_seg(dim3 grid, hipStream_t st, const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y,
                                 int64_t ld_y, const float* mean, const float* invstd, int st_seg_stride, int cdiv, const float* kk,
                                 int64_t kk_seg_stride, int64_t kk_plane, float* dx, int64_t ld_dx, float* dres, int64_t ld_dr,
                                 int64_t rows_per_seg, int64_t rows, int C, const float* relu_ab) {
    if (kk_seg_stride != 0 || st_seg_stride != 0)
        hipLaunchKernelGGL((bwd_apply_kernel<MASK, DRES, TWIN, true>), grid, dim3(256), 0, st, dy, ld_dy, x, ld_x, y, ld_y, mean, invstd, st_seg_stride,
                           cdiv, kk, kk_seg_stride, kk_plane, dx, ld_dx, dres, ld_dr, rows_per_seg, rows, C, relu_ab);
    else
        hipLaunchKernelGGL((bwd_apply_kernel<MASK, DRES, TWIN, false>), grid, dim3(256), 0, st, dy, ld_dy, x, ld_x, y, ld_y, mean, invstd, st_seg_stride,
                           cdiv, kk, kk_seg_stride, kk_plane, dx, ld_dx, dres, ld_dr, rows_per_seg, rows, C, relu_ab);
}
static void launch_bwd_apply(dim3 grid, hipStream_t st, const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                             const float* mean, const float* invstd, int st_seg_stride, int cdiv, const float* kk, int64_t kk_seg_stride,
                             int64_t kk_plane, float* dx, int64_t ld_dx, float* dres, int64_t ld_dr, int64_t rows_per_seg, int64_t rows, int C,
                             const float* relu_ab, int twin_out) {
#define DIGA_BA(MASK, DRES, TWIN)                                                                                                          \
    launch_bwd_apply_seg<MASK, DRES, TWIN>(grid, st, dy, ld_dy, x, ld_x, y, ld_y, mean, invstd, st_seg_stride, cdiv, kk, kk_seg_stride, kk_plane, \
                                           dx, ld_dx, dres, ld_dr, rows_per_seg, rows, C, relu_ab)
    const int mk = y != nullptr ? 1 : relu_ab != nullptr ? 2 : 0;
    if (dres != nullptr) {
        if (twin_out) { if (mk == 1) DIGA_BA(1, true, true); else if (mk == 2) DIGA_BA(2, true, true); else DIGA_BA(0, true, true); }
        else { if (mk == 1) DIGA_BA(1, true, false); else if (mk == 2) DIGA_BA(2, true, false); else DIGA_BA(0, true, false); }
    } else {
        if (twin_out) { if (mk == 1) DIGA_BA(1, false, true); else if (mk == 2) DIGA_BA(2, false, true); else DIGA_BA(0, false, true); }
        else { if (mk == 1) DIGA_BA(1, false, false); else if (mk == 2) DIGA_BA(2, false, false); else DIGA_BA(0, false, false); }
    }
#undef DIGA_BA
}

// out[seg][c] = mean over the segment's rows (SE global average pool) from colstats partials
__global__ void seg_mean_kernel(const float* __restrict__ partial, ColGeom g, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.nseg * g.C) return;
    const int seg = idx / g.C, c = idx - seg * g.C;
    double n, mean, m2;
    merge_channel(partial, g, seg, c, n, mean, m2);
    out[idx] = (float)mean;
}

// out[seg][c] = sum over the segment's rows (bias gradient of a convolution), same merge in double
__global__ void seg_sum_kernel(const float* __restrict__ partial, ColGeom g, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.nseg * g.C) return;
    const int seg = idx / g.C, c = idx - seg * g.C;
    double n, mean, m2;
    merge_channel(partial, g, seg, c, n, mean, m2);
    out[idx] = (float)(mean * n);
}

// out[seg][c] = sum_rows dy*x from bwd_partial's second plane (mean = 0, invstd = 1)
__global__ void seg_dot_kernel(const float* __restrict__ partial, ColGeom g, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.nseg * g.C) return;
    const int seg = idx / g.C, c = idx - seg * g.C;
    double s = 0.0;
    for (int k = 0; k < g.nchunk; ++k) s += (double)partial[((int64_t)seg * g.nchunk + k) * 2 * g.C + g.C + c];
    out[idx] = (float)s;
}

// ---- 3x3 stride-2 pad-1 max-pool (ceil_mode), NHWC; idx = winning tap 0..8 (first maximum in scan order)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
    const int tq = C >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * tq;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % tq) * 4;
        int64_t p = i / tq;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hi = ho * 2 - 1 + t / 3, wi = wo * 2 - 1 + t % 3;
            if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
                const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)n * H + hi) * W + wi) * C + c);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (vv[e] > best[e]) {
                        best[e] = vv[e];
                        arg[e] = t;
                    }
            }
        }
        const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * C + c;
        *reinterpret_cast<float4*>(y + o) = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<uint32_t*>(idx + o) = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) |
                                                 ((uint32_t)arg[3] << 24);
    }
}

// gather form: every input element sums the <=4 windows that cover it and elected it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                          int Wo) {
    const int tq = C >> 2;
    const int64_t total = (int64_t)N * H * W * tq;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % tq) * 4;
        int64_t p = i / tq;
        const int wi = (int)(p % W);
        p /= W;
        const int hi = (int)(p % H);
        const int n = (int)(p / H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // windows ho with ho*2-1 <= hi <= ho*2+1
        const int ho0 = hi / 2, wo0 = wi / 2;    // candidates: ho0, ho0+1 (the latter only when hi is odd)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ho = ho0 + a;
            const int ty = hi - (ho * 2 - 1);
            if (ho >= Ho || ty < 0 || ty > 2) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int wo = wo0 + b;
                const int tx = wi - (wo * 2 - 1);
                if (wo >= Wo || tx < 0 || tx > 2) continue;
                const int tap = ty * 3 + tx;
                const int64_t o = (((int64_t)n * Ho + ho) * Wo + wo) * C + c;
                const uint32_t pk = *reinterpret_cast<const uint32_t*>(idx + o);
                const float4 g = *reinterpret_cast<const float4*>(dy + o);
                if ((int)(pk & 255u) == tap) acc[0] += g.x;
                if ((int)((pk >> 8) & 255u) == tap) acc[1] += g.y;
                if ((int)((pk >> 16) & 255u) == tap) acc[2] += g.z;
                if ((int)(pk >> 24) == tap) acc[3] += g.w;
            }
        }
        *reinterpret_cast<float4*>(dx + (((int64_t)n * H + hi) * W + wi) * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ---- lane-parallel, division-free finalisers -------------------------------------------------------
// Merging K chunk partials serially per channel (with double divisions) cost 0.6-0.8 ms per layer; here 8
// lanes share a channel, the loops contain only double multiply-adds, and the two-pass form
//   mean = sum_k (n_k*s_k + sd_k) / N ;  M2 = sum_k [ sd2_k - sd_k^2/n_k + n_k*(s_k + sd_k/n_k - mean)^2 ]
// needs the reciprocal of just two distinct chunk sizes.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct ChunkInv {
    double n_full, n_last, inv_full, inv_last;
};
__device__ __forceinline__ ChunkInv chunk_inv(const ColGeom& g) {
    ChunkInv ci;
    ci.n_full = (double)g.chunk_rows;
    ci.n_last = (double)(g.rows_per_seg - (int64_t)(g.nchunk - 1) * g.chunk_rows);
    ci.inv_full = 1.0 / ci.n_full;
    ci.inv_last = 1.0 / ci.n_last;
    return ci;
}

// block = kFinCh channels x kFinLn lanes; all (segment, chunk) partials of a channel form one population
constexpr int kFinCh = 16, kFinLn = 16;
__global__ __launch_bounds__(256) void bn_finalize2_kernel(const float* __restrict__ partial, ColGeom g,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                           float* __restrict__ ab, float momentum, float eps,
                                                           const float* __restrict__ counts = nullptr) {
    // counts (nullable): rows of every record when they are not g.chunk_rows-sized row chunks (the Winograd output transform's
    // per-tile-group records, diga_bn_fwd_records); a record may be empty (count 0)
    __shared__ double red[kFinLn][kFinCh];
    const int cl = threadIdx.x % kFinCh, lane = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    const bool live = c < g.C;
    const ChunkInv ci = chunk_inv(g);
    const int K = g.nseg * g.nchunk;
    double s = 0.0;
    if (live)
#pragma unroll 4
        for (int k = lane; k < K; k += kFinLn) {
            const float* p = partial + (int64_t)k * 3 * g.C;
            const double nk = counts != nullptr ? (double)counts[k] : ((k % g.nchunk) == g.nchunk - 1) ? ci.n_last : ci.n_full;
            s += nk * (double)p[2 * g.C + c] + (double)p[c];
        }
    red[lane][cl] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int l = 0; l < kFinLn; ++l) tot += red[l][cl];
    const double N = (double)g.nseg * (double)g.rows_per_seg;
    const double mean = tot / N;
    __syncthreads();
    double m2 = 0.0;
    if (live)
#pragma unroll 4
        for (int k = lane; k < K; k += kFinLn) {
            const float* p = partial + (int64_t)k * 3 * g.C;
            const bool last = (k % g.nchunk) == g.nchunk - 1;
            double nk = last ? ci.n_last : ci.n_full, ik = last ? ci.inv_last : ci.inv_full;
            if (counts != nullptr) {
                nk = (double)counts[k];
                ik = nk > 0.0 ? 1.0 / nk : 0.0;
            }
            const double sd = p[c], sd2 = p[g.C + c], sh = p[2 * g.C + c];
            const double d = sh + sd * ik - mean;
            m2 += sd2 - sd * sd * ik + nk * d * d;
        }
    red[lane][cl] = m2;
    __syncthreads();
    if (lane != 0 || !live) return;
    double M2 = 0.0;
#pragma unroll
    for (int l = 0; l < kFinLn; ++l) M2 += red[l][cl];
    const float var = (float)(M2 / N);
    const float invstd = 1.f / sqrtf(var + eps);
    const float mu = (float)mean;
    save_mean[c] = mu;
    save_invstd[c] = invstd;
    const float a = invstd * gamma[c];
    ab[c] = a;
    ab[g.C + c] = beta[c] - mu * a;
    if (running_mean != nullptr) {
        const float unbiased = N > 1.0 ? (float)(M2 / (N - 1.0)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// Folds `group` consecutive row-chunk partials {sum d, sum d^2, shift} (each over `g.chunk_rows` rows, the last
// one ragged) into one partial per group, re-based on the shift of the group's first chunk.  Block = 4 chunk
// lanes x 64 channels, so every load is a coalesced 256-byte row piece; sums are formed in double.
__global__ __launch_bounds__(256) void merge_partials_kernel(const float* __restrict__ partial, ColGeom g, int group,
                                                             float* __restrict__ merged, const float* __restrict__ counts = nullptr,
                                                             float* __restrict__ mcounts = nullptr) {
    // counts / mcounts (nullable): per-record row counts in, per-group sums out (records of unequal size, see bn_finalize2_kernel)
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const bool live = c < g.C;
    const int k0 = blockIdx.x * group;
    const int k1 = min(k0 + group, g.nchunk);
    const ChunkInv ci = chunk_inv(g);
    double s1 = 0.0, s2 = 0.0;
    float base = 0.f;
    if (counts != nullptr && threadIdx.x == 0 && blockIdx.y == 0) {
        float tot = 0.f;                                   // (integers < 2^24: exact in fp32)
        for (int k = k0; k < k1; ++k) tot += counts[k];
        mcounts[blockIdx.x] = tot;
    }
    if (live) {
        base = partial[((int64_t)k0 * 3 + 2) * g.C + c];
        for (int k = k0 + lane; k < k1; k += 4) {
            const float* p = partial + (int64_t)k * 3 * g.C;
            const double nk = counts != nullptr ? (double)counts[k] : (k == g.nchunk - 1) ? ci.n_last : ci.n_full;
            const double sd = p[c], sd2 = p[g.C + c], dl = (double)p[2 * g.C + c] - (double)base;
            s1 += sd + nk * dl;
            s2 += sd2 + 2.0 * dl * sd + nk * dl * dl;
        }
    }
    red[0][lane][cl] = s1;
    red[1][lane][cl] = s2;
    __syncthreads();
    if (lane != 0 || !live) return;
    float* o = merged + (int64_t)blockIdx.x * 3 * g.C;
    o[c] = (float)(red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
    o[g.C + c] = (float)(red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
    o[2 * g.C + c] = base;
}

// Folds K rows of a [K][W] matrix of partial sums into `parts` rows (part p sums rows p, p + parts, ...): block = 4 row
// lanes x 64 columns, coalesced 256-byte row pieces, sums in double.  Used in front of bn_bwd_finalize2_kernel when the
// backward-data epilogue delivered one partial per 128-row chunk (> 1000 chunks at C2 sizes: walking them with the
// finaliser's 16 lanes per channel took 44 us per layer).
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ in, int K, int W, int parts,
                                                          float* __restrict__ out) {
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, part = blockIdx.y;
    double acc = 0.0;
    if (c < W)
        for (int k = part + lane * parts; k < K; k += 4 * parts) acc += (double)in[(int64_t)k * W + c];
    red[lane][cl] = acc;
    __syncthreads();
    if (lane == 0 && c < W) out[(int64_t)part * W + c] = (float)(red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize2_kernel(const float* __restrict__ partial, ColGeom g,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ invstd, float* __restrict__ kk,
                                                               int training, float* __restrict__ dgamma = nullptr,
                                                               float* __restrict__ dbeta = nullptr) {
    __shared__ double red[2][kFinLn][kFinCh];
    const int cl = threadIdx.x % kFinCh, lane = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    const bool live = c < g.C;
    // (eval mode with a trainable affine pair: the partial sums exist -- they ARE dgamma / dbeta -- but dx takes no mean terms)
    const int K = (training || dgamma) ? g.nseg * g.nchunk : 0;
    double s1 = 0.0, s2 = 0.0;
    if (live)
#pragma unroll 4
        for (int k = lane; k < K; k += kFinLn) {
            s1 += (double)partial[(int64_t)k * 2 * g.C + c];
            s2 += (double)partial[(int64_t)k * 2 * g.C + g.C + c];
        }
    red[0][lane][cl] = s1;
    red[1][lane][cl] = s2;
    __syncthreads();
    if (lane != 0 || !live) return;
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int l = 0; l < kFinLn; ++l) {
        t1 += red[0][l][cl];
        t2 += red[1][l][cl];
    }
    const double M = (double)g.nseg * (double)g.rows_per_seg;
    const float k1 = gamma[c] * invstd[c];
    kk[c] = k1;
    kk[g.C + c] = training ? (float)(k1 * (t1 / M)) : 0.f;
    kk[2 * g.C + c] = training ? (float)(k1 * (t2 / M)) : 0.f;
    // a trainable affine pair (the SegFormer head's BatchNorm): the two column sums ARE its gradients
    if (dgamma) dgamma[c] = (float)t2;
    if (dbeta) dbeta[c] = (float)t1;
}

// one wave per (image, group): lane = (channel j = lane % cpg_pad, chunk lane); requires cpg <= 64
__global__ __launch_bounds__(256) void gn_finalize2_kernel(const float* __restrict__ partial, ColGeom g, int G,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ chan_scale, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, float* __restrict__ ab, float eps) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);     // (n, group)
    if (idx >= g.nseg * G) return;
    const int n_img = idx / G, grp = idx - n_img * G;
    const int cpg = g.C / G;
    const int lanes_per_ch = 64 / cpg > 0 ? 64 / cpg : 1;
    const int j = lane % cpg, cl = lane / cpg;
    const bool act = cl < lanes_per_ch;
    const int c = grp * cpg + j;
    const ChunkInv ci = chunk_inv(g);
    double s = 0.0;
    if (act)
        for (int k = cl; k < g.nchunk; k += lanes_per_ch) {
            const float* p = partial + ((int64_t)n_img * g.nchunk + k) * 3 * g.C;
            const double nk = (k == g.nchunk - 1) ? ci.n_last : ci.n_full;
            s += nk * (double)p[2 * g.C + c] + (double)p[c];
        }
    const double N = (double)g.rows_per_seg * cpg;
    const double mean = wave_sum_f64(s) / N;
    double m2 = 0.0;
    if (act)
        for (int k = cl; k < g.nchunk; k += lanes_per_ch) {
            const float* p = partial + ((int64_t)n_img * g.nchunk + k) * 3 * g.C;
            const bool last = k == g.nchunk - 1;
            const double nk = last ? ci.n_last : ci.n_full, ik = last ? ci.inv_last : ci.inv_full;
            const double sd = p[c], sd2 = p[g.C + c], sh = p[2 * g.C + c];
            const double d = sh + sd * ik - mean;
            m2 += sd2 - sd * sd * ik + nk * d * d;
        }
    const double M2 = wave_sum_f64(m2);
    const float var = (float)(M2 / N);
    const float invstd = 1.f / sqrtf(var + eps);
    const float mu = (float)mean;
    if (lane == 0) {
        save_mean[idx] = mu;
        save_invstd[idx] = invstd;
    }
    if (lane < cpg) {
        const int64_t NC = (int64_t)g.nseg * g.C;
        const int cc = grp * cpg + lane;
        const float sc = chan_scale ? chan_scale[(int64_t)n_img * g.C + cc] : 1.f;
        const float a = invstd * gamma[cc];
        ab[(int64_t)n_img * g.C + cc] = a * sc;
        ab[NC + (int64_t)n_img * g.C + cc] = (beta[cc] - mu * a) * sc;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_finalize2_kernel(const float* __restrict__ partial, ColGeom g, int G,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ chan_scale,
                                                               const float* __restrict__ invstd, float* __restrict__ kk,
                                                               float* __restrict__ chan_sums) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= g.nseg * G) return;
    const int n_img = idx / G, grp = idx - n_img * G;
    const int cpg = g.C / G;
    const int lanes_per_ch = 64 / cpg > 0 ? 64 / cpg : 1;
    const int j = lane % cpg, cl = lane / cpg;
    const bool act = cl < lanes_per_ch;
    const int c = grp * cpg + j;
    double s1 = 0.0, s2 = 0.0;
    if (act)
        for (int k = cl; k < g.nchunk; k += lanes_per_ch) {
            const float* p = partial + ((int64_t)n_img * g.nchunk + k) * 2 * g.C;
            s1 += (double)p[c];
            s2 += (double)p[g.C + c];
        }
    // per-channel totals: sum over the chunk lanes (lanes with equal j), kept in every lane of that channel
    for (int o = cpg; o < 64; o <<= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    const double sc = chan_scale ? (double)chan_scale[(int64_t)n_img * g.C + c] : 1.0;
    s1 *= sc;
    s2 *= sc;
    const int64_t NC = (int64_t)g.nseg * g.C;
    if (lane < cpg) {
        chan_sums[(int64_t)n_img * g.C + c] = (float)s1;
        chan_sums[NC + (int64_t)n_img * g.C + c] = (float)s2;
    }
    double c1 = lane < cpg ? (double)gamma[c] * s1 : 0.0;
    double c2 = lane < cpg ? (double)gamma[c] * s2 : 0.0;
    c1 = wave_sum_f64(c1);
    c2 = wave_sum_f64(c2);
    const double cnt = (double)g.rows_per_seg * cpg;
    const float is = invstd[idx];
    if (lane < cpg) {
        kk[(int64_t)n_img * g.C + c] = (float)sc * gamma[c] * is;
        kk[NC + (int64_t)n_img * g.C + c] = (float)(is * (c1 / cnt));
        kk[2 * NC + (int64_t)n_img * g.C + c] = (float)(is * (c2 / cnt));
    }
}

static ColGeom make_geom(int64_t rows_per_seg, int64_t nseg, int64_t C) {
    ColGeom g;
    g.rows_per_seg = rows_per_seg;
    g.nseg = (int)nseg;
    g.C = (int)C;
    // ~512 blocks in total (2 per CU, 8 row loads in flight per thread), at least 64 rows per chunk
    int64_t want = ceil_div(512, nseg);
    int64_t chunk = ceil_div(rows_per_seg, want);
    if (chunk < 64) chunk = 64;
    if (chunk > rows_per_seg) chunk = rows_per_seg;
    g.chunk_rows = (int)chunk;
    g.nchunk = (int)ceil_div(rows_per_seg, chunk);
    return g;
}

static unsigned ew_blocks(int64_t work) {
    int64_t b = ceil_div(work, 256);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int check_norm(const char* who, int64_t C, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
    DIGA_REQUIRE(C > 0 && C % 4 == 0, DIGA_EINVAL, "%s: C=%lld must be a positive multiple of 4", who, (long long)C);
    for (int64_t ld : lds) DIGA_REQUIRE(ld >= C && ld % 4 == 0, DIGA_EINVAL, "%s: leading dimension %lld invalid", who, (long long)ld);
    for (const void* p : ptrs) DIGA_REQUIRE(p == nullptr || aligned16(p), DIGA_EALIGN, "%s: pointers must be 16-byte aligned", who);
    return DIGA_OK;
}

// ---- the two tiny dense layers of the SE block (G5/model/seg_model_noaux.py:122-137: Linear(1280, 80) -> ReLU -> Linear(80, 1280) ->
// Sigmoid on the [N, 1280] pooled vector).  One wave per output element, lanes over the reduction; the backward pass as three
// small kernels (dz = dy * act'(y) and the bias gradient; dW; dx).  N is the batch (16): nothing here is worth a GEMM library call.
__device__ __forceinline__ float act_apply(float v, int act) { return act == 1 ? fmaxf(v, 0.f) : act == 2 ? 1.f / (1.f + expf(-v)) : v; }

__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, float* __restrict__ y, int N, int K, int O,
                                                               int act) {
    const int wave = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= N * O) return;
    const int n = wave / O, o = wave - n * O;
    const float* xr = x + (int64_t)n * K;
    const float* wr = w + (int64_t)o * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = __builtin_fmaf(xr[k], wr[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) y[(int64_t)n * O + o] = act_apply(acc + (b != nullptr ? b[o] : 0.f), act);
}

// dz[n][o] = dy[n][o] * act'(y[n][o]); db[o] = sum_n dz[n][o]  (one thread per output column o, N rows in sequence)
__global__ __launch_bounds__(256) void small_linear_dz_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              float* __restrict__ dz, float* __restrict__ db, int N, int O, int act) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) {
        const float yv = y[(int64_t)n * O + o], g = dy[(int64_t)n * O + o];
        const float d = act == 1 ? (yv > 0.f ? g : 0.f) : act == 2 ? g * yv * (1.f - yv) : g;
        dz[(int64_t)n * O + o] = d;
        s += d;
    }
    if (db != nullptr) db[o] = s;
}

// dW[o][k] = sum_n dz[n][o] * x[n][k]
__global__ __launch_bounds__(256) void small_linear_dw_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                              float* __restrict__ dw, int N, int K, int O) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)O * K) return;
    const int o = (int)(idx / K), k = (int)(idx - (int64_t)o * K);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = __builtin_fmaf(dz[(int64_t)n * O + o], x[(int64_t)n * K + k], s);
    dw[idx] = s;
}

// dx[n][k] = sum_o dz[n][o] * W[o][k]
__global__ __launch_bounds__(256) void small_linear_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                              float* __restrict__ dx, int N, int K, int O) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (int64_t)n * K);
    float s = 0.f;
    for (int o = 0; o < O; ++o) s = __builtin_fmaf(dz[(int64_t)n * O + o], w[(int64_t)o * K + k], s);
    dx[idx] = s;
}

}  // namespace diga

using namespace diga;

// workspace: partials (3*C floats per (segment, chunk)) + coefficient arrays (up to 3*nseg*C)
extern "C" size_t diga_norm_workspace_bytes(int64_t rows_per_seg, int64_t nseg, int64_t C) {
    if (rows_per_seg <= 0 || nseg <= 0 || C <= 0) return 0;
    const ColGeom g = make_geom(rows_per_seg, nseg, C);
    return ((size_t)nseg * g.nchunk * 3 * C + (size_t)5 * nseg * C + 64) * sizeof(float);
}

extern "C" int diga_bn_fwd(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* residual, int64_t ld_r,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float* save_mean, float* save_invstd, float* save_ab, int64_t M, int64_t C, int training,
                           int relu, int y_twin, unsigned char* relu_bits, float momentum, float eps, void* workspace,
                           size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(!relu_bits || (relu && C % 32 == 0), DIGA_EINVAL, "bn_fwd: relu_bits needs relu and C % 32 == 0");
    DIGA_REQUIRE(!y_twin || (C % 8 == 0 && ld_y == C), DIGA_EINVAL, "bn_fwd: twin output needs C % 8 == 0 and a dense y");
    DIGA_REQUIRE(x && gamma && beta && save_mean && save_invstd && workspace && M > 0, DIGA_EINVAL, "bn_fwd: bad argument");
    DIGA_REQUIRE(y || (save_ab && !y_twin), DIGA_EINVAL, "bn_fwd: y = null (statistics and coefficients only; residual / relu_bits are then the consumer's) needs save_ab");
    DIGA_REQUIRE(training || (running_mean && running_var), DIGA_EINVAL, "bn_fwd: eval mode needs running statistics");
    int rc = check_norm("bn_fwd", C, {ld_x, y ? ld_y : C, residual ? ld_r : C}, {x, y ? (const void*)y : (const void*)x, residual});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= diga_norm_workspace_bytes(M, 1, C), DIGA_EWORKSPACE, "bn_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes per element: statistics pass reads x (4), apply reads x (4) [+ residual (4)], writes y (4)
    ProfScope prof(DIGA_PROF_NORM, st, (double)M * C * (8.0 + (training ? 4.0 : 0.0) + (residual ? 4.0 : 0.0)));
    const ColGeom g = make_geom(M, 1, C);
    float* partial = (float*)workspace;
    float* ab = save_ab != nullptr ? save_ab : partial + (size_t)g.nchunk * 3 * C;     // [2][C] y = fma(x, a, b)
    if (training) {
        hipLaunchKernelGGL(colstats_partial_kernel, dim3(g.nchunk, 1), dim3(kNormThreads), 0, st, x, ld_x, g, partial);
        hipLaunchKernelGGL(bn_finalize2_kernel, dim3((unsigned)ceil_div(C, kFinCh)), dim3(256), 0, st, partial, g, gamma, beta,
                           running_mean, running_var, save_mean, save_invstd, ab, momentum, eps);
    } else {
        hipLaunchKernelGGL(bn_eval_ab_kernel, dim3((unsigned)ceil_div(C, 128)), dim3(128), 0, st, gamma, beta, running_mean,
                           running_var, save_mean, save_invstd, ab, (int)C, eps);
    }
    if (y != nullptr)
        launch_affine(dim3(ew_blocks(M * C / 4)), st, x, ld_x, y, ld_y, residual, ld_r, ab,
                           ab + C, (int64_t)0, M, M, (int)C, relu, y_twin, relu_bits);
    return launch_status("diga_bn_fwd");
}

static int bn_fwd_from_partials(const char* who, const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* residual,
                                int64_t ld_r, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float* save_mean, float* save_invstd, float* save_ab, int64_t M,
                                int64_t C, int relu, int y_twin, unsigned char* relu_bits, float momentum, float eps,
                                const float* partial, int64_t chunk_rows, const float* counts, int64_t n_records,
                                void* workspace, size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(!relu_bits || (relu && C % 32 == 0), DIGA_EINVAL, "%s: relu_bits needs relu and C %% 32 == 0", who);
    DIGA_REQUIRE(x && gamma && beta && save_mean && save_invstd && partial && workspace && M > 0, DIGA_EINVAL, "%s: bad argument", who);
    DIGA_REQUIRE(counts ? (n_records > 0 && n_records < (1 << 30)) : chunk_rows > 0, DIGA_EINVAL, "%s: bad chunk_rows / record count", who);
    // y == nullptr: statistics and coefficients only (save_ab required) -- the consumer applies relu(fma(x, a, b)) on load
    DIGA_REQUIRE(y || (save_ab && !y_twin), DIGA_EINVAL, "%s: y = null (statistics and coefficients only) needs save_ab", who);
    int rc = check_norm(who, C, {ld_x, y ? ld_y : C, residual ? ld_r : C}, {x, y ? (const void*)y : (const void*)x, residual});
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_NORM, st, (double)M * C * (8.0 + (residual ? 4.0 : 0.0)));   // read x [+ residual], write y
    ColGeom g;
    g.rows_per_seg = M;
    g.nseg = 1;
    g.C = (int)C;
    g.chunk_rows = counts ? 1 : (int)chunk_rows;
    g.nchunk = counts ? (int)n_records : (int)ceil_div(M, chunk_rows);
    // the finaliser walks all chunks of a channel with 16 lanes: beyond a few hundred chunks fold them first
    // (fold to <= 96 groups: the finaliser's 16 lanes per channel then walk 6 partials each instead of 24+;
    //  merge + finalise of a 1176-chunk layer: 5.9 + 16.4 us before)
    const int group = g.nchunk > 128 ? (int)ceil_div(g.nchunk, 96) : 1;
    const int ngroup = (int)ceil_div(g.nchunk, group);
    const size_t need = ((group > 1 ? (size_t)ngroup * 3 * C + (counts ? ngroup : 0) : 0) + (size_t)2 * C) * sizeof(float);
    DIGA_REQUIRE(workspace_bytes >= need, DIGA_EWORKSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, need);
    float* ab = save_ab != nullptr ? save_ab : (float*)workspace;
    if (group > 1) {
        float* merged = (float*)workspace + 2 * C;
        float* mcounts = counts ? merged + (size_t)ngroup * 3 * C : nullptr;
        hipLaunchKernelGGL(merge_partials_kernel, dim3(ngroup, (unsigned)ceil_div(C, 64)), dim3(256), 0, st, partial, g, group,
                           merged, counts, mcounts);
        partial = merged;
        counts = mcounts;
        g.chunk_rows = counts ? 1 : (int)(chunk_rows * group);
        g.nchunk = ngroup;
    }
    hipLaunchKernelGGL(bn_finalize2_kernel, dim3((unsigned)ceil_div(C, kFinCh)), dim3(256), 0, st, partial, g, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, ab, momentum, eps, counts);
    if (y != nullptr)
        launch_affine(dim3(ew_blocks(M * C / 4)), st, x, ld_x, y, ld_y, residual, ld_r, ab,
                           ab + C, (int64_t)0, M, M, (int)C, relu, y_twin, relu_bits);
    return launch_status(who);
}

extern "C" int diga_bn_fwd_partials(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* residual,
                                    int64_t ld_r, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float* save_mean, float* save_invstd, float* save_ab, int64_t M,
                                    int64_t C, int relu, int y_twin, unsigned char* relu_bits, float momentum, float eps,
                                    const float* partial, int64_t chunk_rows,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return bn_fwd_from_partials("diga_bn_fwd_partials", x, ld_x, y, ld_y, residual, ld_r, gamma, beta, running_mean, running_var, save_mean,
                                save_invstd, save_ab, M, C, relu, y_twin, relu_bits, momentum, eps, partial, chunk_rows, nullptr, 0,
                                workspace, workspace_bytes, stream);
}

extern "C" int diga_bn_fwd_records(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* residual,
                                   int64_t ld_r, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* save_mean, float* save_invstd, float* save_ab, int64_t M,
                                   int64_t C, int relu, int y_twin, unsigned char* relu_bits, float momentum, float eps,
                                   const float* partial, const float* counts, int64_t n_records,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(counts != nullptr, DIGA_EINVAL, "diga_bn_fwd_records: null counts");
    return bn_fwd_from_partials("diga_bn_fwd_records", x, ld_x, y, ld_y, residual, ld_r, gamma, beta, running_mean, running_var, save_mean,
                                save_invstd, save_ab, M, C, relu, y_twin, relu_bits, momentum, eps, partial, 0, counts, n_records,
                                workspace, workspace_bytes, stream);
}

static int bn_bwd_impl(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                           const float* relu_ab, const float* gamma, const float* save_mean, const float* save_invstd,
                           float* dx, int64_t ld_dx,
                           float* dres, int64_t ld_dr, int64_t M, int64_t C, int training, int dx_twin, void* workspace,
                           size_t workspace_bytes, void* stream, float* dgamma, float* dbeta) {
    DIGA_REQUIRE(!dx_twin || (C % 8 == 0 && ld_dx == C), DIGA_EINVAL, "bn_bwd: twin output needs C % 8 == 0 and a dense dx");
    DIGA_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && workspace && M > 0, DIGA_EINVAL, "bn_bwd: bad argument");
    int rc = check_norm("bn_bwd", C, {ld_dy, ld_x, y ? ld_y : C, ld_dx, dres ? ld_dr : C}, {dy, x, y, dx, dres});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= diga_norm_workspace_bytes(M, 1, C), DIGA_EWORKSPACE, "bn_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // reduce pass reads dy, x [, y] (training only); apply pass reads dy, x [, y], writes dx [, dres]
    ProfScope prof(DIGA_PROF_NORM, st,
                   (double)M * C * ((training ? 8.0 + (y ? 4.0 : 0.0) : 0.0) + 12.0 + (y ? 4.0 : 0.0) + (dres ? 4.0 : 0.0)));
    const ColGeom g = make_geom(M, 1, C);
    float* partial = (float*)workspace;
    float* kk = partial + (size_t)g.nchunk * 3 * C;
    DIGA_REQUIRE(!(y && relu_ab), DIGA_EINVAL, "bn_bwd: pass y or relu_ab, not both");
    if (training || dgamma)
        hipLaunchKernelGGL(bwd_partial_kernel, dim3(g.nchunk, 1), dim3(kNormThreads), 0, st, dy, ld_dy, x, ld_x, y, ld_y,
                           save_mean, save_invstd, 0, 1, g, partial, relu_ab);
    hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3((unsigned)ceil_div(C, kFinCh)), dim3(256), 0, st, partial, g, gamma,
                       save_invstd, kk, training, dgamma, dbeta);
    launch_bwd_apply(dim3(ew_blocks(M * C / 4)), st, dy, ld_dy, x, ld_x, y, ld_y, save_mean,
                       save_invstd, 0, 1, kk, (int64_t)0, (int64_t)C, dx, ld_dx, dres, ld_dr, M, M, (int)C, relu_ab, dx_twin);
    return launch_status(dgamma ? "diga_bn_bwd_affine" : "diga_bn_bwd");
}

extern "C" int diga_bn_bwd(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                           const float* relu_ab, const float* gamma, const float* save_mean, const float* save_invstd,
                           float* dx, int64_t ld_dx,
                           float* dres, int64_t ld_dr, int64_t M, int64_t C, int training, int dx_twin, void* workspace,
                           size_t workspace_bytes, void* stream) {
    return bn_bwd_impl(dy, ld_dy, x, ld_x, y, ld_y, relu_ab, gamma, save_mean, save_invstd, dx, ld_dx, dres, ld_dr, M, C, training,
                       dx_twin, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int diga_bn_bwd_affine(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                                  const float* relu_ab, const float* gamma, const float* save_mean, const float* save_invstd,
                                  float* dx, int64_t ld_dx, float* dgamma, float* dbeta, int64_t M, int64_t C, int training,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(dgamma && dbeta, DIGA_EINVAL, "bn_bwd_affine: dgamma and dbeta are required");
    return bn_bwd_impl(dy, ld_dy, x, ld_x, y, ld_y, relu_ab, gamma, save_mean, save_invstd, dx, ld_dx, nullptr, 0, M, C, training ? 1 : 0, 0,
                       workspace, workspace_bytes, stream, dgamma, dbeta);
}

extern "C" int diga_bn_bwd_partials(const float* g, int64_t ld_g, const float* x, int64_t ld_x, const float* gamma,
                                    const float* save_mean, const float* save_invstd, float* dx, int64_t ld_dx, int64_t M,
                                    int64_t C, int dx_twin, const float* partial, int64_t chunk_rows, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(!dx_twin || (C % 8 == 0 && ld_dx == C), DIGA_EINVAL, "bn_bwd_partials: twin output needs C % 8 == 0 and a dense dx");
    DIGA_REQUIRE(g && x && gamma && save_mean && save_invstd && dx && partial && workspace && M > 0 && chunk_rows > 0, DIGA_EINVAL,
                 "bn_bwd_partials: bad argument");
    int rc = check_norm("bn_bwd_partials", C, {ld_g, ld_x, ld_dx}, {g, x, dx});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= (size_t)(3 + 64) * C * sizeof(float), DIGA_EWORKSPACE, "bn_bwd_partials: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_NORM, st, (double)M * C * 12.0);        // apply pass only: g, x in, dx out
    ColGeom geo;
    geo.rows_per_seg = M;
    geo.nseg = 1;
    geo.C = (int)C;
    geo.chunk_rows = (int)chunk_rows;
    geo.nchunk = (int)ceil_div(M, chunk_rows);
    float* kk = (float*)workspace;
    if (geo.nchunk > 64) {
        // many 128-row chunks: fold them to 32 rows first (the finaliser only needs sum g, sum g*xhat over all rows)
        constexpr int kParts = 32;
        float* folded = kk + 3 * C;
        hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)ceil_div(2 * C, 64), kParts), dim3(256), 0, st, partial, geo.nchunk,
                           (int)(2 * C), kParts, folded);
        partial = folded;
        geo.nchunk = kParts;
        geo.chunk_rows = (int)ceil_div(M, kParts);       // (only nseg * nchunk and rows_per_seg enter the finaliser)
    }
    hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3((unsigned)ceil_div(C, kFinCh)), dim3(256), 0, st, partial, geo, gamma,
                       save_invstd, kk, 1);
    launch_bwd_apply(dim3(ew_blocks(M * C / 4)), st, g, ld_g, x, ld_x, (const float*)nullptr,
                       (int64_t)0, save_mean, save_invstd, 0, 1, kk, (int64_t)0, (int64_t)C, dx, ld_dx, (float*)nullptr,
                       (int64_t)0, M, M, (int)C, (const float*)nullptr, dx_twin);
    return launch_status("diga_bn_bwd_partials");
}

extern "C" int diga_gn_fwd(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* gamma, const float* beta,
                           const float* chan_scale, float* save_mean, float* save_invstd, int64_t N, int64_t HW, int64_t C,
                           int64_t G, int relu, float eps, void* workspace, size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(x && y && gamma && beta && save_mean && save_invstd && workspace && N > 0 && HW > 0, DIGA_EINVAL, "gn_fwd: bad argument");
    DIGA_REQUIRE(G > 0 && C % G == 0, DIGA_EINVAL, "gn_fwd: C must be divisible by the group count");
    int rc = check_norm("gn_fwd", C, {ld_x, ld_y}, {x, y});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= diga_norm_workspace_bytes(HW, N, C), DIGA_EWORKSPACE, "gn_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_NORM, st, (double)N * HW * C * 12.0);      // statistics read, apply read + write
    const ColGeom g = make_geom(HW, N, C);
    float* partial = (float*)workspace;
    float* ab = partial + (size_t)N * g.nchunk * 3 * C;
    hipLaunchKernelGGL(colstats_partial_kernel, dim3(g.nchunk, (unsigned)N), dim3(kNormThreads), 0, st, x, ld_x, g, partial);
    const int cpg_f = (int)(C / G);
    if (cpg_f <= 64 && (cpg_f & (cpg_f - 1)) == 0)
        hipLaunchKernelGGL(gn_finalize2_kernel, dim3((unsigned)ceil_div(N * G, 4)), dim3(256), 0, st, partial, g, (int)G, gamma,
                           beta, chan_scale, save_mean, save_invstd, ab, eps);
    else
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)ceil_div(N * G, 64)), dim3(64), 0, st, partial, g, (int)G, gamma,
                           beta, chan_scale, save_mean, save_invstd, ab, eps);
    launch_affine(dim3(ew_blocks(N * HW * C / 4)), st, x, ld_x, y, ld_y,
                       (const float*)nullptr, (int64_t)0, ab, ab + N * C, C, HW, N * HW, (int)C, relu, 0);
    return launch_status("diga_gn_fwd");
}

extern "C" int diga_gn_bwd(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, const float* y, int64_t ld_y,
                           const float* gamma, const float* chan_scale, const float* save_mean, const float* save_invstd,
                           float* dx, int64_t ld_dx, float* dgamma, float* dbeta, int64_t N, int64_t HW, int64_t C,
                           int64_t G, void* workspace, size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace && N > 0 && HW > 0,
                 DIGA_EINVAL, "gn_bwd: bad argument");
    DIGA_REQUIRE(G > 0 && C % G == 0, DIGA_EINVAL, "gn_bwd: C must be divisible by the group count");
    int rc = check_norm("gn_bwd", C, {ld_dy, ld_x, y ? ld_y : C, ld_dx}, {dy, x, y, dx});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= diga_norm_workspace_bytes(HW, N, C), DIGA_EWORKSPACE, "gn_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_NORM, st, (double)N * HW * C * (20.0 + (y ? 8.0 : 0.0)));   // reduce: dy, x [, y]; apply: dy, x [, y], dx
    const ColGeom g = make_geom(HW, N, C);
    const int cpg = (int)(C / G);
    float* partial = (float*)workspace;
    float* kk = partial + (size_t)N * g.nchunk * 3 * C;       // [3][N][C]
    float* chan_sums = kk + (size_t)3 * N * C;                // [2][N][C]
    hipLaunchKernelGGL(bwd_partial_kernel, dim3(g.nchunk, (unsigned)N), dim3(kNormThreads), 0, st, dy, ld_dy, x, ld_x, y, ld_y,
                       save_mean, save_invstd, (int)G, cpg, g, partial, (const float*)nullptr);
    if (cpg <= 64 && (cpg & (cpg - 1)) == 0)
        hipLaunchKernelGGL(gn_bwd_finalize2_kernel, dim3((unsigned)ceil_div(N * G, 4)), dim3(256), 0, st, partial, g, (int)G,
                           gamma, chan_scale, save_invstd, kk, chan_sums);
    else
        hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((unsigned)ceil_div(N * G, 64)), dim3(64), 0, st, partial, g, (int)G,
                           gamma, chan_scale, save_invstd, kk, chan_sums);
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3((unsigned)ceil_div(C, 128)), dim3(128), 0, st, chan_sums, (int)N, (int)C,
                       dgamma, dbeta);
    launch_bwd_apply(dim3(ew_blocks(N * HW * C / 4)), st, dy, ld_dy, x, ld_x, y, ld_y,
                       save_mean, save_invstd, (int)G, cpg, kk, C, N * C, dx, ld_dx, (float*)nullptr, (int64_t)0, HW, N * HW,
                       (int)C, (const float*)nullptr, 0);
    return launch_status("diga_gn_bwd");
}

extern "C" int diga_avgpool_nhwc(const float* x, int64_t ld_x, float* out, int64_t N, int64_t HW, int64_t C, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(x && out && workspace && N > 0 && HW > 0, DIGA_EINVAL, "avgpool: bad argument");
    int rc = check_norm("avgpool", C, {ld_x}, {x});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= diga_norm_workspace_bytes(HW, N, C), DIGA_EWORKSPACE, "avgpool: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_NORM, st, (double)N * HW * C * 4.0);
    const ColGeom g = make_geom(HW, N, C);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(colstats_partial_kernel, dim3(g.nchunk, (unsigned)N), dim3(kNormThreads), 0, st, x, ld_x, g, partial);
    hipLaunchKernelGGL(seg_mean_kernel, dim3((unsigned)ceil_div(N * C, 256)), dim3(256), 0, st, partial, g, out);
    return launch_status("diga_avgpool_nhwc");
}

// Column sums of a NARROW matrix (C <= 64, any C: the 19-class prediction conv's bias gradient, [M][19] with M = 16 x 192 x 192 for
// the SegFormer head -- torch's column reduce of that shape took 2.3 ms): CP = 32 or 64 column lanes x 256 / CP row lanes per block,
// eight rows in flight per thread, partial sums per block, merged in double by one small block.
template <int CP>
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int C, int64_t chunk_rows,
                                                            float* __restrict__ partial) {
    constexpr int RL = 256 / CP;
    __shared__ float red[RL][CP];
    const int tc = threadIdx.x % CP, tr = threadIdx.x / CP;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = r0 + chunk_rows < M ? r0 + chunk_rows : M;
    float acc = 0.f;
    if (tc < C) {
        int64_t r = r0 + tr;
        for (; r + 7 * RL < r1; r += 8 * RL) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(r + u * RL) * ld + tc];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < r1; r += RL) acc += x[r * ld + tc];
    }
    red[tr][tc] = acc;
    __syncthreads();
    if (tr == 0 && tc < C) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < RL; ++l) s += red[l][tc];
        partial[(int64_t)blockIdx.x * C + tc] = s;
    }
}

__global__ __launch_bounds__(64) void colsum_narrow_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
    const int c = threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += (double)partial[(int64_t)k * C + c];
    out[c] = (float)s;
}

extern "C" int diga_colsum_nhwc(const float* x, int64_t ld_x, float* out, int64_t M, int64_t C, void* workspace, size_t workspace_bytes,
                                void* stream) {
    DIGA_REQUIRE(x && out && workspace && M > 0, DIGA_EINVAL, "colsum: bad argument");
    if (C > 0 && C <= 64 && (C % 4 != 0 || ld_x % 4 != 0)) {
        DIGA_REQUIRE(ld_x >= C, DIGA_EINVAL, "colsum: leading dimension %lld invalid", (long long)ld_x);
        int64_t nblk = ceil_div(M, 256);
        if (nblk > 512) nblk = 512;
        const int64_t chunk = ceil_div(M, nblk);
        nblk = ceil_div(M, chunk);
        DIGA_REQUIRE(workspace_bytes >= (size_t)nblk * C * sizeof(float), DIGA_EWORKSPACE, "colsum: workspace too small");
        hipStream_t st = (hipStream_t)stream;
        ProfScope prof(DIGA_PROF_NORM, st, (double)M * C * 4.0);
        float* partial = (float*)workspace;
        if (C <= 32)
            hipLaunchKernelGGL(colsum_narrow_kernel<32>, dim3((unsigned)nblk), dim3(256), 0, st, x, ld_x, M, (int)C, chunk, partial);
        else
            hipLaunchKernelGGL(colsum_narrow_kernel<64>, dim3((unsigned)nblk), dim3(256), 0, st, x, ld_x, M, (int)C, chunk, partial);
        hipLaunchKernelGGL(colsum_narrow_final_kernel, dim3(1), dim3(64), 0, st, partial, (int)nblk, (int)C, out);
        return launch_status("diga_colsum_nhwc");
    }
    int rc = check_norm("colsum", C, {ld_x}, {x});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= diga_norm_workspace_bytes(M, 1, C), DIGA_EWORKSPACE, "colsum: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_NORM, st, (double)M * C * 4.0);
    const ColGeom g = make_geom(M, 1, C);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(colstats_partial_kernel, dim3(g.nchunk, 1u), dim3(kNormThreads), 0, st, x, ld_x, g, partial);
    hipLaunchKernelGGL(seg_sum_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, st, partial, g, out);
    return launch_status("diga_colsum_nhwc");
}

extern "C" int diga_channel_affine(const float* x, int64_t ld_x, float* y, int64_t ld_y, const float* a, const float* b,
                                   int64_t N, int64_t HW, int64_t C, void* stream) {
    DIGA_REQUIRE(x && y && a && N > 0 && HW > 0, DIGA_EINVAL, "channel_affine: bad argument");
    int rc = check_norm("channel_affine", C, {ld_x, ld_y}, {x, y, a, b});
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)N * HW * C * 8.0);
    launch_affine(dim3(ew_blocks(N * HW * C / 4)), st, x, ld_x, y, ld_y,
                       (const float*)nullptr, (int64_t)0, a, b, C, HW, N * HW, (int)C, 0, 0);
    return launch_status("diga_channel_affine");
}

extern "C" int diga_channel_dot(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, float* out, int64_t N,
                                int64_t HW, int64_t C, void* workspace, size_t workspace_bytes, void* stream) {
    DIGA_REQUIRE(dy && x && out && workspace && N > 0 && HW > 0, DIGA_EINVAL, "channel_dot: bad argument");
    int rc = check_norm("channel_dot", C, {ld_dy, ld_x}, {dy, x});
    if (rc) return rc;
    DIGA_REQUIRE(workspace_bytes >= diga_norm_workspace_bytes(HW, N, C), DIGA_EWORKSPACE, "channel_dot: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_NORM, st, (double)N * HW * C * 8.0);
    const ColGeom g = make_geom(HW, N, C);
    float* partial = (float*)workspace;
    // sum g * x: the backward-partial kernel with null statistics (= mean 0, invstd 1).  (Round 2 copied the pair {0, 1} from
    // the host stack with hipMemcpyAsync: a hidden host synchronisation, and under stream capture a dangling host pointer.)
    hipLaunchKernelGGL(bwd_partial_kernel, dim3(g.nchunk, (unsigned)N), dim3(kNormThreads), 0, st, dy, ld_dy, x, ld_x,
                       (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr, 0, (int)C, g, partial,
                       (const float*)nullptr);
    hipLaunchKernelGGL(seg_dot_kernel, dim3((unsigned)ceil_div(N * C, 256)), dim3(256), 0, st, partial, g, out);
    return launch_status("diga_channel_dot");
}


extern "C" int diga_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int64_t N, int64_t K, int64_t O, int act,
                                     void* stream) {
    DIGA_REQUIRE(x && w && y && N > 0 && K > 0 && O > 0 && act >= 0 && act <= 2 && N * O < (1ll << 24), DIGA_EINVAL,
                 "small_linear_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)(O * K + N * K + N * O) * 4.0);
    hipLaunchKernelGGL(small_linear_fwd_kernel, dim3((unsigned)ceil_div(N * O * 64, 256)), dim3(256), 0, st, x, w, b, y, (int)N, (int)K,
                       (int)O, act);
    return launch_status("diga_small_linear_fwd");
}

extern "C" int diga_small_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dz, float* dx, float* dw,
                                     float* db, int64_t N, int64_t K, int64_t O, int act, void* stream) {
    DIGA_REQUIRE(x && w && y && dy && dz && N > 0 && K > 0 && O > 0 && act >= 0 && act <= 2, DIGA_EINVAL, "small_linear_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)(2 * O * K + 2 * N * K + 3 * N * O) * 4.0);
    hipLaunchKernelGGL(small_linear_dz_kernel, dim3((unsigned)ceil_div(O, 256)), dim3(256), 0, st, dy, y, dz, db, (int)N, (int)O, act);
    if (dw != nullptr)
        hipLaunchKernelGGL(small_linear_dw_kernel, dim3((unsigned)ceil_div(O * K, 256)), dim3(256), 0, st, (const float*)dz, x, dw, (int)N,
                           (int)K, (int)O);
    if (dx != nullptr)
        hipLaunchKernelGGL(small_linear_dx_kernel, dim3((unsigned)ceil_div(N * K, 256)), dim3(256), 0, st, (const float*)dz, w, dx, (int)N,
                           (int)K, (int)O);
    return launch_status("diga_small_linear_bwd");
}

extern "C" int diga_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int64_t N, int64_t H, int64_t W, int64_t C,
                                     int64_t Ho, int64_t Wo, void* stream) {
    DIGA_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, DIGA_EINVAL, "maxpool_fwd: bad argument");
    int rc = check_norm("maxpool_fwd", C, {}, {x, y});
    if (rc) return rc;
    DIGA_REQUIRE((Ho - 1) * 2 - 1 < H && (Wo - 1) * 2 - 1 < W, DIGA_EINVAL, "maxpool_fwd: last window starts outside the input");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)N * C * (H * W * 4.0 + Ho * Wo * 5.0));
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(N * Ho * Wo * C / 4)), dim3(256), 0, st, x, y, idx, (int)N, (int)H,
                       (int)W, (int)C, (int)Ho, (int)Wo);
    return launch_status("diga_maxpool3x3s2_fwd");
}

extern "C" int diga_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t N, int64_t H, int64_t W,
                                     int64_t C, int64_t Ho, int64_t Wo, void* stream) {
    DIGA_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, DIGA_EINVAL, "maxpool_bwd: bad argument");
    int rc = check_norm("maxpool_bwd", C, {}, {dy, dx});
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, st, (double)N * C * (H * W * 4.0 + Ho * Wo * 5.0));
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks(N * H * W * C / 4)), dim3(256), 0, st, dy, idx, dx, (int)N, (int)H,
                       (int)W, (int)C, (int)Ho, (int)Wo);
    return launch_status("diga_maxpool3x3s2_bwd");
}
