// Shared device helpers of the MiT / SegFormer kernels (mit_*.hip): fp16 storage, fp32 accumulation, gfx950 only.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"
#include "../../include/diga_mit.h"

namespace diga {
namespace mit {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));


// blocks b and b+8 share an XCD (round-robin dispatch): give every XCD a contiguous range of tiles (bijective)
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// 64-byte LDS rows (32 halves = one K-step): the four 16-byte k-slots of row r sit at slot ^ swz64(r); conflict-free for
// the ds_read_b128 fragment reads of a 16-row tile (same function as the conv kernels' lds_swz)
__device__ __forceinline__ int swz64(int r) { return ((0x78 >> (((r >> 2) & 3) * 2)) & 3) ^ (((r >> 1) & 1) << 1); }

// 128-byte LDS rows (64 halves, attention K / V / Q / dO images): chunk c (16 B) of row r sits at c ^ swz128(r).
// k = (r >> 1) & 7 -> v = ((k & 3) << 1) | (k >> 2): a permutation of 0..7 over 8 rows of one parity (ds_read_b128 of a
// 16-row tile touches 16 distinct 16-byte slots of the 256-byte bank row) whose upper two bits are distinct over 4
// consecutive row pairs (the transposing read of 8 rows x 32 bytes touches 8 distinct 32-byte positions).
__device__ __forceinline__ int swz128(int r) {
    const int k = (r >> 1) & 7;
    return ((k & 3) << 1) | (k >> 2);
}

// 256-byte LDS rows read with the transposing ds_read_b64_tr_b16 (weight-gradient GEMM): chunk ^ (tr_key(r) << 1)
__device__ __forceinline__ int tr_key(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__device__ __forceinline__ f16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// GELU(u) = u * Phi(u) and its derivative Phi(u) + u * phi(u), Phi = 0.5 (1 + erf(u / sqrt 2)).  erf by Abramowitz & Stegun
// 7.1.26 (|error| <= 1.5e-7, far inside the fp16 storage of both results): one v_rcp, one v_exp and five fmas; the same
// exponential exp(-u^2 / 2) serves erf and the density phi.  (erff() costs ~25 VALU instructions per element: it was the
// largest single cost of the depthwise-conv kernels.)
__device__ __forceinline__ void gelu_terms(float u, float& cdf, float& pdf) {
    const float x = fabsf(u) * 0.70710678118654752f;
    // (__builtin_amdgcn_rcpf: the bare v_rcp_f32, 1 ulp; __frcp_rn is the correctly rounded reciprocal = the IEEE division sequence,
    //  div_scale x 2 + rcp + 4 fma + div_fmas + div_fixup per element -- a third of the depthwise forward's VALU instructions)
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * x);
    const float e = __expf(-x * x);                                  // = exp(-u^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.f - poly * e;
    cdf = 0.5f * (1.f + copysignf(erf_abs, u));
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float u) {
    float cdf, pdf;
    gelu_terms(u, cdf, pdf);
    return u * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float u) {
    float cdf, pdf;
    gelu_terms(u, cdf, pdf);
    return cdf + u * pdf;
}

// out = scale * sum over chunks of partial[chunk][i] (+ existing when accumulate), i < n, fixed order: 32 columns x 8 chunk
// phases per block (phase p adds chunks p, p + 8, ... sequentially, then the phases are added 0..7).
//   MAP 0: out0[i]      MAP 1 (LayerNorm): i < C -> out0[i] (dgamma), else out1[i - C] (dbeta)
//   MAP 2 (depthwise 3x3): t = i / C, c = i % C: t < 9 -> out0[c * 9 + t] (dw [C][9]), t == 9 -> out1[c] (db)
//   COLS x (256 / COLS) phases per block: 32 x 8 for the weight-sized sums; 8 x 32 where n is a few hundred values and the chunks are
//   hundreds (LayerNorm dgamma / dbeta of a 300 000-row token matrix: 20 blocks of 144-deep chains otherwise -- 18 us of latency).
//   (partial2, n2 > 0): a SECOND sum of the same chunk count handled by the blocks behind the first one's (MAP 0 only: the bias slab of
//   a weight-gradient GEMM, written to out1) -- one launch instead of two.
template <int MAP, int COLS = 32>
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partial, int chunks, int n, float* __restrict__ out0,
                                                             float* __restrict__ out1, int C, float scale, int accumulate,
                                                             const float* __restrict__ partial2 = nullptr, int n2 = 0) {
    constexpr int PH = 256 / COLS;
    __shared__ float red[PH][COLS + 1];
    const int col = threadIdx.x % COLS, ph = threadIdx.x / COLS;
    int blk = blockIdx.x;
    if constexpr (MAP == 0) {
        const int nb0 = (n + COLS - 1) / COLS;
        if (blk >= nb0) {                                   // the second sum (block-uniform branch)
            blk -= nb0;
            partial = partial2;
            n = n2;
            out0 = out1;
        }
    }
    const int i = blk * COLS + col;
    float s = 0.f;
    if (i < n)
        for (int k = ph; k < chunks; k += PH) s += partial[(int64_t)k * n + i];
    red[ph][col] = s;
    __syncthreads();
    if (ph == 0 && i < n) {
        float tot = red[0][col];
#pragma unroll
        for (int p = 1; p < PH; ++p) tot += red[p][col];
        tot *= scale;
        float* dst;
        if constexpr (MAP == 0) {
            dst = out0 + i;
        } else if constexpr (MAP == 1) {
            dst = i < C ? out0 + i : out1 + (i - C);
        } else {
            const int t = i / C, c = i - t * C;
            dst = t < 9 ? out0 + (int64_t)c * 9 + t : out1 + c;
        }
        if (accumulate) tot += *dst;
        *dst = tot;
    }
}

}  // namespace mit
}  // namespace diga
