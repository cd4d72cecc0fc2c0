// Spatial-reduction attention of the MiT encoder (G5/model/networks/MixTransfomer.py:120-137): softmax(q k^T * scale) v with
// head_dim 64 and a short, spatially reduced key / value sequence (Nk = N / sr^2: 576 keys for every stage of a 768x768
// crop).  Flash-style: scores never leave registers.  fp16 operands, fp32 accumulation (v_mfma_f32_16x16x32_f16), fp32
// softmax in the log2 domain.
//
// Every product is formed TRANSPOSED so that the result of one MFMA is the B operand of the next without any lane
// movement: S^T = K Q^T has the key on the accumulator row (4 g + e, g = lane >> 4) and the query on the lane column, which
// is exactly the operand layout of  O^T = V^T P^T  once the k-slots (g, j) of that MFMA are read as
//   key(g, j) = 4 g + j (j < 4, first 16-key tile)   |   16 + 4 g + j - 4 (j >= 4, second tile)
// -- and V^T in that same key order is what the transposing ds_read_b64_tr_b16 delivers from a row-major [key][d] image
// (per 16-lane group a 4-row x 16-column block, lane i receiving column i).  Softmax reductions over keys run over a
// lane's 16 registers plus two cross-lane steps (lane ^ 16, lane ^ 32).
//
//   attn_fwd_kernel    O, LSE                 block = 8 waves x 32 queries, loop over 64-key blocks (K, V via LDS-DMA)
//   attn_bwd_dq_kernel dQ, delta = rowsum(dO*O)  same structure, recomputes P from LSE
//   attn_bwd_dkv_kernel dK, dV                block = 4 waves x 16 keys, loop over 64-query blocks (Q, dO via LDS-DMA),
//                                             query range split across blocks, fp32 slabs reduced in fixed order
#include "mit_common.h"

namespace diga {
namespace mit {

struct AttnArgs {
    const _Float16* Q;        // [B*N][ldq], head h at columns h*64..
    const _Float16* K;        // [B*Nk][ldkv]
    const _Float16* V;        // [B*Nk][ldkv]
    int64_t ldq, ldkv;
    _Float16* O;              // [B*N][ldo]           (forward: out; backward: saved forward output)
    int64_t ldo;
    float* lse;               // [B][heads][N], log2 domain
    const _Float16* dO;       // [B*N][ldo]
    _Float16* dQ;             // [B*N][ldq]
    float* delta;             // [B][heads][N]
    float* slab;              // dkv: [splits][B*Nk][2*heads*64] fp32
    int B, heads, N, Nk;
    int q_blocks;             // fwd / dq: blocks of 256 queries per (b, head)
    int k_blocks, splits, qblocks_per_split;   // dkv
    float scale, scale_log2e;
};

constexpr int kRowB = 128;                 // bytes per LDS row: 64 halves
constexpr int kImg = 64 * kRowB;           // one 64-row image: 8 KB

// LDS-DMA fill of a 64-row x 128-byte image from rows (base + row * ld), rows clamped to [0, nrows): wave w of 8 fills
// rows 8 w .. 8 w + 7 (one instruction: lane -> row l >> 3, destination chunk l & 7, source chunk swizzled)
__device__ __forceinline__ void fill_rows8(const _Float16* base, int64_t ld, int row0, int nrows, unsigned char* img, int w8, int lane) {
    const int r = 8 * w8 + (lane >> 3);
    const int src_row = min(row0 + r, nrows - 1);
    const int chunk = (lane & 7) ^ swz128(r);
    glds16(reinterpret_cast<const unsigned char*>(base + (int64_t)src_row * ld) + chunk * 16, img + (8 * w8) * kRowB);
}

// row-wise (ds_read_b128) fragment of a 16-row tile: row = lane & 15, k = 32 half + 8 (lane >> 4) + j
__device__ __forceinline__ f16x8 row_frag(const unsigned char* img, int tile, int half, int lane) {
    const int r = 16 * tile + (lane & 15);
    const int chunk = 4 * half + (lane >> 4);
    return *reinterpret_cast<const f16x8*>(img + r * kRowB + ((chunk ^ swz128(r)) << 4));
}

// transposed fragment: element j of lane (g, i) = img[32 kk + (j < 4 ? 4 g + j : 16 + 4 g + j - 4)][16 dt + i]
__device__ __forceinline__ f16x8 col_frag(const unsigned char* img, int kk, int dt, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int r0 = 32 * kk + 4 * g + q, r1 = r0 + 16;
    const int chunk = 2 * dt + (pp >> 1);
    return tr_frag(img + r0 * kRowB + ((chunk ^ swz128(r0)) << 4) + ((pp & 1) << 3),
                   img + r1 * kRowB + ((chunk ^ swz128(r1)) << 4) + ((pp & 1) << 3));
}

// exp2 as ONE v_exp_f32: exp2f() wraps the instruction in a denormal-range rescue (compare, select, ldexp: five instructions per
// element, 34 v_exp_f32 + 36 v_ldexp_f32 + 57 v_cmp + 104 v_cndmask per key block in the forward) for results below 2^-126 -- which
// vanish in every sum and fp16 cast they enter here
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ f16x8 pack8(const f32x4& a, const f32x4& b) {
    return (f16x8){(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
}

// reductions over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) with the VALU lane swaps of gfx950
// (v_permlane32_swap / v_permlane16_swap: no LDS round trip on the softmax's critical path): swapping a register with itself
// leaves {lower half twice, upper half twice} resp. {even rows twice, odd rows twice} in the two results
__device__ __forceinline__ float group_max(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];          // 2 buffers x (K image + V image)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    int blk = xcd_remap(blockIdx.x, gridDim.x);            // the query blocks of one (image, head) share its K / V through one L2
    const int qb = blk % a.q_blocks;
    blk /= a.q_blocks;
    const int h = blk % a.heads, b = blk / a.heads;
    const int q0 = qb * (128 * QT) + wv * (16 * QT);
    const _Float16* Kb = a.K + (int64_t)b * a.Nk * a.ldkv + h * 64;
    const _Float16* Vb = a.V + (int64_t)b * a.Nk * a.ldkv + h * 64;
    const int nkb = (a.Nk + 63) / 64;

    auto issue = [&](int kb, int buf) {
        unsigned char* base = smem + buf * 2 * kImg;
        fill_rows8(Kb, a.ldkv, kb * 64, a.Nk, base, wv, lane);
        fill_rows8(Vb, a.ldkv, kb * 64, a.Nk, base + kImg, wv, lane);
    };
    issue(0, 0);

    f16x8 qf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = min(q0 + 16 * t + li, a.N - 1);
        const _Float16* qp = a.Q + ((int64_t)b * a.N + q) * a.ldq + h * 64 + 8 * g;
        qf[t][0] = *reinterpret_cast<const f16x8*>(qp);
        qf[t][1] = *reinterpret_cast<const f16x8*>(qp + 32);
    }
    f32x4 o[4][QT];
    float m[QT], l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m[t] = -1e30f;
        l[t] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    for (int kb = 0; kb < nkb; ++kb) {
        const bool ahead = kb + 1 < nkb;
        if (ahead) issue(kb + 1, (kb + 1) & 1);
        if (ahead) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* Ki = smem + (kb & 1) * 2 * kImg;
        const unsigned char* Vi = Ki + kImg;
        f32x4 s[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f16x8 k0 = row_frag(Ki, kt, 0, lane), k1 = row_frag(Ki, kt, 1, lane);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, qf[t][0], acc, 0, 0, 0);
                s[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf[t][1], acc, 0, 0, 0);
            }
        }
        if (kb * 64 + 64 > a.Nk) {                             // last key block of a ragged sequence (uniform branch): keys past Nk
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int t = 0; t < QT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kb * 64 + 16 * kt + 4 * g + e >= a.Nk) s[kt][t][e] = -1e30f;
        }
        f16x8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // m tracks the RAW score maximum (scale > 0); p = exp2(scale_log2e * (s - m)) as one fma per element
            float bm = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) bm = fmaxf(bm, s[kt][t][e]);
            bm = group_max(bm);
            const float mn = fmaxf(m[t], bm);
            if (__any(mn > m[t])) {                                 // wave-uniform: the running maximum rarely moves after the first blocks
                const float alpha = exp2f((m[t] - mn) * a.scale_log2e);
                l[t] *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[d][t][e] *= alpha;
                m[t] = mn;
            }
            const float off = -mn * a.scale_log2e;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = fast_exp2(fmaf(s[kt][t][e], a.scale_log2e, off));
                    s[kt][t][e] = p;
                    ps += p;
                }
            l[t] += ps;
            pf[t][0] = pack8(s[0][t], s[1][t]);
            pf[t][1] = pack8(s[2][t], s[3][t]);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f16x8 vt = col_frag(Vi, kk, d, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) o[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vt, pf[t][kk], o[d][t], 0, 0, 0);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // everyone is done reading before the buffer is refilled
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float lt = group_sum(l[t]);
        const float inv = 1.f / lt;
        const int q = q0 + 16 * t + li;
        if (q < a.N) {
            _Float16* op = a.O + ((int64_t)b * a.N + q) * a.ldo + h * 64 + 4 * g;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                *reinterpret_cast<f16x4*>(op + 16 * d) = (f16x4){(_Float16)(o[d][t][0] * inv), (_Float16)(o[d][t][1] * inv),
                                                                 (_Float16)(o[d][t][2] * inv), (_Float16)(o[d][t][3] * inv)};
            if (g == 0 && a.lse != nullptr) a.lse[((int64_t)b * a.heads + h) * a.N + q] = m[t] * a.scale_log2e + log2f(lt);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, dQ:  P^T = exp2(S^T c - lse), dP^T = V dO^T, dS^T = P^T (dP^T - delta) scale, dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int QT = 2;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    int blk = xcd_remap(blockIdx.x, gridDim.x);            // the query blocks of one (image, head) share its K / V through one L2
    const int qb = blk % a.q_blocks;
    blk /= a.q_blocks;
    const int h = blk % a.heads, b = blk / a.heads;
    const int q0 = qb * 256 + wv * 32;
    const _Float16* Kb = a.K + (int64_t)b * a.Nk * a.ldkv + h * 64;
    const _Float16* Vb = a.V + (int64_t)b * a.Nk * a.ldkv + h * 64;
    const int nkb = (a.Nk + 63) / 64;
    auto issue = [&](int kb, int buf) {
        unsigned char* base = smem + buf * 2 * kImg;
        fill_rows8(Kb, a.ldkv, kb * 64, a.Nk, base, wv, lane);
        fill_rows8(Vb, a.ldkv, kb * 64, a.Nk, base + kImg, wv, lane);
    };
    issue(0, 0);

    f16x8 qf[QT][2], df[QT][2];
    float lse[QT], delta[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = min(q0 + 16 * t + li, a.N - 1);
        const int64_t row = (int64_t)b * a.N + q;
        const _Float16* qp = a.Q + row * a.ldq + h * 64 + 8 * g;
        const _Float16* dp = a.dO + row * a.ldo + h * 64 + 8 * g;
        const _Float16* op = a.O + row * a.ldo + h * 64 + 8 * g;
        qf[t][0] = *reinterpret_cast<const f16x8*>(qp);
        qf[t][1] = *reinterpret_cast<const f16x8*>(qp + 32);
        df[t][0] = *reinterpret_cast<const f16x8*>(dp);
        df[t][1] = *reinterpret_cast<const f16x8*>(dp + 32);
        const f16x8 o0 = *reinterpret_cast<const f16x8*>(op), o1 = *reinterpret_cast<const f16x8*>(op + 32);
        float dsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)df[t][0][e] * (float)o0[e] + (float)df[t][1][e] * (float)o1[e];
        delta[t] = group_sum(dsum);
        lse[t] = a.lse[((int64_t)b * a.heads + h) * a.N + q];
        if (g == 0 && q0 + 16 * t + li < a.N) a.delta[((int64_t)b * a.heads + h) * a.N + q] = delta[t];
    }
    f32x4 dq[4][QT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[d][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < nkb; ++kb) {
        const bool ahead = kb + 1 < nkb;
        if (ahead) issue(kb + 1, (kb + 1) & 1);
        if (ahead) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* Ki = smem + (kb & 1) * 2 * kImg;
        const unsigned char* Vi = Ki + kImg;
        const bool tail = kb * 64 + 64 > a.Nk;
        f16x8 dsf[QT][2];
        f32x4 ds[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f16x8 k0 = row_frag(Ki, kt, 0, lane), k1 = row_frag(Ki, kt, 1, lane);
            const f16x8 v0 = row_frag(Vi, kt, 0, lane), v1 = row_frag(Vi, kt, 1, lane);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, pa = (f32x4){0.f, 0.f, 0.f, 0.f};
                sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, qf[t][0], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf[t][1], sa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, df[t][0], pa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, df[t][1], pa, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = fast_exp2(sa[e] * a.scale_log2e - lse[t]);
                    ds[kt][t][e] = p * (pa[e] - delta[t]) * a.scale;
                }
            }
        }
        if (tail) {                                            // (uniform branch) keys past Nk: clamped copies of the last key
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int t = 0; t < QT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kb * 64 + 16 * kt + 4 * g + e >= a.Nk) ds[kt][t][e] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            dsf[t][0] = pack8(ds[0][t], ds[1][t]);
            dsf[t][1] = pack8(ds[2][t], ds[3][t]);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f16x8 kt_ = col_frag(Ki, kk, d, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) dq[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kt_, dsf[t][kk], dq[d][t], 0, 0, 0);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0 + 16 * t + li;
        if (q < a.N) {
            _Float16* op = a.dQ + ((int64_t)b * a.N + q) * a.ldq + h * 64 + 4 * g;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                *reinterpret_cast<f16x4*>(op + 16 * d) = (f16x4){(_Float16)dq[d][t][0], (_Float16)dq[d][t][1], (_Float16)dq[d][t][2], (_Float16)dq[d][t][3]};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, dK / dV: the block owns 64 keys (16 per wave), walks its range of 64-query blocks.
//   S = Q K^T (row = query 4 g + e, col = key), P = exp2(S c - lse_q), dP = dO V^T, dS = P (dP - delta_q) scale
//   dV += P^T dO, dK += dS^T Q: P / dS tiles are the A operands as they stand (k-slots = queries), dO / Q come transposed
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];          // 2 buffers x (Q image + dO image) + statistics
    float* stat = reinterpret_cast<float*>(smem + 4 * kImg);       // [2 buffers][4 waves][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    int blk = xcd_remap(blockIdx.x, gridDim.x);            // the key blocks of one (image, head, query range) re-read the same Q / dO
    const int kblk = blk % a.k_blocks;
    blk /= a.k_blocks;
    const int split = blk % a.splits;
    blk /= a.splits;
    const int h = blk % a.heads, b = blk / a.heads;
    const int key = min(kblk * 64 + wv * 16 + li, a.Nk - 1);
    const _Float16* kp = a.K + ((int64_t)b * a.Nk + key) * a.ldkv + h * 64 + 8 * g;
    const _Float16* vp = a.V + ((int64_t)b * a.Nk + key) * a.ldkv + h * 64 + 8 * g;
    const f16x8 kf0 = *reinterpret_cast<const f16x8*>(kp), kf1 = *reinterpret_cast<const f16x8*>(kp + 32);
    const f16x8 vf0 = *reinterpret_cast<const f16x8*>(vp), vf1 = *reinterpret_cast<const f16x8*>(vp + 32);
    const _Float16* Qb = a.Q + (int64_t)b * a.N * a.ldq + h * 64;
    const _Float16* Db = a.dO + (int64_t)b * a.N * a.ldo + h * 64;
    const float* lse_b = a.lse + ((int64_t)b * a.heads + h) * a.N;
    const float* del_b = a.delta + ((int64_t)b * a.heads + h) * a.N;
    const int nqb_total = (a.N + 63) / 64;
    const int qb0 = split * a.qblocks_per_split;
    const int qb1 = min(qb0 + a.qblocks_per_split, nqb_total);

    // per block and wave 5 LDS-DMA instructions: rows 16 w .. 16 w + 15 of the Q and dO images (2 + 2) and one 4-byte-wide
    // one for the statistics of those 16 queries: stat[buf][w][0..15] = lse, [16..31] = delta (lanes 32..63 fetch a dummy)
    auto issue = [&](int qb, int buf) {
        unsigned char* base = smem + buf * 2 * kImg;
        fill_rows8(Qb, a.ldq, qb * 64, a.N, base, 2 * wv, lane);
        fill_rows8(Qb, a.ldq, qb * 64, a.N, base, 2 * wv + 1, lane);
        fill_rows8(Db, a.ldo, qb * 64, a.N, base + kImg, 2 * wv, lane);
        fill_rows8(Db, a.ldo, qb * 64, a.N, base + kImg, 2 * wv + 1, lane);
        const int q = min(qb * 64 + 16 * wv + (lane & 15), a.N - 1);
        const float* src = (lane & 16) ? del_b + q : lse_b + q;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stat + buf * 256 + wv * 64), 4, 0, 0);
    };

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (qb0 < qb1) issue(qb0, 0);
    for (int qb = qb0; qb < qb1; ++qb) {
        const int buf = (qb - qb0) & 1;
        const bool ahead = qb + 1 < qb1;
        if (ahead) issue(qb + 1, buf ^ 1);
        if (ahead) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); // the next block's 5 LDS-DMA loads stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* Qi = smem + buf * 2 * kImg;
        const unsigned char* Di = Qi + kImg;
        const float* sl = stat + buf * 256;
        const bool qtail = qb * 64 + 64 > a.N;
        f32x4 p[4], ds[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4 sa = (f32x4){0.f, 0.f, 0.f, 0.f}, pa = (f32x4){0.f, 0.f, 0.f, 0.f};
            sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(row_frag(Qi, qt, 0, lane), kf0, sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(row_frag(Qi, qt, 1, lane), kf1, sa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_16x16x32_f16(row_frag(Di, qt, 0, lane), vf0, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_16x16x32_f16(row_frag(Di, qt, 1, lane), vf1, pa, 0, 0, 0);
            const float4 l4 = *reinterpret_cast<const float4*>(sl + 64 * qt + 4 * g);
            const float4 d4 = *reinterpret_cast<const float4*>(sl + 64 * qt + 16 + 4 * g);
            const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pe = fast_exp2(sa[e] * a.scale_log2e - le[e]);
                p[qt][e] = pe;
                ds[qt][e] = pe * (pa[e] - de[e]) * a.scale;
            }
        }
        if (qtail) {                                           // (uniform branch) rows past N were filled with a clamped query
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (qb * 64 + 16 * qt + 4 * g + e >= a.N) p[qt][e] = ds[qt][e] = 0.f;
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const f16x8 pf = pack8(p[2 * qq], p[2 * qq + 1]);
            const f16x8 sf = pack8(ds[2 * qq], ds[2 * qq + 1]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, col_frag(Di, qq, d, lane), dv[d], 0, 0, 0);
                dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sf, col_frag(Qi, qq, d, lane), dk[d], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // slab[split][b*Nk + key][two*C + h*64 + d]: rows = keys 4 g + e of this wave's 16, cols = d = 16 dt + li
    const int C2 = 2 * a.heads * 64;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int kq = kblk * 64 + wv * 16 + 4 * g + e;
        if (kq >= a.Nk) continue;
        float* row = a.slab + ((int64_t)split * a.B * a.Nk + (int64_t)b * a.Nk + kq) * C2 + h * 64 + li;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            row[16 * d] = dk[d][e];
            row[a.heads * 64 + 16 * d] = dv[d][e];
        }
    }
}

__global__ __launch_bounds__(256) void dkv_reduce_kernel(const float* __restrict__ slab, _Float16* __restrict__ dkv, int64_t n4, int splits) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = reinterpret_cast<const float4*>(slab)[i];
        for (int k = 1; k < splits; ++k) {
            const float4 v = reinterpret_cast<const float4*>(slab)[(int64_t)k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<f16x4*>(dkv)[i] = (f16x4){(_Float16)s.x, (_Float16)s.y, (_Float16)s.z, (_Float16)s.w};
    }
}

}  // namespace mit
}  // namespace diga

using namespace diga;
using namespace diga::mit;

static int attn_check(const char* who, const void* q, const void* kv, int64_t B, int64_t heads, int64_t N, int64_t Nk, int64_t ldq,
                      int64_t ldkv, int64_t ldo) {
    DIGA_REQUIRE(q && kv && B > 0 && heads > 0 && N > 0 && Nk > 0, DIGA_EINVAL, "%s: null pointer / empty shape", who);
    DIGA_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0 && ldq >= heads * 64 && ldkv >= 2 * heads * 64 && ldo >= heads * 64, DIGA_EINVAL,
                 "%s: head_dim is 64; leading dimensions must be multiples of 8 and hold all heads", who);
    DIGA_REQUIRE(aligned16(q) && aligned16(kv), DIGA_EALIGN, "%s: operands must be 16-byte aligned", who);
    DIGA_REQUIRE(B * N < (1ll << 31) && B * Nk < (1ll << 31), DIGA_EINVAL, "%s: too many tokens", who);
    return DIGA_OK;
}

/* q [B*N][ldq] (head h at columns 64h..), kv [B*Nk][ldkv] (K at columns 64h.., V at heads*64 + 64h..): the layouts the q / kv
 * Linear layers produce (MixTransfomer.py:122,130-132).  out [B*N][ldo], lse [B][heads][N] (nullable). */
extern "C" int diga_mit_attention_fwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, void* out, int64_t ldo, float* lse,
                                      int64_t B, int64_t heads, int64_t N, int64_t Nk, float scale, void* stream) {
    int rc = attn_check("mit_attention_fwd", q, kv, B, heads, N, Nk, ldq, ldkv, ldo);
    if (rc) return rc;
    DIGA_REQUIRE(out && aligned16(out), DIGA_EINVAL, "mit_attention_fwd: out");
    hipStream_t st = static_cast<hipStream_t>(stream);
    AttnArgs a{};
    a.Q = static_cast<const _Float16*>(q); a.K = static_cast<const _Float16*>(kv); a.V = a.K + heads * 64;
    a.ldq = ldq; a.ldkv = ldkv; a.O = static_cast<_Float16*>(out); a.ldo = ldo; a.lse = lse;
    a.B = (int)B; a.heads = (int)heads; a.N = (int)N; a.Nk = (int)Nk;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    ProfScope prof(DIGA_PROF_MIT_ATTN_FWD, st, 4.0 * (double)B * heads * (double)N * (double)Nk * 64.0);
    // 32 queries per wave (124 VGPRs, two blocks per CU).  A 64-query variant (K / V fragments read from LDS once per 64 MFMAs,
    // 209 VGPRs, one block per CU) measured 15-25 % SLOWER on every MiT-B5 stage (tools/bench_mit_ops.py --cold: 245 vs 207 us on
    // stage 1): the softmax's latency needs the second resident block.
    a.q_blocks = (int)ceil_div(N, 256);
    hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((unsigned)(B * heads * a.q_blocks)), dim3(512), 4 * kImg, st, a);
    return launch_status("mit_attention_fwd");
}

namespace {
struct DkvPlan {
    int k_blocks, splits, qblocks_per_split;
};
DkvPlan dkv_plan(int64_t B, int64_t heads, int64_t N, int64_t Nk) {
    DkvPlan p;
    p.k_blocks = (int)ceil_div(Nk, 64);
    const int64_t nqb = ceil_div(N, 64);
    int64_t splits = ceil_div(1024, B * heads * p.k_blocks);
    if (splits > ceil_div(nqb, 4)) splits = ceil_div(nqb, 4);
    if (splits < 1) splits = 1;
    p.qblocks_per_split = (int)ceil_div(nqb, splits);
    p.splits = (int)ceil_div(nqb, p.qblocks_per_split);
    return p;
}
}  // namespace

extern "C" size_t diga_mit_attention_bwd_workspace_bytes(int64_t B, int64_t heads, int64_t N, int64_t Nk) {
    if (B <= 0 || heads <= 0 || N <= 0 || Nk <= 0) return 0;
    const DkvPlan p = dkv_plan(B, heads, N, Nk);
    return (size_t)p.splits * (size_t)B * (size_t)Nk * 2 * (size_t)heads * 64 * sizeof(float) + (size_t)B * heads * N * sizeof(float);
}

/* Gradients of diga_mit_attention_fwd: d_out [B*N][ldo] -> dq [B*N][ldq], dkv [B*Nk][ldkv] (same column layout as kv). */
extern "C" int diga_mit_attention_bwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const void* out, const void* d_out,
                                      int64_t ldo, const float* lse, void* dq, void* dkv, void* workspace, size_t workspace_bytes,
                                      int64_t B, int64_t heads, int64_t N, int64_t Nk, float scale, void* stream) {
    int rc = attn_check("mit_attention_bwd", q, kv, B, heads, N, Nk, ldq, ldkv, ldo);
    if (rc) return rc;
    DIGA_REQUIRE(out && d_out && lse && dq && dkv && workspace, DIGA_EINVAL, "mit_attention_bwd: null pointer");
    DIGA_REQUIRE(ldkv == 2 * heads * 64, DIGA_EINVAL, "mit_attention_bwd: dkv is written densely, ldkv must be 2*heads*64");
    DIGA_REQUIRE(aligned16(out) && aligned16(d_out) && aligned16(dq) && aligned16(dkv) && aligned16(workspace), DIGA_EALIGN,
                 "mit_attention_bwd: alignment");
    const DkvPlan p = dkv_plan(B, heads, N, Nk);
    const size_t slab_bytes = (size_t)p.splits * (size_t)B * (size_t)Nk * 2 * (size_t)heads * 64 * sizeof(float);
    DIGA_REQUIRE(workspace_bytes >= slab_bytes + (size_t)B * heads * N * sizeof(float), DIGA_EWORKSPACE, "mit_attention_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    AttnArgs a{};
    a.Q = static_cast<const _Float16*>(q); a.K = static_cast<const _Float16*>(kv); a.V = a.K + heads * 64;
    a.ldq = ldq; a.ldkv = ldkv; a.O = const_cast<_Float16*>(static_cast<const _Float16*>(out)); a.ldo = ldo;
    a.lse = const_cast<float*>(lse); a.dO = static_cast<const _Float16*>(d_out); a.dQ = static_cast<_Float16*>(dq);
    a.slab = static_cast<float*>(workspace);
    a.delta = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + slab_bytes);
    a.B = (int)B; a.heads = (int)heads; a.N = (int)N; a.Nk = (int)Nk;
    a.q_blocks = (int)ceil_div(N, 256);
    a.k_blocks = p.k_blocks; a.splits = p.splits; a.qblocks_per_split = p.qblocks_per_split;
    a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    ProfScope prof(DIGA_PROF_MIT_ATTN_BWD, st, 14.0 * (double)B * heads * (double)N * (double)Nk * 64.0);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(B * heads * a.q_blocks)), dim3(512), 4 * kImg, st, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(B * heads * p.splits * p.k_blocks)), dim3(256), 4 * kImg + 2 * 256 * sizeof(float), st, a);
    const int64_t n4 = B * Nk * 2 * heads * 64 / 4;
    int64_t grid = ceil_div(n4, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(dkv_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, st, a.slab, static_cast<_Float16*>(dkv), n4, p.splits);
    return launch_status("mit_attention_bwd");
}
