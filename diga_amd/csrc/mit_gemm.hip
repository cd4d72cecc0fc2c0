// fp16 MFMA GEMMs of the MiT / SegFormer student (BASELINE configs[4]): every nn.Linear of
// G5/model/networks/MixTransfomer.py (Attention.q/kv/proj :97-99, Mlp.fc1/fc2 :52-55), the patch-embedding and
// spatial-reduction convolutions as GEMMs over gathered rows (OverlapPatchEmbed.proj :200, Attention.sr :105), forward,
// backward-data and backward-weight.  fp16 operands, fp32 accumulation on v_mfma_f32_16x16x32_f16.
//
//   gemm_nt_kernel   C[M,N] = A[M,K] . B[N,K]^T      (forward: B = W;  backward-data: A = dY, B = W^T)
//   gemm_tn_kernel   C[N,K] = sum_m A[m,N] . B[m,K]  (backward-weight: A = dY, B = X; split over m, fp32 slabs)
//
// Both: 128 x 128 (or 128 x 64) block tile, 4 waves as 2 x 2, operands copied global -> LDS by LDS-DMA loads
// (global_load_lds_dwordx4: no registers, no ds_write) into a three-stage ring with counted vmcnt and one raw barrier per
// K-step; LDS images are swizzled through the per-lane SOURCE address.  These layers have short reductions (K = 64..2048)
// and, in the first two stages, hundreds of thousands of rows: most of them are HBM-bound, so
// the kernels keep LDS / register footprints small enough for three blocks per CU rather than chasing MFMA issue.
#include "mit_common.h"

namespace diga {
namespace mit {

// 16 zero bytes every out-of-range LDS-DMA load fetches instead
__device__ __attribute__((aligned(16))) unsigned char g_mit_zero16[16];

struct GemmArgs {
    const _Float16* A;
    int64_t lda;
    const _Float16* B;
    int64_t ldb;
    const float* bias;        // [N] or null
    void* out;                // fp16 or fp32 [M][ldc]
    int64_t ldc;
    const float* res;         // fp32 residual [M][ldr] or null (out = res + seg_scale * (alpha * acc + bias))
    int64_t ldr;
    const float* seg_scale;   // per-segment (image) scale, null = 1: DropPath (MixTransfomer.py:176-177)
    int rows_per_seg;
    int M, N, K;
    int tiles_n;
    int out_f32;
    int accumulate;           // out += result (fp16 out only; read-modify-write)
    int plain16;              // host: fp16 output, no residual / segment scale / accumulate, 16-byte rows -> the whole-tile fp16 epilogue
    float alpha;
};

// ---------------------------------------------------------------------------------------------
// NT: A [M][K], B [N][K], both K-contiguous.  K % 32 == 0.
// ---------------------------------------------------------------------------------------------
// WM = wave rows: the block tile is (64 WM) x (64 TN) with 2 WM waves as WM x 2, wave tile 64 x (32 TN).  WM = 2: 128-row tile,
// 4 waves, three blocks per CU.  WM = 4 (TN = 2 only): 256 x 128 tile, 8 waves, 1.33x the MACs per staged byte, two blocks per
// CU -- taken where the 128-row tiling would leave a mostly empty last round of blocks (diga_mit_gemm_nt below).
template <int TN, int WM>
__global__ __launch_bounds__(128 * WM, 2) void gemm_nt_kernel(GemmArgs a) {
    constexpr int BM = 64 * WM, BN = 64 * TN, MT = 4, NT = 2 * TN, NW = 2 * WM;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int BJ = (BN / 16) / NW;                            // B-tile LDS-DMA instructions per wave and stage
    static_assert(BJ >= 1 && BJ * NW * 16 == BN, "B tile must split evenly over the waves");
    constexpr int kLoads = 2 + BJ;                                // LDS-DMA instructions per wave and stage
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ksteps = a.K / 32;

    // loader geometry: one LDS-DMA instruction fills 16 rows x 64 bytes; lane -> row lane >> 2, destination slot lane & 3
    const int lrow = lane >> 2;
    const int kslot = (lane & 3) ^ swz64(lrow);
    const unsigned char* pa[2];
    const unsigned char* pb[BJ];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = min(m0 + wv * 32 + 16 * j + lrow, a.M - 1);
        pa[j] = reinterpret_cast<const unsigned char*>(a.A + (int64_t)m * a.lda) + kslot * 16;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int n = min(n0 + wv * 16 * BJ + 16 * j + lrow, a.N - 1);
        pb[j] = reinterpret_cast<const unsigned char*>(a.B + (int64_t)n * a.ldb) + kslot * 16;
    }
    auto issue = [&](int ks, int buf) {
        unsigned char* stage = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(pa[j] + ks * 64, stage + (wv * 32 + 16 * j) * 64);
#pragma unroll
        for (int j = 0; j < BJ; ++j) glds16(pb[j] + ks * 64, stage + A_BYTES + (wv * 16 * BJ + 16 * j) * 64);
    };
    auto wait_next = [&](bool newest_in_flight) {
        if (newest_in_flight) {
            if constexpr (kLoads == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    static_assert(kLoads == 3 || kLoads == 4, "vmcnt literals above");

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ swz64(frow)) << 4);
    const int aoff = wm * 64 * 64 + foff;
    const int boff = A_BYTES + wn * 32 * TN * 64 + foff;

    issue(0, 0);
    if (ksteps > 1) issue(1, 1);
    wait_next(ksteps > 1);
    int cur = 0, nx = 2;
    for (int ks = 0; ks < ksteps; ++ks) {
        const bool ahead = ks + 2 < ksteps;
        if (ahead) issue(ks + 2, nx);                      // that stage was last read in step ks - 1 (barrier since)
        const unsigned char* As = smem + cur * STAGE + aoff;
        const unsigned char* Bs = smem + cur * STAGE + boff;
        f16x8 fb[NT], fa[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f16x8*>(Bs + j * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f16x8*>(As + i * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        wait_next(ahead);                                  // own loads of step ks + 1 landed; fragment reads of this step done
        cur = cur == 2 ? 0 : cur + 1;
        nx = nx == 2 ? 0 : nx + 1;
    }

    const int t = threadIdx.x;
    // epilogue, plain fp16 output (q / kv / fc1 and the backward-data GEMMs: no residual, no read-modify-write): alpha and bias in
    // registers, the WHOLE block tile staged as fp16 (BM x (BN + 8) halves fit the operand ring), one barrier, 16-byte row segments
    // out.  (The general path below stages fp32 in 64-row quarters: 2 WM barriers, 8-byte stores.)
    if (a.plain16 && n0 + BN <= a.N) {                     // uniform over the block
        constexpr int LDH = BN + 8;
        _Float16* st16 = reinterpret_cast<_Float16*>(smem);
        static_assert(BM * LDH * 2 <= 3 * STAGE, "fp16 tile image must fit the operand ring");
        float bv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = a.bias != nullptr ? a.bias[n0 + wn * 32 * TN + j * 16 + (lane & 15)] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st16[(wm * 64 + i * 16 + (lane >> 4) * 4 + e) * LDH + wn * 32 * TN + j * 16 + (lane & 15)] =
                        (_Float16)(acc[i][j][e] * a.alpha + bv[j]);
        __syncthreads();
        constexpr int OCT = BN / 8;                        // 16-byte groups per row
        for (int idx = t; idx < BM * OCT; idx += 64 * NW) {
            const int r = idx / OCT, q = idx - r * OCT;
            const int m = m0 + r;
            if (m >= a.M) continue;
            *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(a.out) + (int64_t)m * a.ldc + n0 + q * 8) =
                *reinterpret_cast<const f16x8*>(st16 + r * LDH + q * 8);
        }
        return;
    }
    // epilogue (general: fp32 output, residual, segment scale, read-modify-write): 64-row halves through LDS as fp32, whole row
    // segments out with 16-byte (fp32) / 8-byte (fp16) accesses.  (Two wave rows per pass where the ring holds them: measured neutral.)
    constexpr int LDS_LD = BN + 4;
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int h = 0; h < WM; ++h) {
        if (m0 + h * 64 >= a.M) break;                     // uniform over the block
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
        constexpr int QUADS = BN / 4;                      // float4 groups per row
        for (int idx = t; idx < 64 * QUADS; idx += 64 * NW) {
            const int r = idx / QUADS, q = idx - r * QUADS;
            const int m = m0 + h * 64 + r, n = n0 + q * 4;
            if (m >= a.M || n >= a.N) continue;
            const float4 v4 = *reinterpret_cast<const float4*>(stage + r * LDS_LD + q * 4);
            float v[4] = {v4.x * a.alpha, v4.y * a.alpha, v4.z * a.alpha, v4.w * a.alpha};
            if (a.bias != nullptr) {
                const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            if (a.seg_scale != nullptr) {
                const float s = a.seg_scale[m / a.rows_per_seg];
                v[0] *= s; v[1] *= s; v[2] *= s; v[3] *= s;
            }
            if (a.res != nullptr) {
                const float4 r4 = *reinterpret_cast<const float4*>(a.res + (int64_t)m * a.ldr + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (a.out_f32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (int64_t)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                _Float16* o = reinterpret_cast<_Float16*>(a.out) + (int64_t)m * a.ldc + n;
                if (a.accumulate) {
                    const f16x4 old = *reinterpret_cast<const f16x4*>(o);
                    v[0] += (float)old[0]; v[1] += (float)old[1]; v[2] += (float)old[2]; v[3] += (float)old[3];
                }
                *reinterpret_cast<f16x4*>(o) = (f16x4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// TN (backward-weight): slab[split][n][k] = sum over the split's rows m of A[m][n] * B[m][k].
// A [M][lda] (N columns used), B [M][ldb] (K columns used); N % 8 == 0, K % 8 == 0.  Both operands are row(m)-major while
// the contraction runs over m: LDS image [32 rows][128 columns] per operand, fragments (8 consecutive rows of one column)
// read with the transposing ds_read_b64_tr_b16.
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const _Float16* A;
    int64_t lda;
    const _Float16* B;
    int64_t ldb;
    float* slab;
    float* bias_slab;         // [splits][N] column sums of A (the bias gradient), or null
    int M, N, K;
    int tiles_n, tiles_k, splits, steps_per_split;
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(WgradArgs a) {
    constexpr int MT = 4, NT = 4, ROWB = 256;                     // 128 columns x 2 bytes
    constexpr int PLANE = 32 * ROWB, STAGE = 2 * PLANE;           // 16 KB
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_k = wg % a.tiles_k;
    wg /= a.tiles_k;
    const int tile_n = wg % a.tiles_n;
    const int split = wg / a.tiles_n;
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int p_begin = split * a.steps_per_split * 32;
    const int p_end = min(p_begin + a.steps_per_split * 32, a.M);
    const int ksteps = p_end > p_begin ? (p_end - p_begin + 31) / 32 : 0;

    // loader: per operand 32 rows x 16 chunks = 8 LDS-DMA instructions, 2 per wave: rows 8 wv + 4 j + (lane >> 4), chunk lane & 15
    const int nmax = a.N / 8 - 1, kmax = a.K / 8 - 1;
    auto issue = [&](int ks, int buf) {
        unsigned char* stage = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * wv + 4 * j + (lane >> 4);
            const int p = p_begin + ks * 32 + row;
            const int sw = (lane & 15) ^ (tr_key(row) << 1);
            const bool ok = p < p_end;
            const unsigned char* sa = ok ? reinterpret_cast<const unsigned char*>(a.A + (int64_t)p * a.lda) + (int64_t)min(n0 / 8 + sw, nmax) * 16 : g_mit_zero16;
            const unsigned char* sb = ok ? reinterpret_cast<const unsigned char*>(a.B + (int64_t)p * a.ldb) + (int64_t)min(k0 / 8 + sw, kmax) * 16 : g_mit_zero16;
            glds16(sa, stage + (8 * wv + 4 * j) * ROWB);
            glds16(sb, stage + PLANE + (8 * wv + 4 * j) * ROWB);
        }
    };
    auto wait_next = [&](bool newest_in_flight) {
        if (newest_in_flight) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x4 acc[MT][NT];
    float bsum[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the bias gradient (column sums of A = dY) rides along: the A fragments pass through registers anyway
    const bool do_bias = a.bias_slab != nullptr && tile_k == 0 && wn == 0;
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int row0 = 8 * g + q, row1 = row0 + 4;
    const int key0 = tr_key(row0) << 1, key1 = tr_key(row1) << 1;
    auto off = [&](int tile, int row, int key) { return row * ROWB + (((2 * tile + (pp >> 1)) ^ key) << 4) + ((pp & 1) << 3); };

    if (ksteps > 0) issue(0, 0);
    if (ksteps > 1) issue(1, 1);
    wait_next(ksteps > 1);
    int cur = 0, nx = 2;
    for (int ks = 0; ks < ksteps; ++ks) {
        const bool ahead = ks + 2 < ksteps;
        if (ahead) issue(ks + 2, nx);
        const unsigned char* As = smem + cur * STAGE;
        const unsigned char* Bs = As + PLANE;
        f16x8 fb[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = tr_frag(Bs + off(wn * 4 + j, row0, key0), Bs + off(wn * 4 + j, row1, key1));
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f16x8 fa = tr_frag(As + off(wm * 4 + i, row0, key0), As + off(wm * 4 + i, row1, key1));
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[j], acc[i][j], 0, 0, 0);
            if (do_bias)
                bsum[i] += (((float)fa[0] + (float)fa[1]) + ((float)fa[2] + (float)fa[3])) + (((float)fa[4] + (float)fa[5]) + ((float)fa[6] + (float)fa[7]));
        }
        wait_next(ahead);
        cur = cur == 2 ? 0 : cur + 1;
        nx = nx == 2 ? 0 : nx + 1;
    }
    if (do_bias) {                                         // uniform per wave
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = n0 + wm * 64 + i * 16 + (lane & 15);
            if (g == 0 && n < a.N) a.bias_slab[(int64_t)split * a.N + n] = v;
        }
    }
    float* out = a.slab + (int64_t)split * a.N * a.K;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int k = k0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wm * 64 + i * 16 + (lane >> 4) * 4 + e;
                if (n < a.N && k < a.K) out[(int64_t)n * a.K + k] = acc[i][j][e];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// column sums of an fp16 matrix (bias gradients): partial[chunk][c] over 256-row chunks, then a fixed-order reduce
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const _Float16* __restrict__ x, int64_t ld, float* __restrict__ partial,
                                                             int M, int C, int rows_per_block) {
    // thread -> 8-column group (c8) and row phase; C % 8 == 0
    const int groups = C / 8;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, M);
    __shared__ float red[256 * 8];
    for (int gbase = 0; gbase < groups; gbase += 256) {
        // lanes along the column groups for coalescing: t % gw = group, t / gw = row phase
        const int gw = min(groups - gbase, 256);
        // use the largest power-of-two split of the block over rows
        int phases = 256 / gw;
        if (phases < 1) phases = 1;
        const int gi = threadIdx.x % gw, ph = threadIdx.x / gw;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ph < phases) {
            for (int r = r0 + ph; r < r1; r += phases) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(x + (int64_t)r * ld + (gbase + gi) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
        __syncthreads();
        if (threadIdx.x < gw) {
            float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < phases; ++p)
#pragma unroll
                for (int e = 0; e < 8; ++e) tot[e] += red[(p * gw + threadIdx.x) * 8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) partial[(int64_t)blockIdx.x * C + (gbase + threadIdx.x) * 8 + e] = tot[e];
        }
        __syncthreads();
    }
}

// fp32 [R][C] -> fp16 copy [R][C] and (optionally) the fp16 transpose [C][R]   (weights, once per optimizer step)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ w, _Float16* __restrict__ w16,
                                                             _Float16* __restrict__ wt16, int R, int C) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = w[(int64_t)r * C + c];
            if (w16 != nullptr) w16[(int64_t)r * C + c] = (_Float16)v;
        }
        tile[ty + 8 * i][tx] = v;
    }
    if (wt16 == nullptr) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) wt16[(int64_t)c * R + r] = (_Float16)tile[tx][ty + 8 * i];
    }
}


// All weights of a model in ONE launch (once per forward): tensor t is a [Co][Ci][R][S] fp32 parameter (Linear: R = S = 1), its
// GEMM form is rows[co][k], k = (r * S + s) * Ci + ci, zero-padded to Kp columns:
//   mode 0: w16 [Co][Kp] fp16 and wt16 [Kp][Co] fp16 (forward / backward-data operands)
//   mode 1: fp32 transposes for the depthwise 3x3 (Ci = 1): out_a [R*S][Co] and out_b = the same with the taps reversed
// One block per 32 x 32 tile of (co, k); the tile -> tensor map is a prefix array searched per block.
struct PrepEntry {
    const float* src;
    void* out_a;
    void* out_b;
    int Co, Ci, RS, Kp, mode, tiles_k;
};

__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const PrepEntry* __restrict__ tab, const int64_t* __restrict__ tile_start,
                                                                int n_tensors) {
    __shared__ float tile[32][33];
    const int64_t blk = blockIdx.x;
    int lo = 0, hi = n_tensors - 1;
    while (lo < hi) {                                               // last tensor whose first tile <= blk
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    const PrepEntry e = tab[lo];
    const int local = (int)(blk - tile_start[lo]);
    const int tk = local % e.tiles_k, tc = local / e.tiles_k;
    const int c0 = tc * 32, k0 = tk * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int K = e.RS * e.Ci;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = c0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (co < e.Co && k < K) {
            const int tap = k / e.Ci, ci = k - tap * e.Ci;
            v = e.src[((int64_t)co * e.Ci + ci) * e.RS + tap];
        }
        if (e.mode == 0 && co < e.Co && k < e.Kp) static_cast<_Float16*>(e.out_a)[(int64_t)co * e.Kp + k] = (_Float16)v;
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, co = c0 + tx;
        if (co >= e.Co || k >= e.Kp) continue;
        const float v = tile[tx][ty + 8 * i];
        if (e.mode == 0) {
            static_cast<_Float16*>(e.out_b)[(int64_t)k * e.Co + co] = (_Float16)v;
        } else if (k < K) {
            static_cast<float*>(e.out_a)[(int64_t)k * e.Co + co] = v;
            static_cast<float*>(e.out_b)[(int64_t)(K - 1 - k) * e.Co + co] = v;
        }
    }
}

}  // namespace mit
}  // namespace diga

using namespace diga;
using namespace diga::mit;

extern "C" int diga_mit_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* out, int64_t ldc,
                                int out_f32, const float* residual, int64_t ldr, const float* seg_scale, int64_t rows_per_seg,
                                int accumulate, float alpha, int64_t M, int64_t N, int64_t K, void* stream) {
    DIGA_REQUIRE(A && B && out && M > 0 && N > 0 && K > 0, DIGA_EINVAL, "mit_gemm_nt: null pointer / empty shape");
    DIGA_REQUIRE(K % 32 == 0 && N % 4 == 0, DIGA_EINVAL, "mit_gemm_nt: K %% 32 == 0 and N %% 4 == 0 required (K=%lld N=%lld)",
                 (long long)K, (long long)N);
    DIGA_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && (residual == nullptr || ldr % 4 == 0), DIGA_EINVAL,
                 "mit_gemm_nt: leading dimensions must keep 16-byte rows");
    DIGA_REQUIRE(aligned16(A) && aligned16(B) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, DIGA_EALIGN, "mit_gemm_nt: alignment");
    DIGA_REQUIRE(!(accumulate && out_f32) && !(seg_scale && rows_per_seg <= 0), DIGA_EINVAL, "mit_gemm_nt: bad epilogue options");
    DIGA_REQUIRE(M < (1ll << 31) && N < (1 << 24), DIGA_EINVAL, "mit_gemm_nt: shape too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GemmArgs a;
    a.A = static_cast<const _Float16*>(A); a.lda = lda;
    a.B = static_cast<const _Float16*>(B); a.ldb = ldb;
    a.bias = bias; a.out = out; a.ldc = ldc; a.res = residual; a.ldr = ldr;
    a.seg_scale = seg_scale; a.rows_per_seg = (int)rows_per_seg;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.out_f32 = out_f32; a.accumulate = accumulate; a.alpha = alpha;
    a.plain16 = !out_f32 && residual == nullptr && seg_scale == nullptr && !accumulate && ldc % 8 == 0 && aligned16(out);
    ProfScope prof(DIGA_PROF_MIT_GEMM, st, 2.0 * (double)M * (double)N * (double)K);
    if (N > 64) {
        a.tiles_n = (int)ceil_div(N, 128);
        // 256-row tiles (fewer, fatter blocks, 2 per CU, 1.33x the MACs per staged byte) where the 128-row tiling needs more than
        // one round of blocks (3 per CU = 768 slots) AND the layer is wide: measured on the MiT-B5 shapes (tools/bench_mit_ops.py
        // --cold): N >= 256 gains 5-14 % (q / proj / fc1 / fc2 of stage 3, fc1 of stages 1-2), N = 128 loses 3-5 %, small-M
        // layers (kv, sr: 9216 rows) lose 8-28 %
        const int64_t blocks128 = ceil_div(M, 128) * a.tiles_n;
        const bool big = blocks128 > 768 && N >= 256;
        if (big) {
            constexpr int SH = 3 * (256 * 64 + 128 * 64);
            static bool once = [] { return hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SH) == hipSuccess; }();
            (void)once;
            hipLaunchKernelGGL((gemm_nt_kernel<2, 4>), dim3((unsigned)(ceil_div(M, 256) * a.tiles_n)), dim3(512), SH, st, a);
        } else {
            constexpr int SH = 3 * (128 * 64 + 128 * 64);
            hipLaunchKernelGGL((gemm_nt_kernel<2, 2>), dim3((unsigned)(ceil_div(M, 128) * a.tiles_n)), dim3(256), SH, st, a);
        }
    } else {
        a.tiles_n = 1;
        constexpr int SH = 3 * (128 * 64 + 64 * 64);                // >= 64 x 68 x 4 epilogue stage
        hipLaunchKernelGGL((gemm_nt_kernel<1, 2>), dim3((unsigned)ceil_div(M, 128)), dim3(256), SH, st, a);
    }
    return launch_status("mit_gemm_nt");
}

namespace {
struct TnPlan {
    int tiles_n, tiles_k, splits, steps_per_split;
};
TnPlan tn_plan(int64_t M, int64_t N, int64_t K) {
    TnPlan p;
    p.tiles_n = (int)ceil_div(N, 128);
    p.tiles_k = (int)ceil_div(K, 128);
    const int64_t steps = ceil_div(M, 32);
    const int64_t tiles = (int64_t)p.tiles_n * p.tiles_k;
    int64_t splits = ceil_div(512, tiles);                          // one round of 2 blocks per CU: the slabs cost HBM traffic
    if (splits > ceil_div(steps, 8)) splits = ceil_div(steps, 8);   // at least 8 K-steps per block
    if (splits < 1) splits = 1;
    p.steps_per_split = (int)ceil_div(steps, splits);
    p.splits = (int)ceil_div(steps, p.steps_per_split);
    return p;
}
}  // namespace

extern "C" size_t diga_mit_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const TnPlan p = tn_plan(M, N, K);
    return (size_t)p.splits * ((size_t)N * (size_t)K + (size_t)N) * sizeof(float);
}

extern "C" int diga_mit_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* dw, float* dbias, float scale,
                                int accumulate, void* workspace, size_t workspace_bytes, int64_t M, int64_t N, int64_t K, void* stream) {
    DIGA_REQUIRE(A && B && dw && workspace && M > 0 && N > 0 && K > 0, DIGA_EINVAL, "mit_gemm_tn: null pointer / empty shape");
    DIGA_REQUIRE(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, DIGA_EINVAL,
                 "mit_gemm_tn: N, K and the leading dimensions must be multiples of 8");
    DIGA_REQUIRE(aligned16(A) && aligned16(B), DIGA_EALIGN, "mit_gemm_tn: operands must be 16-byte aligned");
    DIGA_REQUIRE(M < (1ll << 31), DIGA_EINVAL, "mit_gemm_tn: too many rows");
    const TnPlan p = tn_plan(M, N, K);
    DIGA_REQUIRE(workspace_bytes >= (size_t)p.splits * ((size_t)N * (size_t)K + (size_t)N) * sizeof(float), DIGA_EWORKSPACE,
                 "mit_gemm_tn: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    WgradArgs a;
    a.A = static_cast<const _Float16*>(A); a.lda = lda;
    a.B = static_cast<const _Float16*>(B); a.ldb = ldb;
    a.slab = static_cast<float*>(workspace);
    a.bias_slab = dbias != nullptr ? a.slab + (size_t)p.splits * (size_t)N * (size_t)K : nullptr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K;
    a.tiles_n = p.tiles_n; a.tiles_k = p.tiles_k; a.splits = p.splits; a.steps_per_split = p.steps_per_split;
    ProfScope prof(DIGA_PROF_MIT_WGRAD, st, 2.0 * (double)M * (double)N * (double)K);
    constexpr int SH = 3 * 2 * 32 * 256;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(p.tiles_n * p.tiles_k * p.splits), dim3(256), SH, st, a);
    const int64_t n = N * K;
    // fixed-order sum of the split slabs (32 columns x 8 split phases per block: short latency chains even for 500 splits)
    // (the bias slab, when there is one, by the blocks behind the weight's: one launch)
    hipLaunchKernelGGL((partial_reduce_kernel<0, 32>), dim3((unsigned)(ceil_div(n, 32) + (dbias != nullptr ? ceil_div(N, 32) : 0))), dim3(256), 0, st,
                       a.slab, p.splits, (int)n, dw, dbias, (int)n, scale, accumulate, (const float*)a.bias_slab, dbias != nullptr ? (int)N : 0);
    return launch_status("mit_gemm_tn");
}

namespace {
int colsum_rows_per_block(int64_t M) {                  // >= ~1024 blocks for the big token matrices, 32..512 rows each
    int64_t r = ceil_div(M, 1024);
    r = ceil_div(r, 32) * 32;
    if (r < 32) r = 32;
    if (r > 512) r = 512;
    return (int)r;
}
}  // namespace

extern "C" size_t diga_mit_colsum_workspace_bytes(int64_t M, int64_t C) {
    if (M <= 0 || C <= 0) return 0;
    return (size_t)ceil_div(M, colsum_rows_per_block(M)) * (size_t)C * sizeof(float);
}

extern "C" int diga_mit_colsum(const void* x, int64_t ld, float* out, float scale, int accumulate, void* workspace,
                               size_t workspace_bytes, int64_t M, int64_t C, void* stream) {
    DIGA_REQUIRE(x && out && workspace && M > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0, DIGA_EINVAL, "mit_colsum: bad argument");
    DIGA_REQUIRE(aligned16(x), DIGA_EALIGN, "mit_colsum: x must be 16-byte aligned");
    const int rpb = colsum_rows_per_block(M);
    const int chunks = (int)ceil_div(M, rpb);
    DIGA_REQUIRE(workspace_bytes >= (size_t)chunks * (size_t)C * sizeof(float), DIGA_EWORKSPACE, "mit_colsum: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DIGA_PROF_MIT_MISC, st, 2.0 * (double)M * (double)C);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(chunks), dim3(256), 0, st, static_cast<const _Float16*>(x), ld,
                       static_cast<float*>(workspace), (int)M, (int)C, rpb);
    if (chunks >= 64)
        hipLaunchKernelGGL((partial_reduce_kernel<0, 8>), dim3((unsigned)ceil_div(C, 8)), dim3(256), 0, st, static_cast<const float*>(workspace), chunks,
                           (int)C, out, (float*)nullptr, (int)C, scale, accumulate, (const float*)nullptr, 0);
    else
        hipLaunchKernelGGL((partial_reduce_kernel<0, 32>), dim3((unsigned)ceil_div(C, 32)), dim3(256), 0, st, static_cast<const float*>(workspace), chunks,
                           (int)C, out, (float*)nullptr, (int)C, scale, accumulate, (const float*)nullptr, 0);
    return launch_status("mit_colsum");
}

extern "C" int diga_mit_cast_transpose(const float* w, void* w16, void* wt16, int64_t R, int64_t C, void* stream) {
    DIGA_REQUIRE(w && (w16 || wt16) && R > 0 && C > 0, DIGA_EINVAL, "mit_cast_transpose: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)ceil_div(C, 32), (unsigned)ceil_div(R, 32)), dim3(256), 0, st, w,
                       static_cast<_Float16*>(w16), static_cast<_Float16*>(wt16), (int)R, (int)C);
    return launch_status("mit_cast_transpose");
}

static_assert(sizeof(diga::mit::PrepEntry) == sizeof(diga_mit_weight_prep_t), "diga_mit_weight_prep_t mirrors PrepEntry");

extern "C" int diga_mit_weight_prep_multi(const diga_mit_weight_prep_t* table, const int64_t* tile_start, int64_t n_tensors,
                                          int64_t total_tiles, void* stream) {
    DIGA_REQUIRE(table && tile_start && n_tensors > 0 && total_tiles > 0 && total_tiles < (1ll << 31), DIGA_EINVAL,
                 "mit_weight_prep_multi: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(weight_prep_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st,
                       reinterpret_cast<const PrepEntry*>(table), tile_start, (int)n_tensors);
    return launch_status("mit_weight_prep_multi");
}
