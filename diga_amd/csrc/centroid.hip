// Centroid-based dynamic pseudo-label selection.
// Reference: G5/calc_centroids.py:120-176 (Class_Features) and the bilateral-consensus block at
// G5/train_DiGA_gta2city_self_training.py:298-304,327-341.  The reference walks the 19 classes in
// a Python loop with a [B,256,h,w] temporary each, and issues two .item() syncs per (image,class).
//
// Here the feature map [N,D,hw] is streamed once per kernel:
//   centroid_weights : lanes = pixels (coalesced plane reads), the 4 waves of a block split the D
//                      channels, centroids sit transposed in LDS and are read as broadcasts.
//   class_ids/sums   : argmax+label consensus to a byte map, then one wave per channel
//                      accumulates per-class sums in lane-private LDS bins (no atomics on floats).
//   ema_apply        : the order-dependent centroid update, sequential over images on device.
#include "common.h"

namespace diga {

constexpr int kKMax = 32;

// ------------------------------------------------------------------------------------------
// Round 6.  Block = 64 pixels (one per lane) x 4 waves that split the D channels; what bounds this kernel is NOT bandwidth alone:
//   * arithmetic: (c - x)^2 accumulated per (pixel, channel, class) is 2 VALU operations -- 17.2 M pixel-channels x 19 classes at C4 =
//     10.2 M wave instructions, 40 k cycles per SIMD = 17 us of pure issue.  The classes are therefore processed in PAIRS on the packed
//     fp32 pipe (v_pk_add_f32 / v_pk_fma_f32: a centroid pair from one 8-byte LDS read against the pixel's value broadcast to both
//     halves): 9 us;
//   * latency: a wave's channels are a chain of dependent load rounds -- 16 plane loads per lane and round (round 5: 8 with two pixels
//     per lane), and the FIRST round is issued before the centroids are staged and the block meets at its barrier;
//   * occupancy: the partial distances reuse the centroids' LDS (20 KB per block instead of 40) and the kernel is held to 64 VGPRs:
//     eight blocks per CU = eight waves per SIMD, all 1056 blocks of C4 resident at once.
// Per pixel the arithmetic is round 5's: fma chain in channel order inside a wave's quarter of D, the four partials added in wave order.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int K>
__global__ __launch_bounds__(256, 6) void centroid_weights_kernel(const float* __restrict__ feat,
                                                               const float* __restrict__ cent,
                                                               float* __restrict__ weights,
                                                               float* __restrict__ neg_dist, int D, int Krt,
                                                               int64_t HW) {
    constexpr int KP = (K + 3) & ~3;
    constexpr int U = 16;             // plane loads in flight per lane (32 at five waves per SIMD measured slower: 45.3 vs 40.7 us)
    extern __shared__ __align__(16) float smem[];
    float* cT = smem;                 // [D][KP]  transposed centroids; afterwards [4][K][64] per-wave partial squared distances
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 64 + lane;
    const int dper = (D + 3) / 4;
    const int d0 = wv * dper, d1 = (d0 + dper < D) ? d0 + dper : D;
    const float* f = feat + ((int64_t)n * D + d0) * HW + (p < HW ? p : 0);
    const int nd = d1 > d0 ? d1 - d0 : 0;
    const int full = nd / U;          // rounds of U channels (+ a tail of nd % U)
    float x[U];
    if (full > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = f[(int64_t)u * HW];
    }
    // coalesced read of the [K][D] centroids (consecutive threads = consecutive d), transposed on the LDS write
    for (int i = threadIdx.x; i < KP * D; i += 256) {
        const int k = i / D, d = i - k * D;
        cT[d * KP + k] = (k < Krt) ? cent[(int64_t)k * D + d] : 0.f;
    }
    __syncthreads();
    f32x2 acc[KP / 2];
#pragma unroll
    for (int q = 0; q < KP / 2; ++q) acc[q] = (f32x2){0.f, 0.f};
    auto channel = [&](int d, float xv) {
        const f32x2* c2 = reinterpret_cast<const f32x2*>(cT + (size_t)d * KP);
        const f32x2 xx = (f32x2){xv, xv};
#pragma unroll
        for (int q = 0; q < KP / 2; ++q) {
            const f32x2 e = c2[q] - xx;
            acc[q] = __builtin_elementwise_fma(e, e, acc[q]);
        }
    };
    for (int r = 0; r < full; ++r) {
        if (r > 0) {
            // (no register double-buffering: 16 + 20 live values keep the kernel at eight waves per SIMD, and the other seven waves
            //  cover this round trip -- the double-buffered form needed 112 VGPRs = four waves, one block too few per CU for C4)
            const float* fr = f + (int64_t)r * U * HW;
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = fr[(int64_t)u * HW];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) channel(d0 + r * U + u, x[u]);
    }
    for (int d = d0 + full * U; d < d1; ++d) channel(d, f[(int64_t)(d - d0) * HW]);
    __syncthreads();                  // every wave is done with the centroids: their LDS becomes the partials' [4][K][64]
    float* part = smem;
#pragma unroll
    for (int q = 0; q < KP / 2; ++q) {
        if (2 * q < K) part[(wv * K + 2 * q) * 64 + lane] = acc[q].x;
        if (2 * q + 1 < K) part[(wv * K + 2 * q + 1) * 64 + lane] = acc[q].y;
    }
    __syncthreads();
    if (wv != 0 || p >= HW) return;
    float dist[K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = part[(0 * K + k) * 64 + lane] + part[(1 * K + k) * 64 + lane] +
                        part[(2 * K + k) * 64 + lane] + part[(3 * K + k) * 64 + lane];
        dist[k] = -sqrtf(s);
        if (k < Krt) m = fmaxf(m, dist[k]);
    }
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k < Krt) {
            if (neg_dist) neg_dist[((int64_t)n * Krt + k) * HW + p] = dist[k];
            dist[k] = expf(dist[k] - m);
            z += dist[k];
        }
    }
    const float rz = 1.f / z;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (k < Krt) weights[((int64_t)n * Krt + k) * HW + p] = dist[k] * rz;
}

// ------------------------------------------------------------------------------------------
// Block = 64 x 16 output pixels, four rows per thread (rows Y0 + (tid >> 6) + 4 r).  The low-res weights its bilinear taps touch
// (<= 6 rows x <= 66 columns x K classes for the x8 upsampling of the path) are staged in LDS once -- the 4 x K taps of a pixel were
// 76 gathers through L1 before -- and a thread's four label loads are issued before the staging barrier: the kernel is two
// dependent global round trips per block, so fewer, fatter blocks (4096 instead of 16384 at C4) is what shortens it.
// Falls back to direct reads when the footprint does not fit (small upsampling factors / downsampling geometries).
constexpr int kConsRows = 6, kConsCols = 68, kConsRPT = 4;
__global__ __launch_bounds__(256) void argmax_consensus_kernel(const float* __restrict__ wts,
                                                               const long long* __restrict__ pseudo_in,
                                                               long long* __restrict__ pseudo_out,
                                                               long long* __restrict__ feat_pseudo, int K, int h, int w,
                                                               int H, int W, float sy, float sx) {
    extern __shared__ float tile[];        // [K][rows][cols]
    const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * (4 * kConsRPT);
    const int X = X0 + (threadIdx.x & 63);
    const int n = blockIdx.z;
    long long pin[kConsRPT];
#pragma unroll
    for (int r = 0; r < kConsRPT; ++r) {
        const int Y = Y0 + (threadIdx.x >> 6) + 4 * r;
        pin[r] = (X < W && Y < H) ? pseudo_in[((int64_t)n * H + Y) * W + X] : 0;
    }
    // footprint of the block (uniform): taps of the first and last pixel row / column
    int ia, ib, ja, jb;
    float t_;
    bilinear_cell(Y0, sy, h, ia, t_);
    bilinear_cell(min(Y0 + 4 * kConsRPT - 1, H - 1), sy, h, ib, t_);
    bilinear_cell(X0, sx, w, ja, t_);
    bilinear_cell(min(X0 + 63, W - 1), sx, w, jb, t_);
    const int rows = min(ib + 1, h - 1) - ia + 1, cols = min(jb + 1, w - 1) - ja + 1;
    const bool staged = rows <= kConsRows && cols <= kConsCols;
    const float* base_n = wts + ((int64_t)n * K) * h * w;
    if (staged) {
        const int per = rows * cols;
        for (int i = threadIdx.x; i < K * per; i += 256) {
            const int k = i / per, r = (i - k * per) / cols, c = i - k * per - r * cols;
            tile[i] = base_n[(int64_t)k * h * w + (int64_t)(ia + r) * w + ja + c];
        }
        __syncthreads();
    }
    if (X >= W) return;
    int j0;
    float wx;
    bilinear_cell(X, sx, w, j0, wx);
#pragma unroll
    for (int r = 0; r < kConsRPT; ++r) {
        const int Y = Y0 + (threadIdx.x >> 6) + 4 * r;
        if (Y >= H) continue;
        int i0;
        float wy;
        bilinear_cell(Y, sy, h, i0, wy);
        float best = -INFINITY;
        int arg = 0;
        if (staged) {
            const int dj = (w > 1) ? 1 : 0, di = (h > 1) ? cols : 0;
            const float* t0 = tile + (i0 - ia) * cols + (j0 - ja);
            const int per = rows * cols;
            for (int k = 0; k < K; ++k) {
                const float* p = t0 + k * per;
                const float v = (1.f - wy) * ((1.f - wx) * p[0] + wx * p[dj]) + wy * ((1.f - wx) * p[di] + wx * p[di + dj]);
                if (v > best) {  // strict: first maximum wins, as torch.max
                    best = v;
                    arg = k;
                }
            }
        } else {
            const int dj = (w > 1) ? 1 : 0, di = (h > 1) ? w : 0;
            const float* base = base_n + (int64_t)i0 * w + j0;
            for (int k = 0; k < K; ++k) {
                const float* p = base + (int64_t)k * h * w;
                const float v = (1.f - wy) * ((1.f - wx) * p[0] + wx * p[dj]) + wy * ((1.f - wx) * p[di] + wx * p[di + dj]);
                if (v > best) {
                    best = v;
                    arg = k;
                }
            }
        }
        const int64_t o = ((int64_t)n * H + Y) * W + X;
        pseudo_out[o] = (pin[r] == (long long)arg) ? pin[r] : (long long)DIGA_IGNORE_LABEL;
        if (feat_pseudo) feat_pseudo[o] = (long long)arg;
    }
}

// ------------------------------------------------------------------------------------------
// Round 6: the consensus kernel for upsampling factors >= 4 (the path's x8), built for what bounds it.  Round 5's kernel above spends
// ~250 VALU + 76 LDS instructions per output pixel (19 classes x 4 taps read one float at a time, 7 scalar flops each): 4.2 M pixels at
// C4 = 26 us of issue alone, next to 72 MB of traffic worth 12 us -- it is ALU-bound, not bandwidth-bound.  Here:
//   * the staged low-res weights lie CLASS-FASTEST in LDS ([row][col][KP]): a tap's 20 classes are five 16-byte reads, not 19 reads;
//   * two classes per instruction on the packed fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32): 6 packed operations per class pair;
//   * int64 labels travel as 16-byte pairs: a thread owns two horizontally adjacent pixels on two rows (block = 128 x 8 pixels).
// Same expression tree per class as torch's upsample_bilinear2d, h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d), with the inner and the
// outer sum contracted to fma; strict first-maximum argmax.
constexpr int kCons8Rows = 6, kCons8Cols = 36;
template <int K>
__global__ __launch_bounds__(256) void argmax_consensus_pairs_kernel(const float* __restrict__ wts, const long long* __restrict__ pseudo_in,
                                                                     long long* __restrict__ pseudo_out, long long* __restrict__ feat_pseudo,
                                                                     int h, int w, int H, int W, float sy, float sx) {
    constexpr int KP = (K + 3) & ~3;
    __shared__ __align__(16) float tile[kCons8Rows * kCons8Cols * KP];          // [row][col][KP]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int X0 = blockIdx.x * 128, Y0 = blockIdx.y * 8;
    const int X = X0 + 2 * lane;
    const int n = blockIdx.z;
    longlong2 pin[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int Y = Y0 + wv + 4 * r;
        pin[r] = (X < W && Y < H) ? *reinterpret_cast<const longlong2*>(pseudo_in + ((int64_t)n * H + Y) * W + X) : make_longlong2(0, 0);
    }
    int ia, ib, ja, jb;
    float t_;
    bilinear_cell(Y0, sy, h, ia, t_);
    bilinear_cell(min(Y0 + 7, H - 1), sy, h, ib, t_);
    bilinear_cell(X0, sx, w, ja, t_);
    bilinear_cell(min(X0 + 127, W - 1), sx, w, jb, t_);
    const int rows = min(ib + 1, h - 1) - ia + 1, cols = min(jb + 1, w - 1) - ja + 1;      // <= kCons8Rows x kCons8Cols (checked by the host)
    const float* base_n = wts + ((int64_t)n * K) * h * w;
    const int per = rows * cols;
    for (int i = threadIdx.x; i < KP * per; i += 256) {
        const int k = i / per, rc = i - k * per;
        const int r = rc / cols, c = rc - r * cols;
        tile[rc * KP + k] = k < K ? base_n[(int64_t)k * h * w + (int64_t)(ia + r) * w + ja + c] : -INFINITY;
    }
    __syncthreads();
    if (X >= W) return;
    int j0[2];
    float wx[2];
    bilinear_cell(X, sx, w, j0[0], wx[0]);
    bilinear_cell(X + 1, sx, w, j0[1], wx[1]);
    const int dj = (w > 1) ? KP : 0, di = (h > 1) ? cols * KP : 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int Y = Y0 + wv + 4 * r;
        if (Y >= H) continue;
        int i0;
        float wy;
        bilinear_cell(Y, sy, h, i0, wy);
        long long res[2], arg64[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float* t0 = tile + ((i0 - ia) * cols + (j0[e] - ja)) * KP;
            const f32x2 w0 = (f32x2){1.f - wx[e], 1.f - wx[e]}, w1 = (f32x2){wx[e], wx[e]};
            const f32x2 h0 = (f32x2){1.f - wy, 1.f - wy}, h1 = (f32x2){wy, wy};
            float best = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int q = 0; q < KP / 4; ++q) {
                const float4 a4 = *reinterpret_cast<const float4*>(t0 + 4 * q), b4 = *reinterpret_cast<const float4*>(t0 + dj + 4 * q);
                const float4 c4 = *reinterpret_cast<const float4*>(t0 + di + 4 * q), d4 = *reinterpret_cast<const float4*>(t0 + di + dj + 4 * q);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2 a = hh ? (f32x2){a4.z, a4.w} : (f32x2){a4.x, a4.y}, b = hh ? (f32x2){b4.z, b4.w} : (f32x2){b4.x, b4.y};
                    const f32x2 c = hh ? (f32x2){c4.z, c4.w} : (f32x2){c4.x, c4.y}, d = hh ? (f32x2){d4.z, d4.w} : (f32x2){d4.x, d4.y};
                    const f32x2 top = __builtin_elementwise_fma(w1, b, w0 * a), bot = __builtin_elementwise_fma(w1, d, w0 * c);
                    const f32x2 v = __builtin_elementwise_fma(h1, bot, h0 * top);
                    const int k0 = 4 * q + 2 * hh;
                    if (k0 < K && v.x > best) {            // strict: first maximum wins, as torch.max (padding classes hold -inf)
                        best = v.x;
                        arg = k0;
                    }
                    if (k0 + 1 < K && v.y > best) {
                        best = v.y;
                        arg = k0 + 1;
                    }
                }
            }
            const long long pl = e ? pin[r].y : pin[r].x;
            res[e] = (pl == (long long)arg) ? pl : (long long)DIGA_IGNORE_LABEL;
            arg64[e] = (long long)arg;
        }
        const int64_t o = ((int64_t)n * H + Y) * W + X;
        *reinterpret_cast<longlong2*>(pseudo_out + o) = make_longlong2(res[0], res[1]);
        if (feat_pseudo) *reinterpret_cast<longlong2*>(feat_pseudo + o) = make_longlong2(arg64[0], arg64[1]);
    }
}

// ------------------------------------------------------------------------------------------
// ids[n,p] = argmax_k out[n,k,p] if it agrees with the label (or no labels), else K (dead bucket)
__global__ __launch_bounds__(256) void class_ids_kernel(const float* __restrict__ out,
                                                        const float* __restrict__ labels_lr,
                                                        const long long* __restrict__ labels_full,
                                                        uint8_t* __restrict__ ids, int32_t* __restrict__ counts, int K,
                                                        int h, int w, int H, int W, float ry, float rx) {
    __shared__ int cnt[kKMax + 1];
    if (threadIdx.x <= kKMax) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int n = blockIdx.y;
    const int64_t hw = (int64_t)h * w;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < hw) {
        const float* o = out + ((int64_t)n * K) * hw + p;
        // (round 6: the K plane loads of a pixel in flight together, eight at a time -- the run-time-K loop issued them one by one,
        //  a chain of K dependent round trips for a 5 MB pass)
        float best = -INFINITY;
        int arg = 0;
        for (int k0 = 0; k0 < K; k0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (k0 + u < K) ? o[(int64_t)(k0 + u) * hw] : -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < K && (v[u] > best || (k0 + u == 0))) {      // strict: first maximum wins (k = 0 starts the chain, NaN-safe as before)
                    best = v[u];
                    arg = k0 + u;
                }
            }
        }
        int id = arg;
        if (labels_lr != nullptr) {
            const float l = labels_lr[(int64_t)n * hw + p];
            id = (l == (float)arg) ? arg : K;
        } else if (labels_full != nullptr) {
            // F.interpolate(mode='nearest'): src = min(floor(dst * (in/out)), in-1), float arithmetic
            const int y = (int)(p / w), x = (int)(p - (int64_t)y * w);
            int sy_ = (int)floorf((float)y * ry), sx_ = (int)floorf((float)x * rx);
            sy_ = sy_ < H - 1 ? sy_ : H - 1;
            sx_ = sx_ < W - 1 ? sx_ : W - 1;
            const long long l = labels_full[((int64_t)n * H + sy_) * W + sx_];
            id = (l == (long long)arg) ? arg : K;
        }
        ids[(int64_t)n * hw + p] = (uint8_t)id;
        atomicAdd(&cnt[id], 1);
    }
    __syncthreads();
    if (threadIdx.x < K && cnt[threadIdx.x]) atomicAdd(&counts[(int64_t)n * K + threadIdx.x], cnt[threadIdx.x]);
}

// one block per (image, channel): its four waves take a quarter of the plane each (round 5: one wave per plane, four planes per block --
// 2048 waves for the whole chip, each a chain of 131 dependent load rounds at C4), 16 plane loads + 16 id loads in flight per lane and
// round, lane-private class bins in LDS (no float atomics), then a cross-wave + cross-lane sum per class.  A lane's additions keep their
// pixel order; the plane's total is (lanes of quarter 0) + ... + (lanes of quarter 3) per class, a fixed order.
__global__ __launch_bounds__(256) void class_sums_kernel(const float* __restrict__ feat,
                                                         const uint8_t* __restrict__ ids, float* __restrict__ sums,
                                                         int D, int K, int64_t hw) {
    extern __shared__ float bins_raw[];           // [4][K + 1][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int d = blockIdx.x;
    const int n = blockIdx.y;
    float* bins = bins_raw + (size_t)wv * (K + 1) * 64;
    for (int k = 0; k <= K; ++k) bins[k * 64 + lane] = 0.f;
    const float* f = feat + ((int64_t)n * D + d) * hw;
    const uint8_t* id = ids + (int64_t)n * hw;
    const int64_t quarter = ((hw + 255) / 256) * 64;
    const int64_t end = (wv + 1) * quarter < hw ? (wv + 1) * quarter : hw;
    int64_t p = wv * quarter + lane;
    constexpr int U = 16;
    for (; p + (U - 1) * 64 < end; p += U * 64) {
        float v[U];
        int c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c[u] = id[p + 64 * u];
            v[u] = f[p + 64 * u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) bins[c[u] * 64 + lane] += v[u];
    }
    for (; p < end; p += 64) bins[id[p] * 64 + lane] += f[p];
    __syncthreads();
    for (int k = wv; k < K; k += 4) {
        const float* b = bins_raw + (size_t)k * 64 + lane;
        const size_t ws = (size_t)(K + 1) * 64;
        const float t = wave_sum(((b[0] + b[ws]) + b[2 * ws]) + b[3 * ws]);
        if (lane == 0) sums[((int64_t)n * K + k) * D + d] = t;
    }
}

// block = one class; threads = feature channels; sequential over images (order matters)
__global__ __launch_bounds__(1024) void centroid_apply_kernel(float* __restrict__ cent, float* __restrict__ nums,
                                                              const float* __restrict__ sums,
                                                              const int32_t* __restrict__ counts, int N, int K, int D,
                                                              float hw, float momentum, int min_pixels, int mode) {
    __shared__ float sm[16];
    __shared__ float vsum_s;
    const int k = blockIdx.x;
    const int d = threadIdx.x;
    float c = (d < D) ? cent[(int64_t)k * D + d] : 0.f;
    float num = nums[k];
    for (int n = 0; n < N; ++n) {
        const int cnt = counts[(int64_t)n * K + k];
        if (cnt == 0 || cnt < min_pixels) continue;  // uniform over the block
        float v = 0.f;
        if (d < D) {
            // adaptive_avg_pool2d(feat*mask,1) / adaptive_avg_pool2d(mask,1)
            const float mean_fm = sums[((int64_t)n * K + k) * D + d] / hw;
            const float mean_m = (float)cnt / hw;
            v = mean_fm / mean_m;
        }
        float t = wave_sum(v);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += sm[i];
            vsum_s = tot;
        }
        __syncthreads();
        const float vsum = vsum_s;
        __syncthreads();
        if (vsum == 0.f) continue;
        if (mode == 0) {
            c = c * (1.f - momentum) + momentum * v;
            num = fminf(num + 1.f, 3000.f);
        } else {
            c = c * num + v;
            num = num + 1.f;
            c = c / num;
            num = fminf(num, 3000.f);
        }
    }
    if (d < D) cent[(int64_t)k * D + d] = c;
    if (threadIdx.x == 0) nums[k] = num;
}

}  // namespace diga

using namespace diga;

extern "C" int diga_centroid_softmax_weights(const float* feat, const float* centroids, float* weights,
                                             float* neg_dist, int64_t N, int64_t D, int64_t K, int64_t HW,
                                             void* stream) {
    DIGA_REQUIRE(feat && centroids && weights, DIGA_EINVAL, "centroid_softmax_weights: null pointer");
    DIGA_REQUIRE(N > 0 && D > 0 && D <= 1024 && K >= 1 && K <= kKMax && HW > 0, DIGA_EINVAL,
                 "centroid_softmax_weights: bad shape N=%lld D=%lld K=%lld HW=%lld", (long long)N, (long long)D,
                 (long long)K, (long long)HW);
    hipStream_t st = (hipStream_t)stream;
    // D*4 B of features per low-res pixel in, K*4 B of weights out (+ K*4 with distances)
    ProfScope prof(DIGA_PROF_CENTROID_WEIGHTS, st, (double)N * HW * (D * 4.0 + K * 4.0 * (neg_dist ? 2.0 : 1.0)));
#define DIGA_CW_LAUNCH(KT_, KP_)                                                                                             \
    do {                                                                                                                    \
        dim3 grid((unsigned)ceil_div(HW, 64), (unsigned)N);                                                                 \
        const size_t sh = std::max((size_t)D * KP_, (size_t)4 * KT_ * 64) * sizeof(float);                                   \
        DIGA_REQUIRE(sh <= 160 * 1024, DIGA_EINVAL, "centroid_softmax_weights: D too large for LDS");                        \
        (void)hipFuncSetAttribute((const void*)centroid_weights_kernel<KT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL((centroid_weights_kernel<KT_>), grid, dim3(256), sh, st, feat, centroids, weights, neg_dist, (int)D,     \
                           (int)K, HW);                                                                                     \
    } while (0)
    if (K <= 19 && K > 16) DIGA_CW_LAUNCH(19, 20);
    else if (K <= 16) DIGA_CW_LAUNCH(16, 16);
    else DIGA_CW_LAUNCH(32, 32);
#undef DIGA_CW_LAUNCH
    return launch_status("diga_centroid_softmax_weights");
}

extern "C" int diga_upsample_argmax_consensus(const float* weights, const int64_t* pseudo_in, int64_t* pseudo_out,
                                              int64_t* feat_pseudo, int64_t N, int64_t K, int64_t h, int64_t w,
                                              int64_t H, int64_t W, void* stream) {
    DIGA_REQUIRE(weights && pseudo_in && pseudo_out, DIGA_EINVAL, "upsample_argmax_consensus: null pointer");
    DIGA_REQUIRE(N > 0 && K >= 1 && h > 0 && w > 0 && H > 0 && W > 0, DIGA_EINVAL, "upsample_argmax_consensus: bad shape");
    dim3 grid((unsigned)ceil_div(W, 64), (unsigned)ceil_div(H, 4 * kConsRPT), (unsigned)N);
    // low-res weights in, int64 label map read and written [, second map written]
    ProfScope prof(DIGA_PROF_CONSENSUS, (hipStream_t)stream,
                   (double)N * (h * w * K * 4.0 + (double)H * W * (feat_pseudo ? 24.0 : 16.0)));
    const float sy = ac_scale(h, H), sx = ac_scale(w, W);
    // the pair kernel: even W, 16-byte aligned label maps, and an upsampling factor that keeps a 128 x 8 block's taps inside its LDS tile
    // (8 rows span <= 7 * sy + 2 low-res rows, 128 columns <= 127 * sx + 2 columns)
    const bool pairs = (K == 19 || K == 16) && W % 2 == 0 && W >= 2 && h >= 2 && w >= 2 && aligned16(pseudo_in) && aligned16(pseudo_out) &&
                       (!feat_pseudo || aligned16(feat_pseudo)) && 7.f * sy + 3.f <= (float)kCons8Rows && 127.f * sx + 3.f <= (float)kCons8Cols;
    if (pairs) {
        dim3 g2((unsigned)ceil_div(W, 128), (unsigned)ceil_div(H, 8), (unsigned)N);
        if (K == 19)
            hipLaunchKernelGGL((argmax_consensus_pairs_kernel<19>), g2, dim3(256), 0, (hipStream_t)stream, weights, (const long long*)pseudo_in,
                               (long long*)pseudo_out, (long long*)feat_pseudo, (int)h, (int)w, (int)H, (int)W, sy, sx);
        else
            hipLaunchKernelGGL((argmax_consensus_pairs_kernel<16>), g2, dim3(256), 0, (hipStream_t)stream, weights, (const long long*)pseudo_in,
                               (long long*)pseudo_out, (long long*)feat_pseudo, (int)h, (int)w, (int)H, (int)W, sy, sx);
        return launch_status("diga_upsample_argmax_consensus");
    }
    hipLaunchKernelGGL(argmax_consensus_kernel, grid, dim3(256), (size_t)K * kConsRows * kConsCols * sizeof(float), (hipStream_t)stream, weights,
                       (const long long*)pseudo_in, (long long*)pseudo_out, (long long*)feat_pseudo, (int)K, (int)h,
                       (int)w, (int)H, (int)W, sy, sx);
    return launch_status("diga_upsample_argmax_consensus");
}

extern "C" size_t diga_class_mean_workspace_bytes(int64_t N, int64_t hw) {
    return (size_t)((N * hw + 255) / 256) * 256;
}

extern "C" int diga_class_mean_vectors(const float* feat, const float* out, const float* labels_lr,
                                       const int64_t* labels_full, float* sums, int32_t* counts, void* workspace,
                                       size_t workspace_bytes, int64_t N, int64_t D, int64_t K, int64_t h, int64_t w,
                                       int64_t H, int64_t W, void* stream) {
    DIGA_REQUIRE(feat && out && sums && counts && workspace, DIGA_EINVAL, "class_mean_vectors: null pointer");
    DIGA_REQUIRE(N > 0 && D > 0 && K >= 1 && K <= kKMax && h > 0 && w > 0, DIGA_EINVAL, "class_mean_vectors: bad shape");
    DIGA_REQUIRE(!(labels_lr && labels_full), DIGA_EINVAL, "class_mean_vectors: give labels at one resolution only");
    DIGA_REQUIRE(!labels_full || (H > 0 && W > 0), DIGA_EINVAL, "class_mean_vectors: full-res labels need H,W");
    const int64_t hw = h * w;
    DIGA_REQUIRE(workspace_bytes >= diga_class_mean_workspace_bytes(N, hw), DIGA_EWORKSPACE,
                 "class_mean_vectors: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    uint8_t* ids = (uint8_t*)workspace;
    // features D*4 + logits K*4 per low-res pixel + the labels the ids are drawn from
    ProfScope prof(DIGA_PROF_CLASS_MEANS, st, (double)N * hw * (D * 4.0 + K * 4.0 + (labels_full ? 8.0 : 4.0)));
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)N * K * sizeof(int32_t), st);
    DIGA_REQUIRE(e == hipSuccess, (int)e, "class_mean_vectors: memset failed: %s", hipGetErrorString(e));
    const float ry = labels_full ? (float)H / (float)h : 0.f, rx = labels_full ? (float)W / (float)w : 0.f;
    hipLaunchKernelGGL(class_ids_kernel, dim3((unsigned)ceil_div(hw, 256), (unsigned)N), dim3(256), 0, st, out, labels_lr,
                       (const long long*)labels_full, ids, counts, (int)K, (int)h, (int)w, (int)H, (int)W, ry, rx);
    hipLaunchKernelGGL(class_sums_kernel, dim3((unsigned)D, (unsigned)N), dim3(256), (size_t)4 * (K + 1) * 64 * sizeof(float), st, feat, ids,
                       sums, (int)D, (int)K, hw);
    return launch_status("diga_class_mean_vectors");
}

extern "C" int diga_centroid_ema_apply(float* centroids, float* nums, const float* sums, const int32_t* counts,
                                       int64_t N, int64_t K, int64_t D, int64_t hw, float momentum, int min_pixels,
                                       int mode, void* stream) {
    DIGA_REQUIRE(centroids && nums && sums && counts, DIGA_EINVAL, "centroid_ema_apply: null pointer");
    DIGA_REQUIRE(N > 0 && K >= 1 && D >= 1 && D <= 1024 && hw > 0 && (mode == 0 || mode == 1), DIGA_EINVAL,
                 "centroid_ema_apply: bad argument (D <= 1024, mode in {0,1})");
    const int threads = (int)(ceil_div(D, 64) * 64);
    ProfScope prof(DIGA_PROF_CENTROID_APPLY, (hipStream_t)stream, (double)K * D * (N * 4.0 + 8.0));
    hipLaunchKernelGGL(centroid_apply_kernel, dim3((unsigned)K), dim3(threads), 0, (hipStream_t)stream, centroids, nums,
                       sums, counts, (int)N, (int)K, (int)D, (float)hw, momentum, min_pixels, mode);
    return launch_status("diga_centroid_ema_apply");
}
