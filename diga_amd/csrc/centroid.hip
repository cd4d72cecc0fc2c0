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
// PX pixels per lane (block = 64 * PX pixels): the 20 KB centroid image is staged once per 128 pixels, every centroid
// value read from LDS serves PX pixels, and PX * 4 plane loads are in flight per lane.  Per pixel the arithmetic (channel
// order inside a wave's quarter of D, then the four partials in wave order) does not depend on PX.
template <int K, int PX>
__global__ __launch_bounds__(256) void centroid_weights_kernel(const float* __restrict__ feat,
                                                               const float* __restrict__ cent,
                                                               float* __restrict__ weights,
                                                               float* __restrict__ neg_dist, int D, int Krt,
                                                               int64_t HW) {
    constexpr int KP = (K + 3) & ~3;
    constexpr int BP = 64 * PX;       // pixels per block
    extern __shared__ __align__(16) float smem[];
    float* cT = smem;                 // [D][KP]  transposed centroids
    float* part = smem + (size_t)D * KP;  // [4][K][BP] per-wave partial squared distances
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // coalesced read of the [K][D] centroids (consecutive threads = consecutive d), transposed on the LDS write
    for (int i = threadIdx.x; i < KP * D; i += 256) {
        const int k = i / D, d = i - k * D;
        cT[d * KP + k] = (k < Krt) ? cent[(int64_t)k * D + d] : 0.f;
    }
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * BP + lane;
    const int dper = (D + 3) / 4;
    const int d0 = wv * dper, d1 = (d0 + dper < D) ? d0 + dper : D;
    float acc[PX][K];
#pragma unroll
    for (int u = 0; u < PX; ++u)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[u][k] = 0.f;
    const float* f[PX];
#pragma unroll
    for (int u = 0; u < PX; ++u) f[u] = feat + ((int64_t)n * D) * HW + (p0 + 64 * u < HW ? p0 + 64 * u : 0);
    // 8 channels x PX plane loads in flight per lane: a wave's 64 channels are 8 dependent rounds (4-deep: 16 rounds, and the
    // kernel is nothing but that latency chain: 68.7 MB of features at C4)
#pragma unroll 8
    for (int d = d0; d < d1; ++d) {
        float x[PX];
#pragma unroll
        for (int u = 0; u < PX; ++u) x[u] = f[u][(int64_t)d * HW];
        const float4* c4 = reinterpret_cast<const float4*>(cT + (size_t)d * KP);
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) {
            const float4 c = c4[q];
#pragma unroll
            for (int u = 0; u < PX; ++u) {
                const float e0 = c.x - x[u], e1 = c.y - x[u], e2 = c.z - x[u], e3 = c.w - x[u];
                if (q * 4 + 0 < K) acc[u][q * 4 + 0] += e0 * e0;
                if (q * 4 + 1 < K) acc[u][q * 4 + 1] += e1 * e1;
                if (q * 4 + 2 < K) acc[u][q * 4 + 2] += e2 * e2;
                if (q * 4 + 3 < K) acc[u][q * 4 + 3] += e3 * e3;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PX; ++u)
#pragma unroll
        for (int k = 0; k < K; ++k) part[(wv * K + k) * BP + 64 * u + lane] = acc[u][k];
    __syncthreads();
    // softmax over classes: wave w finishes the pixels lane + 64 * u with u % 4 == w (PX <= 4: one group per wave)
    if (wv >= PX) return;
    const int64_t p = p0 + 64 * wv;
    if (p >= HW) return;
    const int col = 64 * wv + lane;
    float dist[K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = part[(0 * K + k) * BP + col] + part[(1 * K + k) * BP + col] +
                        part[(2 * K + k) * BP + col] + part[(3 * K + k) * BP + col];
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
        float best = o[0];
        int arg = 0;
        for (int k = 1; k < K; ++k) {
            const float v = o[(int64_t)k * hw];
            if (v > best) {
                best = v;
                arg = k;
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

// one wave per (image, channel): lane-private class bins in LDS, then a cross-lane sum per class
__global__ __launch_bounds__(256) void class_sums_kernel(const float* __restrict__ feat,
                                                         const uint8_t* __restrict__ ids, float* __restrict__ sums,
                                                         int D, int K, int64_t hw) {
    __shared__ float bins[4][kKMax + 1][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int d = blockIdx.x * 4 + wv;
    const int n = blockIdx.y;
    for (int k = 0; k <= K; ++k) bins[wv][k][lane] = 0.f;
    if (d < D) {
        const float* f = feat + ((int64_t)n * D + d) * hw;
        const uint8_t* id = ids + (int64_t)n * hw;
        // (8 pixels per round: the loads of a round are issued before its first LDS update -- the loop was one dependent
        //  load -> read-modify-write chain per pixel, 131 rounds at C4; a lane's additions keep their order)
        int64_t p = lane;
        for (; p + 7 * 64 < hw; p += 8 * 64) {
            float v[8];
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u] = id[p + 64 * u];
                v[u] = f[p + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) bins[wv][c[u]][lane] += v[u];
        }
        for (; p < hw; p += 64) bins[wv][id[p]][lane] += f[p];
    }
    __syncthreads();
    if (d < D) {
        for (int k = 0; k < K; ++k) {
            const float t = wave_sum(bins[wv][k][lane]);
            if (lane == 0) sums[((int64_t)n * K + k) * D + d] = t;
        }
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
    // pixels per lane: the smallest PX whose blocks all fit on the chip at once (256 CUs x the blocks the LDS footprint allows per CU) --
    // at C4 (8 x 8385 pixels) PX = 2 gives 528 blocks for 512 slots: a second, almost empty round doubled the kernel's time
    auto pick_px = [&](int kp) {
        for (int px = 1; px <= 4; ++px) {
            const size_t sh = ((size_t)D * kp + (size_t)4 * kp * 64 * px) * sizeof(float);
            const int64_t per_cu = std::min<int64_t>(8, (160 * 1024) / (int64_t)sh);
            if (per_cu >= 1 && ceil_div(HW, 64 * px) * N <= 256 * per_cu) return px;
        }
        return 4;
    };
#define DIGA_CW_LAUNCH(KT_, KP_, PX_)                                                                                        \
    do {                                                                                                                    \
        dim3 grid((unsigned)ceil_div(HW, 64 * PX_), (unsigned)N);                                                           \
        const size_t sh = ((size_t)D * KP_ + (size_t)4 * KT_ * 64 * PX_) * sizeof(float);                                    \
        DIGA_REQUIRE(sh <= 160 * 1024, DIGA_EINVAL, "centroid_softmax_weights: D too large for LDS");                        \
        (void)hipFuncSetAttribute((const void*)centroid_weights_kernel<KT_, PX_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL((centroid_weights_kernel<KT_, PX_>), grid, dim3(256), sh, st, feat, centroids, weights, neg_dist, (int)D,     \
                           (int)K, HW);                                                                                     \
    } while (0)
    if (K <= 19 && K > 16) {
        switch (pick_px(20)) {
            case 1: DIGA_CW_LAUNCH(19, 20, 1); break;
            case 2: DIGA_CW_LAUNCH(19, 20, 2); break;
            case 3: DIGA_CW_LAUNCH(19, 20, 3); break;
            default: DIGA_CW_LAUNCH(19, 20, 4); break;
        }
    } else if (K <= 16) {
        switch (pick_px(16)) {
            case 1: DIGA_CW_LAUNCH(16, 16, 1); break;
            case 2: DIGA_CW_LAUNCH(16, 16, 2); break;
            case 3: DIGA_CW_LAUNCH(16, 16, 3); break;
            default: DIGA_CW_LAUNCH(16, 16, 4); break;
        }
    } else {
        dim3 grid((unsigned)ceil_div(HW, 64), (unsigned)N);
        const size_t sh = ((size_t)D * 32 + 4 * 32 * 64) * sizeof(float);
        DIGA_REQUIRE(sh <= 160 * 1024, DIGA_EINVAL, "centroid_softmax_weights: D*K too large for LDS");
        (void)hipFuncSetAttribute((const void*)centroid_weights_kernel<32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipLaunchKernelGGL((centroid_weights_kernel<32, 1>), grid, dim3(256), sh, st, feat, centroids, weights, neg_dist,
                           (int)D, (int)K, HW);
    }
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
    hipLaunchKernelGGL(argmax_consensus_kernel, grid, dim3(256), (size_t)K * kConsRows * kConsCols * sizeof(float), (hipStream_t)stream, weights,
                       (const long long*)pseudo_in, (long long*)pseudo_out, (long long*)feat_pseudo, (int)K, (int)h,
                       (int)w, (int)H, (int)W, ac_scale(h, H), ac_scale(w, W));
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
    hipLaunchKernelGGL(class_sums_kernel, dim3((unsigned)ceil_div(D, 4), (unsigned)N), dim3(256), 0, st, feat, ids, sums,
                       (int)D, (int)K, hw);
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
