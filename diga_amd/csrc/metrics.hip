// Confusion matrix for mIoU (G5/util/metrics.py:32-44: np.bincount(K*gt + pred) over valid gt).
// Integer atomics only: LDS histogram per block, one 64-bit global add per non-empty bin.
#include "common.h"

namespace diga {

__global__ __launch_bounds__(256) void confusion_kernel(const long long* __restrict__ gt,
                                                        const long long* __restrict__ pred,
                                                        unsigned long long* __restrict__ hist, int64_t n, int K) {
    extern __shared__ unsigned int sh[];
    const int bins = K * K;
    for (int i = threadIdx.x; i < bins; i += 256) sh[i] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const long long g = gt[i], p = pred[i];
        if (g >= 0 && g < K && p >= 0 && p < K) atomicAdd(&sh[(int)g * K + (int)p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

}  // namespace diga

extern "C" int diga_confusion_matrix(const int64_t* gt, const int64_t* pred, int64_t* hist, int64_t n, int64_t K,
                                     void* stream) {
    DIGA_REQUIRE(gt && pred && hist && n >= 0 && K >= 1 && K <= 64, DIGA_EINVAL, "confusion_matrix: bad argument");
    if (n == 0) return DIGA_OK;
    int64_t blocks = diga::ceil_div(n, 256 * 16);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(diga::confusion_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(K * K) * sizeof(unsigned int),
                       (hipStream_t)stream, (const long long*)gt, (const long long*)pred, (unsigned long long*)hist, n,
                       (int)K);
    return diga::launch_status("diga_confusion_matrix");
}

// ---------------------------------------------------------------------------------------------
// Two-scale evaluation (G5/train_DiGA_gta2city_warm_up.py:346-359, G5/evaluate_val.py:73-93): the logits of
// the full-size and of the half-size image are both upsampled (bilinear, align_corners) to label size, the
// element-wise maximum is taken and its argmax is scored against the ground truth.  Fused: every thread
// interpolates the 2*K values of its pixel from the two low-res maps and adds to the confusion matrix; the
// [N,K,H,W] upsampled tensors (159 MB per 1024x2048 image, twice) are never written.
// ---------------------------------------------------------------------------------------------
namespace diga {
__global__ __launch_bounds__(256) void two_scale_confusion_kernel(const float* __restrict__ pa, int ha, int wa,
                                                                  const float* __restrict__ pb, int hb, int wb,
                                                                  const long long* __restrict__ gt,
                                                                  long long* __restrict__ pred_out,
                                                                  unsigned long long* __restrict__ hist, int K, int H,
                                                                  int W, float say, float sax, float sby, float sbx) {
    extern __shared__ unsigned int sh[];
    const int bins = K * K;
    for (int i = threadIdx.x; i < bins; i += 256) sh[i] = 0;
    __syncthreads();
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (X < W && Y < H) {
        int ia, ja, ib, jb;
        float wya, wxa, wyb, wxb;
        bilinear_cell(Y, say, ha, ia, wya);
        bilinear_cell(X, sax, wa, ja, wxa);
        bilinear_cell(Y, sby, hb, ib, wyb);
        bilinear_cell(X, sbx, wb, jb, wxb);
        const int dja = wa > 1 ? 1 : 0, dia = ha > 1 ? wa : 0, djb = wb > 1 ? 1 : 0, dib = hb > 1 ? wb : 0;
        const float* A = pa + ((int64_t)n * K) * ha * wa + (int64_t)ia * wa + ja;
        const float* B = pb + ((int64_t)n * K) * hb * wb + (int64_t)ib * wb + jb;
        float best = -INFINITY;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            const float* p = A + (int64_t)k * ha * wa;
            const float* q = B + (int64_t)k * hb * wb;
            const float va = (1.f - wya) * ((1.f - wxa) * p[0] + wxa * p[dja]) + wya * ((1.f - wxa) * p[dia] + wxa * p[dia + dja]);
            const float vb = (1.f - wyb) * ((1.f - wxb) * q[0] + wxb * q[djb]) + wyb * ((1.f - wxb) * q[dib] + wxb * q[dib + djb]);
            const float v = fmaxf(va, vb);
            if (v > best) {
                best = v;
                arg = k;
            }
        }
        const int64_t o = ((int64_t)n * H + Y) * W + X;
        if (pred_out) pred_out[o] = arg;
        if (gt) {
            const long long g = gt[o];
            if (g >= 0 && g < K) atomicAdd(&sh[(int)g * K + arg], 1u);
        }
    }
    __syncthreads();
    if (hist)
        for (int i = threadIdx.x; i < bins; i += 256)
            if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}
}  // namespace diga

extern "C" int diga_two_scale_confusion(const float* pred_a, int64_t ha, int64_t wa, const float* pred_b, int64_t hb,
                                        int64_t wb, const int64_t* gt, int64_t* pred_out, int64_t* hist, int64_t N,
                                        int64_t K, int64_t H, int64_t W, void* stream) {
    DIGA_REQUIRE(pred_a && pred_b && (gt == nullptr || hist != nullptr) && (pred_out || hist), DIGA_EINVAL,
                 "two_scale_confusion: bad argument");
    DIGA_REQUIRE(N > 0 && K >= 1 && K <= 64 && ha > 0 && wa > 0 && hb > 0 && wb > 0 && H > 0 && W > 0, DIGA_EINVAL,
                 "two_scale_confusion: bad shape");
    dim3 grid((unsigned)diga::ceil_div(W, 64), (unsigned)diga::ceil_div(H, 4), (unsigned)N);
    hipLaunchKernelGGL(diga::two_scale_confusion_kernel, grid, dim3(256), (size_t)(K * K) * sizeof(unsigned int),
                       (hipStream_t)stream, pred_a, (int)ha, (int)wa, pred_b, (int)hb, (int)wb, (const long long*)gt,
                       (long long*)pred_out, (unsigned long long*)hist, (int)K, (int)H, (int)W, diga::ac_scale(ha, H),
                       diga::ac_scale(wa, W), diga::ac_scale(hb, H), diga::ac_scale(wb, W));
    return diga::launch_status("diga_two_scale_confusion");
}
