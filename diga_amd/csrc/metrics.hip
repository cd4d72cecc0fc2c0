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
