// OhemCrossEntropy on MI355X (SURVEY section 8f "next" row 4): hard-pixel cross entropy of the Synthia / semi-
// supervised trees of the reference (domain_adaptation/GTA5/util/loss.py:65-122 and its copies).
//
//   p_t   = softmax(score)[target]          over pixels whose target != ignore_label
//   kth   = the min(min_kept, n_valid - 1)-th smallest p_t (0-based, i.e. sorted[k])
//   thr   = max(kth, thresh)
//   loss  = mean of CE over the valid pixels with p_t < thr;   d loss / d score = (softmax - onehot) / n_kept there
//
// The reference sorts all valid probabilities to read one order statistic.  Here the k-th smallest value is found
// exactly by a three-level radix select on the float bits (p_t >= 0, so the unsigned order of the bits is the order
// of the values): 11 + 11 + 10 bit histograms built with integer atomics (order independent => deterministic), each
// followed by a one-block scan that narrows the prefix.  No host synchronisation anywhere; 8 B per pixel of
// workspace (p_t and CE) so the logits are read twice (probabilities, gradient) -- 2 x 76 + 76 (grad) B per pixel
// at C = 19 against the reference's softmax + CE + gather + sort + masked-select chain.
#include "common.h"

namespace diga {

struct OhemState {
    unsigned int n_valid;     // pixels with target != ignore
    unsigned int k;           // remaining 0-based rank inside the current prefix
    unsigned int prefix;      // high bits of the k-th smallest value found so far
    unsigned int n_kept;
    float threshold;
    float loss_sum;
    unsigned int pad[2];
};

constexpr int kOhemBins = 2048;
constexpr float kOhemInvalid = 2.0f;      // stored for ignored pixels: never below any threshold <= 1, never histogrammed

__global__ __launch_bounds__(256) void ohem_clear_kernel(OhemState* st, unsigned int* hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 3 * kOhemBins) hist[i] = 0u;
    if (i == 0) {
        st->n_valid = 0u; st->k = 0u; st->prefix = 0u; st->n_kept = 0u; st->threshold = 0.f; st->loss_sum = 0.f;
    }
}

// pass 1: p_t and CE per pixel, first-level histogram (bits 31..21), valid count
__global__ __launch_bounds__(256) void ohem_prob_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                        float* __restrict__ prob, float* __restrict__ ce,
                                                        unsigned int* __restrict__ hist, OhemState* st, int C, int64_t HW,
                                                        int64_t total, long long ignore) {
    __shared__ unsigned int sh[kOhemBins];
    __shared__ unsigned int nv;
    for (int i = threadIdx.x; i < kOhemBins; i += 256) sh[i] = 0u;
    if (threadIdx.x == 0) nv = 0u;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned int mine = 0u;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const int64_t n = g / HW, p = g - n * HW;
        const long long t = target[g];
        if (t == ignore) {
            prob[g] = kOhemInvalid;
            ce[g] = 0.f;
            continue;
        }
        const float* base = logits + n * C * HW + p;
        float m = base[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, base[(int64_t)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(base[(int64_t)c * HW] - m);
        const int tc = (t >= 0 && t < C) ? (int)t : 0;
        const float xt = base[(int64_t)tc * HW];
        const float lse = m + logf(s);
        const float pt = expf(xt - m) / s;          // softmax(score)[target], as torch evaluates it
        prob[g] = pt;
        ce[g] = lse - xt;
        atomicAdd(&sh[__float_as_uint(pt) >> 21], 1u);
        ++mine;
    }
    atomicAdd(&nv, mine);
    __syncthreads();
    for (int i = threadIdx.x; i < kOhemBins; i += 256)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
    if (threadIdx.x == 0 && nv) atomicAdd(&st->n_valid, nv);
}

// one block: find the bin that holds rank k, append its index to the prefix, reduce k to the rank inside the bin
__global__ __launch_bounds__(256) void ohem_select_kernel(const unsigned int* __restrict__ hist, OhemState* st, int level,
                                                          unsigned int min_kept, float thresh) {
    __shared__ unsigned int part[256];
    const int bins = level == 2 ? 1024 : kOhemBins, per = bins / 256;
    unsigned int k = st->k;
    if (level == 0) {
        const unsigned int nv = st->n_valid;
        k = nv == 0 ? 0u : (min_kept < nv - 1 ? min_kept : nv - 1);
    }
    unsigned int s = 0;
    for (int i = 0; i < per; ++i) s += hist[threadIdx.x * per + i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        int seg = 0;
        for (; seg < 255 && acc + part[seg] <= k; ++seg) acc += part[seg];
        int b = seg * per;
        for (; b < seg * per + per - 1 && acc + hist[b] <= k; ++b) acc += hist[b];
        st->k = k - acc;
        const unsigned int shift = level == 0 ? 21 : level == 1 ? 10 : 0;
        st->prefix |= (unsigned int)b << shift;
        if (level == 2) {
            const float kth = __uint_as_float(st->prefix);
            st->threshold = st->n_valid == 0 ? 0.f : fmaxf(kth, thresh);
        }
    }
}

// histogram of the next digit over the values that share the prefix found so far
__global__ __launch_bounds__(256) void ohem_hist_kernel(const float* __restrict__ prob, unsigned int* __restrict__ hist,
                                                        const OhemState* st, int level, int64_t total) {
    __shared__ unsigned int sh[kOhemBins];
    for (int i = threadIdx.x; i < kOhemBins; i += 256) sh[i] = 0u;
    __syncthreads();
    const unsigned int prefix = st->prefix;
    const unsigned int mask = level == 1 ? 0xFFE00000u : 0xFFFFFC00u;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const unsigned int u = __float_as_uint(prob[g]);
        if ((u & mask) == prefix) atomicAdd(&sh[level == 1 ? (u >> 10) & 2047u : u & 1023u], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kOhemBins; i += 256)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// kept pixels: per-block partial sums of CE (fixed order inside the block) and counts
__global__ __launch_bounds__(256) void ohem_reduce_kernel(const float* __restrict__ prob, const float* __restrict__ ce,
                                                          const OhemState* st, float* __restrict__ psum,
                                                          unsigned int* __restrict__ pcnt, int64_t total) {
    __shared__ float sm[4];
    __shared__ int smi[4];
    const float thr = st->threshold;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float s = 0.f;
    int c = 0;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const bool kept = prob[g] < thr;
        s += kept ? ce[g] : 0.f;
        c += kept ? 1 : 0;
    }
    const float tot = block_sum<4>(s, sm);
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) smi[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        psum[blockIdx.x] = tot;
        pcnt[blockIdx.x] = (unsigned int)(smi[0] + smi[1] + smi[2] + smi[3]);
    }
}

__global__ __launch_bounds__(256) void ohem_finalize_kernel(const float* __restrict__ psum, const unsigned int* __restrict__ pcnt,
                                                            int nblocks, OhemState* st, float* __restrict__ loss) {
    __shared__ double sd[256];
    __shared__ unsigned int sc[256];
    double s = 0.0;
    unsigned int c = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        s += (double)psum[i];
        c += pcnt[i];
    }
    sd[threadIdx.x] = s;
    sc[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0;
        unsigned int tc = 0;
        for (int i = 0; i < 256; ++i) {
            ts += sd[i];
            tc += sc[i];
        }
        st->n_kept = tc;
        st->loss_sum = (float)ts;
        loss[0] = tc == 0 ? __uint_as_float(0x7FC00000u) : (float)(ts / (double)tc);   // mean of nothing is NaN, as torch's
    }
}

__global__ __launch_bounds__(256) void ohem_grad_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                        const float* __restrict__ prob, const OhemState* st,
                                                        float* __restrict__ grad, int C, int64_t HW, int64_t total,
                                                        float grad_scale) {
    const float thr = st->threshold;
    const float gs = st->n_kept == 0 ? 0.f : grad_scale / (float)st->n_kept;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        const int64_t n = g / HW, p = g - n * HW;
        const float* base = logits + n * C * HW + p;
        float* gb = grad + n * C * HW + p;
        if (!(prob[g] < thr)) {
            for (int c = 0; c < C; ++c) gb[(int64_t)c * HW] = 0.f;
            continue;
        }
        float m = base[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, base[(int64_t)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(base[(int64_t)c * HW] - m);
        const float lse = m + logf(s);
        const int tc = (int)target[g];
        for (int c = 0; c < C; ++c)
            gb[(int64_t)c * HW] = (expf(base[(int64_t)c * HW] - lse) - (c == tc ? 1.f : 0.f)) * gs;
    }
}

static int ohem_blocks(int64_t total) {
    int64_t b = ceil_div(total, 256 * 4);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace diga

using namespace diga;

extern "C" size_t diga_ohem_ce_workspace_bytes(int64_t N, int64_t H, int64_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t total = N * H * W;
    return (size_t)total * 2 * sizeof(float) + (size_t)3 * kOhemBins * sizeof(unsigned int) + sizeof(OhemState) +
           (size_t)2 * ohem_blocks(total) * sizeof(float) + 256;
}

extern "C" int diga_ohem_ce_fwd_bwd(const float* logits, const int64_t* target, float* grad, float* loss_out,
                                    void* workspace, size_t workspace_bytes, int64_t N, int64_t C, int64_t H, int64_t W,
                                    int64_t ignore_label, float thresh, int64_t min_kept, float grad_scale, void* stream) {
    DIGA_REQUIRE(logits && target && loss_out && workspace, DIGA_EINVAL, "ohem_ce: null pointer");
    DIGA_REQUIRE(N > 0 && H > 0 && W > 0 && C >= 1 && C <= 1024, DIGA_EINVAL, "ohem_ce: bad shape N=%lld C=%lld H=%lld W=%lld",
                 (long long)N, (long long)C, (long long)H, (long long)W);
    DIGA_REQUIRE(N * H * W < (1ll << 32), DIGA_EINVAL, "ohem_ce: more than 2^32 pixels");
    DIGA_REQUIRE(workspace_bytes >= diga_ohem_ce_workspace_bytes(N, H, W) && aligned16(workspace), DIGA_EWORKSPACE,
                 "ohem_ce: workspace too small or misaligned");
    const int64_t HW = H * W, total = N * HW;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_CE2D, st);
    float* prob = (float*)workspace;
    float* ce = prob + total;
    unsigned int* hist = (unsigned int*)(ce + total);
    OhemState* state = (OhemState*)(hist + 3 * kOhemBins);
    const int nb = ohem_blocks(total);
    float* psum = (float*)(state + 1);
    unsigned int* pcnt = (unsigned int*)(psum + nb);
    const unsigned int mk = (unsigned int)(min_kept < 1 ? 1 : (min_kept > 0xFFFFFFFFll ? 0xFFFFFFFFll : min_kept));
    hipLaunchKernelGGL(ohem_clear_kernel, dim3((3 * kOhemBins + 255) / 256), dim3(256), 0, st, state, hist);
    hipLaunchKernelGGL(ohem_prob_kernel, dim3(nb), dim3(256), 0, st, logits, (const long long*)target, prob, ce, hist, state,
                       (int)C, HW, total, (long long)ignore_label);
    hipLaunchKernelGGL(ohem_select_kernel, dim3(1), dim3(256), 0, st, hist, state, 0, mk, thresh);
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(nb), dim3(256), 0, st, prob, hist + kOhemBins, state, 1, total);
    hipLaunchKernelGGL(ohem_select_kernel, dim3(1), dim3(256), 0, st, hist + kOhemBins, state, 1, mk, thresh);
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(nb), dim3(256), 0, st, prob, hist + 2 * kOhemBins, state, 2, total);
    hipLaunchKernelGGL(ohem_select_kernel, dim3(1), dim3(256), 0, st, hist + 2 * kOhemBins, state, 2, mk, thresh);
    hipLaunchKernelGGL(ohem_reduce_kernel, dim3(nb), dim3(256), 0, st, prob, ce, state, psum, pcnt, total);
    hipLaunchKernelGGL(ohem_finalize_kernel, dim3(1), dim3(256), 0, st, psum, pcnt, nb, state, loss_out);
    if (grad != nullptr)
        hipLaunchKernelGGL(ohem_grad_kernel, dim3(nb), dim3(256), 0, st, logits, (const long long*)target, prob, state, grad, (int)C,
                           HW, total, grad_scale);
    return launch_status("diga_ohem_ce_fwd_bwd");
}
