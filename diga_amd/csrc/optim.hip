// EMA teacher update and duplicate-aware momentum SGD, multi-tensor (one launch for the whole
// model).  Reference: G5/util/utils.py:103-116 (EMA), torch.optim.SGD driven with the param
// groups of G5/model/model_noaux.py:48-77 (SGD).
//
// Both are pure HBM streams: EMA 12 B/param (read t, read s, write t), SGD 20 B/param (read p,
// g, buf; write p, buf) with the k duplicate micro-steps kept in registers.  A block owns one
// chunk of one tensor; lanes move 16 B each.  Contraction is off so that the fp32 results are
// the reference's separately-rounded multiply/add sequences.
#include "common.h"

#pragma clang fp contract(off)

namespace diga {

__global__ __launch_bounds__(256) void ema_flat_kernel(float* __restrict__ t, const float* __restrict__ s,
                                                       int64_t n4, int64_t n, float a, float b) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 tv = reinterpret_cast<float4*>(t)[i];
        const float4 sv = reinterpret_cast<const float4*>(s)[i];
        tv.x = a * tv.x + b * sv.x;
        tv.y = a * tv.y + b * sv.y;
        tv.z = a * tv.z + b * sv.z;
        tv.w = a * tv.w + b * sv.w;
        reinterpret_cast<float4*>(t)[i] = tv;
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) t[i] = a * t[i] + b * s[i];
    }
}

__global__ __launch_bounds__(256) void ema_multi_kernel(float* const* __restrict__ tp,
                                                        const float* const* __restrict__ sp,
                                                        const int64_t* __restrict__ sizes,
                                                        const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_start,
                                                        int64_t chunk_elems, float a, float b) {
    const int ti = chunk_tensor[blockIdx.x];
    const int64_t start = chunk_start[blockIdx.x];
    float* __restrict__ t = tp[ti];
    const float* __restrict__ s = sp[ti];
    const int64_t size = sizes[ti];
    const int64_t end = start + chunk_elems < size ? start + chunk_elems : size;
    const bool vec = (((uintptr_t)t | (uintptr_t)s) & 15u) == 0 && (start & 3) == 0;
    if (vec) {
        const int64_t n4 = (end - start) >> 2;
        float4* t4 = reinterpret_cast<float4*>(t + start);
        const float4* s4 = reinterpret_cast<const float4*>(s + start);
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            float4 tv = t4[i];
            const float4 sv = s4[i];
            tv.x = a * tv.x + b * sv.x;
            tv.y = a * tv.y + b * sv.y;
            tv.z = a * tv.z + b * sv.z;
            tv.w = a * tv.w + b * sv.w;
            t4[i] = tv;
        }
        for (int64_t i = start + n4 * 4 + threadIdx.x; i < end; i += 256) t[i] = a * t[i] + b * s[i];
    } else {
        for (int64_t i = start + threadIdx.x; i < end; i += 256) t[i] = a * t[i] + b * s[i];
    }
}

__device__ __forceinline__ void sgd_elem(float& p, const float g, float& buf, int k, float lr, float mom, float wd,
                                         bool first) {
    for (int r = 0; r < k; ++r) {
        const float d = g + wd * p;
        buf = first ? d : (mom * buf + d);
        p = p - lr * buf;
    }
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(float* const* __restrict__ pp,
                                                        const float* const* __restrict__ gp,
                                                        float* const* __restrict__ bp,
                                                        const int64_t* __restrict__ sizes,
                                                        const int32_t* __restrict__ mult,
                                                        const float* __restrict__ lrs,
                                                        const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_start,
                                                        int64_t chunk_elems, float mom, float wd, int first,
                                                        float gscale, const int32_t* __restrict__ skip_flag) {
    // skip_flag (nullable): set by diga_nonfinite_flag_f32 when this step's gradients hold inf / NaN (a loss-scaled fp16
    // backward that overflowed): the whole step leaves parameters and momentum untouched, on every block alike
    if (skip_flag != nullptr && skip_flag[0] != 0) return;
    const int ti = chunk_tensor[blockIdx.x];
    const int64_t start = chunk_start[blockIdx.x];
    float* __restrict__ p = pp[ti];
    const float* __restrict__ g = gp[ti];
    float* __restrict__ b = bp[ti];
    const int64_t size = sizes[ti];
    const int k = mult[ti];
    const float lr = lrs[ti];
    const int64_t end = start + chunk_elems < size ? start + chunk_elems : size;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)b) & 15u) == 0 && (start & 3) == 0;
    const bool f = first != 0;
    if (vec) {
        const int64_t n4 = (end - start) >> 2;
        float4* p4 = reinterpret_cast<float4*>(p + start);
        const float4* g4 = reinterpret_cast<const float4*>(g + start);
        float4* b4 = reinterpret_cast<float4*>(b + start);
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            float4 pv = p4[i], bv = b4[i];
            const float4 gv = g4[i];
            sgd_elem(pv.x, gv.x * gscale, bv.x, k, lr, mom, wd, f);
            sgd_elem(pv.y, gv.y * gscale, bv.y, k, lr, mom, wd, f);
            sgd_elem(pv.z, gv.z * gscale, bv.z, k, lr, mom, wd, f);
            sgd_elem(pv.w, gv.w * gscale, bv.w, k, lr, mom, wd, f);
            p4[i] = pv;
            b4[i] = bv;
        }
        for (int64_t i = start + n4 * 4 + threadIdx.x; i < end; i += 256) {
            float pv = p[i], bv = b[i];
            sgd_elem(pv, g[i] * gscale, bv, k, lr, mom, wd, f);
            p[i] = pv;
            b[i] = bv;
        }
    } else {
        for (int64_t i = start + threadIdx.x; i < end; i += 256) {
            float pv = p[i], bv = b[i];
            sgd_elem(pv, g[i] * gscale, bv, k, lr, mom, wd, f);
            p[i] = pv;
            b[i] = bv;
        }
    }
}

// flag[0] = 1 and flag[1] += 1 (once per launch) when x holds an inf or NaN
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, int64_t n, int32_t* __restrict__ flag) {
    const int64_t n4 = n >> 2;
    bool bad = false;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = x4[i];
        // (an exponent of all ones: v - v is NaN for inf and NaN, 0 otherwise)
        const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(t == 0.f);
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) bad |= !((x[i] - x[i]) == 0.f);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) {
        if (atomicExch(&flag[0], 1) == 0) atomicAdd(&flag[1], 1);
    }
}

}  // namespace diga

using namespace diga;

extern "C" int diga_nonfinite_flag_f32(const float* x, int64_t n, int32_t* flag, void* stream) {
    DIGA_REQUIRE(x && flag && n >= 0, DIGA_EINVAL, "nonfinite_flag_f32: bad argument");
    DIGA_REQUIRE(aligned16(x), DIGA_EALIGN, "nonfinite_flag_f32: x must be 16-byte aligned");
    if (n == 0) return DIGA_OK;
    int64_t blocks = ceil_div(n / 4 > 0 ? n / 4 : 1, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    ProfScope prof(DIGA_PROF_ELEMENTWISE, (hipStream_t)stream, (double)n * 4.0);
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, flag);
    return launch_status("diga_nonfinite_flag_f32");
}

extern "C" int diga_ema_update_flat(float* teacher, const float* student, int64_t n, float alpha,
                                    float one_minus_alpha, void* stream) {
    DIGA_REQUIRE(teacher && student && n >= 0, DIGA_EINVAL, "ema_update_flat: bad argument");
    DIGA_REQUIRE(aligned16(teacher) && aligned16(student), DIGA_EALIGN, "ema_update_flat: buffers must be 16-byte aligned");
    if (n == 0) return DIGA_OK;
    const int64_t n4 = n / 4;
    int64_t blocks = ceil_div(n4 > 0 ? n4 : 1, 256);
    if (blocks > 4096) blocks = 4096;
    ProfScope prof(DIGA_PROF_EMA, (hipStream_t)stream);
    hipLaunchKernelGGL(ema_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, teacher, student, n4, n,
                       alpha, one_minus_alpha);
    return launch_status("diga_ema_update_flat");
}

extern "C" int diga_ema_update_multi(float* const* teacher_ptrs, const float* const* student_ptrs,
                                     const int64_t* sizes, const int32_t* chunk_tensor, const int64_t* chunk_start,
                                     int64_t n_chunks, int64_t chunk_elems, float alpha, float one_minus_alpha,
                                     void* stream) {
    DIGA_REQUIRE(teacher_ptrs && student_ptrs && sizes && chunk_tensor && chunk_start, DIGA_EINVAL,
                 "ema_update_multi: null table");
    DIGA_REQUIRE(n_chunks >= 0 && chunk_elems > 0 && (chunk_elems % 4) == 0, DIGA_EINVAL,
                 "ema_update_multi: chunk_elems must be a positive multiple of 4");
    if (n_chunks == 0) return DIGA_OK;
    ProfScope prof(DIGA_PROF_EMA, (hipStream_t)stream);
    hipLaunchKernelGGL(ema_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, teacher_ptrs,
                       student_ptrs, sizes, chunk_tensor, chunk_start, chunk_elems, alpha, one_minus_alpha);
    return launch_status("diga_ema_update_multi");
}

extern "C" int diga_sgd_momentum_multi(float* const* param_ptrs, const float* const* grad_ptrs, float* const* buf_ptrs,
                                       const int64_t* sizes, const int32_t* mult, const float* lr,
                                       const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks,
                                       int64_t chunk_elems, float momentum, float weight_decay, int first_step,
                                       float grad_scale, const int32_t* skip_flag, void* stream) {
    DIGA_REQUIRE(param_ptrs && grad_ptrs && buf_ptrs && sizes && mult && lr && chunk_tensor && chunk_start,
                 DIGA_EINVAL, "sgd_momentum_multi: null table");
    DIGA_REQUIRE(n_chunks >= 0 && chunk_elems > 0 && (chunk_elems % 4) == 0, DIGA_EINVAL,
                 "sgd_momentum_multi: chunk_elems must be a positive multiple of 4");
    if (n_chunks == 0) return DIGA_OK;
    ProfScope prof(DIGA_PROF_SGD, (hipStream_t)stream);
    hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, param_ptrs,
                       grad_ptrs, buf_ptrs, sizes, mult, lr, chunk_tensor, chunk_start, chunk_elems, momentum,
                       weight_decay, first_step, grad_scale, skip_flag);
    return launch_status("diga_sgd_momentum_multi");
}
