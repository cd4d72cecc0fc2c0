// Shared host/device helpers for libdiga_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/diga_hip.h"

namespace diga {

void set_error(const char* fmt, ...);
int launch_status(const char* what);  // hipGetLastError() -> 0 or hipError_t, records the string

#define DIGA_REQUIRE(cond, code, ...)     \
    do {                                  \
        if (!(cond)) {                    \
            ::diga::set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

// Brackets the launches of one entry point with HIP events when diga_prof_enable(1) is on.  `work` = the ALGORITHMIC
// work of the call (bytes for the HBM-bound families, FLOPs for the convolutions: what the roofline divides by).
struct ProfScope {
    ProfScope(int tag, hipStream_t st, double work = 0.0);
    ~ProfScope();
    hipEvent_t stop_ = nullptr;
    hipStream_t st_;
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a block of NWAVES*64 threads; result valid in thread 0.  smem: NWAVES floats.
template <int NWAVES>
__device__ __forceinline__ float block_sum(float v, float* smem) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NWAVES; ++i) r += smem[i];
    }
    __syncthreads();
    return r;
}

// align_corners=True bilinear source tap, cell form: value = (1-w1)*x[i0] + w1*x[i0+1] with
// i0 in [0, n_in-2].  Same float arithmetic as torch (src = scale*dst, scale = (in-1)/(out-1)).
__device__ __forceinline__ void bilinear_cell(int dst, float scale, int n_in, int& i0, float& w1) {
    const float src = scale * (float)dst;
    int i = (int)src;
    if (i > n_in - 2) i = n_in - 2;
    if (i < 0) i = 0;
    float w = src - (float)i;
    i0 = i;
    w1 = w > 1.f ? 1.f : w;
}

static inline float ac_scale(int64_t n_in, int64_t n_out) {
    return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
}

}  // namespace diga
