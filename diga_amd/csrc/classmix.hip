// ClassMix / cross-domain mixture: label histogram and label-conditioned paste.
// Reference: the inline blocks at G5/train_DiGA_gta2city_warm_up.py:240-259 and
// G5/train_DiGA_gta2city_self_training.py:259-275,306-325 (per image torch.unique -> host
// random.sample -> ~10 masked assignments per image -> per-image blend).
//
// Here: one histogram launch (8 B/px, integer atomics only -> deterministic), ONE small D->H copy
// of the [B,256] presence table for the host's random.sample (bit-compatible class choice), and
// one paste launch (44 B/px; +16 B/px when labels are pasted too) driven by a [B,256] byte LUT.
#include "common.h"

namespace diga {

__global__ __launch_bounds__(256) void label_hist256_kernel(const long long* __restrict__ labels,
                                                            uint32_t* __restrict__ hist, int64_t HW) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    const long long* lab = labels + (int64_t)b * HW;
    // Round 6: labels as 16-byte PAIRS (two int64 per load, four loads = 64 B per lane in flight per round; round 5: eight 8-byte loads),
    // a wave's pairs are 1 KB contiguous.  An odd HW leaves one label for the tail.
    const int64_t pairs = HW >> 1;
    const int64_t stride = (int64_t)gridDim.x * 256;
    auto count = [&](long long v) {
        bool todo = (v >= 0 && v < 256);
        // wave-aggregated increment: one LDS atomic per distinct value per wave
        while (true) {
            const unsigned long long pending = __ballot(todo);
            if (pending == 0ull) break;
            const int leader = __ffsll((long long)pending) - 1;
            const int lv = __shfl((int)v, leader, 64);
            const bool same = todo && ((int)v == lv);
            const unsigned long long grp = __ballot(same);
            if ((threadIdx.x & 63) == leader) atomicAdd(&sh[lv], (uint32_t)__popcll(grp));
            todo = todo && !same;
        }
    };
    const bool aligned = (reinterpret_cast<uintptr_t>(lab) & 15u) == 0;
    if (aligned) {
        const longlong2* lab2 = reinterpret_cast<const longlong2*>(lab);
        for (int64_t base = (int64_t)blockIdx.x * 256; base < pairs; base += 4 * stride) {
            longlong2 vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * stride + threadIdx.x;
                vv[u] = i < pairs ? lab2[i] : make_longlong2(-1, -1);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                count(vv[u].x);
                count(vv[u].y);
            }
        }
        if ((HW & 1) && blockIdx.x == 0) count(threadIdx.x == 0 ? lab[HW - 1] : -1);
    } else {
        for (int64_t base = (int64_t)blockIdx.x * 256; base < HW; base += stride) {
            const int64_t i = base + threadIdx.x;
            count(i < HW ? lab[i] : -1);
        }
    }
    __syncthreads();
    const uint32_t c = sh[threadIdx.x];
    if (c) atomicAdd(&hist[(int64_t)b * 256 + threadIdx.x], c);
}

template <int VEC>
__global__ __launch_bounds__(256) void classmix_paste_kernel(const float* __restrict__ bg, const float* __restrict__ fg,
                                                             const long long* __restrict__ labels,
                                                             const uint8_t* __restrict__ lut, float* __restrict__ out,
                                                             const long long* __restrict__ bg_labels,
                                                             long long* __restrict__ labels_out, int CH, int64_t HW) {
    __shared__ uint8_t sl[256];
    const int b = blockIdx.y;
    sl[threadIdx.x] = lut[(int64_t)b * 256 + threadIdx.x];
    __syncthreads();
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p >= HW) return;
    bool take[VEC];
    long long lv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        lv[v] = labels[(int64_t)b * HW + p + v];
        take[v] = (lv[v] >= 0 && lv[v] < 256) ? (sl[lv[v]] != 0) : false;
    }
    for (int c = 0; c < CH; ++c) {
        const int64_t off = ((int64_t)b * CH + c) * HW + p;
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(bg + off);
            const float4 f = *reinterpret_cast<const float4*>(fg + off);
            float4 o;
            o.x = take[0] ? f.x : a.x;
            o.y = take[1] ? f.y : a.y;
            o.z = take[2] ? f.z : a.z;
            o.w = take[3] ? f.w : a.w;
            *reinterpret_cast<float4*>(out + off) = o;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) out[off + v] = take[v] ? fg[off + v] : bg[off + v];
        }
    }
    if (labels_out != nullptr) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t o = (int64_t)b * HW + p + v;
            labels_out[o] = take[v] ? lv[v] : bg_labels[o];
        }
    }
}

}  // namespace diga

using namespace diga;

extern "C" int diga_label_hist256(const int64_t* labels, uint32_t* hist, int64_t B, int64_t HW, void* stream) {
    DIGA_REQUIRE(labels && hist && B > 0 && HW > 0, DIGA_EINVAL, "label_hist256: bad argument");
    int64_t bx = ceil_div(HW, 256 * 8);  // ~8 grid-stride rounds per block
    if (bx < 1) bx = 1;
    if (bx > 1024) bx = 1024;
    ProfScope prof(DIGA_PROF_CLASSMIX_HIST, (hipStream_t)stream, (double)B * HW * 8.0);
    hipLaunchKernelGGL(label_hist256_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)labels, hist, HW);
    return launch_status("diga_label_hist256");
}

extern "C" int diga_classmix_paste(const float* bg, const float* fg, const int64_t* labels, const uint8_t* lut,
                                   float* out, const int64_t* bg_labels, int64_t* labels_out, int64_t B, int64_t CH,
                                   int64_t HW, void* stream) {
    DIGA_REQUIRE(bg && fg && labels && lut && out && B > 0 && CH > 0 && HW > 0, DIGA_EINVAL,
                 "classmix_paste: bad argument");
    DIGA_REQUIRE((labels_out == nullptr) || (bg_labels != nullptr), DIGA_EINVAL,
                 "classmix_paste: labels_out needs bg_labels");
    const bool vec4 = (HW % 4 == 0) && aligned16(bg) && aligned16(fg) && aligned16(out);
    hipStream_t st = (hipStream_t)stream;
    // per pixel: int64 label + CH floats of each image in, CH floats out [+ background label in, label out]
    ProfScope prof(DIGA_PROF_CLASSMIX_PASTE, st, (double)B * HW * (8.0 + CH * 12.0 + (labels_out ? 16.0 : 0.0)));
    if (vec4) {
        dim3 grid((unsigned)ceil_div(HW / 4, 256), (unsigned)B);
        hipLaunchKernelGGL((classmix_paste_kernel<4>), grid, dim3(256), 0, st, bg, fg, (const long long*)labels, lut, out,
                           (const long long*)bg_labels, (long long*)labels_out, (int)CH, HW);
    } else {
        dim3 grid((unsigned)ceil_div(HW, 256), (unsigned)B);
        hipLaunchKernelGGL((classmix_paste_kernel<1>), grid, dim3(256), 0, st, bg, fg, (const long long*)labels, lut, out,
                           (const long long*)bg_labels, (long long*)labels_out, (int)CH, HW);
    }
    return launch_status("diga_classmix_paste");
}
