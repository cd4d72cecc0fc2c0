// Probe of ds_read_b64_tr_b16 (the transposing LDS read): prints what lane i of a 16-lane group receives when lane 4q+p
// supplies the address of row q, columns 4p..4p+3 of a block of 16-bit elements with 64-byte rows.
//   hipcc --offload-arch=gfx950 tools/experiments/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    for (int i = threadIdx.x; i < 4096; i += 64) reinterpret_cast<short*>(smem)[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x & 15, g = threadIdx.x >> 4, q = l >> 2, p = l & 3;
    // group g reads rows 4g..4g+3 (row = 32 elements), columns 0..15
    auto* addr = (__attribute__((address_space(3))) s16x4*)(smem + (4 * g + q) * 64 + p * 8);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(addr);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
    short* d;
    hipMalloc(&d, 256 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        if (l % 16 < 3 || l % 16 == 15) printf("lane %2d: %4d %4d %4d %4d   (element = 32*row + col)\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
