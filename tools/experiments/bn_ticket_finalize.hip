// RETIRED in round 6, the same round it was built (was: diga_amd/csrc/norm.hip + a nullable `int32_t* tickets` argument of
// diga_bn_fwd_partials / diga_bn_fwd_records / diga_bn_bwd_partials).  BatchNorm statistics: fold of the producing convolution's partials
// + finaliser in ONE launch -- every fold block takes a ticket for its 64-channel slab (release fence, agent-scope atomicAdd), the block
// that draws the last ticket finalises the slab reading the other blocks' partials with agent-scope atomic loads, and zeroes the counter.
// VERDICT r05 asked for exactly this form ("ticketed last-arriving merge_partials block"; DESIGN section 14.3 had estimated 0.2 ms of the
// step per microsecond taken out of the chain).  Measured on one MI355X, three interleaved rounds of bench.py --lean (C2, fp32):
//     fused 433.9 / 434.2 / 434.6 ms      two launches 409.1 / 410.7 / 408.1 ms       (gpurun_out/r06_ab__.txt, profiles/r06_ab_bn_ticket.txt)
// i.e. +25 ms per step = ~80 us per BatchNorm pass (208 forward + 104 backward): the last-arriving block walks its slab's <= 96 x 3 x 64
// partials ALONE, four 16-channel rounds x two passes of agent-scope loads that do not pipeline (~0.6 us each), where bn_finalize2's 64
// blocks do the same work in parallel with plain pipelined loads in 7.7 us.  AND the results were wrong at the sizes that take the fold
// (> 128 chunks: tests/test_gpu_fullsize_golden.py, traj768) -- stale partials across XCDs or a flaw in the ticket logic, not debugged
// further once the timing was known.  The finaliser chain stays two launches.  Not built.

// Round 6: merge + finalise in ONE launch.  The two-launch chain (merge_partials 6.6 us, bn_finalize2 7.7 us and the gap between them)
// sits on its stream's critical path in front of every BatchNorm apply pass -- 208 times per C2 step, and round 5 measured that chain at
// ~1 : 1 in the step.  Here every merge block takes a ticket for its 64-channel slab after its merged partial is
// out (release fence, agent-scope atomic); the block that draws the LAST ticket of a slab finalises those 64 channels -- the same
// arithmetic as bn_finalize2_kernel, 16 channels x 16 record lanes at a time, four times -- reading the other blocks' partials with
// agent-scope loads (the per-XCD L2s are not coherent with each other for plain loads), and puts the ticket counter back to zero for the
// next call on this stream.  `tickets`: ceil(C / 64) int32, zero on entry, zero on exit (caller-owned, one per stream).
struct FinArgs {
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* save_mean;
    float* save_invstd;
    float* ab;
    float momentum, eps;
};
__global__ __launch_bounds__(256) void merge_finalize_kernel(const float* __restrict__ partial, ColGeom g, int group, float* __restrict__ merged,
                                                             const float* __restrict__ counts, float* __restrict__ mcounts, ColGeom gm,
                                                             FinArgs f, int* __restrict__ tickets) {
    __shared__ double red[2][4][64];
    __shared__ int last;
    merge_partials_block(red, partial, g, group, merged, counts, mcounts, true);
    __threadfence();                                            // this block's merged partial (and mcounts) before its ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(&tickets[blockIdx.y], 1);
        last = (t == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x == 0) tickets[blockIdx.y] = 0;              // (nobody else touches this slab's counter any more in this launch)
    double (*fred)[kFinCh] = reinterpret_cast<double (*)[kFinCh]>(&red[0][0][0]);      // 16 x 16 doubles of the 2 x 4 x 64
#pragma unroll 1
    for (int q = 0; q < 64 / kFinCh; ++q)
        bn_finalize_channels<true>(fred, blockIdx.y * 64 + q * kFinCh, merged, gm, f.gamma, f.beta, f.running_mean, f.running_var, f.save_mean,
                                   f.save_invstd, f.ab, f.momentum, f.eps, mcounts);
}



// colsum_fold + bn_bwd_finalize2 in ONE launch (round 6, the backward twin of merge_finalize_kernel): the fold's grid is (2 C / 64 column
// slabs, parts); a channel slab s (64 channels) owns the two column slabs s and C / 64 + s (sum g, sum g xhat) -- 2 x parts blocks take a
// ticket of tickets[s], the last one finalises the 64 channels.  C % 64 == 0.
__global__ __launch_bounds__(256) void colsum_fold_finalize_kernel(const float* __restrict__ in, int K, int W, int parts, float* __restrict__ out,
                                                                   ColGeom gf, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                                   float* __restrict__ kk, int* __restrict__ tickets) {
    __shared__ double red[2][kFinLn][kFinCh];                   // (the fold uses its first 4 x 64 doubles)
    __shared__ int last;
    colsum_fold_block(reinterpret_cast<double (*)[64]>(&red[0][0][0]), in, K, W, parts, out);
    const int slabs = gf.C / 64, slab = (int)blockIdx.x % slabs;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(&tickets[slab], 1);
        last = (t == 2 * (int)gridDim.y - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x == 0) tickets[slab] = 0;
#pragma unroll 1
    for (int q = 0; q < 64 / kFinCh; ++q)
        bn_bwd_finalize_channels<true>(red, slab * 64 + q * kFinCh, out, gf, gamma, invstd, kk, 1, nullptr, nullptr);
}

