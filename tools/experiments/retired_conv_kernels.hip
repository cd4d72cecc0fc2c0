// Retired generations of the split-bf16 forward kernel, kept for reference only (NOT compiled into libdiga_hip.so; they
// use ConvArgs and the helpers of diga_amd/csrc/conv.hip as of round 3):
//   conv_fwd_x3_kernel   round 1: fp32 operands split in the loader, two LDS buffers -- superseded by conv_fwd_x3w_kernel
//   conv_fwd_x3t_kernel  round 2: pre-split twins, LDS-DMA, ONE MFMA wave per SIMD -- superseded by conv_fwd_x3t8_kernel (two MFMA
//                        waves per SIMD + dead-tap skipping: 3-17 % faster on every C2 layer shape)
template <int TN, bool EPI = false>
__global__ __launch_bounds__(256, 2) void conv_fwd_x3_kernel(ConvArgs a) {
    constexpr int BM = 128, BN = 64 * TN, TM = 2, BK = 32;
    constexpr int CPR = BK / 4, RPP = 256 / CPR, NPA = BM / RPP, NPB = BN / RPP;
    constexpr int A_PLANE = BM * kRowB, B_PLANE = BN * kRowB, BUF = 2 * A_PLANE + 2 * B_PLANE;
    extern __shared__ __align__(16) unsigned char smem_b[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int lr = t / CPR, c4 = (t % CPR) * 4;
    int pixbase[NPA], iy0[NPA], ix0[NPA];
    bool mok[NPA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + lr + RPP * i;
        mok[i] = m < a.M;
        const int mm = mok[i] ? m : 0;
        const int img = mm / HoWo, rem = mm - img * HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pixbase[i] = img * a.Hi * a.Wi;
        iy0[i] = ho * a.sy + a.oy0;
        ix0[i] = wo * a.sx + a.ox0;
    }
    const int RS = a.R * a.S;
    const int cchunks = a.Cin / BK;
    const int ksteps = RS * cchunks;

    float4 ra[NPA], rb[NPB];
    float fa[NPA], fb[NPB];
    int cob[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int co = n0 + lr + RPP * i;
        const bool ok = co < a.Cout;
        cob[i] = ok ? co : a.Cout - 1;
        fb[i] = ok ? 1.f : 0.f;
    }
    auto gload = [&](int ks) {
        const int tap = ks / cchunks, c0 = (ks - tap * cchunks) * BK;
        const int r = tap / a.S, s = tap - r * a.S;
        const int dy = r * a.ody, dx = s * a.odx;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool ok = mok[i] && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const int cy = min(max(iy, 0), a.Hi - 1), cx = min(max(ix, 0), a.Wi - 1);
            ra[i] = *reinterpret_cast<const float4*>(a.in + (int64_t)(pixbase[i] + cy * a.Wi + cx) * a.in_ld + c0 + c4);
            fa[i] = ok ? 1.f : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i)
            rb[i] = *reinterpret_cast<const float4*>(a.wgt + ((int64_t)cob[i] * RS + tap) * a.Cin + c0 + c4);
    };
    auto lstore = [&](int buf) {
        unsigned char* base = smem_b + buf * BUF;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            uint2 hi, lo;
            split4(ra[i], fa[i], hi, lo);
            const int off = (lr + RPP * i) * kRowB + c4 * 2;
            *reinterpret_cast<uint2*>(base + off) = hi;
            *reinterpret_cast<uint2*>(base + A_PLANE + off) = lo;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            uint2 hi, lo;
            split4(rb[i], fb[i], hi, lo);
            const int off = (lr + RPP * i) * kRowB + c4 * 2;
            *reinterpret_cast<uint2*>(base + 2 * A_PLANE + off) = hi;
            *reinterpret_cast<uint2*>(base + 2 * A_PLANE + B_PLANE + off) = lo;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) gload(ks + 1);
        const unsigned char* Ah = smem_b + cur * BUF;
        const unsigned char* Al = Ah + A_PLANE;
        const unsigned char* Bh = Ah + 2 * A_PLANE;
        const unsigned char* Bl = Bh + B_PLANE;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = (wm * 64 + i * 32 + li) * kRowB + s * 32 + lh * 16;
                ah[i] = *reinterpret_cast<const bf16x8_t*>(Ah + off);
                al[i] = *reinterpret_cast<const bf16x8_t*>(Al + off);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = (wn * 32 * TN + j * 32 + li) * kRowB + s * 32 + lh * 16;
                bh[j] = *reinterpret_cast<const bf16x8_t*>(Bh + off);
                bl[j] = *reinterpret_cast<const bf16x8_t*>(Bl + off);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (ks + 1 < ksteps) lstore(cur ^ 1);
        __syncthreads();
    }

    epilogue_tile<TM, TN, EPI>(acc, reinterpret_cast<float*>(smem_b), a, m0, n0, wm, wn, lane, t, tile_m);
}


template <int TN, bool EPI = false>
__global__ __launch_bounds__(512, 1) void conv_fwd_x3t_kernel(ConvArgs a) {
    constexpr int BM = 256, BN = 64 * TN, NT = 2 * TN, MT = 8;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, STAGE = 2 * A_PLANE + 2 * B_PLANE;
    extern __shared__ __align__(16) unsigned char smem_b[];
    // 8 waves: 0-3 read fragments and issue MFMAs, 4-7 (one per SIMD, next to an MFMA wave) only issue the LDS-DMA loads
    // -- an LDS-DMA instruction costs its wave 60-185 cycles of issue time, which a lone in-order wave cannot overlap
    // with its own MFMAs (measured: one wave per SIMD doing both keeps the matrix pipe 45 % busy)
    const int t = threadIdx.x & 255, lane = t & 63, wv = t >> 6;
    const bool loader = threadIdx.x >= 256;
    const int wm = wv >> 1, wn = wv & 1;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wg % a.tiles_n, tile_m = wg / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const unsigned char* twin = reinterpret_cast<const unsigned char*>(a.in);

    // LDS-DMA geometry of this lane: A rows wv*64 + 16 j + (lane >> 2), j = 0..3, destination slot lane & 3
    const int lrow = lane >> 2;
    const int kslot = (lane & 3) ^ lds_swz(lrow);             // 64 wv + 16 j leave bits 1..3 of the row untouched
    int pixbase[4], yx0[4];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = min(m0 + wv * 64 + 16 * j + lrow, a.M - 1);
        const int img = m / HoWo, rem = m - img * HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pixbase[j] = img * a.Hi * a.Wi;
        yx0[j] = ((ho * a.sy + a.oy0) << 16) | ((wo * a.sx + a.ox0) & 0xffff);
    }
    const int RS = a.R * a.S;
    const int cchunks = a.Cin / 32;
    const int ksteps = RS * cchunks;
    const int64_t rowb = (int64_t)a.in_ld * 4;               // twin bytes per pixel (in_ld = channels per pixel row)
    const unsigned char* pa[4];                              // twin row of (pixel of the current tap) + this lane's k-slot
    int l_tap = 0, l_cc = 0;
    auto set_tap = [&](int tap) {
        const int r = tap / a.S, s = tap - r * a.S;
        const int dy = r * a.ody, dx = s * a.odx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = (yx0[j] >> 16) + dy, ix = (int)(short)(yx0[j] & 0xffff) + dx;
            int cy, cx;
            const bool ok = map_tap(a, iy, ix, cy, cx);
            pa[j] = ok ? twin + (int64_t)(pixbase[j] + cy * a.Wi + cx) * rowb + kslot * 32 : nullptr;
        }
    };
    const unsigned char* bimg = a.wgt_img + (int64_t)tile_n * ksteps * (2 * B_PLANE) + (wv * 2 * TN) * 1024 + lane * 16;
    int l_ks = 0;
    auto issue = [&](int buf) {          // LDS-DMA loads of the K-step the loader state points at, then advance it
        unsigned char* stage = smem_b + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char* src = pa[j] != nullptr ? pa[j] + l_cc * 128 : g_zero16;      // 32 channels = 128 twin bytes
            const unsigned char* src_lo = pa[j] != nullptr ? src + 16 : g_zero16;
            unsigned char* dst = stage + (wv * 64 + 16 * j) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_lo,
                                             (__attribute__((address_space(3))) void*)(dst + A_PLANE), 16, 0, 0);
        }
        const unsigned char* bsrc = bimg + (int64_t)l_ks * (2 * B_PLANE);
        unsigned char* bdst = stage + 2 * A_PLANE + (wv * 2 * TN) * 1024;
#pragma unroll
        for (int c = 0; c < 2 * TN; ++c)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + c * 1024),
                                             (__attribute__((address_space(3))) void*)(bdst + c * 1024), 16, 0, 0);
        ++l_ks;
        if (++l_cc == cchunks) {
            l_cc = 0;
            if (++l_tap < RS) set_tap(l_tap);
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ lds_swz(frow)) << 4);
    const int aoff = wm * 128 * 64 + foff;
    const int boff = 2 * A_PLANE + wn * 32 * TN * 64 + foff;

    // Three-stage ring: the loads of K-step ks + 2 are issued at the top of step ks, so an LDS-DMA load has two steps
    // (~5 k cycles) to land -- one step does not cover an HBM miss.  A wave waits only for its own loads of the NEXT
    // stage (counted vmcnt: the newest stage's loads stay in flight), then the raw barrier makes every wave's visible.
    constexpr int kLoadsPerStage = 8 + 2 * TN;
    auto wait_next = [&](bool newest_in_flight) {
        if (newest_in_flight) {
            if constexpr (TN == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    static_assert(kLoadsPerStage == (TN == 2 ? 12 : 10), "vmcnt literals above");
    if (loader) {
        set_tap(0);
        issue(0);
        if (ksteps > 1) issue(1);
        wait_next(ksteps > 1);
        int nx = 2;
        for (int ks = 0; ks < ksteps; ++ks) {
            const bool ahead = ks + 2 < ksteps;
            if (ahead) issue(nx);                          // that stage was last read in step ks - 1 (barrier since)
            wait_next(ahead);
            nx = nx == 2 ? 0 : nx + 1;
        }
        return;                                            // the epilogue's barriers count the surviving waves
    }
    __builtin_amdgcn_s_barrier();                          // stage 0 has landed
    int cur = 0;                                           // stage of step ks
    for (int ks = 0; ks < ksteps; ++ks) {
        const unsigned char* Ah = smem_b + cur * STAGE + aoff;
        const unsigned char* Al = Ah + A_PLANE;
        const unsigned char* Bh = smem_b + cur * STAGE + boff;
        const unsigned char* Bl = Bh + B_PLANE;
        bf16x8_t bh[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bh[j] = *reinterpret_cast<const bf16x8_t*>(Bh + j * 1024);
            bl[j] = *reinterpret_cast<const bf16x8_t*>(Bl + j * 1024);
        }
        // one wave per SIMD: fragment reads run TWO 16-row tiles ahead of their MFMAs (nobody else hides LDS latency)
        bf16x8_t fa[MT][2];
        fa[0][0] = *reinterpret_cast<const bf16x8_t*>(Ah);
        fa[0][1] = *reinterpret_cast<const bf16x8_t*>(Al);
        fa[1][0] = *reinterpret_cast<const bf16x8_t*>(Ah + 1024);
        fa[1][1] = *reinterpret_cast<const bf16x8_t*>(Al + 1024);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT + 4, 0);
#pragma unroll
        for (int i = 0; i < MT; i += 2) {
            if (i + 2 < MT) {
                fa[i + 2][0] = *reinterpret_cast<const bf16x8_t*>(Ah + (i + 2) * 1024);
                fa[i + 2][1] = *reinterpret_cast<const bf16x8_t*>(Al + (i + 2) * 1024);
                fa[i + 3][0] = *reinterpret_cast<const bf16x8_t*>(Ah + (i + 3) * 1024);
                fa[i + 3][1] = *reinterpret_cast<const bf16x8_t*>(Al + (i + 3) * 1024);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
            // two row tiles at a time: 2 NT independent accumulators between the three MFMAs of one accumulator
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i + u][1], bh[j], acc[i + u][j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i + u][0], bl[j], acc[i + u][j], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i + u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i + u][0], bh[j], acc[i + u][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * NT, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this step's fragment reads are done before the stage is released
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }
    __syncthreads();                                       // (4 surviving waves) all MFMA waves are out of the ring

    float* stage = reinterpret_cast<float*>(smem_b);
    constexpr int LDS_LD = BN + 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (m0 + h * 128 >= a.M) break;              // uniform over the block
        if (h) __syncthreads();
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        stage[(i * 16 + (lane >> 4) * 4 + e) * LDS_LD + wn * 32 * TN + j * 16 + (lane & 15)] = acc[i][j][e];
        }
        __syncthreads();
        drain_stage<2, TN, EPI>(stage, a, m0 + h * 128, n0, t, tile_m * 2 + h);
    }
}

// Three pointwise-layer experiments were measured against the 12-wave kernel below and removed again:
//  * (commit 78ff4b5) a persistent 8-wave kernel whose MFMA waves store from registers while the loader waves prefetch the
//    next tile: 1-6 % faster on K <= 512 into >= 1024 channels without statistics, 2-9 % slower elsewhere; +-0 on the step;
//  * (78ff4b5) a self-loading 4-wave 128 x 128 kernel at two blocks per CU: 6-12 % slower;
//  * (7ede725) a persistent kernel whose two wave groups alternate per tile between computing and storing the previous
//    tile from registers + issuing the LDS-DMA loads (the store fully hidden under the next tile's K loop): -33 % on
//    256 -> 1024 channels with the operands warm in the Infinity Cache (a 154 MB twin re-read by every timed launch), but
//    +-0 (-2 ... +18 % with the BatchNorm statistics) with the operands cold in HBM as they are inside a training step
//    (tools/bench_twin.py --cold), and +2 ms on the step.
// In-kernel stamps from the first one gave the number that mattered: a CU stores ~7.4 B/cycle with plain stores,
// ~12 B/cycle with non-temporal ones (store4_stream).  The lesson of the third: time conv kernels COLD.
// Also tried and removed (commit 9129c47): a stream-K launch of the 12-wave kernel against the 0.6-round tile tail (groups of
// N-tile blocks walking equal ranges of (panel, K-step) in step + a fix-up kernel for the cut tiles): -13 % stand-alone on
// the self-training 3x3 layers, +-0 on the C2 shapes, and -3 ... -6 ms per step LOST inside the two-stream step, whose
// second stream already fills the tails.

