// RETIRED in round 6 (was: diga_amd/csrc/conv.hip, gemm_f32_persistent_kernel<JUNC = true> + diga_conv2d_junction_f32).
// conv1 of a bottleneck fused with the residual junction in front of it: eight loader waves apply relu(fma(y3, a, b) + skip) while
// staging the GEMM's A operand, the duty column tile stores the activated tensor + its ReLU mask bits.  Bit-identical to the
// stand-alone apply pass (round 4, tests/test_gpu_bn_box.py::test_fused_residual_junction...), and MEASURED SLOWER: the pair costs
// 1.056 vs 1.005 ms on layer3, 3.66 vs 3.25 ms on layer4 (every column-tile block of a row tile loads and transforms y3 and skip
// again and the GEMM stalls on it), the C2 step 437.6 vs 432.6 ms (round 4).  Default-off since then; removed from
// the product library in round 6.  Not built; kept as the record of what was tried.

// ---- GemmArgs fields
    // JUNC instantiation: A is not read as it is but produced on the way into LDS -- x = relu(fma(A, a[k], b[k]) + skip), the residual
    // junction of a bottleneck (A = conv3's raw output, a / b = its BatchNorm's coefficients, skip = the block input).  The column-tile-0
    // block of every row tile also stores x (the activated tensor the rest of the network reads) and its ReLU mask bits.
    const float* j_skip;
    const float* j_ab;                               // [2][K]
    float* j_x;                                      // [M][j_x_ld]
    unsigned char* j_bits;                           // nullable [M][K / 8]
    int64_t j_skip_ld, j_x_ld;

// ---- kernel body (inside gemm_f32_persistent_kernel<true>, 1024 threads, __launch_bounds__(1024, 4))
    if constexpr (JUNC) {
        // the BatchNorm coefficients [2][K] behind the ring (16 KB are free next to its 144 KB), visible to the loader waves after one
        // block-wide barrier
        float* abs = reinterpret_cast<float*>(smem_b + 3 * STAGE);
        for (int i = threadIdx.x; i < 2 * g.K; i += 1024) abs[i] = g.j_ab[i];
        __syncthreads();
    }
    if (loader && JUNC) {
        // EIGHT loader waves here (block of 1024 threads, 4 waves per SIMD, 128 registers each -- the MFMA branch needs 126): every wave
        // stages 32 rows of A and 16 of W, so that two K-steps of operands in flight are 64 registers instead of 128.
        // A goes global -> registers -> LDS: y3 and skip as float4 per (row, 16-byte chunk), two K-steps in flight (register sets 0 / 1),
        // x = relu(fma(y3, a, b) + skip) -- affine_apply_kernel's expression, bit for bit -- written to the stage in the layout the LDS-DMA
        // path produces (position lane & 7 of row r holds chunk (lane & 7) ^ ((r >> 1) & 7)); the weights keep their LDS-DMA loads.
        const int lw = wv - 8, lrow = lane >> 3, l7 = lane & 7;
        const float* abs = reinterpret_cast<const float*>(smem_b + 3 * STAGE);
        const unsigned char* pb[2];
        float4 Y0[4], S0[4], Y1[4], S1[4];                       // (two named sets: a runtime set index would put them in scratch)
        int r0_0 = 0, r0_1 = 0, cc_0 = 0, cc_1 = 0;
        bool first_0 = false, first_1 = false;
        int l_it = 0, l_cc = 0, l_row0 = 0;
        bool l_first = false;
        // per-lane constants: float offset of this lane's 16-byte chunk inside a K-step's 32 floats, for each of its four rows
        // (the XOR swizzle depends on the row's position in the tile, not on the tile)
        int c4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c4[j] = (l7 ^ (((8 * j + lrow) >> 1) & 7)) << 2;
        auto set_tile = [&](int it) {
            const int t = start + slot + nslots * it;
            const int tile_n = t % g.tiles_n, tile_m = t / g.tiles_n;
            l_row0 = tile_m * 256 + lw * 32 + lrow;
            // which of a row tile's column-tile blocks stores x: alternating with the block's tile count (a block's tile_n is fixed by its
            // slot parity: always the same half of the blocks would carry all the stores)
            l_first = tile_n == ((tile_m >> 4) % g.tiles_n);
            const unsigned char* wb = reinterpret_cast<const unsigned char*>(g.W) + (int64_t)(tile_n * 128 + lw * 16 + lrow) * wrowb;
#pragma unroll
            for (int c = 0; c < 2; ++c) pb[c] = wb + (int64_t)(8 * c) * wrowb + ((l7 ^ (((8 * c + lrow) >> 1) & 7)) << 4);
        };
        auto issue = [&](float4 (&Y)[4], float4 (&S)[4], int& s_row0, int& s_cc, bool& s_first, int buf) {   // loads of one K-step: 8 to
            unsigned char* stage = smem_b + buf * STAGE;                                                      // registers, 2 weight loads to LDS
            s_row0 = l_row0; s_cc = l_cc; s_first = l_first;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // 32-bit element offsets from the uniform base pointers (M * ld < 2^31 elements: checked by the entry point); rows beyond M:
                // clamped loads, nothing stored
                const unsigned row = (unsigned)min(l_row0 + 8 * j, g.M - 1);
                const unsigned col = (unsigned)(l_cc * 32 + c4[j]);
                Y[j] = *reinterpret_cast<const float4*>(g.A + (row * (unsigned)g.a_ld + col));
                S[j] = *reinterpret_cast<const float4*>(g.j_skip + (row * (unsigned)g.j_skip_ld + col));
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[c] + l_cc * 128),
                                                 (__attribute__((address_space(3))) void*)(stage + A_BYTES + (lw * 16 + 8 * c) * 128), 16, 0, 0);
            if (++l_cc == ksteps) {
                l_cc = 0;
                if (++l_it < nmine) set_tile(l_it);
            }
        };
        auto consume = [&](const float4 (&Y)[4], const float4 (&S)[4], int row0, int cc, bool first, int buf) {   // a register set ->
            unsigned char* stage = smem_b + buf * STAGE;                    // stage `buf` (+ x and its mask bits from the duty block)
            const bool bits_on = first && g.j_bits != nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = cc * 32 + c4[j];
                const float4 av = *reinterpret_cast<const float4*>(abs + k), bv = *reinterpret_cast<const float4*>(abs + g.K + k);
                const float4 y = Y[j], sk = S[j];
                float4 o;
                o.x = __builtin_fmaf(y.x, av.x, bv.x) + sk.x; o.y = __builtin_fmaf(y.y, av.y, bv.y) + sk.y;
                o.z = __builtin_fmaf(y.z, av.z, bv.z) + sk.z; o.w = __builtin_fmaf(y.w, av.w, bv.w) + sk.w;
                unsigned v = ((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u)) << c4[j];
                const int row = row0 + 8 * j;
                const bool live = row < g.M;
                o.x = live ? fmaxf(o.x, 0.f) : 0.f; o.y = live ? fmaxf(o.y, 0.f) : 0.f;
                o.z = live ? fmaxf(o.z, 0.f) : 0.f; o.w = live ? fmaxf(o.w, 0.f) : 0.f;
                *reinterpret_cast<float4*>(stage + (lw * 32 + 8 * j) * 128 + lane * 16) = o;
                if (first && live) store4_stream(g.j_x + ((unsigned)row * (unsigned)g.j_x_ld + (unsigned)k), o.x, o.y, o.z, o.w);
                if (bits_on) {
                    // the 8 lanes of a row OR their nibbles into the 32-bit mask word of this K-step (one bit per channel)
                    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
                    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
                    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);     // row_half_mirror
                    if (l7 == 0 && live) *reinterpret_cast<unsigned*>(g.j_bits + ((unsigned)row * (unsigned)(g.K >> 3) + (unsigned)(cc * 4))) = v;
                }
            }
        };
        set_tile(0);
        issue(Y0, S0, r0_0, cc_0, first_0, 0);
        if (total_steps > 1) {
            issue(Y1, S1, r0_1, cc_1, first_1, 1);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // the 10 newest (step 1) may be in flight; step 0 has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        consume(Y0, S0, r0_0, cc_0, first_0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // stage 0 complete
        int nx = 2;
        // iteration gs: issue step gs + 2 into the register set step gs used, then turn step gs + 1's registers into its stage; the
        // loop is unrolled by two so that the sets are named at compile time
        auto half = [&](int gs, float4 (&Yi)[4], float4 (&Si)[4], int& ri, int& ci, bool& fi, const float4 (&Yc)[4], const float4 (&Sc)[4],
                        int rc, int cc, bool fc) {
            const bool ahead = gs + 2 < total_steps;
            const int nb = nx == 0 ? 2 : nx - 1;                 // stage of step gs + 1
            if (ahead) issue(Yi, Si, ri, ci, fi, nx);
            if (gs + 1 < total_steps) {
                if (ahead) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                consume(Yc, Sc, rc, cc, fc, nb);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            nx = nx == 2 ? 0 : nx + 1;
        };
        for (int gs = 0; gs < total_steps; gs += 2) {
            half(gs, Y0, S0, r0_0, cc_0, first_0, Y1, S1, r0_1, cc_1, first_1);                     // even step: set 0 reloaded, set 1 consumed
            if (gs + 1 < total_steps) half(gs + 1, Y1, S1, r0_1, cc_1, first_1, Y0, S0, r0_0, cc_0, first_0);
        }
        return;
    }

// ---- entry points
// conv1 of a bottleneck fused with the residual junction in front of it (G5/model/seg_model_noaux.py:96-101 of block b, :85-87 of
// block b + 1):  x = relu(fma(y3, a, b) + skip)  [stored, with its ReLU mask bits]  and  out = x . W^T  [+ BatchNorm statistics of out]
// in ONE launch of the persistent GEMM -- the loader waves apply the junction on the way into LDS (gemm_f32_persistent_kernel<true>).
extern "C" int diga_conv2d_junction_ok(int64_t M, int64_t K, int64_t Cout) {
    return M > 0 && K % 32 == 0 && K >= 32 && K <= 2048 && Cout % 128 == 0 && ceil_div(M, 256) * (Cout / 128) >= 512 && M < (1ll << 31);
}

extern "C" int diga_conv2d_junction_f32(const float* y3, int64_t y3_ld, const float* skip, int64_t skip_ld, const float* ab, float* x_out,
                                        int64_t x_ld, unsigned char* relu_bits, const float* wgt, float* out, int64_t out_ld,
                                        float* stats_partial, int64_t M, int64_t K, int64_t Cout, void* stream) {
    DIGA_REQUIRE(y3 && skip && ab && x_out && wgt && out, DIGA_EINVAL, "conv2d_junction: null pointer");
    DIGA_REQUIRE(M * std::max({y3_ld, skip_ld, x_ld}) < (1ll << 31), DIGA_EINVAL, "conv2d_junction: tensors beyond 2^31 elements");
    DIGA_REQUIRE(diga_conv2d_junction_ok(M, K, Cout), DIGA_EINVAL,
                 "conv2d_junction: needs K %% 32 == 0, K <= 2048, Cout %% 128 == 0 and >= 512 tiles (diga_conv2d_junction_ok)");
    DIGA_REQUIRE(y3_ld >= K && skip_ld >= K && x_ld >= K && out_ld >= Cout && y3_ld % 4 == 0 && skip_ld % 4 == 0 && x_ld % 4 == 0 &&
                     out_ld % 4 == 0, DIGA_EINVAL, "conv2d_junction: bad leading dimension");
    DIGA_REQUIRE(aligned16(y3) && aligned16(skip) && aligned16(ab) && aligned16(x_out) && aligned16(wgt) && aligned16(out), DIGA_EALIGN,
                 "conv2d_junction: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(DIGA_PROF_CONV_FWD, st, 2.0 * (double)M * (double)Cout * (double)K);
    GemmArgs g;
    g.A = y3; g.W = wgt; g.out = out; g.M = (int)M; g.K = (int)K; g.Cout = (int)Cout;
    g.tiles_m = (int)ceil_div(M, 256); g.tiles_n = (int)(Cout / 128); g.wb_tiles = 0; g.wb_stride = 0;
    g.a_ld = y3_ld; g.out_ld = out_ld; g.bias = nullptr; g.stats = stats_partial;
    g.j_skip = skip; g.j_ab = ab; g.j_x = x_out; g.j_bits = relu_bits; g.j_skip_ld = skip_ld; g.j_x_ld = x_ld;
    const size_t sh = 3 * (256 + 128) * 128 + (size_t)2 * K * sizeof(float);
    (void)hipFuncSetAttribute((const void*)gemm_f32_persistent_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(gemm_f32_persistent_kernel<true>, dim3(256), dim3(1024), sh, st, g);
    return launch_status("diga_conv2d_junction_f32");
}

