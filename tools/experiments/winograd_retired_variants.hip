// RETIRED in round 6 (was: diga_amd/csrc/winograd.hip).  Compile-time A/B variants of the Winograd transforms that were measured
// and lost, removed from the product library; the numbers that retired them:
//   -DDIGA_WINO_EPI_OLD   round 4's backward-data output transform (below): 256 VGPRs, one wave per SIMD, 316 us per launch against
//                         130 us for winoM_output_epi2_kernel (124 VGPRs) -- -4.7 ms per C2 step
//   -DDIGA_WINO_XCD       XCD-contiguous block order of the input transform: FETCH_SIZE -15 %, kernel 141 vs 132 us, step +1.5 ms
//   -DDIGA_WINO_BAND=4    banded tile tables: input transform 177 vs 141 us per launch in the step
//   -DDIGA_WINO6_IN_VEC=float4   6x6 input transform on float4: 108 vs 163 us warm, 195 vs 177 us in the step, step +3 ms
//   -DDIGA_WINO_IN_OCC / _OUT_OCC / _IN_PLAIN_STORE / _EPI_TL     register caps, plain stores, tile lanes: no gain
//   in_ab (DIGA_FUSE_BN1=1: bn1's apply pass folded into conv2's Winograd input transform; entry points
//   diga_conv2d_winograd_f32_ab / diga_conv2d_wgrad_winograd_f32_ab): bit-identical, step 425.6 vs 420.6 ms (+5 ms)
// Not built.

// one output pixel (VW = 4, 2 or 1 channels) of a backward-data convolution through the epilogue of diga_bwd_epilogue_t (the arithmetic
// of wino_output_epi_kernel / drain_stage<EPI>, element for element), its operands already in registers
template <typename V>
__device__ __forceinline__ void wino_epi_pixel_regs(V o, V add, V xin, V ym, unsigned bits, int64_t row, int k, float* __restrict__ y,
                                                    int64_t ld, const WinoEpi& ep, const float* ra, const float* rb, const float* mu,
                                                    const float* is, float* sd, float* sd2) {
    constexpr int VW = sizeof(V) / 4;
    float v[VW], xv[VW], a4[VW], y4[VW];
    *reinterpret_cast<V*>(v) = o;
    *reinterpret_cast<V*>(a4) = add;
    *reinterpret_cast<V*>(xv) = xin;
    *reinterpret_cast<V*>(y4) = ym;
    if (ep.add != nullptr) {
#pragma unroll
        for (int c = 0; c < VW; ++c) v[c] += a4[c];
    }
    if (ep.masky != nullptr) {
#pragma unroll
        for (int c = 0; c < VW; ++c) v[c] = y4[c] > 0.f ? v[c] : 0.f;
    } else if (ep.maskbits != nullptr) {
#pragma unroll
        for (int c = 0; c < VW; ++c) v[c] = ((bits >> c) & 1u) ? v[c] : 0.f;
    } else if (ep.relu_ab != nullptr) {
#pragma unroll
        for (int c = 0; c < VW; ++c) v[c] = __builtin_fmaf(xv[c], ra[c], rb[c]) > 0.f ? v[c] : 0.f;
    }
    *reinterpret_cast<V*>(y + row * ld + k) = *reinterpret_cast<const V*>(v);
#pragma unroll
    for (int c = 0; c < VW; ++c) {
        sd[c] += v[c];
        sd2[c] += v[c] * ((xv[c] - mu[c]) * is[c]);
    }
}

// winoM_output_kernel with the backward-data epilogue: block (g, s) = tiles [g * tpb, (g + 1) * tpb) x channels [64 VW s, 64 VW (s + 1)),
// 256 / TL channel groups x TL tile lanes; partial-row layout of wino_output_epi_kernel (the finaliser only adds the rows up).
template <int M, typename V, int TL>
__global__ __launch_bounds__(256) void winoM_output_epi_kernel(const float* __restrict__ Mb, const int4* __restrict__ tab,
                                                               float* __restrict__ y, int64_t ld, int64_t T, int64_t Tp, int K,
                                                               int H, int W, int d, int tpb, WinoEpi ep) {
    constexpr int A = M + 2;
    constexpr int VW = sizeof(V) / 4;
    constexpr int CG = 256 / TL;                 // channel groups per block; TL tile lanes
    __shared__ float red[2][TL][CG * VW];
    const int q = threadIdx.x % CG, tl = threadIdx.x / CG;
    const int k = (blockIdx.y * CG + q) * VW;
    const bool kok = k < K;
    const int64_t t0 = (int64_t)blockIdx.x * tpb;
    int64_t t1 = t0 + tpb;
    if (t1 > T) t1 = T;
    const int64_t plane = Tp * K;
    float ra[VW], rb[VW], mu[VW], is[VW], sd[VW], sd2[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) ra[c] = rb[c] = mu[c] = is[c] = sd[c] = sd2[c] = 0.f;
    if (kok && ep.relu_ab != nullptr) {
        *reinterpret_cast<V*>(ra) = *reinterpret_cast<const V*>(ep.relu_ab + k);
        *reinterpret_cast<V*>(rb) = *reinterpret_cast<const V*>(ep.relu_ab + K + k);
    }
    if (kok && ep.partials != nullptr) {
        *reinterpret_cast<V*>(mu) = *reinterpret_cast<const V*>(ep.mean + k);
        *reinterpret_cast<V*>(is) = *reinterpret_cast<const V*>(ep.invstd + k);
    }
    if (kok) {
        for (int64_t t = t0 + tl; t < t1; t += TL) {
            const int4 e = tab[t];
            // ONE round trip per tile: the A * A product loads and, right behind them, the epilogue operands (addend, x, mask) of the
            // tile's M * M pixels are all issued before anything is used (the pixel loads used to start only after the transform:
            // two dependent latencies per tile at one or two waves per SIMD -- 2.3 TB/s)
            V mt[A][A];
#pragma unroll
            for (int j = 0; j < A; ++j)
#pragma unroll
                for (int i = 0; i < A; ++i) mt[i][j] = nt_loadv<V>(Mb + t * K + k + (A * i + j) * plane);
            V va[M][M], vx[M][M], vy[M][M];
            unsigned vb[M][M];
            bool ok[M][M];
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const int yy = e.y + i * d, xx = e.z + j * d;
                    ok[i][j] = yy < H && xx < W;
                    const int64_t row = (int64_t)(e.x * H + min(yy, H - 1)) * W + min(xx, W - 1);     // (clamped: loads are unconditional)
                    va[i][j] = ep.add != nullptr ? *reinterpret_cast<const V*>(ep.add + row * ep.add_ld + k) : vzero<V>();
                    vx[i][j] = ep.x != nullptr ? *reinterpret_cast<const V*>(ep.x + row * ep.x_ld + k) : vzero<V>();
                    vy[i][j] = ep.masky != nullptr ? *reinterpret_cast<const V*>(ep.masky + row * ep.masky_ld + k) : vzero<V>();
                    vb[i][j] = ep.maskbits != nullptr ? (unsigned)ep.maskbits[row * ep.maskbits_ld + (k >> 3)] >> (k & 7) : 0u;
                }
            V s[M][A];
#pragma unroll
            for (int j = 0; j < A; ++j) {
                V col_in[A], col[M];
#pragma unroll
                for (int i = 0; i < A; ++i) col_in[i] = mt[i][j];
                Xf<M>::at(col_in, col);
#pragma unroll
                for (int i = 0; i < M; ++i) s[i][j] = col[i];
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
                V o[M];
                Xf<M>::at(s[i], o);
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    if (!ok[i][j]) continue;
                    const int64_t row = (int64_t)(e.x * H + e.y + i * d) * W + e.z + j * d;
                    wino_epi_pixel_regs<V>(o[j], va[i][j], vx[i][j], vy[i][j], vb[i][j], row, k, y, ld, ep, ra, rb, mu, is, sd, sd2);
                }
            }
        }
    }
    if (ep.partials == nullptr) return;
#pragma unroll
    for (int c = 0; c < VW; ++c) {
        red[0][tl][q * VW + c] = sd[c];
        red[1][tl][q * VW + c] = sd2[c];
    }
    __syncthreads();
    const int ch = blockIdx.y * CG * VW + threadIdx.x;
    if (threadIdx.x < CG * VW && ch < K) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int l = 0; l < TL; ++l) {
            a0 += red[0][l][threadIdx.x];
            a1 += red[1][l][threadIdx.x];
        }
        float* sp = ep.partials + (int64_t)blockIdx.x * 2 * K + ch;
        sp[0] = a0;
        sp[K] = a1;
    }
}


template <int M>
static void launch_output_epi_m(const float* Mb, const int4* tab, float* out, int64_t out_ld, const WinoGeom& g, int64_t Cout, int64_t G,
                                int tpb, const WinoEpi& ep, hipStream_t st) {
    using VT = typename Vec<M>::Out;
    constexpr int VW = sizeof(VT) / 4;
    constexpr int TL = kEpiTileLanes;
    hipLaunchKernelGGL((winoM_output_epi_kernel<M, VT, TL>), dim3((unsigned)G, (unsigned)ceil_div(Cout, (256 / TL) * VW)), dim3(256), 0, st, Mb,
                       tab, out, out_ld, g.T, g.Tp, (int)Cout, g.H, g.W, g.d, tpb, ep);
}
