// Retired (NOT compiled): the backward-data epilogue of diga_bwd_epilogue_t inside gemm_f32_persistent_kernel, on the accumulator
// layout (addend / mask / x as dword loads per accumulator element, sum g / sum g * xhat per 64-row chunk in registers).  Parity-
// tested in round 3 and 3.5 ms SLOWER on the step than conv_fwd_dma_kernel<true>'s LDS-staged float4 epilogue:
// every MFMA wave of the CU sits in the epilogue at the same time, and its loads are issued only then.  This is the branch that
// stood at "tile finished" in the kernel's K loop (GemmArgs carried the e_* fields of ConvArgs).
            if constexpr (EPI) {
                // the backward-data epilogue of diga_bwd_epilogue_t, element for element as drain_stage<EPI> applies it, on the
                // accumulator layout: per 32x32 tile 16 rows per lane of one column; the loads of a tile (addend, x, mask) are all
                // issued before their first use; sum g / sum g * xhat per 64-row chunk = this wave's rows
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = col_w + j * 32;
                    float ra = 0.f, rb = 0.f, mu = 0.f, is = 0.f;
                    if (g.e_relu_ab != nullptr) {
                        ra = g.e_relu_ab[col];
                        rb = g.e_relu_ab[g.Cout + col];
                    }
                    if (g.e_partials != nullptr) {
                        mu = g.e_mean[col];
                        is = g.e_invstd[col];
                    }
                    float sd = 0.f, sd2 = 0.f;
#pragma unroll
                    for (int ih = 0; ih < 4; ++ih) {                        // half an accumulator tile at a time (register budget)
                        const int i = ih >> 1, e0 = (ih & 1) * 8;
                        float va[8], vx[8], vy[8];
                        unsigned vb[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + u;
                            const int r = i * 32 + (e & 3) + 8 * (e >> 2);
                            const unsigned row = (unsigned)min(row_w + 4 * lh + r, g.M - 1);      // (32-bit element offsets: host checks)
                            va[u] = g.e_add != nullptr ? g.e_add[row * (unsigned)g.e_add_ld + (unsigned)col] : 0.f;
                            vx[u] = g.e_x != nullptr ? g.e_x[row * (unsigned)g.e_x_ld + (unsigned)col] : 0.f;
                            vy[u] = g.e_masky != nullptr ? g.e_masky[row * (unsigned)g.e_masky_ld + (unsigned)col] : 0.f;
                            vb[u] = g.e_maskbits != nullptr ? g.e_maskbits[row * (unsigned)g.e_maskbits_ld + (unsigned)(col >> 3)] : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + u;
                            const int r = i * 32 + (e & 3) + 8 * (e >> 2);
                            float v = acc[i][j][e] + va[u];
                            if (g.e_masky != nullptr) v = vy[u] > 0.f ? v : 0.f;
                            else if (g.e_maskbits != nullptr) v = ((vb[u] >> (col & 7)) & 1u) ? v : 0.f;
                            else if (g.e_relu_ab != nullptr) v = __builtin_fmaf(vx[u], ra, rb) > 0.f ? v : 0.f;
                            if (r < rows_left) {
                                __builtin_nontemporal_store(v, o + (int64_t)r * g.out_ld + j * 32);
                                sd += v;
                                sd2 += v * ((vx[u] - mu) * is);
                            }
                            acc[i][j][e] = 0.f;
                        }
                        __builtin_amdgcn_sched_barrier(0);                  // (keep the next half's 32 loads from being hoisted up here)
                    }
                    if (g.e_partials != nullptr && row_w < g.M) {
                        sd += __shfl_xor(sd, 32, 64);
                        sd2 += __shfl_xor(sd2, 32, 64);
                        if (lh == 0) {
                            float* sp = g.e_partials + (int64_t)(row_w >> 6) * 2 * g.Cout + col;
                            sp[0] = sd;
                            sp[g.Cout] = sd2;
                        }
                    }
                }
                ks_in_tile = 0;
                ++it;
                continue;
            }
