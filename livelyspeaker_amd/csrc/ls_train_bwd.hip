// Fused backward of the mixer (8 MLPblocks, mlp_module.py:37-91) for the training step, the mirror image of the fused
// training forward (k_step TRAIN variant, ls_step.hip): one 512-thread workgroup holds TWO samples of the batch (70 | 72
// rows), the gradient of the residual stream G stays in registers in the MFMA C/D layout for all layers (wave w owns
// channels [64w, 64w+64)), and per layer, last to first:
//   1. dA2 = G * SiLU'(A2)                         -> global (operand of the batch-level weight-gradient GEMM) and LDS
//   2. dU2 = dA2 . Wch          (v_mfma_f32_16x16x4_f32 against the TRANSPOSED weight image; ragged rows on FMAs / 4x4x1 MFMAs)
//   3. LayerNorm-2 backward: G += rstd * (gy - mean(gy) - xhat * mean(gy * xhat)), gy = dU2 * alpha2; row means via one
//      cross-wave LDS exchange; d alpha2 / d beta2 / d bias as per-workgroup partial column sums
//   4. dA1 = G * SiLU'(A1)                         -> global (operand of the token-weight gradient) and LDS (wave-private columns)
//   5. dU1 = blockdiag(Wt, Wt)^T . dA1   (token mixing on MFMA with the transposed block-diagonal image)
//   6. LayerNorm-1 backward as in 3; per-sample sum over tokens of G -> d(timestep embedding) partial of this layer
// The row means are only known after the whole row has been produced, so the update is applied in two parts (rstd * gy at once,
// the mean terms afterwards from the saved x-hat) and gy never leaves the registers.  Everything the kernel sums is reduced in a fixed order (registers,
// cross-lane butterflies, LDS in wave order): the step stays bit-reproducible.
#include "ls_internal.h"
#include "ls_step_common.h"
#include "ls_train.h"
#include "ls_lanes.h"

namespace ls {

template <int S>
__global__ __launch_bounds__(512) void k_mixer_bwd(const MixerBwdArgs a) {
    constexpr int R = 2 * S;
    constexpr int MK = (R + 3) / 4;
    constexpr int kFullTiles = 4;
    constexpr int NREM = R - 16 * kFullTiles;
    constexpr bool kRemMfma = (NREM % 4 == 0);
    constexpr int NRG = kRemMfma ? NREM / 4 : 1;
    constexpr int NRV = kRemMfma ? 1 : NREM;
    static_assert(NREM > 0 && NREM <= 16 && R <= 16 * kNT, "row tiling");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* psum = smem;                      // [8 waves][80 rows] (s1, s2) pairs of the LayerNorm-backward row sums
    float* U = smem + 2 * kWaves * 16 * kNT; // [R][520] fp32 MFMA operand (dA2, then dA1)
    float* REM = U + R * kUStride;           // [8 waves][2][NREM][16] remainder-row patch

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int s16 = lane & 15, g = lane >> 4, chw = 64 * w + 4 * g;
    auto fresh = [&]() {                     // keeps per-lane addresses phase-local (see ls_step.hip)
        asm volatile("" : "+v"(lane));
        s16 = lane & 15;
        g = lane >> 4;
        chw = 64 * w + 4 * g;
    };
    auto row_of = [&](int t) { return 16 * t + s16; };
    auto grow_of = [&](int t) -> int {       // global row (sample * S + token) or -1 for padding / past the batch
        const int r = 16 * t + s16;
        if (r >= R) return -1;
        const int sq = r >= S ? 1 : 0;
        const int sample = 2 * b + sq;
        return sample < a.B ? sample * S + (r - sq * S) : -1;
    };
    auto growc_of = [&](int t) -> int { const int gr = grow_of(t); return gr >= 0 ? gr : 0; };   // clamped: loads stay branch-free
    const size_t LR = (size_t)a.B * S;       // rows per layer of the saved activations

    f4 G[kCB][kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        const int gr = grow_of(t);
#pragma unroll
        for (int cb = 0; cb < kCB; ++cb)
            G[cb][t] = gr >= 0 ? *reinterpret_cast<const f4*>(a.g + (size_t)gr * kD + chw + 16 * cb) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    f2* pst = reinterpret_cast<f2*>(psum);

    // LayerNorm backward, G += rstd * (gy - m1 - xhat * m2) with gy = dU * alpha and the row means m1 = mean(gy), m2 = mean(gy * xhat),
    // in two parts so that gy never leaves the registers: ln_bwd_tile adds rstd * gy as soon as the MFMAs deliver dU; once the
    // cross-wave row sums are known, ln_bwd_finish subtracts rstd * (m1 + xhat * m2), which needs only the saved x-hat again.
    // (A gy slab per workgroup in global memory, written by the tiles and re-read here, was 1.2 GB of traffic per step.)
    auto ln_bwd_finish = [&](float (&s1)[kNT], float (&s2)[kNT], const float* xsaved, const float* stats) {
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
            float u = s1[t], v = s2[t];
            u = xor32_sum(xor16_sum(u));
            v = xor32_sum(xor16_sum(v));
            if (g == 0) pst[w * 80 + 16 * t + s16] = (f2){u, v};
        }
        __syncthreads();
        float m1[kNT], m2[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) {
                const f2 pp = pst[ww * 80 + 16 * t + s16];
                u += pp.x;
                v += pp.y;
            }
            m1[t] = u * (1.0f / kD);
            m2[t] = v * (1.0f / kD);
        }
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
            // all loads of the tile first, from clamped rows (a guarded load would serialise behind its own s_waitcnt)
            const int gr = grow_of(t), grc = growc_of(t);
            const f2 st = *reinterpret_cast<const f2*>(stats + (size_t)grc * 2);
            f4 x[kCB];
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb) x[cb] = *reinterpret_cast<const f4*>(xsaved + (size_t)grc * kD + chw + 16 * cb);
            if (gr >= 0) {
                const float c1 = st.y * m1[t], c2 = st.y * m2[t];
#pragma unroll
                for (int cb = 0; cb < kCB; ++cb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) G[cb][t][j] -= fmaf(x[cb][j], c2, c1);      // x: saved x-hat
            }
        }
        __syncthreads();                      // psum may be rewritten
    };
    // consume one (tile, channel block) of dU: G += rstd * gy, row partial sums, LayerNorm-parameter partial column sums
    auto ln_bwd_tile = [&](const f4 du, int t, int cb, const f4 x, const f2 st, const f4 al, float& s1, float& s2, f4& pa, f4& pb) {
        if (grow_of(t) < 0) return;
        f4 gy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = x[j];                // the forward saved x-hat itself
            gy[j] = du[j] * al[j];
            s1 += gy[j];
            s2 = fmaf(gy[j], xh, s2);
            pa[j] = fmaf(du[j], xh, pa[j]);
            pb[j] += du[j];
            G[cb][t][j] = fmaf(st.y, gy[j], G[cb][t][j]);
        }
    };
    // column partials of this workgroup: sum over the 16 row lanes, lanes s16 == 0 write [wg][layer][which][512]
    auto write_colpart = [&](f4 v, int l, int which, int cb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = row16_sum(v[j]);
        if (s16 == 0) *reinterpret_cast<f4*>(a.colpart + (((size_t)b * a.layers + l) * 5 + which) * kD + chw + 16 * cb) = v;
    };

    for (int l = a.layers - 1; l >= 0; --l) {
        fresh();
        // ---- 1. dA2 = G * SiLU'(A2): weight-gradient operand + MFMA operand; its column sums are d bias -------------
        {
            const float* A2 = a.a2 + (size_t)l * LR * kD;
            float* dA2 = a.da2 + (size_t)l * LR * kD;
            // the whole A2 tile of this lane first (20 float4 in flight together; no MFMA accumulators are live here): one memory
            // round trip per phase instead of one per channel block
            f4 avs[kCB][kNT];
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb)
#pragma unroll
                for (int t = 0; t < kNT; ++t) avs[cb][t] = *reinterpret_cast<const f4*>(A2 + (size_t)growc_of(t) * kD + chw + 16 * cb);
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb) {
                f4 pbias = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < kNT; ++t) {
                    if (16 * t + s16 >= R) continue;
                    const int gr = grow_of(t);
                    f4 d = (f4){0.f, 0.f, 0.f, 0.f};
                    const f4 av = avs[cb][t];
                    if (gr >= 0) {
                        d = silu_grad4(av, G[cb][t]);
                        *reinterpret_cast<f4*>(dA2 + (size_t)gr * kD + chw + 16 * cb) = d;
                        pbias += d;
                    }
                    *reinterpret_cast<f4*>(&U[row_of(t) * kUStride + chw + 16 * cb]) = d;
                }
                write_colpart(pbias, l, 0, cb);
            }
        }
        __syncthreads();
        // ---- 2./3. dU2 = dA2 . Wch (transposed image) and the LayerNorm-2 backward -------------------------------
        float s1[kNT], s2[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
        const float* X2 = a.x2 + (size_t)l * LR * kD;
        const float* S2 = a.s2 + (size_t)l * LR * 2;
        f2 st2[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t) st2[t] = *reinterpret_cast<const f2*>(S2 + (size_t)growc_of(t) * 2);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            fresh();
            f4 acc[2][kFullTiles];
            float racc[2][NRV];
            f4 racc4[2][NRG];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
                for (int t = 0; t < kFullTiles; ++t) acc[c2][t] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < NRV; ++r) racc[c2][r] = 0.f;
#pragma unroll
                for (int r = 0; r < NRG; ++r) racc4[c2][r] = (f4){0.f, 0.f, 0.f, 0.f};
            }
            const wrsrc_t wrs = wrsrc(a.wchT_img);                        // buffer-descriptor loads, as in k_step (ls_lanes.h: uniform_rsrc)
            const int wsb = (((l * kWaves + w) * 2 + p) * 32) * 2 * 1024;
            // operand addresses off pinned LDS bases + immediate offsets, as in k_step's channel-mixing loop
            typedef const __attribute__((address_space(3))) float* ldsp;
            typedef const __attribute__((address_space(3))) f4* ldsp4;
            ldsp ub0 = (ldsp)(U + s16 * kUStride + 4 * g);
            ldsp ub2 = ub0 + 32 * kUStride;
            ldsp ur = (ldsp)(U + (16 * kFullTiles + (kRemMfma ? (lane & 3) : 0)) * kUStride + 4 * g);
            asm volatile("" : "+v"(ub0), "+v"(ub2), "+v"(ur));
            f4 An[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) An[c2] = wload4(wrs, lane * 16, wsb + c2 * 1024);
#pragma unroll 2
            for (int q = 0; q < 32; ++q) {
                f4 A[2], Bv[kFullTiles], Ur[kRemMfma ? NRG : NRV];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) A[c2] = An[c2];
                {
                    const int qn = (q + 1 < 32) ? q + 1 : 31;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) An[c2] = wload4(wrs, lane * 16, wsb + (qn * 2 + c2) * 1024);
                }
#pragma unroll
                for (int t = 0; t < kFullTiles; ++t) Bv[t] = *(ldsp4)((t < 2 ? ub0 : ub2) + 16 * (t & 1) * kUStride + 16 * q);
#pragma unroll
                for (int r = 0; r < (kRemMfma ? NRG : NRV); ++r)
                    Ur[r] = *(ldsp4)(ur + (kRemMfma ? 4 : 1) * r * kUStride + 16 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int t = 0; t < kFullTiles; ++t) acc[c2][t] = MFMA(A[c2][j], Bv[t][j], acc[c2][t]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        if constexpr (kRemMfma) {
#pragma unroll
                            for (int r = 0; r < NRG; ++r)
                                racc4[c2][r] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[c2][j], Ur[r][j], racc4[c2][r], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int r = 0; r < NRV; ++r) racc[c2][r] = fmaf(A[c2][j], Ur[r][j], racc[c2][r]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            fresh();
            // LayerNorm-2 backward inputs of both channel blocks of this pass: in flight during the remainder-row reduction below
            f4 xs2[2][kNT], al2[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                al2[c2] = *g4(a.ln2a + l * kD + chw + 16 * (2 * p + c2));
#pragma unroll
                for (int t = 0; t < kNT; ++t) xs2[c2][t] = *reinterpret_cast<const f4*>(X2 + (size_t)growc_of(t) * kD + chw + 16 * (2 * p + c2));
            }
            float* rem = REM + w * (2 * NREM * 16);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                if constexpr (kRemMfma) {
#pragma unroll
                    for (int r = 0; r < NRG; ++r) {
                        f4 v = racc4[c2][r];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = xor32_sum(xor16_sum(v[i]));
                        }
                        if (g == 0) *reinterpret_cast<f4*>(&rem[(c2 * NREM + 4 * r + (lane & 3)) * 16 + 4 * (s16 >> 2)]) = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < NRV; ++r) {
                        float v = racc[c2][r];
                        v = xor32_sum(xor16_sum(v));
                        if (g == 0) rem[(c2 * NREM + r) * 16 + s16] = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int cb = 2 * p + c2;
                f4 pa = (f4){0.f, 0.f, 0.f, 0.f}, pb = pa;
                const f4 al = al2[c2];
#pragma unroll
                for (int t = 0; t < kFullTiles; ++t) ln_bwd_tile(acc[c2][t], t, cb, xs2[c2][t], st2[t], al, s1[t], s2[t], pa, pb);
                if (s16 < NREM) {
                    const f4 rv = *reinterpret_cast<const f4*>(&rem[(c2 * NREM + s16) * 16 + 4 * g]);
                    ln_bwd_tile(rv, kFullTiles, cb, xs2[c2][kFullTiles], st2[kFullTiles], al, s1[kFullTiles], s2[kFullTiles], pa, pb);
                }
                write_colpart(pa, l, 1, cb);
                write_colpart(pb, l, 2, cb);
            }
            __builtin_amdgcn_wave_barrier();
        }
        ln_bwd_finish(s1, s2, X2, S2);
        fresh();
        // ---- 4. dA1 = G * SiLU'(A1): token-weight-gradient operand + wave-private columns of the MFMA operand ------
        {
            const float* A1 = a.a1 + (size_t)l * LR * kD;
            float* dA1 = a.da1 + (size_t)l * LR * kD;
            f4 avs[kCB][kNT];                 // as in step 1: the whole A1 tile in flight together
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb)
#pragma unroll
                for (int t = 0; t < kNT; ++t) avs[cb][t] = *reinterpret_cast<const f4*>(A1 + (size_t)growc_of(t) * kD + chw + 16 * cb);
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb) {
#pragma unroll
                for (int t = 0; t < kNT; ++t) {
                    if (16 * t + s16 >= R) continue;
                    const int gr = grow_of(t);
                    f4 d = (f4){0.f, 0.f, 0.f, 0.f};
                    const f4 av = avs[cb][t];
                    if (gr >= 0) {
                        d = silu_grad4(av, G[cb][t]);
                        *reinterpret_cast<f4*>(dA1 + (size_t)gr * kD + chw + 16 * cb) = d;
                    }
                    *reinterpret_cast<f4*>(&U[row_of(t) * kUStride + chw + 16 * cb]) = d;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();      // token mixing contracts over rows: a wave reads back only its own columns
        fresh();
        // ---- 5./6. dU1 = WW^T . dA1 and the LayerNorm-1 backward ----------------------------------------------------
#pragma unroll
        for (int t = 0; t < kNT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
        const float* X1 = a.x1 + (size_t)l * LR * kD;
        const float* S1 = a.s1 + (size_t)l * LR * 2;
        {
            const wrsrc_t wws = wrsrc(a.wwT_img);
            const int wwb = l * kNT * MK * 256;
            typedef const __attribute__((address_space(3))) float* ldsp;
            constexpr int MB = (MK + 7) / 8;
            ldsp upb[MB];
#pragma unroll
            for (int k = 0; k < MB; ++k) upb[k] = (ldsp)(U + 64 * w + s16 + (32 * k + g) * kUStride);
            ldsp uplast = (ldsp)(U + 64 * w + s16 + min(4 * (MK - 1) + g, R - 1) * kUStride);
#pragma unroll
            for (int k = 0; k < MB; ++k) asm volatile("" : "+v"(upb[k]));
            asm volatile("" : "+v"(uplast));
            f4 pa[kCB], pb[kCB];
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb) { pa[cb] = (f4){0.f, 0.f, 0.f, 0.f}; pb[cb] = pa[cb]; }
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                float Bt[MK];
#pragma unroll
                for (int m = 0; m < MK; ++m)
                    if (tokmix_needed(S, t, m)) Bt[m] = wload1(wws, lane * 4, wwb + (t * MK + m) * 256);
                // what the LayerNorm backward of this tile needs: issued before the tile's MFMAs, pinned there
                const f2 st1 = *reinterpret_cast<const f2*>(S1 + (size_t)growc_of(t) * 2);
                f4 xs[kCB], al1[kCB];
#pragma unroll
                for (int cb = 0; cb < kCB; ++cb) {
                    xs[cb] = *reinterpret_cast<const f4*>(X1 + (size_t)growc_of(t) * kD + chw + 16 * cb);
                    al1[cb] = *g4(a.ln1a + l * kD + chw + 16 * cb);
                }
                __builtin_amdgcn_sched_barrier(0);
                f4 acc[kCB];
#pragma unroll
                for (int cb = 0; cb < kCB; ++cb) acc[cb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MK; ++m) {
                    if (tokmix_needed(S, t, m)) {
                        ldsp src = (4 * m + 3 < R) ? upb[m >> 3] + (4 * (m & 7)) * kUStride : uplast;
#pragma unroll
                        for (int cb = 0; cb < kCB; ++cb) acc[cb] = MFMA(src[16 * cb], Bt[m], acc[cb]);
                    }
                }
#pragma unroll
                for (int cb = 0; cb < kCB; ++cb)
                    ln_bwd_tile(acc[cb], t, cb, xs[cb], st1, al1[cb], s1[t], s2[t], pa[cb], pb[cb]);
            }
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb) {
                write_colpart(pa[cb], l, 3, cb);
                write_colpart(pb[cb], l, 4, cb);
            }
        }
        ln_bwd_finish(s1, s2, X1, S1);
        // d(timestep embedding): x1 = x0 + emb broadcast over the tokens -> per-sample sum over tokens of G, per layer
#pragma unroll
        for (int cb = 0; cb < kCB; ++cb) {
            f4 h0 = (f4){0.f, 0.f, 0.f, 0.f}, h1 = h0;
#pragma unroll
            for (int t = 0; t < kNT; ++t) {
                const int r = 16 * t + s16;
                if (r < S) h0 += G[cb][t];
                else if (r < R) h1 += G[cb][t];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h0[j] = row16_sum(h0[j]);
                h1[j] = row16_sum(h1[j]);
            }
            if (s16 == 0) {
                float* dst = a.dembp + ((size_t)l * a.B + 2 * b) * kD + chw + 16 * cb;
                *reinterpret_cast<f4*>(dst) = h0;
                if (2 * b + 1 < a.B) *reinterpret_cast<f4*>(dst + kD) = h1;
            }
        }
    }
    // gradient with respect to the token sequences entering layer 0
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
        const int gr = grow_of(t);
        if (gr >= 0)
#pragma unroll
            for (int cb = 0; cb < kCB; ++cb) *reinterpret_cast<f4*>(a.g + (size_t)gr * kD + chw + 16 * cb) = G[cb][t];
    }
}

hipError_t init_mixer_bwd() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mixer_bwd<35>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)step_lds_bytes(kTED));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_mixer_bwd<36>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)step_lds_bytes(kBEAT));
}

hipError_t launch_mixer_bwd(Variant v, const MixerBwdArgs& a, hipStream_t st) {
    const int wgs = (a.B + 1) / 2;
    if (v == kTED) hipLaunchKernelGGL(k_mixer_bwd<35>, dim3(wgs), dim3(512), step_lds_bytes(kTED), st, a);
    else hipLaunchKernelGGL(k_mixer_bwd<36>, dim3(wgs), dim3(512), step_lds_bytes(kBEAT), st, a);
    return hipGetLastError();
}

}  // namespace ls
