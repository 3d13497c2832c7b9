// "One workgroup per CFG pass" variant of the fused diffusion step, for the bf16x3 split-precision mode.
//
// Why a second decomposition.  With the channel/token contractions on the bf16 matrix cores the step is no longer
// bound by the fp32 FMA lanes but by the LayerNorm / SiLU / operand-conversion phases and their barriers (rocprofv3:
// 46 % of wave time in s_waitcnt/s_barrier, matrix pipe 35 % busy, profiles/r01c).  ls_step.hip keeps one 8-wave
// workgroup (150 KB LDS) per CU, so all eight waves stall at the same barriers.  Here a workgroup is ONE sequence
// (the cond or the uncond pass of a sample): 4 waves, 35/36 rows = 3 token tiles, 74 KB LDS -> TWO independent
// workgroups per CU (one wave of each per SIMD) that hide each other's stalls and overlap one's VALU phases with the
// other's MFMAs.  The price is row padding 48/35 on the (now cheap) MFMAs and a tiny second kernel for the CFG lerp
// + sampler update, because the two passes of a sample no longer meet inside one workgroup.
//
// MEASURED RESULT (MI355X, B=512): this variant is SLOWER than the fused kernel, 0.84 vs 0.67 ms/step.  Two workgroups
// per CU are co-resident as intended, but each streams the full 1 MB/layer of hi+lo weights for 48 rows (fused: 80),
// doubling L2->CU weight traffic to ~10 TB/s, and waves wait MORE (57 % vs 46 % in s_waitcnt).  It is kept as precision
// mode LS_PRECISION_BF16X3_PERPASS for A/B runs (it is parity-tested); the product uses ls_step.hip for both precisions.
//
// Same math, layouts and reference citations as ls_step.hip (see there); wave w owns channels [128w, 128w+128).
#include "ls_step_common.h"

namespace ls {

constexpr int kSW = 4;     // waves per workgroup
constexpr int kSCB = 8;    // 16-channel blocks per wave
constexpr int kSNT = 3;    // token tiles

#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int S, int NPRE, int JF>
__global__ __launch_bounds__(256, 2) void k_seq(const StepArgs a) {
    constexpr int R = S;
    constexpr int KXQ = (JF + 15) / 16, KXP = KXQ * 16;
    constexpr int NOB = (JF + 15) / 16;
    constexpr int NG = (R + 7) / 8;            // 8-row groups of the transposed token-mix operand
    constexpr int KS = (R + 31) / 32;          // bf16 MFMA k steps over source rows
    constexpr int kGrp = 64 * 8 + 16;          // bf16 per 8-row group of a 64-channel half (+32 B against bank aliasing)
    constexpr int kPlane = NG * kGrp;          // one wave-private plane
    constexpr int NU = NOB * kSNT, MAXU = (NU + kSW - 1) / kSW;
    static_assert(R <= 16 * kSNT && KS == 2, "one sequence = 3 token tiles, 2 k steps");
    static_assert(kSW * 2 * kPlane * 2 <= R * kUStride * 4, "wave-private token-mix planes must fit the operand buffer");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    f2* pst = reinterpret_cast<f2*>(smem);                     // [4 waves][48 rows] (mean, M2)
    float* U = smem + 2 * kSW * 16 * kSNT;                     // [R][520] fp32, or two bf16 planes [R][520]

    const int b = blockIdx.x >> 1, sq = blockIdx.x & 1;        // sample, CFG pass (0 cond, 1 uncond)
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int s16 = lane & 15, g = lane >> 4, chw = 128 * w + 4 * g;
    auto fresh = [&]() {            // phase-local addressing (see ls_step.hip: keeps the residual stream out of scratch)
        asm volatile("" : "+v"(lane));
        s16 = lane & 15;
        g = lane >> 4;
        chw = 128 * w + 4 * g;
    };
    auto row_of = [&](int t) { return 16 * t + s16; };
    auto valid_of = [&](int t) { return (16 * t + 15 < R) ? true : (16 * t + s16 < R); };
    auto rowc_of = [&](int t) { const int r = 16 * t + s16; return (16 * t + 15 < R || r < R) ? r : R - 1; };

    f4 X[kSCB][kSNT];

    // ================= embedding (RAG.py:110-122, 184-192) ==========================================
    {
        const unsigned long long gidx = a.call ? a.call->sample_offset + (unsigned long long)b : (unsigned long long)b;
#pragma unroll
        for (int t = 0; t < kSNT; ++t) {
            const int tk = rowc_of(t);
#pragma unroll
            for (int cb = 0; cb < kSCB; ++cb) {
                const int ch = chw + 16 * cb;
                f4 v = (f4){0.f, 0.f, 0.f, 0.f};
                if (valid_of(t)) {
                    if (tk >= NPRE) {
                        const float* st = (sq ? a.static_u : a.static_c) + ((size_t)b * kT + (tk - NPRE)) * kD + ch;
                        v = *reinterpret_cast<const f4*>(st);
                    } else if (tk == 0) {
                        const f4 mu = *reinterpret_cast<const f4*>(a.z_mu + (size_t)b * kD + ch);
                        const f4 sd = *reinterpret_cast<const f4*>(a.z_std + (size_t)b * kD + ch);
                        f4 e;
                        const float* ep = sq ? a.eps_u : a.eps_c;
                        if (ep) {
                            e = *reinterpret_cast<const f4*>(ep + (size_t)b * kD + ch);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                e[j] = philox_normal(a.call, gidx, a.step_id, 1u + sq, (unsigned)(ch + j));
                        }
                        v = mu + e * sd;
                    } else {
                        v = *reinterpret_cast<const f4*>(a.emo_tok + (size_t)b * kD + ch);
                    }
                }
                X[cb][t] = v;
            }
        }
        const float* xin = a.x_in + (size_t)b * kT * JF;
        for (int idx = tid; idx < R * KXP; idx += 256) {
            const int r = idx / KXP, k = idx - r * KXP;
            float v = 0.f;
            if (r >= NPRE && k < JF) v = xin[(r - NPRE) * JF + k];
            U[r * kUStride + k] = v;
        }
        __syncthreads();
        fresh();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f4 acc[4][kSNT];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < kSNT; ++t) acc[c][t] = X[4 * h + c][t];
            gf4p wp = g4(a.W->winx_seq_img) + (size_t)(w * 2 + h) * KXQ * 4 * 64 + lane;
#pragma unroll 2
            for (int q = 0; q < KXQ; ++q) {
                f4 A[4], Bv[kSNT];
#pragma unroll
                for (int c = 0; c < 4; ++c) A[c] = wp[(q * 4 + c) * 64];
#pragma unroll
                for (int t = 0; t < kSNT; ++t)
                    Bv[t] = *reinterpret_cast<const f4*>(&U[rowc_of(t) * kUStride + 16 * q + 4 * g]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int t = 0; t < kSNT; ++t) acc[c][t] = MFMA(A[c][j], Bv[t][j], acc[c][t]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < kSNT; ++t) X[4 * h + c][t] = valid_of(t) ? acc[c][t] : (f4){0.f, 0.f, 0.f, 0.f};
        }
    }

    auto dump_trace = [&](int stage) {
        if (!a.trace) return;
        float* tr = a.trace + (((size_t)b * (a.layers + 1) + stage) * 2 + sq) * S * kD;
#pragma unroll
        for (int t = 0; t < kSNT; ++t)
            if (valid_of(t))
#pragma unroll
                for (int cb = 0; cb < kSCB; ++cb)
                    *reinterpret_cast<f4*>(tr + (size_t)row_of(t) * kD + chw + 16 * cb) = X[cb][t];
    };
    dump_trace(0);

    // LayerNorm statistics: two-pass over the lane's 32 channels, Chan merges across lane groups and the 4 waves
    float mean[kSNT], rstd[kSNT];
    auto ln_stats = [&]() {
#pragma unroll
        for (int t = 0; t < kSNT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int cb = 0; cb < kSCB; ++cb) s += (X[cb][t][0] + X[cb][t][1]) + (X[cb][t][2] + X[cb][t][3]);
            float m = s * (1.0f / 32.0f), m2 = 0.f;
#pragma unroll
            for (int cb = 0; cb < kSCB; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = X[cb][t][j] - m;
                    m2 = fmaf(d, d, m2);
                }
            {
                const float mo = __shfl_xor(m, 16), m2o = __shfl_xor(m2, 16);
                const float d = mo - m;
                m2 = (m2 + m2o) + d * d * 16.0f;
                m = 0.5f * (m + mo);
            }
            {
                const float mo = __shfl_xor(m, 32), m2o = __shfl_xor(m2, 32);
                const float d = mo - m;
                m2 = (m2 + m2o) + d * d * 32.0f;
                m = 0.5f * (m + mo);
            }
            if (g == 0) pst[w * 48 + 16 * t + s16] = (f2){m, m2};
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kSNT; ++t) {
            f2 pw[kSW];
            float ms = 0.f, m2s = 0.f;
#pragma unroll
            for (int ww = 0; ww < kSW; ++ww) {
                pw[ww] = pst[ww * 48 + 16 * t + s16];
                ms += pw[ww].x;
                m2s += pw[ww].y;
            }
            const float mt = ms * (1.0f / kSW);
            float dd = 0.f;
#pragma unroll
            for (int ww = 0; ww < kSW; ++ww) {
                const float d = pw[ww].x - mt;
                dd = fmaf(d, d, dd);
            }
            mean[t] = mt;
            rstd[t] = rsqrtf((m2s + 128.0f * dd) * (1.0f / kD) + 1e-5f);
        }
    };

    // ================= TransMLP: 8 x MLPblock (mlp_module.py:67-91) ================================
    for (int l = 0; l < a.layers; ++l) {
        fresh();
        {
            const float* te = a.temb + (size_t)b * a.temb_stride + chw;
            f4 e[kSCB];
#pragma unroll
            for (int cb = 0; cb < kSCB; ++cb) e[cb] = *reinterpret_cast<const f4*>(te + 16 * cb);
#pragma unroll
            for (int cb = 0; cb < kSCB; ++cb)
#pragma unroll
                for (int t = 0; t < kSNT; ++t)
                    if (valid_of(t)) X[cb][t] += e[cb];
        }
        // ---- block1: LN -> token mixing (bf16x3, wave-private transposed operand) -> SiLU -> residual ----
        ln_stats();
        fresh();
        {
            float nmr[kSNT];
#pragma unroll
            for (int t = 0; t < kSNT; ++t) nmr[t] = -mean[t] * rstd[t];
            __bf16* Th = reinterpret_cast<__bf16*>(U) + (size_t)w * 2 * kPlane;     // this wave's planes
            __bf16* Tl = Th + kPlane;
            gbf8p wwh = (gbf8p)(const bf8*)(a.W->ww_seq_hi_img) + (size_t)l * kSNT * KS * 64 + lane;
            gbf8p wwl = (gbf8p)(const bf8*)(a.W->ww_seq_lo_img) + (size_t)l * kSNT * KS * 64 + lane;
            const int slot = ((s16 & 3) << 2) | (s16 >> 2);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // (a) LN1 of the 64 channels of this half, split hi/lo, stored as UT[row/8][channel slot][row%8]
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cb = 4 * h + c;
                    const f4 al = *g4(a.W->ln1a + l * kD + chw + 16 * cb);
                    const f4 be = *g4(a.W->ln1b + l * kD + chw + 16 * cb);
#pragma unroll
                    for (int t = 0; t < kSNT; ++t) {
                        const int r = row_of(t);
                        const int o = (r >> 3) * kGrp + (16 * c + g) * 8 + (r & 7);
                        if (valid_of(t)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float u = fmaf(fmaf(X[cb][t][j], rstd[t], nmr[t]), al[j], be[j]);
                                const __bf16 hi = (__bf16)u;
                                Th[o + 32 * j] = hi;
                                Tl[o + 32 * j] = (__bf16)(u - (float)hi);
                            }
                        } else if (r < 8 * NG) {          // rows R..8*NG-1: finite zeros (their weights are zero)
#pragma unroll
                            for (int j = 0; j < 4; ++j) { Th[o + 32 * j] = (__bf16)0.f; Tl[o + 32 * j] = (__bf16)0.f; }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // (b) out[channel][row] = sum_r' u[r'][channel] * Wt[row][r']
#pragma unroll
                for (int t = 0; t < kSNT; ++t) {
                    const float bt = valid_of(t) ? g1(a.W->btok_seq)[l * 48 + row_of(t)] : 0.f;
                    f4 acc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = (f4){bt, bt, bt, bt};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const bf8 Bh = wwh[(t * KS + ks) * 64], Bl = wwl[(t * KS + ks) * 64];
                        const int grp = (4 * ks + 3 < NG || 4 * ks + g < NG) ? 4 * ks + g : NG - 1;
                        bf8 Ah[4], Al[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            Ah[c] = *reinterpret_cast<const bf8*>(Th + grp * kGrp + (16 * c + slot) * 8);
                            Al[c] = *reinterpret_cast<const bf8*>(Tl + grp * kGrp + (16 * c + slot) * 8);
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = MFMA_BF(Al[c], Bh, acc[c]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = MFMA_BF(Ah[c], Bl, acc[c]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = MFMA_BF(Ah[c], Bh, acc[c]);
                    }
                    if (valid_of(t)) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int j = 0; j < 4; ++j) X[4 * h + c][t][j] = silu_acc(acc[c][j], X[4 * h + c][t][j]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- block2: LN -> channel mixing (bf16x3) -> SiLU -> residual ---------------------------------
        ln_stats();        // its barrier also orders every wave's private token-mix reads before the shared stores below
        fresh();
        {
            float nmr[kSNT];
#pragma unroll
            for (int t = 0; t < kSNT; ++t) nmr[t] = -mean[t] * rstd[t];
            __bf16* Uh = reinterpret_cast<__bf16*>(U);
            __bf16* Ul = Uh + R * kUStride;
#pragma unroll
            for (int cb = 0; cb < kSCB; ++cb)
#pragma unroll
                for (int t = 0; t < kSNT; ++t)
                    if (valid_of(t)) {
                        bf4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float u = fmaf(X[cb][t][j], rstd[t], nmr[t]);       // LN2 alpha/beta folded into W', b'
                            hi[j] = (__bf16)u;
                            lo[j] = (__bf16)(u - (float)hi[j]);
                        }
                        *reinterpret_cast<bf4*>(&Uh[row_of(t) * kUStride + chw + 16 * cb]) = hi;
                        *reinterpret_cast<bf4*>(&Ul[row_of(t) * kUStride + chw + 16 * cb]) = lo;
                    }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            fresh();
            f4 acc[4][kSNT];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f4 bc = *g4(a.W->bch + l * kD + chw + 16 * (4 * p + c));
#pragma unroll
                for (int t = 0; t < kSNT; ++t) acc[c][t] = bc;
            }
            const size_t wofs = ((size_t)((l * kSW + w) * 2 + p) * 16) * 4 * 64 + lane;
            gbf8p wh = (gbf8p)(const bf8*)(a.W->wch_seq_hi_img) + wofs;
            gbf8p wl = (gbf8p)(const bf8*)(a.W->wch_seq_lo_img) + wofs;
            const __bf16* Uh = reinterpret_cast<const __bf16*>(U);
            const __bf16* Ul = Uh + R * kUStride;
            int rofs[kSNT];
#pragma unroll
            for (int t = 0; t < kSNT; ++t) rofs[t] = rowc_of(t) * kUStride + 8 * g;
            bf8 Ahn[4], Aln[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { Ahn[c] = wh[c * 64]; Aln[c] = wl[c * 64]; }
#pragma unroll 2
            for (int q = 0; q < 16; ++q) {
                bf8 Ah[4], Al[4], Bh[kSNT], Bl[kSNT];
#pragma unroll
                for (int c = 0; c < 4; ++c) { Ah[c] = Ahn[c]; Al[c] = Aln[c]; }
                const int qn = (q + 1 < 16) ? q + 1 : 15;
#pragma unroll
                for (int c = 0; c < 4; ++c) { Ahn[c] = wh[(qn * 4 + c) * 64]; Aln[c] = wl[(qn * 4 + c) * 64]; }
#pragma unroll
                for (int t = 0; t < kSNT; ++t) {
                    Bh[t] = *reinterpret_cast<const bf8*>(Uh + rofs[t] + 32 * q);
                    Bl[t] = *reinterpret_cast<const bf8*>(Ul + rofs[t] + 32 * q);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < kSNT; ++t) acc[c][t] = MFMA_BF(Al[c], Bh[t], acc[c][t]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < kSNT; ++t) acc[c][t] = MFMA_BF(Ah[c], Bl[t], acc[c][t]);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < kSNT; ++t) acc[c][t] = MFMA_BF(Ah[c], Bh[t], acc[c][t]);
            }
            fresh();
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < kSNT; ++t)
                    if (valid_of(t)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) X[4 * p + c][t][j] = silu_acc(acc[c][t][j], X[4 * p + c][t][j]);
                    }
        }
        dump_trace(l + 1);
    }

    // ================= OutputProcess.poseFinal (RAG.py:205-211), bf16x3 through the LDS planes =====
    fresh();
    __syncthreads();                       // every wave is done reading the last LN2 operand
    {
        __bf16* Uh = reinterpret_cast<__bf16*>(U);
        __bf16* Ul = Uh + R * kUStride;
#pragma unroll
        for (int cb = 0; cb < kSCB; ++cb)
#pragma unroll
            for (int t = 0; t < kSNT; ++t)
                if (valid_of(t)) {
                    bf4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hi[j] = (__bf16)X[cb][t][j];
                        lo[j] = (__bf16)(X[cb][t][j] - (float)hi[j]);
                    }
                    *reinterpret_cast<bf4*>(&Uh[row_of(t) * kUStride + chw + 16 * cb]) = hi;
                    *reinterpret_cast<bf4*>(&Ul[row_of(t) * kUStride + chw + 16 * cb]) = lo;
                }
    }
    __syncthreads();
    {
        const __bf16* Uh = reinterpret_cast<const __bf16*>(U);
        const __bf16* Ul = Uh + R * kUStride;
        float* outp = a.out_raw + ((size_t)(b * 2 + sq) * kT) * JF;
#pragma unroll 1
        for (int i = 0; i < MAXU; ++i) {
            const int u = w + kSW * i;                 // wave-uniform work unit (out-block, token tile)
            if (u >= NU) break;
            const int ob = u / kSNT, t = u - ob * kSNT;
            const int r = 16 * t + s16;
            const int rc = r < R ? r : R - 1;
            gbf8p wh = (gbf8p)(const bf8*)(a.W->wout_hi_img) + (size_t)ob * 16 * 64 + lane;
            gbf8p wl = (gbf8p)(const bf8*)(a.W->wout_lo_img) + (size_t)ob * 16 * 64 + lane;
            const int ro = rc * kUStride + 8 * g;
            f4 a0 = (f4){0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const bf8 Ah = wh[q * 64], Al = wl[q * 64];
                const bf8 Bh = *reinterpret_cast<const bf8*>(Uh + ro + 32 * q);
                const bf8 Bl = *reinterpret_cast<const bf8*>(Ul + ro + 32 * q);
                a0 = MFMA_BF(Al, Bh, a0);
                a1 = MFMA_BF(Ah, Bl, a1);
                a2 = MFMA_BF(Ah, Bh, a2);
            }
            const f4 res = (a0 + a1) + a2;
            if (r < R && r >= NPRE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 16 * ob + 4 * g + j;
                    if (c < JF) outp[(size_t)(r - NPRE) * JF + c] = res[j] + g1(a.W->bout)[c];
                }
            }
        }
    }
}

// CFG lerp (cfg_sampler.py:31) + posterior / DDIM update (gaussian_diffusion.py:260-282, 507-558, 745-798) for the
// per-pass variant: out_raw[b][pass][T][JF] -> x_{t-1}.  Element order and arithmetic identical to k_step's epilogue.
__global__ __launch_bounds__(256) void k_cfg_update(const StepArgs a, int JF) {
    const int b = blockIdx.x;
    const float sc = a.scale ? a.scale[b] : 1.0f;
    const size_t base = (size_t)b * kT * JF;
    const float* oc_p = a.out_raw + (size_t)(b * 2) * kT * JF;
    const float* ou_p = oc_p + (size_t)kT * JF;
    const unsigned long long gidx = a.call ? a.call->sample_offset + (unsigned long long)b : (unsigned long long)b;
    for (int idx = threadIdx.x; idx < kT * JF; idx += 256) {
        const int f = idx / JF, c = idx - f * JF;
        const float oc = oc_p[idx], ou = ou_p[idx];
        if (a.fwd_c) a.fwd_c[base + idx] = oc;
        if (a.fwd_u) a.fwd_u[base + idx] = ou;
        float x0 = ou + sc * (oc - ou);
        if (a.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        if (a.x0_out) a.x0_out[base + idx] = x0;
        if (a.sampler != kNone) {
            const float xt = a.x_in[base + idx];
            float nz = 0.f;
            if (a.t_nonzero) {
                if (a.noise) {
                    const size_t bn = a.const_noise ? 0 : (size_t)b;
                    nz = a.noise[(bn * JF + c) * kT + f];
                } else {
                    nz = philox_normal(a.call, gidx, a.step_id, 3u, (unsigned)(c * kT + f));
                }
            }
            float xn;
            if (a.sampler == kDDPM) {
                xn = a.c0 * x0 + a.c1 * xt;
                if (a.t_nonzero) xn += a.c2 * nz;
            } else {
                const float eps = (a.c0 * xt - x0) / a.c1;
                xn = x0 * a.c2 + a.c3 * eps;
                if (a.t_nonzero) xn += a.c4 * nz;
            }
            a.x_out[base + idx] = xn;
        }
    }
}

size_t seq_lds_bytes(Variant v) {
    const int S = (v == kTED) ? 35 : 36;
    return (size_t)(S * kUStride + 2 * kSW * 16 * kSNT) * sizeof(float);
}

hipError_t init_seq_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_seq<35, 1, 27>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(kTED));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_seq<36, 2, 282>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(kBEAT));
}

hipError_t launch_step_seq(Variant v, const StepArgs& a, int batch, hipStream_t st) {
    const size_t lds = seq_lds_bytes(v);
    if (v == kTED) hipLaunchKernelGGL((k_seq<35, 1, 27>), dim3(2 * batch), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_seq<36, 2, 282>), dim3(2 * batch), dim3(256), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cfg_update, dim3(batch), dim3(256), 0, st, a, v == kTED ? 27 : 282);
    return hipGetLastError();
}

}  // namespace ls
