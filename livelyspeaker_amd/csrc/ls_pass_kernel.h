// One-pass-per-workgroup diffusion step for gfx950 (MI355X): one launch = one p_sample / ddim_sample step, like k_step
// (ls_step_kernel.h), but a workgroup holds ONE CFG pass of one sample -- S = 35 | 36 rows, ~78 KB of LDS.  What that buys over k_step's
// one-sample-per-CU mapping is granularity: 128 clips put a workgroup on every CU (k_step: 256), and a batch of 256 k + r clips pays for
// r in half-CU units.  Same reference arithmetic as k_step:
//   ClassifierFreeSampleModel.forward   scripts/model/cfg_sampler.py:24-31   (two INDEPENDENT forwards: what makes the split legal)
//   RAG.forward                         scripts/model/RAG.py:98-133
//   TransMLP / MLPblock / LN_spatial    scripts/model/mlp_module.py:21-91
//   OutputProcess                       scripts/model/RAG.py:205-211
//   p_mean_variance / p_sample / ddim_sample   scripts/diffusion/gaussian_diffusion.py:284-399, 507-558, 745-798
//
// Mapping:
//   * workgroup = (sample b, pass p), blockIdx = b * npass + p.  Two forms of the one template (NW waves):
//       NW = 8   64 channels per wave, as in k_step; ONE workgroup per CU (its registers leave no room for a second).  Two waves per SIMD
//                hide each other's LDS / L2 round trips.  Used when the grid fits the chip once (ls_api.cpp run_pass).
//       NW = 4   128 channels per wave, up to 256 VGPRs; TWO independent workgroups per CU, not phase-locked by a common barrier (wave
//                priority keeps them level).  Used for grids beyond one workgroup per CU.
//     Wave w owns channels [CHW w, CHW w + CHW) of all rows in the MFMA C/D layout (lane & 15 = row of the tile, 4 (lane >> 4) + reg = channel
//     of the 16-channel block), resident in registers for the whole forward.  Rows 32 .. S-1 (3 | 4 of a third tile's 16) are multiplied on
//     the VALU (TED) or by v_mfma_f32_4x4x1 (BEAT) in channel mixing, as k_step treats its ragged tile; TED fp32 also HOLDS them dense
//     (LS_PASS_DENSE below).
//   * the weight images are k_step's (ls_api.cpp build_fused_images): global 16-channel block gb = (8-wave slice gb >> 2, pass (gb >> 1) & 1,
//     c2 = gb & 1), so no second copy of the weights exists (plus wtail / wtok1_hi / wtok1_lo: a few KB for the one-pass token mixing).
//   * CFG combination: each pass writes its poseFinal output [T][J*F] write-through (sc1), every wave drains, barrier, one lane takes a
//     ticket (relaxed agent-scope fetch_add on the sample's counter: zeroed by ls_prepare, and set back to zero by the second taker).
//     The workgroup that draws the odd ticket is the LAST of its sample: it reads the other pass's output with sc1 loads, combines the
//     two in pass order (result independent of which one arrived last), and applies the sampler update.  Nobody waits for anybody:
//     correct for any dispatch order, placement or residency (cdna_hip_programming.md section 6, Guideline 16, counter form).
//     npass = 1 (every guidance scale is 1): no hand-off at all, the grid is one workgroup per sample.
#pragma once
#include "ls_step_common.h"
#include "ls_lanes.h"

namespace ls {

constexpr int kPassNT = 3;                 // 16-row tiles of one pass (S = 35 | 36 -> 48)

// NW waves per workgroup.  4 (128 channels per wave, up to 256 VGPRs at two workgroups per CU): the form that SHARES a CU.  8 (64
// channels per wave, as in k_step): for launches that put at most ONE workgroup on a CU -- two waves per SIMD of the same workgroup hide
// each other's LDS / L2 round trips, which a lone 4-wave workgroup cannot (its SIMDs hold one wave each); its registers (up to 256 per
// wave x 8 waves) leave no room for a second workgroup, so it is never used where two per CU are wanted.
__host__ __device__ constexpr int pass_lds_floats(int S, int NW) {
    // psum [NW waves][48 rows] (mean, M2) | U [S][520] | REM [NW waves][4 blocks][S - 32 rows][16]
    return 2 * NW * 16 * kPassNT + S * kUStride + NW * 4 * (S - 32) * 16;
}

typedef unsigned pass_u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned* pass_gu32p;

// Wave priority (the CU's two workgroups are independent and compete for the SIMDs): 0 none; 1 falls with the layer index (the workgroup
// that is ahead yields, so the two stay level and neither is left to finish alone); 2 the same for the long products only, the
// LayerNorm / token-mixing / epilogue phases always at the top level
#ifndef LS_PASS_PRIO
#define LS_PASS_PRIO 2
#endif
// fp32: rows 32 .. S-1 of the residual stream live DENSE -- lane = channel of a 64-channel half, one register per row (6 | 8 VGPRs) --
// instead of as a third MFMA C/D tile of which 3 | 4 of 16 lanes are real (32 VGPRs, and a full tile's worth of vector instructions in
// every LayerNorm / SiLU phase for them).  The launch is power-bound when two workgroups share a CU (docs/DESIGN_NOTES_r5.md), so the
// instructions not issued are the gain.  bf16x3 keeps the tile form (its products for the ragged rows are padded MFMAs).
#ifndef LS_PASS_DENSE
#define LS_PASS_DENSE 1
#endif
#ifndef LS_PASS_PFD
#define LS_PASS_PFD 2                       // bf16x3 channel mixing: weight fragments requested this many k blocks ahead
#endif
#ifndef LS_PASS_BPREF
#define LS_PASS_BPREF 1                     // channel mixing: the LDS operands of k block q + 1 are requested while block q is multiplied
#endif

// PREC = 1: bf16x3 split precision (opt-in, as in k_step): operands u = hi + lo as two bf16 planes, W.u ~= hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (two terms do not meet the contract: docs/DESIGN_NOTES_r5.md).
template <int S, int NPRE, int JF, int PREC = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void k_pass(const StepArgs a) {
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr int kPassThreads = 64 * NW;
    constexpr int CB = 32 / NW;              // 16-channel blocks owned by one wave
    constexpr int CHW = 16 * CB;             // channels owned by one wave: 128 | 64
    constexpr int NH = CHW / 64;             // 64-channel halves of a wave's channels (the dense ragged rows: lane = channel of a half)
    constexpr int KXQ = (JF + 15) / 16;      // 16-wide k groups of the x_t part of input_mapping
    constexpr int KXP = KXQ * 16;
    constexpr int NOB = (JF + 15) / 16;      // 16-wide output blocks of poseFinal
    constexpr int OSTR = NOB * 16 + 4;
    constexpr int NT = kPassNT;
    constexpr int NREM = S - 32;             // rows of the ragged third tile: 3 (TED) | 4 (BEAT)
    constexpr bool kDense = LS_PASS_DENSE != 0 && PREC == 0 && (NREM % 4 != 0);      // BEAT's 4 ragged rows stay on v_mfma_f32_4x4x1 in the tile form (measured: the dense form's 4 VALU rows cost it 6 %)
    constexpr int NTX = kDense ? 2 : NT;     // row tiles of the residual stream held in the MFMA C/D layout
    constexpr bool kRemMfma = !kDense && (NREM % 4 == 0);
    constexpr int NRG = kRemMfma ? NREM / 4 : 1;
    constexpr int NRV = kRemMfma ? 1 : NREM;
    constexpr int MK = (S + 3) / 4;          // k steps of the token-mix GEMM
    constexpr int MQ = (MK + 3) / 4;         // ... in groups of four (one 16-byte weight fragment per lane)
    constexpr int NU = NOB * NT;             // output-projection work units (wide outputs)
    constexpr int MAXU = (NU + NW - 1) / NW;
    constexpr int KS = (S + 31) / 32;        // k steps (32 source rows) of the bf16 token-mix MFMA
    static_assert(S > 32 && S <= 36, "one pass = two full row tiles + a ragged one of at most 4 rows");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* psum = smem;                          // [NW][48] (mean, M2) pairs of the LayerNorm merge
    float* U = smem + 2 * NW * 16 * NT;          // [S][520] fp32 operand
    float* REM = U + S * kUStride;               // [NW waves][4][NREM][16] ragged-row patch (tile form only)

    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np = a.npass;
    const int bid = blockIdx.x;
    const int p = np == 2 ? (bid & 1) : 0;
    const int bl = np == 2 ? (bid >> 1) : bid;      // launch-local sample
    const int b = a.b0 + bl;                        // sample of the prepared batch
    const bool unc = p == 1;
    int s16 = lane & 15;
    int g = lane >> 4;
    int chw = CHW * w + 4 * g;                      // + 16 cb + j = this lane's channels
    // see k_step: laundering the lane id at phase boundaries keeps per-lane addresses phase-local, so the residual stream stays in registers
    auto fresh = [&]() {
        asm volatile("" : "+v"(lane));
        s16 = lane & 15;
        g = lane >> 4;
        chw = CHW * w + 4 * g;
    };
    auto row_of = [&](int t) { return 16 * t + s16; };
    auto valid_of = [&](int t) { return t < 2 ? true : (s16 < NREM); };
    auto rowc_of = [&](int t) { return t < 2 ? 16 * t + s16 : min(32 + s16, S - 1); };

    f4 X[CB][NT];                                   // kDense: tile 2 exists only while embedding and for poseFinal
    float XR[NH][NREM];                             // kDense: rows 32 + r, channel CHW w + 64 h + lane

    auto stamp = [&](int idx) {
#ifdef LS_DEBUG
        if (a.prof && bid == a.prof_wg && lane == 0 && idx < kProfPoints) a.prof[w * kProfPoints + idx] = __builtin_amdgcn_s_memtime();
        if (a.wgt && tid == 0 && (idx == 0 || idx == 4 + 8 * a.layers)) a.wgt[2 * bid + (idx ? 1 : 0)] = __builtin_amdgcn_s_memtime();
        if (a.wgt && tid == 0 && idx == 0 && bid < 2048)
            a.wgt[2048 + bid] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
#else
        (void)idx;
#endif
    };
    stamp(0);

    // ================= embedding: InputProcess + input_mapping (RAG.py:110-114, 184-192) ==========
    {
        const unsigned long long goff = a.call ? a.call->sample_offset : 0ull;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tk = rowc_of(t);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int ch = chw + 16 * cb;
                f4 v = (f4){0.f, 0.f, 0.f, 0.f};
                if (valid_of(t)) {
                    if (tk >= NPRE) {
                        v = *reinterpret_cast<const f4*>((unc ? a.static_u : a.static_c) + ((size_t)b * kT + (tk - NPRE)) * kD + ch);
                    } else if (tk == 0) {
                        // style token: reparameterize(mu, logvar)  (RAG.py:10-13, 116-120)
                        const f4 mu = *reinterpret_cast<const f4*>(a.z_mu + (size_t)b * kD + ch);
                        const f4 sd = *reinterpret_cast<const f4*>(a.z_std + (size_t)b * kD + ch);
                        f4 e;
                        const float* ep = unc ? a.eps_u : a.eps_c;
                        if (ep) {
                            e = *reinterpret_cast<const f4*>(ep + (size_t)b * kD + ch);
                        } else {
                            float z[4];            // this lane's 4 consecutive channels = one Philox block
                            philox_normal4(a.call, goff + (unsigned long long)b, a.step_id, unc ? 2u : 1u, (unsigned)(ch >> 2), z);
                            e = (f4){z[0], z[1], z[2], z[3]};
                        }
                        v = mu + e * sd;
                    } else {
                        v = *reinterpret_cast<const f4*>(a.emo_tok + (size_t)b * kD + ch);      // BEAT emotion token (scripts_beat/model/RAG.py:125-126)
                    }
                }
                X[cb][t] = v;
            }
        }
        // x_t of this sample -> LDS [S][KXP] (zero for prefix tokens and pad columns): loads first, then the writes, in blocks
        constexpr int NIT = (S * KXP + kPassThreads - 1) / kPassThreads;
        constexpr int CH = 14;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += CH) {
            float xv[CH];
#pragma unroll
            for (int itl = 0; itl < CH; ++itl) {
                if (it0 + itl >= NIT) break;
                const int idx = min(tid + kPassThreads * (it0 + itl), S * KXP - 1);
                const int r = idx / KXP, k = idx - r * KXP;
                const bool live = r >= NPRE && k < JF;
                xv[itl] = a.x_in[(size_t)b * kT * JF + (live ? (r - NPRE) * JF + k : 0)];
                if (!live) xv[itl] = 0.f;
            }
#pragma unroll
            for (int itl = 0; itl < CH; ++itl) {
                if (it0 + itl >= NIT) break;
                const int idx = tid + kPassThreads * (it0 + itl);
                if (idx < S * KXP) {
                    const int r = idx / KXP;
                    U[r * kUStride + (idx - r * KXP)] = xv[itl];
                }
            }
        }
        __syncthreads();
        fresh();
        // winx_img[8][2][KXQ][2][64][4]: 16-channel block 8 w + cb = (8-wave slice 2 w + (cb >> 2), pass (cb >> 1) & 1, c2 = cb & 1)
        const wrsrc_t wrs = wrsrc(a.W->winx_img);
#pragma unroll
        for (int pp = 0; pp < CB / 2; ++pp) {           // two channel blocks at a time: 6 accumulators
            f4 acc[2][NT];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[c2][t] = X[2 * pp + c2][t];
            const int gb0 = CB * w + 2 * pp;            // global 16-channel block = (8-wave slice gb >> 2, pass (gb >> 1) & 1, c2 = gb & 1)
            const int wsb = ((gb0 >> 2) * 2 + ((gb0 >> 1) & 1)) * KXQ * 2 * 1024;
            f4 An[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) An[c2] = wload4(wrs, lane * 16, wsb + c2 * 1024);
#pragma unroll 2
            for (int q = 0; q < KXQ; ++q) {
                f4 A[2], Bv[NT];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) A[c2] = An[c2];
                const int qn = q + 1 < KXQ ? q + 1 : KXQ - 1;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) An[c2] = wload4(wrs, lane * 16, wsb + (qn * 2 + c2) * 1024);
#pragma unroll
                for (int t = 0; t < NT; ++t) Bv[t] = *reinterpret_cast<const f4*>(&U[rowc_of(t) * kUStride + 16 * q + 4 * g]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[c2][t] = MFMA(A[c2][j], Bv[t][j], acc[c2][t]);
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int t = 0; t < NT; ++t) X[2 * pp + c2][t] = valid_of(t) ? acc[c2][t] : (f4){0.f, 0.f, 0.f, 0.f};   // pad rows stay zero
        }
        if constexpr (kDense) {
            // tile 2 -> the dense form, through this wave's own columns of the (now free) operand buffer
            __syncthreads();                        // every wave has read the last x_t operand
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                if (s16 < NREM) *reinterpret_cast<f4*>(&U[(32 + s16) * kUStride + chw + 16 * cb]) = X[cb][2];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int r = 0; r < NREM; ++r) XR[h][r] = U[(32 + r) * kUStride + CHW * w + 64 * h + lane];
        }
    }
    stamp(1);

    // LN_spatial statistics over the 512 channels of each row (mlp_module.py:29-33): two passes over the lane's 32 channels, then Chan's
    // parallel-variance merge over the 4 lane groups (two cross-lane exchanges) and the 4 waves (LDS) -- one workgroup barrier per LayerNorm.
    float mean[NT], rstd[NT];
    float meanR[NREM], rstdR[NREM];                   // kDense: the ragged rows' statistics (wave-uniform)
    auto ln_stats = [&]() {
        f2* pst = reinterpret_cast<f2*>(psum);            // [4 waves][48 rows] (mean, M2) of 128 channels
        if constexpr (kDense) {
            // a row = this wave's 128 channels over the 64 lanes x 2 halves; sums over the wave on the VALU (DPP within a row of 16 lanes,
            // then the two lane-swap exchanges): every lane ends with the total, no trip through SGPRs
            auto wsum = [&](float v) { return xor32_sum(xor16_sum(row16_sum(v))); };
            float mr[NREM], qr[NREM];
#pragma unroll
            for (int r = 0; r < NREM; ++r) {
                float sr = XR[0][r];
                if constexpr (NH == 2) sr += XR[NH - 1][r];
                mr[r] = wsum(sr) * (1.0f / CHW);
            }
#pragma unroll
            for (int r = 0; r < NREM; ++r) {
                const float d0 = XR[0][r] - mr[r];
                float dq = d0 * d0;
                if constexpr (NH == 2) { const float d1 = XR[NH - 1][r] - mr[r]; dq = fmaf(d1, d1, dq); }
                qr[r] = wsum(dq);
            }
#pragma unroll
            for (int r = 0; r < NREM; ++r)
                if (lane == 0) pst[w * 48 + 32 + r] = (f2){mr[r], qr[r]};
        }
#pragma unroll
        for (int t = 0; t < NTX; ++t) {
            f4 sv = X[0][t];
#pragma unroll
            for (int cb = 1; cb < CB; ++cb) sv += X[cb][t];
            const float s = (sv[0] + sv[1]) + (sv[2] + sv[3]);
            float m = s * (1.0f / (4 * CB));
            const f4 mv = (f4){m, m, m, m};
            f4 qv = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f4 d = X[cb][t] - mv;
                qv = __builtin_elementwise_fma(d, d, qv);
            }
            float m2 = (qv[0] + qv[1]) + (qv[2] + qv[3]);
            {   // the lane group 16 lanes away (4 CB + 4 CB values), then 32 lanes away (8 CB + 8 CB)
                float ma, mb, qa, qb;
                xor16_pair(m, ma, mb);
                xor16_pair(m2, qa, qb);
                const float d = mb - ma;
                m2 = (qa + qb) + d * d * (2.0f * CB);
                m = 0.5f * (ma + mb);
            }
            {
                float ma, mb, qa, qb;
                xor32_pair(m, ma, mb);
                xor32_pair(m2, qa, qb);
                const float d = mb - ma;
                m2 = (qa + qb) + d * d * (4.0f * CB);
                m = 0.5f * (ma + mb);
            }
            if (g == 0) pst[w * 48 + 16 * t + s16] = (f2){m, m2};
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTX; ++t) {
            f2 pw[NW];
            f2 acc2 = (f2){0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                pw[ww] = pst[ww * 48 + 16 * t + s16];
                acc2 += pw[ww];
            }
            const float mt = acc2.x * (1.0f / NW);
            float dd = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { const float d = pw[ww].x - mt; dd = fmaf(d, d, dd); }
            mean[t] = mt;
            rstd[t] = rsqrtf((acc2.y + (float)CHW * dd) * (1.0f / kD) + 1e-5f);
        }
        if constexpr (kDense) {
#pragma unroll
            for (int r = 0; r < NREM; ++r) {
                f2 pw[NW];
                f2 acc2 = (f2){0.f, 0.f};
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    pw[ww] = pst[ww * 48 + 32 + r];
                    acc2 += pw[ww];
                }
                const float mt = acc2.x * (1.0f / NW);
                float dd = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) { const float d = pw[ww].x - mt; dd = fmaf(d, d, dd); }
                meanR[r] = mt;
                rstdR[r] = rsqrtf((acc2.y + (float)CHW * dd) * (1.0f / kD) + 1e-5f);
            }
        }
    };
    // the normalised operand of this lane's channels -> LDS [row][520]; LN1 applies alpha / beta here, LN2's are folded into the
    // channel-mix weights on the host (W' = W diag(alpha), b' = b + W beta)
    auto ln_store = [&](auto affine, const f4 (&alv)[CB], const f4 (&bev)[CB], const float (&alR)[NH], const float (&beR)[NH]) {
        constexpr bool alpha = decltype(affine)::value;
        if constexpr (kDense) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int r = 0; r < NREM; ++r) {
                    float u = (XR[h][r] - meanR[r]) * rstdR[r];
                    if (alpha) u = fmaf(u, alR[h], beR[h]);
                    U[(32 + r) * kUStride + CHW * w + 64 * h + lane] = u;
                }
        }
        float nmr[NT];
#pragma unroll
        for (int t = 0; t < NTX; ++t) nmr[t] = -mean[t] * rstd[t];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int t = 0; t < NTX; ++t)
                if (valid_of(t)) {
                    const f4 rs = (f4){rstd[t], rstd[t], rstd[t], rstd[t]}, nm = (f4){nmr[t], nmr[t], nmr[t], nmr[t]};
                    f4 u = __builtin_elementwise_fma(X[cb][t], rs, nm);
                    if (alpha) u = __builtin_elementwise_fma(u, alv[cb], bev[cb]);
                    if constexpr (PREC == 1) {
                        // u = hi + lo (+ O(2^-17 |u|)), hi = bf16_rne(u), lo = bf16_rne(u - hi): two bf16 planes [S][520] in the fp32 buffer's space
                        __bf16* Uh = reinterpret_cast<__bf16*>(U);
                        __bf16* Ul = Uh + S * kUStride;
                        bf4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            hi[j] = (__bf16)u[j];
                            lo[j] = (__bf16)(u[j] - (float)hi[j]);
                        }
                        *reinterpret_cast<bf4*>(&Uh[row_of(t) * kUStride + chw + 16 * cb]) = hi;
                        *reinterpret_cast<bf4*>(&Ul[row_of(t) * kUStride + chw + 16 * cb]) = lo;
                    } else {
                        *reinterpret_cast<f4*>(&U[row_of(t) * kUStride + chw + 16 * cb]) = u;
                    }
                }
    };

    // ================= TransMLP: 8 x MLPblock (mlp_module.py:67-91) ================================
    for (int l = 0; l < a.layers; ++l) {
        fresh();
        if (LS_PASS_PRIO == 1) {
            if (l < 2) __builtin_amdgcn_s_setprio(3); else if (l < 4) __builtin_amdgcn_s_setprio(2); else if (l < 6) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        } else if (LS_PASS_PRIO == 2) {
            __builtin_amdgcn_s_setprio(3);
        }
        {   // x = x + emb  (re-added at the input of EVERY block, mlp_module.py:68-69, 88-89)
            const float* te = a.temb + (size_t)b * a.temb_stride + chw;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f4 e = *reinterpret_cast<const f4*>(te + 16 * cb);
#pragma unroll
                for (int t = 0; t < NTX; ++t)
                    if (valid_of(t)) X[cb][t] += e;
            }
            if constexpr (kDense) {
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float e = a.temb[(size_t)b * a.temb_stride + CHW * w + 64 * h + lane];
#pragma unroll
                    for (int r = 0; r < NREM; ++r) XR[h][r] += e;
                }
            }
        }
        // ---- block1: LN -> token-mixing Conv1d(S,S,1) -> SiLU -> residual -------------------------
        f4 alv[CB], bev[CB];                        // LN1 affine: in flight during the statistics and their barrier
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            alv[cb] = wload4(wrsrc(a.W->ln1a), chw * 4, (l * kD + 16 * cb) * 4);
            bev[cb] = wload4(wrsrc(a.W->ln1b), chw * 4, (l * kD + 16 * cb) * 4);
        }
        float alR[NH], beR[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) { alR[h] = 1.f; beR[h] = 0.f; }
        if constexpr (kDense) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                alR[h] = g1(a.W->ln1a)[l * kD + CHW * w + 64 * h + lane];
                beR[h] = g1(a.W->ln1b)[l * kD + CHW * w + 64 * h + lane];
            }
        }
        ln_stats();
        stamp(2 + 8 * l);
        fresh();
        ln_store(std::true_type{}, alv, bev, alR, beR);
        // no workgroup barrier: token mixing contracts over ROWS, wave w reads back only the 128 channel columns it has just written
        __builtin_amdgcn_wave_barrier();
        stamp(3 + 8 * l);
        fresh();
        if constexpr (PREC == 1) {
            // bf16x3 token mixing: the operand stays row-major (the two bf16 planes) and the MFMA A fragment -- 8 consecutive source rows
            // of one channel per lane -- is gathered by ds_read_b64_tr_b16 (see k_step).  wtok1_hi / lo [l][t][ks][lane][8] =
            // Wt[16 t + (lane & 15)][32 ks + 8 (lane >> 4) + e], zero outside S x S.
            typedef short s4v __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s4v* lds4;
            const wrsrc_t wrh = wrsrc(a.W->wtok1_hi_img), wrl = wrsrc(a.W->wtok1_lo_img);
            const int wsb = l * NT * KS * 1024;
            bf8 Bh[NT][KS], Bl[NT][KS];
            float bt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    Bh[t][ks] = wload8h(wrh, lane * 16, wsb + (t * KS + ks) * 1024);
                    Bl[t][ks] = wload8h(wrl, lane * 16, wsb + (t * KS + ks) * 1024);
                }
                bt[t] = g1(a.W->btok_rows)[l * 80 + rowc_of(t)];
            }
            const __bf16* Ph = reinterpret_cast<const __bf16*>(U);
            const __bf16* Pl = Ph + S * kUStride;
            int ro[KS][2];                                  // rows past the last one are clamped: their weights are 0 and the clamped row is finite
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ro[ks][0] = min(32 * ks + 8 * g + (s16 >> 2), S - 1) * kUStride + CHW * w + 4 * (s16 & 3);
                ro[ks][1] = min(32 * ks + 8 * g + 4 + (s16 >> 2), S - 1) * kUStride + CHW * w + 4 * (s16 & 3);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                bf8 Ah[KS], Al[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const s4v h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Ph + ro[ks][0] + 16 * cb));
                    const s4v h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Ph + ro[ks][1] + 16 * cb));
                    const s4v l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Pl + ro[ks][0] + 16 * cb));
                    const s4v l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(Pl + ro[ks][1] + 16 * cb));
                    Ah[ks] = __builtin_bit_cast(bf8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    Al[ks] = __builtin_bit_cast(bf8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
                f4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = (f4){bt[t], bt[t], bt[t], bt[t]};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[ks], Bh[t][ks], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[ks], Bl[t][ks], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[ks], Bh[t][ks], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (valid_of(t)) X[cb][t] = silu_acc4(acc[t], X[cb][t]);
            }
        } else {
            // out[ch][r] = sum_r' u[r'][ch] * Wt[r][r'] + bt[r] as D[channel][row]: A = u^T from LDS, B = the Conv1d weights (one pass:
            // wtok1_img[l][t][mq][lane][j] = Wt[16 t + (lane & 15)][4 (4 mq + j) + (lane >> 4)]).  Channel block by channel block: the
            // block's MK source values are read once and meet the three row tiles (three independent accumulators).
            const wrsrc_t wrs = wrsrc(a.W->wtok1_img);
            const int wsb = l * NT * MQ * 1024;
            f4 Bt[NT][MQ];
            float bt[NT];
#pragma unroll
            for (int t = 0; t < NTX; ++t) {
#pragma unroll
                for (int m = 0; m < MQ; ++m) Bt[t][m] = wload4(wrs, lane * 16, wsb + (t * MQ + m) * 1024);
                bt[t] = g1(a.W->btok_rows)[l * 80 + rowc_of(t)];
            }
            if constexpr (kDense) {
                // ragged OUTPUT rows on v_mfma_f32_4x4x1 (16 blocks of 4 rows x 4 channels, k = 1): A = Wt[32 + (lane & 3)][k] (the same
                // in every block; wtail[l][k][4]), B = u[k][this lane's channel], and D lands exactly in the dense form (lane = channel,
                // register = row).  2 S instructions of 16 clocks instead of 8 x MK padded 16x16x4 MFMAs of 32.
                float wt[S];
#pragma unroll
                for (int k = 0; k < S; ++k) wt[k] = g1(a.W->wtail)[(l * S + k) * 4 + (lane & 3)];
                f4 acc4[NH][2];                                 // [half][k parity]: independent accumulator chains
#pragma unroll
                for (int h = 0; h < NH; ++h) { acc4[h][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc4[h][1] = acc4[h][0]; }
                typedef const __attribute__((address_space(3))) float* ldsq;
                ldsq ucol = (ldsq)(U + CHW * w + lane);
#pragma unroll
                for (int k = 0; k < S; ++k)
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        acc4[h][k & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wt[k], ucol[k * kUStride + 64 * h], acc4[h][k & 1], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < NREM; ++r) {
                    const float br = g1(a.W->btok_rows)[l * 80 + 32 + r];
#pragma unroll
                    for (int h = 0; h < NH; ++h) XR[h][r] = silu_acc(acc4[h][0][r] + acc4[h][1][r] + br, XR[h][r]);
                }
            }
            typedef const __attribute__((address_space(3))) float* ldsp;
            ldsp up[MK];
#pragma unroll
            for (int m = 0; m < MK; ++m) up[m] = (ldsp)(U + min(4 * m + g, S - 1) * kUStride + CHW * w + s16);   // clamped rows meet zero weights
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                float av[MK];
#pragma unroll
                for (int m = 0; m < MK; ++m) av[m] = up[m][16 * cb];
                f4 acc[NT];
#pragma unroll
                for (int t = 0; t < NTX; ++t) acc[t] = (f4){bt[t], bt[t], bt[t], bt[t]};
#pragma unroll
                for (int m = 0; m < MK; ++m)
#pragma unroll
                    for (int t = 0; t < NTX; ++t) acc[t] = MFMA(av[m], Bt[t][m >> 2][m & 3], acc[t]);
#pragma unroll
                for (int t = 0; t < NTX; ++t)
                    if (valid_of(t)) X[cb][t] = silu_acc4(acc[t], X[cb][t]);
            }
        }
        stamp(4 + 8 * l);
        fresh();
        // ---- block2: LN -> channel-mixing Linear(512,512) -> SiLU -> residual ---------------------
        ln_stats();            // its barrier also orders every wave's token-mix reads before the stores below
        stamp(5 + 8 * l);
        ln_store(std::false_type{}, alv, bev, alR, beR);
        __syncthreads();
        stamp(6 + 8 * l);
        if (LS_PASS_PRIO == 2) {
            if (l < 3) __builtin_amdgcn_s_setprio(2); else if (l < 6) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (PREC == 1) {
            // ---- bf16x3: W'.u ~= hi_w.hi_u + hi_w.lo_u + lo_w.hi_u (bf16 x bf16 products are exact in fp32; the dropped lo.lo term and the
            // split residuals are O(2^-16) relative); all three row tiles padded (bf16 MFMAs are cheap), two channel blocks at a time.
            // wch_hi / lo [L][8][2][16 q][2][64][8]: block 8 w + 2 pp + c2 = (8-wave slice 2 w + (pp >> 1), pass pp & 1, c2)
#pragma unroll
            for (int pp = 0; pp < CB / 2; ++pp) {
                fresh();
                f4 acc[2][NT];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const f4 bc = wload4(wrsrc(a.W->bch), chw * 4, (l * kD + 16 * (2 * pp + c2)) * 4);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[c2][t] = bc;
                }
                const wrsrc_t wrh = wrsrc(a.W->wch_hi_img), wrl = wrsrc(a.W->wch_lo_img);
                const int gb0 = CB * w + 2 * pp;
                const int wsb = ((((l * 8 + (gb0 >> 2)) * 2 + ((gb0 >> 1) & 1)) * 16) * 2) * 1024;
                const __bf16* Uh = reinterpret_cast<const __bf16*>(U);
                const __bf16* Ul = Uh + S * kUStride;
                int rofs[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) rofs[t] = rowc_of(t) * kUStride + 8 * g;
                // One wave per SIMD and workgroup: nothing but this wave's own prefetch hides the L2 round trip of the weight fragments, and a
                // k block is only 18 bf16 MFMAs (~290 clocks) long -- fragments are requested PFD blocks ahead (fully unrolled: the stages
                // are renamed, not moved)
                constexpr int PFD = LS_PASS_PFD;
                bf8 Ahq[PFD][2], Alq[PFD][2];
#pragma unroll
                for (int k = 0; k < PFD; ++k)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        Ahq[k][c2] = wload8h(wrh, lane * 16, wsb + (k * 2 + c2) * 1024);
                        Alq[k][c2] = wload8h(wrl, lane * 16, wsb + (k * 2 + c2) * 1024);
                    }
                // the LDS operands are software-pipelined too: the hi plane of block q + 1 is requested at the top of block q, its lo plane
                // once the hi plane of block q has met its last MFMA (36 live operand registers instead of 48 for a full double buffer)
                bf8 Bhn[NT], Bln[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    Bhn[t] = *reinterpret_cast<const bf8*>(Uh + rofs[t]);
                    Bln[t] = *reinterpret_cast<const bf8*>(Ul + rofs[t]);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    bf8 Ah[2], Al[2], Bh[NT], Bl[NT];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) { Ah[c2] = Ahq[q % PFD][c2]; Al[c2] = Alq[q % PFD][c2]; }
#pragma unroll
                    for (int t = 0; t < NT; ++t) { Bh[t] = Bhn[t]; Bl[t] = Bln[t]; }
                    // pinned: left to itself the scheduler sinks each request to just ahead of its use (one L2 round trip per fragment)
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + PFD < 16) {
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            Ahq[q % PFD][c2] = wload8h(wrh, lane * 16, wsb + ((q + PFD) * 2 + c2) * 1024);
                            Alq[q % PFD][c2] = wload8h(wrl, lane * 16, wsb + ((q + PFD) * 2 + c2) * 1024);
                        }
                    }
                    if (q + 1 < 16) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) Bhn[t] = *reinterpret_cast<const bf8*>(Uh + rofs[t] + 32 * (q + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // term-major order: 6 independent accumulators between two MFMAs on the same one
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[c2][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[c2], Bh[t], acc[c2][t], 0, 0, 0);
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[c2][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[c2], Bh[t], acc[c2][t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + 1 < 16) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) Bln[t] = *reinterpret_cast<const bf8*>(Ul + rofs[t] + 32 * (q + 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[c2][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[c2], Bl[t], acc[c2][t], 0, 0, 0);
                }
                fresh();
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (valid_of(t)) X[2 * pp + c2][t] = silu_acc4(acc[c2][t], X[2 * pp + c2][t]);
                if (pp == CB / 4 - 1) stamp(7 + 8 * l);
            }
        } else {
        // Rows 32 .. S-1 on the VALU (TED) / v_mfma_f32_4x4x1 (BEAT) from the same A-operand registers, as in k_step.
#pragma unroll
        for (int pp = 0; pp < CB / 4; ++pp) {            // 4 channel blocks x 2 full tiles = 8 accumulators per half
            fresh();
            f4 acc[4][2];
            float racc[4][NRV];
            f4 racc4[4][NRG];
            f4 bcv[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const f4 bc = wload4(wrsrc(a.W->bch), chw * 4, (l * kD + 16 * (4 * pp + c4)) * 4);
                bcv[c4] = bc;
                acc[c4][0] = bc; acc[c4][1] = bc;
#pragma unroll
                for (int r = 0; r < NRV; ++r) racc[c4][r] = 0.f;
#pragma unroll
                for (int r = 0; r < NRG; ++r) racc4[c4][r] = (f4){0.f, 0.f, 0.f, 0.f};
            }
            // wch_img[L][8][2][32 q][2][64][4]: block 8 w + 4 pp + c4 = (8-wave slice 2 w + pp, pass c4 >> 1, c2 = c4 & 1)
            const wrsrc_t wrs = wrsrc(a.W->wch_img);
            const int wsb = ((l * 8 + ((CB * w) >> 2) + pp) * 2 * 32) * 2 * 1024;
            auto woff = [&](int q, int c4) { return wsb + ((((c4 >> 1) * 32 + q) * 2) + (c4 & 1)) * 1024; };
            typedef const __attribute__((address_space(3))) float* ldsp;
            typedef const __attribute__((address_space(3))) f4* ldsp4;
            ldsp ub0 = (ldsp)(U + s16 * kUStride + 4 * g);                 // tile t: + 16 t rows
            ldsp ur = (ldsp)(U + (32 + (kRemMfma ? (lane & 3) : 0)) * kUStride + 4 * g);
            asm volatile("" : "+v"(ub0), "+v"(ur));
            f4 An[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) An[c4] = wload4(wrs, lane * 16, woff(0, c4));
            // LDS operands: the two full tiles' fragments meet the first MFMA of the k block, so they are requested one block ahead
            // (LS_PASS_BPREF; one wave per SIMD and workgroup: nobody else hides the LDS round trip); the ragged rows' are first used
            // behind eight MFMAs and are read at the top of their own block
            f4 Bn[2];
            auto ldb = [&](int q) {
#pragma unroll
                for (int t = 0; t < 2; ++t) Bn[t] = *(ldsp4)(ub0 + 16 * t * kUStride + 16 * q);
            };
            if (LS_PASS_BPREF) ldb(0);
#pragma unroll 2
            for (int q = 0; q < 32; ++q) {
                f4 A[4], Bv[2], Ur[kRemMfma ? NRG : NRV];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) A[c4] = An[c4];
                {
                    const int qn = (q + 1 < 32) ? q + 1 : 31;       // branch-free prefetch (the last one re-reads q = 31)
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) An[c4] = wload4(wrs, lane * 16, woff(qn, c4));
                }
                if (!LS_PASS_BPREF) ldb(q);
#pragma unroll
                for (int t = 0; t < 2; ++t) Bv[t] = Bn[t];
#pragma unroll
                for (int r = 0; r < (kRemMfma ? NRG : NRV); ++r) Ur[r] = *(ldsp4)(ur + (kRemMfma ? 4 : 1) * r * kUStride + 16 * q);
                // per k: [8 MFMAs][4 * NREM scalar FMAs], order pinned (k_step: the compiler's own order stalls on the FMAs' ds_reads)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[c4][t] = MFMA(A[c4][j], Bv[t][j], acc[c4][t]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (LS_PASS_BPREF && j == 0) ldb((q + 1 < 32) ? q + 1 : 31);
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        if constexpr (kRemMfma) {
#pragma unroll
                            for (int r = 0; r < NRG; ++r)
                                racc4[c4][r] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[c4][j], Ur[r][j], racc4[c4][r], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int r = 0; r < NRV; ++r) racc[c4][r] = fmaf(A[c4][j], Ur[r][j], racc[c4][r]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            fresh();
            if constexpr (kDense) {
                // ragged rows: lane (g, s16) accumulated channel 16 c4 + s16 of every block c4 over its k subset; summed over the lane groups
                // every lane holds the totals of all four blocks, and block g's is the one of this lane's dense channel 64 pp + lane
                const float bcR = g1(a.W->bch)[l * kD + CHW * w + 64 * pp + lane];
#pragma unroll
                for (int r = 0; r < NREM; ++r) {
                    float v[4];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) v[c4] = xor32_sum(xor16_sum(racc[c4][r]));
                    const float sel = g == 0 ? v[0] : g == 1 ? v[1] : g == 2 ? v[2] : v[3];
                    XR[pp][r] = silu_acc(sel + bcR, XR[pp][r]);
                }
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                    for (int t = 0; t < 2; ++t) X[4 * pp + c4][t] = silu_acc4(acc[c4][t], X[4 * pp + c4][t]);
            } else {
            // ragged rows: sum the 4 k subsets of the lane groups, then [channel-lane][row] -> [row-lane][channel-reg] through a per-wave patch
            float* rem = REM + w * (4 * NREM * 16);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                if constexpr (kRemMfma) {
#pragma unroll
                    for (int r = 0; r < NRG; ++r) {
                        f4 v = racc4[c4][r];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = xor32_sum(xor16_sum(v[i]));
                        // lane (block = lane >> 2, row = lane & 3) holds channels 4 (s16 >> 2) .. + 3 of row 4 r + (lane & 3)
                        if (g == 0) *reinterpret_cast<f4*>(&rem[(c4 * NREM + 4 * r + (lane & 3)) * 16 + 4 * (s16 >> 2)]) = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < NRV; ++r) {
                        const float v = xor32_sum(xor16_sum(racc[c4][r]));
                        if (g == 0) rem[(c4 * NREM + r) * 16 + s16] = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int cb = 4 * pp + c4;
#pragma unroll
                for (int t = 0; t < 2; ++t) X[cb][t] = silu_acc4(acc[c4][t], X[cb][t]);
                if (s16 < NREM) {
                    const f4 rv = *reinterpret_cast<const f4*>(&rem[(c4 * NREM + s16) * 16 + 4 * g]);
                    X[cb][2] = silu_acc4(rv + bcv[c4], X[cb][2]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            }
            if (pp == 0) stamp(7 + 8 * l);
        }
        }
        stamp(9 + 8 * l);
    }

    // ================= OutputProcess.poseFinal (RAG.py:205-211) ====================================
    stamp(2 + 8 * a.layers);
    fresh();
    if (LS_PASS_PRIO) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                       // every wave is done reading the last LN2 operand: U is free
    constexpr bool kOutFromRegs = (NOB <= 2);
    constexpr int OROWS = kOutFromRegs ? NW * S : S;
    static_assert(OROWS * OSTR <= S * kUStride, "OUT overlay must fit the operand buffer");
    float* OUT = U;
    if constexpr (kDense && kOutFromRegs) {
        // the ragged rows back into the tile form the product below takes its B operand in: through this wave's own columns of rows 32 ..
        // of the operand buffer (beyond the OUT overlay)
        static_assert(OROWS * OSTR <= 32 * kUStride, "the tile-2 patch must not meet the OUT overlay");
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < NREM; ++r) U[(32 + r) * kUStride + CHW * w + 64 * h + lane] = XR[h][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
            X[cb][2] = s16 < NREM ? *reinterpret_cast<const f4*>(&U[(32 + s16) * kUStride + chw + 16 * cb]) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (kOutFromRegs) {
        // Narrow output (TED): every wave contracts over ITS OWN 128 channels straight from the residual registers (a valid MFMA B
        // operand; wout_reg_img carries the matching k permutation), writes a [S][32] partial; the 4 partials are summed below.
        f4 acc[NOB][NT];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[ob][t] = (f4){0.f, 0.f, 0.f, 0.f};
        const wrsrc_t wrs = wrsrc(a.W->wout_reg_img);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                // wout_reg_img[8][NOB][4][64][4]: 8-wave slice 2 w + (cb >> 2), block cb & 3
                const f4 A = wload4(wrs, lane * 16, ((((CB * w + cb) >> 2) * NOB + ob) * 4 + ((CB * w + cb) & 3)) * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[ob][t] = MFMA(A[j], X[cb][t][j], acc[ob][t]);
            }
        stamp(3 + 8 * a.layers);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (valid_of(t)) *reinterpret_cast<f4*>(&OUT[(w * S + row_of(t)) * OSTR + 16 * ob + 4 * g]) = acc[ob][t];
    } else {
#pragma unroll
        for (int t = 0; t < NTX; ++t)
            if (valid_of(t))
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<f4*>(&U[row_of(t) * kUStride + chw + 16 * cb]) = X[cb][t];
        if constexpr (kDense) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int r = 0; r < NREM; ++r) U[(32 + r) * kUStride + CHW * w + 64 * h + lane] = XR[h][r];
        }
        __syncthreads();
        f4 res[MAXU];
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = w + NW * i;          // wave-uniform
            res[i] = (f4){0.f, 0.f, 0.f, 0.f};
            if (u < NU) {
                const int ob = u / NT, t = u - ob * NT;
                const int rc = min(16 * t + s16, S - 1);
                const wrsrc_t wrs = wrsrc(a.W->wout_img);
                const int wsb = ob * 32 * 1024;
                const float* up = &U[rc * kUStride + 4 * g];
                f4 a0 = (f4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
                constexpr int QB = 8;              // weight fragments in flight ahead of their use
                f4 An[QB];
#pragma unroll
                for (int k = 0; k < QB; ++k) An[k] = wload4(wrs, lane * 16, wsb + k * 1024);
#pragma unroll 1
                for (int q0 = 0; q0 < 32; q0 += QB) {
                    f4 A[QB];
#pragma unroll
                    for (int k = 0; k < QB; ++k) A[k] = An[k];
                    const int qn = q0 + QB < 32 ? q0 + QB : q0;
#pragma unroll
                    for (int k = 0; k < QB; ++k) An[k] = wload4(wrs, lane * 16, wsb + (qn + k) * 1024);
#pragma unroll
                    for (int k = 0; k < QB; ++k) {
                        const f4 Bv = *reinterpret_cast<const f4*>(up + 16 * (q0 + k));
                        a0 = MFMA(A[k][0], Bv[0], a0);
                        a1 = MFMA(A[k][1], Bv[1], a1);
                        a0 = MFMA(A[k][2], Bv[2], a0);
                        a1 = MFMA(A[k][3], Bv[3], a1);
                    }
                }
                res[i] = a0 + a1;
            }
        }
        stamp(3 + 8 * a.layers);
        fresh();
        __syncthreads();                       // operand buffer is free: overlay OUT[row][c]
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = w + NW * i;
            if (u < NU) {
                const int ob = u / NT, t = u - ob * NT;
                const int r = 16 * t + s16;
                if (r < S) *reinterpret_cast<f4*>(&OUT[r * OSTR + 16 * ob + 4 * g]) = res[i];
            }
        }
    }
    __syncthreads();
    auto out_at = [&](int r, int c) {
        if constexpr (kOutFromRegs) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) v += OUT[(ww * S + r) * OSTR + c];
            return v;
        } else {
            return OUT[r * OSTR + c];
        }
    };

    // ====== hand-off of the pass output, CFG lerp (cfg_sampler.py:31), posterior / DDIM update (gaussian_diffusion.py:260-282,
    //        507-558, 745-798), written back in the internal [B][T][JF] layout ======================
    const float* other = nullptr;
    if (np == 2) {
        float* mine = a.pf + ((size_t)bl * 2 + p) * (kT * JF);
        const wrsrc_t prs = uniform_rsrc(mine);
        for (int idx = tid; idx < kT * JF; idx += kPassThreads) {
            const int f = idx / JF, c = idx - f * JF;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out_at(NPRE + f, c)), prs, idx * 4, 0, 16);     // sc1: write-through
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains
        __syncthreads();
        unsigned* flagw = reinterpret_cast<unsigned*>(psum);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add((pass_gu32p)(a.pcnt + bl), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flagw[0] = old & 1u;                              // odd ticket: the other pass of this step is already out
            // the later one hands the word back as it found it at the start of the step (0): the next step -- a later launch -- needs no
            // reset between launches (a memset node replayed ahead of the loop was observed to write a garbage pattern instead of zeros)
            if (old & 1u) __hip_atomic_store((pass_gu32p)(a.pcnt + bl), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!flagw[0]) { stamp(4 + 8 * a.layers); return; }   // first of the sample: done
        other = a.pf + ((size_t)bl * 2 + (1 - p)) * (kT * JF);
    }
    stamp(3 + 8 * a.layers + 1);
    {
        const float sc = (np == 2 && a.scale) ? a.scale[b] : 1.0f;
        const unsigned long long gidx = (a.call ? a.call->sample_offset : 0ull) + (unsigned long long)b;
        const size_t base = (size_t)b * kT * JF;
        const wrsrc_t ors = uniform_rsrc(np == 2 ? other : a.x_in);
        for (int idx = tid; idx < kT * JF; idx += kPassThreads) {
            const int f = idx / JF, c = idx - f * JF;
            const float bo = g1(a.W->bout)[c];
            const float own = out_at(NPRE + f, c);
            float x0;
            if (np == 2) {
                const float oth = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ors, idx * 4, 0, 16));   // sc1: never this CU's L1
                const float oc = (unc ? oth : own) + bo, ou = (unc ? own : oth) + bo;
                if (a.fwd_c) a.fwd_c[base + idx] = oc;
                if (a.fwd_u) a.fwd_u[base + idx] = ou;
                x0 = ou + sc * (oc - ou);
            } else {
                x0 = own + bo;                                // scale == 1: the CFG combination is the cond output
            }
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
    stamp(4 + 8 * a.layers);
}

}  // namespace ls
