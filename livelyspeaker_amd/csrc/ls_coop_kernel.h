// Sample-split diffusion step for SMALL batches on gfx950 (MI355X): one launch = one p_sample / ddim_sample step, like k_step
// (ls_step_kernel.h), but a sample is spread over 16 workgroups instead of one, so that 4 ... 32 clips fill the chip.
// Same reference arithmetic:
//   ClassifierFreeSampleModel.forward   scripts/model/cfg_sampler.py:24-31
//   RAG.forward                         scripts/model/RAG.py:98-133
//   TransMLP / MLPblock / LN_spatial    scripts/model/mlp_module.py:21-91   (per-sample independence, :67-91, is what makes the split legal)
//   OutputProcess                       scripts/model/RAG.py:205-211
//   p_mean_variance / p_sample / ddim_sample   scripts/diffusion/gaussian_diffusion.py:284-399, 507-558, 745-798
//
// Mapping:
//   * workgroup = (sample b, CFG pass p, channel slice c): 4 waves, the S = 35 | 36 rows of ONE pass x 64 of the 512 channels.
//     Wave w owns channels [64c + 16w, +16) of every row = 12 VGPRs of residual stream, in the MFMA C/D layout exactly as in k_step.
//     blockIdx = (b * npass + p) * 8 + c, so slice c of every sample runs on XCD c (observed round-robin placement): each XCD's L2
//     holds one eighth of the weights (1 MB).  Placement is a speed matter only -- every hand-off below is placement-independent.
//   * what crosses workgroups (the 8 slices of one (sample, pass)), per layer:
//       SYNC1  LayerNorm-1: (mean, M2) of each row over the slice's 64 channels                      288 B per workgroup
//       SYNC2  LayerNorm-2: the same, plus the slice of the raw rows x[S][64] (channel mixing contracts over all 512)  9 KB
//     and once per step the final rows (poseFinal contracts over all 512 channels; the CFG combination needs both passes).
//     Token mixing contracts over ROWS and stays inside the workgroup.
//   * hand-off protocol (cdna_hip_programming.md section 6, Guideline 16, forms R1 / R2): payload = 16-byte write-through (sc1)
//     stores, every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, then the row statistics are published as 8-byte
//     {tag, value} granules with relaxed agent-scope atomic stores -- the granules ARE the flags.  Consumers poll the granules with
//     relaxed agent-scope loads (sc1: L1 bypassed) and read the payload with sc1 loads.  Tags are unique per launch and sync point
//     (StepArgs::epoch + index); the granule words are zeroed by a memset node ahead of every call.  Every spin is bounded: on a
//     timeout the workgroup records it in StepArgs::cerr and carries on (the host then fails the call).
//   * single-buffered payload is safe: a slice rewrites its rows of layer l+1 only after SYNC1(l+1), which every consumer reaches
//     after its reads of layer l; the statistics alternate between two granule areas for the same reason.
#pragma once
#include "ls_step_common.h"
#include "ls_lanes.h"

namespace ls {

constexpr int kCoopThreads = 256;
constexpr int kCoopWaves = 4;
constexpr int kCoopSlices = 8;             // channel slices of 64
constexpr int kCoopRows = 36;              // rows of one pass in the exchange buffers (S <= 36)
constexpr int kCoopU1Stride = 80;          // LDS row stride of the token-mix operand [S][64]: = 16 mod 32, conflict-free column reads
constexpr unsigned kCoopSpinLimit = 1u << 18;       // polls (~1-2 us each) before a hand-off wait gives up: waits are < 1 ms when the slices are resident
// LDS: psum [4][48] f2 | stat [48] f2 | REM [4][4][16] | U [36][520]
constexpr int kCoopLdsFloats = 2 * kCoopWaves * 48 + 2 * 48 + kCoopWaves * 4 * 16 + kCoopRows * kUStride;

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long* gu64p;

__device__ __forceinline__ f4 ld_sc1(wrsrc_t r, int voff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 16));     // aux 16 = sc1: served from L2 / memory, never this CU's L1
}
__device__ __forceinline__ void st_sc1(f4 v, wrsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, voff, 0, 16);       // write-through
}
__device__ __forceinline__ unsigned long long gran_load(const unsigned long long* p) {
    return __hip_atomic_load((gu64p)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(unsigned long long* p, unsigned tag, float v) {
    __hip_atomic_store((gu64p)p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int S, int NPRE, int JF>
__global__ __launch_bounds__(kCoopThreads, 2) void k_coop(const StepArgs a) {
    constexpr int KXQ = (JF + 15) / 16;
    constexpr int KXP = KXQ * 16;
    constexpr int NOB = (JF + 15) / 16;
    constexpr int NT1 = 3;                      // 16-row tiles of one pass
    constexpr int NREM = S - 32;                // rows of the ragged third tile: 3 (TED) | 4 (BEAT)
    constexpr bool kRemMfma = (NREM % 4 == 0);  // BEAT: one v_mfma_f32_4x4x1 row group; TED: scalar FMAs (see k_step)
    constexpr int NRV = kRemMfma ? 1 : NREM;
    constexpr int MK1 = (S + 3) / 4;            // k steps of the token-mix GEMM of one pass
    constexpr int NU = NOB * NT1;               // output-projection work units (out block, row tile)
    static_assert(S > 32 && S <= kCoopRows, "one pass = two full row tiles + a ragged one");
    static_assert(NREM >= 1 && NREM <= 4, "ragged tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    f2* pst = reinterpret_cast<f2*>(smem);                         // [4 waves][48 rows] (mean, M2) over 16 channels
    f2* stat = pst + kCoopWaves * 48;                              // [48 rows] (mean, rstd) over all 512 channels
    float* REM = smem + 2 * kCoopWaves * 48 + 2 * 48;              // [4 waves][4][16] ragged-row patch
    float* U = REM + kCoopWaves * 4 * 16;                          // [36][520] channel-mix operand; overlays: x_t staging, token-mix operand

    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int c = bid & 7;                                          // channel slice
    const int np = a.npass;
    const int pg = bid >> 3;                                        // launch-local (sample, pass) group
    const int p = np == 2 ? (pg & 1) : 0;
    const int bl = np == 2 ? (pg >> 1) : pg;                        // launch-local sample
    const int b = a.b0 + bl;                                        // sample of the prepared batch
    const bool unc = p == 1;
    int s16 = lane & 15;
    int g = lane >> 4;
    int chw = 64 * c + 16 * w + 4 * g;                              // + j = this lane's channels
    auto fresh = [&]() {
        asm volatile("" : "+v"(lane));
        s16 = lane & 15;
        g = lane >> 4;
        chw = 64 * c + 16 * w + 4 * g;
    };
    auto valid_of = [&](int t) { return 16 * t + 15 < S ? true : 16 * t + s16 < S; };
    auto rowc_of = [&](int t) { const int r = 16 * t + s16; return (16 * t + 15 < S || r < S) ? r : S - 1; };

    float* xg = a.cx + (size_t)pg * kCoopRows * kD;                 // raw rows of this (sample, pass): [36][512]
    unsigned long long* gran = a.cgran + (size_t)pg * 2 * kCoopRows * kCoopSlices * 2;   // [2 areas][36 rows][8 slices][2] granules
    unsigned spin_bad = 0;

    f4 X[NT1];

    // ================= embedding: InputProcess + input_mapping (RAG.py:110-114, 184-192) ==========
    {
        const unsigned long long goff = a.call ? a.call->sample_offset : 0ull;
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int tk = rowc_of(t);
            f4 v = (f4){0.f, 0.f, 0.f, 0.f};
            if (valid_of(t)) {
                if (tk >= NPRE) {
                    v = *reinterpret_cast<const f4*>((unc ? a.static_u : a.static_c) + ((size_t)b * kT + (tk - NPRE)) * kD + chw);
                } else if (tk == 0) {                               // style token: reparameterize(mu, logvar)  (RAG.py:10-13, 116-120)
                    const f4 mu = *reinterpret_cast<const f4*>(a.z_mu + (size_t)b * kD + chw);
                    const f4 sd = *reinterpret_cast<const f4*>(a.z_std + (size_t)b * kD + chw);
                    f4 e;
                    const float* ep = unc ? a.eps_u : a.eps_c;
                    if (ep) {
                        e = *reinterpret_cast<const f4*>(ep + (size_t)b * kD + chw);
                    } else {
                        float z[4];
                        philox_normal4(a.call, goff + (unsigned long long)b, a.step_id, unc ? 2u : 1u, (unsigned)(chw >> 2), z);
                        e = (f4){z[0], z[1], z[2], z[3]};
                    }
                    v = mu + e * sd;
                } else {                                            // BEAT emotion token (scripts_beat/model/RAG.py:125-126)
                    v = *reinterpret_cast<const f4*>(a.emo_tok + (size_t)b * kD + chw);
                }
            }
            X[t] = v;
        }
        // x_t of this sample -> LDS [S][KXP] (zero for prefix tokens and pad columns); loads first, then the writes, in blocks
        constexpr int NIT = (S * KXP + kCoopThreads - 1) / kCoopThreads;
        constexpr int CH = 14;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += CH) {
            float xv[CH];
#pragma unroll
            for (int itl = 0; itl < CH; ++itl) {
                if (it0 + itl >= NIT) break;
                const int idx = min(tid + kCoopThreads * (it0 + itl), S * KXP - 1);
                const int r = idx / KXP, k = idx - r * KXP;
                const bool live = r >= NPRE && k < JF;
                xv[itl] = a.x_in[(size_t)b * kT * JF + (live ? (r - NPRE) * JF + k : 0)];
                if (!live) xv[itl] = 0.f;
            }
#pragma unroll
            for (int itl = 0; itl < CH; ++itl) {
                if (it0 + itl >= NIT) break;
                const int idx = tid + kCoopThreads * (it0 + itl);
                if (idx < S * KXP) {
                    const int r = idx / KXP;
                    U[r * kUStride + (idx - r * KXP)] = xv[itl];
                }
            }
        }
        __syncthreads();
        fresh();
        f4 acc[NT1];
#pragma unroll
        for (int t = 0; t < NT1; ++t) acc[t] = X[t];
        // winx_img[8][2][KXQ][2][64][4] (ls_api.cpp build_fused_images): 16-channel block 4c + w = (wave c, pass w >> 1, c2 = w & 1)
        const wrsrc_t wrs = wrsrc(a.W->winx_img);
        const int wsb = ((c * 2 + (w >> 1)) * KXQ * 2 + (w & 1)) * 1024;
        f4 An = wload4(wrs, lane * 16, wsb);
#pragma unroll 2
        for (int q = 0; q < KXQ; ++q) {
            const f4 A = An;
            const int qn = q + 1 < KXQ ? q + 1 : KXQ - 1;
            An = wload4(wrs, lane * 16, wsb + qn * 2048);
            f4 Bv[NT1];
#pragma unroll
            for (int t = 0; t < NT1; ++t) Bv[t] = *reinterpret_cast<const f4*>(&U[rowc_of(t) * kUStride + 16 * q + 4 * g]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT1; ++t) acc[t] = MFMA(A[j], Bv[t][j], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < NT1; ++t) X[t] = valid_of(t) ? acc[t] : (f4){0.f, 0.f, 0.f, 0.f};   // pad rows stay zero
    }

    // LN_spatial statistics (mlp_module.py:29-33) of every row over all 512 channels, across the 8 slice workgroups:
    // lane: two passes over its 4 channels; Chan's parallel-variance merge over the 4 lane groups (VALU swaps), the 4 waves (LDS) and
    // the 8 slices (granules through L2 / memory).  `payload`: the caller has issued this workgroup's sc1 payload stores; they are
    // drained before the granules -- which double as the payload's ready flags -- are published.
    float mean[NT1], rstd[NT1];
    auto ln_sync = [&](int area, unsigned tag, bool payload) {
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const f4 v = X[t];
            float m = ((v[0] + v[1]) + (v[2] + v[3])) * 0.25f;
            const f4 d4 = v - (f4){m, m, m, m};
            float m2 = (d4[0] * d4[0] + d4[1] * d4[1]) + (d4[2] * d4[2] + d4[3] * d4[3]);
            {
                float ma, mb, qa, qb;
                xor16_pair(m, ma, mb);
                xor16_pair(m2, qa, qb);
                const float d = mb - ma;
                m2 = (qa + qb) + d * d * 2.0f;
                m = 0.5f * (ma + mb);
            }
            {
                float ma, mb, qa, qb;
                xor32_pair(m, ma, mb);
                xor32_pair(m2, qa, qb);
                const float d = mb - ma;
                m2 = (qa + qb) + d * d * 4.0f;
                m = 0.5f * (ma + mb);
            }
            if (g == 0) pst[w * 48 + 16 * t + s16] = (f2){m, m2};
        }
        if (payload) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores
        __syncthreads();
        unsigned long long* ga = gran + (size_t)area * kCoopRows * kCoopSlices * 2;
        if (tid < S) {                                                          // row tid: merge the 4 waves, publish the slice's partial
            f2 pw[kCoopWaves];
            float ms = 0.f, qs = 0.f;
#pragma unroll
            for (int ww = 0; ww < kCoopWaves; ++ww) { pw[ww] = pst[ww * 48 + tid]; ms += pw[ww].x; qs += pw[ww].y; }
            const float mt = ms * 0.25f;
            float dd = 0.f;
#pragma unroll
            for (int ww = 0; ww < kCoopWaves; ++ww) { const float d = pw[ww].x - mt; dd = fmaf(d, d, dd); }
            unsigned long long* gp = ga + ((size_t)tid * kCoopSlices + c) * 2;
            gran_store(gp, tag, mt);
            gran_store(gp + 1, tag, qs + 16.0f * dd);
        }
        // gather: thread (row = tid >> 3, slice = tid & 7) + a second row 32 + (tid >> 3) for the first 8 * NREM threads
        const int sl = tid & 7, r0 = tid >> 3, r1 = 32 + (tid >> 3);
        const bool has1 = tid < 8 * NREM;
        const unsigned long long* g0 = ga + ((size_t)r0 * kCoopSlices + sl) * 2;
        const unsigned long long* g1p = ga + ((size_t)(has1 ? r1 : r0) * kCoopSlices + sl) * 2;
        unsigned long long v0, v1, v2, v3;
        for (unsigned spins = 0;; ++spins) {
            v0 = gran_load(g0); v1 = gran_load(g0 + 1);
            v2 = gran_load(g1p); v3 = gran_load(g1p + 1);
            const bool ok = (unsigned)(v0 >> 32) == tag && (unsigned)(v1 >> 32) == tag && (unsigned)(v2 >> 32) == tag && (unsigned)(v3 >> 32) == tag;
            if (__all(ok)) break;
            if (spin_bad || spins > kCoopSpinLimit) { spin_bad = 1; break; }     // after one timeout the launch only drains
            __builtin_amdgcn_s_sleep(1);
        }
        auto merge8 = [&](unsigned long long vm, unsigned long long vq, int row, bool live) {
            const float pm = __uint_as_float((unsigned)vm), pq = __uint_as_float((unsigned)vq);
            float sm = pm;
            sm = dpp_add<0xB1>(sm); sm = dpp_add<0x4E>(sm); sm = dpp_add<0x141>(sm);     // the 8 lanes of one row
            const float mu = sm * 0.125f, d = pm - mu;
            float q = fmaf(64.f * d, d, pq);
            q = dpp_add<0xB1>(q); q = dpp_add<0x4E>(q); q = dpp_add<0x141>(q);
            if (live && sl == 0) stat[row] = (f2){mu, rsqrtf(q * (1.0f / kD) + 1e-5f)};
        };
        merge8(v0, v1, r0, true);
        merge8(v2, v3, r1, has1);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const f2 st = stat[rowc_of(t)];
            mean[t] = st.x;
            rstd[t] = st.y;
        }
    };

    const wrsrc_t xrs = uniform_rsrc(xg);

    // ================= TransMLP: 8 x MLPblock (mlp_module.py:67-91) ================================
    for (int l = 0; l < a.layers; ++l) {
        fresh();
        {   // x = x + emb  (re-added at the input of EVERY block, mlp_module.py:68-69, 88-89)
            const f4 e = *reinterpret_cast<const f4*>(a.temb + (size_t)b * a.temb_stride + chw);
#pragma unroll
            for (int t = 0; t < NT1; ++t)
                if (valid_of(t)) X[t] += e;
        }
        // ---- block1: LN -> token-mixing Conv1d(S,S,1) -> SiLU -> residual -------------------------
        const f4 al1 = wload4(wrsrc(a.W->ln1a), chw * 4, l * kD * 4), be1 = wload4(wrsrc(a.W->ln1b), chw * 4, l * kD * 4);
        ln_sync(0, a.epoch + 2 * l + 1, false);
        fresh();
#pragma unroll
        for (int t = 0; t < NT1; ++t)
            if (valid_of(t)) {
                const float nm = -mean[t] * rstd[t];
                f4 u = __builtin_elementwise_fma(X[t], (f4){rstd[t], rstd[t], rstd[t], rstd[t]}, (f4){nm, nm, nm, nm});
                u = __builtin_elementwise_fma(u, al1, be1);
                *reinterpret_cast<f4*>(&U[(16 * t + s16) * kCoopU1Stride + 16 * w + 4 * g]) = u;
            }
        // token mixing contracts over ROWS: wave w reads back only the 16 channel columns it has just written (LDS operations of one
        // wave execute in order); the barriers of ln_sync ordered these stores after every wave's reads of the previous operand
        __builtin_amdgcn_wave_barrier();
        fresh();
        {
            // out[ch][r] = sum_r' u[r'][ch] * Wt[r][r'] + bt[r] as D[channel][row]: A = u^T from LDS, B = the Conv1d weights in per-lane
            // fragment order: wtok1_img[l][t][m][lane] = Wt[16 t + (lane & 15)][4 m + (lane >> 4)] (zero outside S x S)
            const wrsrc_t wrs = wrsrc(a.W->wtok1_img);
            const int wsb = l * NT1 * MK1 * 256;
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                float Bt[MK1];
#pragma unroll
                for (int m = 0; m < MK1; ++m) Bt[m] = wload1(wrs, lane * 4, wsb + (t * MK1 + m) * 256);
                const float bt = valid_of(t) ? g1(a.W->btok_rows)[l * 80 + 16 * t + s16] : 0.f;
                f4 acc = (f4){bt, bt, bt, bt};
#pragma unroll
                for (int m = 0; m < MK1; ++m) {
                    const int sr = (4 * m + 3 < S) ? 4 * m + g : min(4 * m + g, S - 1);   // clamped rows meet zero weights
                    acc = MFMA(U[sr * kCoopU1Stride + 16 * w + s16], Bt[m], acc);
                }
                if (valid_of(t)) X[t] = silu_acc4(acc, X[t]);
            }
        }
        fresh();
        // ---- block2: LN -> channel-mixing Linear(512,512) -> SiLU -> residual ---------------------
        // publish this slice's raw rows (write-through), then the LayerNorm-2 partials as their ready flags
#pragma unroll
        for (int t = 0; t < NT1; ++t)
            if (valid_of(t)) st_sc1(X[t], xrs, ((16 * t + s16) * kD + chw) * 4);
        ln_sync(1, a.epoch + 2 * l + 2, true);
        fresh();
        {   // all 512 channels of the S rows -> LDS, normalised on the way (LN2's alpha / beta are folded into the channel-mix weights)
            constexpr int NLD = kCoopRows / 2;                       // 18 x (2 rows x 512 floats) per pass of the 256 threads
            f4 xv[NLD];
            const int col = (tid & 127) * 4, rh = tid >> 7;
#pragma unroll
            for (int i = 0; i < NLD; ++i) xv[i] = ld_sc1(xrs, (min(2 * i + rh, S - 1) * kD + col) * 4);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int r = 2 * i + rh;
                if (r < S) {
                    const f2 st = stat[r];
                    const float nm = -st.x * st.y;
                    *reinterpret_cast<f4*>(&U[r * kUStride + col]) = __builtin_elementwise_fma(xv[i], (f4){st.y, st.y, st.y, st.y}, (f4){nm, nm, nm, nm});
                }
            }
        }
        __syncthreads();
        fresh();
        {
            const f4 bc = wload4(wrsrc(a.W->bch), chw * 4, l * kD * 4);
            f4 acc[2];
            acc[0] = bc; acc[1] = bc;
            float racc[NRV];
            f4 racc4 = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < NRV; ++r) racc[r] = 0.f;
            // wch_img[L][8][2][32 q][2][64][4]: 16-channel block 4c + w = (wave c, pass w >> 1, c2 = w & 1)
            const wrsrc_t wrs = wrsrc(a.W->wch_img);
            const int wsb = (((l * 8 + c) * 2 + (w >> 1)) * 32 * 2 + (w & 1)) * 1024;
            typedef const __attribute__((address_space(3))) float* ldsp;
            typedef const __attribute__((address_space(3))) f4* ldsp4;
            ldsp ub0 = (ldsp)(U + s16 * kUStride + 4 * g);
            ldsp ur = (ldsp)(U + (32 + (kRemMfma ? (lane & 3) : 0)) * kUStride + 4 * g);
            asm volatile("" : "+v"(ub0), "+v"(ur));
            constexpr int PF = 4;                                    // weight fragments in flight ahead of their use
            f4 An[PF];
#pragma unroll
            for (int k = 0; k < PF; ++k) An[k] = wload4(wrs, lane * 16, wsb + k * 2048);
#pragma unroll 4
            for (int q = 0; q < 32; ++q) {
                const f4 A = An[0];
#pragma unroll
                for (int k = 0; k + 1 < PF; ++k) An[k] = An[k + 1];
                An[PF - 1] = wload4(wrs, lane * 16, wsb + min(q + PF, 31) * 2048);
                f4 Bv[2], Ur[kRemMfma ? 1 : NRV];
#pragma unroll
                for (int t = 0; t < 2; ++t) Bv[t] = *(ldsp4)(ub0 + 16 * t * kUStride + 16 * q);
#pragma unroll
                for (int r = 0; r < (kRemMfma ? 1 : NRV); ++r) Ur[r] = *(ldsp4)(ur + r * kUStride + 16 * q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = MFMA(A[j], Bv[t][j], acc[t]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (kRemMfma) {
                        racc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(A[j], Ur[0][j], racc4, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int r = 0; r < NRV; ++r) racc[r] = fmaf(A[j], Ur[r][j], racc[r]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            fresh();
            // ragged rows: sum the 4 k subsets, then [channel-lane][row] -> [row-lane][channel-reg] through a per-wave LDS patch
            float* rem = REM + w * (4 * 16);
            if constexpr (kRemMfma) {
                f4 v = racc4;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = xor32_sum(xor16_sum(v[i]));
                if (g == 0) *reinterpret_cast<f4*>(&rem[(lane & 3) * 16 + 4 * (s16 >> 2)]) = v;
            } else {
#pragma unroll
                for (int r = 0; r < NRV; ++r) {
                    const float v = xor32_sum(xor16_sum(racc[r]));
                    if (g == 0) rem[r * 16 + s16] = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t) X[t] = silu_acc4(acc[t], X[t]);
            if (s16 < NREM) {
                const f4 rv = *reinterpret_cast<const f4*>(&rem[s16 * 16 + 4 * g]);
                X[2] = silu_acc4(rv + bc, X[2]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ================= OutputProcess.poseFinal (RAG.py:205-211) + CFG + sampler update =============
    // Every slice publishes its final rows into the second exchange buffer (the first may still be read by a slower slice's
    // layer-(L-1) gather) and raises a flag; the (out block, row tile) units of the sample are then spread over its workgroups and
    // waves, unit 4 j + w to wave w of workgroup j = p * 8 + c, and each unit owner computes BOTH passes of its unit.
    fresh();
    const int j16 = p * kCoopSlices + c;
    const unsigned tagF = a.epoch + 2 * a.layers + 1;
    {
        const wrsrc_t ors = uniform_rsrc(a.cx2 + (size_t)pg * kCoopRows * kD);
#pragma unroll
        for (int t = 0; t < NT1; ++t)
            if (valid_of(t)) st_sc1(X[t], ors, ((16 * t + s16) * kD + chw) * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                             // also: every wave is done with the last channel-mix operand
        unsigned long long* fl = a.cflag + (size_t)bl * 16;
        if (tid == 0) gran_store(fl + j16, tagF, 0.f);
        if (4 * j16 >= NU) {                                         // no unit for this workgroup
            if (spin_bad && lane == 0) atomicOr(a.cerr, 1u);
            return;
        }
        if (tid < 64) {
            const int k = min(tid, np * kCoopSlices - 1);
            for (unsigned spins = 0;; ++spins) {
                const bool ok = (unsigned)(gran_load(fl + k) >> 32) == tagF;
                if (__all(ok)) break;
                if (spin_bad || spins > kCoopSpinLimit) { spin_bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    const int nwv = kCoopWaves * np * kCoopSlices;                   // waves of this sample
    constexpr int MAXR = (NU + 4 * kCoopSlices - 1) / (4 * kCoopSlices);      // unit rounds when a single pass runs (np = 1)
    f4 oacc[2][MAXR];
#pragma unroll
    for (int rd = 0; rd < MAXR; ++rd) { oacc[0][rd] = (f4){0.f, 0.f, 0.f, 0.f}; oacc[1][rd] = oacc[0][rd]; }
    for (int pp = 0; pp < np; ++pp) {
        const wrsrc_t ors = uniform_rsrc(a.cx2 + (size_t)(bl * np + pp) * kCoopRows * kD);
        {
            constexpr int NLD = kCoopRows / 2;
            f4 xv[NLD];
            const int col = (tid & 127) * 4, rh = tid >> 7;
#pragma unroll
            for (int i = 0; i < NLD; ++i) xv[i] = ld_sc1(ors, (min(2 * i + rh, S - 1) * kD + col) * 4);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int r = 2 * i + rh;
                if (r < S) *reinterpret_cast<f4*>(&U[r * kUStride + col]) = xv[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int rd = 0; rd < MAXR; ++rd) {
            const int u = 4 * j16 + w + rd * nwv;                    // wave-uniform
            if (u < NU) {
                const int ob = u / NT1, tt = u - ob * NT1;
                const int rc = min(16 * tt + s16, S - 1);
                const wrsrc_t wrs = wrsrc(a.W->wout_img);            // [NOB][32 q][64][4], k in natural order
                const int wsb = ob * 32 * 1024;
                const float* up = &U[rc * kUStride + 4 * g];
                f4 a0 = (f4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
                constexpr int QB = 8;
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
                if (pp == 0) oacc[0][rd] = a0 + a1; else oacc[1][rd] = a0 + a1;
            }
        }
        __syncthreads();                                             // the operand buffer is restaged for the other pass
    }
    if (spin_bad && lane == 0) atomicOr(a.cerr, 1u);

    // ====== CFG lerp (cfg_sampler.py:31) + posterior / DDIM update (gaussian_diffusion.py:260-282, 507-558, 745-798): lane (s16, g)
    //        holds out columns 16 ob + 4 g .. + 3 of row 16 tt + s16, both passes ======================
#pragma unroll
    for (int rd = 0; rd < MAXR; ++rd) {
        const int u = 4 * j16 + w + rd * nwv;
        if (u >= NU) break;
        const int ob = u / NT1, tt = u - ob * NT1;
        const int r = 16 * tt + s16;
        const int f = r - NPRE;
        if (r >= S || f < 0) continue;
        const float sc = (np == 2 && a.scale) ? a.scale[b] : 1.0f;
        const unsigned long long gidx = (a.call ? a.call->sample_offset : 0ull) + (unsigned long long)b;
        const size_t base = (size_t)b * kT * JF;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int cc = 16 * ob + 4 * g + jj;
            if (cc >= JF) continue;
            const int idx = f * JF + cc;
            const float bo = g1(a.W->bout)[cc];
            const float oc = oacc[0][rd][jj] + bo;
            float x0;
            if (np == 2) {
                const float ou = oacc[1][rd][jj] + bo;
                if (a.fwd_c) a.fwd_c[base + idx] = oc;
                if (a.fwd_u) a.fwd_u[base + idx] = ou;
                x0 = ou + sc * (oc - ou);
            } else {
                x0 = oc;                                             // scale == 1: the CFG combination is the cond output
            }
            if (a.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            if (a.x0_out) a.x0_out[base + idx] = x0;
            if (a.sampler != kNone) {
                const float xt = a.x_in[base + idx];
                float nz = 0.f;
                if (a.t_nonzero) {
                    if (a.noise) {
                        const size_t bn = a.const_noise ? 0 : (size_t)b;
                        nz = a.noise[(bn * JF + cc) * kT + f];
                    } else {
                        nz = philox_normal(a.call, gidx, a.step_id, 3u, (unsigned)(cc * kT + f));
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
}

}  // namespace ls
