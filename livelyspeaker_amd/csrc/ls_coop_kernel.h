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
//   * workgroup = (sample b, CFG pass p, channel slice c): 8 waves, the S = 35 | 36 rows of ONE pass x 64 of the 512 channels.
//     Wave (w, h), w = 0..3, h = 0..1: channels [64c + 16w, +16), in the MFMA C/D layout exactly as in k_step; for the residual
//     stream, LayerNorm, token mixing and the epilogues it owns the rows of half h (h = 0: rows 0..15, h = 1: rows 16..S-1 = one
//     full tile + the ragged 3 | 4 rows); for the channel-mixing MFMAs it takes k blocks 2h, 2h+1 of every slice for ALL rows (each
//     weight fragment is loaded by one wave and meets two row tiles), and the two halves swap partial sums through LDS.
//     blockIdx = (b * npass + p) * 8 + c, so slice c of every sample runs on XCD c (observed round-robin placement): each XCD's L2
//     holds one eighth of the weights (1 MB).  Placement is a speed matter only -- every hand-off below is placement-independent.
//   * what crosses workgroups (the 8 slices of one (sample, pass)), per layer:
//       SYNC1  LayerNorm-1: (mean, M2) of each row over the slice's 64 channels                      288 B per workgroup
//       SYNC2  LayerNorm-2: the same, plus the slice's rows x[S][64] (channel mixing contracts over all 512 channels)  9 KB
//     and once per step partial poseFinal outputs (poseFinal contracts over all 512 channels; the CFG combination needs both passes).
//     Token mixing contracts over ROWS and stays inside the workgroup.
//   * SYNC2 never blocks: LayerNorm 2 is folded AROUND the channel-mixing product (as ls_long.hip does),
//         LN2(x) W'^T + b' = rstd2 * ((x - mu1) W'^T - (mu2 - mu1) wsum) + b',      wsum[n] = sum_k W'[n][k],
//     with the rows centred on the LayerNorm-1 mean every slice already holds (no cancellation: mu2 - mu1 is small), so the MFMAs
//     need no statistic: a workgroup multiplies its OWN 64 channels of k while the other seven slices' rows -- requested all at once
//     when their ready flags are up -- are pulled global -> LDS by LDS-DMA (no VGPR round trip), and applies (mu2, rstd2), merged
//     while the pulls land, in the epilogue.  The rows travel and sit in LDS as [slice][k block of 16][row][16]:
//     every MFMA B-operand read and every DMA chunk is one contiguous, conflict-free 1 KiB.
//   * hand-off protocol (cdna_hip_programming.md section 6, Guideline 16, forms R1 / R2): payload = 16-byte write-through (sc1)
//     stores, every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, then the row statistics are published as 8-byte
//     {tag, value} granules with relaxed agent-scope atomic stores -- the granules ARE the flags.  Consumers poll the granules with
//     relaxed agent-scope loads (sc1: L1 bypassed) and read the payload with sc1 loads.  Tags are unique per launch and sync point
//     (CallParams::tag_base of the call + StepArgs::epoch of the launch + index); the granule words are also zeroed ahead of every call.  Every spin is bounded: on a
//     timeout the workgroup records it in StepArgs::cerr and carries on (the host then fails the call).
//   * single-buffered payload is safe: a slice rewrites its rows of layer l+1 only after SYNC1(l+1), which every consumer reaches
//     after its reads of layer l; the statistics alternate between two granule areas for the same reason.
//   * poseFinal: every slice contracts its own 64 channels for all outputs (operand = its rows, staged in LDS), publishes the
//     partial [S][J*F]; the (frame, 4 outputs) quads of the sample are then dealt out to its 16 workgroups, which sum the 8 partials
//     of each pass in a fixed order, combine the passes (CFG) and apply the sampler update.
#pragma once
#include "ls_step_common.h"
#include "ls_lanes.h"

namespace ls {

constexpr int kCoopThreads = 512;
constexpr int kCoopWaves = 8;
constexpr int kCoopRows = 36;              // rows of one pass in the exchange buffers (S <= 36)
constexpr unsigned kCoopSpinLimit = 1u << 18;       // polls (~1-2 us each) before a hand-off wait gives up: waits are < 1 ms when the slices are resident
// LDS: psum [4][48] f2 | stat [48] f2 | U [32 k blocks of 16 channels][36 rows][16] (all 512 channels of one pass, whatever the slicing)
constexpr int kCoopLdsFloats = 2 * 4 * 48 + 2 * 48 + 32 * kCoopRows * 16;
// The slicing is a template parameter (round 6): NCB = 16-channel blocks per WAVE = 1 | 2 | 4, i.e. slices of 64 | 128 | 256 channels,
// 8 | 4 | 2 slice workgroups per (sample, pass).  NCB = 1 is the round-4 kernel (two workgroups per CU: 32 clips fill the chip); with
// NCB = 2 / 4 a workgroup holds twice / four times the registers (one workgroup per CU: 32 / 64 clips fill the chip), exchanges
// 3/7 / 1/7 of the rows per group and layer, and -- with two slices -- multiplies its own half of k for ~18 k clocks while the other
// half arrives, so the row hand-off leaves the chain altogether.
template <int NCB> struct CoopGeom {
    static_assert(NCB == 1 || NCB == 2 || NCB == 4, "16-channel blocks per wave");
    static constexpr int NS = 8 / NCB;                     // slice workgroups of one (sample, pass)
    static constexpr int KB = 4 * NCB;                     // 16-channel k blocks of a slice
    static constexpr int CPS = 64 * NCB;                   // channels per slice
    static constexpr int SLICEF = KB * kCoopRows * 16;     // one slice's rows in exchange order [KB k blocks][36 rows][16], floats
    static constexpr int U1S = CPS + 16;                   // LDS row stride of the token-mix operand [S][CPS]: = 16 mod 32, conflict-free column reads
    static constexpr int HB = KB / 2;                      // k blocks of one slice a wave half multiplies
    static constexpr int PF = NCB == 1 ? 4 : 2;            // k blocks whose weight fragments are in flight (NCB fragments each)
    static constexpr int MINW = NCB == 1 ? 4 : 2;          // waves per SIMD the register budget allows for (two | one workgroup per CU)
};

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long* gu64p;

__device__ __forceinline__ f4 ld_sc1(wrsrc_t r, int voff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 16));     // aux 16 = sc1: served from L2 / memory, never this CU's L1
}
__device__ __forceinline__ float ld_sc1_1(wrsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 16));
}
__device__ __forceinline__ void st_sc1(f4 v, wrsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, voff, 0, 16);       // write-through
}
__device__ __forceinline__ unsigned long long gran_load(const unsigned long long* p) {
    return __hip_atomic_load((gu64p)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef __attribute__((address_space(3))) void* coop_lds_vp;
// 1 KiB global -> LDS, lane i's 16 bytes to lds + 16 i (LDS-DMA, no VGPR round trip); sc1: never served from this CU's L1
__device__ __forceinline__ void dma_sc1(wrsrc_t r, float* lds, int lane_bytes, int wave_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (coop_lds_vp)lds, 16, lane_bytes, wave_bytes, 0, 16);
}
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for every LDS-DMA pull and
// weight prefetch in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void gran_store(unsigned long long* p, unsigned tag, float v) {
    __hip_atomic_store((gu64p)p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int S, int NPRE, int JF, int NCB>
__global__ __launch_bounds__(kCoopThreads, CoopGeom<NCB>::MINW) void k_coop(const StepArgs a) {
    using G = CoopGeom<NCB>;
    constexpr int NS = G::NS, KB = G::KB, CPS = G::CPS, SLICEF = G::SLICEF, U1S = G::U1S, HB = G::HB, PF = G::PF;
    constexpr int KXQ = (JF + 15) / 16;
    constexpr int KXP = KXQ * 16;
    constexpr int XSTR = KXP + 4;               // LDS row stride of the x_t staging
    constexpr int NOB = (JF + 15) / 16;
    constexpr int NOBP = NOB * 16;              // padded output columns
    constexpr int NT1 = 3;                      // 16-row tiles of one pass
    constexpr int NREM = S - 32;                // rows of the ragged third tile: 3 (TED) | 4 (BEAT)
    constexpr bool kRemMfma = (NREM % 4 == 0);  // BEAT: one v_mfma_f32_4x4x1 row group; TED: scalar FMAs (see k_step)
    constexpr int NRV = kRemMfma ? 1 : NREM;
    constexpr int MK1 = (S + 3) / 4;            // k steps of the token-mix GEMM of one pass
    constexpr int MQ1 = (MK1 + 3) / 4;          // ... in groups of four (one 16-byte weight fragment per lane)
    constexpr int NQUAD = kT * (NOBP / 4);      // (frame, 4 output columns) quads of one sample
    static_assert(S > 32 && S <= kCoopRows, "one pass = two full row tiles + a ragged one");
    static_assert(NREM >= 1 && NREM <= 4, "ragged tile");
    static_assert(S * XSTR <= NS * SLICEF, "x_t staging fits the operand buffer");
    static_assert(S * U1S <= NS * SLICEF && NCB * (kCoopWaves * 256 + kCoopWaves * 64) <= NS * SLICEF, "overlays fit the operand buffer");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    f2* pst = reinterpret_cast<f2*>(smem);                         // [4 waves of a half][48 rows] (mean, M2) over the wave's 16 NCB channels
    f2* stat = pst + 4 * 48;                                       // [48 rows] (mean, rstd) over all 512 channels
    float* U = smem + 2 * 4 * 48 + 2 * 48;                         // [NS slices][KB k blocks][36 rows][16]; overlays: x_t staging, token-mix operand, partial sums

    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3, h = wv >> 2;                              // channel blocks NCB w .. NCB w + NCB - 1 of the slice, row half
    const int bid = blockIdx.x;
    const int np = a.npass;
    // blockIdx -> (group, slice), for speed only (observed round-robin placement: block b runs on XCD b % 8; every hand-off is
    // placement-independent).  xmap 0: slice = bid % NS, i.e. slice c of every group on XCDs c, c + NS, ... (each XCD's L2 holds 1 / NS of
    // the weights; every row hand-off crosses XCDs).  xmap 1: the NS slices of a group on ONE XCD (cheaper hand-offs, every L2 sees all the
    // weights).  xmap 2 (8 slices): slices 0-3 of a group on one XCD, 4-7 on its neighbour.  Measured (profiles/r06_split_variants.md):
    // with one workgroup per CU xmap 1 wins by 13-16 %, with two per CU xmap 2 by 0-4 % -- the host picks by grid size.
    const bool xm = a.xmap == 1, xm2 = NCB == 1 && a.xmap == 2;
    const int c = xm ? ((bid >> 3) & (NS - 1)) : xm2 ? ((bid & 1) * 4 + ((bid >> 3) & 3)) : (bid & (NS - 1));            // channel slice
    const int pg = xm ? ((bid >> 3) / NS * 8 + (bid & 7)) : xm2 ? ((bid >> 5) * 4 + ((bid & 7) >> 1)) : (bid / NS);      // launch-local (sample, pass) group
    if (pg >= a.ngroups) return;                                    // xmap 1 rounds the grid up to whole sets of 8 groups
    const int p = np == 2 ? (pg & 1) : 0;
    const int bl = np == 2 ? (pg >> 1) : pg;                        // launch-local sample
    const int b = a.b0 + bl;                                        // sample of the prepared batch
    const bool unc = p == 1;
    int s16 = lane & 15;
    int g = lane >> 4;
    // 16-channel block cb of this wave, numbered over all 512 channels (the weight images' numbering: block = 4 * (k_step wave) + 2 * pass + c2)
    auto gblk = [&](int cb) { return KB * c + NCB * w + cb; };
    auto chw = [&](int cb) { return 16 * gblk(cb) + 4 * g; };       // + j = this lane's channels of block cb
    auto fresh = [&]() {
        asm volatile("" : "+v"(lane));
        s16 = lane & 15;
        g = lane >> 4;
    };
    // this lane's rows: r0 = 16 h + s16 (always a real row), r1 = 32 + s16 (half 1 only, the ragged rows)
    auto row0 = [&]() { return 16 * h + s16; };
    auto live1 = [&]() { return h == 1 && s16 < NREM; };
    auto row1c = [&]() { return min(32 + s16, S - 1); };

    float* xg = a.cx + (size_t)pg * 32 * kCoopRows * 16;                // centred rows of this (sample, pass), exchange order
    unsigned long long* gran = a.cgran + (size_t)pg * 2 * kCoopRows * NS * 2;   // [2 areas][36 rows][NS slices][2] granules
    unsigned spin_bad = 0;
    // phase stamps, -DLS_DEBUG builds only (tools/coop_profile.py): lane 0 of every wave of one workgroup records s_memtime
    auto stamp = [&](int idx) {
#ifdef LS_DEBUG
        if (a.prof && bid == a.prof_wg && lane == 0 && idx < kProfPoints) a.prof[wv * kProfPoints + idx] = __builtin_amdgcn_s_memtime();
#else
        (void)idx;
#endif
    };
    stamp(0);
    // Wave priority by phase: the hand-off / LayerNorm / epilogue phases sit on the latency chain of the (sample, pass); the long product
    // of the CU's other workgroup is the filler (measured: -3.5 % at 32 TED clips, nothing at 4).
    __builtin_amdgcn_s_setprio(3);

    f4 X0[NCB], X1[NCB];                                            // residual stream: rows r0 / r1 of this lane's 4 channels of each block

    // ================= embedding: InputProcess + input_mapping (RAG.py:110-114, 184-192) ==========
    {
        const unsigned long long goff = a.call ? a.call->sample_offset : 0ull;
        auto base_row = [&](int tk, int ch) -> f4 {
            if (tk >= NPRE) return *reinterpret_cast<const f4*>((unc ? a.static_u : a.static_c) + ((size_t)b * kT + (tk - NPRE)) * kD + ch);
            if (tk == 0) {                                          // style token: reparameterize(mu, logvar)  (RAG.py:10-13, 116-120)
                const f4 mu = *reinterpret_cast<const f4*>(a.z_mu + (size_t)b * kD + ch);
                const f4 sd = *reinterpret_cast<const f4*>(a.z_std + (size_t)b * kD + ch);
                f4 e;
                const float* ep = unc ? a.eps_u : a.eps_c;
                if (ep) {
                    e = *reinterpret_cast<const f4*>(ep + (size_t)b * kD + ch);
                } else {
                    float z[4];
                    philox_normal4(a.call, goff + (unsigned long long)b, a.step_id, unc ? 2u : 1u, (unsigned)(ch >> 2), z);
                    e = (f4){z[0], z[1], z[2], z[3]};
                }
                return mu + e * sd;
            }
            return *reinterpret_cast<const f4*>(a.emo_tok + (size_t)b * kD + ch);       // BEAT emotion token (scripts_beat/model/RAG.py:125-126)
        };
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            X0[cb] = base_row(row0(), chw(cb));
            X1[cb] = live1() ? base_row(row1c(), chw(cb)) : (f4){0.f, 0.f, 0.f, 0.f};
        }
        // x_t of this sample -> LDS [S][KXP] (zero for prefix tokens and pad columns); loads first, then the writes, in blocks
        constexpr int NIT = (S * KXP + kCoopThreads - 1) / kCoopThreads;
        constexpr int CH = 11;
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
                    U[r * XSTR + (idx - r * KXP)] = xv[itl];
                }
            }
        }
        lds_barrier();
        fresh();
        f4 acc0[NCB], acc1[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { acc0[cb] = X0[cb]; acc1[cb] = X1[cb]; }
        // winx_img[8][2][KXQ][2][64][4] (ls_api.cpp build_fused_images): 16-channel block gb = (wave gb >> 2, pass (gb >> 1) & 1, c2 = gb & 1)
        const wrsrc_t wrs = wrsrc(a.W->winx_img);
        auto wsb = [&](int cb) { const int gb = gblk(cb); return (((gb >> 2) * 2 + ((gb >> 1) & 1)) * KXQ * 2 + (gb & 1)) * 1024; };
        constexpr int EPF0 = NCB == 1 ? 4 : 2;
        constexpr int EPF = KXQ < EPF0 ? KXQ : EPF0;                 // k steps whose weight fragments are in flight
        f4 An[EPF][NCB];
#pragma unroll
        for (int k = 0; k < EPF; ++k)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) An[k][cb] = wload4(wrs, lane * 16, wsb(cb) + k * 2048);
        const float* u0 = &U[row0() * XSTR + 4 * g];
        const float* u1 = &U[row1c() * XSTR + 4 * g];
#pragma unroll EPF
        for (int q = 0; q < KXQ; ++q) {
            f4 A[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                A[cb] = An[0][cb];
#pragma unroll
                for (int k = 0; k + 1 < EPF; ++k) An[k][cb] = An[k + 1][cb];
                An[EPF - 1][cb] = wload4(wrs, lane * 16, wsb(cb) + min(q + EPF, KXQ - 1) * 2048);
            }
            const f4 B0 = *reinterpret_cast<const f4*>(u0 + 16 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc0[cb] = MFMA(A[cb][j], B0[j], acc0[cb]);
            if (h) {
                const f4 B1 = *reinterpret_cast<const f4*>(u1 + 16 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc1[cb] = MFMA(A[cb][j], B1[j], acc1[cb]);
            }
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            X0[cb] = acc0[cb];
            X1[cb] = live1() ? acc1[cb] : (f4){0.f, 0.f, 0.f, 0.f};  // pad rows stay zero
        }
    }

    // LN_spatial statistics (mlp_module.py:29-33) of every row over all 512 channels, across the NS slice workgroups:
    // lane: two passes over its 4 NCB channels; Chan's parallel-variance merge over the lane's blocks, the 4 lane groups (VALU swaps), the
    // 4 waves of a half (LDS) and the NS slices (granules through L2 / memory).
    // ln_publish: this slice's (mean, M2) of every row -> granule area `area`.  `payload`: the caller has issued this workgroup's
    // write-through payload stores; they are drained before the granules -- which double as the payload's ready flags -- go out.
    auto lane_part = [&](const f4 (&X)[NCB], float& m, float& m2) {
        float mm[NCB], qq[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const f4 v = X[cb];
            mm[cb] = ((v[0] + v[1]) + (v[2] + v[3])) * 0.25f;
            const f4 d4 = v - (f4){mm[cb], mm[cb], mm[cb], mm[cb]};
            qq[cb] = (d4[0] * d4[0] + d4[1] * d4[1]) + (d4[2] * d4[2] + d4[3] * d4[3]);
        }
        // equal counts n merge as: mean = (ma + mb) / 2, M2 = qa + qb + (mb - ma)^2 n / 2
        if constexpr (NCB >= 2) {
#pragma unroll
            for (int cb = 0; cb < NCB; cb += 2) {
                const float d = mm[cb + 1] - mm[cb];
                qq[cb] = (qq[cb] + qq[cb + 1]) + d * d * 2.0f;
                mm[cb] = 0.5f * (mm[cb] + mm[cb + 1]);
            }
        }
        if constexpr (NCB >= 4) {
            const float d = mm[2] - mm[0];
            qq[0] = (qq[0] + qq[2]) + d * d * 4.0f;
            mm[0] = 0.5f * (mm[0] + mm[2]);
        }
        m = mm[0]; m2 = qq[0];
        {
            float ma, mb, qa, qb;
            xor16_pair(m, ma, mb);
            xor16_pair(m2, qa, qb);
            const float d = mb - ma;
            m2 = (qa + qb) + d * d * (2.0f * NCB);
            m = 0.5f * (ma + mb);
        }
        {
            float ma, mb, qa, qb;
            xor32_pair(m, ma, mb);
            xor32_pair(m2, qa, qb);
            const float d = mb - ma;
            m2 = (qa + qb) + d * d * (4.0f * NCB);
            m = 0.5f * (ma + mb);
        }
    };
    auto ln_publish = [&](int area, unsigned tag, bool payload) {
        float m, m2;
        if (h) {                                                     // wave-uniform; the two rows' chains in one block, so that they interleave
            float n, n2;
            lane_part(X0, m, m2);
            lane_part(X1, n, n2);
            if (g == 0) {
                pst[w * 48 + row0()] = (f2){m, m2};
                pst[w * 48 + 32 + s16] = (f2){n, n2};
            }
        } else {
            lane_part(X0, m, m2);
            if (g == 0) pst[w * 48 + row0()] = (f2){m, m2};
        }
        if (payload) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores
        lds_barrier();
        if (tid < S) {                                                          // row tid: merge the 4 waves' blocks, publish the slice's partial
            f2 pw[4];
            float ms = 0.f, qs = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) { pw[ww] = pst[ww * 48 + tid]; ms += pw[ww].x; qs += pw[ww].y; }
            const float mt = ms * 0.25f;
            float dd = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) { const float d = pw[ww].x - mt; dd = fmaf(d, d, dd); }
            unsigned long long* gp = gran + (size_t)area * kCoopRows * NS * 2 + ((size_t)tid * NS + c) * 2;
            gran_store(gp, tag, mt);
            gran_store(gp + 1, tag, qs + (16.0f * NCB) * dd);
        }
    };
    // ln_gather: wait for all NS slices' partials of every row, merge -> stat[row] = (mean, rstd) and this lane's rows' values
    float mean0, rstd0, mean1, rstd1;
    auto ln_gather = [&](int area, unsigned tag, int stamp_at, bool drain = false) {
        const unsigned long long* ga = gran + (size_t)area * kCoopRows * NS * 2;
        // thread (row = tid / NS, slice = tid % NS); threads beyond the S rows re-read the last row
        const int sl = tid & (NS - 1), r = min(tid / NS, S - 1);
        const bool live = (tid / NS) < S;
        const unsigned long long* g0 = ga + ((size_t)r * NS + sl) * 2;
        unsigned long long v0, v1;
        stamp(stamp_at);
        if (wv * (64 / NS) < S) {                                    // waves whose rows exist poll (wave-uniform)
            for (unsigned spins = 0;; ++spins) {
                v0 = gran_load(g0); v1 = gran_load(g0 + 1);
                const bool ok = (unsigned)(v0 >> 32) == tag && (unsigned)(v1 >> 32) == tag;
                if (__all(ok)) break;
                if (spin_bad || spins > kCoopSpinLimit) { spin_bad = 1; break; }     // after one timeout the launch only drains
                __builtin_amdgcn_s_sleep(1);
            }
            const float pm = __uint_as_float((unsigned)v0), pq = __uint_as_float((unsigned)v1);
            float sm = pm;
            sm = dpp_add<0xB1>(sm);                                  // the NS lanes of one row
            if constexpr (NS >= 4) sm = dpp_add<0x4E>(sm);
            if constexpr (NS >= 8) sm = dpp_add<0x141>(sm);
            const float mu = sm * (1.0f / NS), d = pm - mu;
            float q = fmaf((float)CPS * d, d, pq);
            q = dpp_add<0xB1>(q);
            if constexpr (NS >= 4) q = dpp_add<0x4E>(q);
            if constexpr (NS >= 8) q = dpp_add<0x141>(q);
            if (live && sl == 0) stat[r] = (f2){mu, rsqrtf(q * (1.0f / kD) + 1e-5f)};
        }
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the caller's LDS-DMA pulls have landed before anyone passes the barrier
        lds_barrier();
        const f2 s0 = stat[row0()], s1 = stat[row1c()];
        mean0 = s0.x; rstd0 = s0.y; mean1 = s1.x; rstd1 = s1.y;
    };

    // tags of this launch: the host's per-launch epoch + the call's base from device memory (the launch arguments are frozen in a replayed
    // graph; the base is what makes a replay's tags differ from the previous call's)
    const unsigned ep = a.epoch + (a.call ? a.call->tag_base : 0u);
    const wrsrc_t xrs = uniform_rsrc(xg);
    // The weight table's pointers once, into SGPR descriptors: the hand-off asm statements clobber "memory", so a dereference of a.W
    // inside the layer loop is re-loaded after each of them -- a dependent global load + s_waitcnt vmcnt(0) at the head of every phase.
    const wrsrc_t rs_ln1a = wrsrc(a.W->ln1a), rs_ln1b = wrsrc(a.W->ln1b), rs_wtok1 = wrsrc(a.W->wtok1_img), rs_bch = wrsrc(a.W->bch),
                  rs_wsum = wrsrc(a.W->wsum), rs_wch = wrsrc(a.W->wch_img), rs_wout = wrsrc(a.W->wout_reg_img);
    const gfp p_btok = g1(a.W->btok_rows), p_bout = g1(a.W->bout);
    f4 temb4[NCB];                                                   // the same row at every block
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) temb4[cb] = *reinterpret_cast<const f4*>(a.temb + (size_t)b * a.temb_stride + chw(cb));
    stamp(1);

    // ================= TransMLP: 8 x MLPblock (mlp_module.py:67-91) ================================
    for (int l = 0; l < a.layers; ++l) {
        fresh();
        // x = x + emb  (re-added at the input of EVERY block, mlp_module.py:68-69, 88-89)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            X0[cb] += temb4[cb];
            if (live1()) X1[cb] += temb4[cb];
        }
        // ---- block1: LN -> token-mixing Conv1d(S,S,1) -> SiLU -> residual -------------------------
        ln_publish(0, ep + 2 * l + 1, false);
        // LayerNorm affine, token-mix weights and biases of this wave's row tiles (tile h, and the ragged tile 2 for half 1): requested
        // AFTER the partials are out (22 dword loads per wave ahead of them cost the chain ~1 k clocks of issue), in flight during the
        // exchange -- they depend on no activation.  wtok1_img[l][t][mq][lane][j] = Wt[16 t + (lane & 15)][4 (4 mq + j) + (lane >> 4)]
        f4 al1[NCB], be1[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { al1[cb] = wload4(rs_ln1a, chw(cb) * 4, l * kD * 4); be1[cb] = wload4(rs_ln1b, chw(cb) * 4, l * kD * 4); }
        f4 Bt0[MQ1], Bt1[MQ1];
        float btb0, btb1;
        {
            const int wsb = l * NT1 * MQ1 * 1024;
#pragma unroll
            for (int m = 0; m < MQ1; ++m) {
                Bt0[m] = wload4(rs_wtok1, lane * 16, wsb + (h * MQ1 + m) * 1024);
                Bt1[m] = h ? wload4(rs_wtok1, lane * 16, wsb + (2 * MQ1 + m) * 1024) : (f4){0.f, 0.f, 0.f, 0.f};
            }
            btb0 = p_btok[l * 80 + row0()];
            btb1 = p_btok[l * 80 + row1c()];
        }
        ln_gather(0, ep + 2 * l + 1, 2 + 8 * l + 5);
        stamp(2 + 8 * l);
        fresh();
        const float mu1_0 = mean0, mu1_1 = mean1;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            {
                const float nm = -mean0 * rstd0;
                f4 u = __builtin_elementwise_fma(X0[cb], (f4){rstd0, rstd0, rstd0, rstd0}, (f4){nm, nm, nm, nm});
                u = __builtin_elementwise_fma(u, al1[cb], be1[cb]);
                *reinterpret_cast<f4*>(&U[row0() * U1S + 16 * (NCB * w + cb) + 4 * g]) = u;
            }
            if (live1()) {
                const float nm = -mean1 * rstd1;
                f4 u = __builtin_elementwise_fma(X1[cb], (f4){rstd1, rstd1, rstd1, rstd1}, (f4){nm, nm, nm, nm});
                u = __builtin_elementwise_fma(u, al1[cb], be1[cb]);
                *reinterpret_cast<f4*>(&U[(32 + s16) * U1S + 16 * (NCB * w + cb) + 4 * g]) = u;
            }
        }
        lds_barrier();                                             // token mixing contracts over ROWS: both halves' rows of these channels
        fresh();
        {
            // out[ch][r] = sum_r' u[r'][ch] * Wt[r][r'] + bt[r] as D[channel][row]: A = u^T from LDS, B = the Conv1d weights
            f4 acc0[NCB], acc1[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { acc0[cb] = (f4){btb0, btb0, btb0, btb0}; acc1[cb] = (f4){btb1, btb1, btb1, btb1}; }
#pragma unroll
            for (int m = 0; m < MK1; ++m) {
                const int sr = (4 * m + 3 < S) ? 4 * m + g : min(4 * m + g, S - 1);   // clamped rows meet zero weights
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const float av = U[sr * U1S + 16 * (NCB * w + cb) + s16];
                    acc0[cb] = MFMA(av, Bt0[m >> 2][m & 3], acc0[cb]);
                    if (h) acc1[cb] = MFMA(av, Bt1[m >> 2][m & 3], acc1[cb]);
                }
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                X0[cb] = silu_acc4(acc0[cb], X0[cb]);
                if (live1()) X1[cb] = silu_acc4(acc1[cb], X1[cb]);
            }
        }
        stamp(3 + 8 * l);
        fresh();
        // ---- block2: LN -> channel-mixing Linear(512,512) -> SiLU -> residual ---------------------
        // This slice's rows, centred on the LayerNorm-1 mean: (write-through) to the other slices, and into its own region of the operand
        // buffer once every wave is past its token-mix reads of the overlay (the barrier inside ln_publish); k block NCB w + cb of the slice.
        {
            f4 c0[NCB], c1[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                c0[cb] = X0[cb] - (f4){mu1_0, mu1_0, mu1_0, mu1_0};
                c1[cb] = X1[cb] - (f4){mu1_1, mu1_1, mu1_1, mu1_1};
                st_sc1(c0[cb], xrs, ((c * KB + NCB * w + cb) * kCoopRows * 16 + row0() * 16 + 4 * g) * 4);
                if (live1()) st_sc1(c1[cb], xrs, ((c * KB + NCB * w + cb) * kCoopRows * 16 + (32 + s16) * 16 + 4 * g) * 4);
            }
            ln_publish(1, ep + 2 * l + 2, true);                // LayerNorm-2 partials of the RAW rows
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float* own = U + c * SLICEF + (NCB * w + cb) * (kCoopRows * 16);
                *reinterpret_cast<f4*>(&own[row0() * 16 + 4 * g]) = c0[cb];
                if (live1()) *reinterpret_cast<f4*>(&own[(32 + s16) * 16 + 4 * g]) = c1[cb];
            }
        }
        stamp(4 + 8 * l);
        fresh();
        {
            f4 bc[NCB], ws4[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { bc[cb] = wload4(rs_bch, chw(cb) * 4, l * kD * 4); ws4[cb] = wload4(rs_wsum, chw(cb) * 4, l * kD * 4); }
            f4 acc[NCB][2];
            float racc[NCB][NRV];
            f4 racc4[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                acc[cb][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc[cb][1] = acc[cb][0]; racc4[cb] = acc[cb][0];
#pragma unroll
                for (int r = 0; r < NRV; ++r) racc[cb][r] = 0.f;
            }
            // wch_img[L][8][2][32 q][2][64][4]: 16-channel block gb = (wave gb >> 2, pass (gb >> 1) & 1, c2 = gb & 1); k block q = KB s + q' of slice s.
            // Half h multiplies k blocks q' = HB h .. HB h + HB - 1 of every slice, slices in ring order from its own.
            const wrsrc_t wrs = rs_wch;
            auto wsb = [&](int cb) { const int gb = gblk(cb); return (((l * 8 + (gb >> 2)) * 2 + ((gb >> 1) & 1)) * 32 * 2 + (gb & 1)) * 1024; };
            auto qof = [&](int n) { return ((c + n / HB) & (NS - 1)) * KB + HB * h + (n % HB); };     // the n-th of this wave's 16 k blocks
            constexpr int NKB = NS * HB;                             // = 16
            f4 An[PF][NCB];                                          // weight fragments in flight ahead of their use
#pragma unroll
            for (int k = 0; k < PF; ++k)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) An[k][cb] = wload4(wrs, lane * 16, wsb(cb) + qof(k) * 2048);
            const unsigned long long* ga = gran + (size_t)kCoopRows * NS * 2;         // area 1: row 0's mean granule = slice ready
            const unsigned tag2 = ep + 2 * l + 2;
            lds_barrier();                                           // own slice visible to every wave
            // Pull the other slices (9 NCB chunks of 1 KiB each, dealt over the waves) into their LDS regions as soon as their rows are
            // published: lane s polls slice s.  The pulls fly while this slice's own k blocks are multiplied from LDS.
            {
                for (unsigned spins = 0;; ++spins) {
                    const bool ok = (unsigned)(gran_load(ga + (size_t)(lane & (NS - 1)) * 2) >> 32) == tag2;
                    if (__all(ok)) break;
                    if (spin_bad || spins > kCoopSpinLimit) { spin_bad = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                stamp(70 + 2 * l);
                constexpr int NCHUNK = SLICEF / 256;                 // 1 KiB chunks of a slice: 9 NCB
#pragma unroll 1
                for (int i = 1; i < NS; ++i) {
                    const int s = (c + i) & (NS - 1);
#pragma unroll
                    for (int ch0 = 0; ch0 < NCHUNK; ch0 += kCoopWaves) {
                        const int ch = ch0 + wv;
                        if (ch < NCHUNK) dma_sc1(xrs, U + s * SLICEF + ch * 256, lane * 16, (s * SLICEF + ch * 256) * 4);
                    }
                }
            }
            typedef const __attribute__((address_space(3))) f4* ldsp4;
            // PF k blocks per trip, so that fragment k of the trip lives in An[k] and is re-requested in place: a rolled
            // one-block loop rotates An[] through register moves, and a move of the newest fragment waits for it -- vmcnt(0) at the end of
            // every iteration, i.e. no prefetch across iterations at all.
            static_assert(NKB % PF == 0 && (PF % HB == 0 || HB % PF == 0), "the k-block loop is unrolled by PF");
#pragma unroll 1
            for (int trip = 0; trip < NKB / PF; ++trip) {
                // operands of a k block are read while the previous block of the SAME slice is multiplied (never across the point where
                // the other slices' rows are known to have landed)
                f4 Bn[2], Un[kRemMfma ? 1 : NRV];
                auto ldb = [&](int n) {
                    const int s = (c + n / HB) & (NS - 1), kb = HB * h + (n % HB);
                    const float* ub = U + s * SLICEF + (kb * kCoopRows + s16) * 16 + 4 * g;
                    const float* ubr = U + s * SLICEF + (kb * kCoopRows + 32 + (kRemMfma ? (lane & 3) : 0)) * 16 + 4 * g;
#pragma unroll
                    for (int t = 0; t < 2; ++t) Bn[t] = *(ldsp4)(ub + 16 * t * 16);
#pragma unroll
                    for (int r = 0; r < (kRemMfma ? 1 : NRV); ++r) Un[r] = *(ldsp4)(ubr + r * 16);
                };
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int n = PF * trip + u;
                    if (u == 0 && trip == 0) __builtin_amdgcn_s_setprio(0);
                    if ((PF % HB == 0) ? (trip == 0 && u == HB) : (u == 0 && trip == HB / PF)) {
                        // LayerNorm-2 statistics while the pulls land (their granules came up with the ready flags); its barrier waits for
                        // vmcnt(0) first: this wave's chunks have landed in LDS, and so have the other waves'
                        __builtin_amdgcn_s_setprio(3);
                        ln_gather(1, tag2, 2 + 8 * l + 6, true);
                        stamp(71 + 2 * l);
                        __builtin_amdgcn_s_setprio(0);
                    }
                    const bool head = u == 0 || (PF % HB == 0 && u % HB == 0);                   // first block of a slice (or of the trip)
                    const bool more = u + 1 < PF && !(PF % HB == 0 && (u + 1) % HB == 0);       // the next block is of the same slice
                    if (head) ldb(n);
                    f4 Bv[2], Ur[kRemMfma ? 1 : NRV];
#pragma unroll
                    for (int t = 0; t < 2; ++t) Bv[t] = Bn[t];
#pragma unroll
                    for (int r = 0; r < (kRemMfma ? 1 : NRV); ++r) Ur[r] = Un[r];
                    if (more) ldb(n + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                            for (int t = 0; t < 2; ++t) acc[cb][t] = MFMA(An[u][cb][j], Bv[t][j], acc[cb][t]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            if constexpr (kRemMfma) {
                                racc4[cb] = __builtin_amdgcn_mfma_f32_4x4x1f32(An[u][cb][j], Ur[0][j], racc4[cb], 0, 0, 0);
                            } else {
#pragma unroll
                                for (int r = 0; r < NRV; ++r) racc[cb][r] = fmaf(An[u][cb][j], Ur[r][j], racc[cb][r]);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    // re-requested in place AFTER its last use (requested before, the old and the new fragment are both live and the new one
                    // is copied into place at the back edge -- behind a wait for it)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) An[u][cb] = wload4(wrs, lane * 16, wsb(cb) + qof(min(n + PF, NKB - 1)) * 2048);
                }
            }
            stamp(5 + 8 * l);
            __builtin_amdgcn_s_setprio(3);
            fresh();
            // The two halves swap partial sums through LDS (overlaid on the operand buffer, which every wave has finished reading
            // after the barrier): wave (w, h) keeps row tile h and hands tile 1 - h to wave (w, 1 - h); the ragged rows' partials --
            // summed over the 4 k subsets of the lanes first, [channel-lane][row] -> [row-lane][channel-reg] -- all go to half 1.
            lds_barrier();
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                float* xch = U + (cb * kCoopWaves + wv) * 256;                           // [NCB][8 waves][64 lanes][4]
                float* rag = U + NCB * kCoopWaves * 256 + (cb * kCoopWaves + wv) * 64;   // [NCB][8 waves][4 rows][16 channels]
                *reinterpret_cast<f4*>(&xch[lane * 4]) = h ? acc[cb][0] : acc[cb][1];
                if constexpr (kRemMfma) {
                    f4 v = racc4[cb];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = xor32_sum(xor16_sum(v[i]));
                    if (g == 0) *reinterpret_cast<f4*>(&rag[(lane & 3) * 16 + 4 * (s16 >> 2)]) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < NRV; ++r) {
                        const float v = xor32_sum(xor16_sum(racc[cb][r]));
                        if (g == 0) rag[r * 16 + s16] = v;
                    }
                }
            }
            // LayerNorm 2 around the product: v = rstd2 * (acc - (mu2 - mu1) wsum) + b'
            lds_barrier();                                         // the partner's partials are in LDS
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float* rag = U + NCB * kCoopWaves * 256 + (cb * kCoopWaves + wv) * 64;
                {
                    const f4 mine = h ? acc[cb][1] : acc[cb][0];
                    const f4 sum = mine + *reinterpret_cast<const f4*>(&U[(cb * kCoopWaves + (wv ^ 4)) * 256 + lane * 4]);
                    const float dm = mean0 - mu1_0;
                    const f4 v = (sum - (f4){dm, dm, dm, dm} * ws4[cb]) * (f4){rstd0, rstd0, rstd0, rstd0} + bc[cb];
                    X0[cb] = silu_acc4(v, X0[cb]);
                }
                if (live1()) {
                    const f4 rv = *reinterpret_cast<const f4*>(&rag[s16 * 16 + 4 * g]) +
                                  *reinterpret_cast<const f4*>(&U[NCB * kCoopWaves * 256 + (cb * kCoopWaves + (wv ^ 4)) * 64 + s16 * 16 + 4 * g]);
                    const float dm = mean1 - mu1_1;
                    const f4 v = (rv - (f4){dm, dm, dm, dm} * ws4[cb]) * (f4){rstd1, rstd1, rstd1, rstd1} + bc[cb];
                    X1[cb] = silu_acc4(v, X1[cb]);
                }
            }
        }
        stamp(6 + 8 * l);
    }

    // ================= OutputProcess.poseFinal (RAG.py:205-211) + CFG + sampler update =============
    fresh();
    const int j16 = p * NS + c;
    const int n16 = np * NS;                                         // workgroups of this sample
    const unsigned tagF = ep + 2 * a.layers + 1;
    {
        // this slice's final rows -> LDS [S][CPS] (token-mix operand layout): every wave has to be past its reads of the partial sums
        lds_barrier();
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            *reinterpret_cast<f4*>(&U[row0() * U1S + 16 * (NCB * w + cb) + 4 * g]) = X0[cb];
            if (live1()) *reinterpret_cast<f4*>(&U[(32 + s16) * U1S + 16 * (NCB * w + cb) + 4 * g]) = X1[cb];
        }
        lds_barrier();
        // partial poseFinal over this slice's CPS channels: unit (out block ob, row tile t, 4 k blocks) = 16 MFMAs; wout_reg_img[8][NOB][4][64][4]
        // holds Wout[16 ob + (lane & 15)][64 w8 + 16 q' + 4 (lane >> 4) + j] (ls_api.cpp build_fused_images); this slice: w8 = NCB c .. NCB c + NCB - 1
        float* part = a.cpart + ((size_t)pg * NS + c) * kCoopRows * NOBP;
        const wrsrc_t prs = uniform_rsrc(part);
        const wrsrc_t wrs = rs_wout;
        // out block ob = wv + 8 i, 64 channels (one w8) at a time: four weight fragments for the three row tiles, the next four in flight meanwhile
        constexpr int MAXOB = (NOB + kCoopWaves - 1) / kCoopWaves;
        auto woff = [&](int step) {                                  // step = i * NCB + kq
            const int ob = min(wv + kCoopWaves * (step / NCB), NOB - 1), w8 = NCB * c + (step % NCB);
            return (w8 * NOB + ob) * 4 * 1024;
        };
        f4 An[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) An[qq] = wload4(wrs, lane * 16, woff(0) + qq * 1024);
#pragma unroll 1
        for (int i = 0; i < MAXOB; ++i) {
            const int ob = wv + kCoopWaves * i;                      // wave-uniform
            if (ob >= NOB) break;
            f4 o[NT1];
#pragma unroll
            for (int t = 0; t < NT1; ++t) o[t] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kq = 0; kq < NCB; ++kq) {
                f4 A[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) A[qq] = An[qq];
                const int nxt = woff(i * NCB + kq + 1);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) An[qq] = wload4(wrs, lane * 16, nxt + qq * 1024);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    f4 Bv[NT1];
#pragma unroll
                    for (int t = 0; t < NT1; ++t) Bv[t] = *reinterpret_cast<const f4*>(&U[min(16 * t + s16, S - 1) * U1S + 16 * (4 * kq + qq) + 4 * g]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < NT1; ++t) o[t] = MFMA(A[qq][j], Bv[t][j], o[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < NT1; ++t)
                if (16 * t + s16 < S) st_sc1(o[t], prs, ((16 * t + s16) * NOBP + 16 * ob + 4 * g) * 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        unsigned long long* fl = a.cflag + (size_t)bl * 16;
        if (tid == 0) gran_store(fl + j16, tagF, 0.f);
        stamp(2 + 8 * a.layers);
        if (tid < 64) {
            const int k = min(tid, n16 - 1);
            for (unsigned spins = 0;; ++spins) {
                const bool ok = (unsigned)(gran_load(fl + k) >> 32) == tagF;
                if (__all(ok)) break;
                if (spin_bad || spins > kCoopSpinLimit) { spin_bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        lds_barrier();
    }
    stamp(3 + 8 * a.layers);
    if (spin_bad && lane == 0) atomicOr(a.cerr, 1u);

    // ====== sum of the NS partials of each pass, CFG lerp (cfg_sampler.py:31), posterior / DDIM update (gaussian_diffusion.py:260-282,
    //        507-558, 745-798): quad (frame f, columns 4 cq .. + 3), dealt over the sample's workgroups ======================
    {
        const int per = (NQUAD + n16 - 1) / n16;
        const float sc = (np == 2 && a.scale) ? a.scale[b] : 1.0f;
        const unsigned long long gidx = (a.call ? a.call->sample_offset : 0ull) + (unsigned long long)b;
        const size_t base = (size_t)b * kT * JF;
        const wrsrc_t p0 = uniform_rsrc(a.cpart + (size_t)(bl * np) * NS * kCoopRows * NOBP);
        // Few quads per workgroup (TED: 17): one thread per output ELEMENT (a quad's four columns on four lanes: 4x the threads for the
        // Philox draws and the update; the same sums in the same order).  Many (BEAT: 77): one thread per quad -- measured 1 % faster
        // there at 32 clips, 2 % slower for TED.
        constexpr bool kByElement = 4 * ((NQUAD + 2 * NS - 1) / (2 * NS)) <= 128 * NCB;
        if constexpr (kByElement) {
        for (int e = tid; e < 4 * per; e += kCoopThreads) {
            const int quad = j16 * per + (e >> 2), jj = e & 3;
            if (quad >= NQUAD) break;
            const int f = quad / (NOBP / 4), cq = quad - f * (NOBP / 4);
            const int cc = 4 * cq + jj;
            if (cc >= JF) continue;
            const int off = ((NPRE + f) * NOBP + cc) * 4;
            float pc[NS], pu[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) pc[s] = ld_sc1_1(p0, off + s * kCoopRows * NOBP * 4);
            if (np == 2) {
#pragma unroll
                for (int s = 0; s < NS; ++s) pu[s] = ld_sc1_1(p0, off + (NS + s) * kCoopRows * NOBP * 4);
            }
            const int idx = f * JF + cc;
            const float xt = a.sampler != kNone ? a.x_in[base + idx] : 0.f;
            float oc = pc[0], ou = 0.f;
#pragma unroll
            for (int s = 1; s < NS; ++s) oc += pc[s];
            if (np == 2) {
                ou = pu[0];
#pragma unroll
                for (int s = 1; s < NS; ++s) ou += pu[s];
            }
            const float bo = p_bout[cc];
            oc += bo;
            float x0;
            if (np == 2) {
                ou += bo;
                if (a.fwd_c) a.fwd_c[base + idx] = oc;
                if (a.fwd_u) a.fwd_u[base + idx] = ou;
                x0 = ou + sc * (oc - ou);
            } else {
                x0 = oc;                                             // scale == 1: the CFG combination is the cond output
            }
            if (a.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            if (a.x0_out) a.x0_out[base + idx] = x0;
            if (a.sampler != kNone) {
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
        } else {
        for (int i = tid; i < per; i += kCoopThreads) {
            const int quad = j16 * per + i;
            if (quad >= NQUAD) break;
            const int f = quad / (NOBP / 4), cq = quad - f * (NOBP / 4);
            const int off = ((NPRE + f) * NOBP + 4 * cq) * 4;
            f4 pc[NS], pu[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) pc[s] = ld_sc1(p0, off + s * kCoopRows * NOBP * 4);
            if (np == 2) {
#pragma unroll
                for (int s = 0; s < NS; ++s) pu[s] = ld_sc1(p0, off + (NS + s) * kCoopRows * NOBP * 4);
            }
            f4 oc4 = pc[0], ou4 = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 1; s < NS; ++s) oc4 += pc[s];
            if (np == 2) {
                ou4 = pu[0];
#pragma unroll
                for (int s = 1; s < NS; ++s) ou4 += pu[s];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int cc = 4 * cq + jj;
                if (cc >= JF) continue;
                const int idx = f * JF + cc;
                const float bo = p_bout[cc];
                const float oc = oc4[jj] + bo;
                float x0;
                if (np == 2) {
                    const float ou = ou4[jj] + bo;
                    if (a.fwd_c) a.fwd_c[base + idx] = oc;
                    if (a.fwd_u) a.fwd_u[base + idx] = ou;
                    x0 = ou + sc * (oc - ou);
                } else {
                    x0 = oc;                                         // scale == 1: the CFG combination is the cond output
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
    }
    stamp(5 + 8 * a.layers);
}

}  // namespace ls
