// The eight MLPblocks of a LONG sequence (up to 160 tokens: BASELINE configs[4] "as worded", 150 frames + 2 prefix tokens) in ONE launch,
// sample-split like ls_coop_kernel.h: a (sample, CFG pass) group is spread over FOUR workgroups of 128 channels, one per CU, which exchange
// LayerNorm partials and rows inside the launch.  Replaces sixteen launches per step of the batch-level path (ls_long.hip: a token-mixing
// kernel + a channel-mixing GEMM per layer), the assembly launch (the token sequence is put together in the kernel's load) and the poseFinal
// GEMM (every slice multiplies its 128 channels into all output columns in the kernel's tail; the update kernel sums the four partial
// products in slice order); the x_t projection GEMM in front and the sampler update behind stay launches of their own: 3 per step.  (Folding the
// x_t projection in as well would compute it once per CFG pass -- twice the matrix work of the GEMM it replaces -- and was not built.)
// Same reference arithmetic:
//   TransMLP / MLPblock / LN_spatial    scripts/model/mlp_module.py:21-91   (per-sample independence, :67-91, is what makes the split legal)
// (the reference itself cannot run this shape: token mixing fixes S, scripts_beat/model/RAG.py:56 -- synthetic, self-pinned to oracle/).
//
// Mapping:
//   * workgroup = (group, slice c of 128 channels), 8 waves.  Wave (w, h): channel blocks 2 w, 2 w + 1 of the slice (32 channels, MFMA C/D
//     layout: lane (s16, g) holds channels 4 g .. 4 g + 3 of a block for row s16 of a tile), row tiles 5 h .. 5 h + 4 of the ten -- for
//     EVERYTHING: residual stream, LayerNorms, token mixing, channel mixing (no split of k between the halves, no partial-sum swap).
//   * token mixing  D[channel][row] = sum_r' u[r'][channel] Wt[row][r']: the operand u = LN1(x) of the slice's own channels, all rows, in LDS
//     [160][144]; the Conv1d weights stream from an L2-resident per-lane image, one 16-row k block ahead.
//   * channel mixing contracts over all 512 channels = 32 k blocks of 16, but 160 rows x 512 channels do not fit LDS: the k blocks pass
//     through an 8-slot LDS ring [slot][160 rows][16].  The slice's own eight blocks are written into the ring from registers; the other
//     24 are pulled global -> LDS by LDS-DMA from the exchange buffer, two blocks per pair of iterations into the two slots freed by the
//     previous pair (six blocks = ~17 k clocks ahead of their use).  One workgroup barrier per PAIR of k blocks: it says both "blocks 2 pr,
//     2 pr + 1 have landed for every wave" (every wave has waited for all but its last 16 memory operations) and "everyone is done with pair
//     pr - 1".  VMEM returns in issue order on gfx950, so a wait for a weight fragment also waits for every pull issued before it: the
//     fragments are requested four blocks ahead, behind pulls that are then at least two pairs old.  LayerNorm 2 is folded around the product
//      with the rows centred on the LayerNorm-1 mean.  The rows are stored write-through and NOT waited for: the ready
//     flag goes up from inside the product (pair 1), the other slices' flags are looked at in pair 2.
//   * hand-offs: the protocol of ls_coop_kernel.h (write-through payload, every wave drains, barrier, {tag, value} granules that double as
//     ready flags, bounded spins, tags unique per launch / sync point / call).  The four slices of a group sit on ONE XCD when the grid is
//     dealt round-robin (blockIdx = (group / 8 * 4 + slice) * 8 + group % 8) -- speed only.
#pragma once
#include "ls_coop_kernel.h"

#ifndef LS_MIX_ABL
#define LS_MIX_ABL 0            // A/B builds only (tools/ab_variants.py): 1 no channel-mix MFMAs, 2 no token-mix MFMAs, 4 no ring refills, 8 no k-block loop, 16 no barriers in it -- wrong results
#endif

#ifndef LS_MIX_LNSUM
#define LS_MIX_LNSUM 1          // LayerNorm partials as centred sums (sum, sum of squares around the row's previous mean) instead of (mean, M2) Chan merges
#endif

namespace ls {

constexpr int kMixThreads = 512;
constexpr int kMixUS = 144;                 // LDS row stride of the token-mix operand [160][128]: = 16 mod 32 (the four lane groups read four consecutive rows), 16-byte aligned rows
constexpr int kMixSlots = 8;                // ring slots of [160][16] floats
constexpr int kMixPFW = 4;                  // k blocks the weight fragments are requested ahead
// LDS: pst [4][160] f2 | stat [160] f2 | U: max(token-mix operand 160 x 144, ring 8 x 160 x 16)
constexpr int kMixLdsFloats = 2 * 4 * kMixRows + 2 * kMixRows + kMixRows * kMixUS;
static_assert(kMixSlots * kMixRows * 16 <= kMixRows * kMixUS, "the ring overlays the token-mix operand");

template <int S>
__global__ __launch_bounds__(kMixThreads, 2) void k_mix(const MixArgs a) {
    constexpr int NT = (S + 15) / 16, TH = NT / 2, NS = kMixSlices, NCB = 2, KB = 8, RP = kMixRows, US = kMixUS;
    static_assert(NT == 10 && NT * 16 == RP, "ten row tiles (S in 145 .. 160)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f2* pst = reinterpret_cast<f2*>(smem);                 // [4 waves of a half... both halves own disjoint rows][160] (mean, M2) over the wave's 32 channels
    f2* stat = pst + 4 * RP;                               // [160] (mean, rstd) over all 512 channels
    float* U = smem + 2 * 4 * RP + 2 * RP;                 // token-mix operand [160][144] | ring [8][160][16]

    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3, h = wv >> 2;
    const int bid = blockIdx.x;
    const int c = (bid >> 3) & (NS - 1);                   // slice
    const int pg = (bid >> 3) / NS * 8 + (bid & 7);        // launch-local group: its four slices share bid % 8, i.e. one XCD under round-robin placement
    if (pg >= a.ngroups) return;
    int s16 = lane & 15, g = lane >> 4;
    auto fresh = [&]() { asm volatile("" : "+v"(lane)); s16 = lane & 15; g = lane >> 4; };
    auto gblk = [&](int cb) { return KB * c + NCB * w + cb; };                // 16-channel block over all 512 channels
    auto chw = [&](int cb) { return 16 * gblk(cb) + 4 * g; };
    auto rowi = [&](int i) { return 16 * (TH * h + i) + s16; };               // this lane's row of its i-th tile
    auto live = [&](int i) { return rowi(i) < S; };

    const float* xin = a.x_in + (size_t)pg * a.group_stride;
    float* xout = a.x_out + (size_t)pg * a.group_stride;
    float* xg = a.xg + (size_t)pg * 32 * RP * 16;
    unsigned long long* gran = a.gran + (size_t)pg * (2 * RP + 1) * NS * 2;       // [2 areas][160 rows][4 slices][2] statistics granules + [4 slices][2] rows-ready flags
    const wrsrc_t xrs = uniform_rsrc(xg);
    const unsigned ep = a.epoch + (a.call ? a.call->tag_base : 0u);
    unsigned spin_bad = 0;
    auto stamp = [&](int idx) {
#ifdef LS_DEBUG
        if (a.prof && bid == a.prof_wg && lane == 0 && idx < kProfPoints) a.prof[wv * kProfPoints + idx] = __builtin_amdgcn_s_memtime();
#else
        (void)idx;
#endif
    };
    stamp(0);
    const wrsrc_t rs_ln1a = wrsrc(a.ln1a), rs_ln1b = wrsrc(a.ln1b), rs_wtok = wrsrc(a.wtok_img), rs_wch = wrsrc(a.wch_img), rs_bch = wrsrc(a.bch),
                  rs_wsum = wrsrc(a.wsum);
    const gfp p_btok = g1(a.btok);

    // The token sequence entering block 0: either assembled rows from x_in, or (a.xproj != null) assembled HERE as k_long_assemble did --
    // frame rows = x_t projection + static_{c|u}, row 0 = style token mu + eps * std (reparameterize, RAG.py:10-13, 116-120), row 1 = emotion token
    f4 X[NCB][TH];
    if (a.xproj) {
        const int grp = a.g0 + pg, p = grp / a.B, b = grp - p * a.B;            // group = pass * B + sample
        const float* stc = p ? a.static_u : a.static_c;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const int r = rowi(i), ch = chw(cb);
                f4 v = (f4){0.f, 0.f, 0.f, 0.f};
                if (r < S && r >= a.npre) {
                    const size_t fr = ((size_t)b * (S - a.npre) + (r - a.npre)) * kD + ch;
                    v = *reinterpret_cast<const f4*>(a.xproj + fr) + *reinterpret_cast<const f4*>(stc + fr);
                } else if (r == 0) {
                    const f4 mu = *reinterpret_cast<const f4*>(a.z_mu + (size_t)b * kD + ch), sd = *reinterpret_cast<const f4*>(a.z_std + (size_t)b * kD + ch);
                    f4 e;
                    const float* epp = p ? a.eps_u : a.eps_c;
                    if (epp) e = *reinterpret_cast<const f4*>(epp + (size_t)b * kD + ch);
                    else {
                        float z[4];
                        philox_normal4(a.call, a.call->sample_offset + (unsigned long long)(a.b0 + b), a.step_id, 1u + (unsigned)p, (unsigned)(ch >> 2), z);
                        e = (f4){z[0], z[1], z[2], z[3]};
                    }
                    v = mu + e * sd;
                } else if (r < a.npre) {
                    v = *reinterpret_cast<const f4*>(a.emo_tok + (size_t)b * kD + ch);
                }
                X[cb][i] = v;
            }
    } else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < TH; ++i)
                X[cb][i] = live(i) ? *reinterpret_cast<const f4*>(xin + (size_t)rowi(i) * kD + chw(cb)) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    f4 temb4[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) temb4[cb] = *reinterpret_cast<const f4*>(a.temb + chw(cb));

    // (mean, M2) of the lane's 8 channels of one row, merged over the 4 lane groups: the wave's 32 channels
    auto lane_part = [&](f4 v0, f4 v1, float& m, float& m2) {
        const float ma0 = ((v0[0] + v0[1]) + (v0[2] + v0[3])) * 0.25f, mb0 = ((v1[0] + v1[1]) + (v1[2] + v1[3])) * 0.25f;
        const f4 d0 = v0 - (f4){ma0, ma0, ma0, ma0}, d1 = v1 - (f4){mb0, mb0, mb0, mb0};
        const float qa0 = (d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3]);
        const float qb0 = (d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]);
        const float dd = mb0 - ma0;
        m2 = (qa0 + qb0) + dd * dd * 2.0f;                   // equal counts n: M2 = qa + qb + d^2 n / 2
        m = 0.5f * (ma0 + mb0);
        {
            float ma, mb, qa, qb;
            xor16_pair(m, ma, mb); xor16_pair(m2, qa, qb);
            const float d = mb - ma;
            m2 = (qa + qb) + d * d * 4.0f;
            m = 0.5f * (ma + mb);
        }
        {
            float ma, mb, qa, qb;
            xor32_pair(m, ma, mb); xor32_pair(m2, qa, qb);
            const float d = mb - ma;
            m2 = (qa + qb) + d * d * 8.0f;
            m = 0.5f * (ma + mb);
        }
    };
    // The same statistic as plain sums around a per-row centre c: (sum (x - c), sum (x - c)^2) over the wave's 32 channels.  c = the mean the
    // row had at the previous LayerNorm (stat[row].x: LayerNorm 2 is centred on the LayerNorm-1 mean like the rows it normalises, LayerNorm 1
    // on the previous block's LayerNorm-2 mean, block 0 on zero), so the variance E[d^2] - E[d]^2 has no cancellation to speak of; less than half
    // of the Chan form's vector instructions, and sums merge by addition (four waves, four slices).  -0.55 % of the step at 32 clips.
    auto lane_sums = [&](f4 v0, f4 v1, float cc, float& sm, float& sq) {
        const f4 c4 = (f4){cc, cc, cc, cc};
        const f4 d0 = v0 - c4, d1 = v1 - c4;
        const f4 t = d0 + d1;
        const f4 qq = __builtin_elementwise_fma(d1, d1, d0 * d0);
        sm = (t[0] + t[1]) + (t[2] + t[3]);
        sq = (qq[0] + qq[1]) + (qq[2] + qq[3]);
        float x0, x1;
        xor16_pair(sm, x0, x1); sm = x0 + x1;
        xor16_pair(sq, x0, x1); sq = x0 + x1;
        xor32_pair(sm, x0, x1); sm = x0 + x1;
        xor32_pair(sq, x0, x1); sq = x0 + x1;
    };
    // this slice's partial statistics of every row -> granule area `area`; `payload`: write-through stores of this workgroup are drained first
    // `cen`: the rows already centred on stat[row].x (the caller has them in registers), or null
    auto ln_publish = [&](int area, unsigned tag, bool payload, const f4 (*cen)[TH]) {
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            float m, m2;
            if (LS_MIX_LNSUM && cen) lane_sums(cen[0][i], cen[1][i], 0.f, m, m2);
            else if (LS_MIX_LNSUM) lane_sums(X[0][i], X[1][i], stat[rowi(i)].x, m, m2);
            else lane_part(X[0][i], X[1][i], m, m2);
            if (g == 0) pst[w * RP + rowi(i)] = (f2){m, m2};
        }
        if (payload) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (tid < S) {
            f2 pw[4];
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) pw[ww] = pst[ww * RP + tid];
            unsigned long long* gp = gran + (size_t)area * RP * NS * 2 + ((size_t)tid * NS + c) * 2;
            if (LS_MIX_LNSUM) {
                gran_store(gp, tag, (pw[0].x + pw[1].x) + (pw[2].x + pw[3].x));
                gran_store(gp + 1, tag, (pw[0].y + pw[1].y) + (pw[2].y + pw[3].y));
            } else {
                float ms = 0.f, qs = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) { ms += pw[ww].x; qs += pw[ww].y; }
                const float mt = ms * 0.25f;
                float dd = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) { const float d = pw[ww].x - mt; dd = fmaf(d, d, dd); }
                gran_store(gp, tag, mt);
                gran_store(gp + 1, tag, qs + 32.0f * dd);
            }
        }
    };
    // all slices' partials of every row -> stat[row] = (mean, rstd); pad rows get (0, 0)
    auto ln_gather = [&](int area, unsigned tag) {
        const unsigned long long* ga = gran + (size_t)area * RP * NS * 2;
        // thread (row = tid / 4, slice = tid % 4) covers rows 0 .. 127; the first two waves take rows 128 .. 159 as well, in the SAME poll
        // (both rows' granules are requested before either is looked at: one round trip per gather)
        const int sl = tid & (NS - 1), ra = tid / NS, rb = kMixThreads / NS + tid / NS;
        const bool two = wv < 2;                              // wave-uniform
        const unsigned long long* g0 = ga + ((size_t)min(ra, S - 1) * NS + sl) * 2;
        const unsigned long long* g1 = ga + ((size_t)min(rb, S - 1) * NS + sl) * 2;
        unsigned long long v0, v1, w0 = 0, w1 = 0;
        for (unsigned spins = 0;; ++spins) {
            v0 = gran_load(g0); v1 = gran_load(g0 + 1);
            if (two) { w0 = gran_load(g1); w1 = gran_load(g1 + 1); }
            bool ok = (unsigned)(v0 >> 32) == tag && (unsigned)(v1 >> 32) == tag;
            if (two) ok = ok && (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag;
            if (__all(ok)) break;
            if (spin_bad || spins > kCoopSpinLimit) { spin_bad = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        auto merge = [&](unsigned long long x0, unsigned long long x1, int rr) {
            const float pm = __uint_as_float((unsigned)x0), pq = __uint_as_float((unsigned)x1);
            float sm = pm;
            sm = dpp_add<0xB1>(sm); sm = dpp_add<0x4E>(sm);
            if (LS_MIX_LNSUM) {
                float q = pq;
                q = dpp_add<0xB1>(q); q = dpp_add<0x4E>(q);
                if (sl == 0) {
                    const float dm = sm * (1.0f / kD), var = fmaxf(fmaf(-dm, dm, q * (1.0f / kD)), 0.f);
                    stat[rr] = rr < S ? (f2){stat[rr].x + dm, rsqrtf(var + 1e-5f)} : (f2){0.f, 0.f};        // the sums were taken around the old stat[rr].x
                }
                return;
            }
            const float mu = sm * 0.25f, d = pm - mu;
            float q = fmaf(128.0f * d, d, pq);
            q = dpp_add<0xB1>(q); q = dpp_add<0x4E>(q);
            if (sl == 0) stat[rr] = rr < S ? (f2){mu, rsqrtf(q * (1.0f / kD) + 1e-5f)} : (f2){0.f, 0.f};
        };
        merge(v0, v1, ra);
        if (two) merge(w0, w1, rb);
        lds_barrier();
    };

    if (LS_MIX_LNSUM) {            // centre of block 0's LayerNorm 1: zero
        if (tid < RP) stat[tid] = (f2){0.f, 0.f};
        lds_barrier();
    }
    stamp(1);
    for (int l = 0; l < a.layers; ++l) {
        fresh();
        // x = x + emb  (mlp_module.py:68-69)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < TH; ++i) if (live(i)) X[cb][i] += temb4[cb];
        // ---- block1: LN -> token mixing -> SiLU -> residual ---------------------------------------
        ln_publish(0, ep + 2 * l + 1, false, nullptr);
        f4 al1[NCB], be1[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { al1[cb] = wload4(rs_ln1a, chw(cb) * 4, l * kD * 4); be1[cb] = wload4(rs_ln1b, chw(cb) * 4, l * kD * 4); }
        float btb[TH];
#pragma unroll
        for (int i = 0; i < TH; ++i) btb[i] = p_btok[l * S + min(rowi(i), S - 1)];
        // token-mix weights of the first k block, in flight during the exchange: wtok_img[l][q][mt][lane][e]
        const int tsb = l * NT * NT * 1024;
        f4 Bn[TH];
#pragma unroll
        for (int i = 0; i < TH; ++i) Bn[i] = wload4(rs_wtok, lane * 16, tsb + (0 * NT + TH * h + i) * 1024);
        stamp(2 + 10 * l);                                   // LN1 partials published, weights requested
        ln_gather(0, ep + 2 * l + 1);
        stamp(3 + 10 * l);                                   // LN1 statistics gathered
        fresh();
        float mu1[TH];
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            const f2 st = stat[rowi(i)];
            mu1[i] = st.x;
            const float nm = -st.x * st.y;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                f4 u = __builtin_elementwise_fma(X[cb][i], (f4){st.y, st.y, st.y, st.y}, (f4){nm, nm, nm, nm});
                u = __builtin_elementwise_fma(u, al1[cb], be1[cb]);
                *reinterpret_cast<f4*>(&U[rowi(i) * US + 16 * (NCB * w + cb) + 4 * g]) = u;     // pad rows: beta (finite; they meet zero weights)
            }
        }
        lds_barrier();
        stamp(4 + 10 * l);                                   // LN1 applied, operand in LDS
        fresh();
        {
            f4 acc[NCB][TH];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int i = 0; i < TH; ++i) acc[cb][i] = (f4){btb[i], btb[i], btb[i], btb[i]};
#pragma unroll 1
            for (int q = 0; q < NT; ++q) {
                f4 Bv[TH];
#pragma unroll
                for (int i = 0; i < TH; ++i) Bv[i] = Bn[i];
                const int qn = min(q + 1, NT - 1);
#pragma unroll
                for (int i = 0; i < TH; ++i) Bn[i] = wload4(rs_wtok, lane * 16, tsb + (qn * NT + TH * h + i) * 1024);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float av[NCB];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) av[cb] = U[(16 * q + 4 * e + g) * US + 16 * (NCB * w + cb) + s16];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int i = 0; i < TH; ++i) { if (!(LS_MIX_ABL & 2)) acc[cb][i] = MFMA(av[cb], Bv[i][e], acc[cb][i]); else acc[cb][i][e] += av[cb] * Bv[i][e]; }
                }
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int i = 0; i < TH; ++i) if (live(i)) X[cb][i] = silu_acc4(acc[cb][i], X[cb][i]);
        }
        stamp(5 + 10 * l);                                   // token mixing done
        fresh();
        // ---- block2: LN -> channel mixing -> SiLU -> residual -------------------------------------
        const unsigned tag2 = ep + 2 * l + 2;
        {
            // centred rows: write-through to the other slices (exchange order [k block][row][16]); the LDS copy goes into the ring once every wave
            // is past its token-mix reads of the overlay (the barrier inside ln_publish)
            f4 cen[NCB][TH];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int i = 0; i < TH; ++i) {
                    cen[cb][i] = live(i) ? X[cb][i] - (f4){mu1[i], mu1[i], mu1[i], mu1[i]} : (f4){0.f, 0.f, 0.f, 0.f};
                    st_sc1(cen[cb][i], xrs, ((gblk(cb) * RP + rowi(i)) * 16 + 4 * g) * 4);
                }
            ln_publish(1, tag2, false, cen);  // LayerNorm-2 partials: they need no payload.  The rows are NOT waited for here: 80 KB of write-through stores
                                              // take ~5 k clocks to drain -- the ready flag goes up from inside the product (k block 2), behind the waits that are there anyway
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int i = 0; i < TH; ++i) *reinterpret_cast<f4*>(&U[((NCB * w + cb) * RP + rowi(i)) * 16 + 4 * g]) = cen[cb][i];     // slot = local k block
        }
        stamp(6 + 10 * l);                                   // rows published, ring holds the own slice
        fresh();
        {
            f4 bc[NCB], ws4[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) { bc[cb] = wload4(rs_bch, chw(cb) * 4, l * kD * 4); ws4[cb] = wload4(rs_wsum, chw(cb) * 4, l * kD * 4); }
            f4 acc[NCB][TH];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int i = 0; i < TH; ++i) acc[cb][i] = (f4){0.f, 0.f, 0.f, 0.f};
            // k block n of this wave's order: the slice's own eight first, then the other slices in ring order
            auto qof = [&](int n) { return ((c + (n >> 3)) & (NS - 1)) * KB + (n & 7); };
            auto wsb = [&](int cb) { return (l * 32 + gblk(cb)) * 32 * 1024; };
            constexpr int PFW = kMixPFW;
            f4 An[PFW][NCB];
#pragma unroll
            for (int k = 0; k < PFW; ++k)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) An[k][cb] = wload4(rs_wch, lane * 16, wsb(cb) + qof(k) * 1024);
            unsigned long long* rdy = gran + (size_t)2 * RP * NS * 2;           // [4 slices][2] rows-ready flags
            // one ring refill: k block n -> slot n % 8; ten 1 KiB chunks, wave wv takes chunk wv and chunk 8 + (wv & 1) (the doubled pulls of
            // chunks 8 and 9 write the same bytes: every wave issues exactly two)
            auto refill = [&](int n) {
                const int q = qof(n), slot = n & (kMixSlots - 1);
                dma_sc1(xrs, U + slot * RP * 16 + wv * 256, lane * 16, (q * RP * 16 + wv * 256) * 4);
                dma_sc1(xrs, U + slot * RP * 16 + (8 + (wv & 1)) * 256, lane * 16, (q * RP * 16 + (8 + (wv & 1)) * 256) * 4);
            };
            typedef const __attribute__((address_space(3))) f4* ldsp4;
            // Two k blocks per workgroup barrier (sixteen barriers per layer): the barrier of pair pr says "blocks 2 pr and 2 pr + 1 have landed
            // for every wave" (each wave has waited for all but its last 16 memory operations: every pull older than two pairs) and "everyone is
            // done with pair pr - 1", whose two slots are refilled with blocks 2 pr + 6 and 2 pr + 7.  The second block's LDS operands are read
            // while the first is multiplied.
            static_assert(PFW == 4, "two pairs of weight fragments in flight");
#pragma unroll 1
            for (int t4 = 0; t4 < 8; ++t4)
#pragma unroll
            for (int u2 = 0; u2 < 2; ++u2) {
                const int pr = 2 * t4 + u2;
                if (LS_MIX_ABL & 8) break;
                if (pr == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's row stores (issued ahead of every load still in flight) have drained ...
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                if (!(LS_MIX_ABL & 16)) lds_barrier();
                if (pr == 1 && tid == 0) gran_store(rdy + (size_t)c * 2, tag2, 0.f);        // ... and so have every other wave's: the slice's rows are published
                if (pr == 2) {
                    // the other slices' rows: wait for their ready flags once (long up: four own blocks have been multiplied), then fill the four free slots
                    for (unsigned spins = 0;; ++spins) {
                        const bool ok = (unsigned)(gran_load(rdy + (size_t)(lane & (NS - 1)) * 2) >> 32) == tag2;
                        if (__all(ok)) break;
                        if (spin_bad || spins > kCoopSpinLimit) { spin_bad = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!(LS_MIX_ABL & 4)) { refill(8); refill(9); refill(10); refill(11); }
                    stamp(7 + 10 * l);                                           // four own blocks multiplied, first refills issued
                } else if (pr > 2 && 2 * pr + 7 < 32) {
                    if (!(LS_MIX_ABL & 4)) { refill(2 * pr + 6); refill(2 * pr + 7); }
                }
                if (pr == 3) ln_gather(1, tag2);                                 // LayerNorm-2 statistics (published ahead of the product)
                if (pr == 4) stamp(8 + 10 * l);                                  // own half done
                f4 Bv[TH], Bw[TH];
                {
                    const float* ub = U + ((2 * pr) & (kMixSlots - 1)) * RP * 16 + 4 * g;
#pragma unroll
                    for (int i = 0; i < TH; ++i) Bv[i] = *(ldsp4)(ub + rowi(i) * 16);
#pragma unroll
                    for (int i = 0; i < TH; ++i) Bw[i] = *(ldsp4)(ub + RP * 16 + rowi(i) * 16);
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int n = 2 * pr + v, u = 2 * u2 + v;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                            for (int i = 0; i < TH; ++i) {
                                const float bval = v ? Bw[i][j] : Bv[i][j];
                                if (!(LS_MIX_ABL & 1)) acc[cb][i] = MFMA(An[u][cb][j], bval, acc[cb][i]); else acc[cb][i][j] += An[u][cb][j] * bval;
                            }
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) An[u][cb] = wload4(rs_wch, lane * 16, wsb(cb) + qof(min(n + PFW, 31)) * 1024);
                }
            }
            stamp(9 + 10 * l);                               // product done
            fresh();
            // LayerNorm 2 around the product: v = rstd2 * (acc - (mu2 - mu1) wsum) + b'
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const f2 st = stat[rowi(i)];
                const float dm = st.x - mu1[i];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const f4 v = (acc[cb][i] - (f4){dm, dm, dm, dm} * ws4[cb]) * (f4){st.y, st.y, st.y, st.y} + bc[cb];
                    if (live(i)) X[cb][i] = silu_acc4(v, X[cb][i]);
                }
            }
            lds_barrier();              // every wave is past its reads of the ring and of stat before the next layer writes them
        }
        stamp(11 + 10 * l);                                  // epilogue done (= stamp 1 of the next layer)
    }
    if (spin_bad && lane == 0) atomicOr(a.err, 1u);
    if (!a.pout) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < TH; ++i)
                if (live(i)) *reinterpret_cast<f4*>(xout + (size_t)rowi(i) * kD + chw(cb)) = X[cb][i];
        return;
    }
    // ---- poseFinal (OutputProcess, scripts/model/RAG.py poseFinal Linear) on this slice's 128 channels: a partial product over k = the slice's eight
    // 16-channel blocks into ALL output columns; the update kernel sums the four slices' partials in slice order and adds the bias.  The rows go through
    // the ring slots as in channel mixing (slot = local k block; the end-of-layer barrier has passed), the weights stream from a per-lane image.
    // Wave (w, h): column tiles w, w + 4, ... (five at most), row tiles 5 h .. 5 h + 4.  (Dealing the 18 x 10 (column, row) tiles evenly -- 23 / 22 per
    // wave instead of 25 / 20 -- through a per-tile predicate measured 0.8 % SLOWER: a scalar branch per MFMA costs more than the imbalance.)
    fresh();
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int i = 0; i < TH; ++i)
            *reinterpret_cast<f4*>(&U[((NCB * w + cb) * RP + rowi(i)) * 16 + 4 * g]) = live(i) ? X[cb][i] : (f4){0.f, 0.f, 0.f, 0.f};
    constexpr int PT = 5;
    const wrsrc_t rs_wp = wrsrc(a.wpose_img);
    auto wpb = [&](int t, int ql) { return ((w + 4 * t) * 32 + KB * c + ql) * 1024; };
    const int nt_w = (a.npt - w + 3) / 4;                   // column tiles of this wave (wave-uniform)
    f4 Pn[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) Pn[t] = t < nt_w ? wload4(rs_wp, lane * 16, wpb(t, 0)) : (f4){0.f, 0.f, 0.f, 0.f};
    f4 pacc[PT][TH];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int i = 0; i < TH; ++i) pacc[t][i] = (f4){0.f, 0.f, 0.f, 0.f};
    lds_barrier();
    typedef const __attribute__((address_space(3))) f4* ldsq4;
#pragma unroll 1
    for (int ql = 0; ql < KB; ++ql) {
        f4 Pv[PT], Bv[TH];
#pragma unroll
        for (int t = 0; t < PT; ++t) Pv[t] = Pn[t];
        const int qn = min(ql + 1, KB - 1);
#pragma unroll
        for (int t = 0; t < PT; ++t) if (t < nt_w) Pn[t] = wload4(rs_wp, lane * 16, wpb(t, qn));
#pragma unroll
        for (int i = 0; i < TH; ++i) Bv[i] = *(ldsq4)(U + ql * RP * 16 + 4 * g + rowi(i) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < PT; ++t)
                if (t < nt_w) {
#pragma unroll
                    for (int i = 0; i < TH; ++i) pacc[t][i] = MFMA(Pv[t][j], Bv[i][j], pacc[t][i]);
                }
    }
    fresh();
    {
        const int ldp = 16 * a.npt;
        float* po = a.pout + ((size_t)(a.g0 + pg) * NS + c) * S * ldp;
#pragma unroll
        for (int t = 0; t < PT; ++t)
            if (t < nt_w) {
#pragma unroll
                for (int i = 0; i < TH; ++i)
                    if (live(i)) *reinterpret_cast<f4*>(po + (size_t)rowi(i) * ldp + 16 * (w + 4 * t) + 4 * g) = pacc[t][i];
            }
    }
}

}  // namespace ls
