// WavEncoder Conv1d layers 2-4 (scripts/model/audio_enc.py:12-19) as implicit GEMM on v_mfma_f32_16x16x4_f32:
//     out[b][co][p] = bias[co] + sum_{ci,k} W[co][ci][k] * act(in[b][ci][p*6 + k])
// with act = previous layer's InstanceNorm1d + LeakyReLU(0.3) applied while the input window is staged in LDS.
// MFMA M axis = 16 output channels, N axis = 16 output positions, K = (k, ci) in k-MAJOR order inside each chunk of
// 16 input channels: the four K entries of one MFMA are 4 consecutive input channels at the same tap k, so lane
// (position p, g) reads lds[(ci0+g)][p*6 + k] = lane base + compile-time offset (no per-lane div/mod), and the weight
// operand W[co][ci0+g][k] is pre-permuted on the host into per-lane order (one float4 = 4 consecutive MFMA steps).
// Workgroup = 64 positions x 64 output channels of one sample; 4 consumer waves (16 channels x the 4 position tiles each) + 4 producer waves.
#include "ls_internal.h"
#include "ls_lanes.h"
#include "ls_train.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));       // a float4 at dword alignment (global memory takes it)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kCvK = 15, kCvS = 6, kCvTP = 64, kCvCI = 16, kCvTC = 64;

// Stage stamps for tools/conv_bench.cpp (built with -DLS_CONV_PROF; never in the shipped library): lane 0 of one consumer and one
// producer wave of ONE workgroup records the cycle counter when it reaches / leaves each per-stage barrier.
#ifdef LS_CONV_PROF
__device__ unsigned long long* g_conv_prof = nullptr;      // [8 waves][1024 stages][2]
__device__ int g_conv_prof_wg = 0;
#define CV_STAMP(role, sidx, which)                                                                                       \
    do {                                                                                                                  \
        if (g_conv_prof && (int)blockIdx.z == g_conv_prof_wg && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (sidx) < 1024) \
            g_conv_prof[(((role) * 1024) + (sidx)) * 2 + (which)] = __builtin_readcyclecounter();                          \
    } while (0)
#else
#define CV_STAMP(role, sidx, which) do { } while (0)
#endif
// timing-only ablations for tools/build_conv_bench.sh (-DLS_CONV_ABL=bits; results are wrong): 1 consumers read no LDS operands,
// 2 consumers fetch no weight fragments, 4 producers load nothing, 8 no epilogue, 16 producers do nothing at all
#ifndef LS_CONV_ABL
#define LS_CONV_ABL 0
#endif
#ifndef LS_CONV_WGS
#define LS_CONV_WGS 2          // workgroups per CU the stride-6 forward kernel is compiled for (register budget)
#endif
constexpr int kCvWin = (kCvTP - 1) * kCvS + kCvK;      // 393 input samples per channel per tile
constexpr int kCvWinP = kCvWin + 4;                    // 397: odd stride -> the 4 lane groups (channels) hit different banks

// wimg: [co tile (Cout/16)][chunk (Cin/16)][tap (15)][lane 64][4]: lane (s16, g) of tap k holds W[co = 16 ct + s16][ci = 16 chunk + 4 cig + g][k],
//       cig = 0..3 (build_shared_weights() in ls_api.cpp): one float4 = the four MFMA steps of one tap.
//
// Producer / consumer workgroup (512 threads).  Waves 4-7 only stage: they fetch the NEXT 16-channel window (25 clamped loads per
// thread, in flight together), apply the previous layer's InstanceNorm + LeakyReLU and write it to the other half of a
// double-buffered LDS window.  Waves 0-3 only multiply: wave w = channel tile w x the four position tiles, so one float4 of
// weights per tap feeds 16 MFMAs (15 fragment loads per stage and wave, a ring three taps ahead).  One workgroup barrier per
// (tile, chunk) stage; a workgroup walks `tpw` consecutive 64-position tiles of one (sample, 64-channel group) as ONE pipeline.
//
// Round 3.  In-kernel stamps (tools/conv_bench.cpp -DLS_CONV_PROF, profiles/r03b) of rounds 1-2's form showed the PRODUCERS
// waiting 14.5 k of every 25.4 k-cycle stage: the consumers set the pace, two of them per SIMD at 7.7 k cycles of MFMA issue per
// stage each, because (1) every MFMA had its own ds_read_b32 + v_add_u32 with the wait right behind it (20.5 k cycles per stage in
// the loop) and (2) the per-tile epilogue was 32 dependent DPP reduction chains and 64 separately predicated stores (10 k cycles).
// Tried first: consumer wave = one position tile x all four channel tiles (one LDS value feeds four MFMAs) -- but then every wave
// streams the whole 61 KB weight chunk per stage: 480 KB per CU and stage through the 64 B/clk vector-memory path, behind the
// producers' window loads in the same in-order queues; stages of 26 - 58 k cycles (553 us).  And a one-role form (every wave
// multiplies and stages a quarter of the next window, three 256-thread workgroups per CU): VMEM loads return in order, so a wave
// with window loads in flight waits for them at its next weight fragment (553 us too).
// Kept: the weight reuse of the old mapping (16 MFMAs per fragment) AND one LDS read per four MFMAs, by laying the window out as
// [row][r][pt]: element x = 96 pt + r of a row (position tile pt, offset r = 6 s16 + tap) sits at slot r, component pt, so the four
// position tiles' operands of one (lane, tap) are ONE ds_read_b128.  r runs to 104 > 95, so the first 9 elements of tiles 1..3
// (and the 9 past the last tile) are written twice, also as slots 96..104 of the previous tile.  105 slots = an odd number of
// 16-byte units per row: conflict-free under gfx950's b128 lane groups for every tap (checked exhaustively).  The epilogue keeps
// ONE (count, mean, M2) partial per (channel, 64-position tile) instead of four.  The convolution results are bit-identical to the
// old kernel's (same MFMA order per accumulator); the statistics are merged from coarser partials (double-precision merge, ~1e-8).
constexpr int kCvSlots = 105, kCvRow = 4 * kCvSlots;       // floats per window row in the [r][pt] layout
constexpr int kCvOutLd = kCvTP + 4;                        // row stride of the accumulator hand-off tile (2-way write conflicts only)

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt: every global load in flight --
// the producers' look-ahead window, the consumers' weight ring -- and every output store would have to land before each stage's barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// this thread's 25 window elements through a buffer descriptor over the whole input tensor: SGPR base + one 32-bit lane offset +
// immediates (3.7 instead of 13 matrix-pipe cycles of issue per load, and no address arithmetic at all).  No clamping: past the end of
// a row the loads read the next row (finite values the tail select discards), past the end of the tensor the descriptor returns 0.
__device__ __forceinline__ void conv_stage_load(float (&vals)[25], __amdgpu_buffer_rsrc_t rs, int off) {
#pragma unroll
    for (int q = 0; q < 25; ++q) vals[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 64 * q, 0, 0));
}
// element o = c + 16 q of the row: position tile o / 96 = q / 6, offset r = o % 96 = c + 16 (q % 6) (c < 16)
template <bool TAIL>
__device__ __forceinline__ void conv_stage_store(const float (&vals)[25], float* __restrict__ dst, int c, int validw, float vm, float vr) {
    float* d = dst + 4 * c;
#pragma unroll
    for (int q = 0; q < 25; ++q) {
        const int o = c + 16 * q;
        float v = (vals[q] - vm) * vr;                                  // InstanceNorm1d + LeakyReLU(0.3), audio_enc.py:10-11
        v = fmaxf(v, 0.3f * v);
        if (TAIL) v = o < validw ? v : 0.f;
        if (q < 24) d[64 * (q % 6) + q / 6] = v;                        // slot r, component pt
        if (q >= 6 && q % 6 == 0 && c < 9) d[4 * 96 + q / 6 - 1] = v;   // the same element as slot 96 + r of the previous tile
    }
}

__global__ __launch_bounds__(512, LS_CONV_WGS) void k_conv1d_mfma(const float* __restrict__ in, const float* __restrict__ stats,
                                                           const float* __restrict__ wimg, const float* __restrict__ bias,
                                                           float* __restrict__ out, float* __restrict__ spart, int Cin, int Cout, int Lin, int Lout,
                                                           int ntile, int tpw) {
    __shared__ __attribute__((aligned(16))) float sIn[2][kCvCI * kCvRow];
    __shared__ __attribute__((aligned(16))) float sOut[kCvTC * kCvOutLd];     // a finished tile's accumulators, consumers -> producers
    const int b = blockIdx.z, co0 = blockIdx.y * kCvTC;
    const int t0 = blockIdx.x * tpw, t1 = min(ntile, t0 + tpw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = Cin / kCvCI;
    const int nstage = (t1 - t0) * nchunk;
    if (w >= 4) {
        // ---------------- producers: thread (row sr of the 16-channel window, columns sc + 16 q).  (Loads TWO stages ahead -- window s+2
        // requested before window s+1 is written, straight-line so that the waits really are vmcnt(25) -- measured slower, 496 vs
        // 482 us: fetch latency is not what the stage waits for.)
        const int pt = tid - 256, sr = pt >> 4, sc = pt & 15;
        // descriptor over THIS sample's rows onwards (32-bit offsets: the whole tensor is 4 GB at 4096 clips); it ends with the tensor
        const long long left = (long long)(gridDim.z - b) * Cin * Lin * 4;
        const auto rsin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)b * Cin * Lin, 0, (int)min(left, 0x7fffffffll), 0x00020000);
        auto stage = [&](int sidx, float* buf) {
            const int tl = sidx / nchunk, c = sidx - tl * nchunk;
            const int in0 = (t0 + tl) * kCvTP * kCvS;
            const int validw = min(Lin - in0, kCvWin);                     // >= 1 for every tile that has an output position
            const int row = c * kCvCI + sr;                                 // row of this sample
            const float vm = stats[((size_t)b * Cin + row) * 2], vr = stats[((size_t)b * Cin + row) * 2 + 1];
            float vals[25];
            if (LS_CONV_ABL & 16) return;
            if (LS_CONV_ABL & 4) {
#pragma unroll
                for (int q = 0; q < 25; ++q) vals[q] = vm + (float)q;
            } else conv_stage_load(vals, rsin, (row * Lin + in0 + sc) * 4);
            if (validw >= kCvWin) conv_stage_store<false>(vals, buf + sr * kCvRow, sc, validw, vm, vr);
            else conv_stage_store<true>(vals, buf + sr * kCvRow, sc, validw, vm, vr);
        };
        // The tile epilogue belongs to the producers too (round 3): the consumers drop a finished tile's accumulators into sOut and go
        // straight on multiplying; one stage later the producers -- which wait ~2.5 k cycles per stage anyway -- add the bias, store
        // whole 256-byte row pieces and reduce the InstanceNorm partials (a 4-lane reduction per channel instead of a 16-lane one).
        // In the consumers the 20 output stores sat in the same in-order vmcnt queue as the weight fragments: every tile began by
        // waiting for its predecessor's stores to be acknowledged (timing-only ablation without the epilogue: - 40 us on conv2).
        const int orow = pt >> 2, oq = pt & 3;                             // drain: thread = (channel row, 16 positions)
        auto drain = [&](int tile) {
            if (LS_CONV_ABL & 8) return;
            const int p0 = tile * kCvTP, pbase = p0 + 16 * oq;
            const float bvv = bias[co0 + orow];
            f4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f4*>(&sOut[orow * kCvOutLd + 16 * oq + 4 * i]) + bvv;
            float* o = out + ((size_t)b * Cout + co0 + orow) * Lout + pbase;
            const int nvl = min(16, max(0, Lout - pbase));                  // valid positions of this thread
            if (nvl == 16) {                                                // rows start at any dword: 4-byte-aligned 16-byte stores
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<f4u*>(o + 4 * i) = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (4 * i + e < nvl) o[4 * i + e] = v[i][e];
            }
            if (spart) {
                // (count, mean, M2) of the (channel, 64-position tile): two-pass inside the tile, the four threads of a row = one DPP quad
                const int nv = min(kCvTP, Lout - p0);
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s1 += 4 * i + e < nvl ? v[i][e] : 0.f;
                s1 = dpp_add<0xB1>(s1); s1 = dpp_add<0x4E>(s1);
                const float mean = s1 / (float)nv;
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = 4 * i + e < nvl ? v[i][e] - mean : 0.f; m2 = fmaf(d, d, m2); }
                m2 = dpp_add<0xB1>(m2); m2 = dpp_add<0x4E>(m2);
                if (oq == 0) {
                    float* sp = spart + (((size_t)b * Cout + co0 + orow) * ntile + tile) * 3;
                    sp[0] = (float)nv; sp[1] = mean; sp[2] = m2;
                }
            }
        };
        stage(0, sIn[0]);
        lds_barrier();
        for (int sidx = 0; sidx < nstage; ++sidx) {
            if (sidx + 1 < nstage) stage(sidx + 1, sIn[(sidx + 1) & 1]);
            // the tile whose last chunk was stage sidx - 2 sits in sOut since the barrier before this one
            if (sidx >= 2 && (sidx - 2) % nchunk == nchunk - 1) drain(t0 + (sidx - 2) / nchunk);
            CV_STAMP(w, sidx, 0);
            lds_barrier();
            CV_STAMP(w, sidx, 1);
        }
        lds_barrier();                                                     // the consumers have dropped the last tile
        drain(t1 - 1);
        return;
    }
    // ---------------- consumers: wave w = channel tile w x position tiles 0..3
    const int s16 = lane & 15, g = lane >> 4;
    const int lbase = g * kCvRow + 4 * (s16 * kCvS);                   // + (4 cig) rows + 4 * tap as immediates
#ifndef LS_CONV_WPRE
#define LS_CONV_WPRE 3
#endif
    constexpr int kWPre = LS_CONV_WPRE, kWRing = 5;                    // taps of weights in flight; ring size (15 taps per stage = 3 turns)
    static_assert(kCvK % kWRing == 0 && kWPre < kWRing, "ring positions must line up across stages");
    f4 ring[kWRing];
    const auto wrs = uniform_rsrc(wimg);
    const int wbase = ((int)blockIdx.y * 4 + w) * nchunk * kCvK * 1024;
    auto wld = [&](int off) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, off, 0)); };
#pragma unroll
    for (int k = 0; k < kWPre; ++k) ring[k] = wld(wbase + k * 1024);
    lds_barrier();

    int sidx = 0;
    for (int tile = t0; tile < t1; ++tile) {
        const int p0 = tile * kCvTP;
        f4 acc[4];                                                     // [position tile]
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunk; ++c, ++sidx) {
            const float* sb = sIn[sidx & 1] + lbase;
            const int wp = wbase + c * kCvK * 1024;
            const int wpn = wbase + ((c + 1 < nchunk) ? c + 1 : 0) * kCvK * 1024;            // next stage's chunk (same tile or the next)
            f4 Bv[2][4];                                               // [tap parity][cig] = the four position tiles' operands
#pragma unroll
            for (int cig = 0; cig < 4; ++cig) Bv[0][cig] = *reinterpret_cast<const f4*>(sb + (4 * cig) * kCvRow);
#pragma unroll
            for (int k = 0; k < kCvK; ++k) {
                // tap k: [operand reads of tap k+1, weight fragment of tap k+3] pinned in front of [the 16 MFMAs of tap k]: left to itself
                // the scheduler sinks every load next to its first use and waits for it there
                if (k + 1 < kCvK && !(LS_CONV_ABL & 1)) {
#pragma unroll
                    for (int cig = 0; cig < 4; ++cig) Bv[(k + 1) & 1][cig] = *reinterpret_cast<const f4*>(sb + (4 * cig) * kCvRow + 4 * (k + 1));
                }
                const int kn = k + kWPre;
                if (!(LS_CONV_ABL & 2)) ring[kn % kWRing] = wld(kn < kCvK ? wp + kn * 1024 : wpn + (kn - kCvK) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                const f4 A = ring[k % kWRing];
#pragma unroll
                for (int cig = 0; cig < 4; ++cig)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = MFMA(A[cig], Bv[k & 1][cig][t], acc[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
            CV_STAMP(w, sidx, 0);
            lds_barrier();
            CV_STAMP(w, sidx, 1);
        }
        // hand the tile to the producers: lane (s16, g) holds the accumulators of out[co0 + 16 w + 4 g + j][p0 + 16 t + s16].  sOut was
        // drained two stages ago at the latest (tiles are nchunk >= 2 stages apart).
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) sOut[(16 * w + 4 * g + j) * kCvOutLd + 16 * t + s16] = acc[t][j];
    }
    lds_barrier();                                                         // matches the producers' last barrier: the final tile is in sOut
}

// The same implicit GEMM for a SHORT output (the last encoder layer: 34 positions from 217 inputs).  With one sample per workgroup
// the 34 positions fill 34 of the 64 columns of the tile above (conv4 ran at 0.36 of the MFMA peak, its matrix pipe as busy as
// conv3's for 63 % of the FLOPs -- profiles/r02a).  Here the N axis enumerates (sample, position) pairs of NS samples: NS = 4 gives
// 136 columns = 8.5 tiles of 16 -> 9 tiles, 94 % useful.  A lane's activation address is per-lane data anyway (base + compile-time
// tap offsets), so columns that straddle two samples cost nothing.  Same producer / consumer split, same weight image.
// Round 3: the window is stored channel-minor, [sample][x][slot 4 g + cig <- channel 4 cig + g] with 20 floats per x, so the four
// MFMA steps of one (tap, column tile) -- channels g, 4 + g, 8 + g, 12 + g at the same x -- are ONE ds_read_b128 (rounds 1-2: one
// ds_read_b32 per MFMA in a [channel][x] window, 0.56 of the matrix peak); 20 = 5 x 16 bytes per x puts the 16 columns of a tile
// (x = 6 p + tap) on 16 different bank quads: the four 16-lane groups of the b128 read are conflict-free.  Measured: no faster
// (185 -> 183 us in the training step; tools/conv_bench.cpp ablations: consumers alone 151 us = 0.72 of the peak, the producers'
// ~400 instructions per wave and chunk on the same SIMDs cost the other 38) -- kept for the quarter of the LDS traffic.
#ifndef LS_CS_ABL
#define LS_CS_ABL 0                      // timing-only: 1 producers stage the first chunk only, 2 no weight loads after the prologue
#endif
constexpr int kCsXS = 20;                                      // floats per window position: 16 channels + 4 (bank spread)
template <int NS, int NPT>
__global__ __launch_bounds__(512) void k_conv1d_short(const float* __restrict__ in, const float* __restrict__ stats,
                                                         const float* __restrict__ wimg, const float* __restrict__ bias,
                                                         float* __restrict__ out, int Cin, int Cout, int Lin, int Lout, int B, int wn) {
    extern __shared__ __attribute__((aligned(16))) float sIn[];    // [2][NS][wn][20]
    const int b0 = blockIdx.z * NS, co0 = blockIdx.y * kCvTC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = Cin / kCvCI;
    const int bufsz = NS * wn * kCsXS;                         // wn = input samples a row of the window holds (<= Lin)
    if (w >= 4) {
        // ---------------- producers: 16 threads per row, 16 rows per pass, NS passes per chunk
        const int pt = tid - 256;
        constexpr int NQ = 15;                                 // 16-wide groups of a window row: wn <= 35 * 6 + 15 = 225
        // producer wave v stages channels v, 4 + v, 8 + v, 12 + v: their slots 4 v .. 4 v + 3 are consecutive, so a wave's stores (16 x of
        // 4 channels) spread over all 32 banks (channels 4 v .. 4 v + 3 would put its 64 lanes on 8)
        const int ci = 4 * ((pt >> 4) & 3) + (pt >> 6), slot = 4 * (ci & 3) + (ci >> 2);  // channel 4 cig + g -> slot 4 g + cig
        auto stage = [&](int c, float* dst) {
            float vals[NS][NQ], vm[NS], vr[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {                     // every load of the chunk first (clamped addresses: branch-free, in flight together)
                const int b = min(b0 + s, B - 1);                               // past the batch: staged but never stored
                const size_t row = (size_t)b * Cin + c * kCvCI + ci;
                vm[s] = stats[row * 2]; vr[s] = stats[row * 2 + 1];             // InstanceNorm1d + LeakyReLU(0.3), audio_enc.py:10-11
                const float* src = in + row * Lin;
#pragma unroll
                for (int q = 0; q < NQ; ++q) vals[s][q] = src[min((pt & 15) + 16 * q, Lin - 1)];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float* d = dst + (s * wn + (pt & 15)) * kCsXS + slot;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float v = (vals[s][q] - vm[s]) * vr[s];
                    v = fmaxf(v, 0.3f * v);
                    if ((pt & 15) + 16 * q < wn) d[16 * q * kCsXS] = v;
                }
            }
        };
        stage(0, sIn);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk && !(LS_CS_ABL & 1)) stage(c + 1, sIn + ((c + 1) & 1) * bufsz);
            __syncthreads();
        }
        return;
    }
    // ---------------- consumers: wave w = channel tile w x all NPT column tiles
    const int s16 = lane & 15, g = lane >> 4;
    f4 acc[NPT];
    int lb[NPT];                                               // lane's window base per column tile (floats)
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
        const int j = min(16 * t + s16, NS * Lout - 1);
        const int s = j / Lout, p = j - s * Lout;
        lb[t] = (s * wn + p * kCvS) * kCsXS + 4 * g;
    }
    const f4 bv = *reinterpret_cast<const f4*>(bias + co0 + 16 * w + 4 * g);
    __syncthreads();
    constexpr int kWPre = 3, kWRing = 5;                       // weight taps in flight (see k_conv1d_mfma)
    f4 ring[kWRing];
    const auto wrs = uniform_rsrc(wimg);
    const int wbase = ((int)blockIdx.y * 4 + w) * nchunk * kCvK * 1024;
    auto wld = [&](int off) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, off, 0)); };
#pragma unroll
    for (int k = 0; k < kWPre; ++k) ring[k] = wld(wbase + k * 1024);
    for (int c = 0; c < nchunk; ++c) {
        const float* sb = sIn + (c & 1) * bufsz;
        const int wp = wbase + c * kCvK * 1024;
        const int wpn = wbase + ((c + 1 < nchunk) ? c + 1 : 0) * kCvK * 1024;
        f4 Bv[2][NPT];                                         // the operands of tap k + 1 are read before tap k's 36 MFMAs
#pragma unroll
        for (int t = 0; t < NPT; ++t) Bv[0][t] = *reinterpret_cast<const f4*>(sb + lb[t]);
#pragma unroll
        for (int k = 0; k < kCvK; ++k) {
            const f4 A = ring[k % kWRing];
            if (!(LS_CS_ABL & 2)) ring[(k + kWPre) % kWRing] = wld((k + kWPre < kCvK) ? wp + (k + kWPre) * 1024 : wpn + (k + kWPre - kCvK) * 1024);
            if (k + 1 < kCvK) {
#pragma unroll
                for (int t = 0; t < NPT; ++t) Bv[(k + 1) & 1][t] = *reinterpret_cast<const f4*>(sb + lb[t] + (k + 1) * kCsXS);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cig = 0; cig < 4; ++cig)
#pragma unroll
                for (int t = 0; t < NPT; ++t) acc[t] = MFMA(A[cig], Bv[k & 1][t][cig], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int j = 16 * t + s16;
        const int s = j / Lout, p = j - s * Lout;
        if (j < NS * Lout && b0 + s < B) {
#pragma unroll
            for (int e = 0; e < 4; ++e) out[((size_t)(b0 + s) * Cout + co0 + 16 * w + 4 * g + e) * Lout + p] = acc[t][e] + bv[e];
        }
    }
}

// stats[row] = (mean, 1/sqrt(biased var + 1e-5)) from np partial (count, mean, M2) triples per row, merged in index order
// with the parallel-variance update (Chan, Golub, LeVeque): exact-arithmetic equivalent of the two-pass statistics.
// `lpr` lanes per row (a power of two >= 4 covering np where it can): conv3's four partials per row take 4 lanes, not a whole
// wave (round 3: 11 + 18 + 32 us of pure launch width at B = 512 before).  The merge tree depends on lpr, i.e. on np only.
__global__ __launch_bounds__(256) void k_stats_merge(const float* __restrict__ spart, float* __restrict__ stats, int rows, int np, int lpr) {
    const int sub = threadIdx.x & (lpr - 1);
    const int row = (blockIdx.x * 256 + threadIdx.x) / lpr;
    const bool live = row < rows;
    const float* sp = spart + (size_t)(live ? row : 0) * np * 3;
    // merged in double: the mean decides the sign of every normalised activation, i.e. which LeakyReLU slope its gradient
    // gets; keeping it within 1 ulp of the exact mean makes that decision agree with a two-pass fp32 reference
    auto merge = [](double& n, double& mean, double& m2, double nb, double mb, double qb) {
        const double nt = n + nb;
        if (nt > 0.0) {
            const double delta = mb - mean, f = nb / nt;
            mean += delta * f;
            m2 += qb + delta * delta * (n * f);
            n = nt;
        }
    };
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int i = sub; i < np; i += lpr) merge(n, mean, m2, (double)sp[3 * i], (double)sp[3 * i + 1], (double)sp[3 * i + 2]);
    for (int o = 1; o < lpr; o <<= 1) {                                // butterfly: every lane of the row ends with the full merge
        const double nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), qb = __shfl_xor(m2, o);
        merge(n, mean, m2, nb, mb, qb);
    }
    if (live && sub == 0) {
        stats[(size_t)row * 2] = (float)mean;
        stats[(size_t)row * 2 + 1] = (float)(1.0 / sqrt(m2 / n + 1e-5));
    }
}

hipError_t launch_stats_merge(const float* spart, float* stats, int rows, int np, hipStream_t st) {
    int lpr = 4;
    while (lpr < np && lpr < 64) lpr <<= 1;
    const long long threads = (long long)rows * lpr;
    hipLaunchKernelGGL(k_stats_merge, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, spart, stats, rows, np, lpr);
    return hipGetLastError();
}

// out_stats != null: also produce the InstanceNorm statistics of the OUTPUT (fused, no second pass over it); spart is a
// workspace of B * Cout * ceil(Lout / 64) * 4 * 3 floats
hipError_t launch_conv1d_mfma(const float* in, const float* stats, const float* wimg, const float* bias, float* out, float* out_stats,
                              float* spart, int B, int Cin, int Cout, int Lin, int Lout, hipStream_t st) {
    if (Cin % kCvCI || Cin < 2 * kCvCI || Cout % kCvTC || !stats || (out_stats && !spart)) return hipErrorInvalidValue;   // >= 2 chunks per tile (hand-off timing)
    if (!out_stats && Lout <= 36 && (Lout - 1) * kCvS + kCvK <= Lin) {
        // short output without statistics (conv4: 34 positions): columns = (sample, position) pairs of 4 samples, 9 tiles
        constexpr int NS = 4, NPT = 9;
        const int wn = (Lout - 1) * kCvS + kCvK;                          // window positions per sample
        const size_t lds = (size_t)2 * NS * wn * kCsXS * sizeof(float);
        static bool attr_set = false;                                     // > 64 KiB of dynamic LDS needs the opt-in, once per process
        if (!attr_set) {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv1d_short<NS, NPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (ea != hipSuccess) return ea;
            attr_set = true;
        }
        if (NS * Lout <= 16 * NPT && lds <= 160 * 1024) {
            hipLaunchKernelGGL((k_conv1d_short<NS, NPT>), dim3(1, Cout / kCvTC, (B + NS - 1) / NS), dim3(512), lds, st, in, stats, wimg, bias, out,
                               Cin, Cout, Lin, Lout, B, wn);
            return hipGetLastError();
        }
    }
    const int ntile = (Lout + kCvTP - 1) / kCvTP;
    // tiles per workgroup: a workgroup walks `tpw` tiles of one (sample, 64-channel group) as one pipeline (one exposed window fill
    // per workgroup); the chip holds 512 workgroups at a time (2 per CU), so pick the split whose last round is fullest:
    // cost = rounds x (stages per workgroup + 1)
    const int nchunk = Cin / kCvCI;
    int tpw = ntile;
    long long best = -1;
    for (int nx = 1; nx <= ntile; ++nx) {
        const int t = (ntile + nx - 1) / nx;
        const long long wgs = (long long)((ntile + t - 1) / t) * (Cout / kCvTC) * B;
        const long long cost = ((wgs + 511) / 512) * ((long long)t * nchunk + 1);
        if (best < 0 || cost < best) { best = cost; tpw = t; }
    }
    dim3 grid((ntile + tpw - 1) / tpw, Cout / kCvTC, B);
    hipLaunchKernelGGL(k_conv1d_mfma, grid, dim3(512), 0, st, in, stats, wimg, bias, out, out_stats ? spart : nullptr, Cin, Cout, Lin, Lout, ntile, tpw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !out_stats) return e;
    return launch_stats_merge(spart, out_stats, B * Cout, ntile, st);
}

// conv1 (audio_enc.py:10): Conv1d(1, 32, 15, stride 5, padding 1600) on the raw waveform + the InstanceNorm statistics of
// its output, one pass.  Thread = one output position, all 32 channels in registers; the 480 weights are wave-uniform
// (scalar loads), the 5130-sample input window of the workgroup's 1024 positions sits in LDS; a thread owns 4 positions
// 256 apart so that the statistics' cross-lane reductions are amortised.  Bound by the 517 MB output write.
constexpr int kC1fQ = 4, kC1fP = 256 * kC1fQ, kC1fWin = (kC1fP - 1) * 5 + 15;
__global__ __launch_bounds__(256) void k_conv1_fwd(const float* __restrict__ wav, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ out, float* __restrict__ spart, int Lin, int Lout, int pad) {
    __shared__ float sw[kC1fWin + 1];
    const int b = blockIdx.y, p0 = blockIdx.x * kC1fP, tid = threadIdx.x;
    const int x0 = p0 * 5 - pad;
    {   // all 21 loads of the thread first, then the LDS writes (as a rolled loop every load waited for the one before it: 21 memory
        // latencies in a row in front of each workgroup's first FMA)
        constexpr int NW = (kC1fWin + 255) / 256;
        float wv_[NW];
        const float* wb = wav + (size_t)b * Lin;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int x = x0 + tid + 256 * q;
            const float v = wb[min(max(x, 0), Lin - 1)];
            wv_[q] = (x >= 0 && x < Lin) ? v : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NW; ++q) if (tid + 256 * q < kC1fWin) sw[tid + 256 * q] = wv_[q];
    }
    __syncthreads();
    // thread tid owns positions p0 + tid + 256 q: every store instruction of a wave covers 64 consecutive positions
    float win[kC1fQ][15];
    bool valid[kC1fQ];
    int nv = 0;                                                      // valid positions of this WAVE (uniform)
    const int wv = tid >> 6;
#pragma unroll
    for (int q = 0; q < kC1fQ; ++q) {
#pragma unroll
        for (int k = 0; k < 15; ++k) win[q][k] = sw[(tid + 256 * q) * 5 + k];
        valid[q] = p0 + tid + 256 * q < Lout;
        nv += min(64, max(0, Lout - (p0 + 256 * q + 64 * wv)));
    }
    const int np = gridDim.x * 4;
#pragma unroll 2
    for (int co = 0; co < 32; ++co) {
        float v[kC1fQ];
        float s1 = 0.f;
#pragma unroll
        for (int q = 0; q < kC1fQ; ++q) {
            v[q] = bias[co];
#pragma unroll
            for (int k = 0; k < 15; ++k) v[q] = fmaf(w[co * 15 + k], win[q][k], v[q]);
            if (valid[q]) {
                out[((size_t)b * 32 + co) * Lout + p0 + tid + 256 * q] = v[q];
                s1 += v[q];
            }
        }
        if (spart) {
            s1 = wave_sum(s1);
            const float mean = nv > 0 ? s1 / (float)nv : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int q = 0; q < kC1fQ; ++q) {
                const float d = valid[q] ? v[q] - mean : 0.f;
                m2 = fmaf(d, d, m2);
            }
            m2 = wave_sum(m2);
            if ((tid & 63) == 0) {
                float* sp = spart + (((size_t)b * 32 + co) * np + blockIdx.x * 4 + wv) * 3;
                sp[0] = (float)nv; sp[1] = mean; sp[2] = m2;
            }
        }
    }
}

hipError_t launch_conv1_fwd(const float* wav, const float* w, const float* bias, float* out, float* out_stats, float* spart, int B, int Lin,
                            int Lout, int pad, hipStream_t st) {
    if (out_stats && !spart) return hipErrorInvalidValue;
    dim3 grid((Lout + kC1fP - 1) / kC1fP, B);
    hipLaunchKernelGGL(k_conv1_fwd, grid, dim3(256), 0, st, wav, w, bias, out, out_stats ? spart : nullptr, Lin, Lout, pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !out_stats) return e;
    return launch_stats_merge(spart, out_stats, B * 32, (int)grid.x * 4, st);
}

}  // namespace ls

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the stride-6 conv layers as an implicit GEMM (training step):
//     dW[co][ci][k] = sum_{b,p} dC[b][co][p] * act(in[b][ci][6p + k])
// MFMA M axis = 16 output channels, N axis = 16 weight columns (ci*15 + k), K = 4 positions per MFMA (pos = 4m + g).
// Workgroup = 64 output channels x 240 columns (16 input channels) x a run of 64-position tiles of the flattened
// (sample, tile) sequence (768 runs = 3 resident workgroups per CU whatever the batch); per tile it stages dC [64][64] and the
// activation window [16][393] in LDS.  Wave w owns column tiles w, w+4, w+8, w+12 (15 tiles).
// Round 3 (conv2 / conv3 / conv4 at B = 512: 457 / 339 / 303 us -> 460 / 325 / 236, tools/conv_bwd_bench.cpp):
//  * conv4 runs here too (rounds 1-2: im2col + GEMM); steps past the valid positions of a tile are skipped (its 34 positions
//    cost 9 steps, not 16);
//  * both operands are read as ONE ds_read_b128 per four MFMA steps (before: one ds_read_b32 per operand and step):
//      dcs  [co][16 chunks of 4]: chunk (4M + g) ^ (co & 15) of row co holds positions 16M + 4e + g, e = 0..3 -- lane (co, g)'s A
//           values of steps m = 4M .. 4M+3; the XOR makes the four 16-lane groups of a b128 read conflict-free;
//      acts [ci][33 rows][16]: row r = 6g + k (0..32) holds act[24m + r] for m = 0..15, i.e. lane (ci, k, g)'s B values of all 16
//           steps in step order (rows 24..32 repeat rows 0..8 one step later); chunk M is stored at M ^ ((r >> 1) & 3);
//  * the next tile's values stay RAW in registers while the current tile is multiplied and are normalised (InstanceNorm +
//    LeakyReLU, audio_enc.py:10-11) on their way into LDS (before: at fetch time, an s_waitcnt on every load in front of the MFMAs);
//    loads go through per-tile buffer descriptors (the window's tail past the sample reads 0; dC past the last position is zeroed,
//    which makes every mask on the window side unnecessary: a finite value times an exact 0).
// What bounds it (ablations of tools/build_conv_bench.sh, conv2): MFMAs + operand reads + barriers alone 337 us against a 279 us
// matrix floor (16 column tiles for 15), + the window / dC write phase 56, + the global loads 35..90.  The fp32 matrix pipe and the
// fp32 VALU are the same lanes, so the ~1 staging instruction per MFMA is paid in full; a 512-thread form with every LDS image
// multi-buffered, the dC tile by LDS-DMA and the staging interleaved into the MFMA groups (one barrier per tile) measured SLOWER
// (491 / 327 / 266): one workgroup per CU leaves nothing to fill its barrier skew and operand-read bursts.
// Partial sums per workgroup run go to a workspace that k_partial_reduce sums in index order (deterministic).
#ifndef LS_WG_ABL
#define LS_WG_ABL 0                      // timing-only ablations for tools/conv_bwd_bench.cpp: 1 no LDS write phase after the first tile,
#endif                                   // 2 no MFMAs, 8 no global loads after the first tile
namespace ls {

constexpr int kWgCo = 64, kWgCi = 16, kWgPT = 64;                          // 64 output channels x 16 input channels (240 columns)
constexpr int kWgWin = (kWgPT - 1) * 6 + 15;                               // 393
constexpr int kWgRS = 16, kWgCS = 34 * kWgRS;                              // 33 rows of 16 steps per channel (+ one row: bank shift, dummy slot)
constexpr int kWgNA = 28;                                                  // window values per thread: 4 channels x 7 pieces of 64

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))      // 3 workgroups per CU: 50.8 KB of LDS each, <= 168 VGPRs
void k_conv_wgrad(const float* __restrict__ dc, long long sb, long long sc, const float* __restrict__ in, const float* __restrict__ stats,
                  float* __restrict__ partial, int Cin, int Cout, int Lin, int Lout, int ntl, int ntot, int tpg) {
    __shared__ __attribute__((aligned(16))) float dcs[kWgCo * kWgPT];
    __shared__ __attribute__((aligned(16))) float acts[kWgCi * kWgCS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int ci0 = blockIdx.x * kWgCi, co0 = blockIdx.y * kWgCo;
    const int t0 = blockIdx.z * tpg, t1 = min(ntot, t0 + tpg);
    const int W = Cin * 15;

    f4 acc[4][4];                                   // [co tile][column tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = (f4){0.f, 0.f, 0.f, 0.f};
    // operand read offsets in bytes; step group M is reached by XOR (the swizzles live in address bits the rest leaves clear)
    int boff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = 16 * (w + 4 * c) + s16;       // column inside the 240-column chunk (tile 15, w = 3 / c = 3, is computed and dropped)
        const bool real = j < 240;
        const int cij = real ? j / 15 : 0, kj = real ? j - 15 * (j / 15) : 0;
        const int r = 6 * g + kj;
        boff[c] = (cij * kWgCS + r * kWgRS) * 4 + (((r >> 1) & 3) << 4);
    }
    const int aoff = s16 * kWgPT * 4 + (((s16 & 12) | ((g ^ s16) & 3)) << 4);
    // window staging: wave w owns channels 4w .. 4w+3, 7 pieces of 64 values each; value x -> row x % 24, step x / 24 (and, for rows
    // 0..8, a second copy in row 24 + x % 24 one step earlier); values without a slot go to the spare row (no branch).  Both byte
    // addresses of a piece share one register (16 bits each).
    unsigned wad[7];
#pragma unroll
    for (int pc = 0; pc < 7; ++pc) {
        const int x = 64 * pc + lane, m = x / 24, r = x - 24 * m;
        const int base = 4 * w * kWgCS, dummy = base + 33 * kWgRS + (lane & 15);
        const int wa = (x < kWgWin && m < 16) ? base + r * kWgRS + ((((m >> 2) ^ (r >> 1)) & 3) << 2) + (m & 3) : dummy;
        const int r2 = r + 24, m2 = m - 1;
        const int wd = (x < kWgWin && r < 9 && m >= 1) ? base + r2 * kWgRS + ((((m2 >> 2) ^ (r2 >> 1)) & 3) << 2) + (m2 & 3) : dummy;
        wad[pc] = (unsigned)(wa * 4) | ((unsigned)(wd * 4) << 16);
    }
    static_assert(kWgCi * kWgCS * 4 < 65536, "window addresses are packed in 16 bits");
    // dC staging: thread = (channel w + 4q, position `lane`) of the [64][64] tile
    const int dwo = (w * kWgPT + (((((lane >> 4) << 2) | (lane & 3)) ^ w) << 2) + ((lane >> 2) & 3)) * 4;   // row w; row w + 4q: ^ ((q & 3) << 6), + q KB

    constexpr int ND = kWgCo * kWgPT / 256;
    float rd[ND], ra[kWgNA];
    float fr[4], fn[4];                              // rstd and -mean * rstd of the fetched tile's four channels (wave-uniform)
    int fp0 = 0;
    auto fetch = [&](int t) {
        const int b = t / ntl, p0 = (t - b * ntl) * kWgPT;
        const auto rdc = uniform_rsrc(dc + (size_t)b * sb + (size_t)co0 * sc);
        const int vo = min(p0 + lane, Lout - 1) * 4;                            // clamped position: branch-free load
#pragma unroll
        for (int q = 0; q < ND; ++q) rd[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdc, vo, (w + 4 * q) * (int)sc * 4, 0));
        fp0 = p0;
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const size_t row = (size_t)b * Cin + ci0 + 4 * w + cl;
            const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
            fr[cl] = rstd; fn[cl] = -mean * rstd;
            const auto rin = uniform_rsrc(in + row * Lin + p0 * 6, (Lin - p0 * 6) * 4);    // past the sample's end: reads 0
#pragma unroll
            for (int pc = 0; pc < 7; ++pc) ra[cl * 7 + pc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, lane * 4 + 256 * pc, 0, 0));
        }
    };
    if (t0 < t1) fetch(t0);
    for (int t = t0; t < t1; ++t) {
        __syncthreads();
        if (!(LS_WG_ABL & 1) || t == t0) {
            const bool inside = fp0 + lane < Lout;
#pragma unroll
            for (int q = 0; q < ND; ++q)
                *reinterpret_cast<float*>(reinterpret_cast<char*>(dcs) + ((dwo ^ ((q & 3) << 6)) + q * 4 * kWgPT * 4)) = inside ? rd[q] : 0.f;
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                char* ab = reinterpret_cast<char*>(acts) + cl * kWgCS * 4;
#pragma unroll
                for (int pc = 0; pc < 7; ++pc) {
                    float v = fmaf(ra[cl * 7 + pc], fr[cl], fn[cl]);     // InstanceNorm1d + LeakyReLU(0.3), audio_enc.py:10-11
                    v = fmaxf(v, 0.3f * v);
                    *reinterpret_cast<float*>(ab + (wad[pc] & 0xffffu)) = v;
                    *reinterpret_cast<float*>(ab + (wad[pc] >> 16)) = v;
                }
            }
        }
        __syncthreads();
        const int mcnt = (min(kWgPT, Lout - (t % ntl) * kWgPT) + 3) >> 2;            // steps that hold data (uniform), >= 1
        if (t + 1 < t1 && !(LS_WG_ABL & 8)) fetch(t + 1);
        __builtin_amdgcn_sched_barrier(0);                           // the loads are issued before the first MFMA, not sunk next to their use
        const char* ab = reinterpret_cast<const char*>(dcs);
        const char* bb = reinterpret_cast<const char*>(acts);
        // groups of four steps; not unrolled: with all 16 steps in one block the scheduler hoists every operand read to the top and
        // spills the staging registers (scratch reloads then queue behind the 44 loads in flight)
#pragma unroll 1
        for (int M = 0; M < ((LS_WG_ABL & 2) ? 0 : (mcnt >> 2)); ++M) {
            f4 A[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) A[i] = *reinterpret_cast<const f4*>(ab + ((aoff ^ (M << 6)) + i * 16 * kWgPT * 4));
            f4 Bv = *reinterpret_cast<const f4*>(bb + (boff[0] ^ (M << 4)));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f4 Bn = Bv;
                if (c < 3) Bn = *reinterpret_cast<const f4*>(bb + (boff[c + 1] ^ (M << 4)));    // one column tile ahead of the MFMAs that use it
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][c] = MFMA(A[i][e], Bv[e], acc[i][c]);
                Bv = Bn;
            }
        }
        if ((mcnt & 3) && !(LS_WG_ABL & 2)) {                        // the tile's last, partly filled group (uniform)
            const int M = mcnt >> 2, ne = mcnt & 3;
            f4 A[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) A[i] = *reinterpret_cast<const f4*>(ab + ((aoff ^ (M << 6)) + i * 16 * kWgPT * 4));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f4 Bv = *reinterpret_cast<const f4*>(bb + (boff[c] ^ (M << 4)));
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    if (e >= ne) break;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][c] = MFMA(A[i][e], Bv[e], acc[i][c]);
                }
            }
        }
    }
    // lane (column s16 of tile, g) holds output channels co0 + 16 i + 4 g + e
    float* pz = partial + (size_t)blockIdx.z * Cout * W;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (16 * (w + 4 * c) >= 240) continue;
        const int J = ci0 * 15 + 16 * (w + 4 * c) + s16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) pz[(size_t)(co0 + 16 * i + 4 * g + e) * W + J] = acc[i][c][e];
    }
}

// dC element (b, co, p) at dc[b*sb + co*sc + p]; partial needs *ngroups * Cout * Cin * 15 floats, *ngroups <= conv_wgrad_groups()
int conv_wgrad_groups(int Cin, int Cout) {                                     // 768 = 3 resident workgroups per CU
    const int n = 768 / ((Cin / kWgCi) * (Cout / kWgCo));
    return n < 1 ? 1 : n;
}

hipError_t launch_conv_wgrad(const float* dc, long long sb, long long sc, const float* in, const float* stats, float* partial, int B, int Cin,
                             int Cout, int Lin, int Lout, int* ngroups, hipStream_t st) {
    if (Cin % kWgCi || Cout % kWgCo || !stats || B < 1) return hipErrorInvalidValue;
    const int ntl = (Lout + kWgPT - 1) / kWgPT, ntot = B * ntl;
    const int tpg = (ntot + conv_wgrad_groups(Cin, Cout) - 1) / conv_wgrad_groups(Cin, Cout);
    const int nz = (ntot + tpg - 1) / tpg;
    *ngroups = nz;
    hipLaunchKernelGGL(k_conv_wgrad, dim3(Cin / kWgCi, Cout / kWgCo, nz), dim3(256), 0, st, dc, sb, sc, in, stats, partial, Cin, Cout, Lin, Lout,
                       ntl, ntot, tpg);
    return hipGetLastError();
}

}  // namespace ls

// ---------------------------------------------------------------------------------------------------------------------
// Data gradient of the stride-6 conv layers as an implicit GEMM, fused with LeakyReLU' and the first half of the
// InstanceNorm backward (training step).  With x = 6q + r:
//     dAct[b][ci][x] = sum_co sum_{t : r + 6t <= 14} W[co][ci][r + 6t] * dC[b][co][q - t]
// i.e. six phase GEMMs (r = 0..5) with 3,3,3,2,2,2 taps -> 15 (r,t) pairs, no wasted multiplies.  MFMA M axis = 16 input
// channels, N axis = 16 values of q, K = 4 output channels per MFMA for one (r,t).  Workgroup = (sample, 64 q = 384 x,
// 32 input channels); wave w owns channel tile w&1 and q tiles 2(w>>1), 2(w>>1)+1 with one accumulator per phase.
// dC [64 co][66 q incl. halo] is staged in LDS per 64-output-channel chunk; the weight operand comes from a per-lane
// image [ci tile][co group of 4][pair][lane] rebuilt after every optimiser step (k_build_dgrad_img).
// Epilogue: y = InstanceNorm(c_raw), dy = dAct * lrelu'(y) is written to dc, and per-(sample, channel) partial sums of
// dy and dy*y go to `partial`; k_in_finalize turns dy into d c_raw = rstd * (dy - mean(dy) - y * mean(dy*y)).
#ifndef LS_DG_ABL
#define LS_DG_ABL 0                      // timing-only (tools/conv_bwd_bench.cpp): 16 no waveform gathers, 32 two S1 products instead of 48
#endif
namespace ls {

constexpr int kDgQT = 64, kDgDld = kDgQT + 4, kDgCo = 64;
constexpr int kDgEpS = 6 * kDgQT + 4;            // row stride of the epilogue tile [32][6*64] in LDS

__global__ void k_build_dgrad_img(const float* __restrict__ w, float* __restrict__ img, int Cin, int Cout) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(Cin / 16) * (Cout / 4) * 16 * 64;
    if (i >= total) return;
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63), pq = (int)((i >> 8) & 3);
    size_t rr = i >> 10;
    const int cog = (int)(rr % (Cout / 4)), cit = (int)(rr / (Cout / 4));
    const int pair = 4 * pq + e;
    float v = 0.f;
    if (pair < 15) {
        const int t = pair < 12 ? pair / 6 : 2, r = pair < 12 ? pair % 6 : pair - 12;
        const int co = 4 * cog + (lane >> 4), ci = 16 * cit + (lane & 15);
        v = w[((size_t)co * Cin + ci) * 15 + r + 6 * t];
    }
    img[i] = v;
}

// Both operand images (forward / weight-gradient layout of ls_api.cpp, data-gradient layout above) of the three stride-6 layers in ONE
// launch (the training step rebuilt them in six: 30 us of launch latency per step)
struct ConvImgJobs { const float* w[3]; float* img[3]; float* dimg[3]; int Cin[3], Cout[3]; int first[7]; };   // first[j]: first block of job j
__global__ __launch_bounds__(256) void k_build_conv_imgs(const ConvImgJobs a) {
    int j = 0;
#pragma unroll
    for (int q = 1; q < 6; ++q) j += (int)blockIdx.x >= a.first[q];
    const int L = j >> 1, Cin = a.Cin[L], Cout = a.Cout[L];
    const size_t i = (size_t)((int)blockIdx.x - a.first[j]) * 256 + threadIdx.x;
    const float* w = a.w[L];
    if (!(j & 1)) {                                                 // forward image: [co tile][chunk][k][lane][cig]
        if (i >= (size_t)Cout * Cin * 15) return;
        const int cig = (int)(i & 3), lane = (int)((i >> 2) & 63);
        size_t r = i >> 8;
        const int k = (int)(r % 15); r /= 15;
        const int nchunk = Cin / 16;
        const int ch = (int)(r % nchunk), ct = (int)(r / nchunk);
        a.img[L][i] = w[((size_t)(16 * ct + (lane & 15)) * Cin + 16 * ch + 4 * cig + (lane >> 4)) * 15 + k];
    } else {                                                        // data-gradient image (k_build_dgrad_img)
        if (i >= (size_t)(Cin / 16) * (Cout / 4) * 16 * 64) return;
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63), pq = (int)((i >> 8) & 3);
        size_t rr = i >> 10;
        const int cog = (int)(rr % (Cout / 4)), cit = (int)(rr / (Cout / 4));
        const int pair = 4 * pq + e;
        float v = 0.f;
        if (pair < 15) {
            const int t = pair < 12 ? pair / 6 : 2, r = pair < 12 ? pair % 6 : pair - 12;
            v = w[((size_t)(4 * cog + (lane >> 4)) * Cin + 16 * cit + (lane & 15)) * 15 + r + 6 * t];
        }
        a.dimg[L][i] = v;
    }
}

hipError_t launch_build_conv_imgs(const float* const w[3], float* const img[3], float* const dimg[3], const int Cin[3], const int Cout[3], hipStream_t st) {
    ConvImgJobs a;
    int nb = 0;
    for (int L = 0; L < 3; ++L) {
        a.w[L] = w[L]; a.img[L] = img[L]; a.dimg[L] = dimg[L]; a.Cin[L] = Cin[L]; a.Cout[L] = Cout[L];
        a.first[2 * L] = nb;     nb += (int)(((size_t)Cout[L] * Cin[L] * 15 + 255) / 256);
        a.first[2 * L + 1] = nb; nb += (int)(((size_t)(Cin[L] / 16) * (Cout[L] / 4) * 16 * 64 + 255) / 256);
    }
    a.first[6] = nb;
    hipLaunchKernelGGL(k_build_conv_imgs, dim3(nb), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_build_dgrad_img(const float* w, float* img, int Cin, int Cout, hipStream_t st) {
    const size_t total = (size_t)(Cin / 16) * (Cout / 4) * 16 * 64;
    hipLaunchKernelGGL(k_build_dgrad_img, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, img, Cin, Cout);
    return hipGetLastError();
}

// dC element (b, co, p) at dc_in[b*sb + co*sc + p*sp]
// FUSE1 (the layer above conv1: Cin == 32, so one workgroup holds all 32 channels of its 384 positions): dy is not written at all.
// Its only consumer is conv1's weight gradient  dW1[c][k] = sum_{b,x} d c1[b][c][x] wav[b][5x + k - pad],  d c1 = a dy + b' c_raw + c'
// per (b, c) row (k_in_bwd_coef), i.e.  a S1 + b' S3 + c' S2  with  S1 = sum_x dy wav,  S2 = sum_x wav,  S3 = sum_x c_raw wav -- and
// S2, S3 do not depend on the backward pass (S3 = bias S2 + w1 . R with R the 15 x 15 autocorrelation of the strided waveform,
// k_wav_moments).  So the epilogue writes dy back into its LDS tile and multiplies it with the waveform on the matrix pipe
// (M = 16 channels, N = the 15 taps, K = 4 positions; 48 MFMAs per wave on top of the main loop's 480): s1part[b][tile][x half][32][16].
// Saves the 517 MB dy write here and conv1's weight-gradient kernel, which read dy and c_raw again (1.03 GB).
// WIDE: the tile shape for SHORT layers (round 3): 64 input channels x 48 q per workgroup, wave = one channel tile x three q tiles (45 MFMAs
// per weight fragment instead of 30, every wave its own fragments).  conv4's 37 q fill 37 of 48 tile positions instead of 37 of 64, conv3's
// 219 fill 219 of 240 instead of 219 of 256.  The epilogue handles the 64 channels as two halves of 32 through the same LDS tile.
template <bool FUSE1, bool WIDE>
__global__ __launch_bounds__(256) void k_conv_dgrad(const float* __restrict__ dc_in, long long sb, long long sc, long long sp,
                                                    const float* __restrict__ wimg, const float* __restrict__ craw, const float* __restrict__ stats,
                                                    float* __restrict__ dc_out, float* __restrict__ partial, int Cin, int Cout, int Lx, int Lout,
                                                    const float* __restrict__ wav, int Lw, int wpad, float* __restrict__ s1part) {
    static_assert(!(FUSE1 && WIDE), "the fused form is conv2's: 32 input channels");
    constexpr int NCT = WIDE ? 4 : 2, NQW = WIDE ? 3 : 2, QT = WIDE ? 48 : kDgQT, EPS = 6 * QT + 4;   // channel tiles, q tiles per wave, q per workgroup
    __shared__ __attribute__((aligned(16))) float smem[32 * EPS > kDgCo * kDgDld ? 32 * EPS : kDgCo * kDgDld];
    float* dcs = smem;                 // main loop: dC [64 co][QT + 2 q + pad]; epilogue: dAct tile [32 ci][6 QT x + pad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, ci0 = blockIdx.y * 16 * NCT, q0 = blockIdx.x * QT;
    const int cit = WIDE ? w : (w & 1), qh = WIDE ? 0 : (w >> 1);

    f4 acc[NQW][6];
#pragma unroll
    for (int qt = 0; qt < NQW; ++qt)
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[qt][r] = (f4){0.f, 0.f, 0.f, 0.f};

    const auto wrs = uniform_rsrc(wimg);                           // weight image through a buffer descriptor (ls_lanes.h)
    const int wp = (((int)blockIdx.y * NCT + cit) * (Cout / 4)) * 4 * 1024;
    const int bbase = g * kDgDld + 32 * qh + s16 + 2;             // + 4*cogl*Dld*... see below: row = 4*cogl + g

    for (int cc = 0; cc < Cout / kDgCo; ++cc) {
        __syncthreads();
        {
            // all of this thread's loads first (clamped addresses: branch-free, in flight together), then the LDS writes
            constexpr int kN = kDgCo * (QT + 2), kPer = (kN + 255) / 256;
            float v[kPer];
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int idx = min(tid + 256 * i, kN - 1);
                const int co = idx / (QT + 2), jj = idx - co * (QT + 2);
                const int pc = min(max(q0 - 2 + jj, 0), Lout - 1);
                v[i] = dc_in[(size_t)b * sb + (size_t)(cc * kDgCo + co) * sc + (size_t)pc * sp];
            }
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int idx = tid + 256 * i;
                const int co = idx / (QT + 2), jj = idx - co * (QT + 2);
                const int p = q0 - 2 + jj;
                if (idx < kN) dcs[co * kDgDld + jj] = (p >= 0 && p < Lout) ? v[i] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int cogl = 0; cogl < kDgCo / 4; ++cogl) {
            const int wc = wp + (cc * (kDgCo / 4) + cogl) * 4 * 1024;
            f4 A[4];
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) A[pq] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, wc + pq * 1024, 0));
            const float* br = dcs + (4 * cogl) * kDgDld + bbase;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float Bq[NQW];
#pragma unroll
                for (int qt = 0; qt < NQW; ++qt) Bq[qt] = br[16 * qt - t];
#pragma unroll
                for (int r = 0; r < (t < 2 ? 6 : 3); ++r) {
                    const int pair = t < 2 ? t * 6 + r : 12 + r;
#pragma unroll
                    for (int qt = 0; qt < NQW; ++qt) acc[qt][r] = MFMA(A[pair >> 2][pair & 3], Bq[qt], acc[qt][r]);
                }
            }
        }
    }
    // Epilogue.  A lane's accumulators are 6 consecutive x of 4 channels (x = 6 q + r), 24 B apart from the next lane's: written
    // straight to HBM that is 48 scattered dword stores per lane (rocprofv3: 953 MB written for a 517 MB tensor, 35 % MFMA
    // busy, 62 % of wave time in s_waitcnt).  The tile goes through LDS instead ([32 channels][384 x], reusing the operand
    // buffer) and every wave then streams whole channel rows: coalesced c_raw loads and dy stores, one (sum dy, sum dy*y) pair
    // per row and workgroup.
    float* ep = smem;
    const int x0 = 6 * q0, nx = min(6 * QT, Lx - x0);                      // valid x of this tile (>= 1)
    const int nslot = gridDim.x * 2;
    constexpr int NK = (6 * QT + 63) / 64;                                 // 64-wide pieces of a tile row
#pragma unroll
    for (int half = 0; half < (WIDE ? 2 : 1); ++half) {
        __syncthreads();                                                   // the operand buffer / the previous half's tile is free
        if (!WIDE || (cit >> 1) == half) {
#pragma unroll
            for (int qt = 0; qt < NQW; ++qt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* dst = ep + (16 * (WIDE ? (cit & 1) : cit) + 4 * g + e) * EPS + 6 * (32 * qh + 16 * qt + s16);
#pragma unroll
                    for (int r2 = 0; r2 < 3; ++r2) *reinterpret_cast<float2*>(dst + 2 * r2) = make_float2(acc[qt][2 * r2][e], acc[qt][2 * r2 + 1][e]);
                }
        }
        __syncthreads();
        float cv[8][NK];                                                   // the accumulators are dead: their registers hold the c_raw tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* cr = craw + ((size_t)b * Cin + ci0 + 32 * half + w + 4 * i) * Lx + x0;
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) cv[i][kq] = cr[min(lane + 64 * kq, nx - 1)];   // clamped: branch-free loads in flight together
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = w + 4 * i;
            const size_t row = (size_t)b * Cin + ci0 + 32 * half + c;
            const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
            float* dst = dc_out + row * Lx + x0;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) {
                const int xl = lane + 64 * kq;
                const bool in_tile = (6 * QT) % 64 == 0 || xl < 6 * QT;    // the last piece of a 288-wide row is partial
                const float y = (cv[i][kq] - mean) * rstd;
                const float av = in_tile ? ep[c * EPS + min(xl, 6 * QT - 1)] : 0.f;
                const float dy = y >= 0.f ? av : 0.3f * av;
                if (xl < nx) {
                    if (!FUSE1) dst[xl] = dy;
                    s1 += dy;
                    s2 += dy * y;
                }
                if (FUSE1 && !(LS_DG_ABL & 64)) ep[c * EPS + xl] = xl < nx ? dy : 0.f;          // dy replaces dAct in the tile (own element)
            }
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) {                                                 // slot layout kept for the consumers: 2 per position tile
                float* pp = partial + (row * nslot + blockIdx.x * 2) * 2;
                pp[0] = s1; pp[1] = s2; pp[2] = 0.f; pp[3] = 0.f;
            }
        }
    }
    if (FUSE1 && !(LS_DG_ABL & 128)) {
        // S1[c][k] over this tile: wave = (channel tile w & 1, x half w >> 1); step m multiplies x = 192 (w >> 1) + 4 m + g
        const int mt = w & 1, kh = w >> 1;
        float Bw[48];
        const auto rw = uniform_rsrc(wav + (size_t)b * Lw, Lw * 4);      // past the end: the conv's zero padding (range check)
        const int xb = x0 + 192 * kh + g;
        const int i0 = 5 * xb + s16 - wpad;                              // + 20 m
        if (5 * x0 >= wpad) {                                            // (uniform) no index of this tile is negative
#pragma unroll
            for (int m = 0; m < 48; ++m) Bw[m] = (LS_DG_ABL & 16) ? 1.f + m : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, (i0 + 20 * m) * 4, 0, 0));
        } else {
            // the left padding.  NOT through the range check: the compiler folds the 80 m into the instruction's immediate offset, and the
            // hardware does not wrap a negative (= huge unsigned) register offset + immediate back into range -- elements whose own index
            // was valid came back as 0
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                const int i = i0 + 20 * m;
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, max(i, 0) * 4, 0, 0));
                Bw[m] = i >= 0 ? v : 0.f;
            }
        }
        __syncthreads();                                                 // every row of the tile holds dy now
        const float* ar = ep + (16 * mt + s16) * EPS + 192 * kh + g;
        f4 a1 = (f4){0.f, 0.f, 0.f, 0.f}, a2 = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < ((LS_DG_ABL & 32) ? 2 : 48); m += 2) {
            a1 = MFMA(ar[4 * m], Bw[m], a1);
            a2 = MFMA(ar[4 * m + 4], Bw[m + 1], a2);
        }
        a1 += a2;
        // lane (k = s16, g) holds channels 16 mt + 4 g + e
        float* o = s1part + ((((size_t)b * gridDim.x + blockIdx.x) * 2 + kh) * 32 + 16 * mt + 4 * g) * 16 + s16;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[16 * e] = a1[e];
    }
}

// S2[k] = sum_p wavpad[5p + k] and R[j][k] = sum_p wavpad[5p + j] wavpad[5p + k] over the Lout positions of conv1 (zero padding `pad`)
// of every sample: mom[b][part][16][16] with R in [j][k], j, k < 15, and S2[j] in column 15, summed over the gridDim.x parts by the
// consumer.  Workgroup = (part of the positions, sample); the 15 x 16 products are ONE MFMA per four positions (the operand with the
// ones column is both A and B); the operand comes out of a coalesced LDS copy of the workgroup's stretch of the waveform (one
// workgroup per sample with a 64-address gather in front of every MFMA took 160 us at B = 512).
constexpr int kMomWin = 5200;                             // window floats per workgroup: 5 * 4 * (steps per workgroup) + 15 <= kMomWin
// parts per sample: 8, or as many as it takes for a part's stretch of the waveform to fit the window (longer audio than the reference's)
int wav_moment_parts(int Lout) {
    int parts = 8;
    while (80 * (((Lout + 3) / 4 + 4 * parts - 1) / (4 * parts)) + 16 > kMomWin) ++parts;
    return parts;
}
__global__ __launch_bounds__(256) void k_wav_moments(const float* __restrict__ wav, float* __restrict__ mom, int Lw, int Lout, int pad) {
    __shared__ float wv[kMomWin];
    __shared__ float red[4][16][16];
    const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int nparts = gridDim.x;
    const int nstep = (Lout + 3) / 4, per = (nstep + 4 * nparts - 1) / (4 * nparts);   // steps of 4 positions per wave
    // the workgroup's stretch of the (zero-padded, Lout-limited) waveform, coalesced: positions [4 * 4 * part * per, + 16 * per)
    const int pa = 16 * part * per, x0 = 5 * pa - pad, xend = min(5 * (Lout - 1) + 15 - pad, Lw);   // valid wav indices: [max(x0,0), xend)
    const float* wb = wav + (size_t)b * Lw;
    {
        constexpr int NQ = (kMomWin + 255) / 256;                           // all loads first (a rolled loop waits for each in turn)
        float t[NQ];
#pragma unroll
        for (int q = 0; q < ((LS_DG_ABL & 512) ? 1 : NQ); ++q) {
            const int x = x0 + tid + 256 * q;
            const float v = wb[min(max(x, 0), Lw - 1)];
            t[q] = (x >= 0 && x < xend) ? v : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) if (tid + 256 * q < 80 * per + 16) wv[tid + 256 * q] = t[q];
    }
    __syncthreads();
    // (positions >= Lout: their taps beyond the last valid position's window are zeroed above only partly -- mask them out here)
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    const int ma = (part * 4 + w) * per;
    const float* lw_ = wv + 20 * (w * per) + 5 * g + s16;                      // + 20 per step
    for (int q0 = 0; q0 < ((LS_DG_ABL & 256) ? 8 : per); q0 += 8) {                                      // eight operand reads, then eight MFMAs on two accumulators
        float v[8], one[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + u, p = 4 * (ma + q) + g;
            const bool ok = q < per && ma + q < nstep && p < Lout;
            v[u] = ok ? lw_[20 * min(q, per - 1)] : 0.f;
            one[u] = ok ? 1.f : 0.f;                                           // B's column 15 is all ones: R[j][15] = S2[j]
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc = MFMA(s16 == 15 ? 0.f : v[u], s16 == 15 ? one[u] : v[u], acc);
            acc2 = MFMA(s16 == 15 ? 0.f : v[u + 1], s16 == 15 ? one[u + 1] : v[u + 1], acc2);
        }
    }
    acc += acc2;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[w][4 * g + e][s16] = acc[e];
    __syncthreads();
    const int j = tid >> 4, k = tid & 15;
    mom[((size_t)b * nparts + part) * 256 + tid] = ((red[0][j][k] + red[1][j][k]) + red[2][j][k]) + red[3][j][k];
}

// out[b][c*15 + k] = a S1 + b' S3 + c' S2 (see k_conv_dgrad<FUSE1>); a following k_partial_reduce over b gives dW1
__global__ __launch_bounds__(512) void k_conv1_wgrad_finish(const float* __restrict__ s1part, int nt2, const float* __restrict__ coef,
                                                            const float* __restrict__ mom, int nparts, const float* __restrict__ w1,
                                                            const float* __restrict__ bias1, float* __restrict__ out) {
    __shared__ float msum[256];
    const int b = blockIdx.x, c = threadIdx.x >> 4, k = threadIdx.x & 15;
    if (threadIdx.x < 256) {                                               // the sample's moments: parts summed in index order
        const float* mb = mom + (size_t)b * nparts * 256 + threadIdx.x;
        float a = 0.f;
#pragma unroll 8
        for (int q = 0; q < nparts; ++q) a += mb[q * 256];
        msum[threadIdx.x] = a;
    }
    const float* sp = s1part + (size_t)b * nt2 * 512 + threadIdx.x;
    float s1 = 0.f;
#pragma unroll 6
    for (int i = 0; i < nt2; ++i) s1 += sp[(size_t)i * 512];              // fixed order
    __syncthreads();
    if (k == 15) return;
    const float s2 = msum[k * 16 + 15];
    float s3 = bias1[c] * s2;
#pragma unroll
    for (int j = 0; j < 15; ++j) s3 = fmaf(w1[c * 15 + j], msum[j * 16 + k], s3);
    const float* cf = coef + ((size_t)b * 32 + c) * 4;
    out[(size_t)b * 480 + c * 15 + k] = cf[0] * s1 + cf[1] * s3 + cf[2] * s2;
}

hipError_t launch_wav_moments(const float* wav, float* mom, int B, int Lw, int Lout, int pad, hipStream_t st) {
    hipLaunchKernelGGL(k_wav_moments, dim3(wav_moment_parts(Lout), B), dim3(256), 0, st, wav, mom, Lw, Lout, pad);
    return hipGetLastError();
}

// one workgroup per (sample, channel) row: d c_raw = rstd * (dy - mean(dy) - y * mean(dy * y)), in place on dc
__global__ __launch_bounds__(256) void k_in_finalize(float* __restrict__ dc, const float* __restrict__ craw, const float* __restrict__ stats,
                                                     const float* __restrict__ partial, int nslot, int L) {
    const size_t row = blockIdx.x;
    float a = 0.f, c2 = 0.f;
    for (int i = 0; i < nslot; ++i) {                                  // fixed order, every thread the same
        a += partial[(row * nslot + i) * 2];
        c2 += partial[(row * nslot + i) * 2 + 1];
    }
    const float m1 = a / (float)L, m2 = c2 / (float)L;
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    const float* cr = craw + row * L;
    float* dr = dc + row * L;
    for (int x = threadIdx.x; x < L; x += 256) {
        const float y = (cr[x] - mean) * rstd;
        dr[x] = rstd * (dr[x] - m1 - y * m2);
    }
}

hipError_t launch_conv_dgrad(const float* dc_in, long long sb, long long sc, long long sp, const float* wimg, const float* craw,
                             const float* stats, float* dc_out, float* partial, int B, int Cin, int Cout, int Lx, int Lout, bool finalize,
                             int* nslot, hipStream_t st) {
    if (Cin % 32 || Cout % kDgCo) return hipErrorInvalidValue;
    const int nq = (Lx + 5) / 6;
    // tile shape: 64 channels x 48 q where that wastes fewer tile positions than 32 x 64 (short layers: conv4's 37 q, conv3's 219)
    const int t64 = (nq + 63) / 64 * 64, t48 = (nq + 47) / 48 * 48;
    const bool wide = Cin % 64 == 0 && t48 < t64;
    dim3 grid(wide ? t48 / 48 : t64 / 64, wide ? Cin / 64 : Cin / 32, B);
    if (nslot) *nslot = (int)grid.x * 2;
    if (wide)
        hipLaunchKernelGGL((k_conv_dgrad<false, true>), grid, dim3(256), 0, st, dc_in, sb, sc, sp, wimg, craw, stats, dc_out, partial, Cin, Cout, Lx, Lout,
                           (const float*)nullptr, 0, 0, (float*)nullptr);
    else
        hipLaunchKernelGGL((k_conv_dgrad<false, false>), grid, dim3(256), 0, st, dc_in, sb, sc, sp, wimg, craw, stats, dc_out, partial, Cin, Cout, Lx, Lout,
                           (const float*)nullptr, 0, 0, (float*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !finalize) return e;
    hipLaunchKernelGGL(k_in_finalize, dim3(B * Cin), dim3(256), 0, st, dc_out, craw, stats, partial, (int)grid.x * 2, Lx);
    return hipGetLastError();
}

// The layer above conv1 (Cin == 32) with conv1's weight gradient folded in: no dy tensor.  rowpart [B*32][*nslot][2] as
// launch_conv_dgrad(finalize = false) leaves it; work needs B * (2 * tiles * 512 + 128 + 480) floats; mom from launch_wav_moments;
// out_part[B][480] is summed over B by the caller (launch_partial_reduce).
hipError_t launch_conv_dgrad_conv1(const float* dc_in, long long sb, long long sc, long long sp, const float* wimg, const float* craw,
                                   const float* stats, float* rowpart, int B, int Cout, int Lx, int Lout, const float* wav, int Lw, int wpad,
                                   const float* mom, const float* w1, const float* bias1, float* work, float** out_part, hipStream_t st) {
    if (Cout % kDgCo) return hipErrorInvalidValue;
    const int nq = (Lx + 5) / 6;
    dim3 grid((nq + kDgQT - 1) / kDgQT, 1, B);
    const int nslot = (int)grid.x * 2, nt2 = (int)grid.x * 2;
    float* s1part = work;
    float* coef = s1part + (size_t)B * nt2 * 512;
    float* outp = coef + (size_t)B * 128;
    hipLaunchKernelGGL((k_conv_dgrad<true, false>), grid, dim3(256), 0, st, dc_in, sb, sc, sp, wimg, craw, stats, (float*)nullptr, rowpart, 32, Cout, Lx, Lout,
                       wav, Lw, wpad, s1part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if ((e = launch_in_bwd_coef(stats, rowpart, nslot, B * 32, Lx, coef, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv1_wgrad_finish, dim3(B), dim3(512), 0, st, s1part, nt2, coef, mom, wav_moment_parts(Lx), w1, bias1, outp);
    *out_part = outp;
    return hipGetLastError();
}

}  // namespace ls
