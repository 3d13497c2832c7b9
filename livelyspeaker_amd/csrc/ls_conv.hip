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
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kCvK = 15, kCvS = 6, kCvTP = 64, kCvCI = 16, kCvTC = 64;

// Stage stamps for tools/conv_bench.cpp (built with -DLS_CONV_PROF; never in the shipped library): lane 0 of one consumer and one
// producer wave of ONE workgroup records the cycle counter when it reaches / leaves each per-stage barrier.
#ifdef LS_CONV_PROF
__device__ unsigned long long* g_conv_prof = nullptr;      // [2 roles][1024 stages][2]
__device__ int g_conv_prof_wg = 0;
#define CV_STAMP(role, sidx, which)                                                                                       \
    do {                                                                                                                  \
        if (g_conv_prof && (int)blockIdx.z == g_conv_prof_wg && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (sidx) < 1024) \
            g_conv_prof[(((role) * 1024) + (sidx)) * 2 + (which)] = __builtin_readcyclecounter();                          \
    } while (0)
#else
#define CV_STAMP(role, sidx, which) do { } while (0)
#endif
constexpr int kCvWin = (kCvTP - 1) * kCvS + kCvK;      // 393 input samples per channel per tile
constexpr int kCvWinP = kCvWin + 4;                    // 397: odd stride -> the 4 lane groups (channels) hit different banks

// wimg: [co tile (Cout/16)][chunk (Cin/16)][step4 (15)][lane 64][4]: element e of step4 q is MFMA step s = 4q+e,
//       tap k = s / 4 ... see build_conv_image() in ls_api.cpp: step s -> (k = s / 4, cig = s % 4), ci = 4*cig + g
// Producer / consumer workgroup (512 threads): waves 0-3 only multiply (wave w = channel tile w x the four position tiles:
// one float4 of weights per tap feeds 16 MFMAs; activations from LDS, weights from L2), waves 4-7 only stage -- they fetch the
// NEXT chunk's raw input window (25 values per thread, clamped addresses: branch-free and in flight together), apply the
// previous layer's InstanceNorm + LeakyReLU and write it to the other half of a double-buffered LDS window while the consumers
// run the current chunk's 240 MFMAs.  One workgroup barrier per chunk.  (As a single-role 256-thread kernel the memory side
// alone took 276 us and the MFMA side alone 336 us for conv2, and they overlapped poorly: 515 us; this form: 476 us.)
__global__ __launch_bounds__(512) void k_conv1d_mfma(const float* __restrict__ in, const float* __restrict__ stats,
                                                        const float* __restrict__ wimg, const float* __restrict__ bias,
                                                        float* __restrict__ out, float* __restrict__ spart, int Cin, int Cout, int Lin, int Lout,
                                                        int ntile, int tpw) {
    // A workgroup walks `tpw` consecutive 64-position tiles of one (sample, 64-channel group): the (tile, chunk) stages form ONE
    // pipeline, so the producers' first fetch and the consumers' epilogue of a tile overlap neighbouring stages.
    // Measured on MI355X at B = 512 (profiles/r02b): conv2 505 -> 499 us, conv3 347 -> 338 us with this and the weight ring below --
    // and 502 us with the producers additionally software-pipelined two stages deep (raw loads given two stage-times to land; not
    // kept).  So neither the per-tile prologue, nor the weight fetch, nor the activation fetch latency sets the 12 us stage time
    // (8 us of it is MFMA issue of the two co-resident workgroups).  Cutting the producers' arithmetic from 10 to 4 VALU operations
    // per element on interior tiles (no clamp, no tail mask, max() for the LeakyReLU) changed nothing either (496 us): the producer
    // side is not the limiter in any of its aspects; the consumers' loop -- one LDS operand read per MFMA, 50 % of the LDS pipe
    // with two workgroups per CU -- sets the pace at 59 % matrix-pipe occupancy.
    __shared__ float sIn[2][kCvCI * kCvWinP];
    const int b = blockIdx.z, co0 = blockIdx.y * kCvTC;
    const int t0 = blockIdx.x * tpw, t1 = min(ntile, t0 + tpw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = Cin / kCvCI;
    const int nstage = (t1 - t0) * nchunk;
    constexpr int NV = (kCvWin + 15) / 16;
    if (w >= 4) {
        // ---------------- producers
        const int pt = tid - 256;
        auto stage = [&](int sidx, float* dst) {
            const int tl = sidx / nchunk, c = sidx - tl * nchunk;
            const int in0 = (t0 + tl) * kCvTP * kCvS;
            const int validw = Lin - in0;                                   // >= 1 for every tile that has an output position
            const size_t row = (size_t)b * Cin + c * kCvCI + (pt >> 4);
            const float vm = stats[row * 2], vr = stats[row * 2 + 1];   // InstanceNorm1d + LeakyReLU(0.3), audio_enc.py:10-11
            const float* src = in + row * Lin + in0;
            float vals[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) vals[q] = src[min((pt & 15) + 16 * q, validw - 1)];      // clamped: branch-free, in flight together
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int o = (pt & 15) + 16 * q;
                float v = (vals[q] - vm) * vr;
                v = v >= 0.f ? v : 0.3f * v;
                if (o < kCvWin) dst[(pt >> 4) * kCvWinP + o] = o < validw ? v : 0.f;
            }
        };
        stage(0, sIn[0]);
        __syncthreads();
        for (int sidx = 0; sidx < nstage; ++sidx) {
            if (sidx + 1 < nstage) stage(sidx + 1, sIn[(sidx + 1) & 1]);
            if (w == 4) CV_STAMP(1, sidx, 0);
            __syncthreads();
            if (w == 4) CV_STAMP(1, sidx, 1);
        }
        return;
    }
    // ---------------- consumers
    const int s16 = lane & 15, g = lane >> 4;
    const int lbase = g * kCvWinP + s16 * kCvS;                // + (4*cig)*WinP + 16*pt*S + k per step
    const f4 bv = *reinterpret_cast<const f4*>(bias + co0 + 16 * w + 4 * g);
    constexpr int kWPre = 3, kWRing = 5;                       // taps in flight; ring size (15 taps per chunk = 3 turns of the ring:
    static_assert(kCvK % kWRing == 0 && kWPre < kWRing, "ring positions must line up across chunks");   // positions repeat per chunk)
    f4 ring[kWRing];
    // weight image through a buffer descriptor (ls_lanes.h: 3.7 instead of 16.8 matrix-pipe cycles per load): SGPR offset = this wave's
    // channel tile, + chunk, + tap; VGPR offset = lane * 16
    const auto wrs = uniform_rsrc(wimg);
    const int wbase = ((int)blockIdx.y * 4 + w) * nchunk * kCvK * 1024;
    auto wld = [&](int off) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, off, 0)); };
    __syncthreads();
    int sidx = 0;
    for (int tile = t0; tile < t1; ++tile) {
        const int p0 = tile * kCvTP;
        f4 acc[4];                                             // [position tile]
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunk; ++c, ++sidx) {
            const float* sb = sIn[sidx & 1];
            // weight operand: a ring of kWPre taps in flight (one float4 per tap = 16 MFMAs = 512 issue cycles; an L2 round trip under
            // load is longer than that, so a distance of one tap stalled every tap); the ring runs on into the next chunk's image
            const int wp = wbase + c * kCvK * 1024;
            const int wpn = wbase + ((c + 1 < nchunk) ? c + 1 : 0) * kCvK * 1024;            // next stage's chunk (same tile or the next)
            if (sidx == 0) {
#pragma unroll
                for (int k = 0; k < kWPre; ++k) ring[k] = wld(wp + k * 1024);
            }
#pragma unroll
            for (int k = 0; k < kCvK; ++k) {
                const f4 A = ring[k % kWRing];
                ring[(k + kWPre) % kWRing] = wld((k + kWPre < kCvK) ? wp + (k + kWPre) * 1024 : wpn + (k + kWPre - kCvK) * 1024);
#pragma unroll
                for (int cig = 0; cig < 4; ++cig) {
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) {
                        const float Bv = sb[lbase + (4 * cig) * kCvWinP + 16 * pt * kCvS + k];
                        acc[pt] = MFMA(A[cig], Bv, acc[pt]);
                    }
                }
            }
            if (w == 0) CV_STAMP(0, sidx, 0);
            __syncthreads();
            if (w == 0) CV_STAMP(0, sidx, 1);
        }
        // epilogue of this tile (registers -> global only): runs while the producers stage the next tile's second chunk
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int p = p0 + 16 * pt + s16;
            const bool valid = p < Lout;
            const int nv = min(16, max(0, Lout - (p0 + 16 * pt)));
            const float inv_nv = nv > 0 ? 1.0f / (float)nv : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + 16 * w + 4 * g + j;
                const float v = acc[pt][j] + bv[j];
                if (valid) out[((size_t)b * Cout + co) * Lout + p] = v;
                if (spart) {
                    const float s1 = row16_sum(valid ? v : 0.f);
                    const float mean = s1 * inv_nv;
                    const float d = valid ? v - mean : 0.f;
                    const float m2 = row16_sum(d * d);
                    if (s16 == 0) {
                        float* sp = spart + (((size_t)b * Cout + co) * (ntile * 4) + tile * 4 + pt) * 3;
                        sp[0] = (float)nv; sp[1] = mean; sp[2] = m2;
                    }
                }
            }
        }
    }
}

// The same implicit GEMM for a SHORT output (the last encoder layer: 34 positions from 217 inputs).  With one sample per workgroup
// the 34 positions fill 34 of the 64 columns of the tile above (conv4 ran at 0.36 of the MFMA peak, its matrix pipe as busy as
// conv3's for 63 % of the FLOPs -- profiles/r02a).  Here the N axis enumerates (sample, position) pairs of NS samples: NS = 4 gives
// 136 columns = 8.5 tiles of 16 -> 9 tiles, 94 % useful.  A lane's activation address is per-lane data anyway (base + compile-time
// tap / channel offsets), so columns that straddle two samples cost nothing.  Same producer / consumer split, same weight image.
template <int NS, int NPT>
__global__ __launch_bounds__(512) void k_conv1d_short(const float* __restrict__ in, const float* __restrict__ stats,
                                                         const float* __restrict__ wimg, const float* __restrict__ bias,
                                                         float* __restrict__ out, int Cin, int Cout, int Lin, int Lout, int B, int winp) {
    extern __shared__ float sIn[];                             // [2][NS][16][winp]
    const int b0 = blockIdx.z * NS, co0 = blockIdx.y * kCvTC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = Cin / kCvCI;
    const int wn = (Lout - 1) * kCvS + kCvK;                   // input samples a row of the window holds (<= Lin)
    const int bufsz = NS * kCvCI * winp;
    if (w >= 4) {
        // ---------------- producers: 16 threads per row, 16 rows per pass, NS passes per chunk
        const int pt = tid - 256;
        constexpr int NQ = 15;                                 // 16-wide groups of a window row: wn <= 35 * 6 + 15 = 225
        auto stage = [&](int c, float* dst) {
            float vals[NS][NQ], vm[NS], vr[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {                     // every load of the chunk first (clamped addresses: branch-free, in flight together)
                const int b = min(b0 + s, B - 1);                               // past the batch: staged but never stored
                const size_t row = (size_t)b * Cin + c * kCvCI + (pt >> 4);
                vm[s] = stats[row * 2]; vr[s] = stats[row * 2 + 1];             // InstanceNorm1d + LeakyReLU(0.3), audio_enc.py:10-11
                const float* src = in + row * Lin;
#pragma unroll
                for (int q = 0; q < NQ; ++q) vals[s][q] = src[min((pt & 15) + 16 * q, Lin - 1)];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float* d = dst + (s * kCvCI + (pt >> 4)) * winp;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int o = (pt & 15) + 16 * q;
                    float v = (vals[s][q] - vm[s]) * vr[s];
                    v = v >= 0.f ? v : 0.3f * v;
                    if (o < wn) d[o] = v;
                }
            }
        };
        stage(0, sIn);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk) stage(c + 1, sIn + ((c + 1) & 1) * bufsz);
            __syncthreads();
        }
        return;
    }
    // ---------------- consumers: wave w = channel tile w x all NPT column tiles
    const int s16 = lane & 15, g = lane >> 4;
    f4 acc[NPT];
    int lb[NPT];                                               // lane's window base per column tile
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
        const int j = min(16 * t + s16, NS * Lout - 1);
        const int s = j / Lout, p = j - s * Lout;
        lb[t] = (s * kCvCI + g) * winp + p * kCvS;
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
#pragma unroll
        for (int k = 0; k < kCvK; ++k) {
            const f4 A = ring[k % kWRing];
            ring[(k + kWPre) % kWRing] = wld((k + kWPre < kCvK) ? wp + (k + kWPre) * 1024 : wpn + (k + kWPre - kCvK) * 1024);
#pragma unroll
            for (int cig = 0; cig < 4; ++cig) {
#pragma unroll
                for (int t = 0; t < NPT; ++t) {
                    const float Bv = sb[lb[t] + (4 * cig) * winp + k];
                    acc[t] = MFMA(A[cig], Bv, acc[t]);
                }
            }
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
__global__ __launch_bounds__(256) void k_stats_merge(const float* __restrict__ spart, float* __restrict__ stats, int rows, int np) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);              // one wave per row
    if (row >= rows) return;
    const float* sp = spart + (size_t)row * np * 3;
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
    for (int i = lane; i < np; i += 64) merge(n, mean, m2, (double)sp[3 * i], (double)sp[3 * i + 1], (double)sp[3 * i + 2]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                                 // butterfly: every lane ends with the full merge
        const double nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), qb = __shfl_xor(m2, o);
        merge(n, mean, m2, nb, mb, qb);
    }
    if (lane == 0) {
        stats[(size_t)row * 2] = (float)mean;
        stats[(size_t)row * 2 + 1] = (float)(1.0 / sqrt(m2 / n + 1e-5));
    }
}

hipError_t launch_stats_merge(const float* spart, float* stats, int rows, int np, hipStream_t st) {
    hipLaunchKernelGGL(k_stats_merge, dim3((rows + 3) / 4), dim3(256), 0, st, spart, stats, rows, np);
    return hipGetLastError();
}

// out_stats != null: also produce the InstanceNorm statistics of the OUTPUT (fused, no second pass over it); spart is a
// workspace of B * Cout * ceil(Lout / 64) * 4 * 3 floats
hipError_t launch_conv1d_mfma(const float* in, const float* stats, const float* wimg, const float* bias, float* out, float* out_stats,
                              float* spart, int B, int Cin, int Cout, int Lin, int Lout, hipStream_t st) {
    if (Cin % kCvCI || Cout % kCvTC || !stats || (out_stats && !spart)) return hipErrorInvalidValue;
    if (!out_stats && Lout <= 36 && (Lout - 1) * kCvS + kCvK <= Lin) {
        // short output without statistics (conv4: 34 positions): columns = (sample, position) pairs of 4 samples, 9 tiles
        constexpr int NS = 4, NPT = 9;
        const int winp = ((Lout - 1) * kCvS + kCvK) | 1;                  // odd row stride: the 4 lane groups (channels) hit different banks
        const size_t lds = (size_t)2 * NS * kCvCI * winp * sizeof(float);
        static bool attr_set = false;                                     // > 64 KiB of dynamic LDS needs the opt-in, once per process
        if (!attr_set) {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv1d_short<NS, NPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (ea != hipSuccess) return ea;
            attr_set = true;
        }
        if (NS * Lout <= 16 * NPT && lds <= 160 * 1024) {
            hipLaunchKernelGGL((k_conv1d_short<NS, NPT>), dim3(1, Cout / kCvTC, (B + NS - 1) / NS), dim3(512), lds, st, in, stats, wimg, bias, out,
                               Cin, Cout, Lin, Lout, B, winp);
            return hipGetLastError();
        }
    }
    const int ntile = (Lout + kCvTP - 1) / kCvTP;
    // tiles per workgroup: the whole row when that still leaves >= 2 workgroups per CU; otherwise split rows until it does
    int tpw = ntile;
    while (tpw > 1 && (long long)((ntile + tpw - 1) / tpw) * (Cout / kCvTC) * B < 512) tpw = (tpw + 1) / 2;
    dim3 grid((ntile + tpw - 1) / tpw, Cout / kCvTC, B);
    hipLaunchKernelGGL(k_conv1d_mfma, grid, dim3(512), 0, st, in, stats, wimg, bias, out, out_stats ? spart : nullptr, Cin, Cout, Lin, Lout, ntile, tpw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !out_stats) return e;
    return launch_stats_merge(spart, out_stats, B * Cout, ntile * 4, st);
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
    for (int i = tid; i < kC1fWin; i += 256) {
        const int x = x0 + i;
        const float v = wav[(size_t)b * Lin + min(max(x, 0), Lin - 1)];
        sw[i] = (x >= 0 && x < Lin) ? v : 0.f;
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
// Workgroup = 64 output channels x 240 columns (16 input channels) x a group of samples; per 64-position tile it stages
// dC [64][64] and the activation window [16][398] (InstanceNorm + LeakyReLU applied on the way in) in LDS; lane (j, g)
// reads act_lds[ci_j][k_j + 6g + 24m] = lane base + immediate.  Wave w owns column tiles w, w+4, w+8, w+12 (15 tiles).
// Partial sums per sample group go to a workspace that k_partial_reduce sums in index order (deterministic).
namespace ls {

constexpr int kWgCo = 64, kWgCi = 16, kWgPT = 64, kWgDld = kWgPT + 4;      // 64 output channels x 16 input channels (240 columns)
constexpr int kWgWin = (kWgPT - 1) * 6 + 15, kWgWinP = kWgWin + 1;      // 393 -> 394

__global__ __launch_bounds__(256) void k_conv_wgrad(const float* __restrict__ dc, const float* __restrict__ in, const float* __restrict__ stats,
                                                    float* __restrict__ partial, int Cin, int Cout, int Lin, int Lout, int spw, int B) {
    __shared__ float dcs[kWgCo * kWgDld];
    __shared__ float acts[kWgCi * kWgWinP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int ci0 = blockIdx.x * kWgCi, co0 = blockIdx.y * kWgCo;
    const int b0 = blockIdx.z * spw, b1 = min(B, b0 + spw);
    const int W = Cin * 15;
    const int ntile = w == 3 ? 3 : 4;               // column tiles of this wave

    f4 acc[4][4];                                   // [co tile][column tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = (f4){0.f, 0.f, 0.f, 0.f};
    int bbase[4];                                   // lane's activation base per column tile: ci_j * WinP + k_j + 6 g
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = 16 * (w + 4 * c) + s16;       // column inside the 240-column chunk (tile 15 of wave 3 is unused)
        const int cij = j / 15, kj = j - 15 * cij;
        bbase[c] = (c < ntile ? cij : 0) * kWgWinP + kj + 6 * g;
    }
    const int abase = s16 * kWgDld + g;             // + 16 i * Dld + 4 m

    // Software pipeline over (sample, 64-position tile): the next tile's global loads (16 dC + 25 activation values per
    // thread, InstanceNorm + LeakyReLU applied in registers) are in flight while the current tile is multiplied.
    constexpr int ND = kWgCo * kWgPT / 256, NA = (kWgWin + 15) / 16;
    float rd[ND], ra[NA];
    const int ntl = (Lout + kWgPT - 1) / kWgPT, ntot = (b1 - b0) * ntl;
    auto fetch = [&](int t) {
        const int b = b0 + t / ntl, p0 = (t % ntl) * kWgPT;
#pragma unroll
        for (int q = 0; q < ND; ++q) {
            const int idx = tid + 256 * q, co = idx >> 6, p = idx & 63;
            const float v = dc[((size_t)b * Cout + co0 + co) * Lout + min(p0 + p, Lout - 1)];    // clamped address: branch-free load
            rd[q] = p0 + p < Lout ? v : 0.f;
        }
        const int ci = tid >> 4;
        const size_t row = (size_t)b * Cin + ci0 + ci;
        const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
        const float* src = in + row * Lin + p0 * 6;
        const int valid = Lin - p0 * 6;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int o = (tid & 15) + 16 * q;
            float v = (src[min(o, valid - 1)] - mean) * rstd;       // clamped address keeps the 25 loads branch-free and in flight together
            v = v >= 0.f ? v : 0.3f * v;
            ra[q] = o < valid ? v : 0.f;
        }
    };
    if (ntot > 0) fetch(0);
    for (int t = 0; t < ntot; ++t) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ND; ++q) {
            const int idx = tid + 256 * q;
            dcs[(idx >> 6) * kWgDld + (idx & 63)] = rd[q];
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int o = (tid & 15) + 16 * q;
            if (o < kWgWin) acts[(tid >> 4) * kWgWinP + o] = ra[q];
        }
        __syncthreads();
        if (t + 1 < ntot) fetch(t + 1);
#pragma unroll 4
        for (int m = 0; m < kWgPT / 4; ++m) {
            float A[4], Bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) A[i] = dcs[abase + 16 * i * kWgDld + 4 * m];
#pragma unroll
            for (int c = 0; c < 4; ++c) Bv[c] = acts[bbase[c] + 24 * m];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i][c] = MFMA(A[i], Bv[c], acc[i][c]);
        }
    }
    // lane (column s16 of tile, g) holds output channels co0 + 16 i + 4 g + e
    float* pz = partial + (size_t)blockIdx.z * Cout * W;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= ntile) continue;
        const int J = ci0 * 15 + 16 * (w + 4 * c) + s16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) pz[(size_t)(co0 + 16 * i + 4 * g + e) * W + J] = acc[i][c][e];
    }
}

hipError_t launch_conv_wgrad(const float* dc, const float* in, const float* stats, float* partial, int B, int Cin, int Cout, int Lin, int Lout,
                             int spw, int* ngroups, hipStream_t st) {
    if (Cin % kWgCi || Cout % kWgCo || !stats || spw < 1) return hipErrorInvalidValue;
    const int nz = (B + spw - 1) / spw;
    *ngroups = nz;
    hipLaunchKernelGGL(k_conv_wgrad, dim3(Cin / kWgCi, Cout / kWgCo, nz), dim3(256), 0, st, dc, in, stats, partial, Cin, Cout, Lin, Lout, spw, B);
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

hipError_t launch_build_dgrad_img(const float* w, float* img, int Cin, int Cout, hipStream_t st) {
    const size_t total = (size_t)(Cin / 16) * (Cout / 4) * 16 * 64;
    hipLaunchKernelGGL(k_build_dgrad_img, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, img, Cin, Cout);
    return hipGetLastError();
}

// dC element (b, co, p) at dc_in[b*sb + co*sc + p*sp]
__global__ __launch_bounds__(256) void k_conv_dgrad(const float* __restrict__ dc_in, long long sb, long long sc, long long sp,
                                                    const float* __restrict__ wimg, const float* __restrict__ craw, const float* __restrict__ stats,
                                                    float* __restrict__ dc_out, float* __restrict__ partial, int Cin, int Cout, int Lx, int Lout) {
    __shared__ __attribute__((aligned(16))) float smem[32 * kDgEpS > kDgCo * kDgDld ? 32 * kDgEpS : kDgCo * kDgDld];
    float* dcs = smem;                 // main loop: dC [64 co][66 q + pad]; epilogue: dAct tile [32 ci][384 x + pad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, ci0 = blockIdx.y * 32, q0 = blockIdx.x * kDgQT;
    const int cit = w & 1, qh = w >> 1;

    f4 acc[2][6];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[qt][r] = (f4){0.f, 0.f, 0.f, 0.f};

    const auto wrs = uniform_rsrc(wimg);                           // weight image through a buffer descriptor (ls_lanes.h)
    const int wp = (((int)blockIdx.y * 2 + cit) * (Cout / 4)) * 4 * 1024;
    const int bbase = g * kDgDld + 32 * qh + s16 + 2;             // + 4*cogl*Dld*... see below: row = 4*cogl + g

    for (int cc = 0; cc < Cout / kDgCo; ++cc) {
        __syncthreads();
        {
            // all of this thread's loads first (clamped addresses: branch-free, in flight together), then the LDS writes
            constexpr int kN = kDgCo * (kDgQT + 2), kPer = (kN + 255) / 256;
            float v[kPer];
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int idx = min(tid + 256 * i, kN - 1);
                const int co = idx / (kDgQT + 2), jj = idx - co * (kDgQT + 2);
                const int pc = min(max(q0 - 2 + jj, 0), Lout - 1);
                v[i] = dc_in[(size_t)b * sb + (size_t)(cc * kDgCo + co) * sc + (size_t)pc * sp];
            }
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const int idx = tid + 256 * i;
                const int co = idx / (kDgQT + 2), jj = idx - co * (kDgQT + 2);
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
                const float B0 = br[-t], B1 = br[16 - t];
#pragma unroll
                for (int r = 0; r < (t < 2 ? 6 : 3); ++r) {
                    const int pair = t < 2 ? t * 6 + r : 12 + r;
                    acc[0][r] = MFMA(A[pair >> 2][pair & 3], B0, acc[0][r]);
                    acc[1][r] = MFMA(A[pair >> 2][pair & 3], B1, acc[1][r]);
                }
            }
        }
    }
    // Epilogue.  A lane's accumulators are 6 consecutive x of 4 channels (x = 6 q + r), 24 B apart from the next lane's: written
    // straight to HBM that is 48 scattered dword stores per lane (rocprofv3: 953 MB written for a 517 MB tensor, 35 % MFMA
    // busy, 62 % of wave time in s_waitcnt).  The tile goes through LDS instead ([32 channels][384 x], reusing the operand
    // buffer) and every wave then streams whole channel rows: coalesced c_raw loads and dy stores, one (sum dy, sum dy*y) pair
    // per row and workgroup.
    __syncthreads();
    float* ep = smem;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float* dst = ep + (16 * cit + 4 * g + e) * kDgEpS + 6 * (32 * qh + 16 * qt + s16);
#pragma unroll
            for (int r2 = 0; r2 < 3; ++r2) *reinterpret_cast<float2*>(dst + 2 * r2) = make_float2(acc[qt][2 * r2][e], acc[qt][2 * r2 + 1][e]);
        }
    __syncthreads();
    const int x0 = 6 * q0, nx = min(6 * kDgQT, Lx - x0);                  // valid x of this tile (>= 1)
    const int nslot = gridDim.x * 2;
    float cv[8][6];                                                      // the accumulators are dead: their registers hold the c_raw tile
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float* cr = craw + ((size_t)b * Cin + ci0 + w + 4 * i) * Lx + x0;
#pragma unroll
        for (int k = 0; k < 6; ++k) cv[i][k] = cr[min(lane + 64 * k, nx - 1)];   // clamped: 48 branch-free loads in flight together
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = w + 4 * i;
        const size_t row = (size_t)b * Cin + ci0 + c;
        const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
        float* dst = dc_out + row * Lx + x0;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int xl = lane + 64 * k;
            const float y = (cv[i][k] - mean) * rstd;
            const float av = ep[c * kDgEpS + xl];
            const float dy = y >= 0.f ? av : 0.3f * av;
            if (xl < nx) {
                dst[xl] = dy;
                s1 += dy;
                s2 += dy * y;
            }
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {                                                 // slot layout kept for the consumers: 2 per position tile
            float* pp = partial + (row * nslot + blockIdx.x * 2) * 2;
            pp[0] = s1; pp[1] = s2; pp[2] = 0.f; pp[3] = 0.f;
        }
    }
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
    dim3 grid((nq + kDgQT - 1) / kDgQT, Cin / 32, B);
    if (nslot) *nslot = (int)grid.x * 2;
    hipLaunchKernelGGL(k_conv_dgrad, grid, dim3(256), 0, st, dc_in, sb, sc, sp, wimg, craw, stats, dc_out, partial, Cin, Cout, Lx, Lout);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !finalize) return e;
    hipLaunchKernelGGL(k_in_finalize, dim3(B * Cin), dim3(256), 0, st, dc_out, craw, stats, partial, (int)grid.x * 2, Lx);
    return hipGetLastError();
}

}  // namespace ls
