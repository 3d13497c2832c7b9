// WavEncoder Conv1d layers 2-4 (scripts/model/audio_enc.py:12-19) as implicit GEMM on v_mfma_f32_16x16x4_f32:
//     out[b][co][p] = bias[co] + sum_{ci,k} W[co][ci][k] * act(in[b][ci][p*6 + k])
// with act = previous layer's InstanceNorm1d + LeakyReLU(0.3) applied while the input window is staged in LDS.
// MFMA M axis = 16 output channels, N axis = 16 output positions, K = (k, ci) in k-MAJOR order inside each chunk of
// 16 input channels: the four K entries of one MFMA are 4 consecutive input channels at the same tap k, so lane
// (position p, g) reads lds[(ci0+g)][p*6 + k] = lane base + compile-time offset (no per-lane div/mod), and the weight
// operand W[co][ci0+g][k] is pre-permuted on the host into per-lane order (one float4 = 4 consecutive MFMA steps).
// Workgroup = 64 positions x 64 output channels of one sample: wave w owns 16 positions and 4 channel tiles.
#include "ls_internal.h"
#include "ls_train.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kCvK = 15, kCvS = 6, kCvTP = 64, kCvCI = 16, kCvTC = 64;
constexpr int kCvWin = (kCvTP - 1) * kCvS + kCvK;      // 393 input samples per channel per tile
constexpr int kCvWinP = kCvWin + 4;                    // 397: odd stride -> the 4 lane groups (channels) hit different banks
constexpr int kCvSteps = kCvK * kCvCI / 4;             // 60 MFMA k-steps per chunk

// wimg: [co tile (Cout/16)][chunk (Cin/16)][step4 (15)][lane 64][4]: element e of step4 q is MFMA step s = 4q+e,
//       tap k = s / 4 ... see build_conv_image() in ls_api.cpp: step s -> (k = s / 4, cig = s % 4), ci = 4*cig + g
__global__ __launch_bounds__(256) void k_conv1d_mfma(const float* __restrict__ in, const float* __restrict__ stats,
                                                     const float* __restrict__ wimg, const float* __restrict__ bias,
                                                     float* __restrict__ out, int Cin, int Cout, int Lin, int Lout) {
    __shared__ float sIn[kCvCI * kCvWinP];
    const int b = blockIdx.z, co0 = blockIdx.y * kCvTC, p0 = blockIdx.x * kCvTP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int nchunk = Cin / kCvCI;
    const int in0 = p0 * kCvS;

    f4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f4){0.f, 0.f, 0.f, 0.f};
    const int lbase = g * kCvWinP + (16 * w + s16) * kCvS;     // + (4*cig)*WinP + k per step

    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();
        {   // 16 threads per input channel; clamped addresses keep the 25 loads branch-free and in flight together
            const int ci = tid >> 4;
            const size_t row = (size_t)b * Cin + c * kCvCI + ci;
            const float m = stats[row * 2], r = stats[row * 2 + 1];         // InstanceNorm1d + LeakyReLU(0.3), audio_enc.py:10-11
            const float* src = in + row * Lin + in0;
            const int valid = Lin - in0;                                    // >= 1 for every tile that has an output position
            float vals[(kCvWin + 15) / 16];
#pragma unroll
            for (int q = 0; q < (kCvWin + 15) / 16; ++q) vals[q] = src[min((tid & 15) + 16 * q, valid - 1)];
#pragma unroll
            for (int q = 0; q < (kCvWin + 15) / 16; ++q) {
                const int o = (tid & 15) + 16 * q;
                float v = (vals[q] - m) * r;
                v = v >= 0.f ? v : 0.3f * v;
                if (o < kCvWin) sIn[ci * kCvWinP + o] = o < valid ? v : 0.f;
            }
        }
        __syncthreads();
        const f4* wp = reinterpret_cast<const f4*>(wimg) + ((size_t)(blockIdx.y * 4) * nchunk + c) * kCvK * 64 + lane;
#pragma unroll
        for (int k = 0; k < kCvK; ++k) {
            f4 A[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) A[t] = wp[((size_t)t * nchunk * kCvK + k) * 64];
#pragma unroll
            for (int cig = 0; cig < 4; ++cig) {
                const float Bv = sIn[lbase + (4 * cig) * kCvWinP + k];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = MFMA(A[t][cig], Bv, acc[t]);
            }
        }
    }
    const int p = p0 + 16 * w + s16;
    if (p < Lout) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + 16 * t + 4 * g + j;
                if (co < Cout) out[((size_t)b * Cout + co) * Lout + p] = acc[t][j] + bias[co];
            }
    }
}

hipError_t launch_conv1d_mfma(const float* in, const float* stats, const float* wimg, const float* bias, float* out, int B,
                              int Cin, int Cout, int Lin, int Lout, hipStream_t st) {
    if (Cin % kCvCI || Cout % kCvTC || !stats) return hipErrorInvalidValue;
    dim3 grid((Lout + kCvTP - 1) / kCvTP, Cout / kCvTC, B);
    hipLaunchKernelGGL(k_conv1d_mfma, grid, dim3(256), 0, st, in, stats, wimg, bias, out, Cin, Cout, Lin, Lout);
    return hipGetLastError();
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

constexpr int kWgCo = 64, kWgCi = 16, kWgCols = kWgCi * 15, kWgPT = 64, kWgDld = kWgPT + 4;
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
