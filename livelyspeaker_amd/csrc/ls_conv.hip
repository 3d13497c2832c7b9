// WavEncoder Conv1d layers 2-4 (scripts/model/audio_enc.py:12-19) as implicit GEMM on v_mfma_f32_16x16x4_f32:
//     out[b][co][p] = bias[co] + sum_{ci,k} W[co][ci][k] * act(in[b][ci][p*6 + k])
// with act = previous layer's InstanceNorm1d + LeakyReLU(0.3) applied while the input window is staged in LDS.
// MFMA M axis = 16 output channels, N axis = 16 output positions, K = (k, ci) in k-MAJOR order inside each chunk of
// 16 input channels: the four K entries of one MFMA are 4 consecutive input channels at the same tap k, so lane
// (position p, g) reads lds[(ci0+g)][p*6 + k] = lane base + compile-time offset (no per-lane div/mod), and the weight
// operand W[co][ci0+g][k] is pre-permuted on the host into per-lane order (one float4 = 4 consecutive MFMA steps).
// Workgroup = 64 positions x 64 output channels of one sample: wave w owns 16 positions and 4 channel tiles.
#include "ls_internal.h"

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
        for (int idx = tid; idx < kCvCI * kCvWin; idx += 256) {
            const int ci = idx / kCvWin, o = idx - ci * kCvWin;
            const int gi = in0 + o;
            float v = 0.f;
            if (gi < Lin) {
                const size_t row = (size_t)b * Cin + c * kCvCI + ci;
                v = in[row * Lin + gi];
                const float m = stats[row * 2], r = stats[row * 2 + 1];     // InstanceNorm1d + LeakyReLU(0.3), audio_enc.py:10-11
                v = (v - m) * r;
                v = v >= 0.f ? v : 0.3f * v;
            }
            sIn[ci * kCvWinP + o] = v;
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
