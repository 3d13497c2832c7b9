// Generic fp32 MFMA GEMM for the once-per-call linears and the SAG decoder:
//     C[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] ) (+ R[m][n])          (nn.Linear: y = x W^T + b)
// v_mfma_f32_16x16x4_f32 in the same transposed form as the step kernel (features on the MFMA M axis, rows of A on
// N), so a lane's 4 accumulator registers are 4 consecutive features of one row and the store is a float4.
// 128x128 tile per 256-thread workgroup, 4 waves as 2x2 of 64x64, K staged through LDS in chunks of 32 with row
// stride 36 floats (conflict-free ds_read_b128); the next chunk is fetched into registers while the current one is
// multiplied.  Arbitrary M, N, K (edges are zero-filled / masked).
#include "ls_internal.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kGBM = 128, kGBN = 128, kGBK = 32, kGLd = kGBK + 4;

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__global__ __launch_bounds__(256, 3) void k_gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                    const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                                    float* __restrict__ C, int ldc, int M, int N, int K, int act) {
    __shared__ __attribute__((aligned(16))) float sA[kGBM * kGLd];
    __shared__ __attribute__((aligned(16))) float sW[kGBN * kGLd];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * kGBM, n0 = blockIdx.x * kGBN;
    const int s16 = lane & 15, g = lane >> 4;

    f4 acc[4][4];                                   // [feature tile][row tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const bool vec_ok = ((lda | ldw) & 3) == 0 && (((size_t)A | (size_t)W) & 15) == 0;
    // this thread's share of a 128 x 32 tile of A and of W: 4 float4 each (row idx>>3, k offset (idx&7)*4)
    f4 ra[4], rw[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 3, k = k0 + (idx & 7) * 4;
            ra[i] = (f4){0.f, 0.f, 0.f, 0.f};
            rw[i] = ra[i];
            if (m0 + r < M) {
                const float* p = A + (size_t)(m0 + r) * lda + k;
                if (vec_ok && k + 3 < K) ra[i] = *reinterpret_cast<const f4*>(p);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < K) ra[i][e] = p[e];
                }
            }
            if (n0 + r < N) {
                const float* p = W + (size_t)(n0 + r) * ldw + k;
                if (vec_ok && k + 3 < K) rw[i] = *reinterpret_cast<const f4*>(p);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < K) rw[i][e] = p[e];
                }
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += kGBK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<f4*>(&sA[r * kGLd + c4]) = ra[i];
            *reinterpret_cast<f4*>(&sW[r * kGLd + c4]) = rw[i];
        }
        __syncthreads();
        if (k0 + kGBK < K) fetch(k0 + kGBK);        // next tile's global loads overlap this tile's MFMAs
#pragma unroll
        for (int kk = 0; kk < kGBK / 16; ++kk) {
            f4 af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const f4*>(&sA[(wm * 64 + 16 * j + s16) * kGLd + 16 * kk + 4 * g]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f4 wf = *reinterpret_cast<const f4*>(&sW[(wn * 64 + 16 * i + s16) * kGLd + 16 * kk + 4 * g]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(wf[e], af[j][e], acc[i][j]);
            }
        }
    }
    // epilogue: lane (row = s16 of row tile j, g) holds features n0 + wn*64 + 16*i + 4*g + {0..3} -> one float4 per (i, j)
    const bool cvec = ((ldc | (R ? ldr : 0)) & 3) == 0 && (((size_t)C | (size_t)R | (size_t)bias) & 15) == 0;
    auto activate = [&](float v) {
        if (act == 1) return v / (1.0f + expf(-v));
        if (act == 2) return expf(0.5f * v);
        if (act == 3) return gelu_exact(v);
        return v;
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + 16 * j + s16;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + 16 * i + 4 * g;
            if (n >= N) continue;
            if (cvec && n + 3 < N) {
                f4 v = acc[i][j];
                if (bias) v += *reinterpret_cast<const f4*>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = activate(v[e]);
                if (R) v += *reinterpret_cast<const f4*>(R + (size_t)m * ldr + n);
                *reinterpret_cast<f4*>(C + (size_t)m * ldc + n) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < N) {
                        float v = activate(acc[i][j][e] + (bias ? bias[n + e] : 0.f));
                        if (R) v += R[(size_t)m * ldr + n + e];
                        C[(size_t)m * ldc + n + e] = v;
                    }
                }
            }
        }
    }
}

// act: 0 none, 1 SiLU, 2 exp(0.5 y), 3 exact GELU (F.gelu default, nn.TransformerDecoderLayer activation="gelu")
hipError_t launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr,
                          float* C, int ldc, int M, int N, int K, int act, hipStream_t st) {
    dim3 grid((N + kGBN - 1) / kGBN, (M + kGBM - 1) / kGBM);
    hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, st, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, act);
    return hipGetLastError();
}

}  // namespace ls
