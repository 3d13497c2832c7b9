// Strided fp32 MFMA GEMM for the training step (forward, data-gradient and weight-gradient products):
//     C[m][n] (+)= act( sum_k A(m,k) * B(n,k) + bias[n] )
// A(m,k) and B(n,k) are addressed through two-level strides on both the row and the reduction index
//     off(m,k) = (m / ri) * ro + (m % ri) * rs  +  (k / ki) * ko + (k % ki) * ks
// so nn.Linear forward (A = X[m][k], B = W[n][k]), dX = dY W (B = W read down its columns), dW = dY^T X (both operands
// read down their columns, reduction over the batch rows) and the [B][C][L] conv tensors all run on the same kernel
// without materialised transposes.  The template flag says which index is contiguous in memory for each operand, which
// decides how threads are laid over the tile when it is staged (coalesced global reads) and how it sits in LDS:
//   K-contiguous operand: LDS [row][k] (stride 36), MFMA operand = one ds_read_b128;
//   row-contiguous operand: LDS [k][row] (stride 132), staged with float4 writes, MFMA operand = 4 ds_read_b32.
// v_mfma_f32_16x16x4_f32, 128x128 tile / 256 threads, K chunk 32, same transposed accumulator form as ls_gemm.hip.
// Split-K (gridDim.z > 1) writes partial tiles to a workspace which k_splitk_reduce sums in a fixed order
// (deterministic: no float atomics anywhere in the training step).
#include "ls_internal.h"
#include "ls_train.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kTM = 128, kTK = 32, kLdK = kTK + 4, kLdR = kTM + 4;

__device__ __forceinline__ size_t op_off(const GemmOperand& o, int r, int k) {
    return (size_t)(r / o.ri) * o.ro + (size_t)(r % o.ri) * o.rs + (size_t)(k / o.ki) * o.ko + (size_t)(k % o.ki) * o.ks;
}

// Stage a 128 x 32 tile of one operand.  KC: threads adjacent along k (operand is k-contiguous) -> LDS [row][k];
// otherwise threads adjacent along rows -> LDS [k][row].
template <bool KC>
__device__ __forceinline__ void stage(float* s, const GemmOperand& o, int r0, int R, int k0, int k1, int tid) {
    if (KC) {
        const bool vec = o.vec;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            f4 v = (f4){0.f, 0.f, 0.f, 0.f};
            const int k = k0 + c4;
            if (r0 + r < R) {
                if (vec && k + 3 < k1) v = *reinterpret_cast<const f4*>(o.p + op_off(o, r0 + r, k));
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < k1) v[e] = o.p[op_off(o, r0 + r, k + e)];
                }
            }
            *reinterpret_cast<f4*>(&s[r * kLdK + c4]) = v;
        }
    } else {
        const bool vec = o.vec;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int kk = idx >> 5, r4 = (idx & 31) * 4;
            f4 v = (f4){0.f, 0.f, 0.f, 0.f};
            const int k = k0 + kk;
            if (k < k1) {
                if (vec && r0 + r4 + 3 < R) v = *reinterpret_cast<const f4*>(o.p + op_off(o, r0 + r4, k));
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (r0 + r4 + e < R) v[e] = o.p[op_off(o, r0 + r4 + e, k)];
                }
            }
            *reinterpret_cast<f4*>(&s[kk * kLdR + r4]) = v;
        }
    }
}

template <bool KC>
__device__ __forceinline__ f4 frag(const float* s, int row, int kk, int g) {
    if (KC) return *reinterpret_cast<const f4*>(&s[row * kLdK + 16 * kk + 4 * g]);
    f4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = s[(16 * kk + 4 * g + e) * kLdR + row];
    return v;
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void k_gemm_tr(const GemmArgs a) {
    constexpr int SA = AK ? kTM * kLdK : kTK * kLdR;
    constexpr int SB = BK ? kTM * kLdK : kTK * kLdR;
    __shared__ __attribute__((aligned(16))) float sA[SA];
    __shared__ __attribute__((aligned(16))) float sB[SB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.y * kTM, n0 = blockIdx.x * kTM;
    const int s16 = lane & 15, g = lane >> 4;
    const int z = blockIdx.z;
    const int kbeg = z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);

    f4 acc[4][4];                                   // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbeg; k0 < kend; k0 += kTK) {
        __syncthreads();
        stage<AK>(sA, a.A, m0, a.M, k0, kend, tid);
        stage<BK>(sB, a.B, n0, a.N, k0, kend, tid);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTK / 16; ++kk) {
            f4 bf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf[i] = frag<BK>(sB, wn * 64 + 16 * i + s16, kk, g);
                af[i] = frag<AK>(sA, wm * 64 + 16 * i + s16, kk, g);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(bf[i][e], af[j][e], acc[i][j]);
        }
    }
    // lane (m = s16 of m tile j, g) holds n = n0 + wn*64 + 16*i + 4*g + {0..3}
    const bool partial = gridDim.z > 1;
    float* Cz = partial ? a.ws + (size_t)z * a.M * a.N : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + 16 * j + s16;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + 16 * i + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= a.N) continue;
                float v = acc[i][j][e];
                if (partial) { Cz[(size_t)m * a.N + n + e] = v; continue; }
                if (a.bias) v += a.bias[n + e];
                const size_t co = (size_t)(m / a.cri) * a.cro + (size_t)(m % a.cri) * a.crs + (size_t)(n + e) * a.cns;
                if (a.Cpre) a.Cpre[co] = v;
                if (a.act == 1) v = v / (1.0f + expf(-v));
                if (a.R) v += a.R[co];
                if (a.accumulate) v += a.C[co];
                a.C[co] = v;
            }
        }
    }
}

// C[m][n] (+)= sum_z ws[z][m][n] (+ bias); fixed summation order
__global__ void k_splitk_reduce(const GemmArgs a, int Z) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)a.M * a.N) return;
    const int m = (int)(i / a.N), n = (int)(i % a.N);
    float v = 0.f;
    for (int z = 0; z < Z; ++z) v += a.ws[(size_t)z * a.M * a.N + i];
    if (a.bias) v += a.bias[n];
    const size_t co = (size_t)(m / a.cri) * a.cro + (size_t)(m % a.cri) * a.crs + (size_t)n * a.cns;
    if (a.accumulate) v += a.C[co];
    a.C[co] = v;
}

hipError_t launch_gemm_tr(GemmArgs a, bool a_kcontig, bool b_kcontig, int splits, hipStream_t st) {
    if (splits < 1) splits = 1;
    // chunk is a multiple of the K tile so that every split starts on a tile boundary
    int kchunk = ((a.K + splits - 1) / splits + kTK - 1) / kTK * kTK;
    splits = (a.K + kchunk - 1) / kchunk;
    a.kchunk = kchunk;
    if (splits > 1 && (!a.ws || (size_t)splits * a.M * a.N > a.ws_floats)) return hipErrorInvalidValue;
    if (splits > 1 && (a.Cpre || a.R || a.act)) return hipErrorInvalidValue;
    dim3 grid((a.N + kTM - 1) / kTM, (a.M + kTM - 1) / kTM, splits);
    if (a_kcontig && b_kcontig) hipLaunchKernelGGL((k_gemm_tr<true, true>), grid, dim3(256), 0, st, a);
    else if (a_kcontig && !b_kcontig) hipLaunchKernelGGL((k_gemm_tr<true, false>), grid, dim3(256), 0, st, a);
    else if (!a_kcontig && b_kcontig) hipLaunchKernelGGL((k_gemm_tr<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_gemm_tr<false, false>), grid, dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splits == 1) return e;
    const size_t n = (size_t)a.M * a.N;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, splits);
    return hipGetLastError();
}

}  // namespace ls
