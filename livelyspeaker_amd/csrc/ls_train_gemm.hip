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
// The next K tile is fetched into registers while the current one is multiplied (software pipelining).
// v_mfma_f32_16x16x4_f32, 128x128 tile / 256 threads, K chunk 32, same transposed accumulator form as ls_gemm.hip.
// Split-K (gridDim.z > 1) writes partial tiles to a workspace which k_splitk_reduce sums in a fixed order
// (deterministic: no float atomics anywhere in the training step).
#include "ls_internal.h"
#include "ls_train.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kTM = 128, kTK = 32, kLdK = kTK + 4, kLdR = kTM + 4;

// two-level index -> offset; the single-level case (inner extent INT_MAX) skips the integer division
__device__ __forceinline__ size_t lvl(int i, int inner, long long so, long long si) {
    if (inner == INT_MAX) return (size_t)i * si;
    return (size_t)(i / inner) * so + (size_t)(i % inner) * si;
}

// Fetch this thread's share of a 128 x 32 operand tile into registers (global loads only; they stay in flight while the
// previous tile is multiplied).  KC: threads adjacent along k (operand is k-contiguous); otherwise adjacent along rows.
// roff[]: the thread's row offsets, computed once per kernel.
template <bool KC>
__device__ __forceinline__ void fetch(f4 (&v)[4], const GemmOperand& o, const size_t (&roff)[4], int r0, int R, int k0, int k1, int tid) {
    if (KC) {
        const int c4 = (tid & 7) * 4, k = k0 + c4;
        size_t koff[4];
        const bool vec = o.vec && k + 3 < k1;
        if (vec) koff[0] = lvl(k, o.ki, o.ko, o.ks);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) koff[e] = lvl(k + e, o.ki, o.ko, o.ks);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 3) + 32 * i;
            v[i] = (f4){0.f, 0.f, 0.f, 0.f};
            if (r0 + r < R) {
                if (vec) v[i] = *reinterpret_cast<const f4*>(o.p + roff[i] + koff[0]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < k1) v[i][e] = o.p[roff[i] + koff[e]];
                }
            }
        }
    } else {
        const int r4 = (tid & 31) * 4;
        const bool vec = o.vec && r0 + r4 + 3 < R;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + (tid >> 5) + 8 * i;
            v[i] = (f4){0.f, 0.f, 0.f, 0.f};
            if (k < k1) {
                const size_t koff = lvl(k, o.ki, o.ko, o.ks);
                if (vec) v[i] = *reinterpret_cast<const f4*>(o.p + roff[0] + koff);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (r0 + r4 + e < R) v[i][e] = o.p[roff[e] + koff];
                }
            }
        }
    }
}

// registers -> LDS: K-contiguous operands as [row][k] (stride 36), row-contiguous ones as [k][row] (stride 132)
template <bool KC>
__device__ __forceinline__ void put(float* s, const f4 (&v)[4], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KC) *reinterpret_cast<f4*>(&s[((tid >> 3) + 32 * i) * kLdK + (tid & 7) * 4]) = v[i];
        else *reinterpret_cast<f4*>(&s[((tid >> 5) + 8 * i) * kLdR + (tid & 31) * 4]) = v[i];
    }
}

// per-thread row offsets of the tile rows this thread stages (KC: rows (tid>>3) + 32 i; else rows 4 (tid&31) + e)
template <bool KC>
__device__ __forceinline__ void row_offsets(const GemmOperand& o, size_t (&roff)[4], int r0, int R, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = r0 + (KC ? (tid >> 3) + 32 * i : (tid & 31) * 4 + i);
        if (r >= R) r = R - 1;
        roff[i] = lvl(r, o.ri, o.ro, o.rs);
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
__global__ __launch_bounds__(256, 3) void k_gemm_tr(const GemmArgs a) {
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

    size_t roffA[4], roffB[4];
    row_offsets<AK>(a.A, roffA, m0, a.M, tid);
    row_offsets<BK>(a.B, roffB, n0, a.N, tid);
    f4 ra[4], rb[4];
    fetch<AK>(ra, a.A, roffA, m0, a.M, kbeg, kend, tid);
    fetch<BK>(rb, a.B, roffB, n0, a.N, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += kTK) {
        __syncthreads();
        put<AK>(sA, ra, tid);
        put<BK>(sB, rb, tid);
        __syncthreads();
        if (k0 + kTK < kend) {              // next tile's global loads overlap this tile's MFMAs
            fetch<AK>(ra, a.A, roffA, m0, a.M, k0 + kTK, kend, tid);
            fetch<BK>(rb, a.B, roffB, n0, a.N, k0 + kTK, kend, tid);
        }
#pragma unroll
        for (int kk = 0; kk < kTK / 16; ++kk) {
            f4 af[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = frag<AK>(sA, wm * 64 + 16 * j + s16, kk, g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // one B fragment live at a time keeps the kernel at 3 waves / SIMD
                const f4 bf = frag<BK>(sB, wn * 64 + 16 * i + s16, kk, g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = MFMA(bf[e], af[j][e], acc[i][j]);
            }
        }
    }
    // lane (m = s16 of m tile j, g) holds n = n0 + wn*64 + 16*i + 4*g + {0..3}
    const bool partial = gridDim.z > 1;
    float* Cz = partial ? a.ws + (size_t)z * a.M * a.N : nullptr;
    const bool cvec = a.cns == 1 && (a.crs & 3) == 0 && (a.cri == INT_MAX || (a.cro & 3) == 0) && ((uintptr_t)a.C & 15) == 0 &&
                      (!a.Cpre || ((uintptr_t)a.Cpre & 15) == 0) && (!a.R || ((uintptr_t)a.R & 15) == 0) &&
                      (!a.bias || ((uintptr_t)a.bias & 15) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + 16 * j + s16;
        if (m >= a.M) continue;
        const size_t crow = partial ? (size_t)m * a.N : lvl(m, a.cri, a.cro, a.crs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + 16 * i + 4 * g;
            if (n >= a.N) continue;
            f4 v = acc[i][j];
            if (partial) {
                if ((a.N & 3) == 0) *reinterpret_cast<f4*>(&Cz[crow + n]) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < a.N) Cz[crow + n + e] = v[e];
                }
                continue;
            }
            if (cvec && n + 3 < a.N) {
                const size_t co = crow + n;
                if (a.bias) v += *reinterpret_cast<const f4*>(a.bias + n);
                if (a.Cpre) *reinterpret_cast<f4*>(a.Cpre + co) = v;
                if (a.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
                }
                if (a.R) v += *reinterpret_cast<const f4*>(a.R + co);
                if (a.accumulate) v += *reinterpret_cast<const f4*>(a.C + co);
                *reinterpret_cast<f4*>(a.C + co) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= a.N) continue;
                    float x = v[e];
                    if (a.bias) x += a.bias[n + e];
                    const size_t co = crow + (size_t)(n + e) * a.cns;
                    if (a.Cpre) a.Cpre[co] = x;
                    if (a.act == 1) x = x / (1.0f + expf(-x));
                    if (a.R) x += a.R[co];
                    if (a.accumulate) x += a.C[co];
                    a.C[co] = x;
                }
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
