// Strided fp32 MFMA GEMM (the sampler's once-per-call stage, the SAG decoder, the long-sequence path, the FGD evaluator and the training step's forward, data-gradient and weight-gradient products):
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
// v_mfma_f32_16x16x4_f32, 128x128 tile / 256 threads, K chunk 32, same transposed accumulator form as the step kernel.
// Split-K (gridDim.z > 1) writes partial tiles to a workspace which k_splitk_reduce sums in a fixed order
// (deterministic: no float atomics anywhere in the training step).
//
// DMA variant (both operands K-contiguous, full tiles -- every nn.Linear of the sampler, the SAG decoder and the long path): the
// K tiles go global -> LDS with `buffer_load_dwordx4 ... lds` into a double buffer, one barrier per K tile.  Measured on the MI355X
// (tools/vmem_cost.cpp, issue cost in cycles of a saturated fp32 matrix pipe, 3 waves / SIMD): global_load_dwordx4 16.8 +
// ds_write_b128 28.0 per 16 B of a lane through registers, 5.7 through the LDS-DMA path -- the register-staged loop spends 13 %
// of the pipe's time on its 12 staging instructions per 64 MFMAs.  The DMA destination is lane-linear (1 KB per wave instruction =
// 8 rows x 128 B), so rows sit unpadded at 128 B and the 16-byte chunk c of row r is stored at position c ^ (r & 7): each lane
// FETCHES the chunk that belongs at its position, and the fragment reads apply the same XOR (conflict-free ds_read_b128).
#include "ls_internal.h"
#include "ls_train.h"
#include "ls_lanes.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kTM = 128, kTK = 32, kLdK = kTK + 4, kLdR = kTM + 4;
#ifndef LS_GEMM_HALF_MAX
#define LS_GEMM_HALF_MAX 768      // LDS-DMA grids of at most this many 64-row tiles run as 32-row half tiles (see launch_gemm_tr)
#endif

// epilogue activations: 0 none, 1 SiLU, 2 exp(0.5 y) (std from log-variance), 3 exact GELU (F.gelu default)
__device__ __forceinline__ float gemm_act(float v, int act) {
    if (act == 1) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));   // the step kernel's SiLU (v_exp_f32 + v_rcp_f32)
    if (act == 2) return expf(0.5f * v);
    if (act == 3) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    return v;
}

// The same for a float4, dispatched ONCE per four elements (round 5).  With the scalar form inside `for (e)` the compiler kept the three
// `act ==` tests per element: ~100 scalar branches per lane and tile and one exposed v_exp -> v_rcp chain after the other -- SiLU cost
// 4 %, GELU 10 % of a 77824 x 512 x 512 product (tools/gemm_bench.cpp).  Same operations in the same order per element: bit-identical results.
__device__ __forceinline__ f4 gemm_act4(f4 v, int act) {
    if (act == 1) {
        const f4 t = v * -1.4426950408889634f;
        f4 e;
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(t[j]);
        e = e + 1.0f;
        f4 s;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = __builtin_amdgcn_rcpf(e[j]);
        return v * s;
    }
    if (act == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = expf(0.5f * v[j]);
        return v;
    }
    if (act == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752f));
        return v;
    }
    return v;
}

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

// The same for the common case -- full 128-row tile, whole K tiles, float4-legal, single-level reduction index: four
// unconditional loads off one running pointer (`at` already includes the thread's k position; it advances by one K tile per
// iteration).  The general fetch above costs ~450 executed VALU/SALU instructions per K tile (bounds tests, two-level index
// arithmetic), which on gfx950 come straight out of the fp32 MFMA issue time.
template <bool KC, int NI = 4>
__device__ __forceinline__ void fetch_fast(f4 (&v)[4], const float* at, const size_t (&roff)[4], long long ks) {
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const f4*>(KC ? at + roff[i] : at + (long long)(8 * i) * ks);
}

// registers -> LDS: K-contiguous operands as [row][k] (stride 36), row-contiguous ones as [k][row] (stride 132)
template <bool KC, int NI = 4>
__device__ __forceinline__ void put(float* s, const f4 (&v)[4], int tid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
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

// MT: 16-row m tiles per wave -> 128 (MT = 4) or 64 (MT = 2) rows of A per workgroup.  The 64-row tile exists for products whose
// 128 x 128 tile count fills the chip badly (e.g. 544 tiles on 768 workgroup slots: one CU in eight runs three tiles while the
// others run two); only the K-contiguous fast path is instantiated with it.
typedef __attribute__((address_space(3))) void* lds_vp;

// one K tile of both operands, global -> LDS: wave wv issues MT + 4 buffer_load_dwordx4 ... lds, each filling 8 rows (1 KB)
template <int MT>
__device__ __forceinline__ void gemm_dma_issue(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, float* sA, float* sB, int wv,
                                               const int (&voA)[MT], const int (&voB)[4], int k0) {
    const int kb = __builtin_amdgcn_readfirstlane(k0 * 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vp)&sA[(wv * MT + i) * 256], 16, voA[i], kb, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vp)&sB[(wv * 4 + i) * 256], 16, voB[i], kb, 0, 0);
}

template <bool AK, bool BK, bool FAST, int MT, bool DMA>
constexpr int gemm_lds_a() { return DMA ? 2 * 32 * MT * kTK : (AK ? 32 * MT * kLdK : kTK * kLdR); }
template <bool BK, bool DMA>
constexpr int gemm_lds_b() { return DMA ? 2 * kTM * kTK : (BK ? kTM * kLdK : kTK * kLdR); }

// one output tile (32 MT rows x 128 columns) of problem bz: tile row `by` (in units of 32 MT rows), tile column `bx`
template <bool AK, bool BK, bool FAST, int MT = 4, bool DMA = false>
__device__ __forceinline__ void gemm_tile(GemmArgs a, float* __restrict__ sA, float* __restrict__ sB, int bx, int by, int bz) {
    static_assert(MT == 4 || ((MT == 2 || MT == 1) && AK && FAST), "64- / 32-row tiles: K-contiguous A on the fast path only");
    static_assert(!DMA || (AK && BK && FAST), "LDS-DMA staging: both operands K-contiguous, full tiles");
    static_assert(MT != 1 || DMA, "32-row tiles exist in the LDS-DMA kernel only");
    constexpr int BM = 32 * MT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = by * BM, n0 = bx * kTM;
    const int s16 = lane & 15, g = lane >> 4;
    const int zb = bz / a.splits, z = bz - zb * a.splits;     // (problem of the batch, K split)
    const int kbeg = z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    a.A.p += (size_t)zb * a.bsA;
    a.B.p += (size_t)zb * a.bsB;
    a.C += (size_t)zb * a.bsC;
    if (a.R) a.R += (size_t)zb * a.bsC;             // the residual and the pre-activation copy share C's addressing, batch stride included;
    if (a.Cpre) a.Cpre += (size_t)zb * a.bsC;       // the bias is shared by the problems of a batch

    f4 acc[4][MT];                                  // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    // lane (m = s16 of m tile j, g) holds n = n0 + wn*64 + 16*i + 4*g + {0..3}
    const bool partial = a.splits > 1;
    float* Cz = partial ? a.ws + (size_t)bz * a.M * a.N : nullptr;
    const bool cvec = a.cns == 1 && (a.crs & 3) == 0 && (a.cri == INT_MAX || (a.cro & 3) == 0) && ((uintptr_t)a.C & 15) == 0 &&
                      (!a.Cpre || ((uintptr_t)a.Cpre & 15) == 0) && (!a.R || ((uintptr_t)a.R & 15) == 0) &&
                      (!a.bias || ((uintptr_t)a.bias & 15) == 0);
    // Full tile with float4-legal output (`fastout`): the bias and every residual / accumulate operand are fetched by fetch_addends()
    // BEFORE the first store -- interleaved as load -> add -> store per element the compiler must keep each load behind the previous
    // store, one L2 round trip after the other, 4 * MT in a row at the end of every workgroup.  The DMA loop calls it ahead of its last
    // K tile, so the operands land under that tile's MFMAs.
    // The register-staged variants (128-row tiles: 64 accumulator registers) fetch them one 16-row tile at a time inside the epilogue
    // instead -- all 4 * MT float4 at once would double their register footprint and cost a wave of occupancy.
    const bool fastout = FAST && cvec && !partial;
    f4 biasv[4], rv[4][DMA ? MT : 1];
    f4 wsv[DMA ? 4 : 1], addv[DMA ? 4 : 1];
    float lmean[DMA ? MT : 1], lrstd[DMA ? MT : 1];
    auto fetch_addends = [&](int m0, int n0) {
        if (!DMA || !fastout) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) biasv[i] = a.bias ? *reinterpret_cast<const f4*>(a.bias + n0 + wn * 64 + 16 * i + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
        if constexpr (DMA) {
            if (a.ln_part) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wsv[i] = *reinterpret_cast<const f4*>(a.wsum + n0 + wn * 64 + 16 * i + 4 * g);
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    // the row's statistics from its eight 64-column partials (equal counts): mean of means; M2 = sum M2_p + 64 sum (mean_p - mean)^2
                    const f4* pp = reinterpret_cast<const f4*>(a.ln_part + (size_t)(m0 + wm * 16 * MT + 16 * j + s16) * 16);
                    const f4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
                    const float mean = (((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) * 0.125f;
                    float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]));
                    float dq = 0.f;
#define LS_DQ(v) { const float d = (v) - mean; dq = fmaf(d, d, dq); }
                    LS_DQ(p0[0]) LS_DQ(p0[2]) LS_DQ(p1[0]) LS_DQ(p1[2]) LS_DQ(p2[0]) LS_DQ(p2[2]) LS_DQ(p3[0]) LS_DQ(p3[2])
#undef LS_DQ
                    m2 = fmaf(64.f, dq, m2);
                    lmean[j] = mean;
                    lrstd[j] = 1.0f / sqrtf(m2 * (1.0f / 512.f) + 1e-5f);
                }
            }
            if (a.addn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) addv[i] = *reinterpret_cast<const f4*>(a.addn + n0 + wn * 64 + 16 * i + 4 * g);
            }
        }
        if (a.R || a.accumulate) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const size_t co = lvl(m0 + wm * 16 * MT + 16 * j + s16, a.cri, a.cro, a.crs) + n0 + wn * 64 + 4 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rv[i][j] = a.R ? *reinterpret_cast<const f4*>(a.R + co + 16 * i) : (f4){0.f, 0.f, 0.f, 0.f};
                    if (a.accumulate) rv[i][j] += *reinterpret_cast<const f4*>(a.C + co + 16 * i);
                }
            }
        }
    };
    auto epilogue = [&](int m0, int n0) {
        if (fastout) {
            size_t co[MT];
#pragma unroll
            for (int j = 0; j < MT; ++j) co[j] = lvl(m0 + wm * 16 * MT + 16 * j + s16, a.cri, a.cro, a.crs) + n0 + wn * 64 + 4 * g;
            if (!DMA) {
#pragma unroll
                for (int i = 0; i < 4; ++i) biasv[i] = a.bias ? *reinterpret_cast<const f4*>(a.bias + n0 + wn * 64 + 16 * i + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                if (!DMA && (a.R || a.accumulate)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        rv[i][0] = a.R ? *reinterpret_cast<const f4*>(a.R + co[j] + 16 * i) : (f4){0.f, 0.f, 0.f, 0.f};
                        if (a.accumulate) rv[i][0] += *reinterpret_cast<const f4*>(a.C + co[j] + 16 * i);
                    }
                }
                f4 outv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f4 v;
                    if constexpr (DMA) {
                        if (a.ln_part) v = (acc[i][j] - lmean[j] * wsv[i]) * lrstd[j] + biasv[i];
                        else v = acc[i][j] + biasv[i];
                    } else v = acc[i][j] + biasv[i];
                    if (a.Cpre) *reinterpret_cast<f4*>(a.Cpre + co[j] + 16 * i) = v;
                    v = gemm_act4(v, a.act);
                    if (a.R || a.accumulate) v += rv[i][DMA ? j : 0];
                    if constexpr (DMA) { if (a.addn) v += addv[i]; }
                    *reinterpret_cast<f4*>(a.C + co[j] + 16 * i) = v;
                    outv[i] = v;
                }
                if constexpr (DMA) {
                    if (a.part_out) {
                        // (mean, M2) of this wave's 64 output columns of row m: 16 values in the lane, the other 48 in the lanes s16 + 16 g'
                        float sm = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) sm += (outv[i][0] + outv[i][1]) + (outv[i][2] + outv[i][3]);
                        sm = xor32_sum(xor16_sum(sm));
                        const float mean = sm * (1.0f / 64.f);
                        float q = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float d = outv[i][e] - mean; q = fmaf(d, d, q); }
                        q = xor32_sum(xor16_sum(q));
                        if (g == 0) {
                            float* po = a.part_out + ((size_t)(m0 + wm * 16 * MT + 16 * j + s16) * (a.N / 64) + (n0 / 64 + wn)) * 2;
                            po[0] = mean; po[1] = q;
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + wm * 16 * MT + 16 * j + s16;
            if (!FAST && m >= a.M) continue;
            const size_t crow = partial ? (size_t)m * a.N : lvl(m, a.cri, a.cro, a.crs);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + 16 * i + 4 * g;
                if (!FAST && n >= a.N) continue;
                f4 v = acc[i][j];
                if (partial) {
                    if ((a.N & 3) == 0) *reinterpret_cast<f4*>(&Cz[crow + n]) = v;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < a.N) Cz[crow + n + e] = v[e];
                    }
                    continue;
                }
                if (cvec && (FAST || n + 3 < a.N)) {
                    const size_t co = crow + n;
                    if (a.bias) v += *reinterpret_cast<const f4*>(a.bias + n);
                    if (a.Cpre) *reinterpret_cast<f4*>(a.Cpre + co) = v;
                    v = gemm_act4(v, a.act);
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
                        x = gemm_act(x, a.act);
                        if (a.R) x += a.R[co];
                        if (a.accumulate) x += a.C[co];
                        a.C[co] = x;
                    }
                }
            }
        }
    };

    if constexpr (DMA) {
        const auto rsA = uniform_rsrc(a.A.p), rsB = uniform_rsrc(a.B.p);
        // wave wv stages blocks wv*MT + i of A and wv*4 + i of B (8 rows each); lane = (row in block, position in row)
        const int rl = lane >> 3, ch = (lane & 7) ^ rl;             // this lane's position holds chunk ch of its row
        // fragment (row, k half kk): chunk 4 kk + g of the row, stored at position (4 kk + g) ^ (row & 7); row & 7 = s16 & 7 for every tile
        const int o0 = ((g ^ (s16 & 7)) * 4), o1 = o0 ^ 16;
        const int fa = (wm * 16 * MT + s16) * kTK, fb = (wn * 64 + s16) * kTK;
        int voA[MT], voB[4];
#pragma unroll
        for (int i = 0; i < MT; ++i) voA[i] = (int)(((long long)(m0 + (wv * MT + i) * 8 + rl) * a.A.rs + ch * 4) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) voB[i] = (int)(((long long)(n0 + (wv * 4 + i) * 8 + rl) * a.B.rs + ch * 4) * 4);
        gemm_dma_issue<MT>(rsA, rsB, sA, sB, wv, voA, voB, kbeg);
        __syncthreads();                                    // carries the vmcnt(0) of the DMA above
        int buf = 0;
        // (A persistent form -- one workgroup per resident slot walking tiles blockIdx.x, + gridDim.x, ..., the next output tile's first K
        // tile requested ahead of the current one's last -- measured SLOWER, 229 -> 265 us at 17408 x 1536 x 512: with 4.25 tiles per slot
        // the static walk ends 5 : 4 unbalanced, which costs more than the hidden first fetch gains.)
        for (int k0 = kbeg; k0 < kend; k0 += kTK) {
            // the other buffer was last read before the previous barrier: the next K tile lands in it while this one is multiplied
            if (k0 + kTK >= kend) fetch_addends(m0, n0);
            if (k0 + kTK < kend) gemm_dma_issue<MT>(rsA, rsB, sA + (buf ^ 1) * BM * kTK, sB + (buf ^ 1) * kTM * kTK, wv, voA, voB, k0 + kTK);
            const float* cA = sA + buf * BM * kTK + fa;
            const float* cB = sB + buf * kTM * kTK + fb;
            f4 af[2][MT], bf[2];
#pragma unroll
            for (int j = 0; j < MT; ++j) af[0][j] = *reinterpret_cast<const f4*>(cA + 16 * j * kTK + o0);
            bf[0] = *reinterpret_cast<const f4*>(cB + o0);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int kk = st >> 2, i = st & 3;
                if (st + 1 < 8) bf[(st + 1) & 1] = *reinterpret_cast<const f4*>(cB + 16 * ((st + 1) & 3) * kTK + (((st + 1) >> 2) ? o1 : o0));
                if (st == 0) {
#pragma unroll
                    for (int j = 0; j < MT; ++j) af[1][j] = *reinterpret_cast<const f4*>(cA + 16 * j * kTK + o1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < MT; ++j) acc[i][j] = MFMA(bf[st & 1][e], af[kk][j][e], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            buf ^= 1;
        }
    } else {
    size_t roffA[4], roffB[4];
    row_offsets<AK>(a.A, roffA, m0, a.M, tid);
    row_offsets<BK>(a.B, roffB, n0, a.N, tid);
    f4 ra[4], rb[4];
    // FAST: running pointers of this thread's share of the current K tile
    const float* pa = nullptr;
    const float* pb = nullptr;
    if (FAST) {
        pa = AK ? a.A.p + kbeg + (tid & 7) * 4 : a.A.p + roffA[0] + (long long)(kbeg + (tid >> 5)) * a.A.ks;
        pb = BK ? a.B.p + kbeg + (tid & 7) * 4 : a.B.p + roffB[0] + (long long)(kbeg + (tid >> 5)) * a.B.ks;
        fetch_fast<AK, MT>(ra, pa, roffA, a.A.ks);
        fetch_fast<BK>(rb, pb, roffB, a.B.ks);
    } else {
        fetch<AK>(ra, a.A, roffA, m0, a.M, kbeg, kend, tid);
        fetch<BK>(rb, a.B, roffB, n0, a.N, kbeg, kend, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += kTK) {
        __syncthreads();
        put<AK, MT>(sA, ra, tid);
        put<BK>(sB, rb, tid);
        __syncthreads();
        if (FAST) {                         // next tile's global loads overlap this tile's MFMAs; the last iteration re-reads its
            const long long adv = k0 + kTK < kend ? kTK : 0;          // own tile instead of branching around the loads
            pa += AK ? adv : adv * a.A.ks;
            pb += BK ? adv : adv * a.B.ks;
            fetch_fast<AK, MT>(ra, pa, roffA, a.A.ks);
            fetch_fast<BK>(rb, pb, roffB, a.B.ks);
            __builtin_amdgcn_sched_barrier(0);      // or the scheduler sinks these loads below the MFMAs, right in front of their use
        } else if (k0 + kTK < kend) {
            fetch<AK>(ra, a.A, roffA, m0, a.M, k0 + kTK, kend, tid);
            fetch<BK>(rb, a.B, roffB, n0, a.N, k0 + kTK, kend, tid);
        }
        if (FAST) {
            // fragment loads run one step ahead of the MFMAs that use them (8 steps per K tile: 2 k-halves x 4 n tiles): the fast
            // staging path leaves the registers for a second A set and a second B fragment at 3 waves / SIMD
            f4 af[2][MT], bf[2];
#pragma unroll
            for (int j = 0; j < MT; ++j) af[0][j] = frag<AK>(sA, wm * 16 * MT + 16 * j + s16, 0, g);
            bf[0] = frag<BK>(sB, wn * 64 + s16, 0, g);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int kk = st >> 2, i = st & 3;
                if (st + 1 < 8) bf[(st + 1) & 1] = frag<BK>(sB, wn * 64 + 16 * ((st + 1) & 3) + s16, (st + 1) >> 2, g);
                if (st == 0) {
#pragma unroll
                    for (int j = 0; j < MT; ++j) af[1][j] = frag<AK>(sA, wm * 16 * MT + 16 * j + s16, 1, g);
                }
                __builtin_amdgcn_sched_barrier(0);  // pins [LDS reads of the next step][MFMAs of this step]
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < MT; ++j) acc[i][j] = MFMA(bf[st & 1][e], af[kk][j][e], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < kTK / 16; ++kk) {
                f4 af[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) af[j] = frag<AK>(sA, wm * 16 * MT + 16 * j + s16, kk, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {       // one B fragment live at a time keeps the kernel at 3 waves / SIMD
                    const f4 bf = frag<BK>(sB, wn * 64 + 16 * i + s16, kk, g);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < MT; ++j) acc[i][j] = MFMA(bf[e], af[j][e], acc[i][j]);
                }
            }
        }
    }
    }
    epilogue(m0, n0);
}

template <bool AK, bool BK, bool FAST, int MT = 4>
__global__ __launch_bounds__(256, 3) void k_gemm_tr(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[gemm_lds_a<AK, BK, FAST, MT, false>()];
    __shared__ __attribute__((aligned(16))) float sB[gemm_lds_b<BK, false>()];
    gemm_tile<AK, BK, FAST, MT, false>(a, sA, sB, blockIdx.x, blockIdx.y, blockIdx.z);
}

// LDS-DMA kernel: 64 x 128 tiles, and -- for the tiles of a last, badly filled round -- 32 x 128 half tiles.  Workgroups are handed
// out in blockIdx order as slots free up, so a 1-D grid [nbig whole tiles | 2 (ntiles - nbig) half tiles] is a dynamic schedule for
// free: 1088 tiles on the chip's 768 slots (17408 x 512, the SAG decoder's out-proj / FFN2) are 768 whole tiles + 640 halves =
// one round + one short round instead of two rounds of which the second is 42 % full.  Every output element is still produced by one
// workgroup with the same reduction order, so results do not depend on the split (tests/test_sag.py's batch-composition test).
__global__ __launch_bounds__(256, 3) void k_gemm_dma(GemmArgs a, int gx, int nbig) {
    __shared__ __attribute__((aligned(16))) float sA[gemm_lds_a<true, true, true, 2, true>()];
    __shared__ __attribute__((aligned(16))) float sB[gemm_lds_b<true, true>()];
    if (gridDim.z > 1 || (int)blockIdx.x < nbig) {            // whole tile (batched / split launches keep the 3-D grid: gx = 0)
        if (gx == 0) gemm_tile<true, true, true, 2, true>(a, sA, sB, blockIdx.x, blockIdx.y, blockIdx.z);
        else gemm_tile<true, true, true, 2, true>(a, sA, sB, (int)blockIdx.x % gx, (int)blockIdx.x / gx, 0);
    } else {
        const int u = (int)blockIdx.x - nbig, t = nbig + (u >> 1);
        gemm_tile<true, true, true, 1, true>(a, sA, sB, t % gx, 2 * (t / gx) + (u & 1), 0);
    }
}

// C[m][n] (+)= sum_z ws[z][m][n] (+ bias); fixed summation order
__global__ void k_splitk_reduce(const GemmArgs a, int Z) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)a.M * a.N) return;
    const int m = (int)(i / a.N), n = (int)(i % a.N);
    const float* ws = a.ws + (size_t)blockIdx.y * Z * a.M * a.N;          // blockIdx.y: problem of the batch
    float* C = a.C + (size_t)blockIdx.y * a.bsC;
    float v = 0.f;
#pragma unroll 8                                    // independent loads in flight; the sum keeps its z order
    for (int z = 0; z < Z; ++z) v += ws[(size_t)z * a.M * a.N + i];
    if (a.bias) v += a.bias[n];
    const size_t co = (size_t)(m / a.cri) * a.cro + (size_t)(m % a.cri) * a.crs + (size_t)n * a.cns;
    if (a.accumulate) v += C[co];
    C[co] = v;
}

hipError_t launch_gemm_tr(GemmArgs a, bool a_kcontig, bool b_kcontig, int splits, hipStream_t st) {
    if (splits < 1) splits = 1;
    // chunk is a multiple of the K tile so that every split starts on a tile boundary
    int kchunk = ((a.K + splits - 1) / splits + kTK - 1) / kTK * kTK;
    splits = (a.K + kchunk - 1) / kchunk;
    a.kchunk = kchunk;
    a.splits = splits;
    if (a.nbatch < 1) a.nbatch = 1;
    if (splits > 1 && (!a.ws || (size_t)a.nbatch * splits * a.M * a.N > a.ws_floats)) return hipErrorInvalidValue;
    if (splits > 1 && (a.Cpre || a.R || a.act)) return hipErrorInvalidValue;
    dim3 grid((a.N + kTM - 1) / kTM, (a.M + kTM - 1) / kTM, a.nbatch * splits);
    // fast staging path: full tiles, whole K tiles in every split, float4-legal operands with a single-level reduction index
    const bool fast = a.M % kTM == 0 && a.N % kTM == 0 && a.K % kTK == 0 && a.A.vec && a.B.vec && a.A.ki == INT_MAX && a.B.ki == INT_MAX;
    // 64-row tiles when they balance the 256 CUs better: every CU works through ceil(workgroups / 256) tiles (co-resident ones share
    // its matrix pipes), so the efficiency of a grid is (workgroups / 256) / ceil(workgroups / 256) -- 544 tiles: 0.71, 1088 half tiles: 0.85
    auto balance = [](long long wgs) { const double per = (double)wgs / 256.0; return per / (double)((wgs + 255) / 256); };
    const long long t128 = (long long)grid.x * grid.y * grid.z;
    const bool half = fast && a_kcontig && b_kcontig && balance(2 * t128) > balance(t128) + 0.05;
    // LDS-DMA staging: single-level rows and every byte offset of an operand inside 31 bits (buffer addressing)
    const bool dma = fast && a_kcontig && b_kcontig && a.A.ri == INT_MAX && a.B.ri == INT_MAX &&
                     (long long)a.M * a.A.rs < (1ll << 29) && (long long)a.N * a.B.rs < (1ll << 29);
    if (dma) {
        // always 64-row tiles: 48 KB of double buffer lets three workgroups share a CU (the 128-row form: 64 KB, two); measured over the
        // sampler's shapes (tools/gemm_bench.cpp) the 128-row DMA tile lost to this one and, at some, to the register-staged kernel
        grid.y *= 2;
        if (grid.z == 1 && (long long)grid.x * grid.y <= LS_GEMM_HALF_MAX) {
            // a grid of at most one 64-row tile per slot: every tile as two 32-row halves -- twice the workgroups, half the MFMAs per wave and
            // K tile.  A lone workgroup's K tile costs 1.3 us (64 MFMAs per wave + issue + barrier), so what shortens a small product is
            // fewer MFMAs per wave, not a deeper prefetch (a three-buffer form with tiles k+1 and k+2 in flight measured SLOWER, 29.5 vs
            // 26.0 us at 384 x 512 x 512; removed).  Measured: 384 x 512 x 512 (a 4-clip batch's channel mixing) 26 -> 13 us, 4608 rows
            // 43 -> 32 us, 9728 rows 62 -> 55 us; above 768 tiles whole tiles win by 2-3 %.
            hipLaunchKernelGGL(k_gemm_dma, dim3((unsigned)(2 * grid.x * grid.y)), dim3(256), 0, st, a, (int)grid.x, 0);
        } else if (grid.z == 1) {
            // a last round that fills less than ~60 % of the 768 slots runs as half tiles (a half tile takes ~0.6 of a whole one)
            const long long tiles = (long long)grid.x * grid.y, slots = 768;
            const long long rem = tiles % slots;
            const int nbig = (rem > 0 && rem * 10 < slots * 6 && tiles > slots / 2) ? (int)(tiles - rem) : (int)tiles;
            hipLaunchKernelGGL(k_gemm_dma, dim3((unsigned)(nbig + 2 * (tiles - nbig))), dim3(256), 0, st, a, (int)grid.x, nbig);
        } else {
            hipLaunchKernelGGL(k_gemm_dma, grid, dim3(256), 0, st, a, 0, 0);
        }
    } else if (half) {
        grid.y *= 2;
        hipLaunchKernelGGL((k_gemm_tr<true, true, true, 2>), grid, dim3(256), 0, st, a);
    } else {
#define LS_GEMM_LAUNCH(AKV, BKV)                                                                             \
    do {                                                                                                     \
        if (fast) hipLaunchKernelGGL((k_gemm_tr<AKV, BKV, true>), grid, dim3(256), 0, st, a);                \
        else hipLaunchKernelGGL((k_gemm_tr<AKV, BKV, false>), grid, dim3(256), 0, st, a);                    \
    } while (0)
        if (a_kcontig && b_kcontig) LS_GEMM_LAUNCH(true, true);
        else if (a_kcontig && !b_kcontig) LS_GEMM_LAUNCH(true, false);
        else if (!a_kcontig && b_kcontig) LS_GEMM_LAUNCH(false, true);
        else LS_GEMM_LAUNCH(false, false);
#undef LS_GEMM_LAUNCH
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || splits == 1) return e;
    const size_t n = (size_t)a.M * a.N;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256), a.nbatch), dim3(256), 0, st, a, splits);
    return hipGetLastError();
}

// nn.Linear-shaped entry (y = x W^T + b) used by the sampler's once-per-call stage, the SAG decoder and the FGD evaluator:
//     C[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] ) (+ R[m][n]),   both operands K-contiguous
// act: 0 none, 1 SiLU, 2 exp(0.5 y), 3 exact GELU (F.gelu default, nn.TransformerDecoderLayer activation="gelu")
hipError_t launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr,
                          float* C, int ldc, int M, int N, int K, int act, hipStream_t st) {
    if (R && ldr != ldc) return hipErrorInvalidValue;           // the residual shares C's addressing
    GemmArgs a{};
    a.A = op_rows(A, lda, M, K);
    a.B = op_rows(W, ldw, N, K);
    a.C = C;
    a.cri = INT_MAX; a.cro = 0; a.crs = ldc; a.cns = 1;
    a.bias = bias; a.R = R; a.act = act;
    a.M = M; a.N = N; a.K = K;
    return launch_gemm_tr(a, true, true, 1, st);
}

}  // namespace ls
