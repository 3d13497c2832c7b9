// Device-side helpers of the fused step kernel (ls_step.hip) and the training backward (ls_train_bwd.hip).
#pragma once
#include <type_traits>
#include "ls_internal.h"
#include "ls_philox.h"
#include "ls_lanes.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) bf8* gbf8p;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ablation switches exist in -DLS_DEBUG builds only; in the shipped library the test is a compile-time false
#ifdef LS_DEBUG
#define LS_ABLATED(args, bit) (((args).ablate & (bit)) != 0)
#else
#define LS_ABLATED(args, bit) false
#endif

// x * sigmoid(x); v_exp_f32 + v_rcp_f32 (each ~1 ulp), far inside the 1e-3 parity budget.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// r + SiLU(v) in 5 VALU ops: v_mul (exp2 scale), v_exp, v_add, v_rcp, v_fma -- these epilogues are issue-bound
__device__ __forceinline__ float silu_acc(float v, float r) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
    return fmaf(v, s, r);
}
// the same for a float4 of accumulators: the three non-transcendental operations as packed f32 (v_pk_mul / v_pk_add / v_pk_fma:
// two elements per issue) -- these epilogues run after the MFMA loop, where packed math is not the anti-lever it is next to MFMAs
__device__ __forceinline__ f4 silu_acc4(f4 v, f4 r) {
#ifdef LS_SILU_SCALAR
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = silu_acc(v[j], r[j]);
    return o;
#else
    const f2 c = (f2){-1.4426950408889634f, -1.4426950408889634f}, one = (f2){1.0f, 1.0f};
    f2 v0 = (f2){v[0], v[1]}, v1 = (f2){v[2], v[3]};
    f2 t0 = v0 * c, t1 = v1 * c;
    f2 e0 = (f2){__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])}, e1 = (f2){__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])};
    e0 = e0 + one; e1 = e1 + one;
    const f2 s0 = (f2){__builtin_amdgcn_rcpf(e0[0]), __builtin_amdgcn_rcpf(e0[1])}, s1 = (f2){__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1])};
    const f2 o0 = __builtin_elementwise_fma(v0, s0, (f2){r[0], r[1]}), o1 = __builtin_elementwise_fma(v1, s1, (f2){r[2], r[3]});
    return (f4){o0[0], o0[1], o1[0], o1[1]};
#endif
}

// g * SiLU'(a) for a float4, SiLU'(a) = s (1 + a (1 - s)), s = sigmoid(a); non-transcendental operations as packed f32
__device__ __forceinline__ f4 silu_grad4(f4 a, f4 g) {
    const f4 c = (f4){-1.4426950408889634f, -1.4426950408889634f, -1.4426950408889634f, -1.4426950408889634f};
    const f4 one = (f4){1.0f, 1.0f, 1.0f, 1.0f};
    const f4 t = a * c;
    f4 e;
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(t[j]);
    e = e + one;
    f4 sg;
#pragma unroll
    for (int j = 0; j < 4; ++j) sg[j] = __builtin_amdgcn_rcpf(e[j]);
    return g * (sg * __builtin_elementwise_fma(a, one - sg, one));
}

// Pointers fetched from the DevWeights block are generic to the compiler; cast them to the global
// address space so loads are global_load (vmcnt only) instead of flat_load (vmcnt AND lgkmcnt, which
// would make every LDS wait in the GEMM loops also wait for the weight prefetch).
typedef const __attribute__((address_space(1))) f4* gf4p;
typedef const __attribute__((address_space(1))) float* gfp;
__device__ __forceinline__ gf4p g4(const float* p) { return (gf4p)(const f4*)p; }
__device__ __forceinline__ gfp g1(const float* p) { return (gfp)p; }

// Weight images are read through a buffer descriptor (ls_lanes.h: uniform_rsrc): SGPR base + lane * 16 + an SGPR offset that walks the image.
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
__device__ __forceinline__ wrsrc_t wrsrc(const void* base) { return uniform_rsrc(base); }
__device__ __forceinline__ f4 wload4(wrsrc_t r, int lane_bytes, int wave_bytes) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, wave_bytes, 0));
}
__device__ __forceinline__ bf8 wload8h(wrsrc_t r, int lane_bytes, int wave_bytes) {
    return __builtin_bit_cast(bf8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, wave_bytes, 0));
}
__device__ __forceinline__ float wload1(wrsrc_t r, int lane_bytes, int wave_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, wave_bytes, 0));
}


// which (tile, k step) pairs of the block-diagonal token-mixing GEMM touch a non-zero block
__host__ __device__ constexpr bool tokmix_needed(int S, int t, int m) {
    const int R = 2 * S;
    const int r_lo = 16 * t;
    if (r_lo >= R) return false;
    const int r_hi = (16 * t + 15 < R - 1) ? 16 * t + 15 : R - 1;
    const int src_lo = (r_lo / S) * S, src_hi = (r_hi / S + 1) * S - 1;
    return !(4 * m + 3 < src_lo || 4 * m > src_hi);
}

// same for the bf16 token-mix MFMA, whose k step covers 32 source rows
__host__ __device__ constexpr bool tokmix_needed32(int S, int t, int ks) {
    const int R = 2 * S;
    const int r_lo = 16 * t;
    if (r_lo >= R) return false;
    const int r_hi = (16 * t + 15 < R - 1) ? 16 * t + 15 : R - 1;
    const int src_lo = (r_lo / S) * S, src_hi = (r_hi / S + 1) * S - 1;
    return !(32 * ks + 31 < src_lo || 32 * ks > src_hi);
}

}  // namespace ls
