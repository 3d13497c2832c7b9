// Cross-lane reductions on the VALU (DPP / permlane-swap) instead of __shfl_xor, which compiles to ds_bpermute_b32: every
// exchange then is an LDS instruction with LDS latency, and an epilogue that reduces 16 values over 16 lanes issues 128 of them.
#pragma once
#include <hip/hip_runtime.h>

namespace ls {

// Buffer descriptor over [p, p + 2 GiB) held in SGPRs: loads through it take an SGPR base + one 32-bit VGPR byte offset + an SGPR
// offset.  Measured on the MI355X (tools/vmem_cost.cpp, 3 waves / SIMD under a saturated fp32 matrix pipe): a global_load_dwordx4 with
// a per-lane 64-bit address costs 16.8 matrix-pipe cycles of issue, buffer_load_dwordx4 in this form 3.7 (and `... lds`, straight
// into LDS, 5.7 against 16.8 + 28.0 for load + ds_write_b128), and the per-load 64-bit pointer arithmetic becomes s_add.
// The readfirstlane pair (through UNSIGNED temporaries: the builtin returns int, and an int OR-ed into the 64-bit address
// sign-extends) keeps the descriptor in SGPRs where the compiler cannot prove the pointer uniform; without it every load is wrapped
// in a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((unsigned long long)hi << 32) | (unsigned long long)lo), 0, 0x7fffffff, 0x00020000);
}

// the same over [p, p + bytes): a load at or past `bytes` returns 0 (raw buffer, range-checked on the byte offset)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, int bytes) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((unsigned long long)hi << 32) | (unsigned long long)lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {       // v + v[lane selected by the DPP control]
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15); every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float v) {
    v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]: lane ^ 1
    v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]: lane ^ 2
    v = dpp_add<0x141>(v);       // row_half_mirror: the other quad of the half row
    v = dpp_add<0x140>(v);       // row_mirror: the other half row
    return v;
}

// sum over the whole wave; the result is wave-uniform
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48));
}

// maximum over the whole wave (wave-uniform result)
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ float wave_max(float v) {
    v = dpp_max<0xB1>(v);
    v = dpp_max<0x4E>(v);
    v = dpp_max<0x141>(v);
    v = dpp_max<0x140>(v);
    const int i = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}

// v + v[lane ^ 32] / v + v[lane ^ 16] with gfx950's v_permlane32_swap / v_permlane16_swap (VALU, no LDS round trip):
// swap(a, b) exchanges the upper half (odd rows) of a with the lower half (even rows) of b; with a = b = v the two results
// hold v's lower and upper halves (even and odd rows) broadcast over the pair, so their sum is the xor-exchange sum.
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// both members of the exchange pair: (value of the even row, value of the odd row) / (lower half, upper half)
__device__ __forceinline__ void xor16_pair(float v, float& even, float& odd) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    even = __uint_as_float(r[0]); odd = __uint_as_float(r[1]);
}
__device__ __forceinline__ void xor32_pair(float v, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}
// v[lane ^ 32] / v[lane ^ 16] themselves
__device__ __forceinline__ float xor32_get(float v, int lane) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor16_get(float v, int lane) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane & 16) ? r[0] : r[1]);
}

}  // namespace ls
