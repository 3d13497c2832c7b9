// Philox4x32-10 counter RNG + Box-Muller, shared by the step kernel and the x_T fill kernel.
// Throughput-mode noise only: parity mode consumes the host-drawn tape in the reference's draw order
// (gaussian_diffusion.py:700-743).  oracle/philox_oracle.py restates these streams in numpy (block function pinned to
// Random123's known-answer vectors), so a Philox-mode run can be replayed through the CPU oracle.
#pragma once
#include "ls_internal.h"

namespace ls {

__device__ __forceinline__ void philox4x32(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// one Box-Muller pair from two 32-bit words: (r cos 2 pi u1, r sin 2 pi u1), r = sqrt(-2 ln u0), u in (0, 1].
// v_log_f32 is log2 and v_sin_f32 / v_cos_f32 take REVOLUTIONS, so u1 goes in as it is: no 2 pi multiply, no range reduction.
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& zc, float& zs) {
    const float u0 = ((float)a + 0.5f) * 2.3283064365386963e-10f;
    const float u1 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));      // -2 ln 2 * log2(u0)
    zc = rad * __builtin_amdgcn_cosf(u1);
    zs = rad * __builtin_amdgcn_sinf(u1);
}

// Elements 4*blk .. 4*blk+3 of stream (step_id, stream) of global sample gidx: ONE Philox block, two Box-Muller pairs.
// Keyed by the global sample index, so the streams are invariant to how the batch is sharded over GPUs.
__device__ __forceinline__ void philox_normal4(const CallParams* cp, unsigned long long gidx, unsigned step_id, unsigned stream,
                                               unsigned blk, float z[4]) {
    unsigned c[4] = {blk, step_id * 4u + stream, (unsigned)gidx, (unsigned)(gidx >> 32)};
    philox4x32(c, (unsigned)cp->seed, (unsigned)(cp->seed >> 32));
    box_muller(c[0], c[1], z[0], z[1]);
    box_muller(c[2], c[3], z[2], z[3]);
}

// single element e of the same stream (the sampler's step noise: one or two elements per thread)
__device__ __forceinline__ float philox_normal(const CallParams* cp, unsigned long long gidx, unsigned step_id, unsigned stream,
                                               unsigned e) {
    unsigned c[4] = {e >> 2, step_id * 4u + stream, (unsigned)gidx, (unsigned)(gidx >> 32)};
    philox4x32(c, (unsigned)cp->seed, (unsigned)(cp->seed >> 32));
    float zc, zs;
    box_muller((e & 2) ? c[2] : c[0], (e & 2) ? c[3] : c[1], zc, zs);
    return (e & 1) ? zs : zc;
}

}  // namespace ls
