// Philox4x32-10 counter RNG + Box-Muller, shared by the step kernel and the x_T fill kernel.
// Perf-mode noise only: parity mode consumes the host-drawn tape in the reference's draw order.
#pragma once
#include "ls_internal.h"

namespace ls {

// ---- Philox4x32-10 counter RNG (perf mode; parity mode reads the host noise tape) ---------------
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

// N(0,1) for element e of stream (step_id, stream) of global sample gidx: invariant to how the
// batch is sharded over GPUs.
__device__ __forceinline__ float philox_normal(const CallParams* cp, unsigned long long gidx, unsigned step_id,
                               unsigned stream, unsigned e) {
    unsigned c[4] = {e >> 2, step_id * 4u + stream, (unsigned)gidx, (unsigned)(gidx >> 32)};
    philox4x32(c, (unsigned)cp->seed, (unsigned)(cp->seed >> 32));
    const unsigned a = (e & 2) ? c[2] : c[0], b = (e & 2) ? c[3] : c[1];
    const float u0 = ((float)a + 0.5f) * 2.3283064365386963e-10f;   // (0,1]
    const float u1 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.0f * __logf(u0));
    float sn, cs;
    __sincosf(6.283185307179586f * u1, &sn, &cs);
    return rad * ((e & 1) ? sn : cs);
}


}  // namespace ls
