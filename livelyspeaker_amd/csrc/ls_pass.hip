// Instantiations and launch entry of the one-pass-per-workgroup step kernel (ls_pass_kernel.h): TED (S = 35, J*F = 27) and BEAT (S = 36, J*F = 282),
// exact fp32 and bf16x3, as 4-wave workgroups (two per CU) and as 8-wave workgroups (one per CU).
#include "ls_pass_kernel.h"

namespace ls {

static size_t pass_lds_bytes(Variant v, int nw) { return (size_t)pass_lds_floats(v == kTED ? 35 : 36, nw) * sizeof(float); }

// Opt in to > 64 KiB dynamic LDS once per process (outside stream capture).
hipError_t init_pass_kernels() {
    struct K { const void* f; Variant v; int nw; };
    const K ks[] = {{reinterpret_cast<const void*>(k_pass<35, 1, 27, 0, 4>), kTED, 4},   {reinterpret_cast<const void*>(k_pass<35, 1, 27, 1, 4>), kTED, 4},
                    {reinterpret_cast<const void*>(k_pass<35, 1, 27, 0, 8>), kTED, 8},   {reinterpret_cast<const void*>(k_pass<35, 1, 27, 1, 8>), kTED, 8},
                    {reinterpret_cast<const void*>(k_pass<36, 2, 282, 0, 4>), kBEAT, 4}, {reinterpret_cast<const void*>(k_pass<36, 2, 282, 1, 4>), kBEAT, 4},
                    {reinterpret_cast<const void*>(k_pass<36, 2, 282, 0, 8>), kBEAT, 8}, {reinterpret_cast<const void*>(k_pass<36, 2, 282, 1, 8>), kBEAT, 8}};
    for (const K& k : ks) {
        hipError_t e = hipFuncSetAttribute(k.f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds_bytes(k.v, k.nw));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// One launch = `nsamples` samples starting at a.b0, a.npass workgroups each.  No workgroup waits for another one, so the grid may be any
// size.  waves: 4 = the form two of which share a CU; 8 = one workgroup per CU (the caller picks it when the grid fits the chip once).
hipError_t launch_step_pass(Variant v, int prec, int waves, const StepArgs& a, int nsamples, hipStream_t st) {
    if (nsamples < 1 || (a.npass != 1 && a.npass != 2) || (waves != 4 && waves != 8)) return hipErrorInvalidValue;
    const dim3 grid(nsamples * a.npass), block(64 * waves);
    const size_t lds = pass_lds_bytes(v, waves);
#define LS_PASS_LAUNCH(S_, NPRE_, JF_)                                                                            \
    do {                                                                                                          \
        if (waves == 4) {                                                                                         \
            if (prec == 0) hipLaunchKernelGGL((k_pass<S_, NPRE_, JF_, 0, 4>), grid, block, lds, st, a);           \
            else hipLaunchKernelGGL((k_pass<S_, NPRE_, JF_, 1, 4>), grid, block, lds, st, a);                     \
        } else {                                                                                                  \
            if (prec == 0) hipLaunchKernelGGL((k_pass<S_, NPRE_, JF_, 0, 8>), grid, block, lds, st, a);           \
            else hipLaunchKernelGGL((k_pass<S_, NPRE_, JF_, 1, 8>), grid, block, lds, st, a);                     \
        }                                                                                                         \
    } while (0)
    if (v == kTED) LS_PASS_LAUNCH(35, 1, 27);
    else LS_PASS_LAUNCH(36, 2, 282);
#undef LS_PASS_LAUNCH
    return hipGetLastError();
}

}  // namespace ls
