// Instantiations and launch entry of the one-pass-per-workgroup step kernel (ls_pass_kernel.h): TED (S = 35, J*F = 27) and BEAT (S = 36, J*F = 282).
#include "ls_pass_kernel.h"

namespace ls {

static size_t pass_lds_bytes(Variant v) { return (size_t)pass_lds_floats(v == kTED ? 35 : 36) * sizeof(float); }

// Opt in to > 64 KiB dynamic LDS once per process (outside stream capture).
hipError_t init_pass_kernels() {
    const void* ted[] = {reinterpret_cast<const void*>(k_pass<35, 1, 27, 0>), reinterpret_cast<const void*>(k_pass<35, 1, 27, 1>)};
    const void* beat[] = {reinterpret_cast<const void*>(k_pass<36, 2, 282, 0>), reinterpret_cast<const void*>(k_pass<36, 2, 282, 1>)};
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipFuncSetAttribute(ted[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds_bytes(kTED));
        if (e == hipSuccess) e = hipFuncSetAttribute(beat[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds_bytes(kBEAT));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// One launch = `nsamples` samples starting at a.b0, a.npass workgroups each.  No workgroup waits for another one, so the grid may be any
// size (two workgroups are resident per CU).
hipError_t launch_step_pass(Variant v, int prec, const StepArgs& a, int nsamples, hipStream_t st) {
    if (nsamples < 1 || (a.npass != 1 && a.npass != 2)) return hipErrorInvalidValue;
    const dim3 grid(nsamples * a.npass);
    if (v == kTED) {
        if (prec == 0) hipLaunchKernelGGL((k_pass<35, 1, 27, 0>), grid, dim3(kPassThreads), pass_lds_bytes(kTED), st, a);
        else hipLaunchKernelGGL((k_pass<35, 1, 27, 1>), grid, dim3(kPassThreads), pass_lds_bytes(kTED), st, a);
    } else {
        if (prec == 0) hipLaunchKernelGGL((k_pass<36, 2, 282, 0>), grid, dim3(kPassThreads), pass_lds_bytes(kBEAT), st, a);
        else hipLaunchKernelGGL((k_pass<36, 2, 282, 1>), grid, dim3(kPassThreads), pass_lds_bytes(kBEAT), st, a);
    }
    return hipGetLastError();
}

}  // namespace ls
