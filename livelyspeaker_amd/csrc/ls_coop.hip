// Instantiations and launch entry of the sample-split step kernel (ls_coop_kernel.h): TED (S = 35, J*F = 27) and BEAT (S = 36, J*F = 282),
// each with 8 | 4 | 2 slice workgroups per (sample, pass) (NCB = 1 | 2 | 4 sixteen-channel blocks per wave).
#include "ls_coop_kernel.h"

namespace ls {

size_t coop_lds_bytes() { return (size_t)kCoopLdsFloats * sizeof(float); }

// Opt in to > 64 KiB dynamic LDS once per process (outside stream capture).
hipError_t init_coop_kernels() {
    const void* ks[] = {reinterpret_cast<const void*>(k_coop<35, 1, 27, 1>), reinterpret_cast<const void*>(k_coop<36, 2, 282, 1>),
                        reinterpret_cast<const void*>(k_coop<35, 1, 27, 2>), reinterpret_cast<const void*>(k_coop<36, 2, 282, 2>),
                        reinterpret_cast<const void*>(k_coop<35, 1, 27, 4>), reinterpret_cast<const void*>(k_coop<36, 2, 282, 4>)};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds_bytes());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// One launch = `nsamples` samples starting at a.b0, a.npass passes each, 8 / ncb slice workgroups per (sample, pass).  The caller keeps
// the grid within what is resident at once (ncb = 1: two workgroups per CU; 2 and 4: one per CU): the slices of a group wait for each other.
hipError_t launch_step_coop(Variant v, int ncb, const StepArgs& a, int nsamples, hipStream_t st) {
    if (nsamples < 1 || (a.npass != 1 && a.npass != 2) || (ncb != 1 && ncb != 2 && ncb != 4)) return hipErrorInvalidValue;
    StepArgs c = a;
    c.ngroups = nsamples * a.npass;
    const int ns = 8 / ncb;
    if (ncb != 1 && c.xmap == 2) c.xmap = 0;
    const dim3 grid((c.xmap ? (c.ngroups + 7) / 8 * 8 : c.ngroups) * ns), block(kCoopThreads);      // xmap 1 / 2 deal whole sets of 8 groups over the XCDs
    const size_t lds = coop_lds_bytes();
    if (v == kTED) {
        if (ncb == 1) hipLaunchKernelGGL((k_coop<35, 1, 27, 1>), grid, block, lds, st, c);
        else if (ncb == 2) hipLaunchKernelGGL((k_coop<35, 1, 27, 2>), grid, block, lds, st, c);
        else hipLaunchKernelGGL((k_coop<35, 1, 27, 4>), grid, block, lds, st, c);
    } else {
        if (ncb == 1) hipLaunchKernelGGL((k_coop<36, 2, 282, 1>), grid, block, lds, st, c);
        else if (ncb == 2) hipLaunchKernelGGL((k_coop<36, 2, 282, 2>), grid, block, lds, st, c);
        else hipLaunchKernelGGL((k_coop<36, 2, 282, 4>), grid, block, lds, st, c);
    }
    return hipGetLastError();
}

}  // namespace ls
