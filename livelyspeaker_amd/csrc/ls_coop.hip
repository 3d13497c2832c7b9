// Instantiations and launch entry of the sample-split step kernel (ls_coop_kernel.h): TED (S = 35, J*F = 27) and BEAT (S = 36, J*F = 282).
#include "ls_coop_kernel.h"

namespace ls {

size_t coop_lds_bytes() { return (size_t)kCoopLdsFloats * sizeof(float); }

// Opt in to > 64 KiB dynamic LDS once per process (outside stream capture).
hipError_t init_coop_kernels() {
    const void* ks[] = {reinterpret_cast<const void*>(k_coop<35, 1, 27>), reinterpret_cast<const void*>(k_coop<36, 2, 282>)};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds_bytes());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// One launch = `nsamples` samples starting at a.b0, a.npass passes each, 8 slice workgroups per (sample, pass).  The caller keeps the
// grid within what is resident at once (kCoopMaxGroups groups = 512 workgroups, two per CU): the slices of a group wait for each other.
hipError_t launch_step_coop(Variant v, const StepArgs& a, int nsamples, hipStream_t st) {
    if (nsamples < 1 || (a.npass != 1 && a.npass != 2)) return hipErrorInvalidValue;
    StepArgs c = a;
    c.ngroups = nsamples * a.npass;
    const dim3 grid((a.xmap ? (c.ngroups + 7) / 8 * 8 : c.ngroups) * kCoopSlices);
    if (v == kTED) hipLaunchKernelGGL((k_coop<35, 1, 27>), grid, dim3(kCoopThreads), coop_lds_bytes(), st, c);
    else hipLaunchKernelGGL((k_coop<36, 2, 282>), grid, dim3(kCoopThreads), coop_lds_bytes(), st, c);
    return hipGetLastError();
}

}  // namespace ls
