// Instantiation and launch entry of the long-sequence mixer kernel (ls_mix_kernel.h): S = 152 (150 frames + 2 prefix tokens, the synthetic
// BEAT150 shape); other lengths keep the batch-level kernels of ls_long.hip.
#include "ls_mix_kernel.h"

namespace ls {

size_t mix_lds_bytes() { return (size_t)kMixLdsFloats * sizeof(float); }
bool mix_supports(int S) { return S == 152; }

hipError_t init_mix_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_mix<152>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_lds_bytes());
}

// One launch = a.ngroups (sample, pass) groups, four slice workgroups each, all resident at once (the caller keeps 4 * ngroups <= CUs)
hipError_t launch_mix(int S, const MixArgs& a, hipStream_t st) {
    if (!mix_supports(S) || a.ngroups < 1) return hipErrorInvalidValue;
    const dim3 grid((a.ngroups + 7) / 8 * 8 * kMixSlices), block(kMixThreads);
    hipLaunchKernelGGL((k_mix<152>), grid, block, mix_lds_bytes(), st, a);
    return hipGetLastError();
}

}  // namespace ls
