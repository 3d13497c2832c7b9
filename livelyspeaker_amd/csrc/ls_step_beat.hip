// BEAT instantiations (S = 36: style + emotion token + 34 frames, J*F = 282) of the fused step kernel (ls_step_kernel.h).
#include "ls_step_kernel.h"

namespace ls {

hipError_t init_step_kernels_beat() {
    const void* ks[] = {reinterpret_cast<const void*>(k_step<36, 2, 282, 0>), reinterpret_cast<const void*>(k_step<36, 2, 282, 1>),
                        reinterpret_cast<const void*>(k_step<36, 2, 282, 0, 1>), reinterpret_cast<const void*>(k_step<36, 2, 282, 0, 0, 1>),
                        reinterpret_cast<const void*>(k_step<36, 2, 282, 1, 0, 1>)};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds_bytes(kBEAT));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_step_beat(int prec, int pair, const StepArgs& a, int batch, hipStream_t st) {
    const size_t lds = step_lds_bytes(kBEAT);
    if (pair) {
        const dim3 grid((batch + 1) / 2);
        if (prec == 0) hipLaunchKernelGGL((k_step<36, 2, 282, 0, 0, 1>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((k_step<36, 2, 282, 1, 0, 1>), grid, dim3(512), lds, st, a);
    } else if (prec == 0) {
        hipLaunchKernelGGL((k_step<36, 2, 282, 0>), dim3(batch), dim3(512), lds, st, a);
    } else {
        hipLaunchKernelGGL((k_step<36, 2, 282, 1>), dim3(batch), dim3(512), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_train_mixer_fwd_beat(const StepArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((k_step<36, 2, 282, 0, 1>), dim3((a.tr_B + 1) / 2), dim3(512), step_lds_bytes(kBEAT), st, a);
    return hipGetLastError();
}

}  // namespace ls
