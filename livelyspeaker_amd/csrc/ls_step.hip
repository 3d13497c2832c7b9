// TED instantiations (S = 35: style token + 34 frames, J*F = 27) of the fused step kernel (ls_step_kernel.h) and the
// variant-independent launch entry points.  BEAT lives in ls_step_beat.hip so the two sets compile in parallel.
#include "ls_step_kernel.h"

namespace ls {

size_t step_lds_bytes(Variant v) {
    const int S = (v == kTED) ? 35 : 36;
    const int nrem = 2 * S - 64;
    return (size_t)(2 * S * kUStride + 2 * kWaves * 16 * kNT + kWaves * 2 * nrem * 16) * sizeof(float);
}

// Opt in to >64 KiB dynamic LDS once per process (must happen outside stream capture).
hipError_t init_step_kernels_ted() {
    const void* ks[] = {reinterpret_cast<const void*>(k_step<35, 1, 27, 0>), reinterpret_cast<const void*>(k_step<35, 1, 27, 1>),
                        reinterpret_cast<const void*>(k_step<35, 1, 27, 0, 1>), reinterpret_cast<const void*>(k_step<35, 1, 27, 0, 0, 1>),
                        reinterpret_cast<const void*>(k_step<35, 1, 27, 1, 0, 1>)};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds_bytes(kTED));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t init_step_kernels() {
    hipError_t e = init_step_kernels_ted();
    return e != hipSuccess ? e : init_step_kernels_beat();
}

hipError_t launch_step_ted(int prec, int pair, const StepArgs& a, int batch, hipStream_t st) {
    const size_t lds = step_lds_bytes(kTED);
    if (pair) {
        const dim3 grid((batch + 1) / 2);
        if (prec == 0) hipLaunchKernelGGL((k_step<35, 1, 27, 0, 0, 1>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((k_step<35, 1, 27, 1, 0, 1>), grid, dim3(512), lds, st, a);
    } else if (prec == 0) {
        hipLaunchKernelGGL((k_step<35, 1, 27, 0>), dim3(batch), dim3(512), lds, st, a);
    } else {
        hipLaunchKernelGGL((k_step<35, 1, 27, 1>), dim3(batch), dim3(512), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_step(Variant v, int prec, int pair, const StepArgs& a, int batch, hipStream_t st) {
    return v == kTED ? launch_step_ted(prec, pair, a, batch, st) : launch_step_beat(prec, pair, a, batch, st);
}

hipError_t launch_train_mixer_fwd_beat(const StepArgs& a, hipStream_t st);
hipError_t launch_train_mixer_fwd(Variant v, const StepArgs& a, hipStream_t st) {
    if (v != kTED) return launch_train_mixer_fwd_beat(a, st);
    hipLaunchKernelGGL((k_step<35, 1, 27, 0, 1>), dim3((a.tr_B + 1) / 2), dim3(512), step_lds_bytes(kTED), st, a);
    return hipGetLastError();
}

}  // namespace ls
