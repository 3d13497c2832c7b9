// What the fp32 matrix pipe sustains on this box: every SIMD of every CU issuing independent v_mfma_f32_16x16x4_f32 back to back
// (no memory traffic), at 1, 2 and 4 waves per SIMD, for ~2 ms -- the clock the chip holds under that load is part of the answer.
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/mfma_peak.cpp -o variants/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, int random_data) {
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    // operands with full-entropy mantissas (what real activations / weights toggle in the datapath), eight of each per lane
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        unsigned h = (threadIdx.x * 8 + i) * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a[i] = random_data ? (float)(int)(h & 0xffffff) * (1.0f / 8388608.f) - 1.0f : threadIdx.x * 1e-9f;
        h *= 3266489917u; h ^= h >> 16;
        b[i] = random_data ? ((float)(int)(h & 0xffffff) * (1.0f / 8388608.f) - 1.0f) * 0.01f : 1.0f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + r) & 7], b[(i + 3 * r) & 7], acc[i % NACC], 0, 0, 0);
    }
    f4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int random_data : {0, 1})
    for (int wgs_per_cu : {1, 2, 4}) {
        const int grid = 256 * wgs_per_cu, iters = 20000 / wgs_per_cu;
        hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, out, iters, random_data);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, out, iters, random_data);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * 4 * iters * 32 * 2048.0;
        std::printf("%s operands, %d waves/SIMD: %.3f ms  %.1f TFLOP/s (%.3f of 157.3) = %.2f GHz if the pipe never idles\n", random_data ? "random" : "trivial",
                    wgs_per_cu, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3, flop / (ms * 1e-3) / (256.0 * 256.0) * 1e-9);
    }
    // dependency distance: the same instruction count with 1 / 2 / 4 accumulators in rotation (1 = every MFMA waits for the previous one)
    auto dist = [&](auto kern, int nacc) {
        hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, 10000, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, 10000, 1);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = 512.0 * 4 * 10000 * 32 * 2048.0;
        std::printf("2 waves/SIMD, %d accumulator(s) in rotation: %.3f ms  %.1f TFLOP/s (%.3f)\n", nacc, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
    };
    dist(k_mfma<1>, 1); dist(k_mfma<2>, 2); dist(k_mfma<4>, 4);
    return 0;
}
