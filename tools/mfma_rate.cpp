// Matrix-pipe issue rate of v_mfma_f32_16x16x4_f32 against the number of waves per SIMD that issue it and the number of independent
// accumulators per wave:  hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.cpp -o variants/mfma_rate && variants/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, bool SAMEAB>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(SAMEAB ? a[0] : a[(i + r) & 7], SAMEAB ? b[0] : b[(i * 3 + r) & 7], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, bool SAMEAB>
static void run(int threads, float* d, unsigned long long* dc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, SAMEAB>), dim3(256), dim3(threads), 0, 0, d, iters, dc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SAMEAB>), dim3(256), dim3(threads), 0, 0, d, iters, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double per_simd = (double)iters * 4 * NACC * (threads / 64) / 4.0;      // MFMAs per SIMD
    std::printf("  %d waves/SIMD, %2d accumulators, %s operands: %.1f cycles per MFMA and SIMD (counter), %.1f TFLOP/s\n", threads / 256, NACC,
                SAMEAB ? "fixed" : "rotating", (double)c / per_simd, 256.0 * (threads / 64) * iters * 4 * NACC * 2048.0 / (ms * 1e-3) * 1e-12);
}

int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    unsigned long long* dc; hipMalloc(&dc, 8);
    for (int threads : {256, 512, 768, 1024}) {
        run<8, false>(threads, d, dc);
        run<16, false>(threads, d, dc);
        run<8, true>(threads, d, dc);
    }
    return 0;
}
