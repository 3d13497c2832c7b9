// One-off probe: which workgroups of a 512-thread / 76.7 KB-LDS launch share a CU on MI355X?  (tools/probes, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out, unsigned long long* t) {
    extern __shared__ float s[];
    s[threadIdx.x] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        t[blockIdx.x] = __builtin_amdgcn_s_memtime();
    }
    // stay resident long enough for the whole grid to be placed
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) __builtin_amdgcn_s_sleep(10);
    if (s[threadIdx.x] == 0.f) out[0] = 0;
}
int main(int argc, char** argv) {
    int nwg = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* d; unsigned long long* dt;
    hipMalloc(&d, nwg * 8); hipMalloc(&dt, nwg * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 78000);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 76700, 0, d, dt);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * nwg);
    hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < nwg; ++i) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        printf("bid %3d xcc %u se %u sh %u cu %2u  key %u\n", i, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (xcc << 8) | ((hw >> 8) & 0xff));
    }
    return 0;
}
