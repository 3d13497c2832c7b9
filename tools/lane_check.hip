// Hardware check of the cross-lane helpers in csrc/ls_lanes.h against __shfl_xor (run on the GPU box: hipcc + ./a.out).
#include <cstdio>
#include <cmath>
#include "../livelyspeaker_amd/csrc/ls_lanes.h"
__global__ void k(const float* x, float* o) {
    const int l = threadIdx.x;
    const float v = x[l];
    float a = v; for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m);
    float w = v; for (int m = 1; m < 64; m <<= 1) w += __shfl_xor(w, m);
    o[l] = ls::row16_sum(v) - a;
    o[64 + l] = ls::wave_sum(v) - w;
    o[128 + l] = ls::xor32_sum(v) - (v + __shfl_xor(v, 32));
    o[192 + l] = ls::xor16_sum(v) - (v + __shfl_xor(v, 16));
    o[256 + l] = ls::xor32_get(v, l) - __shfl_xor(v, 32);
    o[320 + l] = ls::xor16_get(v, l) - __shfl_xor(v, 16);
    float mx = v; for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    o[384 + l] = ls::wave_max(v) - mx;
}
int main() {
    float hx[64], ho[448], *dx, *dout;
    for (int i = 0; i < 64; ++i) hx[i] = (float)(i * i % 37) + 0.25f * i;      // exactly representable, sums exact
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    const char* names[7] = {"row16_sum", "wave_sum", "xor32_sum", "xor16_sum", "xor32_get", "xor16_get", "wave_max"};
    int bad = 0;
    for (int t = 0; t < 7; ++t) {
        float m = 0; for (int i = 0; i < 64; ++i) m = fmaxf(m, fabsf(ho[64 * t + i]));
        printf("%s max |diff| = %g\n", names[t], m);
        bad += m != 0.f;
    }
    return bad;
}
