// Stand-alone timing of launch_gemm_nt (the nn.Linear-shaped fp32 MFMA GEMM) on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -x hip -I livelyspeaker_amd/csrc -I include tools/gemm_bench.cpp livelyspeaker_amd/csrc/ls_gemm.hip -o variants/gemm_bench
//   variants/gemm_bench M N K [act] [residual]   ->  us per launch and TFLOP/s (HIP events over 50 launches, after 1500 untimed ones: the shader clock needs a few hundred ms of load to reach its ceiling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "ls_internal.h"
#include "ls_train.h"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: gemm_bench M N K [act] [residual]\n"); return 2; }
    const int M = std::atoi(argv[1]), N = std::atoi(argv[2]), K = std::atoi(argv[3]);
    const int act = argc > 4 ? std::atoi(argv[4]) : 0, res = argc > 5 ? std::atoi(argv[5]) : 0;
    float *A, *W, *C, *b, *R;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&b, N * 4);
    hipMalloc(&R, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K), hw((size_t)N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u + 977u >> 4) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hb(N), hr((size_t)M * N);
    for (int i = 0; i < N; ++i) hb[i] = (float)((i * 2246822519u >> 9) & 0xffff) / 65536.f - 0.5f;
    for (size_t i = 0; i < hr.size(); ++i) hr[i] = (float)((i * 3266489917u >> 7) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(R, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 1500; ++i) ls::launch_gemm_nt(A, K, W, K, b, res ? R : nullptr, N, C, N, M, N, K, act, st);
    const int n = 50;
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) ls::launch_gemm_nt(A, K, W, K, b, res ? R : nullptr, N, C, N, M, N, K, act, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / n;
    std::printf("gemm M=%d N=%d K=%d act=%d res=%d: %.1f us  %.1f TFLOP/s (%.3f of 157.3)\n", M, N, K, act, res, us, 2.0 * M * N * K / us * 1e-6,
                2.0 * M * N * K / us * 1e-6 / 157.3);
    {      // spot check against a host evaluation of act(dot + bias) + residual
        std::vector<float> c((size_t)M * N);
        hipMemcpy(c.data(), C, c.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 4096; ++t) {
            const int m = (int)((t * 2654435761u) % (unsigned)M), n = (int)((t * 40503u + 17u) % (unsigned)N);
            double ref = hb[n];
            for (int k = 0; k < K; ++k) ref += (double)h[(size_t)m * K + k] * hw[(size_t)n * K + k];
            if (act == 1) ref = ref / (1.0 + std::exp(-ref));
            if (act == 2) ref = std::exp(0.5 * ref);
            if (act == 3) ref = 0.5 * ref * (1.0 + std::erf(ref * 0.70710678118654752));
            if (res) ref += hr[(size_t)m * N + n];
            const double d = std::fabs(ref - c[(size_t)m * N + n]);
            if (d > worst) worst = d;
        }
        std::printf("    max |C - host| over 4096 sampled entries: %.3g %s\n", worst, worst < 1e-3 ? "ok" : "MISMATCH");
    }
    return 0;
}
