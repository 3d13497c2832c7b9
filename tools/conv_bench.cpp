// Stand-alone timing (and, with -DLS_CONV_PROF, a per-stage barrier timeline) of the WavEncoder's stride-6 conv layers on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DLS_CONV_PROF] -I livelyspeaker_amd/csrc -I include \
//         tools/conv_bench.cpp livelyspeaker_amd/csrc/ls_conv.hip -o variants/conv_bench
//   variants/conv_bench [B]    ->  us per launch / TFLOP/s of conv2, conv3, conv4 at batch B (default 512), a host spot check,
//                                  and (PROF) for one workgroup: per stage, when its consumer / producer wave reached and left the barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "ls_internal.h"

#ifdef LS_CONV_PROF
namespace ls { extern __device__ unsigned long long* g_conv_prof; extern __device__ int g_conv_prof_wg; }
#endif

static float frand(size_t i, unsigned m) { return (float)(((i * m + 12345u) >> 7) & 0xffff) / 65536.f - 0.5f; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? std::atoi(argv[1]) : 512;
    const int Cin[3] = {32, 64, 128}, Cout[3] = {64, 128, 256}, Lin[3] = {7891, 1313, 217}, Lout[3] = {1313, 217, 34};
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {   // conv1 (1 -> 32 channels, k 15, stride 5, padding 1600) + the InstanceNorm statistics of its output: bound by the 517 MB it writes
        const int lin = 36267, lout = 7891;
        float *dw, *dwt, *db, *dout, *dos, *dsp;
        hipMalloc(&dw, (size_t)B * lin * 4); hipMalloc(&dwt, 480 * 4); hipMalloc(&db, 32 * 4); hipMalloc(&dout, (size_t)B * 32 * lout * 4);
        hipMalloc(&dos, (size_t)B * 32 * 2 * 4); hipMalloc(&dsp, (size_t)B * 32 * 8 * 4 * 3 * 4);
        hipMemset(dw, 0, (size_t)B * lin * 4); hipMemset(dwt, 0, 480 * 4); hipMemset(db, 0, 32 * 4);
        for (int with = 1; with >= 0; --with) {
            auto run = [&]() { return ls::launch_conv1_fwd(dw, dwt, db, dout, with ? dos : nullptr, dsp, B, lin, lout, 1600, st); };
            for (int i = 0; i < 600; ++i) run();
            hipEventRecord(e0, st);
            for (int i = 0; i < 20; ++i) run();
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1000.0 / 20;
            std::printf("conv1 B=%d L %d->%d %s: %.1f us = %.2f TB/s written\n", B, lin, lout, with ? "with its output statistics" : "without statistics", us,
                        (double)B * 32 * lout * 4 / us * 1e-6);
        }
        hipFree(dw); hipFree(dwt); hipFree(db); hipFree(dout); hipFree(dos); hipFree(dsp);
    }
    for (int L = 0; L < 3; ++L) {
        const int ci = Cin[L], co = Cout[L], li = Lin[L], lo = Lout[L];
        const size_t nin = (size_t)B * ci * li, nout = (size_t)B * co * lo, nw = (size_t)co * ci * 15;
        std::vector<float> hin(nin), hw(nw), hb(co), hst((size_t)B * ci * 2), img(nw);
        for (size_t i = 0; i < nin; ++i) hin[i] = frand(i, 2654435761u);
        for (size_t i = 0; i < nw; ++i) hw[i] = frand(i, 40503u) * 0.2f;
        for (int i = 0; i < co; ++i) hb[i] = frand(i, 97u);
        for (size_t r = 0; r < (size_t)B * ci; ++r) { hst[2 * r] = frand(r, 31u) * 0.1f; hst[2 * r + 1] = 1.5f + frand(r, 17u); }
        // image [co tile][chunk][k][lane][cig] = W[co = 16 ct + (lane & 15)][ci = 16 chunk + 4 cig + (lane >> 4)][k]  (ls_api.cpp)
        size_t o = 0;
        for (int ct = 0; ct < co / 16; ++ct)
            for (int ch = 0; ch < ci / 16; ++ch)
                for (int k = 0; k < 15; ++k)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int cig = 0; cig < 4; ++cig)
                            img[o++] = hw[((size_t)(16 * ct + (lane & 15)) * ci + 16 * ch + 4 * cig + (lane >> 4)) * 15 + k];
        float *din, *dst, *dimg, *db, *dout, *dos, *dsp;
        hipMalloc(&din, nin * 4); hipMalloc(&dst, hst.size() * 4); hipMalloc(&dimg, nw * 4); hipMalloc(&db, co * 4); hipMalloc(&dout, nout * 4);
        hipMalloc(&dos, (size_t)B * co * 2 * 4); hipMalloc(&dsp, (size_t)B * co * ((lo + 63) / 64) * 12 * 4);
        hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(dst, hst.data(), hst.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dimg, img.data(), nw * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), co * 4, hipMemcpyHostToDevice);
        const bool stats = L < 2;
        auto run = [&]() { return ls::launch_conv1d_mfma(din, dst, dimg, db, dout, stats ? dos : nullptr, dsp, B, ci, co, li, lo, st); };
        // (400 untimed launches: the shader clock needs a few hundred ms of load to reach its 2.4 GHz ceiling, and the host check of
        // the previous layer lets it fall again)
        for (int i = 0; i < 400; ++i) if (run() != hipSuccess) { std::printf("launch failed\n"); return 1; }
        const int n = 20;
        hipEventRecord(e0, st);
        for (int i = 0; i < n; ++i) run();
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / n, fl = 2.0 * B * co * lo * ci * 15;
        std::printf("conv%d B=%d %dx%d L %d->%d: %.1f us (incl. stats merge)  %.1f TFLOP/s (%.3f of 157.3)\n", L + 2, B, ci, co, li, lo, us, fl / us * 1e-6,
                    fl / us * 1e-6 / 157.3);
        std::vector<float> hout(nout);
        hipMemcpy(hout.data(), dout, nout * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 2048; ++t) {
            const int b = (int)((t * 2654435761u) % (unsigned)B), c = (int)((t * 40503u + 17u) % (unsigned)co), p = (int)((t * 7919u + 3u) % (unsigned)lo);
            double ref = hb[c];
            for (int i = 0; i < ci; ++i) {
                const float m = hst[((size_t)b * ci + i) * 2], r = hst[((size_t)b * ci + i) * 2 + 1];
                for (int k = 0; k < 15; ++k) {
                    float v = (hin[((size_t)b * ci + i) * li + p * 6 + k] - m) * r;
                    v = v >= 0.f ? v : 0.3f * v;
                    ref += (double)hw[((size_t)c * ci + i) * 15 + k] * v;
                }
            }
            const double d = std::fabs(ref - hout[((size_t)b * co + c) * lo + p]);
            if (d > worst) worst = d;
        }
        std::printf("    max |out - host| over 2048 sampled entries: %.3g %s\n", worst, worst < 2e-3 ? "ok" : "MISMATCH");
#ifdef LS_CONV_PROF
        if (L < 2) {
            unsigned long long* dprof; hipMalloc(&dprof, 8 * 1024 * 2 * 8); hipMemset(dprof, 0, 8 * 1024 * 2 * 8);
            const int wg = B / 2 + 3;
            hipMemcpyToSymbol(HIP_SYMBOL(ls::g_conv_prof), &dprof, sizeof dprof);
            hipMemcpyToSymbol(HIP_SYMBOL(ls::g_conv_prof_wg), &wg, sizeof wg);
            run(); hipStreamSynchronize(st);
            std::vector<unsigned long long> p(8 * 1024 * 2);
            hipMemcpy(p.data(), dprof, p.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long* nul = nullptr;
            hipMemcpyToSymbol(HIP_SYMBOL(ls::g_conv_prof), &nul, sizeof nul);
            unsigned long long t0 = ~0ull;
            for (int wv = 0; wv < 8; ++wv) if (p[wv * 2048] && p[wv * 2048] < t0) t0 = p[wv * 2048];
            std::printf("    barrier timeline of workgroup z=%d: per stage, when each wave REACHED the barrier (cycles before its release; waves 0-3 multiply, 4-7 stage) | release time\n", wg);
            double wait[8] = {0}; int ns = 0; unsigned long long last = 0;
            for (int s = 0; s < 1024 && p[2 * s]; ++s) {
                unsigned long long rel = 0;
                for (int wv = 0; wv < 8; ++wv) if (p[wv * 2048 + 2 * s] > rel) rel = p[wv * 2048 + 2 * s];   // the last arrival releases it
                if (s < 16) {
                    std::printf("      stage %3d:", s);
                    for (int wv = 0; wv < 8; ++wv) std::printf(" %6lld", (long long)(rel - p[wv * 2048 + 2 * s]));
                    std::printf(" | %8lld (+%lld)\n", (long long)(rel - t0), (long long)(rel - (last ? last : t0)));
                }
                for (int wv = 0; wv < 8; ++wv) wait[wv] += (double)(rel - p[wv * 2048 + 2 * s]);
                last = rel; ++ns;
            }
            if (ns) {
                std::printf("    %d stages, mean stage %.0f cycles; mean wait at the barrier per wave:", ns, (double)(last - t0) / ns);
                for (int wv = 0; wv < 8; ++wv) std::printf(" %.0f", wait[wv] / ns);
                std::printf("\n");
            }
            hipFree(dprof);
        }
#endif
        hipFree(din); hipFree(dst); hipFree(dimg); hipFree(db); hipFree(dout); hipFree(dos); hipFree(dsp);
    }
    return 0;
}
