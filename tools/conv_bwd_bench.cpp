// Stand-alone timing + host spot checks of the WavEncoder's BACKWARD conv kernels (training step) on the GPU box:
//   tools/build_conv_bench.sh            (builds variants/conv_bwd_bench next to variants/conv_bench)
//   variants/conv_bwd_bench [B]      ->  us per launch / TFLOP/s of the weight and data gradients of conv2, conv3, conv4 at batch B
//                                        (default 512) with sampled entries checked against a host evaluation in double, and conv1's
//                                        weight gradient (timing; its parity is tests/test_gpu_train.py's)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "ls_internal.h"
#include "ls_train.h"


static float frand(size_t i, unsigned m) { return (float)(((i * m + 12345u) >> 7) & 0xffff) / 65536.f - 0.5f; }

template <class F>
static double time_us(hipStream_t st, F&& run, int n = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // the shader clock needs a few hundred ms of load to reach its ceiling (rocm-smi: 109 MHz idle, 2.4 GHz under bench.py); a timing
    // that starts cold reads up to 20 % long
    for (int i = 0; i < 300; ++i) run();              // (every call: the host checks between two timings let the clock fall again)
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) run();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1000.0 / n;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? std::atoi(argv[1]) : 512;
    const int Cin[3] = {32, 64, 128}, Cout[3] = {64, 128, 256}, Lin[3] = {7891, 1313, 217}, Lout[3] = {1313, 217, 34};
    hipStream_t st; hipStreamCreate(&st);
    for (int L = 0; L < 3; ++L) {
        const int ci = Cin[L], co = Cout[L], li = Lin[L], lo = Lout[L], W = ci * 15;
        const size_t nin = (size_t)B * ci * li, ndc = (size_t)B * co * lo, nw = (size_t)co * W;
        std::vector<float> hin(nin), hdc(ndc), hst((size_t)B * ci * 2), hw(nw);
        for (size_t i = 0; i < nin; ++i) hin[i] = frand(i, 2654435761u);
        for (size_t i = 0; i < ndc; ++i) hdc[i] = frand(i, 40503u) * 0.25f;
        for (size_t i = 0; i < nw; ++i) hw[i] = frand(i, 7919u) * 0.2f;
        for (size_t r = 0; r < (size_t)B * ci; ++r) { hst[2 * r] = frand(r, 31u) * 0.1f; hst[2 * r + 1] = 1.5f + frand(r, 17u); }
        // conv4's dC arrives channel-contiguous ([b][p][co], what ls_train_api.cpp hands over); conv2 / conv3 position-contiguous
        const bool chan = false;   // (conv4: ls_train_api.cpp transposes the [b][p][co] gradient it receives first)
        const long long sb = (long long)co * lo, sc = chan ? 1 : lo, sp = chan ? co : 1;
        auto DC = [&](int b, int c, int p) { return hdc[(size_t)b * sb + (size_t)c * sc + (size_t)p * sp]; };
        auto ACT = [&](int b, int i, int x) {
            float v = (hin[((size_t)b * ci + i) * li + x] - hst[((size_t)b * ci + i) * 2]) * hst[((size_t)b * ci + i) * 2 + 1];
            return v >= 0.f ? v : 0.3f * v;
        };
        float *din, *ddc, *dst, *dw, *dpart, *dgw, *dimg, *dout, *drow;
        const int ngmax = ls::conv_wgrad_groups(ci, co);
        hipMalloc(&din, nin * 4); hipMalloc(&ddc, ndc * 4); hipMalloc(&dst, hst.size() * 4); hipMalloc(&dw, nw * 4); hipMalloc(&dgw, nw * 4);
        hipMalloc(&dpart, (size_t)ngmax * nw * 4); hipMalloc(&dimg, (size_t)(ci / 16) * (co / 4) * 16 * 64 * 4); hipMalloc(&dout, nin * 4);
        hipMalloc(&drow, (size_t)B * ci * (((li + 5) / 6 + 63) / 64) * 2 * 2 * 4 + 4096);
        hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(ddc, hdc.data(), ndc * 4, hipMemcpyHostToDevice);
        hipMemcpy(dst, hst.data(), hst.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
        const double fl = 2.0 * B * co * lo * ci * 15;

        // ---- weight gradient ----
        int ng = 0;
        auto wg = [&]() {
            ls::launch_conv_wgrad(ddc, sb, sc, din, dst, dpart, B, ci, co, li, lo, &ng, st);
            ls::launch_partial_reduce(dpart, ng, (long long)nw, (int)nw, dgw, 0, st);
        };
        double us = time_us(st, wg);
        std::printf("conv%d wgrad B=%d %dx%d L %d->%d (%d runs): %.1f us incl. reduce  %.1f TFLOP/s (%.3f of 157.3)\n", L + 2, B, ci, co, li, lo, ng, us,
                    fl / us * 1e-6, fl / us * 1e-6 / 157.3);
        {
            std::vector<float> g(nw);
            hipMemcpy(g.data(), dgw, nw * 4, hipMemcpyDeviceToHost);
            double worst = 0, scale = 0;
            for (int t = 0; t < 48; ++t) {
                const int c = (int)((t * 2654435761u) % (unsigned)co), i = (int)((t * 40503u + 5u) % (unsigned)ci), k = (int)((t * 7u + 3u) % 15u);
                double ref = 0;
                for (int b = 0; b < B; ++b)
                    for (int p = 0; p < lo; ++p) ref += (double)DC(b, c, p) * ACT(b, i, 6 * p + k);
                const double d = std::fabs(ref - g[((size_t)c * ci + i) * 15 + k]);
                if (d > worst) worst = d;
                if (std::fabs(ref) > scale) scale = std::fabs(ref);
            }
            std::printf("    max |dW - host| over 48 sampled entries: %.3g (largest entry %.3g) %s\n", worst, scale, worst < 2e-4 * scale + 1e-3 ? "ok" : "MISMATCH");
        }


        // ---- data gradient (+ LeakyReLU', row partials; finalize pass timed with it as the training step runs it for conv3 / conv4) ----
        ls::launch_build_dgrad_img(dw, dimg, ci, co, st);
        int nslot = 0;
        auto dg = [&](bool fin) { ls::launch_conv_dgrad(ddc, sb, sc, sp, dimg, din, dst, dout, drow, B, ci, co, li, lo, fin, &nslot, st); };
        us = time_us(st, [&]() { dg(false); });
        const double us_fin = time_us(st, [&]() { dg(true); });
        std::printf("conv%d dgrad: %.1f us  %.1f TFLOP/s (%.3f of 157.3); with the InstanceNorm finalize pass %.1f us\n", L + 2, us, fl / us * 1e-6,
                    fl / us * 1e-6 / 157.3, us_fin);
        {
            dg(false);
            hipStreamSynchronize(st);
            std::vector<float> o(nin);
            hipMemcpy(o.data(), dout, nin * 4, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int t = 0; t < 2048; ++t) {
                const int b = (int)((t * 2654435761u) % (unsigned)B), i = (int)((t * 40503u + 17u) % (unsigned)ci), x = (int)((t * 7919u + 3u) % (unsigned)li);
                double ref = 0;
                for (int k = 0; k < 15; ++k) {
                    if ((x - k) % 6 || x - k < 0) continue;
                    const int p = (x - k) / 6;
                    if (p >= lo) continue;
                    for (int c = 0; c < co; ++c) ref += (double)hw[((size_t)c * ci + i) * 15 + k] * DC(b, c, p);
                }
                const float y = (hin[((size_t)b * ci + i) * li + x] - hst[((size_t)b * ci + i) * 2]) * hst[((size_t)b * ci + i) * 2 + 1];
                if (y < 0.f) ref *= 0.3;
                const double d = std::fabs(ref - o[((size_t)b * ci + i) * li + x]);
                if (d > worst) worst = d;
            }
            std::printf("    max |dy - host| over 2048 sampled entries: %.3g %s\n", worst, worst < 2e-3 ? "ok" : "MISMATCH");
        }
        if (L == 0) {
            // conv1's weight gradient consumes conv2's dgrad output (dy + row partials) and the raw conv1 output (here: din)
            const int lw = (li - 1) * 5 + 15 - 3200;
            float *dwav, *dp1, *dg1;
            hipMalloc(&dwav, (size_t)B * lw * 4); hipMalloc(&dp1, ((size_t)B * ((li + 255) / 256) * 480 + (size_t)B * 128) * 4); hipMalloc(&dg1, 480 * 4);
            hipMemset(dwav, 0, (size_t)B * lw * 4);
            int nchunk = 0;
            us = time_us(st, [&]() {
                ls::launch_conv1_wgrad(dout, din, dst, drow, nslot, dwav, dp1, B, lw, li, 5, 1600, &nchunk, st);
                ls::launch_partial_reduce(dp1, B * nchunk, 480, 480, dg1, 0, st);
            });
            std::printf("conv1 wgrad (reads dy + c_raw: %.0f MB): %.1f us = %.2f TB/s\n", 2.0 * nin * 4 / 1e6, us, 2.0 * nin * 4 / us * 1e-6);
            // the same two results from the fused form (what the training step runs): no dy tensor, conv1's gradient out of the epilogue
            {
                std::vector<float> hwav((size_t)B * lw), hw1(480), hb1(32);
                for (size_t i = 0; i < hwav.size(); ++i) hwav[i] = frand(i, 104729u);
                for (int i = 0; i < 480; ++i) hw1[i] = frand(i, 1299709u) * 0.3f;
                for (int i = 0; i < 32; ++i) hb1[i] = frand(i, 15485863u) * 0.1f;
                hipMemcpy(dwav, hwav.data(), hwav.size() * 4, hipMemcpyHostToDevice);
                float *dw1, *db1, *dmom, *dwork, *dgf;
                hipMalloc(&dw1, 480 * 4); hipMalloc(&db1, 32 * 4); hipMalloc(&dmom, (size_t)B * ls::wav_moment_parts(li) * 256 * 4); hipMalloc(&dgf, 480 * 4);
                hipMalloc(&dwork, (size_t)B * (2 * (((li + 5) / 6 + 63) / 64) * 512 + 608) * 4);
                hipMemcpy(dw1, hw1.data(), 480 * 4, hipMemcpyHostToDevice); hipMemcpy(db1, hb1.data(), 32 * 4, hipMemcpyHostToDevice);
                // c_raw consistent with the waveform (the fused form derives sum c_raw * wav from the weights): overwrite din
                std::vector<float> hc((size_t)B * 32 * li);
                for (int b = 0; b < B; ++b)
                    for (int c = 0; c < 32; ++c)
                        for (int p = 0; p < li; ++p) {
                            float v = hb1[c];
                            for (int k = 0; k < 15; ++k) {
                                const int x = 5 * p + k - 1600;
                                if (x >= 0 && x < lw) v = fmaf(hw1[c * 15 + k], hwav[(size_t)b * lw + x], v);
                            }
                            hc[((size_t)b * 32 + c) * li + p] = v;
                        }
                hipMemcpy(din, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
                // reference: the unfused pair on the same inputs
                dg(false);
                ls::launch_conv1_wgrad(dout, din, dst, drow, nslot, dwav, dp1, B, lw, li, 5, 1600, &nchunk, st);
                ls::launch_partial_reduce(dp1, B * nchunk, 480, 480, dg1, 0, st);
                float* outp = nullptr;
                auto fused = [&]() {
                    ls::launch_wav_moments(dwav, dmom, B, lw, li, 1600, st);
                    ls::launch_conv_dgrad_conv1(ddc, sb, sc, sp, dimg, din, dst, drow, B, co, li, lo, dwav, lw, 1600, dmom, dw1, db1, dwork, &outp, st);
                    ls::launch_partial_reduce(outp, B, 480, 480, dgf, 0, st);
                };
                const double usf = time_us(st, fused);
                {
                    const double usk = time_us(st, [&]() { ls::launch_conv_dgrad_conv1(ddc, sb, sc, sp, dimg, din, dst, drow, B, co, li, lo, dwav, lw, 1600, dmom, dw1, db1, dwork, &outp, st); });
                    const double usm = time_us(st, [&]() { ls::launch_wav_moments(dwav, dmom, B, lw, li, 1600, st); });
                    std::printf("    parts: fused dgrad + coefficients + finish %.1f us, waveform moments %.1f us\n", usk, usm);
                }
                std::vector<float> g1(480), g2(480);
                hipMemcpy(g1.data(), dg1, 480 * 4, hipMemcpyDeviceToHost); hipMemcpy(g2.data(), dgf, 480 * 4, hipMemcpyDeviceToHost);
                double worst = 0, scale = 0;
                for (int i = 0; i < 480; ++i) { worst = std::fmax(worst, std::fabs((double)g1[i] - g2[i])); scale = std::fmax(scale, std::fabs((double)g1[i])); }
                std::printf("conv2 dgrad + conv1 wgrad FUSED (waveform moments + epilogue products + finish + batch sum): %.1f us; max |dW1 - unfused| %.3g (largest entry %.3g) %s\n",
                            usf, worst, scale, worst < 3e-4 * scale ? "ok" : "MISMATCH");
                hipFree(dw1); hipFree(db1); hipFree(dmom); hipFree(dwork); hipFree(dgf);
            }
            hipFree(dwav); hipFree(dp1); hipFree(dg1);
        }
        hipFree(din); hipFree(ddc); hipFree(dst); hipFree(dw); hipFree(dpart); hipFree(dgw); hipFree(dimg); hipFree(dout); hipFree(drow);
    }
    return 0;
}
