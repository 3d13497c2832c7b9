// Once-per-sampling-call kernels for gfx950: everything in RAG.forward that depends on neither x_t nor
// t and that the reference recomputes 2x per step:
//   WavEncoder (4 x Conv1d k15 + InstanceNorm1d + LeakyReLU 0.3)     scripts/model/audio_enc.py:6-25
//   static part of input_mapping, speaker mu / logvar                scripts/model/RAG.py:110-120
//   timestep-embedding table                                         scripts/model/mlp_module.py:123-136
// plus layout conversion, q_sample and the Philox x_T fill.
#include "ls_internal.h"
#include "ls_philox.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Direct Conv1d, kernel 15.  Workgroup = (64 output positions) x (32 output channels) of one sample;
// lane = position, wave = 8 output channels.  Input window + weight slab are staged in LDS per chunk
// of input channels; the previous layer's InstanceNorm + LeakyReLU(0.3) is applied while staging
// (stats = per-(sample,channel) {mean, rstd}), so normalised activations are never written to HBM.
// ---------------------------------------------------------------------------------------------
constexpr int kConvK = 15;
constexpr int kConvTP = 64;
constexpr int kConvTC = 32;
constexpr int kConvCI = 16;   // input channels per LDS chunk

template <int STRIDE>
__global__ __launch_bounds__(256) void k_conv1d(const float* __restrict__ in, const float* __restrict__ stats,
                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ out, int Cin, int Cout, int Lin, int Lout,
                                                int pad) {
    constexpr int WIN = (kConvTP - 1) * STRIDE + kConvK;          // input window per tile
    constexpr int WINP = WIN + 1;
    __shared__ float sIn[kConvCI * WINP];
    __shared__ __attribute__((aligned(16))) float sW[kConvCI * kConvK * kConvTC];   // [ci*15+k][32 channels]

    const int b = blockIdx.z;
    const int co0 = blockIdx.y * kConvTC;
    const int p0 = blockIdx.x * kConvTP;
    const int tid = threadIdx.x;
    const int p = tid & 63;
    const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);      // channels co0 + 8*cg .. +7
    const int in0 = p0 * STRIDE - pad;

    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;

    for (int ci0 = 0; ci0 < Cin; ci0 += kConvCI) {
        const int nci = (Cin - ci0 < kConvCI) ? Cin - ci0 : kConvCI;
        __syncthreads();
        for (int idx = tid; idx < nci * WIN; idx += 256) {
            const int ci = idx / WIN, o = idx - ci * WIN;
            const int gi = in0 + o;
            float v = 0.f;
            if (gi >= 0 && gi < Lin) {
                v = in[((size_t)b * Cin + ci0 + ci) * Lin + gi];
                if (stats) {   // InstanceNorm1d(affine=False, eps 1e-5, biased var) + LeakyReLU(0.3), audio_enc.py:10-11
                    const float m = stats[((size_t)b * Cin + ci0 + ci) * 2], r = stats[((size_t)b * Cin + ci0 + ci) * 2 + 1];
                    v = (v - m) * r;
                    v = v >= 0.f ? v : 0.3f * v;
                }
            }
            sIn[ci * WINP + o] = v;
        }
        for (int idx = tid; idx < nci * kConvK * kConvTC; idx += 256) {
            const int c = idx & (kConvTC - 1), ck = idx / kConvTC;     // ck = ci*15 + k
            const int ci = ck / kConvK, k = ck - ci * kConvK;
            float v = 0.f;
            if (co0 + c < Cout) v = w[((size_t)(co0 + c) * Cin + ci0 + ci) * kConvK + k];
            sW[ck * kConvTC + c] = v;
        }
        __syncthreads();
        for (int ci = 0; ci < nci; ++ci) {
#pragma unroll
            for (int k = 0; k < kConvK; ++k) {
                const float v = sIn[ci * WINP + p * STRIDE + k];
                const f4 w0 = *reinterpret_cast<const f4*>(&sW[(ci * kConvK + k) * kConvTC + cg * 8]);
                const f4 w1 = *reinterpret_cast<const f4*>(&sW[(ci * kConvK + k) * kConvTC + cg * 8 + 4]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = fmaf(v, w0[c], acc[c]);
                    acc[4 + c] = fmaf(v, w1[c], acc[4 + c]);
                }
            }
        }
    }
    if (p0 + p < Lout) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int co = co0 + cg * 8 + c;
            if (co < Cout) out[((size_t)b * Cout + co) * Lout + p0 + p] = acc[c] + bias[co];
        }
    }
}

hipError_t launch_conv1d(const float* in, const float* stats, const float* w, const float* bias, float* out,
                         int B, int Cin, int Cout, int Lin, int Lout, int stride, int pad, hipStream_t st) {
    dim3 grid((Lout + kConvTP - 1) / kConvTP, (Cout + kConvTC - 1) / kConvTC, B);
    if (stride == 5)
        hipLaunchKernelGGL(k_conv1d<5>, grid, dim3(256), 0, st, in, stats, w, bias, out, Cin, Cout, Lin, Lout, pad);
    else if (stride == 6)
        hipLaunchKernelGGL(k_conv1d<6>, grid, dim3(256), 0, st, in, stats, w, bias, out, Cin, Cout, Lin, Lout, pad);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// per-row mean and 1/sqrt(biased var + 1e-5) over L (two-pass, like torch's InstanceNorm on fp32)
__global__ __launch_bounds__(256) void k_instnorm_stats(const float* __restrict__ x, float* __restrict__ stats, int L) {
    __shared__ float red[4];
    const size_t rowi = blockIdx.x;
    const float* xr = x + rowi * L;
    const int tid = threadIdx.x;
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    float s = 0.f;
    for (int i = tid; i < L; i += 256) s += xr[i];
    const float mean = block_sum(s) / (float)L;
    float q = 0.f;
    for (int i = tid; i < L; i += 256) {
        const float d = xr[i] - mean;
        q += d * d;
    }
    const float var = block_sum(q) / (float)L;
    if (tid == 0) {
        stats[rowi * 2] = mean;
        stats[rowi * 2 + 1] = 1.0f / sqrtf(var + 1e-5f);
    }
}

hipError_t launch_instnorm_stats(const float* x, float* stats, int rows, int L, hipStream_t st) {
    hipLaunchKernelGGL(k_instnorm_stats, dim3(rows), dim3(256), 0, st, x, stats, L);
    return hipGetLastError();
}

__global__ void k_gather_rows(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out,
                              int rows, int width, int table_rows) {
    const int r = blockIdx.x;
    long long i = idx[r];
    if (i < 0) i = 0;
    if (i >= table_rows) i = table_rows - 1;
    for (int c = threadIdx.x; c < width; c += blockDim.x) out[(size_t)r * width + c] = table[(size_t)i * width + c];
}

hipError_t launch_gather_rows(const float* table, const int64_t* idx, float* out, int rows, int width,
                              int table_rows, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_rows, dim3(rows), dim3(256), 0, st, table, idx, out, rows, width, table_rows);
    return hipGetLastError();
}

// InputProcess features without the x_t columns (RAG.py:110-112, 184-192):
// row (b,f) = [origin_x[b,:,f] if f < n_pre_seq else 0 | indicator bit | audio feature (cond) or 0 (uncond)]
__global__ void k_build_feats(const float* __restrict__ origin_x, const float* __restrict__ conv4,
                              float* __restrict__ feat_c, float* __restrict__ feat_u, int JF, int n_pre_seq) {
    const int b = blockIdx.x / kT, f = blockIdx.x % kT;
    const int KF = JF + 1 + kAudioFeat;
    float* fc = feat_c + (size_t)blockIdx.x * KF;
    float* fu = feat_u + (size_t)blockIdx.x * KF;
    for (int c = threadIdx.x; c < KF; c += blockDim.x) {
        float vc, vu;
        if (c < JF) {
            vc = vu = (f < n_pre_seq) ? origin_x[((size_t)b * JF + c) * kT + f] : 0.f;
        } else if (c == JF) {
            vc = vu = (f < n_pre_seq) ? 1.f : 0.f;
        } else {
            vc = conv4[((size_t)b * kAudioFeat + (c - JF - 1)) * kT + f];
            vu = 0.f;                                                   // mask_cond(force_mask), RAG.py:82-83
        }
        fc[c] = vc;
        fu[c] = vu;
    }
}

hipError_t launch_build_feats(const float* origin_x, const float* conv4, float* feat_c, float* feat_u,
                              int B, int JF, int n_pre_seq, hipStream_t st) {
    hipLaunchKernelGGL(k_build_feats, dim3(B * kT), dim3(256), 0, st, origin_x, conv4, feat_c, feat_u, JF, n_pre_seq);
    return hipGetLastError();
}

// [B][JF][T] (reference [B,J,F,T]) <-> internal [B][T][JF]
__global__ void k_to_internal(const float* __restrict__ src, float* __restrict__ dst, int JF) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < kT * JF; i += blockDim.x) {
        const int f = i / JF, c = i - f * JF;
        dst[(size_t)b * kT * JF + i] = src[((size_t)b * JF + c) * kT + f];
    }
}
__global__ void k_from_internal(const float* __restrict__ src, float* __restrict__ dst, int JF) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < kT * JF; i += blockDim.x) {
        const int c = i / kT, f = i - c * kT;
        dst[(size_t)b * kT * JF + i] = src[((size_t)b * kT + f) * JF + c];
    }
}
hipError_t launch_to_internal(const float* s, float* d, int B, int JF, hipStream_t st) {
    hipLaunchKernelGGL(k_to_internal, dim3(B), dim3(256), 0, st, s, d, JF);
    return hipGetLastError();
}
hipError_t launch_from_internal(const float* s, float* d, int B, int JF, hipStream_t st) {
    hipLaunchKernelGGL(k_from_internal, dim3(B), dim3(256), 0, st, s, d, JF);
    return hipGetLastError();
}

// conv4 [B][256][T] -> audio feature [B][T][256] (audio_enc.py:25 transpose), for ls_read("audio_feat")
__global__ void k_transpose_feat(const float* __restrict__ conv4, float* __restrict__ out) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < kT * kAudioFeat; i += blockDim.x) {
        const int f = i / kAudioFeat, c = i - f * kAudioFeat;
        out[(size_t)b * kT * kAudioFeat + i] = conv4[((size_t)b * kAudioFeat + c) * kT + f];
    }
}
hipError_t launch_transpose_feat(const float* conv4, float* out, int B, hipStream_t st) {
    hipLaunchKernelGGL(k_transpose_feat, dim3(B), dim3(256), 0, st, conv4, out);
    return hipGetLastError();
}

// q_sample (gaussian_diffusion.py:240-258): out = a*x0 + b*noise; layout-agnostic elementwise
__global__ void k_q_sample(const float* x0, const float* noise, float* out,   // out may alias x0 or noise

                           size_t n, float a, float b) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a * x0[i] + b * noise[i];
}
hipError_t launch_q_sample(const float* x0, const float* noise, float* out, size_t n, float a, float b, hipStream_t st) {
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_q_sample, dim3(blocks), dim3(256), 0, st, x0, noise, out, n, a, b);
    return hipGetLastError();
}

// Philox x_T (perf mode): element index follows the reference layout (c*T+f) so it is layout independent.
__global__ void k_randn_fill(float* __restrict__ out, int JF, const CallParams* __restrict__ call, unsigned stream_id) {
    const int b = blockIdx.x;
    const unsigned long long gidx = call->sample_offset + (unsigned long long)b;
    for (int i = threadIdx.x; i < kT * JF; i += blockDim.x) {
        const int f = i / JF, c = i - f * JF;
        out[(size_t)b * kT * JF + i] = philox_normal(call, gidx, 0xFFFFFFu, stream_id, (unsigned)(c * kT + f));
    }
}
hipError_t launch_randn_fill(float* out, int B, int JF, const CallParams* call, unsigned stream_id, hipStream_t st) {
    hipLaunchKernelGGL(k_randn_fill, dim3(B), dim3(256), 0, st, out, JF, call, stream_id);
    return hipGetLastError();
}

}  // namespace ls
