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

// (the WavEncoder convolutions and their fused InstanceNorm statistics live in ls_conv.hip)

__global__ void k_gather_rows(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out,
                              int rows, int width, int table_rows, int idx_stride) {
    const int r = blockIdx.x;
    long long i = idx[(size_t)r * idx_stride];
    if (i < 0) i = 0;
    if (i >= table_rows) i = table_rows - 1;
    for (int c = threadIdx.x; c < width; c += blockDim.x) out[(size_t)r * width + c] = table[(size_t)i * width + c];
}

hipError_t launch_gather_rows(const float* table, const int64_t* idx, float* out, int rows, int width,
                              int table_rows, hipStream_t st, int idx_stride) {
    hipLaunchKernelGGL(k_gather_rows, dim3(rows), dim3(256), 0, st, table, idx, out, rows, width, table_rows, idx_stride);
    return hipGetLastError();
}

// InputProcess features without the x_t columns (RAG.py:110-112, 184-192):
// row (b,f) = [origin_x[b,:,f] if f < n_pre_seq else 0 | indicator bit | audio feature (cond) or 0 (uncond)]
__global__ void k_build_feats(const float* __restrict__ origin_x, const float* __restrict__ conv4,
                              float* __restrict__ feat_c, float* __restrict__ feat_u, int JF, int KFP, int n_pre_seq, int T) {
    const int b = blockIdx.x / T, f = blockIdx.x % T;
    const int KF = JF + 1 + kAudioFeat;
    float* fc = feat_c + (size_t)blockIdx.x * KFP;          // rows padded with zeros to KFP (a whole number of GEMM K tiles)
    float* fu = feat_u + (size_t)blockIdx.x * KFP;
    for (int c = threadIdx.x; c < KFP; c += blockDim.x) {
        float vc, vu;
        if (c < JF) {
            vc = vu = (f < n_pre_seq) ? origin_x[((size_t)b * JF + c) * T + f] : 0.f;
        } else if (c == JF) {
            vc = vu = (f < n_pre_seq) ? 1.f : 0.f;
        } else if (c < KF) {
            vc = conv4[((size_t)b * kAudioFeat + (c - JF - 1)) * T + f];
            vu = 0.f;                                                   // mask_cond(force_mask), RAG.py:82-83
        } else {
            vc = vu = 0.f;
        }
        fc[c] = vc;
        fu[c] = vu;
    }
}

hipError_t launch_build_feats(const float* origin_x, const float* conv4, float* feat_c, float* feat_u,
                              int B, int JF, int KFP, int n_pre_seq, hipStream_t st, int T) {
    hipLaunchKernelGGL(k_build_feats, dim3(B * T), dim3(256), 0, st, origin_x, conv4, feat_c, feat_u, JF, KFP, n_pre_seq, T);
    return hipGetLastError();
}

// [B][JF][T] (reference [B,J,F,T]) <-> internal [B][T][JF]
__global__ void k_to_internal(const float* __restrict__ src, float* __restrict__ dst, int JF, int T) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * JF; i += blockDim.x) {
        const int f = i / JF, c = i - f * JF;
        dst[(size_t)b * T * JF + i] = src[((size_t)b * JF + c) * T + f];
    }
}
__global__ void k_from_internal(const float* __restrict__ src, float* __restrict__ dst, int JF, int T) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * JF; i += blockDim.x) {
        const int c = i / T, f = i - c * T;
        dst[(size_t)b * T * JF + i] = src[((size_t)b * T + f) * JF + c];
    }
}
hipError_t launch_to_internal(const float* s, float* d, int B, int JF, hipStream_t st, int T) {
    hipLaunchKernelGGL(k_to_internal, dim3(B), dim3(256), 0, st, s, d, JF, T);
    return hipGetLastError();
}
hipError_t launch_from_internal(const float* s, float* d, int B, int JF, hipStream_t st, int T) {
    hipLaunchKernelGGL(k_from_internal, dim3(B), dim3(256), 0, st, s, d, JF, T);
    return hipGetLastError();
}

// conv4 [B][256][T] -> audio feature [B][T][256] (audio_enc.py:25 transpose), for ls_read("audio_feat")
__global__ void k_transpose_feat(const float* __restrict__ conv4, float* __restrict__ out, int T) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * kAudioFeat; i += blockDim.x) {
        const int f = i / kAudioFeat, c = i - f * kAudioFeat;
        out[(size_t)b * T * kAudioFeat + i] = conv4[((size_t)b * kAudioFeat + c) * T + f];
    }
}
hipError_t launch_transpose_feat(const float* conv4, float* out, int B, hipStream_t st, int T) {
    hipLaunchKernelGGL(k_transpose_feat, dim3(B), dim3(256), 0, st, conv4, out, T);
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
__global__ void k_randn_fill(float* __restrict__ out, int JF, const CallParams* __restrict__ call, unsigned stream_id, int T) {
    const int b = blockIdx.x;
    const unsigned long long gidx = call->sample_offset + (unsigned long long)b;
    for (int i = threadIdx.x; i < T * JF; i += blockDim.x) {
        const int f = i / JF, c = i - f * JF;
        out[(size_t)b * T * JF + i] = philox_normal(call, gidx, 0xFFFFFFu, stream_id, (unsigned)(c * T + f));
    }
}
hipError_t launch_randn_fill(float* out, int B, int JF, const CallParams* call, unsigned stream_id, hipStream_t st, int T) {
    hipLaunchKernelGGL(k_randn_fill, dim3(B), dim3(256), 0, st, out, JF, call, stream_id, T);
    return hipGetLastError();
}

}  // namespace ls
