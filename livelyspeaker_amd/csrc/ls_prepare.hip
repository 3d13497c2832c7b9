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

// InputProcess features without the x_t columns (RAG.py:110-112, 184-192), split by what they multiply:
//   feat_p[(b,f)] = [origin_x[b,:,f] if f < n_pre_seq else 0 | indicator bit | 0 pad]   (KPP columns: shared by both CFG passes)
//   feat_a[(b,f)] = audio feature conv4[b,:,f]                                          (256 columns: the cond pass only)
// so that static_u = feat_p . Wpre^T + b  and  static_c = static_u + feat_a . Waud^T  (mask_cond zeroes the audio term of the uncond
// pass, RAG.py:82-83): K = KPP + 256 in total instead of 2 x (KPP + 256).
// One workgroup per (sample, run of <= 48 frames): conv4's [256][frames] block is staged through LDS so that both its reads (whole row pieces)
// and the [frame][256] writes are coalesced (round 3; one workgroup per (sample, frame) read it with a stride of T floats between lanes: 25 us
// at B = 512).
constexpr int kBfTC = 48;
__global__ __launch_bounds__(256) void k_build_feats(const float* __restrict__ origin_x, const float* __restrict__ conv4,
                                                     float* __restrict__ feat_p, float* __restrict__ feat_a, int JF, int KPP, int n_pre_seq, int T) {
    __shared__ float tile[kAudioFeat * (kBfTC + 1)];
    const int b = blockIdx.x, f0 = blockIdx.y * kBfTC, nf = min(kBfTC, T - f0), tid = threadIdx.x;
    const float* src = conv4 + (size_t)b * kAudioFeat * T + f0;
    {   // thread = channel `tid`: its row piece, all loads first (a rolled load -> LDS loop waits for each load in turn)
        float v[kBfTC];
#pragma unroll
        for (int f = 0; f < kBfTC; ++f) v[f] = src[(size_t)tid * T + min(f, nf - 1)];
#pragma unroll
        for (int f = 0; f < kBfTC; ++f) tile[tid * (kBfTC + 1) + f] = v[f];
    }
    float* fp = feat_p + ((size_t)b * T + f0) * KPP;
    for (int i = tid; i < nf * KPP; i += 256) {
        const int f = f0 + i / KPP, c = i % KPP;
        float v = 0.f;
        if (f < n_pre_seq) v = c < JF ? origin_x[((size_t)b * JF + c) * T + f] : (c == JF ? 1.f : 0.f);
        fp[i] = v;
    }
    __syncthreads();
    float* fa = feat_a + ((size_t)b * T + f0) * kAudioFeat;
    for (int i = tid; i < nf * kAudioFeat; i += 256) fa[i] = tile[(i & (kAudioFeat - 1)) * (kBfTC + 1) + i / kAudioFeat];
}

hipError_t launch_build_feats(const float* origin_x, const float* conv4, float* feat_p, float* feat_a,
                              int B, int JF, int KPP, int n_pre_seq, hipStream_t st, int T) {
    hipLaunchKernelGGL(k_build_feats, dim3(B, (T + kBfTC - 1) / kBfTC), dim3(256), 0, st, origin_x, conv4, feat_p, feat_a, JF, KPP, n_pre_seq, T);
    return hipGetLastError();
}

// speaker style (RAG.py:116-119): [mu | logvar] rows of the fused projection -> z_mu, z_logvar, z_std = exp(0.5 logvar)
__global__ void k_split_style(const float* __restrict__ ml, float* __restrict__ mu, float* __restrict__ lv, float* __restrict__ sd, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = i / kD, c = i - b * kD;
    const float m = ml[(size_t)b * 2 * kD + c], l = ml[(size_t)b * 2 * kD + kD + c];
    mu[i] = m; lv[i] = l; sd[i] = expf(0.5f * l);
}
hipError_t launch_split_style(const float* ml, float* mu, float* lv, float* sd, int B, hipStream_t st) {
    const int n = B * kD;
    hipLaunchKernelGGL(k_split_style, dim3((n + 255) / 256), dim3(256), 0, st, ml, mu, lv, sd, n);
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

// One workgroup per sample: the same arithmetic as the fused step kernel's epilogue (ls_step_kernel.h), coefficients per sample.
// p_sample / ddim_sample (gaussian_diffusion.py:507-558, 745-798) with a [B] timestep tensor whose entries differ.
__global__ void k_sampler_update(const float* __restrict__ x_t, const float* __restrict__ x0v, const float* __restrict__ noise,
                                 const float* __restrict__ table, const int64_t* __restrict__ indices, int n_steps,
                                 float* __restrict__ out, int JF, int T, int sampler) {
    const int b = blockIdx.x;
    long long i = indices[b];
    i = i < 0 ? 0 : (i >= n_steps ? n_steps - 1 : i);
    const float* coef = table + (size_t)i * 8;
    const float nzf = coef[0], c0 = coef[1], c1 = coef[2], c2 = coef[3], c3 = coef[4], c4 = coef[5];
    const bool t_nonzero = nzf != 0.f;
    const size_t base = (size_t)b * T * JF;
    for (int idx = threadIdx.x; idx < T * JF; idx += blockDim.x) {
        const int f = idx / JF, c = idx - f * JF;
        const float xt = x_t[base + idx], x0 = x0v[base + idx];
        const float nz = t_nonzero ? noise[((size_t)b * JF + c) * T + f] : 0.f;
        float xn;
        if (sampler == kDDPM) {
            xn = c0 * x0 + c1 * xt;
            if (t_nonzero) xn += c2 * nz;
        } else {
            const float eps = (c0 * xt - x0) / c1;
            xn = x0 * c2 + c3 * eps;
            if (t_nonzero) xn += c4 * nz;
        }
        out[base + idx] = xn;
    }
}
hipError_t launch_sampler_update(const float* x_t, const float* x0, const float* noise, const float* table, const int64_t* indices,
                                 int n_steps, float* out, int B, int JF, int T, int sampler, hipStream_t st) {
    hipLaunchKernelGGL(k_sampler_update, dim3(B), dim3(256), 0, st, x_t, x0, noise, table, indices, n_steps, out, JF, T, sampler);
    return hipGetLastError();
}

// gaussian_diffusion.py:314-320 (mix) + :365-371 (clamp) + :260-282, 507-558, 745-798 (update); one workgroup per sample
__global__ void k_inpaint_update(const InpaintArgs a) {
    const int b = blockIdx.x;
    const size_t base = (size_t)b * a.T * a.JF;
    const unsigned long long gidx = (a.call ? a.call->sample_offset : 0ull) + (unsigned long long)b;
    for (int idx = threadIdx.x; idx < a.T * a.JF; idx += blockDim.x) {
        const int f = idx / a.JF, c = idx - f * a.JF;
        const size_t ri = ((size_t)b * a.JF + c) * a.T + f;                 // the same element in the reference layout
        float x0 = a.x0[base + idx];
        if (a.maskf[base + idx] != 0.f) {
            float given = a.motion[base + idx];
            if (a.renoise) {
                const float n = a.inoise ? a.inoise[ri] : philox_normal(a.call, gidx, a.step_id, 4u, (unsigned)(c * a.T + f));
                given = a.qa * given + a.qb * n;                               // q_sample(inpainted_motion, t - 1)
            }
            x0 = given;
        }
        if (a.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        a.x0[base + idx] = x0;
        if (a.dump) a.dump[base + idx] = x0;
        if (a.sampler == kNone) continue;
        const float xt = a.x_t[base + idx];
        float nz = 0.f;
        if (a.t_nonzero) nz = a.noise ? a.noise[a.const_noise ? ((size_t)c * a.T + f) : ri] : philox_normal(a.call, gidx, a.step_id, 3u, (unsigned)(c * a.T + f));
        float xn;
        if (a.sampler == kDDPM) {
            xn = a.c0 * x0 + a.c1 * xt;
            if (a.t_nonzero) xn += a.c2 * nz;
        } else {
            const float eps = (a.c0 * xt - x0) / a.c1;
            xn = x0 * a.c2 + a.c3 * eps;
            if (a.t_nonzero) xn += a.c4 * nz;
        }
        a.out[base + idx] = xn;
    }
}
hipError_t launch_inpaint_update(const InpaintArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(k_inpaint_update, dim3(B), dim3(256), 0, st, a);
    return hipGetLastError();
}
__global__ void k_bytes_to_float(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] ? 1.f : 0.f;
}
hipError_t launch_bytes_to_float(const unsigned char* src, float* dst, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_bytes_to_float, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st, src, dst, n);
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
