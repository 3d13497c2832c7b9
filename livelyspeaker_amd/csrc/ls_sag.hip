// SAG decoder: Decoder_TRANSFORMER.forward, scripts/model/motionclip_module.py:98-183 --
// timequeries = mapping([prefix poses | bit]) + pe; 3 x nn.TransformerDecoderLayer (post-norm, 4 heads x 128,
// self-attention over the 34 frame queries, cross-attention to a memory of length 1 = the CLIP text feature,
// FFN 512-1024-512 with exact GELU); finallayer -> [B, J, F, T].  Runs once per batch (it produces init_image for
// the RAG refine loop), so the small kernels here are plain VALU; the linears use the fp32 MFMA GEMM (ls_gemm.hip).
#include "ls_internal.h"
#include "ls_lanes.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));

// rows r = b*T + f;  q[r][d] = b_map[d] + [f < n_pre] (W_map[d][:JF] . x[b,:,f] + W_map[d][JF]) + pe[f][d]
// (motionclip_module.py:155-168: pre_cond is zero beyond the prefix poses, so mapping() leaves just its bias there)
__global__ void k_sag_queries(const float* __restrict__ x, const float* __restrict__ wmap, const float* __restrict__ bmap,
                              const float* __restrict__ pe, float* __restrict__ q, int JF, int n_pre, int D) {
    extern __shared__ float sx[];
    const int r = blockIdx.x, b = r / kT, f = r % kT;
    if (f < n_pre)
        for (int c = threadIdx.x; c < JF; c += blockDim.x) sx[c] = x[((size_t)b * JF + c) * kT + f];
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float v = bmap[d];
        if (f < n_pre) {
            const float* w = wmap + (size_t)d * (JF + 1);
            for (int c = 0; c < JF; ++c) v = fmaf(w[c], sx[c], v);
            v += w[JF];
        }
        q[(size_t)r * D + d] = v + pe[(size_t)f * D + d];
    }
}

// nn.MultiheadAttention self-attention of one (sample, head): softmax(q k^T / sqrt(hd)) v over the T=34 queries.
// qkv rows are [q | k | v] of width 3*D (in_proj of the packed weight).
template <int HD>
__global__ __launch_bounds__(256) void k_sag_attention(const float* __restrict__ qkv, float* __restrict__ out, int D) {
    __shared__ float sq[kT][HD + 1], sk[kT][HD + 1], sv[kT][HD + 1], sp[kT][kT + 1];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < kT * HD; i += 256) {
        const int t = i / HD, d = i % HD;
        const float* row = qkv + (size_t)(b * kT + t) * 3 * D + h * HD + d;
        sq[t][d] = row[0];
        sk[t][d] = row[D];
        sv[t][d] = row[2 * D];
    }
    __syncthreads();
    const float scale = rsqrtf((float)HD);
    for (int i = tid; i < kT * kT; i += 256) {
        const int a = i / kT, c = i % kT;
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < HD; ++d) s = fmaf(sq[a][d] * scale, sk[c][d], s);      // torch scales q before q.k^T
        sp[a][c] = s;
    }
    __syncthreads();
    if (tid < kT) {
        float m = -INFINITY;
        for (int c = 0; c < kT; ++c) m = fmaxf(m, sp[tid][c]);
        float sum = 0.f;
        for (int c = 0; c < kT; ++c) { const float e = expf(sp[tid][c] - m); sp[tid][c] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int c = 0; c < kT; ++c) sp[tid][c] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < kT * HD; i += 256) {
        const int t = i / HD, d = i % HD;
        float o = 0.f;
#pragma unroll 2
        for (int c = 0; c < kT; ++c) o = fmaf(sp[t][c], sv[c][d], o);
        out[(size_t)(b * kT + t) * D + h * HD + d] = o;
    }
}

// y = LayerNorm(x (+ bc[b])) * w + beta, eps 1e-5, biased variance (nn.LayerNorm); one wave per row of D = 512
__global__ __launch_bounds__(256) void k_layernorm512(const float* __restrict__ x, const float* __restrict__ bc,
                                                      const float* __restrict__ w, const float* __restrict__ beta,
                                                      float* __restrict__ y, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const f4* xr = reinterpret_cast<const f4*>(x + (size_t)r * kD);
    f4 v0 = xr[lane], v1 = xr[lane + 64];
    if (bc) {
        const f4* br = reinterpret_cast<const f4*>(bc + (size_t)(r / kT) * kD);
        v0 += br[lane];
        v1 += br[lane + 64];
    }
    float s = (v0[0] + v0[1]) + (v0[2] + v0[3]) + (v1[0] + v1[1]) + (v1[2] + v1[3]);
    s = wave_sum(s);
    const float mean = s * (1.0f / kD);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float a = v0[e] - mean, c = v1[e] - mean; q += a * a + c * c; }
    q = wave_sum(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / kD) + 1e-5f);
    const f4* wr = reinterpret_cast<const f4*>(w);
    const f4* be = reinterpret_cast<const f4*>(beta);
    f4* yr = reinterpret_cast<f4*>(y + (size_t)r * kD);
    yr[lane] = (v0 - mean) * rstd * wr[lane] + be[lane];
    yr[lane + 64] = (v1 - mean) * rstd * wr[lane + 64] + be[lane + 64];
}

// finallayer + "zero for padded area" + permute to [B, J*F, T] (motionclip_module.py:172-176)
__global__ void k_sag_final(const float* __restrict__ xh, const float* __restrict__ wf, const float* __restrict__ bf,
                            const unsigned char* __restrict__ mask, float* __restrict__ out, int JF, int D) {
    extern __shared__ float sx[];
    const int r = blockIdx.x, b = r / kT, f = r % kT;
    for (int d = threadIdx.x; d < D; d += blockDim.x) sx[d] = xh[(size_t)r * D + d];
    __syncthreads();
    const bool keep = mask ? mask[r] != 0 : true;
    for (int c = threadIdx.x; c < JF; c += blockDim.x) {
        float v = bf[c];
        const float* w = wf + (size_t)c * D;
        for (int d = 0; d < D; ++d) v = fmaf(w[d], sx[d], v);
        out[((size_t)b * JF + c) * kT + f] = keep ? v : 0.f;
    }
}

hipError_t launch_sag_queries(const float* x, const float* wmap, const float* bmap, const float* pe, float* q, int B,
                              int JF, int n_pre, int D, hipStream_t st) {
    hipLaunchKernelGGL(k_sag_queries, dim3(B * kT), dim3(256), JF * sizeof(float), st, x, wmap, bmap, pe, q, JF, n_pre, D);
    return hipGetLastError();
}
hipError_t launch_sag_attention(const float* qkv, float* out, int B, int heads, int D, hipStream_t st) {
    if (D / heads != 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_sag_attention<128>), dim3(B, heads), dim3(256), 0, st, qkv, out, D);
    return hipGetLastError();
}
hipError_t launch_layernorm512(const float* x, const float* bc, const float* w, const float* beta, float* y, int rows,
                               hipStream_t st) {
    hipLaunchKernelGGL(k_layernorm512, dim3((rows + 3) / 4), dim3(256), 0, st, x, bc, w, beta, y, rows);
    return hipGetLastError();
}
hipError_t launch_sag_final(const float* xh, const float* wf, const float* bf, const unsigned char* mask, float* out, int B,
                            int JF, int D, hipStream_t st) {
    hipLaunchKernelGGL(k_sag_final, dim3(B * kT), dim3(64), D * sizeof(float), st, xh, wf, bf, mask, out, JF, D);
    return hipGetLastError();
}

}  // namespace ls
