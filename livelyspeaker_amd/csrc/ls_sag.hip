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
// qc (optional): the DISTINCT query rows only -- [B * n_pre] prefix rows (b, f < n_pre), then the T - n_pre rows f >= n_pre, which are
// the same for every sample (b_map + pe[f]): the first layer's packed in_proj runs over these rows instead of all B * T.
__global__ void k_sag_queries(const float* __restrict__ x, const float* __restrict__ wmap, const float* __restrict__ bmap,
                              const float* __restrict__ pe, float* __restrict__ q, float* __restrict__ qc, int JF, int n_pre, int D) {
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
        v += pe[(size_t)f * D + d];
        q[(size_t)r * D + d] = v;
        if (qc) {
            if (f < n_pre) qc[((size_t)b * n_pre + f) * D + d] = v;
            else if (b == 0) qc[((size_t)gridDim.x / kT * n_pre + (f - n_pre)) * D + d] = v;
        }
    }
}

// nn.MultiheadAttention self-attention of one (sample, head): softmax(q k^T / sqrt(hd)) v over the T = 34 queries -- the only
// QK^T / softmax / attn.V in the model (motionclip_module.py:98-183).  qkv rows are [q | k | v] of width 3*D (packed in_proj).
// Workgroup = one sample, 4 waves, looping over its heads: the NEXT head's Q / K / V are fetched into registers while the current
// head is computed (as one workgroup per (sample, head) the kernel was a chain of latencies -- global load, LDS, three barriers --
// repeated over four rounds of 512 resident workgroups: 46 us at B = 512 for 2.4 MFLOP and 52 KB per pair).  Q (pre-scaled), K, V
// [34][HD] are staged in LDS with row stride HD + 4 (conflict-free
// ds_read_b128); both contractions run on v_mfma_f32_16x16x4_f32 with T padded to 3 tiles of 16 (rows past 33 are clamped on the
// read and never stored / masked):
//   scores = Q K^T : 3 x 3 tiles, K-dim = HD in the k-permuted float4 order (lane (row, g) holds d = 16q + 4g + e for step e)
//   softmax        : one wave per query row, lane = key; row max and row sum are wavefront reductions (DPP / readlane)
//   out = P V      : 3 x (HD/16) tiles, K-dim = 36 keys (P's columns 34, 35 are written as zero)
template <int HD>
// n_pre_c > 0: qkv holds the distinct rows only (k_sag_queries' qc order): sample b's frame t lives at row b * n_pre_c + t for
// t < n_pre_c and at the shared row B * n_pre_c + (t - n_pre_c) otherwise.
__global__ __launch_bounds__(256) void k_sag_attention(const float* __restrict__ qkv, float* __restrict__ out, int D, int heads, int n_pre_c) {
    constexpr int LQ = HD + 4, LP = 37, KP = 36;
    __shared__ __attribute__((aligned(16))) float sq[kT * LQ], sk[kT * LQ], sv[kT * LQ];
    __shared__ float sp[kT * LP];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const float scale = rsqrtf((float)HD);
    constexpr int kF4 = kT * HD / 4, kPer = (kF4 + 255) / 256;
    f4 vq[kPer], vk[kPer], vv[kPer];
    // this thread's share of one head's Q / K / V: float4 loads with a clamped index (branch-free, all in flight together)
    auto fetch = [&](int h) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int idx = min(tid + 256 * j, kF4 - 1), t = idx / (HD / 4), d4 = idx % (HD / 4);
            const int rr = n_pre_c <= 0 ? b * kT + t : (t < n_pre_c ? b * n_pre_c + t : (int)gridDim.x * n_pre_c + (t - n_pre_c));
            const float* row = qkv + (size_t)rr * 3 * D + h * HD + 4 * d4;
            vq[j] = *reinterpret_cast<const f4*>(row);
            vk[j] = *reinterpret_cast<const f4*>(row + D);
            vv[j] = *reinterpret_cast<const f4*>(row + 2 * D);
        }
    };
    fetch(0);
    for (int h = 0; h < heads; ++h) {
        if (h) __syncthreads();                                   // the previous head's P.V is done reading sv / sp
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int idx = tid + 256 * j, t = idx / (HD / 4), d = 4 * (idx % (HD / 4));
            if (idx < kF4) {
                *reinterpret_cast<f4*>(&sq[t * LQ + d]) = vq[j] * scale;      // torch scales q before q.k^T
                *reinterpret_cast<f4*>(&sk[t * LQ + d]) = vk[j];
                *reinterpret_cast<f4*>(&sv[t * LQ + d]) = vv[j];
            }
        }
        __syncthreads();
        if (h + 1 < heads) fetch(h + 1);                          // in flight during this head's scores / softmax / P.V
        // ---- scores: tile (mt, nt) = w, w + 4, w + 8 of the 3 x 3 grid; lane holds S[a = 16 mt + 4 g + r][c = 16 nt + s16]
        for (int tile = w; tile < 9; tile += 4) {
            const int mt = tile / 3, nt = tile % 3;
            const float* qa = sq + min(16 * mt + s16, kT - 1) * LQ + 4 * g;
            const float* kb = sk + min(16 * nt + s16, kT - 1) * LQ + 4 * g;
            f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < HD / 16; ++q) {
                const f4 av = *reinterpret_cast<const f4*>(qa + 16 * q), bv = *reinterpret_cast<const f4*>(kb + 16 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
            }
            const int c = 16 * nt + s16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * mt + 4 * g + r;
                if (a < kT && c < kT) sp[a * LP + c] = acc[r];
            }
        }
        __syncthreads();
        // ---- softmax over the keys of each query row: one wave per row, lane = key
        for (int a = w; a < kT; a += 4) {
            const float v = lane < kT ? sp[a * LP + lane] : -INFINITY;
            const float m = wave_max(v);
            const float e = lane < kT ? expf(v - m) : 0.f;
            const float inv = 1.0f / wave_sum(e);
            if (lane < KP) sp[a * LP + lane] = e * inv;                   // columns 34, 35: zero (K padding of the P.V product)
        }
        __syncthreads();
        // ---- out = P V: wave w owns feature tiles w, w + 4 of HD / 16; lane holds O[a = 16 mt + 4 g + r][d = 16 nt + s16]
        for (int nt = w; nt < HD / 16; nt += 4) {
            f4 acc[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) acc[mt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KP / 4; ++ks) {
                const float bv = sv[min(4 * ks + g, kT - 1) * LQ + 16 * nt + s16];    // rows 34, 35 meet P's zero columns
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    const float av = sp[min(16 * mt + s16, kT - 1) * LP + 4 * ks + g];
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[mt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = 16 * mt + 4 * g + r;
                    if (a < kT) out[(size_t)(b * kT + a) * D + h * HD + 16 * nt + s16] = acc[mt][r];
                }
        }
    }
}

// y = LayerNorm(x (+ bc[b])) * w + beta, eps 1e-5, biased variance (nn.LayerNorm); one wave per row of D = 512
__global__ __launch_bounds__(256) void k_layernorm512(const float* __restrict__ x, const float* __restrict__ bc, int bc_stride,
                                                      const float* __restrict__ w, const float* __restrict__ beta,
                                                      float* __restrict__ y, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const f4* xr = reinterpret_cast<const f4*>(x + (size_t)r * kD);
    f4 v0 = xr[lane], v1 = xr[lane + 64];
    if (bc) {
        const f4* br = reinterpret_cast<const f4*>(bc + (size_t)(r / kT) * bc_stride);
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

// y = LN2( LN1(x) + bc[b] ): norm1 and norm2 of a decoder layer in one pass over the rows -- between them the layer only adds its
// cross-attention vector (one memory token: a per-sample constant, ls_sag_api.cpp), so the intermediate never has to leave registers.
// Same arithmetic, in the same order, as two k_layernorm512 launches.
__global__ __launch_bounds__(256) void k_layernorm512x2(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ bc, int bc_stride, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ y, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const f4* xr = reinterpret_cast<const f4*>(x + (size_t)r * kD);
    const f4* br = reinterpret_cast<const f4*>(bc + (size_t)(r / kT) * bc_stride);
    f4 v0 = xr[lane], v1 = xr[lane + 64];
    const f4 c0 = br[lane], c1 = br[lane + 64];
    const f4 *pw1 = reinterpret_cast<const f4*>(w1), *pb1 = reinterpret_cast<const f4*>(b1), *pw2 = reinterpret_cast<const f4*>(w2), *pb2 = reinterpret_cast<const f4*>(b2);
    const f4 g10 = pw1[lane], g11 = pw1[lane + 64], e10 = pb1[lane], e11 = pb1[lane + 64];
    const f4 g20 = pw2[lane], g21 = pw2[lane + 64], e20 = pb2[lane], e21 = pb2[lane + 64];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float s = (v0[0] + v0[1]) + (v0[2] + v0[3]) + (v1[0] + v1[1]) + (v1[2] + v1[3]);
        const float mean = wave_sum(s) * (1.0f / kD);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = v0[e] - mean, c = v1[e] - mean; q += a * a + c * c; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / kD) + 1e-5f);
        if (pass == 0) {
            v0 = ((v0 - mean) * rstd * g10 + e10) + c0;
            v1 = ((v1 - mean) * rstd * g11 + e11) + c1;
        } else {
            v0 = (v0 - mean) * rstd * g20 + e20;
            v1 = (v1 - mean) * rstd * g21 + e21;
        }
    }
    f4* yr = reinterpret_cast<f4*>(y + (size_t)r * kD);
    yr[lane] = v0;
    yr[lane + 64] = v1;
}

// finallayer + "zero for padded area" + permute to [B, J*F, T] (motionclip_module.py:172-176) on v_mfma_f32_16x16x4_f32.
// Workgroup = one sample: out[c][f] = sum_d W[c][d] x[f][d] as D[feature tile][frame tile] (NCT x 3 tiles of 16), both operands
// read straight from global memory in the k-permuted float4 order (lane (row, g) holds d = 16q + 4g .. +3: one float4 = 4 MFMA
// steps).  Each of the 4 waves contracts over its own quarter of D (its loads are issued together), the partial tiles are summed
// through LDS, and the [c][f] result leaves in the reference layout with coalesced stores along f.
// (History at B = 512: one thread per output feature 154 us; one wave per row with wave-wide sums 70 us; this form: see profiles/.)
__global__ __launch_bounds__(256) void k_sag_final(const float* __restrict__ xh, const float* __restrict__ wf, const float* __restrict__ bf,
                                                   const unsigned char* __restrict__ mask, float* __restrict__ out, int JF, int D) {
    constexpr int NCT = 2, NFT = 3;                               // 32 features x 48 (34) frames per pass
    __shared__ float part[4][NCT * NFT][4][64];                   // [wave][tile][reg][lane]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int kq = D / 64;                                        // 16-wide k groups per wave (D = 512: 8)
    const float* xb = xh + (size_t)b * kT * D + (size_t)w * (D / 4) + 4 * g;
    for (int c0 = 0; c0 < JF; c0 += 16 * NCT) {                   // TED: one pass (27 features); BEAT: nine (282)
        f4 acc[NCT][NFT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) acc[ct][ft] = (f4){0.f, 0.f, 0.f, 0.f};
        const float* wb = wf + (size_t)w * (D / 4) + 4 * g;
        for (int q = 0; q < kq; ++q) {
            f4 A[NCT], Bv[NFT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) A[ct] = *reinterpret_cast<const f4*>(wb + (size_t)min(c0 + 16 * ct + s16, JF - 1) * D + 16 * q);
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) Bv[ft] = *reinterpret_cast<const f4*>(xb + (size_t)min(16 * ft + s16, kT - 1) * D + 16 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int ft = 0; ft < NFT; ++ft) acc[ct][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ct][e], Bv[ft][e], acc[ct][ft], 0, 0, 0);
        }
        __syncthreads();                                          // the previous pass's readers are done with `part`
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[w][ct * NFT + ft][r][lane] = acc[ct][ft][r];
        __syncthreads();
        // tile (ct, ft), register r of lane (s16, g): feature c = c0 + 16 ct + 4 g + r, frame f = 16 ft + s16
        for (int idx = tid; idx < NCT * NFT * 256; idx += 256) {
            const int tile = idx >> 8, r = (idx >> 6) & 3, ln = idx & 63;
            const int ct = tile / NFT, ft = tile - ct * NFT;
            const int c = c0 + 16 * ct + 4 * (ln >> 4) + r, f = 16 * ft + (ln & 15);
            if (c < JF && f < kT) {
                const float v = ((part[0][tile][r][ln] + part[1][tile][r][ln]) + (part[2][tile][r][ln] + part[3][tile][r][ln])) + bf[c];
                const bool keep = mask ? mask[b * kT + f] != 0 : true;
                out[((size_t)b * JF + c) * kT + f] = keep ? v : 0.f;
            }
        }
    }
}

hipError_t launch_sag_queries(const float* x, const float* wmap, const float* bmap, const float* pe, float* q, float* qc, int B,
                              int JF, int n_pre, int D, hipStream_t st) {
    hipLaunchKernelGGL(k_sag_queries, dim3(B * kT), dim3(256), JF * sizeof(float), st, x, wmap, bmap, pe, q, qc, JF, n_pre, D);
    return hipGetLastError();
}
hipError_t launch_sag_attention(const float* qkv, float* out, int B, int heads, int D, int n_pre_c, hipStream_t st) {
    if (D / heads != 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_sag_attention<128>), dim3(B), dim3(256), 0, st, qkv, out, D, heads, n_pre_c);
    return hipGetLastError();
}
hipError_t launch_layernorm512(const float* x, const float* bc, int bc_stride, const float* w, const float* beta, float* y, int rows,
                               hipStream_t st) {
    hipLaunchKernelGGL(k_layernorm512, dim3((rows + 3) / 4), dim3(256), 0, st, x, bc, bc_stride, w, beta, y, rows);
    return hipGetLastError();
}
hipError_t launch_layernorm512x2(const float* x, const float* w1, const float* b1, const float* bc, int bc_stride, const float* w2,
                                 const float* b2, float* y, int rows, hipStream_t st) {
    hipLaunchKernelGGL(k_layernorm512x2, dim3((rows + 3) / 4), dim3(256), 0, st, x, w1, b1, bc, bc_stride, w2, b2, y, rows);
    return hipGetLastError();
}
hipError_t launch_sag_final(const float* xh, const float* wf, const float* bf, const unsigned char* mask, float* out, int B,
                            int JF, int D, hipStream_t st) {
    if (D % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sag_final, dim3(B), dim3(256), 0, st, xh, wf, bf, mask, out, JF, D);
    return hipGetLastError();
}

}  // namespace ls
