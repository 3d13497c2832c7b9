// Batch-level non-GEMM kernels of the training step (the mixer itself runs in the fused kernels of
// ls_step.hip / ls_train_bwd.hip): q_sample + feature build, token-weight gradients, reparameterisation + KLD, Huber /
// velocity losses and their gradient, conv1's weight gradient, mixer weight images, deterministic partial-sum reductions
// (no float atomics), AdamW.  All fp32.
#include "ls_internal.h"
#include "ls_train.h"
#include "ls_lanes.h"

namespace ls {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kDm = 512;

__device__ __forceinline__ float sigmoidf_(float a) { return 1.0f / (1.0f + expf(-a)); }
__device__ __forceinline__ float silu_f_(float a) { return a * sigmoidf_(a); }
__device__ __forceinline__ float silu_grad(float a) {
    const float s = sigmoidf_(a);
    return s * (1.0f + a * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------------------------
// q_sample (gaussian_diffusion.py:240-258) fused into the InputProcess / input_mapping operand (RAG.py:106-114, 184-192):
// feat[(b,t)][:] = [x_t[b,:,t] | origin_x[b,:,t] (t < n_pre_seq) | bit | conv4[b,:,t] * (1 - drop[b])], zero padded to KFP
// One workgroup per sample (round 3): conv4's [256][T] block goes through an LDS transpose, so its reads are row pieces and the [T][256] writes
// whole rows (one workgroup per (sample, frame) read it with a stride of T floats between lanes: 38 us at B = 512).
constexpr int kBftT = 34;                                                  // frames of the trained models (ls_train_create checks)
__global__ __launch_bounds__(256) void k_build_feat_train(const float* __restrict__ x_start, const float* __restrict__ noise,
                                   const float* __restrict__ origin_x, const float* __restrict__ c4, const float* __restrict__ drop,
                                   const float* __restrict__ ca, const float* __restrict__ cb, float* __restrict__ feat,
                                   float* __restrict__ x_t, TrainDims d, int n_pre_seq) {
    __shared__ float tile[256 * (kBftT + 1)];
    const int b = blockIdx.x, tid = threadIdx.x, T = d.T;
    {   // thread = audio channel `tid`: its T values, all loads first
        float v[kBftT];
        const float* src = c4 + ((size_t)b * 256 + tid) * T;
#pragma unroll
        for (int t = 0; t < kBftT; ++t) v[t] = src[t];
#pragma unroll
        for (int t = 0; t < kBftT; ++t) tile[tid * (kBftT + 1) + t] = v[t];
    }
    float* fb = feat + (size_t)b * T * d.KFP;
    const float a0 = ca[b], b0 = cb[b], keep = 1.0f - drop[b];
    for (int i = tid; i < d.JF * T; i += 256) {                            // x_t = q_sample(x_start, t, noise) and the prefix poses: [JF][T] in, [T][KFP] out
        const int j = i / T, t = i - j * T;
        const size_t g = (size_t)b * d.JF * T + i;
        const float v = a0 * x_start[g] + b0 * noise[g];
        x_t[g] = v;
        fb[(size_t)t * d.KFP + j] = v;
        fb[(size_t)t * d.KFP + d.JF + j] = t < n_pre_seq ? origin_x[g] : 0.f;
    }
    const int npad = d.KFP - d.KF + 1;                                     // the bit column and the zero padding
    for (int i = tid; i < T * npad; i += 256) {
        const int t = i / npad, c = i - t * npad;
        if (c == 0) fb[(size_t)t * d.KFP + 2 * d.JF] = t < n_pre_seq ? 1.f : 0.f;
        else fb[(size_t)t * d.KFP + d.KF + c - 1] = 0.f;
    }
    __syncthreads();
    for (int i = tid; i < T * 256; i += 256) {
        const int t = i >> 8, c = i & 255;
        fb[(size_t)t * d.KFP + 2 * d.JF + 1 + c] = tile[c * (kBftT + 1) + t] * keep;
    }
}

hipError_t launch_build_feat_train(const float* x_start, const float* noise, const float* origin_x, const float* c4, const float* drop,
                                   const float* ca, const float* cb, float* feat, float* x_t, TrainDims d, int n_pre_seq, hipStream_t st) {
    if (d.T != kBftT) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_build_feat_train, dim3(d.B), dim3(256), 0, st, x_start, noise, origin_x, c4, drop, ca, cb, feat, x_t, d, n_pre_seq);
    return hipGetLastError();
}

// Two GEMM-friendly copies of input_mapping.weight [512][KF], rebuilt from the master parameters every step:
//   wpad[n][k] = W[n][k] (k < KF), 0 up to KFP (whole 32-deep K tiles, 16-byte-aligned rows);  waT[j][n] = W[n][a0 + j] (the 256 audio columns,
//   reduction index contiguous: the operand of d audio features = dH . W[:, a0:])
__global__ void k_build_inmap_images(const float* __restrict__ w, float* __restrict__ wpad, float* __restrict__ waT, int KF, int KFP, int a0) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < kDm * KFP) {
        const int n = i / KFP, k = i - n * KFP;
        wpad[i] = k < KF ? w[(size_t)n * KF + k] : 0.f;
    }
    if (i < 256 * kDm) {
        const int j = i / kDm, n = i - j * kDm;
        waT[i] = w[(size_t)n * KF + a0 + j];
    }
}

hipError_t launch_build_inmap_images(const float* w, float* wpad, float* waT, int KF, int KFP, int a0, hipStream_t st) {
    const int total = kDm * KFP > 256 * kDm ? kDm * KFP : 256 * kDm;
    hipLaunchKernelGGL(k_build_inmap_images, dim3((total + 255) / 256), dim3(256), 0, st, w, wpad, waT, KF, KFP, a0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// da = g * silu'(apre); per-wave partial column sums of da in partial[wave][512] (-> bias gradient)
__global__ __launch_bounds__(256) void k_silu_bwd_colsum(const float* __restrict__ g, const float* __restrict__ apre,
                                                         float* __restrict__ da, float* __restrict__ partial, int rows, int nwaves) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    f4 s0 = (f4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
    for (int row = gw; row < rows; row += nwaves) {
        const f4* gr = reinterpret_cast<const f4*>(g + (size_t)row * kDm);
        const f4* ar = reinterpret_cast<const f4*>(apre + (size_t)row * kDm);
        f4 d0 = gr[lane], d1 = gr[64 + lane];
        const f4 a0 = ar[lane], a1 = ar[64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d0[e] *= silu_grad(a0[e]);
            d1[e] *= silu_grad(a1[e]);
        }
        f4* o = reinterpret_cast<f4*>(da + (size_t)row * kDm);
        o[lane] = d0; o[64 + lane] = d1;
        s0 += d0; s1 += d1;
    }
    f4* po = reinterpret_cast<f4*>(partial + (size_t)gw * kDm);
    po[lane] = s0; po[64 + lane] = s1;
}

hipError_t launch_silu_bwd_colsum(const float* g, const float* apre, float* da, float* partial, int rows, int nwaves, hipStream_t st) {
    hipLaunchKernelGGL(k_silu_bwd_colsum, dim3(nwaves / 4), dim3(256), 0, st, g, apre, da, partial, rows, nwaves);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Token mixing (Conv1d(S,S,1) over the token axis, mlp_module.py:51-55), backward.  (The forward and the data gradient run inside
// the fused mixer kernels.)  Token-weight gradient of every layer in one launch, on the matrix cores:
//   dWt[l][s'][s] = sum_{b,c} dA1[l][b][s'][c] U1[l][b][s][c],   d bt[l][s'] = sum_{b,c} dA1[l][b][s'][c],   U1 = alpha1 * x-hat1 + beta1
// Per sample this is a 35 x 35 (36 x 36) product over K = 512 channels: M = s' and N = s are padded to 3 tiles of 16, both operands
// are read straight from global memory as float4s in the k-permuted order the contraction allows (lane (row, g) holds channels
// 16q + 4g + e for MFMA step e), U1 is rebuilt from the saved x-hat on the way, and the first padding row of the N operand is set
// to ones so that column S of the product IS the bias gradient.  A wave accumulates kTokSpv samples in its 9 accumulators; the four
// waves of a workgroup are then summed through LDS in wave order (deterministic) into one partial per workgroup:
//   pw[l][wg][s'][s], pb[l][wg][s'].
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kTokSpv = 2, kTokMaxS = 47;            // samples per wave; S + 1 <= 48 (three tiles incl. the ones row)

__global__ __launch_bounds__(256) void k_tokmix_wgrad(const float* __restrict__ da, const float* __restrict__ xh1, const float* __restrict__ l1a,
                                                      const float* __restrict__ l1b, float* __restrict__ pw, float* __restrict__ pb, int S, int B) {
    __shared__ __attribute__((aligned(16))) float red[4][9][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int l = blockIdx.y;
    const size_t lbase = (size_t)l * B * S * kDm;
    f4v acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};
    int rowc[3];
    float vmask[3], ones[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int r = 16 * t + s16;
        rowc[t] = min(r, S - 1);                       // clamped row: the loads stay branch-free
        vmask[t] = r < S ? 1.f : 0.f;
        ones[t] = r == S ? 1.f : 0.f;                  // N operand: row S is all ones -> column S of the product = sum_c dA1
    }
    const float* al = l1a + l * kDm + 4 * g;
    const float* be = l1b + l * kDm + 4 * g;
    for (int i = 0; i < kTokSpv; ++i) {
        const int b = (blockIdx.x * 4 + w) * kTokSpv + i;
        if (b >= B) break;                             // wave-uniform
        const float* dab = da + lbase + (size_t)b * S * kDm + 4 * g;
        const float* xhb = xh1 + lbase + (size_t)b * S * kDm + 4 * g;
#pragma unroll 2
        for (int q = 0; q < kDm / 16; ++q) {
            const f4v a4 = *reinterpret_cast<const f4v*>(al + 16 * q), b4 = *reinterpret_cast<const f4v*>(be + 16 * q);
            f4v A[3], Bm[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                A[t] = *reinterpret_cast<const f4v*>(dab + (size_t)rowc[t] * kDm + 16 * q);
                Bm[t] = *reinterpret_cast<const f4v*>(xhb + (size_t)rowc[t] * kDm + 16 * q);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    A[t][e] *= vmask[t];
                    Bm[t][e] = fmaf(fmaf(Bm[t][e], a4[e], b4[e]), vmask[t], ones[t]);
                }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mt][e], Bm[nt][e], acc[mt][nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) *reinterpret_cast<f4v*>(&red[w][3 * mt + nt][lane][0]) = acc[mt][nt];
    __syncthreads();
    // lane (s16, g) of tile (mt, nt), register r: s' = 16 mt + 4 g + r, s = 16 nt + s16
    const size_t blk = (size_t)l * gridDim.x + blockIdx.x;
    for (int o = tid; o < 9 * 256; o += 256) {
        const int tile = o >> 8, ln = (o >> 2) & 63, r = o & 3;
        const float v = ((red[0][tile][ln][r] + red[1][tile][ln][r]) + red[2][tile][ln][r]) + red[3][tile][ln][r];
        const int sp = 16 * (tile / 3) + 4 * (ln >> 4) + r, sc = 16 * (tile % 3) + (ln & 15);
        if (sp < S) {
            if (sc < S) pw[blk * S * S + (size_t)sp * S + sc] = v;
            else if (sc == S) pb[blk * S + sp] = v;
        }
    }
}

// *ngroups = partials per layer written (pw[L][*ngroups][S*S], pb[L][*ngroups][S])
hipError_t launch_tokmix_wgrad(const float* da, const float* xh1, const float* l1a, const float* l1b, float* pw, float* pb, int B, int S,
                               int layers, int* ngroups, hipStream_t st) {
    if (S > kTokMaxS) return hipErrorInvalidValue;
    const int nwg = (B + 4 * kTokSpv - 1) / (4 * kTokSpv);
    *ngroups = nwg;
    hipLaunchKernelGGL(k_tokmix_wgrad, dim3(nwg, layers), dim3(256), 0, st, da, xh1, l1a, l1b, pw, pb, S, B);
    return hipGetLastError();
}

// Channel-mix weight gradient from the product with x-hat: the forward operand was U2 = alpha2 * x-hat2 + beta2, so
//   dW[o][i] = sum_r dA2[r][o] U2[r][i] = alpha2[i] * (dA2^T x-hat2)[o][i] + beta2[i] * db[o],   db[o] = sum_r dA2[r][o] (the bias gradient).
// One thread per element of [L][512][512]; dw / db are addressed through the flat gradient array (per-layer stride lstride).
__global__ __launch_bounds__(256) void k_wch_affine(float* __restrict__ dw, const float* __restrict__ db, const float* __restrict__ l2a,
                                                    const float* __restrict__ l2b, long long lstride) {
    const int l = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;       // e = o * 512 + i
    const int o = e >> 9, i = e & 511;
    float* p = dw + (size_t)l * lstride + e;
    *p = fmaf(*p, l2a[l * kDm + i], l2b[l * kDm + i] * db[(size_t)l * lstride + o]);
}

hipError_t launch_wch_affine(float* dw, const float* db, const float* l2a, const float* l2b, long long lstride, int layers, hipStream_t st) {
    hipLaunchKernelGGL(k_wch_affine, dim3(kDm * kDm / 256, layers), dim3(256), 0, st, dw, db, l2a, l2b, lstride);
    return hipGetLastError();
}

// out[c] (+)= sum_{i<n} partial[i*stride + c] in a fixed order (deterministic): block = 64 columns x 16 index lanes,
// lane q sums i = q, q+16, ... with four independent accumulators, the 16 lanes are then combined through LDS in order.
// blockIdx.y selects an independent group (e.g. a layer): partial += y * pgstride, out += y * ogstride.
__global__ __launch_bounds__(1024) void k_partial_reduce(const float* __restrict__ partial, int n, long long stride, int cols,
                                                         float* __restrict__ out, int accumulate, long long pgstride, long long ogstride) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    partial += (size_t)blockIdx.y * pgstride;
    out += (size_t)blockIdx.y * ogstride;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        int i = q;
#pragma unroll 4                                    // 16 loads in flight per lane: these launches are pure latency
        for (; i + 48 < n; i += 64) {
            s0 += partial[(size_t)i * stride + c];
            s1 += partial[(size_t)(i + 16) * stride + c];
            s2 += partial[(size_t)(i + 32) * stride + c];
            s3 += partial[(size_t)(i + 48) * stride + c];
        }
        for (; i < n; i += 16) s0 += partial[(size_t)i * stride + c];
    }
    red[q][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && c < cols) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][cl];
        out[c] = accumulate ? out[c] + v : v;
    }
}

hipError_t launch_partial_reduce(const float* partial, int n, long long stride, int cols, float* out, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(k_partial_reduce, dim3((cols + 63) / 64), dim3(1024), 0, st, partial, n, stride, cols, out, accumulate, 0LL, 0LL);
    return hipGetLastError();
}

hipError_t launch_partial_reduce_groups(const float* partial, int n, long long stride, int cols, float* out, int groups, long long pgstride,
                                        long long ogstride, hipStream_t st) {
    hipLaunchKernelGGL(k_partial_reduce, dim3((cols + 63) / 64, groups), dim3(1024), 0, st, partial, n, stride, cols, out, 0, pgstride, ogstride);
    return hipGetLastError();
}

// Column sums of a row-strided matrix: partial[blk][c] = sum over this block's rows of in[(r / ri) * ro + (r % ri) * rs + c].
// Block = 64 columns x 4 row lanes; a following k_partial_reduce over blocks finishes the sum in a fixed order.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ in, int ri, long long ro, long long rs, int rows, int cols,
                                                float* __restrict__ partial, int rows_per_block) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + q; r < r1; r += 4) s += in[(size_t)(r / ri) * ro + (size_t)(r % ri) * rs + c];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < cols) partial[(size_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

hipError_t launch_colsum(const float* in, int ri, long long ro, long long rs, int rows, int cols, float* partial, int nblk, hipStream_t st) {
    const int rpb = (rows + nblk - 1) / nblk;
    hipLaunchKernelGGL(k_colsum, dim3((cols + 63) / 64, nblk), dim3(256), 0, st, in, ri, ro, rs, rows, cols, partial, rpb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// style token = reparameterize(mu, logvar) (RAG.py:10-13, :118-120), optional emotion token (scripts_beat/model/RAG.py:125),
// and the per-sample sum of (1 + lv - mu^2 - exp(lv)) for the KLD term (gaussian_diffusion.py:1392)
__global__ __launch_bounds__(512) void k_style_fwd(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                                                   const float* __restrict__ emo_w, const int64_t* __restrict__ emo, int emo_stride,
                                                   float* __restrict__ x0, float* __restrict__ kld_partial, int S, int NPRE) {
    __shared__ float red[8];
    const int b = blockIdx.x, c = threadIdx.x;
    const size_t i = (size_t)b * kDm + c;
    const float m = mu[i], l = lv[i];
    x0[((size_t)b * S) * kDm + c] = m + eps[i] * expf(0.5f * l);
    if (NPRE == 2) x0[((size_t)b * S + 1) * kDm + c] = emo_w[(size_t)emo[(size_t)b * emo_stride] * kDm + c];
    float v = wave_sum(1.0f + l - m * m - expf(l));
    if ((c & 63) == 0) red[c >> 6] = v;
    __syncthreads();
    if (c == 0) kld_partial[b] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

hipError_t launch_style_fwd(const float* mu, const float* lv, const float* eps, const float* emo_w, const int64_t* emo, int emo_stride,
                            float* x0, float* kld_partial, int B, int S, int NPRE, hipStream_t st) {
    hipLaunchKernelGGL(k_style_fwd, dim3(B), dim3(kDm), 0, st, mu, lv, eps, emo_w, emo, emo_stride, x0, kld_partial, S, NPRE);
    return hipGetLastError();
}

// d mu = g0 + kw * mu / (B*512);  d logvar = g0 * eps * 0.5 * exp(0.5 lv) - kw * 0.5 * (1 - exp(lv)) / (B*512)
__global__ __launch_bounds__(512) void k_style_bwd(const float* __restrict__ g0, const float* __restrict__ mu, const float* __restrict__ lv,
                                                   const float* __restrict__ eps, float* __restrict__ dmu, float* __restrict__ dlv, int S,
                                                   float kscale) {
    const int b = blockIdx.x, c = threadIdx.x;
    const size_t i = (size_t)b * kDm + c;
    const float gs = g0[((size_t)b * S) * kDm + c];
    const float l = lv[i];
    dmu[i] = gs + kscale * mu[i];
    dlv[i] = gs * eps[i] * 0.5f * expf(0.5f * l) - 0.5f * kscale * (1.0f - expf(l));
}

hipError_t launch_style_bwd(const float* g0, const float* mu, const float* lv, const float* eps, float* dmu, float* dlv, int B, int S,
                            float kld_weight, hipStream_t st) {
    hipLaunchKernelGGL(k_style_bwd, dim3(B), dim3(kDm), 0, st, g0, mu, lv, eps, dmu, dlv, S, kld_weight / ((float)B * kDm));
    return hipGetLastError();
}

// table[idx[i]][c] += src[i][c]: embedding gradients.  Deterministic with repeated indices and still parallel: block i
// does the work only if it is the FIRST occurrence of its index, and then adds every later occurrence in order.
__global__ __launch_bounds__(256) void k_scatter_rows(const float* __restrict__ src, long long src_stride, const int64_t* __restrict__ idx,
                                                      int idx_stride, int n, int cols, float* __restrict__ table) {
    const int i = blockIdx.x;
    const int64_t me = idx[(size_t)i * idx_stride];
    for (int j = 0; j < i; ++j)
        if (idx[(size_t)j * idx_stride] == me) return;            // uniform across the block
    for (int c = threadIdx.x; c < cols; c += 256) {
        float acc = table[(size_t)me * cols + c];
        for (int j = i; j < n; ++j)
            if (idx[(size_t)j * idx_stride] == me) acc += src[(size_t)j * src_stride + c];
        table[(size_t)me * cols + c] = acc;
    }
}

hipError_t launch_scatter_rows(const float* src, long long src_stride, const int64_t* idx, int idx_stride, int n, int cols, float* table,
                               hipStream_t st) {
    hipLaunchKernelGGL(k_scatter_rows, dim3(n), dim3(256), 0, st, src, src_stride, idx, idx_stride, n, cols, table);
    return hipGetLastError();
}

__global__ void k_scale_rows(float* __restrict__ x, const float* __restrict__ drop, int per_sample, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= 1.0f - drop[i / per_sample];
}

hipError_t launch_scale_rows(float* x, const float* drop, int B, int per_sample, hipStream_t st) {
    const size_t n = (size_t)B * per_sample;
    hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, drop, per_sample, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// compute_huber on poses and on frame-to-frame velocities (gaussian_diffusion.py:21-24, :1381-1387) and d loss / d out.
// out / dout: [(b,t)][JF]; x_start: [b][JF][t].  partial[block] = {sum huber(rot), sum huber(vel)} (before * beta / N)
__global__ __launch_bounds__(256) void k_loss(const float* __restrict__ out, const float* __restrict__ x_start, float* __restrict__ dout,
                                              float* __restrict__ partial, TrainDims d, float lambda_vel) {
    __shared__ float red[2][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float beta = 0.1f, ib = 10.0f;
    const float in_rot = 1.0f / ((float)d.B * d.JF * d.T), in_vel = lambda_vel / ((float)d.B * d.JF * (d.T - 1));
    float srot = 0.f, svel = 0.f;
    if (i < d.B * d.JF) {
        const int b = i / d.JF, j = i % d.JF;
        const float* xs = x_start + (size_t)i * d.T;
        float po = 0.f, px = 0.f, gprev = 0.f;
        for (int t = 0; t < d.T; ++t) {
            const size_t oi = ((size_t)b * d.T + t) * d.JF + j;
            const float o = out[oi], x = xs[t];
            const float z = (o - x) * ib, az = fabsf(z);
            srot += az < 1.f ? 0.5f * z * z : az - 0.5f;
            float gcur = fminf(fmaxf(z, -1.f), 1.f) * in_rot;
            if (t > 0) {
                const float zv = ((o - po) - (x - px)) * ib, azv = fabsf(zv);
                svel += azv < 1.f ? 0.5f * zv * zv : azv - 0.5f;
                const float gv = fminf(fmaxf(zv, -1.f), 1.f) * in_vel;
                gcur += gv;
                gprev -= gv;
                dout[oi - d.JF] = gprev;
            }
            gprev = gcur;
            po = o; px = x;
        }
        dout[((size_t)b * d.T + d.T - 1) * d.JF + j] = gprev;
    }
    (void)beta;
    srot = wave_sum(srot); svel = wave_sum(svel);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = srot; red[1][threadIdx.x >> 6] = svel; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

hipError_t launch_loss(const float* out, const float* x_start, float* dout, float* partial, TrainDims d, float lambda_vel, hipStream_t st) {
    hipLaunchKernelGGL(k_loss, dim3((d.B * d.JF + 255) / 256), dim3(256), 0, st, out, x_start, dout, partial, d, lambda_vel);
    return hipGetLastError();
}

// terms = {rot_mse, vel_mse, kld, loss = rot + lambda_vel*vel, total = loss + kld_weight*kld}  (train_loop.py:178)
__global__ void k_finish_terms(const float* __restrict__ lp, int n_loss, const float* __restrict__ kp, int n_kld, float* __restrict__ terms,
                               TrainDims d, float lambda_vel, float kld_weight) {
    if (blockIdx.x != 0) return;
    // one wave: lane l adds entries l, l + 64, ... in index order, then a fixed butterfly (a lone thread walking all 566 entries took
    // 31 us of dependent loads)
    const int l = threadIdx.x;
    double r = 0.0, v = 0.0, k = 0.0;
    for (int i = l; i < n_loss; i += 64) { r += lp[2 * i]; v += lp[2 * i + 1]; }
    for (int i = l; i < n_kld; i += 64) k += kp[i];
    for (int o = 32; o >= 1; o >>= 1) { r += __shfl_xor(r, o); v += __shfl_xor(v, o); k += __shfl_xor(k, o); }
    if (l != 0) return;
    const float rot = (float)(r * 0.1 / ((double)d.B * d.JF * d.T));
    const float vel = (float)(v * 0.1 / ((double)d.B * d.JF * (d.T - 1)));
    const float kld = (float)(-0.5 * k / ((double)d.B * kDm));
    terms[0] = rot; terms[1] = vel; terms[2] = kld;
    terms[3] = rot + lambda_vel * vel;
    terms[4] = terms[3] + kld_weight * kld;
}

hipError_t launch_finish_terms(const float* loss_partial, int n_loss, const float* kld_partial, int n_kld, float* terms, TrainDims d,
                               float lambda_vel, float kld_weight, hipStream_t st) {
    hipLaunchKernelGGL(k_finish_terms, dim3(1), dim3(64), 0, st, loss_partial, n_loss, kld_partial, n_kld, terms, d, lambda_vel, kld_weight);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Audio encoder backward helpers (audio_enc.py:9-20); the stride-6 layers' gradients are implicit GEMMs in ls_conv.hip.
// dst[b][c][r] = src[b][r][c] for R <= 64 rows: conv4's output gradient arrives as [B][T][256] (rows of the feature GEMM's dA) and
// the implicit-GEMM gradients stage [channel][position] tiles.  Workgroup = (64 channels, sample); both sides coalesced.
__global__ __launch_bounds__(256) void k_transpose_rc(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    __shared__ float tile[64][65];
    const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    const float* s = src + (size_t)b * R * C;
    for (int i = tid; i < R * 64; i += 256) tile[i >> 6][i & 63] = s[(size_t)(i >> 6) * C + c0 + (i & 63)];
    __syncthreads();
    float* d = dst + ((size_t)b * C + c0) * R;                    // 64 channel rows of R floats: one contiguous block
    for (int o = tid; o < R * 64; o += 256) d[o] = tile[o % R][o / R];
}

hipError_t launch_transpose_rc(const float* src, float* dst, int B, int R, int C, hipStream_t st) {
    if (R > 64 || C % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_transpose_rc, dim3(C / 64, B), dim3(256), 0, st, src, dst, R, C);
    return hipGetLastError();
}

// conv1 (Cin = 1, k 15, stride 5, pad 1600) weight gradient: dW1[co][k] = sum_{b,p} dC1[b][co][p] * wav[b][5p + k - 1600].
// STAND-ALONE form: the training step folds these products into the epilogue of conv2's data gradient (k_conv_dgrad<FUSE1>,
// ls_conv.hip) and never writes dy; this kernel is what tools/conv_bwd_bench.cpp checks that against and times beside it.
// dC1 is never materialised: k_conv_dgrad left dy = dAct * lrelu'(y) and per-row partial sums of dy and dy*y, and the InstanceNorm
// backward d c1 = rstd * (dy - mean(dy) - y * mean(dy*y)), y = (c_raw - mean) * rstd, is two FMAs per element here:
// d c1 = a dy + b c_raw + c with per-row a = rstd, b = -rstd^2 mean(dy*y), c = -rstd mean(dy) - b mean (saves a read-modify-write
// pass over the 517 MB tensor).
// Round 3: the products run on the matrix pipe (rounds 1-2: 480 threads x 256 FMAs with one ds_read_b32 each behind the staging:
// 361 us at B = 512 for two 517 MB streams).  Workgroup = (256-position chunk, sample): every wave instruction of the staging reads
// 256 contiguous bytes of one channel row, the d c1 tile [32][256] and the chunk's waveform window go to LDS; then MFMA M axis = 16
// channels, N axis = the 15 taps (+ one dead column), K = 4 positions: a lane's A values are one ds_read_b128 per four steps (step e
// of group M multiplies positions 16M + 4g + e), the B values lane base + immediates in the window.  A workgroup walks four chunks
// with the next one's 70 loads per thread in flight behind the current one's MFMAs.
// partial[(b, chunk)][co*15 + k]; a following k_partial_reduce sums them in index order.
constexpr int kC1P = 256, kC1Ld = kC1P + 4, kC1Win = kC1P * 5 + 16;

// coef[row] = (a, b, c) of d c1 = a dy + b c_raw + c from the row's statistics and its partial sums of dy and dy*y (fixed order)
__global__ __launch_bounds__(256) void k_in_bwd_coef(const float* __restrict__ stats, const float* __restrict__ rowpart, int nslot, int rows, int L,
                                                     float* __restrict__ coef) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float sa = 0.f, sc2 = 0.f;
    const float2* rp = reinterpret_cast<const float2*>(rowpart) + (size_t)row * nslot;
#pragma unroll 8
    for (int i = 0; i < nslot; ++i) {                                      // fixed order; unrolled so that the loads are in flight together
        const float2 v = rp[i];
        sa += v.x; sc2 += v.y;
    }
    const float mean = stats[(size_t)row * 2], rstd = stats[(size_t)row * 2 + 1];
    const float m1 = sa / (float)L, m2 = sc2 / (float)L;
    const float bb = -rstd * rstd * m2;
    coef[(size_t)row * 4] = rstd;
    coef[(size_t)row * 4 + 1] = bb;
    coef[(size_t)row * 4 + 2] = -rstd * m1 - bb * mean;
    coef[(size_t)row * 4 + 3] = 0.f;
}

constexpr int kC1Run = 4;                               // chunks per workgroup: one accumulator set, one partial, the next chunk's loads in flight
__global__ __launch_bounds__(256) void k_conv1_wgrad(const float* __restrict__ dy, const float* __restrict__ craw, const float* __restrict__ coef,
                                                     const float* __restrict__ wav, float* __restrict__ partial, int Lin, int Lout, int pad) {
    __shared__ __attribute__((aligned(16))) float dcs[32 * kC1Ld];
    __shared__ float wavs[kC1Win];
    __shared__ float rowc[32][4];                        // a, b, c of the 32 (b, co) rows
    __shared__ float red[4][32][16];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.x * kC1Run, c1 = min(c0 + kC1Run, (Lout + kC1P - 1) / kC1P);
    // a chunk's values: position p0 + tid of every channel (clamped: branch-free, all in flight together) and 5 window values
    float vd[32], vc[32], vw[6];
    const float* wb = wav + (size_t)b * Lin;
    auto fetch = [&](int c) {
        const int p0 = c * kC1P;
        const size_t o = (size_t)b * 32 * Lout + min(p0 + tid, Lout - 1);
#pragma unroll
        for (int ch = 0; ch < 32; ++ch) { vd[ch] = dy[o + (size_t)ch * Lout]; vc[ch] = craw[o + (size_t)ch * Lout]; }
        const int x0 = p0 * 5 - pad;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int x = x0 + tid + 256 * q;
            const float v = wb[min(max(x, 0), Lin - 1)];
            vw[q] = (x >= 0 && x < Lin) ? v : 0.f;
        }
    };
    if (tid < 128) rowc[tid >> 2][tid & 3] = coef[(size_t)b * 128 + tid];
    fetch(c0);
    f4 acc[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
    const float* bw = wavs + 5 * (64 * w + 4 * g) + s16;  // + 80 M + 5 e
    const float* aw = dcs + s16 * kC1Ld + 64 * w + 4 * g; // + 16 mt rows, + 16 M
    for (int c = c0; c < c1; ++c) {
        __syncthreads();                                 // the previous chunk's MFMAs have read dcs / wavs (first pass: rowc is written)
        {
            const bool inside = c * kC1P + tid < Lout;
#pragma unroll
            for (int ch = 0; ch < 32; ++ch) {
                const float v = fmaf(rowc[ch][0], vd[ch], fmaf(rowc[ch][1], vc[ch], rowc[ch][2]));
                dcs[ch * kC1Ld + tid] = inside ? v : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) if (tid + 256 * q < kC1Win) wavs[tid + 256 * q] = vw[q];
        }
        __syncthreads();
        if (c + 1 < c1) fetch(c + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int M = 0; M < 4; ++M) {
            const f4 A0 = *reinterpret_cast<const f4*>(aw + 16 * M), A1 = *reinterpret_cast<const f4*>(aw + 16 * kC1Ld + 16 * M);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float Bv = bw[80 * M + 5 * e];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[e], Bv, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[e], Bv, acc[1], 0, 0, 0);
            }
        }
    }
    // lane (k = s16, g) holds channels 16 mt + 4 g + e; the four waves' sums are combined in wave order
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[w][16 * mt + 4 * g + e][s16] = acc[mt][e];
    __syncthreads();
    for (int i = tid; i < 480; i += 256) {
        const int co = i / 15, k = i - 15 * co;
        partial[((size_t)b * gridDim.x + blockIdx.x) * 480 + i] = ((red[0][co][k] + red[1][co][k]) + red[2][co][k]) + red[3][co][k];
    }
}

hipError_t launch_in_bwd_coef(const float* stats, const float* rowpart, int nslot, int rows, int L, float* coef, hipStream_t st) {
    hipLaunchKernelGGL(k_in_bwd_coef, dim3((rows + 255) / 256), dim3(256), 0, st, stats, rowpart, nslot, rows, L, coef);
    return hipGetLastError();
}

hipError_t launch_conv1_wgrad(const float* dy, const float* craw, const float* stats, const float* rowpart, int nslot, const float* wav,
                              float* partial, int B, int Lin, int Lout, int stride, int pad, int* nchunk, hipStream_t st) {
    if (stride != 5) return hipErrorInvalidValue;
    const int nc = ((Lout + kC1P - 1) / kC1P + kC1Run - 1) / kC1Run;      // workgroup runs per sample
    *nchunk = nc;
    // the per-row coefficients once (rounds 1-2 and this round's first form: every one of the 31 chunk workgroups of a sample summed
    // the 42 partials of its 32 rows again, serially, before its first barrier -- 190 of 496 us)
    float* coef = partial + (size_t)B * nc * 480;
    launch_in_bwd_coef(stats, rowpart, nslot, B * 32, Lout, coef, st);
    hipLaunchKernelGGL(k_conv1_wgrad, dim3(nc, B), dim3(256), 0, st, dy, craw, coef, wav, partial, Lin, Lout, pad);
    return hipGetLastError();
}

// per-lane MFMA operand image of a stride-6 conv weight (layout documented in ls_conv.hip), rebuilt after every optimiser step
__global__ void k_build_conv_img(const float* __restrict__ w, float* __restrict__ img, int Cin, int Cout) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)Cout * Cin * 15;
    if (i >= total) return;
    const int cig = (int)(i & 3), lane = (int)((i >> 2) & 63);
    size_t r = i >> 8;
    const int k = (int)(r % 15); r /= 15;
    const int nchunk = Cin / 16;
    const int ch = (int)(r % nchunk), ct = (int)(r / nchunk);
    const int co = 16 * ct + (lane & 15), ci = 16 * ch + 4 * cig + (lane >> 4);
    img[i] = w[((size_t)co * Cin + ci) * 15 + k];
}

hipError_t launch_build_conv_img(const float* w, float* img, int Cin, int Cout, hipStream_t st) {
    const size_t total = (size_t)Cout * Cin * 15;
    hipLaunchKernelGGL(k_build_conv_img, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, img, Cin, Cout);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Per-lane MFMA operand images of the mixer weights for the fused training forward (k_step TRAIN variant), rebuilt from
// the flat master parameters after every optimiser step.  Same layouts as ls_api.cpp build_images, except that LN2's
// affine is NOT folded into the channel-mix weights (alpha2 / beta2 are trained):
//   wch[l][w][p][q][c2][lane][j] = W_l[n = 64w + 16(2p+c2) + (lane&15)][k = 16q + 4(lane>>4) + j]
//   ww[l][t][m][lane] = blockdiag(Wt_l, Wt_l)[r = 16t + (lane&15)][r' = 4m + (lane>>4)],  btok[l][r] = bt_l[r % S]
__global__ void k_build_train_images(const TrainImgArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int R = 2 * a.S;
    if (i < (size_t)a.L * kDm * kDm) {
        const int j = (int)(i & 3), lane = (int)((i >> 2) & 63), c2 = (int)((i >> 8) & 1), q = (int)((i >> 9) & 31);
        const int p = (int)((i >> 14) & 1), w = (int)((i >> 15) & 7), l = (int)(i >> 18);
        const int n = 64 * w + 16 * (2 * p + c2) + (lane & 15), k = 16 * q + 4 * (lane >> 4) + j;
        a.wch[i] = a.P[a.base + (long long)l * a.lstride + a.o_w + (long long)n * kDm + k];
        a.wchT[i] = a.P[a.base + (long long)l * a.lstride + a.o_w + (long long)k * kDm + n];       // operand of dU = dA . W
    }
    if (i < (size_t)a.L * 5 * a.MK * 64) {
        const int lane = (int)(i & 63);
        size_t rr = i >> 6;
        const int m = (int)(rr % a.MK); rr /= a.MK;
        const int t = (int)(rr % 5), l = (int)(rr / 5);
        const int r = 16 * t + (lane & 15), rp = 4 * m + (lane >> 4);
        float v = 0.f;
        float vT = 0.f;
        if (r < R && rp < R && r / a.S == rp / a.S) {
            v = a.P[a.base + (long long)l * a.lstride + a.o_wt + (long long)(r % a.S) * a.S + (rp % a.S)];
            vT = a.P[a.base + (long long)l * a.lstride + a.o_wt + (long long)(rp % a.S) * a.S + (r % a.S)];
        }
        a.ww[i] = v;
        a.wwT[i] = vT;
    }
    if (i < (size_t)a.L * 80) {
        const int l = (int)(i / 80), r = (int)(i % 80);
        a.btok[i] = r < R ? a.P[a.base + (long long)l * a.lstride + a.o_bt + (r % a.S)] : 0.f;
    }
    if (i < (size_t)a.L * kDm) {
        const int l = (int)(i / kDm), c = (int)(i % kDm);
        const long long lb = a.base + (long long)l * a.lstride;
        a.bch[i] = a.P[lb + a.o_b + c];
        a.l1a[i] = a.P[lb + a.o_a1 + c]; a.l1b[i] = a.P[lb + a.o_b1 + c];
        a.l2a[i] = a.P[lb + a.o_a2 + c]; a.l2b[i] = a.P[lb + a.o_b2 + c];
    }
}

hipError_t launch_build_train_images(const TrainImgArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.L * kDm * kDm;
    hipLaunchKernelGGL(k_build_train_images, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// torch.optim.AdamW (decoupled weight decay, bias-corrected), train_loop.py:57-59
__global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                        float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
}

hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                        float bc1, float bc2, hipStream_t st) {
    hipLaunchKernelGGL(k_adamw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, sqrtf(bc2));
    return hipGetLastError();
}

}  // namespace ls
