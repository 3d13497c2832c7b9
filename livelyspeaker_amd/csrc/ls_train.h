// Training-step internals: strided GEMM arguments and kernel launchers shared by
// ls_gemm.hip, ls_train_kernels.hip and ls_train_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstddef>
#include <cstdint>

namespace ls {

// element (r, k) of an operand lives at p[(r / ri) * ro + (r % ri) * rs + (k / ki) * ko + (k % ki) * ks]
struct GemmOperand {
    const float* p;
    int ri; long long ro, rs;
    int ki; long long ko, ks;
    int vec;                 // float4 loads along the contiguous index are legal (set by gemm_operand())
};

struct GemmArgs {
    GemmOperand A, B;        // C[m][n] = sum_k A(m,k) * B(n,k)
    float* C;                // element (m, n) at C[(m / cri) * cro + (m % cri) * crs + n * cns]; Cpre and R share the addressing
    int cri; long long cro, crs, cns;
    const float* bias;       // [N] or null
    float* Cpre;             // pre-activation copy or null
    const float* R;          // residual added after the activation or null
    int act;                 // 0 none, 1 SiLU
    int accumulate;          // C += result
    int M, N, K, kchunk;
    float* ws;               // split-K workspace [batch][split][M][N]
    size_t ws_floats;
    // batched form: blockIdx.z = batch * splits + split; problem b reads A.p + b*bsA, B.p + b*bsB and writes C + b*bsC
    // (Cpre / R follow C's batch stride; the bias is shared by all problems of the batch)
    int nbatch, splits;
    long long bsA, bsB, bsC;
    // LayerNorm fused around the product (long-sequence channel mixing, ls_long.hip; LDS-DMA variant only).  With W' = W diag(alpha)
    // and bias' = bias + W beta folded on the host, LN(x) W^T = rstd (x W'^T - mean * wsum) + bias', wsum[n] = sum_k W'[n][k]: the
    // product runs on the RAW rows and the normalisation is two per-row scalars in the epilogue.
    const float* ln_part;    // [M][8][2] (mean, M2) of the eight 64-column groups of every A row, or null
    const float* wsum;       // [N]
    const float* addn;       // [N] vector added to every output row after the residual (the next block's `x += emb`), or null
    float* part_out;         // [M][N / 64][2] (mean, M2) of every 64-column group of the OUTPUT rows (the next LayerNorm's partials), or null
};

inline GemmOperand gemm_operand(const float* p, int ri, long long ro, long long rs, int ki, long long ko, long long ks, bool kcontig,
                                int rows, int K) {
    GemmOperand o{p, ri, ro, rs, ki, ko, ks, 0};
    const bool aligned = ((uintptr_t)p & 15) == 0;
    auto m4 = [](long long v) { return (v & 3) == 0; };
    if (kcontig) o.vec = aligned && ks == 1 && (ki >= K || (m4(ki) && m4(ko))) && m4(rs) && (ri >= rows || m4(ro));
    else o.vec = aligned && rs == 1 && (ri >= rows || (m4(ri) && m4(ro))) && m4(ks) && (ki >= K || m4(ko));
    return o;
}
// row-major matrix [rows][ld], reduction index contiguous
inline GemmOperand op_rows(const float* p, long long ld, int rows, int K) { return gemm_operand(p, INT_MAX, 0, ld, INT_MAX, 0, 1, true, rows, K); }
// the same memory read down its columns: operand row r = matrix column r, reduction index = matrix row
inline GemmOperand op_cols(const float* p, long long ld, int rows, int K) { return gemm_operand(p, INT_MAX, 0, 1, INT_MAX, 0, ld, false, rows, K); }

hipError_t launch_gemm_tr(GemmArgs a, bool a_kcontig, bool b_kcontig, int splits, hipStream_t st);

struct TrainDims { int B, S, T, NPRE, JF, KF, KFP, D, L; };   // KF = 2*JF+1+256 input_mapping fan-in, KFP = padded to 32

// ---- forward ----
hipError_t launch_build_feat_train(const float* x_start, const float* noise, const float* origin_x, const float* c4, const float* drop,
                                   const float* ca, const float* cb, float* feat, float* x_t, TrainDims d, int n_pre_seq, hipStream_t st);
hipError_t launch_style_fwd(const float* mu, const float* lv, const float* eps, const float* emo_w, const int64_t* emo, int emo_stride,
                            float* x0, float* kld_partial, int B, int S, int NPRE, hipStream_t st);
hipError_t launch_loss(const float* out, const float* x_start, float* dout, float* partial, TrainDims d, float lambda_vel, hipStream_t st);
hipError_t launch_finish_terms(const float* loss_partial, int n_loss, const float* kld_partial, int n_kld, float* terms, TrainDims d,
                               float lambda_vel, float kld_weight, hipStream_t st);
// mixer weight images for the fused training forward (k_step TRAIN variant): flat-parameter offsets in, images out
struct TrainImgArgs {
    const float* P;
    long long base, lstride, o_a1, o_b1, o_wt, o_bt, o_a2, o_b2, o_w, o_b;     // layer 0 base, per-layer stride, offsets inside a layer
    int L, S, MK;
    float* wch; float* bch; float* ww; float* btok; float* l1a; float* l1b; float* l2a; float* l2b;
    float* wchT; float* wwT;      // transposed images for the fused backward (dU = dA . W)
};
hipError_t launch_build_train_images(const TrainImgArgs& a, hipStream_t st);
// fused mixer backward (ls_train_bwd.hip): saved activations in, dA2 / dA1 (weight-gradient operands) and partials out
struct MixerBwdArgs {
    float* g;                 // [B*S][512] in: d loss / d mixer output; out: d loss / d token sequences entering layer 0
    const float* a2; const float* a1; const float* x2; const float* x1; const float* s2; const float* s1;   // [L][B*S][512 | 2]; x1 / x2 = x-hat
    float* da2; float* da1;   // [L][B*S][512]
    float* colpart;           // [workgroups][L][5][512]: d bias(block2), d alpha2, d beta2, d alpha1, d beta1 partial column sums
    float* dembp;             // [L][B][512] per-layer d(timestep embedding)
    const float* wchT_img; const float* wwT_img; const float* ln2a; const float* ln1a;   // images / [L][512]
    int B, layers;
};
hipError_t init_mixer_bwd();
hipError_t launch_mixer_bwd(Variant v, const MixerBwdArgs& a, hipStream_t st);
// ---- backward ----
hipError_t launch_silu_bwd_colsum(const float* g, const float* apre, float* da, float* partial, int rows, int nwaves, hipStream_t st);
// token-weight gradient partials of all layers (MFMA): pw[L][*ngroups][S*S], pb[L][*ngroups][S]; xh1 = saved x-hat of LayerNorm 1,
// l1a / l1b [L][512]
hipError_t launch_tokmix_wgrad(const float* da, const float* xh1, const float* l1a, const float* l1b, float* pw, float* pb, int B, int S,
                               int layers, int* ngroups, hipStream_t st);
// dW[l][o][i] = dW[l][o][i] * alpha2[l][i] + beta2[l][i] * db[l][o] (dW came from the product with x-hat instead of U2)
hipError_t launch_wch_affine(float* dw, const float* db, const float* l2a, const float* l2b, long long lstride, int layers, hipStream_t st);
hipError_t launch_partial_reduce(const float* partial, int n, long long stride, int cols, float* out, int accumulate, hipStream_t st);
// the same for `groups` independent (partial, out) pairs in one launch: group y reads partial + y*pgstride, writes out + y*ogstride
hipError_t launch_partial_reduce_groups(const float* partial, int n, long long stride, int cols, float* out, int groups, long long pgstride,
                                        long long ogstride, hipStream_t st);
// partial[nblk][cols]; follow with launch_partial_reduce(partial, nblk, cols, cols, ...)
hipError_t launch_colsum(const float* in, int ri, long long ro, long long rs, int rows, int cols, float* partial, int nblk, hipStream_t st);
hipError_t launch_style_bwd(const float* g0, const float* mu, const float* lv, const float* eps, float* dmu, float* dlv, int B, int S,
                            float kld_weight, hipStream_t st);
hipError_t launch_scatter_rows(const float* src, long long src_stride, const int64_t* idx, int idx_stride, int n, int cols, float* table,
                               hipStream_t st);
hipError_t launch_scale_rows(float* x, const float* drop, int B, int per_sample, hipStream_t st);
hipError_t launch_build_inmap_images(const float* w, float* wpad, float* waT, int KF, int KFP, int a0, hipStream_t st);
// ---- audio encoder backward ----
// dst[b][c][r] = src[b][r][c], R <= 64, C % 64 == 0
hipError_t launch_transpose_rc(const float* src, float* dst, int B, int R, int C, hipStream_t st);
// conv1 weight gradient with the InstanceNorm backward of its output applied on the fly (dy + row partials from
// launch_conv_dgrad(..., finalize = false)); partial[(b, chunk)][480]; *nchunk = chunks per sample
hipError_t launch_conv1_wgrad(const float* dy, const float* craw, const float* stats, const float* rowpart, int nslot, const float* wav,
                              float* partial, int B, int Lin, int Lout, int stride, int pad, int* nchunk, hipStream_t st);
// implicit-GEMM weight gradient of a stride-6 conv layer (ls_conv.hip); dC element (b, co, p) at dc[b*sb + co*sc + p];
// partial[*ngroups][Cout][Cin*15] with *ngroups <= conv_wgrad_groups(Cin, Cout) whatever the batch
int conv_wgrad_groups(int Cin, int Cout);
hipError_t launch_conv_wgrad(const float* dc, long long sb, long long sc, const float* in, const float* stats, float* partial, int B,
                             int Cin, int Cout, int Lin, int Lout, int* ngroups, hipStream_t st);
// implicit-GEMM data gradient + LeakyReLU' + InstanceNorm backward of a stride-6 conv layer (ls_conv.hip)
hipError_t launch_build_dgrad_img(const float* w, float* img, int Cin, int Cout, hipStream_t st);
// finalize = false leaves dy in dc_out and the per-row partial sums in partial[row][*nslot][2] for a fused consumer
hipError_t launch_conv_dgrad(const float* dc_in, long long sb, long long sc, long long sp, const float* wimg, const float* craw,
                             const float* stats, float* dc_out, float* partial, int B, int Cin, int Cout, int Lx, int Lout, bool finalize,
                             int* nslot, hipStream_t st);
// conv2's data gradient with conv1's weight gradient folded into its epilogue (ls_conv.hip, k_conv_dgrad<FUSE1>): no dy tensor
int wav_moment_parts(int Lout);                                                                                      // parts per sample
hipError_t launch_wav_moments(const float* wav, float* mom, int B, int Lw, int Lout, int pad, hipStream_t st);      // mom[B][parts][16][16]
hipError_t launch_in_bwd_coef(const float* stats, const float* rowpart, int nslot, int rows, int L, float* coef, hipStream_t st);
hipError_t launch_conv_dgrad_conv1(const float* dc_in, long long sb, long long sc, long long sp, const float* wimg, const float* craw,
                                   const float* stats, float* rowpart, int B, int Cout, int Lx, int Lout, const float* wav, int Lw, int wpad,
                                   const float* mom, const float* w1, const float* bias1, float* work, float** out_part, hipStream_t st);
// forward + data-gradient operand images of conv2..conv4 in one launch
hipError_t launch_build_conv_imgs(const float* const w[3], float* const img[3], float* const dimg[3], const int Cin[3], const int Cout[3], hipStream_t st);
hipError_t launch_build_conv_img(const float* w, float* img, int Cin, int Cout, hipStream_t st);
// ---- optimiser ----
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                        float bc1, float bc2, hipStream_t st);

}  // namespace ls
