// Internal declarations shared by the host side (ls_api.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ls {

constexpr int kD = 512;        // latent_dim (RAG.py:38; parser default 512)
constexpr int kWaves = 8;      // waves per workgroup of the step kernel
constexpr int kCB = 4;         // 16-channel blocks owned by one wave  (8 waves * 4 * 16 = 512)
constexpr int kNT = 5;         // 16-row token tiles per workgroup     (2*S = 70|72 rows -> 80)
constexpr int kUStride = 520;  // LDS row stride (floats) of the staged GEMM operand: conflict-free ds_read_b128
constexpr int kT = 34;         // frames
constexpr int kAudioFeat = 256;
constexpr int kPeRows = 5000;
constexpr int kProfPoints = 96;  // PositionalEncoding max_len (mlp_module.py:105)

enum SamplerKind { kDDPM = 0, kDDIM = 1, kNone = 2 };

// Per-call values that may change between replays of a captured graph live in device memory.
struct CallParams {
    unsigned long long seed;
    unsigned long long sample_offset;
    unsigned tag_base;          // sample-split kernel: added to every hand-off tag of the call, advanced by the host per sampling call, so a
    unsigned pad_;              // granule left by an EARLIER call can never pass for this call's (whatever the zeroing ahead of the loop did)
};

// Weight images in MFMA operand order (built by ls_api.cpp build_images); lives in device memory so
// the kernel fetches each pointer with one s_load where it is used instead of pinning ~22 SGPR pairs.
struct DevWeights {
    const float* wch_img;    // [L][8][2 passes][32 q][2 cb][64 lanes][4]   channel-mix Linear(512,512)
    const float* bch;        // [L][512]
    const float* wsum;       // [L][512] row sums of the LN2-folded channel-mix weights (sample-split kernel: LayerNorm 2 around the product)
    // bf16x3 split-precision mode: W' = hi + lo with hi = bf16(W'), lo = bf16(W' - hi); operand order of
    // v_mfma_f32_16x16x32_bf16 (lane (n, g) holds k = 32q + 8g .. +7): [L][8][2 passes][16 q][2 cb][64 lanes][8]
    const unsigned short* wch_hi_img;
    const unsigned short* wch_lo_img;
    const unsigned short* ww_hi_img;   // [L][5][KS][64][8]  block-diagonal token weights, bf16 hi / lo planes
    const unsigned short* ww_lo_img;
    const float* ln1a; const float* ln1b; const float* ln2a; const float* ln2b;   // [L][512]
    const float* ww_img;     // [L][5][MK][64]   block-diagonal token-mix operand
    const float* wtok1_img;  // [L][3][ceil(S/16)][64][4]   token-mix operand of ONE pass (sample-split kernel, ls_coop_kernel.h)
    const float* wtail;      // [L][S][4]   Wt[32 + i][k]: the ragged output rows' token-mix weights (one-pass-per-workgroup kernel, fp32)
    const unsigned short* wtok1_hi_img;   // [L][3][ceil(S/32)][64][8]  the same as bf16 hi / lo planes (one-pass-per-workgroup kernel, bf16x3)
    const unsigned short* wtok1_lo_img;
    const float* btok_rows;  // [L][80]
    const float* winx_img;   // [8][2 passes][KXQ][2 cb][64][4]   x_t columns of input_mapping
    const float* wout_img;   // [NOB][32][64][4]   poseFinal, k in natural order (operand staged in LDS)
    const float* wout_reg_img; // [8][NOB][4][64][4] poseFinal, k in residual-register order (operand = registers)
    const float* bout;       // [NOB*16]
};

// Arguments of one launch of the fused step kernel (one workgroup = one sample = cond+uncond rows).
struct StepArgs {
    // activations (internal layout [B][T][JF])
    const float* x_in;
    float* x_out;        // x_{t-1}                       (sampler != kNone)
    float* x0_out;       // pred_xstart (CFG-combined)    (nullable)
    float* fwd_c;        // raw cond / uncond model outputs (nullable)
    float* fwd_u;
    // prepared conditioning
    const float* static_c;   // [B][T][512]  W_in[:,JF:] . [prefix poses | bit | audio] + b
    const float* static_u;   // same with the audio term masked
    const float* z_mu;       // [B][512]
    const float* z_std;      // [B][512] exp(0.5*logvar)
    const float* emo_tok;    // [B][512] or null
    const float* scale;      // [B]
    const float* temb;       // timestep embedding row(s)
    int temb_stride;         // floats between samples (0: one row shared by the batch)
    // noise
    const float* eps_c;      // [B][512] or null -> Philox
    const float* eps_u;
    const float* noise;      // [B][JF][T] (reference layout) or null -> Philox
    int const_noise;
    const CallParams* call;  // Philox key / sample offset
    unsigned step_id;        // Philox stream selector
    const struct DevWeights* W;   // weight images (device memory, constant per model)
    int layers;
    int batch;               // samples in this launch (PAIR variant: two per workgroup, the last one may be single)
    // sampler update
    int sampler;             // SamplerKind
    int t_nonzero;           // 1[t != 0]
    int clip_denoised;       // clamp pred_xstart to [-1,1] (process_xstart, gaussian_diffusion.py:365-371)
    float c0, c1, c2, c3, c4;
    // DDPM: x' = c0*x0 + c1*x_t + nz*c2*noise
    // DDIM: eps = (c0*x_t - x0)/c1 ; x' = x0*c2 + c3*eps + nz*c4*noise
    // ---- training-forward variant (k_step<..., TRAIN = 1>, ls_train_api.cpp): a workgroup holds samples 2b (rows 0..S-1) and
    // 2b+1 (rows S..2S-1) of a single pass and writes every tensor the backward needs straight from registers.
    // All are [L][tr_B*S][512] (stats [L][tr_B*S][2]); temb/temb_stride give one timestep-embedding row per SAMPLE.
    // tr_x1 / tr_x2: the NORMALISED inputs x-hat = (x - mean) * rstd of LayerNorm 1 / 2 (pre-affine); tr_a1 / tr_a2: the
    // pre-activations of the two mixes; tr_s1 / tr_s2: (mean, rstd).
    const float* tr_x0;      // [tr_B*S][512] token sequences entering layer 0
    float* tr_x1; float* tr_a1; float* tr_x2; float* tr_a2;
    float* tr_s1; float* tr_s2;
    float* tr_xout;          // [tr_B*S][512] output of the last layer
    int tr_B;
    float* trace;            // [B][L+1][2S][512] or null
    // ---- sample-split kernel (k_coop, ls_coop_kernel.h): exchange workspaces of ONE launch (kCoopMaxGroups (sample, pass) groups)
    float* cx;               // [groups][32 k blocks of 16 channels][36 rows][16] rows entering channel mixing, centred on the LayerNorm-1 mean
    float* cpart;            // [groups][NS slices][36][J*F padded to 16s] partial poseFinal outputs of each slice (NS = 8 | 4 | 2)
    unsigned long long* cgran;   // [groups][2 areas][36 rows][NS slices][2] {tag, value} granules: (mean, M2) partials of the two LayerNorms
    unsigned long long* cflag;   // [samples][16] {tag, -} ready flags of the final rows
    unsigned* cerr;          // set non-zero by a workgroup whose bounded spin ran out
    unsigned epoch;          // tag base of this launch: unique among the launches since the granule words were last zeroed
    int b0;                  // first sample of this launch
    int npass;               // 2: cond + uncond (CFG); 1: cond only (every guidance scale is 1)
    int ngroups;             // (sample, pass) groups of this launch
    int xmap;                // blockIdx -> (group, slice) mapping, see k_coop
    // ---- one-pass-per-workgroup kernel (k_pass, ls_pass_kernel.h): b0 / npass as above, plus the CFG hand-off of ONE launch
    float* pf;               // [samples][2 passes][T][J*F] poseFinal output of each pass (write-through)
    unsigned* pcnt;          // [samples] arrival tickets: zeroed by ls_prepare, back at zero after every step
    int xpad_ready;          // batch-level / long-sequence path: the previous step of this loop left x_in's padded copy behind (k_long_update), no k_long_padx needed
#ifdef LS_DEBUG
    // Profiling builds only (tools/phase_profile.py, tools/ab_variants.py compile their own -DLS_DEBUG variant of the library):
    // the shipped library has neither the fields nor the code that reads them, so no environment variable can change its results.
    unsigned long long* prof;  // [8 waves][kProfPoints] s_memtime stamps of workgroup prof_wg (env LS_PROF)
    int prof_wg;
    unsigned long long* wgt;   // [workgroups][2] s_memtime at the start / end of EVERY workgroup (tools/wg_timeline.py), or null
    int ablate;              // env LS_ABLATE: 1 skip channel-mix MFMAs, 2 skip token-mix, 4 skip LN stats (results are wrong)
#endif
};

// dataset variant of the compiled kernel
enum Variant { kTED = 0, kBEAT = 1 };

// Arguments of one diffusion step of the long-sequence path (ls_long.hip): same roles as StepArgs, plus the batch-level workspaces.
struct LongStepArgs {
    int B, T, S, npre, JF, JFP, ldo, layers;
    int b0;                                // index of this launch's first sample in the prepared batch (its Philox streams); pointers are already shifted
    int tokpad;                            // token axis of the wtp image: 48 (S <= 48) or 160
    const float* x_in; float* x_out; float* x0_out; float* fwd_c; float* fwd_u;      // internal layout [B][T][JF]
    const float* static_c; const float* static_u; const float* z_mu; const float* z_std; const float* emo_tok; const float* scale;
    const float* temb;                     // one row (the timestep is uniform over the batch in sampling)
    const float* eps_c; const float* eps_u; const float* noise; int const_noise;
    const CallParams* call; unsigned step_id;
    // weights (row-major, as in the state dict)
    const float* winx;                     // [512][JFP]  x_t columns of input_mapping
    const float* ln1a; const float* ln1b; const float* ln2a; const float* ln2b;      // [L][512]
    const float* wt; const float* bt;      // [L][S][S], [L][S]   token-mixing Conv1d(S, S, 1)
    const float* wtp;                      // [L][160 x 160] the same, zero-padded, in k_long_tokmix's per-lane fragment order, or null
    const float* wc; const float* bc;      // [L][512][512], [L][512]
    const float* wcf; const float* bcf; const float* wsum;   // fused form (S <= 160): Wc diag(alpha2), bc + Wc beta2, row sums of the folded weight
    float* part1; float* part2;            // [rows padded to 128][8][2] (mean, M2) partials of LayerNorm 1 / 2 per 64-channel group
    const float* wout; const float* bout;  // [ldo = JF padded to 128s][512] (zero rows beyond JF), [JF]
    // workspaces
    float* xproj;                          // [B*T rounded up to 128][512]
    float* xpad;                           // [B*T rounded up to 128][JFP]  x_t with zero pad columns / rows (operand of the projection)
    float* X; float* U;                    // [2*B*S][512]
    float* OUT;                            // [2*B*S][ldo]
    // the eight blocks in one launch per resident set of groups (ls_mix_kernel.h) instead of sixteen batch-level launches: when mix_cap > 0
    const float* mix_wtok; const float* mix_wch;   // per-lane operand images (MixArgs)
    const float* mix_wpose; float* mix_pout; int mix_npt;     // poseFinal inside the mixer (mix_pout != null): partial products per slice, summed by k_long_update
    float* mix_xg; unsigned long long* mix_gran; unsigned* mix_err;
    unsigned mix_epoch0;                   // tag base of this step's first mixer launch (one kCoopEpochStride per launch)
    int mix_cap;                           // (sample, pass) groups per launch (4 workgroups each, all resident); 0 = batch-level kernels
#ifdef LS_DEBUG
    unsigned long long* prof; int prof_wg;
#endif
    int xpad_ready;                        // xpad already holds x_in (written by the previous step's k_long_update): skip k_long_padx
    int sampler, t_nonzero, clip_denoised;
    float c0, c1, c2, c3, c4;
};
hipError_t launch_step_long(const LongStepArgs& a, hipStream_t st);

constexpr int kMixSlices = 4;               // slice workgroups of a (sample, pass) group
constexpr int kMixRows = 160;               // padded rows (ten 16-row tiles)
// long-sequence mixer kernel (ls_mix.hip): the eight MLPblocks of up to 160 tokens in one launch, four slice workgroups per (sample, pass)
struct MixArgs {
    const float* x_in;          // [groups][S][512] token sequences entering block 0 (group = launch-local (pass, sample) index g0 + ...)
    float* x_out;               // same layout, after block L - 1
    const float* temb;          // [512] timestep embedding row (the timestep is uniform over the batch in sampling)
    const float* ln1a; const float* ln1b;     // [L][512]
    const float* wtok_img;      // [L][10 q][10 mt][64][4]: Wt[16 mt + s16][16 q + 4 e + g], zero beyond S
    const float* btok;          // [L][S]
    const float* wch_img;       // [L][32 gb][32 q][64][4]: W'[16 gb + s16][16 q + 4 g + j], W' = W diag(alpha2)
    const float* bch; const float* wsum;      // [L][512] folded bias, row sums of W'
    float* xg;                  // [groups][32 k blocks][160][16] exchange: rows entering channel mixing, centred on the LayerNorm-1 mean
    unsigned long long* gran;   // [groups]([2 areas][160 rows] + 1)[4 slices][2] {tag, value} granules: LayerNorm partials, rows-ready flags
    unsigned* err;              // set non-zero by a workgroup whose bounded spin ran out
    const CallParams* call;
    unsigned epoch;             // tag base of this launch
    int ngroups, layers;
    long long group_stride;     // floats between groups in x_in / x_out (S * 512)
    // assembly inside the kernel (xproj != null; x_in is then unused): what k_long_assemble read
    // poseFinal inside the kernel (pout != null; x_out is then not written): every slice multiplies its own 128 channels into all output columns,
    // the four partial products are summed (in slice order) by the update kernel
    const float* wpose_img;     // [npt column tiles][32 q][64][4]: Wout[16 nb + s16][16 q + 4 g + j], zero rows beyond JF
    float* pout;                // [all groups of the batch][4 slices][S][16 npt] partial poseFinal products (no bias)
    int npt;                    // 16-column tiles of the output (JF = 282: 18); at most 20 (five per wave)
    const float* xproj;         // [B * T][512] x_t columns of input_mapping
    const float* static_c; const float* static_u; const float* z_mu; const float* z_std; const float* emo_tok; const float* eps_c; const float* eps_u;
#ifdef LS_DEBUG
    unsigned long long* prof; int prof_wg;      // phase stamps of one workgroup (tools/mix_profile.py)
#endif
    int g0, B, b0, npre;        // first group of this launch (group = pass * B + sample), batch, index of sample 0 in the prepared batch (Philox), prefix tokens
    unsigned step_id;
};

bool mix_supports(int S);
hipError_t init_mix_kernels();
hipError_t launch_mix(int S, const MixArgs& a, hipStream_t st);

// sample-split step kernel for small batches (ls_coop.hip): 16 workgroups per sample (2 passes x 8 channel slices of 4 waves)
constexpr int kCoopMaxGroups = 64;     // (sample, pass) groups of one launch: 512 workgroups = two per CU, all resident at once
constexpr unsigned kCoopEpochStride = 64;   // hand-off tags per launch: 2 * layers + 1 of them are used, so layers <= 31 (the reference: 8)
hipError_t init_coop_kernels();
// ncb = 1 | 2 | 4: 8 | 4 | 2 slice workgroups per (sample, pass) (16-channel blocks per wave)
hipError_t launch_step_coop(Variant v, int ncb, const StepArgs& a, int nsamples, hipStream_t st);

// one-pass-per-workgroup step kernel (ls_pass.hip): npass workgroups of 4 waves per sample, two workgroups per CU
hipError_t init_pass_kernels();
hipError_t launch_step_pass(Variant v, int prec, int waves, const StepArgs& a, int nsamples, hipStream_t st);

// prec: 0 = exact fp32 MFMA (default), 1 = bf16x3 split-precision channel mixing (opt-in, parity-gated at 1e-3)
// pair: 0 = CFG (cond + uncond pass of one sample per workgroup), 1 = single pass (guidance scale 1: two samples per workgroup)
hipError_t launch_step(Variant v, int prec, int pair, const StepArgs& a, int batch, hipStream_t st);
hipError_t launch_step_ted(int prec, int pair, const StepArgs& a, int batch, hipStream_t st);     // ls_step.hip
hipError_t launch_step_beat(int prec, int pair, const StepArgs& a, int batch, hipStream_t st);    // ls_step_beat.hip
hipError_t init_step_kernels_ted();
hipError_t init_step_kernels_beat();
// training forward of the mixer: ceil(tr_B / 2) workgroups (ls_step.hip, TRAIN variant)
hipError_t launch_train_mixer_fwd(Variant v, const StepArgs& a, hipStream_t st);
size_t step_lds_bytes(Variant v);
hipError_t init_step_kernels();

// ---- once-per-call kernels (ls_prepare.hip) ------------------------------------------------
// stride-6 layers on MFMA (ls_conv.hip); wimg = per-lane operand image.  out_stats != null: the InstanceNorm statistics of
// the output are produced by the same pass (spart: workspace, B*Cout*ceil(Lout/64)*12 floats)
hipError_t launch_conv1d_mfma(const float* in, const float* stats, const float* wimg, const float* bias, float* out, float* out_stats,
                              float* spart, int B, int Cin, int Cout, int Lin, int Lout, hipStream_t st);
// conv1 (Cin = 1, k 15, stride 5) + output statistics in one pass (spart: B*32*ceil(Lout/256)*12 floats)
hipError_t launch_conv1_fwd(const float* wav, const float* w, const float* bias, float* out, float* out_stats, float* spart, int B, int Lin,
                            int Lout, int pad, hipStream_t st);
// C[M][N] = act(A[M][K] . W[N][K]^T + bias) (+ R): fp32 MFMA GEMM (ls_gemm.hip); act 3 = exact GELU
hipError_t launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr,
                          float* C, int ldc, int M, int N, int K, int act, hipStream_t st);
// out[r] = table[idx[r * idx_stride]] (rows clamped into the table)
hipError_t launch_gather_rows(const float* table, const int64_t* idx, float* out, int rows, int width,
                              int table_rows, hipStream_t st, int idx_stride = 1);
// feat_p [B*T][KPP] = [prefix poses | bit | pad], feat_a [B*T][256] = audio feature (see ls_prepare.hip)
hipError_t launch_build_feats(const float* origin_x, const float* conv4, float* feat_p, float* feat_a,
                              int B, int JF, int KPP, int n_pre_seq, hipStream_t st, int T = kT);
hipError_t launch_split_style(const float* ml, float* mu, float* lv, float* sd, int B, hipStream_t st);
hipError_t launch_to_internal(const float* src_bjft, float* dst_btc, int B, int JF, hipStream_t st, int T = kT);
hipError_t launch_from_internal(const float* src_btc, float* dst_bjft, int B, int JF, hipStream_t st, int T = kT);
hipError_t launch_q_sample(const float* x0, const float* noise, float* out, size_t n, float a, float b,
                           hipStream_t st);
// posterior / DDIM update with PER-SAMPLE coefficients (ls_step with `indices`): x_t, x0, out in the internal [B][T][JF] layout, noise
// in the reference layout [B][JF][T]; table [n_steps][8] = {1[t != 0], c0, c1, c2, c3, c4, -, -} with the meanings of StepArgs, one
// row per schedule index; indices [B] (device) selects each sample's row (clamped into the table)
hipError_t launch_sampler_update(const float* x_t, const float* x0, const float* noise, const float* table, const int64_t* indices,
                                 int n_steps, float* out, int B, int JF, int T, int sampler, hipStream_t st);
// p_mean_variance's inpainting branch + process_xstart + the sampler update, after a denoiser launch with sampler = kNone: x0 (internal
// layout, in: CFG-combined model output, out: pred_xstart) = maskf ? given : x0, given = motion (internal) or qa * motion + qb * n with
// n = inoise (reference layout) or the Philox stream 4 when `renoise`; then clamp, optional dump copy, DDPM / DDIM update as in k_step
struct InpaintArgs {
    const float* x_t; float* x0; const float* maskf; const float* motion; const float* inoise; const float* noise; float* out; float* dump;
    const CallParams* call; unsigned step_id;
    int JF, T, renoise, const_noise, sampler, t_nonzero, clip;
    float qa, qb, c0, c1, c2, c3, c4;
};
hipError_t launch_inpaint_update(const InpaintArgs& a, int B, hipStream_t st);
// bytes [n] -> 0.f / 1.f
hipError_t launch_bytes_to_float(const unsigned char* src, float* dst, size_t n, hipStream_t st);
hipError_t launch_randn_fill(float* out_btc, int B, int JF, const CallParams* call, unsigned stream_id,
                             hipStream_t st, int T = kT);
hipError_t launch_transpose_feat(const float* conv4, float* out_btc, int B, hipStream_t st, int T = kT);

// ---- SAG decoder kernels (ls_sag.hip) ----------------------------------------------------------
hipError_t launch_sag_queries(const float* x, const float* wmap, const float* bmap, const float* pe, float* q, float* qc, int B,
                              int JF, int n_pre, int D, hipStream_t st);
hipError_t launch_sag_attention(const float* qkv, float* out, int B, int heads, int D, int n_pre_c, hipStream_t st);
// y = LayerNorm(x (+ bc[row / T], rows of bc bc_stride floats apart))
hipError_t launch_layernorm512(const float* x, const float* bc, int bc_stride, const float* w, const float* beta, float* y, int rows,
                               hipStream_t st);
hipError_t launch_layernorm512x2(const float* x, const float* w1, const float* b1, const float* bc, int bc_stride, const float* w2,
                                 const float* b2, float* y, int rows, hipStream_t st);
hipError_t launch_sag_final(const float* xh, const float* wf, const float* bf, const unsigned char* mask, float* out, int B,
                            int JF, int D, hipStream_t st);

}  // namespace ls
