/*
 * ls_hip.h -- C-ABI of the MI355X-native RAG denoising / diffusion-sampling engine.
 *
 * The reference (zyhbili/LivelySpeaker) is pure Python/PyTorch: it has no FFI, plugin registry
 * or operator API for this path.  Its "operator interface" is the pair
 *     model(x, timesteps, y=dict)                      scripts/model/RAG.py:98-133
 *     diffusion.p_sample_loop / ddim_sample_loop(...)  scripts/diffusion/gaussian_diffusion.py:608-671, 895-943
 * The entry points below are what a ctypes binding on the reference side would call to replace
 * those two; each cites the reference code it stands in for.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * LS_E* code and never throws across the ABI; ls_last_error() gives the message.  One handle owns
 * one GPU (one HIP stream, its captured hipGraphs, all device buffers); a handle is not
 * thread-safe, distinct handles are independent.  Pointers in ls_cond / ls_sample_args /
 * ls_forward_args are host pointers unless the struct's on_device flag is set, in which case they
 * are device pointers on the handle's GPU.  The library never frees or retains caller memory.
 * All tensors are fp32, C-contiguous, in the reference's layouts ([B, njoints, nfeats, nframes]).
 */
#ifndef LS_HIP_H
#define LS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LS_ABI_VERSION 5

enum {
    LS_OK = 0,
    LS_EINVAL = -1,   /* bad argument / shape / unknown key            */
    LS_ESTATE = -2,   /* call order (weights / schedule / prepare missing) */
    LS_EHIP = -3,     /* a HIP runtime call failed                     */
    LS_ENOMEM = -4,
    LS_EUNSUPPORTED = -5
};

enum { LS_SAMPLER_DDPM = 0, LS_SAMPLER_DDIM = 1 };
enum { LS_NOISE_TAPE = 0, LS_NOISE_PHILOX = 1 };
/* Arithmetic of the channel-mixing GEMM (92 % of the FLOPs). FP32 (default): v_mfma_f32_16x16x4_f32, exact fp32
 * products.  BF16X3 (opt-in): each fp32 operand split into bf16 hi+lo, three v_mfma_f32_16x16x32_bf16 per product
 * (hi.hi + hi.lo + lo.hi, fp32 accumulate): ~2^-16 relative product error, parity-gated at the 1e-3 contract. */
enum { LS_PRECISION_FP32 = 0, LS_PRECISION_BF16X3 = 1 };

typedef struct ls_handle ls_handle;

/* Static shape of the denoiser: RAG.__init__ (scripts/model/RAG.py:17-77; BEAT variant
 * scripts_beat/model/RAG.py:56,72-74) + get_model_args (scripts/mdm_utils/model_util.py:20-37). */
typedef struct ls_config {
    int32_t njoints;          /* 9 (TED) | 47 (BEAT)                               */
    int32_t nfeats;           /* 3 | 6                                             */
    int32_t nframes;          /* 34 = the reference's (token-mixing conv fixes it): fused step kernel.  Any other
                                 value selects the synthetic long-sequence path (e.g. 150 frames, BASELINE configs[4]'s
                                 wording; the reference cannot run it -- perf-only, checked against this repo's oracle) */
    int32_t n_prefix_tokens;  /* 1 = [style] | 2 = [style, emotion]                */
    int32_t n_pre_seq;        /* 4 prefix poses (RAG.py:70)                        */
    int32_t latent_dim;       /* 512                                               */
    int32_t layers;           /* 8                                                 */
    int32_t audio_len;        /* 36267 | 36266 raw samples -> 34 audio frames (must yield nframes) */
    int32_t n_speakers;       /* 1400 (RAG.py:65)                                  */
    int32_t n_emotions;       /* 0 | 8                                             */
    int32_t device;           /* HIP device ordinal                                */
    int32_t reserved;
} ls_config;

/* GaussianDiffusion.__init__ tables after SpacedDiffusion (gaussian_diffusion.py:168-204,
 * respace.py:74-88), fp64 as the reference keeps them; cast to fp32 per step exactly like
 * _extract_into_tensor (gaussian_diffusion.py:1651-1664).  Each array has n_steps entries. */
typedef struct ls_schedule {
    int32_t n_steps;
    int32_t reserved;
    const int64_t* timestep_map;                 /* _WrappedModel, respace.py:125-130 */
    const double* sqrt_alphas_cumprod;           /* q_sample, :240-258               */
    const double* sqrt_one_minus_alphas_cumprod;
    const double* posterior_mean_coef1;          /* q_posterior_mean_variance :260-282 */
    const double* posterior_mean_coef2;
    const double* posterior_log_variance_clipped;/* p_sample :507-558 (FIXED_SMALL)  */
    const double* alphas_cumprod;                /* ddim_sample :745-798             */
    const double* alphas_cumprod_prev;
    const double* sqrt_recip_alphas_cumprod;     /* _predict_eps_from_xstart :418-422 */
    const double* sqrt_recipm1_alphas_cumprod;
} ls_schedule;

/* model_kwargs['y'] of the callers (scripts/test_RAG_ted.py:64-70): only the keys RAG.forward
 * reads.  origin_x is NOT mutated here (the Python shim reproduces RAG.py:110's in-place zeroing). */
typedef struct ls_cond {
    int32_t batch;
    int32_t on_device;
    const float* audio_input;   /* [B, audio_len]                                  */
    const float* origin_x;      /* [B, J, F, T]; frames >= n_pre_seq are ignored   */
    const int64_t* vid_indices; /* [B] < n_speakers                                */
    const int64_t* emo;         /* [B, T] emotion ids as the callers hold y['emo'] (frame 0 is read,
                                   scripts_beat/model/RAG.py:125), or NULL (TED); same shape as ls_train_batch.emo */
    const float* scale;         /* [B] guidance scale (cfg_sampler.py:31)          */
} ls_cond;

/* One call of p_sample_loop / ddim_sample_loop with ClassifierFreeSampleModel as the model. */
typedef struct ls_sample_args {
    int32_t sampler;            /* LS_SAMPLER_*                                     */
    int32_t noise_mode;         /* LS_NOISE_*                                       */
    int32_t skip_timesteps;     /* gaussian_diffusion.py:712 / :982                 */
    int32_t const_noise;        /* :545-546 / :706-707 (TAPE mode only)             */
    int32_t on_device;
    int32_t use_graph;          /* 1: capture the step loop in a hipGraph and replay */
    int32_t clip_denoised;      /* clamp pred_xstart to [-1,1] (callers pass False)  */
    int32_t two_pass_always;    /* 0: when every scale == 1 the uncond pass is skipped (out_u + 1*(out_c - out_u) = out_c,
                                   cfg_sampler.py:31; the callers run guidance_param = 1); 1: always evaluate both passes.
                                   NOT bit-identical: in fp32 out_u + 1*(out_c - out_u) differs from out_c by up to one rounding
                                   of the larger term, so the two settings agree to ~1e-5 on samples (tests: <= 3e-4), not bitwise;
                                   with 0, ls_step / ls_forward-style raw outputs of the uncond pass are not produced. */
    float eta;                  /* DDIM eta (callers never pass it: 0)              */
    int32_t n_dump;             /* dump_steps (DDPM only, :660-671)                 */
    const int32_t* dump_steps;  /* executed-step counters (0 = first executed step), host memory */
    float* dump_out;            /* [n_dump, B, J, F, T] pred_xstart                 */
    const float* x_init;        /* [B,J,F,T] x_T = the loop's first randn; NULL only with PHILOX */
    const float* init_image;    /* [B,J,F,T] or NULL (zeros when skip_timesteps>0)  */
    const float* eps_tape;      /* TAPE: [n_exec, 2, B, latent_dim] style eps (cond, uncond) */
    const float* noise_tape;    /* TAPE: [n_exec, B, J, F, T] per-step randn_like(x) */
    uint64_t seed;              /* PHILOX key                                       */
    uint64_t sample_offset;     /* PHILOX: global index of sample 0 (shard-invariant streams) */
    float* out;                 /* [B, J, F, T]                                     */
    /* Segmented TAPE mode (seg_count > 0): this call runs the executed-step counters [seg_begin, seg_begin + seg_count) of the loop
     * and eps_tape / noise_tape hold THOSE steps only ([seg_count, 2, B, D] / [seg_count, B, J, F, T]) -- the reference's
     * "identical seeds" mode draws two style eps and one randn_like(x) per step from the host generator
     * (gaussian_diffusion.py:700-743, RAG.py:120), 4 GB for 512 clips x 1000 steps if drawn in one piece.  seg_begin == 0 starts the
     * loop (x_init / init_image are read then); x_t stays in the handle between segments; segments must follow each other in
     * order; `out` (and dump_out) are written by the segment that ends the loop, which is also the only one that waits for the GPU.
     * Host tapes (on_device == 0) are uploaded on the handle's COPY stream into a two-slot device buffer while the previous
     * segment's steps run; they should be page-locked, and a segment's host buffers may be reused as soon as the NEXT segment call
     * has returned.  Plain launches (use_graph is ignored).  seg_count == 0: the whole loop in one call (everything above). */
    int32_t seg_begin;
    int32_t seg_count;
    /* p_mean_variance's inpainting branch (gaussian_diffusion.py:314-320; BEAT tree scripts_beat/...:319), off while inpaint_mask is
     * NULL: every step's (CFG-combined) model output is replaced, where the mask is set, by the given motion -- re-noised with
     * q_sample(inpainted_motion, t - 1) while t > 0 when inpaint_noised (the TED tree: its randn_like is inpaint_noise[k] in TAPE mode,
     * the device stream 4 in PHILOX mode), as it is otherwise (the BEAT tree) -- before clip_denoised and the sampler update.  The loop
     * then runs the denoiser and the update as two launches per step.  Not combined with segments. */
    const unsigned char* inpaint_mask;   /* [B,J,F,T] bytes (non-zero = take the given motion), or NULL           */
    const float* inpainted_motion;       /* [B,J,F,T]                                                            */
    const float* inpaint_noise;          /* TAPE + inpaint_noised: [n_exec,B,J,F,T]; entries of steps with t == 0 unused */
    int32_t inpaint_noised;
    int32_t reserved2;
} ls_sample_args;

/* One RAG.forward pair (cond / uncond) and optionally the CFG combination, for model(x,t,y)
 * parity (RAG.py:98-133, cfg_sampler.py:24-31). Any of the three outputs may be NULL. */
typedef struct ls_forward_args {
    int32_t on_device;
    int32_t no_sync;            /* on_device only: return without waiting for the GPU (order consumers with ls_stream_order) */
    const float* x;             /* [B,J,F,T]                                        */
    const int64_t* timesteps;   /* [B] model-scale t in [0, 5000)                   */
    const float* eps_cond;      /* [B, latent_dim] randn_like of reparameterize (RAG.py:12) */
    const float* eps_uncond;    /* [B, latent_dim]                                  */
    float* out_cond;            /* [B,J,F,T]                                        */
    float* out_uncond;
    float* out_cfg;             /* out_u + scale*(out_c - out_u)                    */
    float* trace;               /* debug: [B, layers+1, 2*S, latent_dim] residual stream, or NULL */
} ls_forward_args;

/* One p_sample / ddim_sample step (gaussian_diffusion.py:507-558, 745-798) at schedule index i -- or, as the reference's
 * signature allows (`t` is a [B] tensor), at one schedule index PER SAMPLE (`indices`). */
typedef struct ls_step_args {
    int32_t sampler;
    int32_t index;              /* schedule index i (model sees timestep_map[i]); ignored when indices != NULL */
    int32_t on_device;
    float eta;
    int32_t clip_denoised;
    int32_t two_pass_always;    /* as in ls_sample_args                             */
    const float* x;             /* [B,J,F,T]                                        */
    const float* eps_cond;
    const float* eps_uncond;
    const float* noise;         /* [B,J,F,T]                                        */
    float* sample;              /* [B,J,F,T]                                        */
    float* pred_xstart;         /* [B,J,F,T] or NULL                                */
    const int64_t* indices;     /* [B] schedule index per sample, or NULL (uniform `index`).  When the entries differ the denoiser runs
                                   with one timestep-embedding row per sample and the posterior / DDIM update is a separate
                                   elementwise kernel with per-sample coefficients (same arithmetic as the fused epilogue).  HOST
                                   indices are validated (a constant vector takes the fused path); DEVICE indices
                                   (indices_on_device) are never read by the host -- no round trip in a step-by-step caller --
                                   always take the per-sample path and are clamped into [0, n_steps). */
    int32_t no_sync;            /* on_device only: return without waiting for the GPU (order consumers with ls_stream_order) */
    int32_t indices_on_device;
    const unsigned char* inpaint_mask;   /* as in ls_sample_args; uniform `index` only                          */
    const float* inpainted_motion;
    const float* inpaint_noise;          /* [B,J,F,T] the randn_like of q_sample(inpainted_motion, t - 1), or NULL = un-noised */
} ls_step_args;

typedef struct ls_timing {
    float prepare_ms;           /* last ls_prepare, GPU time (HIP events on the handle's stream); -1 while an ls_prepare_async is in flight */
    float loop_ms;              /* last ls_sample: first step launch .. last step done */
    float total_ms;             /* last ls_sample incl. layout conversion and copies */
    int32_t n_step_launches;
    int32_t graph_replayed;     /* 1 if the loop ran as a hipGraph replay           */
    int32_t single_pass;        /* 1 if the loop ran the single-pass (scale == 1) kernel */
    float tape_upload_ms;       /* segmented TAPE mode: summed GPU-side duration of the tape uploads of the last loop (copy stream) */
    int32_t n_segments;         /* segments the last loop ran in (1 = one call)     */
    int32_t step_path;          /* kernels the last loop's steps ran on: 0 one workgroup per sample (fused), 1 batch-level, 2 sample-split, 3 one workgroup per (sample, pass) */
    int32_t tail_samples;       /* a batch the plan splits (e.g. full fused rounds + a partial one): samples of the second piece, run on ... */
    int32_t tail_path;          /* ... 1 the batch-level, 2 the sample-split, 3 the one-pass-per-workgroup kernels (0: none) */
    int32_t tail2_samples;      /* third piece of the plan (e.g. 416 clips = 256 fused + 128 one-pass-per-workgroup + 32 sample-split) */
    int32_t tail2_path;
    int32_t n_cus;              /* compute units of the handle's device (hipDeviceProp.multiProcessorCount): what the plans and the
                                   sample-split kernel's residency are derived from */
    int32_t coop_slices;        /* slice workgroups per (sample, pass) the sample-split piece of the last loop ran with: 8 | 4 | 2 (0: the plan had no such piece);
                                   a long-sequence model (nframes != 34, step_path 1): 4 = its eight blocks ran in the one-launch mixer kernel, 0 = as batch-level launches */
} ls_timing;

int ls_abi_version(void);
int ls_create(const ls_config* cfg, ls_handle** out);
void ls_destroy(ls_handle* h);
const char* ls_last_error(const ls_handle* h);   /* h may be NULL: error of the last failed ls_create */

/* load_state_dict (scripts/mdm_utils/model_util.py:5-10): key = reference state-dict key, data =
 * fp32 host array of n elements.  '*.pe' buffers are accepted and ignored (recomputed).
 * ls_commit_weights builds the MFMA-ordered device images; it fails if a required key is missing. */
int ls_set_weight(ls_handle* h, const char* key, const float* data, size_t n);
int ls_commit_weights(ls_handle* h);

int ls_set_precision(ls_handle* h, int mode);             /* LS_PRECISION_*; default FP32 */
/* Which kernels a 34-frame model's steps run on.  0 (default): chosen per prepared batch -- the fused kernel gives every sample a
 * workgroup (= one CU: a step costs one CU's time for eight layers however small the batch); the sample-split kernel spreads a
 * sample over 16 workgroups (2 CFG passes x 8 channel slices) that exchange LayerNorm partials and rows through L2 inside ONE launch
 * per step, and takes the small batches; 1: always one workgroup per sample; 2: the batch-level kernels of the long-sequence path
 * (21 launches per step; exact fp32, both CFG passes always evaluated); 3: always the sample-split kernel (exact fp32); 4: always the
 * one-pass-per-workgroup kernel (a workgroup of 4 waves per (sample, CFG pass), two independent workgroups per CU, the passes combined by
 * the later of the two: half-CU granularity, exact fp32; grids of up to one workgroup per CU run as 8-wave workgroups, larger ones as
 * 4-wave workgroups, two per CU); 5: the same kernel with the 4-wave / two-per-CU form at EVERY grid size (what mode 4 and the plans of
 * a device with fewer CUs reach only beyond one workgroup per CU; this selector pins that form to the reference's fixtures); 6 / 7 / 8:
 * the sample-split kernel with its slicing forced to 4 / 2 / 8 channel slices per (sample, CFG pass) (mode 3 chooses per piece).  Same
 * arithmetic every way, different summation order: results agree to ~1e-5, not bitwise.  Takes effect at the next ls_prepare;
 * ls_timing.step_path reports what ran.
 * A long-sequence model (nframes != 34) always runs on the batch-level kernels, except that with 145..160 tokens the eight blocks and
 * poseFinal of a SAMPLING loop run in one launch of the sample-split mixer (ls_timing.coop_slices == 4; 3 launches per step instead of
 * 21) where every launch is at least 7/8 full: 28-32 / 60-64 / 92-96 clips on 256 compute units.  There mode 2 forces the batch-level
 * kernels and mode 3 the mixer at every batch size; the other modes return LS_EUNSUPPORTED.  Steps with per-sample timesteps (ls_step, ls_forward) keep
 * the batch-level kernels. */
int ls_set_path(ls_handle* h, int mode);
/* The plan mode 0 makes for `batch` clips on a device of `n_cus` compute units, without a handle or a GPU (what ls_prepare decides, exposed
 * for inspection and for the CPU test suite): out10 = {pieces, then (kernel family as in ls_timing.step_path, first clip, clips) for up to
 * three pieces}; *ms (nullable) = the step-time model's estimate.  beat: 0 TED / 1 BEAT cost table; single_pass: every guidance scale is 1. */
int ls_plan_query(int beat, int batch, int single_pass, int precision, int n_cus, int* out10, float* ms);
/* Slice workgroups per (sample, CFG pass) -- 8, 4 or 2 -- the sample-split kernel uses for a plan piece of `groups` (sample, pass) groups
 * (clips x 2 under CFG, clips x 1 in the single-pass form) on a device of `n_cus` compute units; negative error code otherwise. */
int ls_plan_coop_slices(int beat, int groups, int n_cus);
int ls_set_schedule(ls_handle* h, const ls_schedule* s);
int ls_prepare(ls_handle* h, const ls_cond* c);           /* once per sampling call */
/* The same, enqueued on the handle's stream WITHOUT waiting: later calls on this handle are ordered behind it, so the caller may
 * overlap it with work on another stream (LivelySpeaker: the SAG decode, scripts/test_LivelySpeaker_ted.py:88-113, needs none of it).
 * Device-resident inputs must stay valid until the next synchronising call on this handle; HOST inputs (on_device == 0) have been
 * copied out of the caller's buffers when the call returns (it waits for those copies, not for the kernels);
 * ls_timing.prepare_ms reads -1 until the work is known to be done. */
int ls_prepare_async(ls_handle* h, const ls_cond* c);
int ls_sample(ls_handle* h, const ls_sample_args* a);
int ls_forward(ls_handle* h, const ls_forward_args* a);
int ls_step(ls_handle* h, const ls_step_args* a);
/* elementwise q_sample (gaussian_diffusion.py:240-258) at schedule index i; pointers per on_device */
int ls_q_sample(ls_handle* h, int index, int on_device, size_t n, const float* x_start,
                const float* noise, float* out);

/* The x_T draw of PHILOX mode on its own (what ls_sample uses when x_init == NULL): out [batch,J,F,T] ~ N(0,1),
 * stream keyed by (seed, sample_offset + b). Lets tests check the device RNG's moments and shard-invariance. */
int ls_philox_x_init(ls_handle* h, int batch, uint64_t seed, uint64_t sample_offset, int on_device, float* out);

/* Read back a prepared intermediate into host memory (parity tests of the once-per-call stages):
 * "audio_feat" [B,T,256], "static_c"/"static_u" [B,T,D], "z_mu"/"z_logvar"/"z_std" [B,D],
 * "temb" [n_steps,D].  Returns the element count, or a negative error. */
long long ls_read(ls_handle* h, const char* name, float* host_out, size_t capacity);

/* Multi-GPU sharding of one batch (the reference's dist_util.py:18-41 is a stub): the path has no
 * per-step exchange, so a C-ABI integrator shards by (1) splitting the batch contiguously with ls_shard_range, (2) giving every
 * rank's handle the same weights (broadcast them with whatever communicator the host application has -- RCCL ncclBroadcast of
 * the arrays passed to ls_set_weight), (3) passing ls_sample_args.sample_offset = first so the PHILOX streams follow the global
 * sample index (results do not depend on the GPU count), (4) gathering the [count, J, F, T] outputs.  livelyspeaker_amd/shard.py
 * is that recipe over torch.distributed.  Rank r of `world` owns samples [first, first + count). */
int ls_shard_range(int64_t total, int32_t world, int32_t rank, int64_t* first, int64_t* count);
int ls_get_timing(const ls_handle* h, ls_timing* out);
int ls_synchronize(ls_handle* h);

/* Stream ordering instead of host synchronisation.  Every handle runs on its own non-blocking HIP stream, which nothing orders
 * against the caller's streams.  ls_stream_order(device, first, then): work enqueued on `then` AFTER this call waits for the work
 * enqueued on `first` BEFORE it (an event recorded on `first`, hipStreamWaitEvent on `then`); both are hipStream_t values
 * (NULL = the device's default stream).  A caller whose inputs were produced on its own stream orders (its stream, handle stream)
 * before an entry point instead of synchronising the host, and (handle stream, its stream) after a no_sync call before it
 * consumes device outputs.  The accessors return each handle's stream. */
int ls_stream_order(int device, void* first, void* then);
void* ls_stream(const ls_handle* h);

/* ---- torch's CPU normal stream, natively ("identical seeds" mode off the Python critical path) ----------------------------------
 * The reference draws every normal of a sampling loop from torch's global CPU generator (gaussian_diffusion.py:700-743; RAG.py:10-13,
 * 120), so torch.manual_seed(s) fixes the sample.  These two entry points make the same draws from the same state: `state` is the
 * 5056-byte blob of torch.get_rng_state() (mt19937 words + the cached Box-Muller sample), updated in place -- hand it back with
 * torch.set_rng_state().  Host memory only; no GPU involved.  n_threads workers share the transcendental part.
 * ls_trng_randn: torch.randn(n), float32 contiguous (n >= 16: the 16-block float Box-Muller; n < 16: the per-element double path).
 * ls_trng_fill_steps: the per-step draws of n_steps sampling steps in the reference's order -- randn(B,1,D) of the cond pass, of the
 * uncond pass, then randn_like(x) -- into eps [n_steps][2][B][D] and noise [n_steps][B][J][F][T]; x is the model-output-shaped view
 * (memory order [T][B][J][F], consumed in that order by torch's per-element path) except at the first step when first_contiguous.
 * variant: the float transform of the contiguous draws -- 0 = libm (torch's DEFAULT-capability kernel), 1..4 = the Cephes polynomials
 * of torch's AVX2 / AVX512 kernels under the four ways their multiply-adds can have been contracted to FMAs; livelyspeaker_amd finds
 * the variant that reproduces torch bit for bit once per process and keeps torch's generator if none does. */
int ls_trng_randn(uint8_t* state, size_t state_bytes, float* out, size_t n, int variant, int n_threads);
int ls_trng_fill_steps(uint8_t* state, size_t state_bytes, int B, int D, int J, int F, int T, int n_steps, int first_contiguous,
                       float* eps, float* noise, int variant, int n_threads);
/* The per-element double pairs whose samples are stored as floats are evaluated by a vectorised restatement and taken from it only when
 * every double within its error margin rounds to the same float; the others go through libm as in torch (ls_torch_rng.cpp).
 * ls_trng_stats: pairs evaluated / pairs sent back to libm so far in this process.  ls_trng_pairs_debug (tests): the vectorised
 * evaluation alone over np pairs (4 np words) next to libm's doubles; isa 0 / 1 / 2 = base / AVX2 / AVX-512 clone, -1 = this machine's. */
int ls_trng_stats(uint64_t* pairs, uint64_t* redone);
/* Long fills of the native stream start their generator threads from JUMPED mt19937 states (csrc/ls_mt_jump.h: t^J modulo the
 * characteristic polynomial, applied as an XOR of ~10 k windows of a 33-block expansion of the state) instead of behind one thread that
 * walks the whole stream.  ls_trng_set_jump(0) restores the sequential scout (returns the previous setting); ls_trng_jump_check compares
 * the two ways of reaching the state `words` (a multiple of 624) further on: 0 = identical. */
int ls_trng_set_jump(int on);
int ls_trng_jump_check(uint32_t seed, uint64_t words, int* support);
int ls_trng_pairs_debug(const uint32_t* words, int np, int isa, double* fast_c, double* fast_s, float* zc, float* zs, uint8_t* redo,
                        double* libm_c, double* libm_s);

/* ---- SAG decoder ---------------------------------------------------------------
 * Decoder_TRANSFORMER (scripts/model/motionclip_module.py:98-183), called as SAG.decoder(batch) at
 * scripts/test_LivelySpeaker_ted.py:88 to produce init_image for the RAG refine loop.  Separate handle: it is a
 * different network with its own checkpoint (SAG.pth, keys 'decoder.*' with the prefix stripped). */
typedef struct ls_sag ls_sag;
typedef struct ls_sag_config {
    int32_t njoints, nfeats, nframes;   /* 9, 3, 34                                   */
    int32_t latent_dim, ff_size;        /* 512, 1024                                  */
    int32_t num_layers, num_heads;      /* 3, 4                                       */
    int32_t n_pre_poses;                /* 4                                          */
    int32_t device;
    int32_t reserved;
} ls_sag_config;
int ls_sag_create(const ls_sag_config* cfg, ls_sag** out);
void ls_sag_destroy(ls_sag* h);
const char* ls_sag_last_error(const ls_sag* h);
int ls_sag_set_weight(ls_sag* h, const char* key, const float* data, size_t n);
int ls_sag_commit_weights(ls_sag* h);
/* batch['x'] [B,J,F,T], batch['z'] [B,latent] (CLIP text feature), batch['mask'] [B,T] bytes or NULL (all true)
 * -> batch['output'] [B,J,F,T] */
int ls_sag_decode(ls_sag* h, int batch, int on_device, const float* x, const float* z, const unsigned char* mask,
                  float* out);
/* The same, enqueued only (device pointers): returns without waiting for the GPU; `out` is complete once ls_sag_stream() has reached
 * this point (order consumers with ls_stream_order).  Lets a caller that iterates batches decode batch n + 1 while batch n is refined. */
int ls_sag_decode_async(ls_sag* h, int batch, const float* x, const float* z, const unsigned char* mask, float* out);
float ls_sag_last_decode_ms(const ls_sag* h);   /* GPU time of the last decode (HIP events on the handle's stream); waits for an asynchronous one */
void* ls_sag_stream(const ls_sag* h);

/* ---- caller-side post-processing of sampled clips ---------------------------------
 * scripts/test_RAG_ted.py:84-111 (layout change, mean add, per-bone normalisation, joint-angle change curve, motion
 * beats) and convert_dir_vec_to_pose (scripts/utils/data_utils.py:77-97).  Stateless; dataset constants are passed in. */
typedef struct ls_post_config {
    int32_t njoints;            /* direction vectors per frame: 9 (TED)                          */
    int32_t n_pairs;            /* angle pairs: 4 (test_RAG_ted.py:24-29)                        */
    int32_t n_pose_joints;      /* 10                                                            */
    float thres;                /* 0.03 (:32)                                                    */
    int32_t pair_a[8], pair_b[8];
    float change_angle[8];      /* (:30)                                                         */
    int32_t bone_parent[16], bone_child[16];   /* dir_vec_pairs (data_utils.py:13-14)            */
    float bone_len[16];
    float mean_dir_vec[48];     /* (:22)                                                         */
} ls_post_config;
/* sample [B,J,3,34] -> aligned [B,34,J*3], pose [B,34,n_pose_joints,3], angle_diff [B,34], beat_mask [B,34] (bytes);
 * any output may be NULL.  Pointers are device pointers iff on_device. */
int ls_ted_post(int device, int on_device, int batch, const ls_post_config* c, const float* sample, float* aligned,
                float* pose, float* angle_diff, unsigned char* beat_mask);

/* BEAT twin of the caller plumbing (scripts_beat/test_RAG_beat.py:86, 101): the sampled tensor [B,J,6,34] in the reference layout
 * -> decoded_motions [B,34,J*6] (`.permute(0, 3, 1, 2).reshape(tar_pose.shape)`) and pred_euler [B,34,J*3] in degrees
 * (`matrix_to_euler_angles(rotation_6d_to_matrix(.), "XYZ") / pi * 180`, scripts_beat/dataloaders/rot_utils.py:218-257, 513-534:
 * Gram-Schmidt on the two 3-vectors, then (atan2(-m12, m22), asin(m02), atan2(-m01, m00))).  Either output may be NULL. */
int ls_beat_post(int device, int on_device, int batch, int njoints, const float* sample, float* decoded, float* euler_deg);

/* ---- training step ----------------------------------------------------------------
 * One optimisation step of the RAG denoiser as TrainLoop.run_step runs it (scripts/train_utils/train_loop.py:146-186):
 *   x_t = q_sample(x_start, t, noise)                                  gaussian_diffusion.py:1281-1282
 *   out = RAG.forward(x_t, t, y) in training mode (mask_cond dropout)  scripts/model/RAG.py:79-133
 *   loss = huber(x_start, out) + lambda_vel * huber(velocities) + kld_weight * KLD(z_mu, z_logvar)
 *                                                                      gaussian_diffusion.py:1347-1396, train_loop.py:178
 *   backward through every parameter, then torch.optim.AdamW           train_loop.py:57-59, fp16_util.py:183-187
 * The trainer owns fp32 master parameters, gradients and Adam moments as FLAT device arrays with one layout
 * (ls_train_param_info); gradients are written to a caller-provided device array so that a data-parallel caller can
 * all-reduce it (RCCL) between ls_train_forward_backward and ls_train_adamw.  Random draws are inputs, in the
 * reference's order: t, noise = randn_like(x_start), drop = bernoulli(cond_mask_prob) [B], eps = randn_like(z_mu). */
typedef struct ls_trainer ls_trainer;
typedef struct ls_train_config {
    ls_config model;
    float lambda_vel;         /* 1.0  (parser_util.py:109)            */
    float kld_weight;         /* 0.01 (train_loop.py:178)             */
    int32_t diffusion_steps;  /* length of the q_sample tables        */
    int32_t reserved;
} ls_train_config;
typedef struct ls_train_batch {
    int32_t batch;
    int32_t on_device;            /* 1: tensor pointers below are device pointers (t stays a HOST pointer) */
    const float* x_start;         /* [B,J,F,T] motion                                            */
    const int64_t* t;             /* [B] HOST: diffusion timestep per sample (schedule_sampler)  */
    const float* noise;           /* [B,J,F,T]                                                   */
    const float* drop;            /* [B] 1.0 = drop the audio condition for this sample          */
    const float* eps;             /* [B,512]                                                     */
    const float* audio_input;     /* [B,audio_len]                                               */
    const float* origin_x;        /* [B,J,F,T]                                                   */
    const int64_t* vid_indices;   /* [B]                                                         */
    const int64_t* emo;           /* [B,T] or NULL (TED)                                         */
} ls_train_batch;
typedef struct ls_train_terms { float rot_mse, vel_mse, kld, loss, total; float fwd_ms, bwd_ms, reserved; } ls_train_terms;

int ls_train_create(const ls_train_config* cfg, ls_trainer** out);
void ls_train_destroy(ls_trainer* h);
const char* ls_train_last_error(const ls_trainer* h);
/* q_sample tables of GaussianDiffusion.__init__ (fp64, diffusion_steps entries) and _WrappedModel's timestep map */
int ls_train_set_schedule(ls_trainer* h, const double* sqrt_alphas_cumprod, const double* sqrt_one_minus_alphas_cumprod,
                          const int64_t* timestep_map);
/* parameter table: reference state-dict key, offset and element count inside the flat arrays */
int ls_train_param_count(const ls_trainer* h);
int64_t ls_train_flat_size(const ls_trainer* h);
int ls_train_param_info(const ls_trainer* h, int index, char* key, size_t key_cap, int64_t* offset, int64_t* numel);
int ls_train_set_weight(ls_trainer* h, const char* key, const float* data, size_t n);   /* host -> master params   */
int ls_train_get_weight(ls_trainer* h, const char* key, float* out, size_t n);           /* master params -> host   */
/* optimizer state for checkpoint / resume (opt%09d.pt, train_loop.py:222-227): which = 1 exp_avg, 2 exp_avg_sq */
int ls_train_get_moment(ls_trainer* h, int which, const char* key, float* out, size_t n);
int ls_train_set_moment(ls_trainer* h, int which, const char* key, const float* data, size_t n);
int64_t ls_train_get_step(const ls_trainer* h);
int ls_train_set_step(ls_trainer* h, int64_t step);
/* forward + loss + backward; grad: DEVICE array of ls_train_flat_size floats, overwritten */
int ls_train_forward_backward(ls_trainer* h, const ls_train_batch* b, float* grad, ls_train_terms* terms);
/* AdamW on the master parameters from a DEVICE gradient array; the step counter is the trainer's (starts at 1) */
int ls_train_adamw(ls_trainer* h, const float* grad, float lr, float beta1, float beta2, float eps, float weight_decay);
/* debug / test access to a named internal tensor of the last forward ("out" [B,T,JF], "x_t", "audio_feat" ...) */
int ls_train_read(ls_trainer* h, const char* what, float* out, size_t n);
void* ls_train_stream(const ls_trainer* h);

/* ---- FGD feature extractor -------------------------------------------------------
 * EmbeddingNet(...).pose_encoder in eval mode: poses [B, n_frames, pose_dim] -> latent mean [B, base]
 * (scripts/model/embedding_net.py:41-83, 261-270; BEAT HalfEmbeddingNet scripts_beat/model/motion_autoencoder.py:38-73,
 * 156-167).  Keys are the auto-encoder checkpoint's ('gen_dict'): pose_encoder.* are used, decoder.* / fc_logvar are
 * accepted and ignored.  Frechet distance / diversity on the features stay host code (ted_evaluator.py:61-152). */
typedef struct ls_eval ls_eval;
typedef struct ls_eval_config {
    int32_t pose_dim;   /* 27 TED                                   */
    int32_t n_frames;   /* 34                                       */
    int32_t base;       /* latent size: 32 TED | vae_length BEAT    */
    int32_t hidden1;    /* out_net widths: 256, 128 TED | 4*base, 2*base BEAT */
    int32_t hidden2;
    int32_t device;
} ls_eval_config;
int ls_eval_create(const ls_eval_config* cfg, ls_eval** out);
void ls_eval_destroy(ls_eval* h);
const char* ls_eval_last_error(const ls_eval* h);
int ls_eval_set_weight(ls_eval* h, const char* key, const float* data, size_t n);
int ls_eval_commit_weights(ls_eval* h);
int ls_eval_features(ls_eval* h, int batch, int on_device, const float* poses, float* feat);
void* ls_eval_stream(const ls_eval* h);

#ifdef __cplusplus
}
#endif
#endif /* LS_HIP_H */
