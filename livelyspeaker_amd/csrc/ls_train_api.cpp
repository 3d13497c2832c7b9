// C-ABI of the training step (include/ls_hip.h "training step").
// Orchestrates the forward (saving activations), the losses, the backward and AdamW on one HIP stream.
// Layout of the flat parameter / gradient / moment arrays: one entry per reference state-dict key, every entry starts
// on a 4-float boundary (float4 loads); the padding stays zero in all four arrays.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ls_hip.h"
#include "ls_internal.h"
#include "ls_train.h"

using namespace ls;

namespace {

std::string g_train_create_error;

struct Buf {
    void* p = nullptr;       // what the kernels read: the buffer's own allocation, or a caller's device tensor for the duration of a call
    void* own = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n > bytes) {
            if (own) { hipError_t e = hipFree(own); if (e != hipSuccess) return e; own = nullptr; bytes = 0; }
            hipError_t e = hipMalloc(&own, n);
            if (e != hipSuccess) return e;
            bytes = n;
        }
        p = own;
        return hipSuccess;
    }
    void alias(const void* q) { p = const_cast<void*>(q); }
    void release() { if (own) (void)hipFree(own); p = own = nullptr; bytes = 0; }
    float* f() const { return static_cast<float*>(p); }
};

struct Param { std::string key; int64_t off, n; };

const int kCin[4] = {1, 32, 64, 128}, kCout[4] = {32, 64, 128, 256}, kStride[4] = {5, 6, 6, 6}, kPad[4] = {1600, 0, 0, 0}, kKey[4] = {0, 3, 6, 9};
#ifndef LS_TRAIN_FORK_DEFAULT
#define LS_TRAIN_FORK_DEFAULT 1      // A/B builds: 0 = the whole backward on one stream
#endif
constexpr int kSpk = 256, kAud = 256, kNW = 1024;   // kD, kPeRows come from ls_internal.h; kNW: waves of the row-loop kernels

}  // namespace

struct ls_trainer {
    ls_train_config cfg{};
    TrainDims d{};
    int convL[5] = {0, 0, 0, 0, 0};
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    // The backward forks behind the mixer's data gradient: the mixer's parameter gradients (the batched 512 x 512 weight-gradient product,
    // token-mixing weights, LayerNorm / bias reductions, timestep embedder) run on `side` while the input stage and the WavEncoder backward run
    // on `stream`; the two meet again before the step's closing event.  The side branch has its own split-K and reduction workspaces.
    hipStream_t side = nullptr;
    hipEvent_t evs[2] = {nullptr, nullptr};
    bool fork = LS_TRAIN_FORK_DEFAULT != 0;
    // this step's batch: the caller's device tensors in place, or the handle's copies of host inputs (ls_train_forward_backward
    // synchronises before it returns, so nothing is read after the call)
    const float *audio_in = nullptr, *in_x = nullptr, *in_noise = nullptr, *in_origin = nullptr, *in_drop = nullptr, *in_eps = nullptr;
    const int64_t *in_vid = nullptr, *in_emo = nullptr;
    Buf wpad, waT;                                             // input_mapping.weight zero-padded to [512][KFP]; its audio columns transposed [256][512]
    Buf hostpack;                                              // q_sample coefficients + timestep rows of the batch: [ca | cb | tidx], one upload
    std::string err;
    std::vector<Param> table;
    std::map<std::string, int> index;
    int64_t flat = 0;
    Buf P, M, V;                         // master params, Adam moments
    int64_t adam_step = 0;
    bool have_sched = false;
    std::vector<double> sac, s1mac;
    std::vector<int64_t> tmap;
    Buf pe;
    // batch-sized buffers
    int capB = 0;
    Buf x_start, noise, drop, eps, audio, origin_x, vid, emo, ca, cb, tidx;
    Buf c[4], st[3], img[4], dimg[4], feat, x_t, zc, mu, lv, pe_rows, pre1, hid, emb, xcur;
    Buf X1, A1, X2, A2, S1, S2, dA2, dA1, colpart, dembp;      // [L][B*S][512] (S1/S2: [L][B*S][2]) written by the fused training forward; X1 / X2 hold x-hat
    Buf twch, tbch, tww, tbtok, tl1a, tl1b, tl2a, tl2b, tdevw, twchT, twwT;    // mixer weight images + the DevWeights block k_step reads
    TrainImgArgs img_args{};
    Buf out, dout, lossp, kldp, terms, G, part, pw, pb, demb, dmu, dlv, dzc, dhid, dAf, dAt, col, dc[3], wmom, ws, part2, ws2;
    size_t ws_floats = 0;
    int B = 0;
    bool have_forward = false;
};

namespace {

int fail(ls_trainer* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_train_create_error = buf;
    return code;
}

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail((h), LS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

void add_param(ls_trainer* h, const std::string& key, int64_t n) {
    h->index[key] = (int)h->table.size();
    h->table.push_back({key, h->flat, n});
    h->flat += (n + 3) / 4 * 4;
}

float* P(ls_trainer* h, const std::string& key) { return h->P.f() + h->table[h->index.at(key)].off; }
float* Gr(ls_trainer* h, float* grad, const std::string& key) { return grad + h->table[h->index.at(key)].off; }
std::string lk(int l, const char* s) { return "backbone.mlps." + std::to_string(l) + "." + s; }
std::string ck(int i, const char* s) { return "audio_encoder.feat_extractor." + std::to_string(kKey[i]) + "." + s; }

// Batch inputs: host arrays are copied in; device tensors are read in place (the call is synchronous, the caller's tensors outlive it,
// and no kernel writes them) -- ten device-to-device copies per step otherwise, the 74 MB waveform among them.
int ingest(ls_trainer* h, Buf& dst, const void* src, size_t bytes, bool on_device) {
    if (on_device && bytes && ((uintptr_t)src & 15) == 0) { dst.alias(src); return LS_OK; }       // (an unaligned view is copied)
    HIPCHK(h, dst.ensure(bytes ? bytes : 4));
    if (bytes) HIPCHK(h, hipMemcpyAsync(dst.p, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    return LS_OK;
}

int splits_for(int M, int N, int K) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int s = (768 + tiles - 1) / tiles;
    const int smax = K / 64 > 0 ? K / 64 : 1;        // at least two 32-deep K tiles per split (K / 256 left a 512 x 512 x 512 weight gradient on 32 workgroups: 25 us)
    if (s > smax) s = smax;
    return s < 1 ? 1 : s;
}

GemmArgs gemm(GemmOperand A, GemmOperand B, float* C, long long ldc, int M, int N, int K) {
    GemmArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.cri = INT_MAX; a.cro = 0; a.crs = ldc; a.cns = 1;
    a.M = M; a.N = N; a.K = K;
    return a;
}

// GEMM launch that splits K when the output has too few 128x128 tiles to fill the chip and the epilogue allows it
// (bias and accumulation are applied by the split-K reduction; activation / pre-activation copy / residual are not)
hipError_t gemm_run(ls_trainer* h, GemmArgs a, bool a_k, bool b_k) {
    int s = 1;
    const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (!a.act && !a.Cpre && !a.R && tiles < 256 && h->ws.p) {
        a.ws = h->ws.f(); a.ws_floats = h->ws_floats;
        s = splits_for(a.M, a.N, a.K);
        while (s > 1 && (size_t)s * a.M * a.N > h->ws_floats) --s;
    }
    return launch_gemm_tr(a, a_k, b_k, s, h->stream);
}

// weight gradient: C[n_out][n_in] = sum over rows of dY(:, n_out) * X(:, n_in); split over the row index
hipError_t wgrad(ls_trainer* h, GemmOperand dy_cols, GemmOperand x_cols, bool a_k, bool b_k, float* C, long long ldc, int M, int N, int K) {
    GemmArgs a = gemm(dy_cols, x_cols, C, ldc, M, N, K);
    a.ws = h->ws.f(); a.ws_floats = h->ws_floats;
    int s = splits_for(M, N, K);
    while (s > 1 && (size_t)s * M * N > h->ws_floats) --s;
    return launch_gemm_tr(a, a_k, b_k, s, h->stream);
}

// the same product for `nbatch` problems laid out at fixed strides (the per-layer weight gradients of the mixer): one launch, the
// K splits sized so that all problems together fill the chip once -- 8x fewer partial tiles than 8 separate launches
hipError_t wgrad_batched(ls_trainer* h, GemmOperand dy_cols, GemmOperand x_cols, float* C, long long ldc, int M, int N, int K, int nbatch,
                         long long bs_dy, long long bs_x, long long bs_c) {
    GemmArgs a = gemm(dy_cols, x_cols, C, ldc, M, N, K);
    a.ws = h->ws.f(); a.ws_floats = h->ws_floats;
    a.nbatch = nbatch; a.bsA = bs_dy; a.bsB = bs_x; a.bsC = bs_c;
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128) * nbatch;
    int s = (768 + tiles - 1) / tiles;
    const int smax = K / 64 > 0 ? K / 64 : 1;        // at least two 32-deep K tiles per split (K / 256 left a 512 x 512 x 512 weight gradient on 32 workgroups: 25 us)
    if (s > smax) s = smax;
    while (s > 1 && (size_t)nbatch * s * M * N > h->ws_floats) --s;
    return launch_gemm_tr(a, false, false, s < 1 ? 1 : s, h->stream);
}

int ensure_batch(ls_trainer* h, int B) {
    if (B <= h->capB) return LS_OK;
    const TrainDims& d0 = h->d;
    const size_t R = (size_t)B * d0.S, nx = (size_t)B * d0.JF * d0.T;
    const int* L = h->convL;
    auto E = [&](Buf& b, size_t floats) { return b.ensure(floats * sizeof(float)); };
    HIPCHK(h, E(h->x_start, nx)); HIPCHK(h, E(h->noise, nx)); HIPCHK(h, E(h->origin_x, nx)); HIPCHK(h, E(h->x_t, nx));
    HIPCHK(h, E(h->drop, B)); HIPCHK(h, E(h->eps, (size_t)B * kD)); HIPCHK(h, E(h->audio, (size_t)B * L[0]));
    HIPCHK(h, h->vid.ensure((size_t)B * 8)); HIPCHK(h, h->emo.ensure((size_t)B * d0.T * 8)); HIPCHK(h, h->tidx.ensure((size_t)B * 8));
    HIPCHK(h, E(h->ca, B)); HIPCHK(h, E(h->cb, B));
    for (int i = 0; i < 4; ++i) HIPCHK(h, E(h->c[i], (size_t)B * kCout[i] * L[i + 1]));
    // (dc[0], the gradient of conv1's output, is never materialised: its only consumer, conv1's weight gradient, is folded into the
    //  epilogue of conv2's data gradient -- 517 MB at B = 512)
    for (int i = 0; i < 3; ++i) { HIPCHK(h, E(h->st[i], (size_t)B * kCout[i] * 2)); if (i) HIPCHK(h, E(h->dc[i], (size_t)B * kCout[i] * L[i + 1])); }
    HIPCHK(h, E(h->wmom, (size_t)B * wav_moment_parts(L[1]) * 256));
    HIPCHK(h, E(h->feat, (size_t)B * d0.T * d0.KFP)); HIPCHK(h, E(h->wpad, (size_t)kD * d0.KFP)); HIPCHK(h, E(h->waT, (size_t)kAud * kD));
    HIPCHK(h, E(h->zc, (size_t)B * kSpk)); HIPCHK(h, E(h->dzc, (size_t)B * kSpk));
    for (Buf* b : {&h->mu, &h->lv, &h->pe_rows, &h->pre1, &h->hid, &h->emb, &h->demb, &h->dmu, &h->dlv, &h->dhid}) HIPCHK(h, E(*b, (size_t)B * kD));
    for (Buf* b : {&h->xcur, &h->G}) HIPCHK(h, E(*b, R * kD));
    for (Buf* b : {&h->X1, &h->A1, &h->X2, &h->A2, &h->dA2, &h->dA1}) HIPCHK(h, E(*b, (size_t)d0.L * R * kD));
    HIPCHK(h, E(h->colpart, (size_t)((B + 1) / 2) * d0.L * 5 * kD)); HIPCHK(h, E(h->dembp, (size_t)d0.L * B * kD));
    for (Buf* b : {&h->S1, &h->S2}) HIPCHK(h, E(*b, (size_t)d0.L * R * 2));
    HIPCHK(h, E(h->out, (size_t)B * d0.T * d0.JF)); HIPCHK(h, E(h->dout, (size_t)B * d0.T * d0.JF));
    HIPCHK(h, E(h->lossp, 2 * ((size_t)B * d0.JF / 256 + 2))); HIPCHK(h, E(h->kldp, B)); HIPCHK(h, E(h->terms, 8));
    HIPCHK(h, E(h->part, (size_t)kNW * 2 * kD + (size_t)B * 128 + 4096 * 512 + (size_t)B * 32 * 2 * 2 * ((L[1] + 5) / 6 / 64 + 1)));
    HIPCHK(h, E(h->pw, (size_t)d0.L * B * 4 * d0.S * d0.S)); HIPCHK(h, E(h->pb, (size_t)d0.L * B * 4 * d0.S));
    HIPCHK(h, E(h->dAf, (size_t)B * d0.T * kAud)); HIPCHK(h, E(h->dAt, (size_t)B * d0.T * kAud));
    // col: the partial-sum workspace of the implicit-GEMM weight gradients (one [Cout][Cin*15] image per workgroup run; the number of
    // runs does not grow with the batch) and of conv1's, and the InstanceNorm partials of the forward convs
    size_t colmax = 0;
    for (int i = 1; i < 4; ++i) {
        const size_t need = (size_t)conv_wgrad_groups(kCin[i], kCout[i]) * kCout[i] * kCin[i] * 15;
        if (need > colmax) colmax = need;
    }
    const size_t c1need = (size_t)B * (2 * (((L[1] + 5) / 6 + 63) / 64) * 512 + 128 + 480);     // conv1's S1 tile partials + row coefficients + per-sample gradients
    if (c1need > colmax) colmax = c1need;
    for (int i = 0; i < 3; ++i) {       // InstanceNorm partials of the forward convs
        const size_t need = (size_t)B * kCout[i] * ((L[i + 1] + 63) / 64) * 4 * 3;
        if (need > colmax) colmax = need;
    }
    HIPCHK(h, E(h->col, colmax));
    h->ws_floats = (size_t)48 << 20;
    HIPCHK(h, E(h->ws, h->ws_floats));
    if (h->fork) { HIPCHK(h, E(h->ws2, h->ws_floats)); HIPCHK(h, h->part2.ensure(h->part.bytes)); }
    h->capB = B;
    return LS_OK;
}

// layer l of a saved-activation family [L][R][512] / statistics [L][R][2]
static inline float* lay(const Buf& b, int l, int R) { return b.f() + (size_t)l * R * kD; }
static inline float* lay2(const Buf& b, int l, int R) { return b.f() + (size_t)l * R * 2; }

// locals shared by the stages of one step
#define TRAIN_LOCALS(h, d)                                                                                   \
    const int B = (d).B, S = (d).S, T = (d).T, R = B * S, BT = B * T, JF = (d).JF;                           \
    const int* L = (h)->convL;                                                                               \
    hipStream_t st = (h)->stream;                                                                            \
    float* part = (h)->part.f();                                                                             \
    (void)S; (void)T; (void)R; (void)BT; (void)JF; (void)L; (void)part

// column sums of a row-strided matrix into dst[cols] (bias gradients), deterministic two-stage reduction
static hipError_t colsum_to(ls_trainer* h, const float* in, int ri, long long ro, long long rs, int rows, int cols, float* dst) {
    int nblk = rows / 64;
    if (nblk < 1) nblk = 1;
    if (nblk > 512) nblk = 512;
    hipError_t e = launch_colsum(in, ri, ro, rs, rows, cols, h->part.f(), nblk, h->stream);
    if (e != hipSuccess) return e;
    return launch_partial_reduce(h->part.f(), nblk, cols, cols, dst, 0, h->stream);
}

// forward pass with every activation the backward needs kept in HBM; ends with the losses and d loss / d out
static int train_forward(ls_trainer* h, const TrainDims& d) {
    TRAIN_LOCALS(h, d);
    // WavEncoder (audio_enc.py:9-25): raw conv outputs + InstanceNorm statistics are kept for the backward
    // (the column buffer is free during the forward: it serves as the statistics-partials workspace)
    HIPCHK(h, launch_conv1_fwd(h->audio_in, P(h, ck(0, "weight")), P(h, ck(0, "bias")), h->c[0].f(), h->st[0].f(), h->col.f(), B, L[0], L[1],
                               kPad[0], st));
    {   // the stride-6 layers' operand images (forward / weight gradient and data gradient), rebuilt from the master weights: one launch
        const float* w3[3] = {P(h, ck(1, "weight")), P(h, ck(2, "weight")), P(h, ck(3, "weight"))};
        float* im[3] = {h->img[1].f(), h->img[2].f(), h->img[3].f()};
        float* dm[3] = {h->dimg[1].f(), h->dimg[2].f(), h->dimg[3].f()};
        HIPCHK(h, launch_build_conv_imgs(w3, im, dm, kCin + 1, kCout + 1, st));
    }
    for (int i = 1; i < 4; ++i) {
        HIPCHK(h, launch_conv1d_mfma(h->c[i - 1].f(), h->st[i - 1].f(), h->img[i].f(), P(h, ck(i, "bias")), h->c[i].f(), i < 3 ? h->st[i].f() : nullptr,
                                     h->col.f(), B, kCin[i], kCout[i], L[i], L[i + 1], st));
    }
    HIPCHK(h, launch_build_feat_train(h->in_x, h->in_noise, h->in_origin, h->c[3].f(), h->in_drop, h->hostpack.f(), h->hostpack.f() + B, h->feat.f(),
                                      h->x_t.f(), d, h->cfg.model.n_pre_seq, st));
    {   // input_mapping (RAG.py:114) -> frame rows of the token sequence
        // operands with 16-byte-aligned rows and a reduction length of whole K tiles (the master weight's rows are 311 floats: that product
        // ran the GEMM's generic path at 34 % matrix-pipe occupancy, 101 us; the audio-gradient product below likewise, 27 %)
        HIPCHK(h, launch_build_inmap_images(P(h, "input_mapping.weight"), h->wpad.f(), h->waT.f(), d.KF, d.KFP, 2 * d.JF + 1, st));
        GemmArgs a = gemm(op_rows(h->feat.f(), d.KFP, BT, d.KFP), op_rows(h->wpad.f(), d.KFP, kD, d.KFP),
                          h->xcur.f() + (size_t)d.NPRE * kD, kD, BT, kD, d.KFP);
        a.cri = T; a.cro = (long long)S * kD; a.crs = kD;
        a.bias = P(h, "input_mapping.bias");
        HIPCHK(h, gemm_run(h, a, true, true));
    }
    HIPCHK(h, launch_gather_rows(P(h, "speaker_embedding.weight"), h->in_vid, h->zc.f(), B, kSpk,
                                 h->cfg.model.n_speakers, st));
    for (int k = 0; k < 2; ++k) {
        GemmArgs a = gemm(op_rows(h->zc.f(), kSpk, B, kSpk), op_rows(P(h, k ? "speaker_logvar.weight" : "speaker_mu.weight"), kSpk, kD, kSpk),
                          k ? h->lv.f() : h->mu.f(), kD, B, kD, kSpk);
        a.bias = P(h, k ? "speaker_logvar.bias" : "speaker_mu.bias");
        HIPCHK(h, gemm_run(h, a, true, true));
    }
    HIPCHK(h, launch_style_fwd(h->mu.f(), h->lv.f(), h->in_eps, d.NPRE == 2 ? P(h, "emotion_embedding.weight") : nullptr,
                               h->in_emo, T, h->xcur.f(), h->kldp.f(), B, S, d.NPRE, st));
    // TimestepEmbedder (mlp_module.py:123-136)
    HIPCHK(h, launch_gather_rows(h->pe.f(), reinterpret_cast<const int64_t*>(h->hostpack.f() + 2 * B), h->pe_rows.f(), B, kD, kPeRows, st));
    {
        GemmArgs a = gemm(op_rows(h->pe_rows.f(), kD, B, kD), op_rows(P(h, "backbone.embed_timestep.time_embed.0.weight"), kD, kD, kD), h->hid.f(),
                          kD, B, kD, kD);
        a.bias = P(h, "backbone.embed_timestep.time_embed.0.bias"); a.Cpre = h->pre1.f(); a.act = 1;
        HIPCHK(h, gemm_run(h, a, true, true));
        GemmArgs b2 = gemm(op_rows(h->hid.f(), kD, B, kD), op_rows(P(h, "backbone.embed_timestep.time_embed.2.weight"), kD, kD, kD), h->emb.f(), kD,
                           B, kD, kD);
        b2.bias = P(h, "backbone.embed_timestep.time_embed.2.bias");
        HIPCHK(h, gemm_run(h, b2, true, true));
    }
    {   // TransMLP: all 8 MLPblocks (mlp_module.py:67-91) in ONE launch of the fused kernel the sampler uses (ls_step.hip,
        // TRAIN variant): a workgroup keeps two samples' residual streams in registers and writes x-hat1 / A1 / x-hat2 / A2
        // and the LayerNorm statistics of every layer for the backward; the last layer's output lands back in xcur.
        h->img_args.P = h->P.f();
        HIPCHK(h, launch_build_train_images(h->img_args, st));
        StepArgs a{};
        a.temb = h->emb.f(); a.temb_stride = kD;
        a.W = static_cast<const DevWeights*>(h->tdevw.p);
        a.layers = d.L;
        a.sampler = kNone;
        a.tr_x0 = h->xcur.f(); a.tr_xout = h->xcur.f(); a.tr_B = B;
        a.tr_x1 = h->X1.f(); a.tr_a1 = h->A1.f(); a.tr_x2 = h->X2.f(); a.tr_a2 = h->A2.f();
        a.tr_s1 = h->S1.f(); a.tr_s2 = h->S2.f();
        HIPCHK(h, launch_train_mixer_fwd(d.NPRE == 2 ? kBEAT : kTED, a, st));
    }
    {   // OutputProcess.poseFinal on the frame rows (RAG.py:128-129, 205-211)
        GemmArgs a = gemm(gemm_operand(h->xcur.f() + (size_t)d.NPRE * kD, T, (long long)S * kD, kD, INT_MAX, 0, 1, true, BT, kD),
                          op_rows(P(h, "output_process.poseFinal.weight"), kD, JF, kD), h->out.f(), JF, BT, JF, kD);
        a.bias = P(h, "output_process.poseFinal.bias");
        HIPCHK(h, gemm_run(h, a, true, true));
    }
    const int nlb = (B * JF + 255) / 256;
    HIPCHK(h, launch_loss(h->out.f(), h->in_x, h->dout.f(), h->lossp.f(), d, h->cfg.lambda_vel, st));
    HIPCHK(h, launch_finish_terms(h->lossp.f(), nlb, h->kldp.f(), B, h->terms.f(), d, h->cfg.lambda_vel, h->cfg.kld_weight, st));
    return LS_OK;
}

// poseFinal and the 8 MLPblocks; leaves d loss / d [style | (emotion) | input_mapping rows] in h->G and d emb in h->demb
static int train_backward_mixer(ls_trainer* h, const TrainDims& d, float* grad) {
    TRAIN_LOCALS(h, d);
    // poseFinal
    HIPCHK(h, wgrad(h, op_cols(h->dout.f(), JF, JF, BT), gemm_operand(h->xcur.f() + (size_t)d.NPRE * kD, INT_MAX, 0, 1, T, (long long)S * kD, kD, false, kD, BT),
                    false, false, Gr(h, grad, "output_process.poseFinal.weight"), kD, JF, kD, BT));
    HIPCHK(h, colsum_to(h, h->dout.f(), INT_MAX, 0, JF, BT, JF, Gr(h, grad, "output_process.poseFinal.bias")));
    HIPCHK(h, hipMemsetAsync(h->G.p, 0, (size_t)R * kD * 4, st));
    {
        GemmArgs a = gemm(op_rows(h->dout.f(), JF, BT, JF), op_cols(P(h, "output_process.poseFinal.weight"), kD, kD, JF),
                          h->G.f() + (size_t)d.NPRE * kD, kD, BT, kD, JF);
        a.cri = T; a.cro = (long long)S * kD; a.crs = kD;
        HIPCHK(h, gemm_run(h, a, true, false));
    }
    // All 8 MLPblocks backward in ONE launch (ls_train_bwd.hip): G stays in registers across the layers; the kernel leaves
    // dA2 / dA1 for the batch-level weight-gradient products below, per-workgroup partial column sums for the bias and
    // LayerNorm parameters, the per-layer d(timestep embedding) and, in G, the gradient of the layer-0 input.
    const int nwg = (B + 1) / 2;
    {
        MixerBwdArgs a{};
        a.g = h->G.f();
        a.a2 = h->A2.f(); a.a1 = h->A1.f(); a.x2 = h->X2.f(); a.x1 = h->X1.f(); a.s2 = h->S2.f(); a.s1 = h->S1.f();
        a.da2 = h->dA2.f(); a.da1 = h->dA1.f(); a.colpart = h->colpart.f(); a.dembp = h->dembp.f();
        a.wchT_img = h->twchT.f(); a.wwT_img = h->twwT.f(); a.ln2a = h->tl2a.f(); a.ln1a = h->tl1a.f();
        a.B = B; a.layers = d.L;
        HIPCHK(h, launch_mixer_bwd(d.NPRE == 2 ? kBEAT : kTED, a, st));
    }
    return LS_OK;
}

// parameter gradients of the 8 MLPblocks from what k_mixer_bwd left (dA2 / dA1, partial column sums, d emb partials); independent of the
// input stage and of the WavEncoder backward: runs on whichever stream h->stream is at the time (the side branch of the fork)
static int train_backward_mixer_params(ls_trainer* h, const TrainDims& d, float* grad) {
    TRAIN_LOCALS(h, d);
    const int nwg = (B + 1) / 2;
    // per-workgroup partial column sums -> bias / LayerNorm-parameter gradients of ALL layers, one launch per family
    // (alpha and beta are adjacent both in colpart and in the flat layout; the per-layer stride of the flat layout is constant)
    const long long cps = (long long)d.L * 5 * kD, ls = h->img_args.lstride;
    float* g0 = grad + h->img_args.base;
    HIPCHK(h, launch_partial_reduce_groups(h->colpart.f(), nwg, cps, kD, g0 + h->img_args.o_b, d.L, 5 * kD, ls, st));
    HIPCHK(h, launch_partial_reduce_groups(h->colpart.f() + kD, nwg, cps, 2 * kD, g0 + h->img_args.o_a2, d.L, 5 * kD, ls, st));
    HIPCHK(h, launch_partial_reduce_groups(h->colpart.f() + 3 * kD, nwg, cps, 2 * kD, g0 + h->img_args.o_a1, d.L, 5 * kD, ls, st));
    // token weights of all layers: dWt[s'][s] = sum_{b,c} dA1[b][s'][c] U1[b][s][c], d bt[s'] = sum_{b,c} dA1[b][s'][c]
    int ntw = 0;
    HIPCHK(h, launch_tokmix_wgrad(h->dA1.f(), h->X1.f(), h->tl1a.f(), h->tl1b.f(), h->pw.f(), h->pb.f(), B, S, d.L, &ntw, st));
    HIPCHK(h, launch_partial_reduce_groups(h->pw.f(), ntw, (long long)S * S, S * S, g0 + h->img_args.o_wt, d.L, (long long)ntw * S * S, ls, st));
    HIPCHK(h, launch_partial_reduce_groups(h->pb.f(), ntw, S, S, g0 + h->img_args.o_bt, d.L, (long long)ntw * S, ls, st));
    // channel-mix weight gradients of all layers: dW[l] = dA2[l]^T U2[l] with U2 = alpha2 * x-hat2 + beta2: the product runs on the saved
    // x-hat2 and the affine is applied to the 512 x 512 result (the bias gradients d b = colsum(dA2) were reduced above)
    HIPCHK(h, wgrad_batched(h, op_cols(lay(h->dA2, 0, R), kD, kD, R), op_cols(lay(h->X2, 0, R), kD, kD, R), g0 + h->img_args.o_w, kD, kD, kD, R, d.L,
                            (long long)R * kD, (long long)R * kD, ls));
    HIPCHK(h, launch_wch_affine(g0 + h->img_args.o_w, g0 + h->img_args.o_b, h->tl2a.f(), h->tl2b.f(), ls, d.L, st));
    HIPCHK(h, launch_partial_reduce(h->dembp.f(), d.L, (long long)B * kD, B * kD, h->demb.f(), 0, st));
    return LS_OK;
}

// style / speaker / emotion embeddings, input_mapping, timestep embedder; leaves d loss / d audio features in h->dAf
static int train_backward_inputs(ls_trainer* h, const TrainDims& d, float* grad) {
    TRAIN_LOCALS(h, d);
    // G = d loss / d [style | (emotion) | input_mapping rows]
    HIPCHK(h, launch_style_bwd(h->G.f(), h->mu.f(), h->lv.f(), h->in_eps, h->dmu.f(), h->dlv.f(), B, S, h->cfg.kld_weight, st));
    for (int k = 0; k < 2; ++k) {
        const float* dz = k ? h->dlv.f() : h->dmu.f();
        const char* wk = k ? "speaker_logvar.weight" : "speaker_mu.weight";
        HIPCHK(h, wgrad(h, op_cols(dz, kD, kD, B), op_cols(h->zc.f(), kSpk, kSpk, B), false, false, Gr(h, grad, wk), kSpk, kD, kSpk, B));
        HIPCHK(h, colsum_to(h, dz, INT_MAX, 0, kD, B, kD, Gr(h, grad, k ? "speaker_logvar.bias" : "speaker_mu.bias")));
        GemmArgs a = gemm(op_rows(dz, kD, B, kD), op_cols(P(h, wk), kSpk, kSpk, kD), h->dzc.f(), kSpk, B, kSpk, kD);
        a.accumulate = k;
        HIPCHK(h, gemm_run(h, a, true, false));
    }
    HIPCHK(h, launch_scatter_rows(h->dzc.f(), kSpk, h->in_vid, 1, B, kSpk, Gr(h, grad, "speaker_embedding.weight"), st));
    if (d.NPRE == 2)
        HIPCHK(h, launch_scatter_rows(h->G.f() + kD, (long long)S * kD, h->in_emo, T, B, kD,
                                      Gr(h, grad, "emotion_embedding.weight"), st));
    // input_mapping
    const float* dH = h->G.f() + (size_t)d.NPRE * kD;
    HIPCHK(h, wgrad(h, gemm_operand(dH, INT_MAX, 0, 1, T, (long long)S * kD, kD, false, kD, BT), op_cols(h->feat.f(), d.KFP, d.KF, BT), false, false,
                    Gr(h, grad, "input_mapping.weight"), d.KF, kD, d.KF, BT));
    HIPCHK(h, colsum_to(h, dH, T, (long long)S * kD, kD, BT, kD, Gr(h, grad, "input_mapping.bias")));
    {   // d audio features = dH . Win[:, 2JF+1:], then the mask_cond scale
        GemmArgs a = gemm(gemm_operand(dH, T, (long long)S * kD, kD, INT_MAX, 0, 1, true, BT, kD),
                          op_rows(h->waT.f(), kD, kAud, kD), h->dAf.f(), kAud, BT, kAud, kD);
        HIPCHK(h, gemm_run(h, a, true, true));
        HIPCHK(h, launch_scale_rows(h->dAf.f(), h->in_drop, B, T * kAud, st));
    }
    return LS_OK;
}

// TimestepEmbedder backward from d emb (reduced by train_backward_mixer_params: same stream, behind it)
static int train_backward_temb(ls_trainer* h, const TrainDims& d, float* grad) {
    TRAIN_LOCALS(h, d);
    // TimestepEmbedder
    {
        const char* w0 = "backbone.embed_timestep.time_embed.0.weight";
        const char* w2 = "backbone.embed_timestep.time_embed.2.weight";
        HIPCHK(h, wgrad(h, op_cols(h->demb.f(), kD, kD, B), op_cols(h->hid.f(), kD, kD, B), false, false, Gr(h, grad, w2), kD, kD, kD, B));
        HIPCHK(h, colsum_to(h, h->demb.f(), INT_MAX, 0, kD, B, kD, Gr(h, grad, "backbone.embed_timestep.time_embed.2.bias")));
        GemmArgs a = gemm(op_rows(h->demb.f(), kD, B, kD), op_cols(P(h, w2), kD, kD, kD), h->dhid.f(), kD, B, kD, kD);
        HIPCHK(h, gemm_run(h, a, true, false));
        HIPCHK(h, launch_silu_bwd_colsum(h->dhid.f(), h->pre1.f(), h->dhid.f(), part, B, kNW, st));
        HIPCHK(h, launch_partial_reduce(part, kNW, kD, kD, Gr(h, grad, "backbone.embed_timestep.time_embed.0.bias"), 0, st));
        HIPCHK(h, wgrad(h, op_cols(h->dhid.f(), kD, kD, B), op_cols(h->pe_rows.f(), kD, kD, B), false, false, Gr(h, grad, w0), kD, kD, kD, B));
    }
    return LS_OK;
}

// WavEncoder backward, last layer first
static int train_backward_audio(ls_trainer* h, const TrainDims& d, float* grad) {
    TRAIN_LOCALS(h, d);
    // WavEncoder backward, last layer first.  dC4(b, co, p) = dAf[(b*T + p)][co]
    {
        const int W4 = kCin[3] * 15;
        // dC4(b, co, p) = dAf[(b*T + p)][co]: transposed once to [b][co][p] so that both gradients stage position-contiguous rows
        // (rounds 1-2 read it in place: a 1 KB stride between the lanes of every load)
        HIPCHK(h, launch_transpose_rc(h->dAf.f(), h->dAt.f(), B, T, kAud, st));
        int ng = 0;                       // implicit GEMM like conv2 / conv3 (rounds 1-2: im2col + GEMM, 105 + 188 us at B = 512)
        HIPCHK(h, launch_conv_wgrad(h->dAt.f(), (long long)T * kAud, T, h->c[2].f(), h->st[2].f(), h->col.f(), B, kCin[3], kCout[3], L[3], L[4], &ng, st));
        HIPCHK(h, launch_partial_reduce(h->col.f(), ng, (long long)kCout[3] * W4, kCout[3] * W4, Gr(h, grad, ck(3, "weight")), 0, st));
        HIPCHK(h, colsum_to(h, h->dAf.f(), INT_MAX, 0, kAud, BT, kAud, Gr(h, grad, ck(3, "bias"))));
        // data gradient: implicit GEMM + LeakyReLU' + InstanceNorm backward
        HIPCHK(h, launch_conv_dgrad(h->dAt.f(), (long long)T * kAud, T, 1, h->dimg[3].f(), h->c[2].f(), h->st[2].f(), h->dc[2].f(), part, B,
                                    kCin[3], kCout[3], L[3], L[4], true, nullptr, st));
    }
    for (int i = 2; i >= 1; --i) {      // conv3 (i=2), conv2 (i=1): dC_i = dc[i] [B][Cout_i][L_{i+1}]
        const int C = kCout[i], Lo = L[i + 1], W = kCin[i] * 15;
        {   // weight gradient: implicit GEMM straight from the raw conv output of the layer below (no im2col)
            int ng = 0;
            HIPCHK(h, launch_conv_wgrad(h->dc[i].f(), (long long)C * Lo, Lo, h->c[i - 1].f(), h->st[i - 1].f(), h->col.f(), B, kCin[i], C, L[i], Lo, &ng,
                                        st));
            HIPCHK(h, launch_partial_reduce(h->col.f(), ng, (long long)C * W, C * W, Gr(h, grad, ck(i, "weight")), 0, st));
        }
        // (bias gradients of conv1..3 stay exactly 0: a bias that feeds an InstanceNorm cannot change the output; the
        //  reference's autograd returns rounding noise of ~1e-7 there)
        if (i == 2) {
            HIPCHK(h, launch_conv_dgrad(h->dc[i].f(), (long long)C * Lo, Lo, 1, h->dimg[i].f(), h->c[i - 1].f(), h->st[i - 1].f(), h->dc[i - 1].f(), part, B,
                                        kCin[i], C, L[i], Lo, true, nullptr, st));
        } else {
            // conv2's data gradient is consumed only by conv1's weight gradient (conv1's input is data): folded into its epilogue, the
            // gradient tensor itself is never written (k_conv_dgrad<FUSE1>); per-sample results are summed over the batch in index order
            float* outp = nullptr;
            // (on a side stream under the forward's first convs this latency-bound 50 us kernel cost the forward 200 us: measured, not kept)
            HIPCHK(h, launch_wav_moments(h->audio_in, h->wmom.f(), B, L[0], L[1], kPad[0], st));
            HIPCHK(h, launch_conv_dgrad_conv1(h->dc[i].f(), (long long)C * Lo, Lo, 1, h->dimg[i].f(), h->c[0].f(), h->st[0].f(), part, B, C, L[1], Lo,
                                              h->audio_in, L[0], kPad[0], h->wmom.f(), P(h, ck(0, "weight")), P(h, ck(0, "bias")), h->col.f(), &outp, st));
            HIPCHK(h, launch_partial_reduce(outp, B, 480, 480, Gr(h, grad, ck(0, "weight")), 0, st));
        }
    }
    return LS_OK;
}

}  // namespace

extern "C" {

int ls_train_create(const ls_train_config* cfg, ls_trainer** out) {
    if (!cfg || !out) return fail(nullptr, LS_EINVAL, "ls_train_create: null argument");
    const ls_config& m = cfg->model;
    if (m.latent_dim != kD || m.nframes != 34 || m.layers < 1 || m.layers > 16 || (m.n_prefix_tokens != 1 && m.n_prefix_tokens != 2) ||
        (m.n_prefix_tokens == 2 && m.n_emotions <= 0) || m.njoints <= 0 || m.nfeats <= 0 || m.n_speakers <= 0 || cfg->diffusion_steps <= 0)
        return fail(nullptr, LS_EUNSUPPORTED, "ls_train_create: unsupported configuration");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, LS_EHIP, "ls_train_create: no HIP device (the training step has no CPU path)");
    if (m.device < 0 || m.device >= n) return fail(nullptr, LS_EINVAL, "ls_train_create: device %d out of range", m.device);
    ls_trainer* h = new ls_trainer();
    h->cfg = *cfg;
    TrainDims& d = h->d;
    d.B = 0; d.T = m.nframes; d.NPRE = m.n_prefix_tokens; d.S = d.T + d.NPRE; d.JF = m.njoints * m.nfeats;
    d.KF = 2 * d.JF + 1 + kAud; d.KFP = (d.KF + 31) / 32 * 32; d.D = kD; d.L = m.layers;      // KFP: whole 32-deep K tiles (the input_mapping product takes the GEMM's LDS-DMA path)
    h->convL[0] = m.audio_len;
    for (int i = 0; i < 4; ++i) h->convL[i + 1] = (h->convL[i] + 2 * kPad[i] - 15) / kStride[i] + 1;
    if (h->convL[4] != d.T) { delete h; return fail(nullptr, LS_EINVAL, "ls_train_create: audio_len %d gives %d audio frames, need %d", m.audio_len, h->convL[4], d.T); }
    for (int l = 0; l < d.L; ++l) {
        add_param(h, lk(l, "block1.0.alpha"), kD); add_param(h, lk(l, "block1.0.beta"), kD);
        add_param(h, lk(l, "block1.1.weight"), (int64_t)d.S * d.S); add_param(h, lk(l, "block1.1.bias"), d.S);
        add_param(h, lk(l, "block2.0.alpha"), kD); add_param(h, lk(l, "block2.0.beta"), kD);
        add_param(h, lk(l, "block2.1.weight"), (int64_t)kD * kD); add_param(h, lk(l, "block2.1.bias"), kD);
    }
    for (int j : {0, 2}) {
        add_param(h, "backbone.embed_timestep.time_embed." + std::to_string(j) + ".weight", (int64_t)kD * kD);
        add_param(h, "backbone.embed_timestep.time_embed." + std::to_string(j) + ".bias", kD);
    }
    add_param(h, "input_mapping.weight", (int64_t)kD * d.KF); add_param(h, "input_mapping.bias", kD);
    add_param(h, "speaker_embedding.weight", (int64_t)m.n_speakers * kSpk);
    add_param(h, "speaker_mu.weight", (int64_t)kD * kSpk); add_param(h, "speaker_mu.bias", kD);
    add_param(h, "speaker_logvar.weight", (int64_t)kD * kSpk); add_param(h, "speaker_logvar.bias", kD);
    for (int i = 0; i < 4; ++i) { add_param(h, ck(i, "weight"), (int64_t)kCout[i] * kCin[i] * 15); add_param(h, ck(i, "bias"), kCout[i]); }
    add_param(h, "output_process.poseFinal.weight", (int64_t)d.JF * kD); add_param(h, "output_process.poseFinal.bias", d.JF);
    if (d.NPRE == 2) add_param(h, "emotion_embedding.weight", (int64_t)m.n_emotions * kD);

    auto bail = [&](const char* what, hipError_t e) {
        fail(nullptr, LS_EHIP, "ls_train_create: %s: %s", what, hipGetErrorString(e));
        ls_train_destroy(h);
        return LS_EHIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(m.device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (auto& ev : h->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    {   // the side branch at the HIGHEST stream priority: a stream of another priority gets a hardware queue of its own (streams of one
        // priority share a small pool of queues, and two that land on the same queue run their kernels in order: in a process with a
        // dozen other streams -- bench.py -- the fork bought nothing until this), and its short kernels go ahead of the long conv launches
        int lo = 0, hi = 0;
        if ((e = hipDeviceGetStreamPriorityRange(&lo, &hi)) != hipSuccess) return bail("hipDeviceGetStreamPriorityRange", e);
        if ((e = hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi)) != hipSuccess) return bail("hipStreamCreateWithPriority", e);
    }
    for (auto& ev : h->evs) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    for (Buf* b : {&h->P, &h->M, &h->V}) {
        if ((e = b->ensure((size_t)h->flat * 4)) != hipSuccess) return bail("hipMalloc(params)", e);
        if ((e = hipMemsetAsync(b->p, 0, (size_t)h->flat * 4, h->stream)) != hipSuccess) return bail("hipMemset", e);
    }
    // PositionalEncoding.pe (mlp_module.py:104-116), a buffer, recomputed
    std::vector<float> pe((size_t)kPeRows * kD);
    for (int p = 0; p < kPeRows; ++p)
        for (int i = 0; i < kD / 2; ++i) {
            const float div = expf((float)(2 * i) * (-logf(10000.0f) / (float)kD));
            pe[(size_t)p * kD + 2 * i] = sinf((float)p * div);
            pe[(size_t)p * kD + 2 * i + 1] = cosf((float)p * div);
        }
    if ((e = h->pe.ensure(pe.size() * 4)) != hipSuccess) return bail("hipMalloc(pe)", e);
    if ((e = hipMemcpy(h->pe.p, pe.data(), pe.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(pe)", e);
    for (int i = 1; i < 4; ++i) {
        if ((e = h->img[i].ensure((size_t)kCout[i] * kCin[i] * 15 * 4)) != hipSuccess) return bail("hipMalloc(img)", e);
        if ((e = h->dimg[i].ensure((size_t)kCout[i] * kCin[i] * 16 * 4)) != hipSuccess) return bail("hipMalloc(dimg)", e);
    }
    {   // mixer weight images of the fused training forward (rebuilt from the master parameters every step)
        const int MK = (2 * d.S + 3) / 4;
        const size_t nimg[8] = {(size_t)d.L * kD * kD, (size_t)d.L * kD, (size_t)d.L * 5 * MK * 64, (size_t)d.L * 80,
                                (size_t)d.L * kD, (size_t)d.L * kD, (size_t)d.L * kD, (size_t)d.L * kD};
        Buf* ib[8] = {&h->twch, &h->tbch, &h->tww, &h->tbtok, &h->tl1a, &h->tl1b, &h->tl2a, &h->tl2b};
        for (int i = 0; i < 8; ++i)
            if ((e = ib[i]->ensure(nimg[i] * 4)) != hipSuccess) return bail("hipMalloc(train images)", e);
        DevWeights dw{};
        dw.wch_img = h->twch.f(); dw.bch = h->tbch.f(); dw.ww_img = h->tww.f(); dw.btok_rows = h->tbtok.f();
        dw.ln1a = h->tl1a.f(); dw.ln1b = h->tl1b.f(); dw.ln2a = h->tl2a.f(); dw.ln2b = h->tl2b.f();
        if ((e = h->tdevw.ensure(sizeof dw)) != hipSuccess) return bail("hipMalloc(DevWeights)", e);
        if ((e = hipMemcpy(h->tdevw.p, &dw, sizeof dw, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(DevWeights)", e);
        auto off = [&](int l, const char* sfx) { return h->table[h->index.at(lk(l, sfx))].off; };
        TrainImgArgs& ia = h->img_args;
        ia.base = off(0, "block1.0.alpha");
        ia.lstride = d.L > 1 ? off(1, "block1.0.alpha") - ia.base : 0;
        ia.o_a1 = 0; ia.o_b1 = off(0, "block1.0.beta") - ia.base; ia.o_wt = off(0, "block1.1.weight") - ia.base;
        ia.o_bt = off(0, "block1.1.bias") - ia.base; ia.o_a2 = off(0, "block2.0.alpha") - ia.base; ia.o_b2 = off(0, "block2.0.beta") - ia.base;
        ia.o_w = off(0, "block2.1.weight") - ia.base; ia.o_b = off(0, "block2.1.bias") - ia.base;
        ia.L = d.L; ia.S = d.S; ia.MK = MK;
        ia.wch = h->twch.f(); ia.bch = h->tbch.f(); ia.ww = h->tww.f(); ia.btok = h->tbtok.f();
        ia.l1a = h->tl1a.f(); ia.l1b = h->tl1b.f(); ia.l2a = h->tl2a.f(); ia.l2b = h->tl2b.f();
        if ((e = h->twchT.ensure(nimg[0] * 4)) != hipSuccess || (e = h->twwT.ensure(nimg[2] * 4)) != hipSuccess) return bail("hipMalloc(transposed images)", e);
        ia.wchT = h->twchT.f(); ia.wwT = h->twwT.f();
        if ((e = init_step_kernels()) != hipSuccess || (e = init_mixer_bwd()) != hipSuccess) return bail("hipFuncSetAttribute", e);
    }
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail("sync", e);
    *out = h;
    return LS_OK;
}

void ls_train_destroy(ls_trainer* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.model.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    std::vector<Buf*> all = {&h->P, &h->M, &h->V, &h->pe, &h->x_start, &h->noise, &h->drop, &h->eps, &h->audio, &h->origin_x, &h->vid, &h->emo, &h->hostpack, &h->wpad, &h->waT,
                             &h->ca, &h->cb, &h->tidx, &h->feat, &h->x_t, &h->zc, &h->mu, &h->lv, &h->pe_rows, &h->pre1, &h->hid, &h->emb, &h->xcur,
                             &h->out, &h->dout, &h->lossp, &h->kldp, &h->terms, &h->G, &h->part, &h->pw, &h->pb, &h->demb, &h->dmu,
                             &h->dlv, &h->dzc, &h->dhid, &h->dAf, &h->dAt, &h->col, &h->wmom, &h->ws, &h->part2, &h->ws2};
    for (int i = 0; i < 4; ++i) { all.push_back(&h->c[i]); all.push_back(&h->img[i]); all.push_back(&h->dimg[i]); }
    for (int i = 0; i < 3; ++i) { all.push_back(&h->st[i]); all.push_back(&h->dc[i]); }
    for (Buf* b : {&h->X1, &h->A1, &h->X2, &h->A2, &h->S1, &h->S2, &h->twch, &h->tbch, &h->tww, &h->tbtok, &h->tl1a, &h->tl1b, &h->tl2a,
                   &h->tl2b, &h->tdevw, &h->twchT, &h->twwT, &h->dA2, &h->dA1, &h->colpart, &h->dembp})
        all.push_back(b);
    for (Buf* b : all) b->release();
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : h->evs) if (ev) (void)hipEventDestroy(ev);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* ls_train_last_error(const ls_trainer* h) { return h ? h->err.c_str() : g_train_create_error.c_str(); }

int ls_train_set_schedule(ls_trainer* h, const double* sac, const double* s1mac, const int64_t* tmap) {
    if (!h || !sac || !s1mac || !tmap) return fail(h, LS_EINVAL, "ls_train_set_schedule: null argument");
    const int n = h->cfg.diffusion_steps;
    h->sac.assign(sac, sac + n); h->s1mac.assign(s1mac, s1mac + n); h->tmap.assign(tmap, tmap + n);
    for (int i = 0; i < n; ++i)
        if (tmap[i] < 0 || tmap[i] >= kPeRows) return fail(h, LS_EINVAL, "ls_train_set_schedule: timestep_map[%d] = %lld out of range", i, (long long)tmap[i]);
    h->have_sched = true;
    return LS_OK;
}

int ls_train_param_count(const ls_trainer* h) { return h ? (int)h->table.size() : 0; }
int64_t ls_train_flat_size(const ls_trainer* h) { return h ? h->flat : 0; }

int ls_train_param_info(const ls_trainer* h, int index, char* key, size_t key_cap, int64_t* offset, int64_t* numel) {
    if (!h || index < 0 || index >= (int)h->table.size()) return LS_EINVAL;
    const Param& p = h->table[index];
    if (key && key_cap) { strncpy(key, p.key.c_str(), key_cap - 1); key[key_cap - 1] = 0; }
    if (offset) *offset = p.off;
    if (numel) *numel = p.n;
    return LS_OK;
}

int ls_train_set_weight(ls_trainer* h, const char* key, const float* data, size_t n) {
    if (!h || !key || !data) return fail(h, LS_EINVAL, "ls_train_set_weight: null argument");
    std::string k(key);
    if (k.size() >= 3 && k.compare(k.size() - 3, 3, ".pe") == 0) return LS_OK;
    auto it = h->index.find(k);
    if (it == h->index.end()) return fail(h, LS_EINVAL, "ls_train_set_weight: unexpected key '%s'", key);
    const Param& p = h->table[it->second];
    if ((int64_t)n != p.n) return fail(h, LS_EINVAL, "ls_train_set_weight: '%s' has %zu elements, expected %lld", key, n, (long long)p.n);
    HIPCHK(h, hipSetDevice(h->cfg.model.device));
    HIPCHK(h, hipMemcpy(h->P.f() + p.off, data, n * 4, hipMemcpyHostToDevice));
    return LS_OK;
}

int ls_train_get_weight(ls_trainer* h, const char* key, float* out, size_t n) {
    if (!h || !key || !out) return fail(h, LS_EINVAL, "ls_train_get_weight: null argument");
    auto it = h->index.find(key);
    if (it == h->index.end()) return fail(h, LS_EINVAL, "ls_train_get_weight: unknown key '%s'", key);
    const Param& p = h->table[it->second];
    if ((int64_t)n != p.n) return fail(h, LS_EINVAL, "ls_train_get_weight: '%s' has %lld elements, got room for %zu", key, (long long)p.n, n);
    HIPCHK(h, hipSetDevice(h->cfg.model.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->P.f() + p.off, n * 4, hipMemcpyDeviceToHost));
    return LS_OK;
}

static int moment_io(ls_trainer* h, int which, const char* key, float* out, const float* in, size_t n) {
    if (!h || !key || (!out && !in)) return fail(h, LS_EINVAL, "ls_train_*_moment: null argument");
    if (which != 1 && which != 2) return fail(h, LS_EINVAL, "ls_train_*_moment: which must be 1 (exp_avg) or 2 (exp_avg_sq)");
    auto it = h->index.find(key);
    if (it == h->index.end()) return fail(h, LS_EINVAL, "ls_train_*_moment: unknown key '%s'", key);
    const Param& p = h->table[it->second];
    if ((int64_t)n != p.n) return fail(h, LS_EINVAL, "ls_train_*_moment: '%s' has %lld elements, got %zu", key, (long long)p.n, n);
    HIPCHK(h, hipSetDevice(h->cfg.model.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float* dev = (which == 1 ? h->M.f() : h->V.f()) + p.off;
    if (out) HIPCHK(h, hipMemcpy(out, dev, n * 4, hipMemcpyDeviceToHost));
    else HIPCHK(h, hipMemcpy(dev, in, n * 4, hipMemcpyHostToDevice));
    return LS_OK;
}
int ls_train_get_moment(ls_trainer* h, int which, const char* key, float* out, size_t n) { return moment_io(h, which, key, out, nullptr, n); }
int ls_train_set_moment(ls_trainer* h, int which, const char* key, const float* data, size_t n) { return moment_io(h, which, key, nullptr, data, n); }
int64_t ls_train_get_step(const ls_trainer* h) { return h ? h->adam_step : 0; }
int ls_train_set_step(ls_trainer* h, int64_t step) {
    if (!h || step < 0) return fail(h, LS_EINVAL, "ls_train_set_step: bad argument");
    h->adam_step = step;
    return LS_OK;
}

int ls_train_forward_backward(ls_trainer* h, const ls_train_batch* tb, float* grad, ls_train_terms* terms) {
    if (!h || !tb || !grad) return fail(h, LS_EINVAL, "ls_train_forward_backward: null argument");
    if (!h->have_sched) return fail(h, LS_ESTATE, "ls_train_forward_backward: ls_train_set_schedule has not been called");
    const int B = tb->batch;
    if (B <= 0) return fail(h, LS_EINVAL, "ls_train_forward_backward: batch must be positive");
    if (!tb->x_start || !tb->t || !tb->noise || !tb->drop || !tb->eps || !tb->audio_input || !tb->origin_x || !tb->vid_indices)
        return fail(h, LS_EINVAL, "ls_train_forward_backward: a required tensor is null");
    TrainDims d = h->d;
    d.B = B;
    if (d.NPRE == 2 && !tb->emo) return fail(h, LS_EINVAL, "ls_train_forward_backward: emo is required for the 2-prefix-token model");
    HIPCHK(h, hipSetDevice(h->cfg.model.device));
    int rc = ensure_batch(h, B);
    if (rc != LS_OK) return rc;
    hipStream_t st = h->stream;
    const bool od = tb->on_device != 0;
    const size_t nx = (size_t)B * d.JF * d.T * 4;
    const int* L = h->convL;
    const int T = d.T;

    // ---- host side of q_sample / timestep lookup ----
    // [ca (B floats) | cb (B floats) | timestep-table rows (B int64)] in one staging vector and one upload
    std::vector<float> pack((size_t)4 * B);
    int64_t* tm = reinterpret_cast<int64_t*>(pack.data() + 2 * B);
    for (int b = 0; b < B; ++b) {
        const int64_t t = tb->t[b];
        if (t < 0 || t >= h->cfg.diffusion_steps) return fail(h, LS_EINVAL, "ls_train_forward_backward: t[%d] = %lld out of range", b, (long long)t);
        pack[b] = (float)h->sac[t]; pack[B + b] = (float)h->s1mac[t]; tm[b] = h->tmap[t];
    }
    if ((rc = ingest(h, h->hostpack, pack.data(), (size_t)16 * B, false))) return rc;
    // device-resident batch tensors are read in place (rounds 1-2 copied each into the handle: nine 4 us copies in front of every step)
    auto in_place = [&](Buf& own, const void* src, size_t bytes, const void** out) -> int {
        if (od) { *out = src; return LS_OK; }
        int r = ingest(h, own, src, bytes, false);
        *out = own.p;
        return r;
    };
    if ((rc = in_place(h->x_start, tb->x_start, nx, (const void**)&h->in_x)) || (rc = in_place(h->noise, tb->noise, nx, (const void**)&h->in_noise)) ||
        (rc = in_place(h->origin_x, tb->origin_x, nx, (const void**)&h->in_origin)) || (rc = in_place(h->drop, tb->drop, B * 4, (const void**)&h->in_drop)) ||
        (rc = in_place(h->eps, tb->eps, (size_t)B * kD * 4, (const void**)&h->in_eps)) ||
        (rc = in_place(h->audio, tb->audio_input, (size_t)B * L[0] * 4, (const void**)&h->audio_in)) ||
        (rc = in_place(h->vid, tb->vid_indices, B * 8, (const void**)&h->in_vid)))
        return rc;
    h->in_emo = nullptr;
    if (d.NPRE == 2 && (rc = in_place(h->emo, tb->emo, (size_t)B * T * 8, (const void**)&h->in_emo))) return rc;
    HIPCHK(h, hipStreamSynchronize(st));      // host staging vectors go out of scope below
    HIPCHK(h, hipEventRecord(h->ev[0], st));
    HIPCHK(h, hipMemsetAsync(grad, 0, (size_t)h->flat * 4, st));

    if ((rc = train_forward(h, d)) != LS_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev[1], st));
    if ((rc = train_backward_mixer(h, d, grad)) != LS_OK) return rc;
    if (h->fork) {
        HIPCHK(h, hipEventRecord(h->evs[0], st));
        HIPCHK(h, hipStreamWaitEvent(h->side, h->evs[0], 0));
        {   // the helpers launch on h->stream with h->ws / h->part: the side branch borrows the names for its launches
            std::swap(h->stream, h->side); std::swap(h->ws, h->ws2); std::swap(h->part, h->part2);
            rc = train_backward_mixer_params(h, d, grad);
            if (rc == LS_OK) rc = train_backward_temb(h, d, grad);
            std::swap(h->stream, h->side); std::swap(h->ws, h->ws2); std::swap(h->part, h->part2);
            if (rc != LS_OK) return rc;
        }
        HIPCHK(h, hipEventRecord(h->evs[1], h->side));
        if ((rc = train_backward_inputs(h, d, grad)) != LS_OK || (rc = train_backward_audio(h, d, grad)) != LS_OK) return rc;
        HIPCHK(h, hipStreamWaitEvent(st, h->evs[1], 0));
    } else if ((rc = train_backward_mixer_params(h, d, grad)) != LS_OK || (rc = train_backward_inputs(h, d, grad)) != LS_OK ||
               (rc = train_backward_temb(h, d, grad)) != LS_OK || (rc = train_backward_audio(h, d, grad)) != LS_OK)
        return rc;
    HIPCHK(h, hipEventRecord(h->ev[2], st));
    float tv[8] = {0};
    HIPCHK(h, hipMemcpyAsync(tv, h->terms.p, 5 * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    h->B = B;
    h->have_forward = true;
    if (terms) {
        terms->rot_mse = tv[0]; terms->vel_mse = tv[1]; terms->kld = tv[2]; terms->loss = tv[3]; terms->total = tv[4];
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]); terms->fwd_ms = ms;
        (void)hipEventElapsedTime(&ms, h->ev[1], h->ev[2]); terms->bwd_ms = ms;
        terms->reserved = 0.f;
    }
    return LS_OK;
}

int ls_train_adamw(ls_trainer* h, const float* grad, float lr, float beta1, float beta2, float eps, float weight_decay) {
    if (!h || !grad) return fail(h, LS_EINVAL, "ls_train_adamw: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.model.device));
    h->adam_step += 1;
    const float bc1 = 1.0f - powf(beta1, (float)h->adam_step), bc2 = 1.0f - powf(beta2, (float)h->adam_step);
    HIPCHK(h, launch_adamw(h->P.f(), grad, h->M.f(), h->V.f(), (size_t)h->flat, lr, beta1, beta2, eps, weight_decay, bc1, bc2, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LS_OK;
}

int ls_train_read(ls_trainer* h, const char* what, float* out, size_t n) {
    if (!h || !what || !out) return fail(h, LS_EINVAL, "ls_train_read: null argument");
    if (!h->have_forward) return fail(h, LS_ESTATE, "ls_train_read: no forward has run");
    const TrainDims& d = h->d;
    const size_t B = (size_t)h->B;
    const std::string w(what);
    const float* src = nullptr;
    size_t need = 0;
    if (w == "out") { src = h->out.f(); need = B * d.T * d.JF; }
    else if (w == "x_t") { src = h->x_t.f(); need = B * d.JF * d.T; }
    else if (w == "audio_feat") { src = h->c[3].f(); need = B * kAud * d.T; }
    else if (w == "z_mu") { src = h->mu.f(); need = B * kD; }
    else if (w == "z_logvar") { src = h->lv.f(); need = B * kD; }
    else if (w == "emb") { src = h->emb.f(); need = B * kD; }
    else if (w == "x_last") { src = h->xcur.f(); need = B * d.S * kD; }
    else if (w.size() == 2 && w[0] == 'c' && w[1] >= '1' && w[1] <= '4') {          // raw conv outputs [B][Cout][L]
        const int i = w[1] - '1';
        src = h->c[i].f(); need = B * kCout[i] * h->convL[i + 1];
    } else if (w.size() == 3 && w[0] == 's' && w[1] == 't' && w[2] >= '1' && w[2] <= '3') {   // (mean, rstd) per (sample, channel)
        const int i = w[2] - '1';
        src = h->st[i].f(); need = B * kCout[i] * 2;
    }
    else return fail(h, LS_EINVAL, "ls_train_read: unknown tensor '%s'", what);
    if (n != need) return fail(h, LS_EINVAL, "ls_train_read: '%s' has %zu elements, buffer has %zu", what, need, n);
    HIPCHK(h, hipSetDevice(h->cfg.model.device));
    HIPCHK(h, hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost));
    return LS_OK;
}

void* ls_train_stream(const ls_trainer* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

}  // extern "C"
