// C-ABI of the SAG decoder (include/ls_hip.h, "ls_sag_*"): replaces SAG.decoder(batch) of
// scripts/test_LivelySpeaker_ted.py:88 = Decoder_TRANSFORMER.forward (scripts/model/motionclip_module.py:138-183).
#include "ls_hip.h"
#include "ls_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace ls;

namespace {
std::string g_sag_create_error;
struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    float* f() const { return static_cast<float*>(p); }
};
}  // namespace

struct ls_sag {
    ls_sag_config cfg{};
    int JF = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, std::vector<float>> w;
    std::map<std::string, Buf> dw;      // device copies under the same keys
    bool committed = false;
    Buf pe, xin, zin, mask, q, qc, qkv, attn, t1, ca, x2, hid, t3, out;
    Buf wcross, bcross;      // cross-attention of ALL layers as one [L*D][D] matrix (see ls_sag_commit_weights)
    hipEvent_t ev[2] = {nullptr, nullptr};
    float last_ms = 0.f;
    bool pending_ms = false;   // ls_sag_decode_async enqueued: last_ms is read from the events when asked for
};

namespace {
int sfail(ls_sag* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_sag_create_error = buf;
    return code;
}
#define SCHK(h, expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess)                                                                                 \
            return sfail((h), LS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
}  // namespace

extern "C" {

const char* ls_sag_last_error(const ls_sag* h) { return h ? h->err.c_str() : g_sag_create_error.c_str(); }

int ls_sag_create(const ls_sag_config* cfg, ls_sag** out) {
    if (!cfg || !out) return sfail(nullptr, LS_EINVAL, "ls_sag_create: null argument");
    *out = nullptr;
    if (cfg->latent_dim != kD) return sfail(nullptr, LS_EUNSUPPORTED, "latent_dim must be %d", kD);
    if (cfg->nframes != kT) return sfail(nullptr, LS_EUNSUPPORTED, "nframes must be %d", kT);
    if (cfg->num_heads < 1 || cfg->latent_dim / cfg->num_heads != 128)
        return sfail(nullptr, LS_EUNSUPPORTED, "head dim must be 128 (latent 512, 4 heads)");
    if (cfg->num_layers < 1 || cfg->ff_size < 1 || cfg->njoints < 1 || cfg->nfeats < 1 || cfg->n_pre_poses < 0 || cfg->n_pre_poses > kT)
        return sfail(nullptr, LS_EINVAL, "bad SAG config");
    hipError_t e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return sfail(nullptr, LS_EHIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(e));
    ls_sag* h = new ls_sag();
    h->cfg = *cfg;
    h->JF = cfg->njoints * cfg->nfeats;
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return sfail(nullptr, LS_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    for (auto& ev : h->ev)
        if (hipEventCreate(&ev) != hipSuccess) { delete h; return sfail(nullptr, LS_EHIP, "hipEventCreate failed"); }
    // PositionalEncoding rows 0..T-1 (motionclip_module.py:11-28), fp32 like the torch buffer
    std::vector<float> pe((size_t)kT * kD);
    const float cexp = (float)(-std::log(10000.0) / kD);
    for (int i = 0; i < kD / 2; ++i) {
        const float div = expf((float)(2 * i) * cexp);
        for (int p = 0; p < kT; ++p) {
            pe[(size_t)p * kD + 2 * i] = sinf((float)p * div);
            pe[(size_t)p * kD + 2 * i + 1] = cosf((float)p * div);
        }
    }
    if (h->pe.ensure(pe.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(h->pe.p, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return sfail(nullptr, LS_EHIP, "pe upload failed");
    }
    *out = h;
    return LS_OK;
}

void ls_sag_destroy(ls_sag* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& kv : h->dw) kv.second.release();
    Buf* all[] = {&h->pe, &h->xin, &h->zin, &h->mask, &h->q, &h->qc, &h->qkv, &h->attn, &h->t1, &h->ca, &h->x2, &h->hid, &h->t3, &h->out,
                  &h->wcross, &h->bcross};
    for (Buf* b : all) b->release();
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ls_sag_set_weight(ls_sag* h, const char* key, const float* data, size_t n) {
    if (!h || !key || (!data && n)) return sfail(h, LS_EINVAL, "ls_sag_set_weight: null argument");
    const std::string k(key);
    if (k.size() >= 3 && k.compare(k.size() - 3, 3, ".pe") == 0) return LS_OK;
    h->w[k].assign(data, data + n);
    h->committed = false;
    return LS_OK;
}

int ls_sag_commit_weights(ls_sag* h) {
    if (!h) return LS_EINVAL;
    SCHK(h, hipSetDevice(h->cfg.device));
    const int D = kD, FF = h->cfg.ff_size, JF = h->JF;
    auto need = [&](const std::string& key, size_t want) -> int {
        auto it = h->w.find(key);
        if (it == h->w.end()) return sfail(h, LS_ESTATE, "missing weight '%s'", key.c_str());
        if (it->second.size() != want) return sfail(h, LS_EINVAL, "weight '%s' has %zu elements, expected %zu", key.c_str(), it->second.size(), want);
        Buf& b = h->dw[key];
        SCHK(h, b.ensure(want * sizeof(float)));
        SCHK(h, hipMemcpy(b.p, it->second.data(), want * sizeof(float), hipMemcpyHostToDevice));
        return LS_OK;
    };
    int rc;
    char key[160];
    for (int l = 0; l < h->cfg.num_layers; ++l) {       // nn.TransformerDecoderLayer parameters (motionclip_module.py:122-128)
        struct { const char* s; size_t n; } items[] = {
            {"self_attn.in_proj_weight", (size_t)3 * D * D}, {"self_attn.in_proj_bias", (size_t)3 * D},
            {"self_attn.out_proj.weight", (size_t)D * D}, {"self_attn.out_proj.bias", (size_t)D},
            {"multihead_attn.in_proj_weight", (size_t)3 * D * D}, {"multihead_attn.in_proj_bias", (size_t)3 * D},
            {"multihead_attn.out_proj.weight", (size_t)D * D}, {"multihead_attn.out_proj.bias", (size_t)D},
            {"linear1.weight", (size_t)FF * D}, {"linear1.bias", (size_t)FF}, {"linear2.weight", (size_t)D * FF}, {"linear2.bias", (size_t)D},
            {"norm1.weight", (size_t)D}, {"norm1.bias", (size_t)D}, {"norm2.weight", (size_t)D}, {"norm2.bias", (size_t)D},
            {"norm3.weight", (size_t)D}, {"norm3.bias", (size_t)D}};
        for (auto& it : items) {
            snprintf(key, sizeof key, "seqTransDecoder.layers.%d.%s", l, it.s);
            if ((rc = need(key, it.n)) != LS_OK) return rc;
        }
    }
    if ((rc = need("finallayer.weight", (size_t)JF * D)) != LS_OK) return rc;      // :131
    if ((rc = need("finallayer.bias", JF)) != LS_OK) return rc;
    if ((rc = need("mapping.weight", (size_t)D * (JF + 1))) != LS_OK) return rc;   // :133  Linear(28,512)
    if ((rc = need("mapping.bias", D)) != LS_OK) return rc;
    {   // Cross-attention to a memory of length 1 (the CLIP text feature): softmax over one key is 1, so every layer adds
        //     out_proj(v_proj(z)) = (W_out W_v) z + (W_out b_v + b_out)
        // to each of its rows -- a per-sample vector that depends on z only.  The products W_out W_v are formed here once (in double),
        // for all layers stacked as one [L*D][D] matrix, so a decode needs ONE small GEMM instead of two per layer.
        const int L = h->cfg.num_layers;
        std::vector<float> wc((size_t)L * D * D), bc((size_t)L * D);
        std::vector<double> row(D);
        for (int l = 0; l < L; ++l) {
            snprintf(key, sizeof key, "seqTransDecoder.layers.%d.", l);
            const std::string P(key);
            const float* Wv = h->w[P + "multihead_attn.in_proj_weight"].data() + (size_t)2 * D * D;     // rows 2D..3D of in_proj: v_proj
            const float* bv = h->w[P + "multihead_attn.in_proj_bias"].data() + 2 * D;
            const float* Wo = h->w[P + "multihead_attn.out_proj.weight"].data();
            const float* bo = h->w[P + "multihead_attn.out_proj.bias"].data();
            for (int i = 0; i < D; ++i) {
                std::fill(row.begin(), row.end(), 0.0);
                double bacc = bo[i];
                for (int j = 0; j < D; ++j) {
                    const double wij = Wo[(size_t)i * D + j];
                    const float* wvj = Wv + (size_t)j * D;
                    for (int k = 0; k < D; ++k) row[k] += wij * (double)wvj[k];
                    bacc += wij * (double)bv[j];
                }
                for (int k = 0; k < D; ++k) wc[((size_t)l * D + i) * D + k] = (float)row[k];
                bc[(size_t)l * D + i] = (float)bacc;
            }
        }
        SCHK(h, h->wcross.ensure(wc.size() * sizeof(float)));
        SCHK(h, h->bcross.ensure(bc.size() * sizeof(float)));
        SCHK(h, hipMemcpy(h->wcross.p, wc.data(), wc.size() * sizeof(float), hipMemcpyHostToDevice));
        SCHK(h, hipMemcpy(h->bcross.p, bc.data(), bc.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    h->committed = true;
    return LS_OK;
}

static int sag_decode_impl(ls_sag* h, int batch, int on_device, const float* x, const float* z, const unsigned char* mask, float* out, bool wait) {
    if (!h || !x || !z || !out) return sfail(h, LS_EINVAL, "ls_sag_decode: null argument");
    if (!h->committed) return sfail(h, LS_ESTATE, "ls_sag_decode before ls_sag_commit_weights");
    if (batch < 1) return sfail(h, LS_EINVAL, "batch must be >= 1");
    SCHK(h, hipSetDevice(h->cfg.device));
    const int B = batch, D = kD, FF = h->cfg.ff_size, JF = h->JF, M = B * kT, H = h->cfg.num_heads;
    hipStream_t st = h->stream;
    const hipMemcpyKind in = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const size_t nx = (size_t)B * JF * kT * sizeof(float);
    SCHK(h, h->xin.ensure(nx)); SCHK(h, h->zin.ensure((size_t)B * D * sizeof(float))); SCHK(h, h->out.ensure(nx));
    SCHK(h, hipMemcpyAsync(h->xin.p, x, nx, in, st));
    SCHK(h, hipMemcpyAsync(h->zin.p, z, (size_t)B * D * sizeof(float), in, st));
    const unsigned char* dmask = nullptr;
    if (mask) {
        SCHK(h, h->mask.ensure((size_t)M));
        SCHK(h, hipMemcpyAsync(h->mask.p, mask, (size_t)M, in, st));
        dmask = static_cast<const unsigned char*>(h->mask.p);
    }
    const size_t nm = (size_t)M * D * sizeof(float);
    SCHK(h, h->q.ensure(nm)); SCHK(h, h->qkv.ensure(3 * (nm + (size_t)128 * D * sizeof(float))));     /* + the compact first layer's tile padding */ SCHK(h, h->attn.ensure(nm)); SCHK(h, h->t1.ensure(nm));
    SCHK(h, h->x2.ensure(nm)); SCHK(h, h->t3.ensure(nm));
    SCHK(h, h->hid.ensure((size_t)M * FF * sizeof(float)));
    const int LD = h->cfg.num_layers * D;
    SCHK(h, h->ca.ensure((size_t)B * LD * sizeof(float)));
    auto W = [&](const std::string& k) { return h->dw[k].f(); };
    SCHK(h, hipEventRecord(h->ev[0], st));
    // cross-attention terms of all layers: ca[b][l*D + i] = (W_out_l W_v_l) z_b + (W_out_l b_v_l + b_out_l)
    SCHK(h, launch_gemm_nt(h->zin.f(), D, h->wcross.f(), D, h->bcross.f(), nullptr, 0, h->ca.f(), LD, B, LD, D, 0, st));
    // First layer: the query rows of frames f >= n_pre are b_map + pe[f] for EVERY sample, so their q / k / v projections are too.  Its
    // packed in_proj runs over the distinct rows only (B * n_pre + T - n_pre instead of B * T: 12 % of them at T = 34, n_pre = 4),
    // padded to whole 128-row tiles so that it stays on the GEMM's full-tile path; the attention kernel maps (sample, frame) to them.
    const int npre = h->cfg.n_pre_poses;
    const bool compact = npre > 0 && npre < kT;
    const int Mc = compact ? (B * npre + (kT - npre) + 127) / 128 * 128 : 0;
    if (compact) {
        const void* old = h->qc.p;
        SCHK(h, h->qc.ensure((size_t)Mc * D * sizeof(float)));
        if (old != h->qc.p) SCHK(h, hipMemsetAsync(h->qc.p, 0, (size_t)Mc * D * sizeof(float), st));       // pad rows: finite inputs
    }
    SCHK(h, launch_sag_queries(h->xin.f(), W("mapping.weight"), W("mapping.bias"), h->pe.f(), h->q.f(), compact ? h->qc.f() : nullptr, B, JF, npre, D, st));
    float* xcur = h->q.f();
    char pre[96];
    for (int l = 0; l < h->cfg.num_layers; ++l) {
        snprintf(pre, sizeof pre, "seqTransDecoder.layers.%d.", l);
        const std::string P(pre);
        // self-attention block: x = norm1(x + out_proj(softmax(q k^T / sqrt(128)) v))
        const bool lc = compact && l == 0;
        SCHK(h, launch_gemm_nt(lc ? h->qc.f() : xcur, D, W(P + "self_attn.in_proj_weight"), D, W(P + "self_attn.in_proj_bias"), nullptr, 0, h->qkv.f(), 3 * D,
                               lc ? Mc : M, 3 * D, D, 0, st));
        SCHK(h, launch_sag_attention(h->qkv.f(), h->attn.f(), B, H, D, lc ? npre : 0, st));
        SCHK(h, launch_gemm_nt(h->attn.f(), D, W(P + "self_attn.out_proj.weight"), D, W(P + "self_attn.out_proj.bias"), xcur, D, h->t1.f(), D, M, D, D, 0, st));
        // ... norm1, then the cross-attention block x = norm2(x + ca_l[b]) (the per-sample vector computed above), in one pass
        SCHK(h, launch_layernorm512x2(h->t1.f(), W(P + "norm1.weight"), W(P + "norm1.bias"), h->ca.f() + (size_t)l * D, LD, W(P + "norm2.weight"),
                                      W(P + "norm2.bias"), h->x2.f(), M, st));
        // feed-forward: x = norm3(x + linear2(gelu(linear1(x))))
        SCHK(h, launch_gemm_nt(h->x2.f(), D, W(P + "linear1.weight"), D, W(P + "linear1.bias"), nullptr, 0, h->hid.f(), FF, M, FF, D, 3, st));
        SCHK(h, launch_gemm_nt(h->hid.f(), FF, W(P + "linear2.weight"), FF, W(P + "linear2.bias"), h->x2.f(), D, h->t3.f(), D, M, D, FF, 0, st));
        SCHK(h, launch_layernorm512(h->t3.f(), nullptr, 0, W(P + "norm3.weight"), W(P + "norm3.bias"), h->q.f(), M, st));
        xcur = h->q.f();
    }
    SCHK(h, launch_sag_final(xcur, W("finallayer.weight"), W("finallayer.bias"), dmask, h->out.f(), B, JF, D, st));
    SCHK(h, hipEventRecord(h->ev[1], st));
    SCHK(h, hipMemcpyAsync(out, h->out.p, nx, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (!wait) { h->pending_ms = true; return LS_OK; }      // the caller orders consumers behind ls_sag_stream (ls_stream_order)
    SCHK(h, hipStreamSynchronize(st));
    SCHK(h, hipEventElapsedTime(&h->last_ms, h->ev[0], h->ev[1]));
    h->pending_ms = false;
    return LS_OK;
}

int ls_sag_decode(ls_sag* h, int batch, int on_device, const float* x, const float* z, const unsigned char* mask, float* out) {
    return sag_decode_impl(h, batch, on_device, x, z, mask, out, true);
}

// The same decode, enqueued only (device pointers): returns without waiting for the GPU, so a caller that iterates batches can decode
// batch n + 1 on this handle's stream while batch n is refined on another handle's (scripts/test_LivelySpeaker_ted.py:57-113 runs them
// back to back).  `out` is complete once ls_sag_stream() has reached this point: order its consumers with ls_stream_order.  The
// handle's staging buffers are reused by the next decode on the same (in-order) stream.
int ls_sag_decode_async(ls_sag* h, int batch, const float* x, const float* z, const unsigned char* mask, float* out) {
    return sag_decode_impl(h, batch, 1, x, z, mask, out, false);
}

float ls_sag_last_decode_ms(const ls_sag* h) {
    if (!h) return -1.f;
    if (h->pending_ms) {           // an asynchronous decode: its span is read once it has finished (waits for it)
        ls_sag* m = const_cast<ls_sag*>(h);
        if (hipEventSynchronize(m->ev[1]) == hipSuccess && hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]) == hipSuccess) m->pending_ms = false;
    }
    return h->last_ms;
}

void* ls_sag_stream(const ls_sag* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

}  // extern "C"
