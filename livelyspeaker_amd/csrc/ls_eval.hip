// FGD feature extractor: PoseEncoderConv.forward in eval mode (scripts/model/embedding_net.py:41-83,
// BEAT scripts_beat/model/motion_autoencoder.py:38-73) behind ls_eval_*.  BatchNorm (eval: running statistics) is folded
// into the preceding conv / linear weights when the weights are committed; nn.LeakyReLU(True) in out_net is
// negative_slope = 1.0, i.e. the identity, so out_net + fc_mu are four plain linears (run on the strided MFMA GEMM).
// The convs are tiny (1.1 MFLOP per clip): a direct kernel, one thread per output element.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/ls_hip.h"
#include "ls_internal.h"
#include "ls_train.h"

namespace ls {

// out[b][co][p] = lrelu( bias[co] + sum_{ci,k} w[co][ci][k] * in(b, ci, p*stride + k) ), in(b,c,l) = in[b*sb + c*sc + l*sl]
__global__ void k_conv_small(const float* __restrict__ in, long long sb, long long sc, long long sl, const float* __restrict__ w,
                             const float* __restrict__ bias, float* __restrict__ out, int Cin, int Cout, int K, int stride, int Lout,
                             float slope, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % Lout);
    const int co = (int)((i / Lout) % Cout);
    const size_t b = i / ((size_t)Lout * Cout);
    const float* ib = in + b * sb + (size_t)(p * stride) * sl;
    const float* wr = w + (size_t)co * Cin * K;
    float acc = bias[co];
    for (int ci = 0; ci < Cin; ++ci)
        for (int k = 0; k < K; ++k) acc = fmaf(wr[ci * K + k], ib[(size_t)ci * sc + (size_t)k * sl], acc);
    out[i] = acc >= 0.f ? acc : slope * acc;
}

}  // namespace ls

using namespace ls;

struct ls_eval {
    ls_eval_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, std::vector<float>> w;
    bool committed = false;
    float* cw[4] = {nullptr, nullptr, nullptr, nullptr};
    float* cb[4] = {nullptr, nullptr, nullptr, nullptr};
    float* lw[4] = {nullptr, nullptr, nullptr, nullptr};
    float* lb[4] = {nullptr, nullptr, nullptr, nullptr};
    float *in = nullptr, *a[4] = {nullptr, nullptr, nullptr, nullptr}, *h[4] = {nullptr, nullptr, nullptr, nullptr};
    int capB = 0;
};

namespace {

std::string g_eval_create_error;

int efail(ls_eval* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_eval_create_error = buf;
    return code;
}

#define EHIP(h, expr)                                                                                                     \
    do {                                                                                                                  \
        hipError_t e__ = (expr);                                                                                          \
        if (e__ != hipSuccess) return efail((h), LS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

struct ConvShape { int cin, cout, k, stride, lin, lout; };

void shapes(const ls_eval_config& c, ConvShape (&cs)[4], int (&lin)[4][2]) {
    const int b = c.base;
    const int spec[4][4] = {{c.pose_dim, b, 3, 1}, {b, 2 * b, 3, 1}, {2 * b, 2 * b, 4, 2}, {2 * b, b, 3, 1}};
    int L = c.n_frames;
    for (int i = 0; i < 4; ++i) {
        cs[i] = {spec[i][0], spec[i][1], spec[i][2], spec[i][3], L, (L - spec[i][2]) / spec[i][3] + 1};
        L = cs[i].lout;
    }
    const int flat = cs[3].cout * cs[3].lout;
    const int dims[5] = {flat, c.hidden1, c.hidden2, b, b};
    for (int i = 0; i < 4; ++i) { lin[i][0] = dims[i]; lin[i][1] = dims[i + 1]; }
}

int upload(ls_eval* h, float*& dst, const std::vector<float>& v) {
    if (dst) { EHIP(h, hipFree(dst)); dst = nullptr; }
    EHIP(h, hipMalloc(reinterpret_cast<void**>(&dst), v.size() * 4));
    EHIP(h, hipMemcpy(dst, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return LS_OK;
}

}  // namespace

extern "C" {

int ls_eval_create(const ls_eval_config* cfg, ls_eval** out) {
    if (!cfg || !out) return efail(nullptr, LS_EINVAL, "ls_eval_create: null argument");
    if (cfg->pose_dim <= 0 || cfg->base <= 0 || cfg->hidden1 <= 0 || cfg->hidden2 <= 0 || cfg->n_frames < 12)
        return efail(nullptr, LS_EINVAL, "ls_eval_create: bad configuration");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return efail(nullptr, LS_EHIP, "ls_eval_create: no HIP device (no CPU path exists)");
    if (cfg->device < 0 || cfg->device >= n) return efail(nullptr, LS_EINVAL, "ls_eval_create: device %d out of range", cfg->device);
    ls_eval* h = new ls_eval();
    h->cfg = *cfg;
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return efail(nullptr, LS_EHIP, "ls_eval_create: stream creation failed");
    }
    *out = h;
    return LS_OK;
}

void ls_eval_destroy(ls_eval* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 4; ++i)
        for (float* p : {h->cw[i], h->cb[i], h->lw[i], h->lb[i], h->a[i], h->h[i]}) if (p) (void)hipFree(p);
    if (h->in) (void)hipFree(h->in);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* ls_eval_last_error(const ls_eval* h) { return h ? h->err.c_str() : g_eval_create_error.c_str(); }

int ls_eval_set_weight(ls_eval* h, const char* key, const float* data, size_t n) {
    if (!h || !key || !data) return efail(h, LS_EINVAL, "ls_eval_set_weight: null argument");
    const std::string k(key);
    // the auto-encoder checkpoint also holds the decoder and fc_logvar: accepted and ignored (never on this path)
    if (k.rfind("decoder.", 0) == 0 || k.find("fc_logvar") != std::string::npos || k.find("num_batches_tracked") != std::string::npos) return LS_OK;
    if (k.rfind("pose_encoder.", 0) != 0) return efail(h, LS_EINVAL, "ls_eval_set_weight: unexpected key '%s'", key);
    h->w[k].assign(data, data + n);
    h->committed = false;
    return LS_OK;
}

int ls_eval_commit_weights(ls_eval* h) {
    if (!h) return LS_EINVAL;
    EHIP(h, hipSetDevice(h->cfg.device));
    ConvShape cs[4];
    int lin[4][2];
    shapes(h->cfg, cs, lin);
    auto get = [&](const std::string& k, size_t n, const std::vector<float>** out) -> int {
        auto it = h->w.find(k);
        if (it == h->w.end()) return efail(h, LS_ESTATE, "ls_eval_commit_weights: missing key '%s'", k.c_str());
        if (it->second.size() != n) return efail(h, LS_EINVAL, "ls_eval_commit_weights: '%s' has %zu elements, expected %zu", k.c_str(), it->second.size(), n);
        *out = &it->second;
        return LS_OK;
    };
    // fold y = (z - rm) / sqrt(rv + 1e-5) * gamma + beta into the producer of z (rows of W scaled, bias shifted)
    auto fold = [&](const std::string& bn, int nout, size_t per_row, std::vector<float>& W, std::vector<float>& Bv) -> int {
        const std::vector<float>*g, *be, *rm, *rv;
        int rc;
        if ((rc = get(bn + "weight", nout, &g)) || (rc = get(bn + "bias", nout, &be)) || (rc = get(bn + "running_mean", nout, &rm)) ||
            (rc = get(bn + "running_var", nout, &rv)))
            return rc;
        for (int o = 0; o < nout; ++o) {
            const float s = (*g)[o] / sqrtf((*rv)[o] + 1e-5f);
            for (size_t j = 0; j < per_row; ++j) W[(size_t)o * per_row + j] *= s;
            Bv[o] = (Bv[o] - (*rm)[o]) * s + (*be)[o];
        }
        return LS_OK;
    };
    int rc;
    for (int i = 0; i < 4; ++i) {
        const std::string p = "pose_encoder.net." + std::to_string(i) + (i < 3 ? ".0." : ".");
        const std::vector<float>*w, *b;
        const size_t per = (size_t)cs[i].cin * cs[i].k;
        if ((rc = get(p + "weight", per * cs[i].cout, &w)) || (rc = get(p + "bias", cs[i].cout, &b))) return rc;
        std::vector<float> W(*w), Bv(*b);
        if (i < 3 && (rc = fold("pose_encoder.net." + std::to_string(i) + ".1.", cs[i].cout, per, W, Bv))) return rc;
        if ((rc = upload(h, h->cw[i], W)) || (rc = upload(h, h->cb[i], Bv))) return rc;
    }
    const char* lk[4] = {"pose_encoder.out_net.0.", "pose_encoder.out_net.3.", "pose_encoder.out_net.6.", "pose_encoder.fc_mu."};
    const char* bk[4] = {"pose_encoder.out_net.1.", "pose_encoder.out_net.4.", nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
        const std::vector<float>*w, *b;
        if ((rc = get(std::string(lk[i]) + "weight", (size_t)lin[i][0] * lin[i][1], &w)) || (rc = get(std::string(lk[i]) + "bias", lin[i][1], &b))) return rc;
        std::vector<float> W(*w), Bv(*b);
        if (bk[i] && (rc = fold(bk[i], lin[i][1], lin[i][0], W, Bv))) return rc;
        if ((rc = upload(h, h->lw[i], W)) || (rc = upload(h, h->lb[i], Bv))) return rc;
    }
    h->committed = true;
    return LS_OK;
}

int ls_eval_features(ls_eval* h, int batch, int on_device, const float* poses, float* feat) {
    if (!h || !poses || !feat) return efail(h, LS_EINVAL, "ls_eval_features: null argument");
    if (!h->committed) return efail(h, LS_ESTATE, "ls_eval_features: weights are not committed");
    if (batch <= 0) return efail(h, LS_EINVAL, "ls_eval_features: batch must be positive");
    EHIP(h, hipSetDevice(h->cfg.device));
    ConvShape cs[4];
    int lin[4][2];
    shapes(h->cfg, cs, lin);
    const int B = batch, T = h->cfg.n_frames, D = h->cfg.pose_dim;
    if (B > h->capB) {
        for (int i = 0; i < 4; ++i) {
            if (h->a[i]) EHIP(h, hipFree(h->a[i]));
            if (h->h[i]) EHIP(h, hipFree(h->h[i]));
            h->a[i] = h->h[i] = nullptr;
            EHIP(h, hipMalloc(reinterpret_cast<void**>(&h->a[i]), (size_t)B * cs[i].cout * cs[i].lout * 4));
            EHIP(h, hipMalloc(reinterpret_cast<void**>(&h->h[i]), (size_t)B * lin[i][1] * 4));
        }
        if (h->in) EHIP(h, hipFree(h->in));
        h->in = nullptr;
        EHIP(h, hipMalloc(reinterpret_cast<void**>(&h->in), (size_t)B * T * D * 4));
        h->capB = B;
    }
    hipStream_t st = h->stream;
    EHIP(h, hipMemcpyAsync(h->in, poses, (size_t)B * T * D * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    for (int i = 0; i < 4; ++i) {
        // layer 0 reads poses [B][T][D] as (b, c = d, l = t): poses.transpose(1, 2) (embedding_net.py:71)
        const float* src = i == 0 ? h->in : h->a[i - 1];
        const long long sb = i == 0 ? (long long)T * D : (long long)cs[i].cin * cs[i].lin;
        const long long sc = i == 0 ? 1 : cs[i].lin, sl = i == 0 ? D : 1;
        const size_t total = (size_t)B * cs[i].cout * cs[i].lout;
        hipLaunchKernelGGL(k_conv_small, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, sb, sc, sl, h->cw[i], h->cb[i], h->a[i],
                           cs[i].cin, cs[i].cout, cs[i].k, cs[i].stride, cs[i].lout, i < 3 ? 0.2f : 1.0f, total);
        EHIP(h, hipGetLastError());
    }
    const float* x = h->a[3];                     // flatten(1): [B][base][12] is already [B][12*base]
    for (int i = 0; i < 4; ++i) {
        GemmArgs a{};
        a.A = op_rows(x, lin[i][0], B, lin[i][0]);
        a.B = op_rows(h->lw[i], lin[i][0], lin[i][1], lin[i][0]);
        a.C = h->h[i];
        a.cri = INT_MAX; a.cro = 0; a.crs = lin[i][1]; a.cns = 1;
        a.bias = h->lb[i];
        a.M = B; a.N = lin[i][1]; a.K = lin[i][0];
        EHIP(h, launch_gemm_tr(a, true, true, 1, st));
        x = h->h[i];
    }
    EHIP(h, hipMemcpyAsync(feat, h->h[3], (size_t)B * h->cfg.base * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    EHIP(h, hipStreamSynchronize(st));
    return LS_OK;
}

void* ls_eval_stream(const ls_eval* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

}  // extern "C"
