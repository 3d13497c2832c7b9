// torch's CPU normal stream, restated natively: the host draws of the "identical seeds" mode (noise_source = 'torch_cpu') without
// torch's generator on the critical path.
//
// Contract being reproduced: the reference draws every normal of a sampling loop from torch's global CPU generator
// (scripts/diffusion/gaussian_diffusion.py:700-743: x_T, then per step randn_like(x); scripts/model/RAG.py:10-13, 120: randn_like of the
// style token in both CFG passes), so `torch.manual_seed(s)` fixes the sample.  torch (2.x, ATen/native/cpu/DistributionTemplates.h,
// ATen/core/DistributionsHelper.h, ATen/core/MT19937RNGEngine.h) makes those draws two ways:
//   * contiguous float tensor of >= 16 elements (randn(B,1,512), randn(*shape)): n 24-bit uniforms (one mt19937 word each), then
//     Box-Muller on 16-blocks -- element j pairs with j + 8: r = sqrt(-2 log(1 - u[j])), theta = 2 pi u[j+8] -- in FLOAT arithmetic
//     (std::log / cos / sin of float in torch's DEFAULT kernel; Cephes polynomials with compiler-contracted FMAs in its AVX2 / AVX512
//     kernels -- both restated below, the Python side picks the variant that reproduces torch on this machine, or keeps torch's generator);
//     a tail that is not a multiple of 16 re-draws the LAST 16 elements;
//   * anything else (randn_like of the model-output-shaped view whose memory order is [T][B][J][F]): one element at a time in MEMORY
//     order through normal_distribution<double>: two 53-bit uniforms (two mt19937 words each, first word = high half),
//     r = sqrt(-2 log1p(-u2)), theta = 2 pi u1, the cos branch is returned and the sin branch cached in the generator for the next
//     element (also across calls).
// The generator state travels as torch.get_rng_state()'s 5056-byte blob (layout probed in tests/test_torch_rng.py), updated in place,
// so torch.set_rng_state() leaves torch's generator exactly where the reference's draws would have left it.
//
// Speed: the mt19937 words are produced sequentially (they must be), the transcendental part is spread over worker threads.
#include "ls_hip.h"

// every product and sum below rounds where it is written: no contraction by THIS compiler (the FMAs of the restated kernels are explicit)
#pragma clang fp contract(off)

#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int kN = 624, kM = 397;
constexpr size_t kStateBytes = 5056;
constexpr size_t kOffLeft = 8, kOffSeeded = 12, kOffNext = 16, kOffState = 24, kOffNormalY = 5024, kOffNormalValid = 5040;

struct Mt {
    uint32_t st[kN];
    int left;
    uint32_t next;
    double cached;
    int cached_valid;

    bool load(const uint8_t* blob) {
        int32_t l, seeded;
        uint64_t nx;
        memcpy(&l, blob + kOffLeft, 4);
        memcpy(&seeded, blob + kOffSeeded, 4);
        memcpy(&nx, blob + kOffNext, 8);
        if (!seeded || l <= 0 || l > kN || nx > (uint64_t)kN) return false;
        left = l; next = (uint32_t)nx;
        for (int i = 0; i < kN; ++i) { uint64_t v; memcpy(&v, blob + kOffState + 8 * (size_t)i, 8); st[i] = (uint32_t)v; }
        memcpy(&cached, blob + kOffNormalY, 8);
        int32_t cv;
        memcpy(&cv, blob + kOffNormalValid, 4);
        cached_valid = cv;
        return true;
    }
    void store(uint8_t* blob) const {
        int32_t l = left;
        uint64_t nx = next;
        memcpy(blob + kOffLeft, &l, 4);
        memcpy(blob + kOffNext, &nx, 8);
        for (int i = 0; i < kN; ++i) { uint64_t v = st[i]; memcpy(blob + kOffState + 8 * (size_t)i, &v, 8); }
        const double c = cached_valid ? cached : 0.0;
        memcpy(blob + kOffNormalY, &c, 8);
        int32_t cv = cached_valid ? 1 : 0;
        memcpy(blob + kOffNormalValid, &cv, 4);
    }
    static uint32_t twist(uint32_t u, uint32_t v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); }
    void next_state() {
        uint32_t* p = st;
        left = kN; next = 0;
        for (int j = kN - kM + 1; --j; p++) *p = p[kM] ^ twist(p[0], p[1]);
        for (int j = kM; --j; p++) *p = p[kM - kN] ^ twist(p[0], p[1]);
        *p = p[kM - kN] ^ twist(p[0], st[0]);
    }
    uint32_t word() {
        if (--left == 0) next_state();
        uint32_t y = st[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    float uf() { return (float)(word() & ((1u << 24) - 1)) * (1.0f / (float)(1u << 24)); }                 // uniform_real<float>
    double ud() {                                                                                           // uniform_real<double>: random64
        const uint64_t hi = word(), lo = word();
        return (double)(((hi << 32) | lo) & ((1ull << 53) - 1)) * (1.0 / (double)(1ull << 53));
    }
};

// ---- the float transform of the contiguous path -------------------------------------------------------------------------------
// variant 0: normal_fill_16<float> as written (std::log / std::cos / std::sin of float): torch's DEFAULT-capability kernel.
// variants 1..4: normal_fill_16_AVX2, which torch's AVX2 AND AVX512 kernels use: Cephes' single-precision log and sincos in Julien
// Pommier's formulation (public: sse_mathfun / avx_mathfun, zlib licence), restated per lane here with the multiply-adds contracted to
// FMAs the way torch's compiler (GCC, -ffp-contract=fast) did.  Where a sum has two products to choose from the contraction is the
// compiler's pick, hence the variants: bit 0 = the log's  p(x) x z + e q1, bit 1 = the cosine's  y z - 0.5 z  (set: the second product
// is the fused one).  The Python side finds the variant that reproduces torch bit for bit on this machine (or none).
inline float as_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t as_u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

#define LS_FMA_TARGET __attribute__((target("fma")))      // fmaf as one instruction (through libm it is a call per operation)
LS_FMA_TARGET inline float cephes_logf(float x, int variant) {
    const bool invalid = x <= 0.0f;
    x = std::fmax(x, as_f(0x00800000u));                    // cut off denormalized stuff
    int e_i = (int)(as_u(x) >> 23) - 0x7f;
    x = as_f((as_u(x) & ~0x7f800000u) | as_u(0.5f));         // keep only the fractional part
    float e = (float)e_i + 1.0f;
    const bool lt = x < 0.707106781186547524f;
    const float tmp0 = lt ? x : 0.0f;
    x = x - 1.0f;
    e = e - (lt ? 1.0f : 0.0f);
    x = x + tmp0;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = fmaf(y, x, -1.1514610310E-1f);
    y = fmaf(y, x, 1.1676998740E-1f);
    y = fmaf(y, x, -1.2420140846E-1f);
    y = fmaf(y, x, 1.4249322787E-1f);
    y = fmaf(y, x, -1.6668057665E-1f);
    y = fmaf(y, x, 2.0000714765E-1f);
    y = fmaf(y, x, -2.4999993993E-1f);
    y = fmaf(y, x, 3.3333331174E-1f);
    y = y * x;
    if (variant & 1) y = fmaf(e, -2.12194440e-4f, y * z);
    else y = fmaf(y, z, e * -2.12194440e-4f);
    y = fmaf(-z, 0.5f, y);
    x = fmaf(e, 0.693359375f, x + y);
    return invalid ? as_f(0xffffffffu) : x;
}

LS_FMA_TARGET inline void cephes_sincosf(float xin, int variant, float& s, float& c) {
    uint32_t sign_sin = as_u(xin) & 0x80000000u;
    float x = as_f(as_u(xin) & 0x7fffffffu);
    float y = x * 1.27323954473516f;                         // 4 / pi
    int j = (int)y;                                          // cvttps
    j = (j + 1) & ~1;
    y = (float)j;
    const uint32_t swap_sign_sin = ((uint32_t)(j & 4)) << 29;
    const bool poly = (j & 2) == 0;
    x = fmaf(y, -0.78515625f, x);
    x = fmaf(y, -2.4187564849853515625e-4f, x);
    x = fmaf(y, -3.77489497744594108e-8f, x);
    const uint32_t sign_cos = ((uint32_t)(~(j - 2) & 4)) << 29;
    sign_sin ^= swap_sign_sin;
    const float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = fmaf(yc, z, -1.388731625493765E-003f);
    yc = fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    if (variant & 2) yc = fmaf(-z, 0.5f, yc * z);
    else yc = fmaf(yc, z, -(z * 0.5f));
    yc = yc + 1.0f;
    float ys = -1.9515295891E-4f;
    ys = fmaf(ys, z, 8.3321608736E-3f);
    ys = fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = fmaf(ys, x, x);
    const float rs = poly ? ys : yc, rc = poly ? yc : ys;
    s = as_f(as_u(rs) ^ sign_sin);
    c = as_f(as_u(rc) ^ sign_cos);
}

LS_FMA_TARGET void fill16_cephes(float* d, int variant) {
    const float two_pi = 2.0f * 3.14159265358979323846;
    for (int j = 0; j < 8; ++j) {
        const float u1 = 1.0f - d[j];
        const float u2 = d[j + 8];
        const float radius = std::sqrt(-2.0f * cephes_logf(u1, variant - 1));
        const float theta = two_pi * u2;
        float sn, cs;
        cephes_sincosf(theta, variant - 1, sn, cs);
        d[j] = fmaf(radius * cs, 1.0f, 0.0f);
        d[j + 8] = fmaf(radius * sn, 1.0f, 0.0f);
    }
}

// in place on 16 uniforms
void fill16(float* d, int variant) {
    if (variant == 0) {
        for (int j = 0; j < 8; ++j) {
            const float u1 = 1 - d[j];
            const float u2 = d[j + 8];
            const float radius = std::sqrt(-2 * std::log(u1));
            const float theta = 2.0f * 3.14159265358979323846 * u2;
            d[j] = radius * std::cos(theta) * 1.0f + 0.0f;
            d[j + 8] = radius * std::sin(theta) * 1.0f + 0.0f;
        }
        return;
    }
    fill16_cephes(d, variant);
}

void parallel_for(size_t n, int threads, const std::function<void(size_t, size_t)>& fn);

}  // namespace

#include <functional>

namespace {

void parallel_for(size_t n, int threads, const std::function<void(size_t, size_t)>& fn) {
    if (threads <= 1 || n < 4096) { fn(0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const size_t a = (size_t)t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        th.emplace_back(fn, a, b);
    }
    for (auto& x : th) x.join();
}

// one deferred piece of transcendental work: a contiguous float draw (uniforms already in place) or a serial double draw
struct ContigJob { float* out; size_t n; std::vector<float> tail; bool has_tail; int variant; };
struct SerialJob {
    float* out;                      // contiguous [B][J][F][T]
    int B, J, F, T;
    bool permuted;                   // true: element e in MEMORY order [T][B][J][F]; false: e in [B][J][F][T] order (fewer than 16 elements)
    size_t n;
    bool lead_cached; double lead;   // element 0 comes from the generator's cached sample
    std::vector<double> u;           // (u1, u2) per pair for elements lead_cached .. n
};

size_t dst_index(const SerialJob& j, size_t e) {
    if (!j.permuted) return e;
    const size_t bjf = (size_t)j.B * j.J * j.F, t = e / bjf, r = e - t * bjf;       // r = (b * J + j) * F + f
    return r * j.T + t;
}

void gen_contig(Mt& g, float* out, size_t n, ContigJob& job, int variant) {
    job.variant = variant;
    for (size_t i = 0; i < n; ++i) out[i] = g.uf();
    job.out = out; job.n = n; job.has_tail = (n % 16) != 0;
    if (job.has_tail) { job.tail.resize(16); for (int i = 0; i < 16; ++i) job.tail[i] = g.uf(); }
}
void run_contig(ContigJob& job, int threads) {
    const size_t blocks = job.n / 16;
    parallel_for(blocks, threads, [&](size_t a, size_t b) { for (size_t i = a; i < b; ++i) fill16(job.out + 16 * i, job.variant); });
    if (job.has_tail) {              // "recompute the last 16 values": they overlap the last full block
        fill16(job.tail.data(), job.variant);
        memcpy(job.out + job.n - 16, job.tail.data(), 16 * sizeof(float));
    }
}

void gen_serial(Mt& g, SerialJob& j) {
    j.lead_cached = false;
    size_t e = 0;
    if (g.cached_valid && j.n > 0) { j.lead_cached = true; j.lead = g.cached; g.cached_valid = 0; e = 1; }
    const size_t pairs = (j.n - e + 1) / 2;
    j.u.resize(2 * pairs);
    for (size_t p = 0; p < pairs; ++p) { j.u[2 * p] = g.ud(); j.u[2 * p + 1] = g.ud(); }
}
// returns the sample left over for the generator's cache when the element count is odd
void run_serial(SerialJob& j, int threads, Mt& g) {
    const size_t e0 = j.lead_cached ? 1 : 0, pairs = j.u.size() / 2;
    if (j.lead_cached) j.out[dst_index(j, 0)] = (float)(j.lead * 1.0 + 0.0);
    double leftover = 0.0;
    bool has_left = false;
    parallel_for(pairs, threads, [&](size_t a, size_t b) {
        for (size_t p = a; p < b; ++p) {
            const double u1 = j.u[2 * p], u2 = j.u[2 * p + 1];
            const double r = ::sqrt(-2.0 * ::log1p(-u2));
            const double theta = 2.0 * 3.14159265358979323846 * u1;
            const double zs = r * ::sin(theta), zc = r * ::cos(theta);
            const size_t e = e0 + 2 * p;
            j.out[dst_index(j, e)] = (float)(zc * 1.0 + 0.0);
            if (e + 1 < j.n) j.out[dst_index(j, e + 1)] = (float)(zs * 1.0 + 0.0);
            else { leftover = zs; has_left = true; }             // only the last pair of the job can get here
        }
    });
    if (has_left) { g.cached = leftover; g.cached_valid = 1; }
}

}  // namespace

extern "C" {

// torch.randn(n) (contiguous float32) / torch.randn_like of a [T][B][J][F]-memory-order view, from the state blob; see ls_hip.h
int ls_trng_randn(uint8_t* state, size_t state_bytes, float* out, size_t n, int variant, int n_threads) {
    if (!state || state_bytes != kStateBytes || (!out && n) || variant < 0 || variant > 4) return LS_EINVAL;
    if (variant > 0 && !__builtin_cpu_supports("fma")) return LS_EUNSUPPORTED;
    Mt g;
    if (!g.load(state)) return LS_EINVAL;
    if (n >= 16) {
        ContigJob job;
        gen_contig(g, out, n, job, variant);
        run_contig(job, n_threads);
    } else {
        SerialJob j{out, 1, 1, 1, (int)n, false, n, false, 0.0, {}};
        gen_serial(g, j);
        run_serial(j, 1, g);
    }
    g.store(state);
    return LS_OK;
}

int ls_trng_fill_steps(uint8_t* state, size_t state_bytes, int B, int D, int J, int F, int T, int n_steps, int first_contiguous, float* eps,
                       float* noise, int variant, int n_threads) {
    if (variant < 0 || variant > 4) return LS_EINVAL;
    if (variant > 0 && !__builtin_cpu_supports("fma")) return LS_EUNSUPPORTED;
    if (!state || state_bytes != kStateBytes || !eps || !noise || B < 1 || D < 1 || J < 1 || F < 1 || T < 1 || n_steps < 0) return LS_EINVAL;
    Mt g;
    if (!g.load(state)) return LS_EINVAL;
    const size_t ne = (size_t)B * D, nx = (size_t)B * J * F * T;
    // One step at a time: the mt19937 words sequentially, then that step's transcendental work on the worker threads.  (Generating a
    // whole segment's words first would need the cached-sample hand-over between steps before the transforms have run.)
    for (int k = 0; k < n_steps; ++k) {
        float* ec = eps + (size_t)(2 * k) * ne;
        float* eu = ec + ne;
        float* nz = noise + (size_t)k * nx;
        ContigJob c0, c1, c2;
        SerialJob s{nz, B, J, F, T, true, nx, false, 0.0, {}};
        const bool contig_noise = (k == 0 && first_contiguous) && nx >= 16;
        if (ne >= 16) { gen_contig(g, ec, ne, c0, variant); gen_contig(g, eu, ne, c1, variant); }
        else {          // never the case for D = 512; kept exact anyway
            for (float* p : {ec, eu}) { SerialJob t{p, 1, 1, 1, (int)ne, false, ne, false, 0.0, {}}; gen_serial(g, t); run_serial(t, 1, g); }
        }
        if (contig_noise) gen_contig(g, nz, nx, c2, variant);
        else {
            if (k == 0 && first_contiguous) s.permuted = false;          // fewer than 16 elements: serial, in [B][J][F][T] order
            gen_serial(g, s);
        }
        if (ne >= 16) { run_contig(c0, n_threads); run_contig(c1, n_threads); }
        if (contig_noise) run_contig(c2, n_threads);
        else run_serial(s, n_threads, g);
    }
    g.store(state);
    return LS_OK;
}

}  // extern "C"
